"""Per-workgroup timeline of one conv launch (diagnostic build with s_memrealtime stamps, 100 MHz ticks)."""
import sys, ctypes, torch
sys.path.insert(0, '/root/repo')
from two_stage_object_detection_amd import hip_ops, _ffi
dev = torch.device('cuda:0')
H, W, Cin, Cout, k, tile, split = [int(v) for v in sys.argv[1:8]]
x = torch.randn(1, H, W, Cin, device=dev)
w = torch.randn(Cout, k, k, Cin, device=dev) / (Cin * k * k) ** 0.5
res = torch.randn(1, H, W, Cout, device=dev)
L = _ffi.lib()
stamps = torch.zeros(8192 * 8, dtype=torch.int64, device=dev)
L.tsod_debug_set_stamps.argtypes = [ctypes.c_void_p]
for _ in range(3):
    hip_ops.conv2d_nhwc(x, w, pad=k // 2, tile=tile, split_k=split, residual=res, act=1, slope=0.25)
torch.cuda.synchronize()
L.tsod_debug_set_stamps(stamps.data_ptr())
stamps.zero_()
torch.cuda.synchronize()
hip_ops.conv2d_nhwc(x, w, pad=k // 2, tile=tile, split_k=split, residual=res, act=1, slope=0.25)
torch.cuda.synchronize()
s = stamps.view(-1, 8).cpu()
s = s[s[:, 0] > 0]
t0 = s[:, 0].min()
rel = (s[:, :5] - t0).float() / 100.0      # us
print('workgroups', s.shape[0], 'kernel span us', float(rel[:, 4].max()))
import numpy as np
r = rel.numpy()
order = np.argsort(r[:, 0])
for name, col in (('start', 0), ('prologue done', 1), ('kloop done', 2), ('stores issued', 3), ('stores drained', 4)):
    v = r[:, col]
    print(f"{name:16s} min {v.min():7.2f} p10 {np.percentile(v,10):7.2f} median {np.median(v):7.2f} p90 {np.percentile(v,90):7.2f} max {v.max():7.2f}")
d = r[:, 1] - r[:, 0]; print('prologue dur  median %.2f p90 %.2f' % (np.median(d), np.percentile(d, 90)))
d = r[:, 2] - r[:, 1]; print('k-loop dur    median %.2f p90 %.2f' % (np.median(d), np.percentile(d, 90)))
d = r[:, 3] - r[:, 2]; print('epilogue dur  median %.2f p90 %.2f' % (np.median(d), np.percentile(d, 90)))
d = r[:, 4] - r[:, 3]; print('store drain   median %.2f p90 %.2f' % (np.median(d), np.percentile(d, 90)))
d = r[:, 4] - r[:, 0]; print('WG lifetime   median %.2f p90 %.2f' % (np.median(d), np.percentile(d, 90)))
# concurrency over time
ev = sorted([(a, 1) for a in r[:, 0]] + [(b, -1) for b in r[:, 4]])
cur = 0; last = 0; hist = {}
for t, dlt in ev:
    hist[cur] = hist.get(cur, 0) + (t - last); last = t; cur += dlt
tot = sum(hist.values())
print('time-weighted resident WGs: mean %.0f' % (sum(k_ * v for k_, v in hist.items()) / tot))
