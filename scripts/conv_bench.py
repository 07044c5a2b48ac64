"""Per-layer conv microbenchmark: every (tile, split) candidate on representative ResNet-50 shapes."""
import sys, torch
sys.path.insert(0, '/root/repo')
from two_stage_object_detection_amd import hip_ops
from two_stage_object_detection_amd._ffi import TILE_NAMES
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
shapes = [  # name, H, W, Cin, Cout, k, stride
    ("l1.conv2 3x3 64->64", 200, 334, 64, 64, 3, 1),
    ("l1.conv3 1x1 64->256", 200, 334, 64, 256, 1, 1),
    ("l1.conv1 1x1 256->64", 200, 334, 256, 64, 1, 1),
    ("l2.conv2 3x3 128->128", 100, 167, 128, 128, 3, 1),
    ("l2.conv3 1x1 128->512", 100, 167, 128, 512, 1, 1),
    ("l3.conv2 3x3 256->256", 50, 84, 256, 256, 3, 1),
    ("l3.conv3 1x1 256->1024", 50, 84, 256, 1024, 1, 1),
    ("l4.conv2 3x3 512->512", 25, 42, 512, 512, 3, 1),
    ("l4.conv3 1x1 512->2048", 25, 42, 512, 2048, 1, 1),
]
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
for name, H, W, Cin, Cout, k, s in shapes:
    x = torch.randn(B, H, W, Cin, device=dev)
    w = torch.randn(Cout, k, k, Cin, device=dev) / (Cin * k * k) ** 0.5
    res = torch.randn(B, H, W, Cout, device=dev)
    flops = 2 * B * H * W * Cout * Cin * k * k
    out = []
    for tile in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11):
        for split in (1, -1, 2, 4, 8):
            K = Cin * k * k
            if split > 1 and (K // 32) // split < 2: continue
            if split > 1 and B * H * W > 20000: continue
            t = timeit(lambda: hip_ops.conv2d_nhwc(x, w, pad=k // 2, tile=tile, split_k=split, residual=res, act=1, slope=0.25))
            out.append((t, tile, split))
    out.sort()
    best = out[0]
    line = " ".join(f"{TILE_NAMES[t]}/s{sp}:{flops / tt / 1e9:5.1f}" for tt, t, sp in out[:6])
    print(f"{name:26s} M={B*H*W:7d} best {flops / best[0] / 1e9:6.1f} TF/s {best[0]*1e3:7.1f} us | {line}", flush=True)
