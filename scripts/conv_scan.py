"""Time one conv config over a list of M (rows) to expose wave-quantisation behaviour."""
import sys, torch
sys.path.insert(0, '/root/repo')
from two_stage_object_detection_amd import hip_ops
dev = torch.device('cuda:0')
Cin, Cout, k, tile = [int(v) for v in sys.argv[1:5]]
def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps
for W in [int(v) for v in sys.argv[5:]]:
    x = torch.randn(1, 64, W, Cin, device=dev)
    w = torch.randn(Cout, k, k, Cin, device=dev) / (Cin * k * k) ** 0.5
    M = 64 * W
    import os
    for split in [int(v) for v in os.environ.get("SPLITS", "1,-1").split(",")]:
        t = timeit(lambda: hip_ops.conv2d_nhwc(x, w, pad=k // 2, tile=tile, split_k=split))
        fl = 2 * M * Cout * Cin * k * k
        print(f"M={M:6d} tiles64={M//64:5d} split {split:2d}: {t*1e3:7.1f} us  {fl/t/1e9:6.1f} TF/s", flush=True)
