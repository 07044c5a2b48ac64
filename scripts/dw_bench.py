"""Time the depthwise 3x3 kernel on the HarDNet shapes (stride-4 maps of an 800x1333 image).

    python scripts/dw_bench.py [H W [N]]

Prints microseconds and the algorithmic HBM rate (read C + write C floats per pixel) per shape; the
"slice" rows write into a channel slice of a wider buffer, as the zero-copy concat does.
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_object_detection_amd import hip_ops  # noqa: E402

dev = torch.device("cuda:0")
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (200, 334)
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1


def bench(C, out_pitch, stride, reps=20):
    x = torch.randn(N, H, W, C, device=dev)
    w = torch.randn(3, 3, C, device=dev)
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = torch.zeros(N, OH, OW, out_pitch, device=dev)
    flush = torch.empty(96 << 20, device=dev)            # 384 MB: evicts L2 + Infinity Cache between runs
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    ts = []
    for _ in range(reps):
        flush.zero_()
        e0.record()
        hip_ops.dwconv3x3_nhwc(x, w, sc, sh, stride=stride, out=out, out_off=0, C=C)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    t = ts[len(ts) // 2]
    byt = 4.0 * N * C * (H * W + OH * OW)
    print(f"C={C:5d} out_pitch={out_pitch:5d} stride={stride}  {t:8.1f} us  {byt / t / 1e6:6.2f} TB/s")


for C in (16, 20, 28, 64, 104, 160, 256, 320, 412, 640, 1024):
    bench(C, C, 1)
for C in (16, 64, 160):
    bench(C, 4 * C + 48, 1)
for C in (96, 320, 640, 1024):
    bench(C, C, 2)

# calibration: a plain device copy of the same bytes under the same cold-cache protocol
for C in (160, 640, 1024):
    x = torch.randn(1, H, W, C, device=dev)
    y = torch.empty_like(x)
    flush = torch.empty(96 << 20, device=dev)
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    ts = []
    for _ in range(20):
        flush.zero_()
        e0.record()
        y.copy_(x)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(f"copy C={C:5d}  {ts[10]:8.1f} us  {8.0 * x.numel() / ts[10] / 1e6:6.2f} TB/s")
