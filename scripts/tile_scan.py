"""TF/s of every conv tile variant on a few large-M shapes (where only the K loop matters).
    python scripts/tile_scan.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_object_detection_amd import hip_ops  # noqa: E402
from two_stage_object_detection_amd._ffi import TILE_IDS, TILE_NAMES  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


for (H, W, Cin, Cout, k) in ((100, 167, 128, 128, 3), (100, 167, 512, 128, 1), (50, 84, 256, 256, 3), (200, 334, 64, 64, 3)):
    x = torch.randn(B, H, W, Cin, device=dev)
    w = torch.randn(Cout, k, k, Cin, device=dev) / (Cin * k * k) ** 0.5
    fl = 2 * B * H * W * Cout * Cin * k * k
    row = []
    for t in TILE_IDS:
        ms = timeit(lambda: hip_ops.conv2d_nhwc(x, w, pad=k // 2, tile=t, split_k=1))
        row.append(f"{TILE_NAMES[t]}={fl / ms / 1e9:.0f}")
    print(f"B={B} {H}x{W} {Cin}->{Cout} k{k}: " + " ".join(row), flush=True)
