"""Time the per-image top-k kernel at the detector's sizes: python scripts/sort_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_object_detection_amd import hip_ops  # noqa: E402

dev = torch.device("cuda:0")
for B, n, n_pre in ((1, 9450, 3000), (1, 37800, 3000), (16, 9450, 3000), (8, 37800, 3000), (1, 37800, 12000)):
    g = torch.Generator().manual_seed(3)
    keys = torch.rand(B, n, generator=g).to(dev) * 0.2 + 0.4        # fg probabilities bunch up around 0.5
    boxes = torch.randn(B, n, 4, generator=g).to(dev)
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    ts = []
    for _ in range(20):
        e0.record()
        hip_ops.sort_topk_desc(keys, boxes, n_pre)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    print(f"B={B:3d} n={n:6d} n_pre={n_pre:6d}  {ts[10]:8.1f} us")
