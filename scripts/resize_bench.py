"""Time the input-step kernel (u8 HWC -> antialiased bilinear -> f32 NHWC4) against its HBM roofline and the CPU oracle.
    python scripts/resize_bench.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402  (baseline leg only)
from two_stage_object_detection_amd import hip_ops  # noqa: E402

dev = torch.device("cuda:0")
for (H, W), (OH, OW) in (((1080, 1920), (600, 600)), ((1080, 1920), (800, 1333)), ((2160, 3840), (800, 1333)), ((480, 640), (800, 1333))):
    img = torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, generator=torch.Generator().manual_seed(0))
    d = img.to(dev)
    out = torch.empty((OH, OW, 4), device=dev)
    flush = torch.empty(96 << 20, device=dev)
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    ts = []
    for _ in range(20):
        flush.zero_()
        e0.record()
        hip_ops.resize_bilinear_aa(d, OH, OW, out=out)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    t = ts[10]
    byt = H * W * 3 + OH * OW * 16
    torch.set_num_threads(16)
    oracle.eval_transform(img, (OH, OW))
    t0 = time.perf_counter()
    for _ in range(5):
        oracle.eval_transform(img, (OH, OW))
    cpu = (time.perf_counter() - t0) / 5
    print(f"{H}x{W} -> {OH}x{OW}: {t:7.1f} us  {byt / t / 1e3:7.1f} GB/s algorithmic ({byt / 1e6:.1f} MB)   CPU oracle {cpu * 1e3:.2f} ms")
