"""Times of the non-GEMM kernels of one B=1 ResNet-50 detector forward (maxpool, layout), HIP events.
    python scripts/tail_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_object_detection_amd import hip_ops  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, reps=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


x = torch.randn(1, 400, 667, 64, device=dev)
print(f"maxpool 400x667x64: {timeit(lambda: hip_ops.maxpool3x3s2_nhwc(x)):.1f} us (85.4 MB algorithmic)")
img = torch.rand(1, 3, 800, 1333, device=dev)
print(f"nchw->nhwc4 800x1333: {timeit(lambda: hip_ops.nchw_to_nhwc(img, 4)):.1f} us (29.9 MB)")
feat = torch.randn(1, 25, 42, 2048, device=dev)
g = torch.Generator().manual_seed(0)
xy = torch.rand(1, 300, 2, generator=g) * torch.tensor([1000.0, 600.0])
rois = torch.cat([xy, xy + torch.rand(1, 300, 2, generator=g) * 300 + 16], dim=-1).to(dev)
idx = torch.zeros(1, dtype=torch.int32, device=dev)
print(f"roi_pool_avg 300 x 2048: {timeit(lambda: hip_ops.roi_pool_avg_nhwc(feat, rois, idx, 800, 1333)):.1f} us")
print(f"roi_align_avg 300 x 2048: {timeit(lambda: hip_ops.roi_align_avg_nhwc(feat, rois, idx, 800, 1333)):.1f} us")

# proposal path at the detector's sizes: realistic overlap structure = boxes decoded from a random-init RPN on one image
from two_stage_object_detection_amd import _ffi  # noqa: E402
L = _ffi.lib()
for B, A in ((1, 9450), (8, 9450)):
    g = torch.Generator().manual_seed(5)
    ctr = torch.rand(B, A, 2, generator=g) * torch.tensor([1333.0, 800.0])
    wh = torch.rand(B, A, 2, generator=g) * 300 + 20
    boxes = torch.cat([ctr - wh / 2, ctr + wh / 2], dim=-1).clamp_(0, 1333).to(dev)
    keys = torch.rand(B, A, generator=g).to(dev)
    counts, idx, bs, ks = hip_ops.sort_topk_desc(keys, boxes, 3000)
    print(f"B={B}: sort_topk {A} -> 3000: {timeit(lambda: hip_ops.sort_topk_desc(keys, boxes, 3000)):.1f} us (incl. host-side allocation gaps)")
    keep, rois, n_kept, status = hip_ops.nms_sorted(bs, counts, 0.7, 300)
    print(f"B={B}: nms (mask + scan) 3000 -> 300: {timeit(lambda: hip_ops.nms_sorted(bs, counts, 0.7, 300)):.1f} us, kept {n_kept.tolist()}")
