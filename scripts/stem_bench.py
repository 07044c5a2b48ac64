#!/usr/bin/env python3
"""One-launch stem (tsod_stem_fp16x2) against the three launches it replaces (layout, 7x7 conv, max pool) at 3x800x1336.
   python scripts/stem_bench.py [batch ...]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_object_detection_amd import hip_ops as ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
H, W = 800, 1336
w = (torch.randn(64, 3, 7, 7, generator=g) / math.sqrt(147)).to(dev)
scale, shift = (torch.rand(64, generator=g) + 0.5).to(dev), (torch.randn(64, generator=g) * 0.1).to(dev)
wfrag, e = ops.pack_stem_wfrag(w)
bn = torch.cat([scale, shift])
wp = ops.pack_conv_weight(w, cin_pad=4, kw_pad=8)


def timed(fn, reps=20):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for B in [int(a) for a in sys.argv[1:]] or [1, 8]:
    x = torch.randn(B, 3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
    oh, ow = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = torch.empty(B, (oh - 1) // 2 + 1, (ow - 1) // 2 + 1, 64, device=dev)
    words = ops.new_amax_words(dev, 3)
    t_f = timed(lambda: ops.stem_fused(x, wfrag, e, bn, 0.25, out=out, amax_out=words[0]))
    x4 = torch.empty(B, H, W, 4, device=dev)
    y = torch.empty(B, oh, ow, 64, device=dev)
    p = torch.empty_like(out)
    from two_stage_object_detection_amd._ffi import check, lib, ptr, stream_ptr

    def unfused():
        check(lib().tsod_nchw_to_nhwc_amax_f32(ptr(x), B, 3, H, W, ptr(x4), 4, 4, ptr(words[1]), stream_ptr()), "layout")
        ops.conv2d_nhwc(x4, wp, stride=2, pad=3, kw_logical=7, scale=scale, shift=shift, act=1, slope=0.25, precision=2, amax_in=words[1],
                        amax_out=words[2], out=y)
        check(lib().tsod_maxpool3x3s2_f32(ptr(y), B, oh, ow, 64, 64, ptr(p), 64, stream_ptr()), "maxpool")
    t_u = timed(unfused)
    err = (p - out).abs().max().item()
    flops = 2 * B * oh * ow * 64 * 147
    gb = 4 * (x.numel() + out.numel()) / 1e9
    print(f"B={B}: one launch {t_f:8.1f} us ({flops / t_f / 1e6:6.1f} TFLOP/s-eq, {gb / t_f * 1e6:5.0f} GB/s algorithmic)   three launches {t_u:8.1f} us   "
          f"ratio {t_u / t_f:.2f}   max |difference| {err:.2e}", flush=True)
