#!/usr/bin/env python3
"""Fused bottleneck (tsod_bottleneck_fp16x2) against the three conv launches it replaces, at layer1's size.
   python scripts/bottleneck_bench.py [batch ...]"""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_object_detection_amd import hip_ops as ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
C, H, W = 256, 200, 334
w1 = (torch.randn(64, C, 1, 1, generator=g) / math.sqrt(C)).to(dev)
w2 = (torch.randn(64, 64, 3, 3, generator=g) / 24.0).to(dev)
w3 = (torch.randn(C, 64, 1, 1, generator=g) / 8.0).to(dev)
bn = [(torch.rand(n, generator=g) + 0.5).to(dev) if i % 2 == 0 else (torch.randn(n, generator=g) * 0.1).to(dev) for i, n in enumerate((64, 64, 64, 64, C, C))]
w1p, w2p, w3p = ops.pack_conv_weight(w1), ops.pack_conv_weight(w2), ops.pack_conv_weight(w3)
stream, exps = ops.pack_bottleneck_wstream(w1.view(64, C), w2p, w3.view(C, 64))
bnv = torch.cat(bn)


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
    x = torch.randn(B, H, W, C, generator=torch.Generator().manual_seed(1)).to(dev)
    words = ops.absmax(x, ops.new_amax_words(dev, 1))
    out = torch.empty_like(x)
    t_f = timed(lambda: ops.bottleneck_fused(x, stream, exps, bnv, C, 0.25, out=out, amax_in=words))
    y1 = torch.empty(B, H, W, 64, device=dev)
    y2 = torch.empty_like(y1)
    o3 = torch.empty_like(x)
    wy = ops.new_amax_words(dev, 2)

    def unfused(tiles):
        ops.conv2d_nhwc(x, w1p, scale=bn[0], shift=bn[1], act=1, slope=0.25, precision=2, amax_in=words, amax_out=wy[0], out=y1, tile=tiles[0][0], split_k=tiles[0][1])
        ops.conv2d_nhwc(y1, w2p, pad=1, scale=bn[2], shift=bn[3], act=1, slope=0.25, precision=2, amax_in=wy[0], amax_out=wy[1], out=y2, tile=tiles[1][0], split_k=tiles[1][1])
        ops.conv2d_nhwc(y2, w3p, scale=bn[4], shift=bn[5], residual=x, act=1, slope=0.25, precision=2, amax_in=wy[1], out=o3, tile=tiles[2][0], split_k=tiles[2][1])
    # the tiles the round-3 tables pinned for layer1.1 (b1: 64x64s1 x3; b8: 64x64s1k64, 128x64s1, 64x128s1)
    t_u = min(timed(lambda: unfused(t)) for t in (((8, 1), (8, 1), (8, -1)), ((10, 1), (14, 1), (15, 1)), ((0, 0), (0, 0), (0, 0))))
    err = (o3 - out).abs().max().item()
    flops = 2 * B * H * W * (C * 64 + 576 * 64 + 64 * C)
    print(f"B={B}: fused {t_f:8.1f} us ({flops / t_f / 1e6:6.1f} TFLOP/s-eq)   three launches {t_u:8.1f} us   ratio {t_u / t_f:.2f}   max |fused - unfused| {err:.2e}", flush=True)
