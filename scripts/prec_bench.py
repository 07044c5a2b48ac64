"""f32-MFMA vs bf16x3 conv path per ResNet-50 layer shape: best (tile, split) of each precision, HIP-event timed.
    python scripts/prec_bench.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_object_detection_amd import hip_ops  # noqa: E402
from two_stage_object_detection_amd._ffi import BF16X3_TILE_IDS, DMA_TILE_IDS, TILE_IDS, TILE_NAMES  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
shapes = [("l1.conv2 3x3 64->64", 200, 334, 64, 64, 3), ("l1.conv3 1x1 64->256", 200, 334, 64, 256, 1),
          ("l1.conv1 1x1 256->64", 200, 334, 256, 64, 1), ("l2.conv2 3x3 128->128", 100, 167, 128, 128, 3),
          ("l2.conv3 1x1 128->512", 100, 167, 128, 512, 1), ("l2.conv1 1x1 512->128", 100, 167, 512, 128, 1),
          ("l3.conv2 3x3 256->256", 50, 84, 256, 256, 3), ("l3.conv3 1x1 256->1024", 50, 84, 256, 1024, 1),
          ("l3.conv1 1x1 1024->256", 50, 84, 1024, 256, 1), ("l4.conv2 3x3 512->512", 25, 42, 512, 512, 3),
          ("l4.conv3 1x1 512->2048", 25, 42, 512, 2048, 1), ("l4.conv1 1x1 2048->512", 25, 42, 2048, 512, 1)]


def timeit(fn, reps=6):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


for name, H, W, Cin, Cout, k in shapes:
    x = torch.randn(B, H, W, Cin, device=dev)
    w = torch.randn(Cout, k, k, Cin, device=dev) / (Cin * k * k) ** 0.5
    res = torch.randn(B, H, W, Cout, device=dev)
    flops = 2 * B * H * W * Cout * Cin * k * k
    best = {}
    for prec, tiles in ((0, TILE_IDS), (1, BF16X3_TILE_IDS)):
        out = []
        for tile in tiles:
            for split in (1, -1, -2, 2, 3, 4, 6, 8, 12):
                if split > 1 and ((Cin * k * k) // 32) // split < 2:
                    continue
                if split > 1 and B * H * W > 20000:
                    continue
                if split == -2 and not (prec == 1 and tile in DMA_TILE_IDS):
                    continue
                t = timeit(lambda: hip_ops.conv2d_nhwc(x, w, pad=k // 2, tile=tile, split_k=split, residual=res, act=1,
                                                       slope=0.25, precision=prec))
                out.append((t, tile, split))
        best[prec] = min(out)
    f, b = best[0], best[1]
    print(f"{name:26s} M={B * H * W:7d}  f32 {f[0] * 1e3:7.1f} us {flops / f[0] / 1e9:6.1f} TF/s ({TILE_NAMES[f[1]]}/s{f[2]})   "
          f"bf16x3 {b[0] * 1e3:7.1f} us {flops / b[0] / 1e9:6.1f} TF/s-eq ({TILE_NAMES[b[1]]}/s{b[2]})   x{f[0] / b[0]:.2f}", flush=True)
