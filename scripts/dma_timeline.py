"""Where a conv_dma_kernel launch spends its time (diagnostic build: make -C two_stage_object_detection_amd/csrc diag).

    TSOD_LIB=two_stage_object_detection_amd/libtsod_diag.so python scripts/dma_timeline.py [B] [layer names, comma list]

Thread 0 of every workgroup stamps s_memrealtime at entry / exit and accumulates s_memtime cycles in its prologues (ring fill
until stage 0 is visible), K loops and epilogues (transpose + stores, or slab store + ticket + combine).  Printed per
(layer shape, tile, schedule): HIP-event time per launch; workgroups; first-to-last START skew; per-workgroup LIFETIME
(mean / max); mean and max microseconds in prologue / K loop / epilogue; the in-kernel clock; and how long after the first
start the last workgroup ended (= the kernel's own span, launch overhead excluded)."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_object_detection_amd import hip_ops  # noqa: E402
from two_stage_object_detection_amd._ffi import TILE_NAMES, TsodError, lib  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
L = lib()
L.tsod_debug_set_dma_stamps.argtypes = [ctypes.c_void_p]
buf = torch.zeros(8 * 8192, dtype=torch.int64, device=dev)
SHAPES = [(50, 84, 256, 256, 3, "layer3.conv2"), (25, 42, 512, 512, 3, "layer4.conv2"), (100, 167, 128, 128, 3, "layer2.conv2"),
          (50, 84, 1024, 256, 1, "layer3.conv1"), (50, 84, 256, 1024, 1, "layer3.conv3"), (25, 42, 2048, 512, 1, "layer4.conv1"),
          (200, 334, 64, 64, 3, "layer1.conv2"), (100, 167, 128, 512, 1, "layer2.conv3"), (100, 167, 512, 128, 1, "layer2.conv1"),
          (25, 42, 512, 2048, 1, "layer4.conv3")]
# (register-staged tiles carry no stamps: their rows show the launch time only)
SCHEDS = [(8, 1), (8, -1), (10, 1), (14, 1), (15, 1), (15, -1), (16, 1), (17, 1), (18, 1), (18, 3), (20, 1), (22, 1), (22, 3), (22, -1), (22, -2), (19, -1), (24, 1), (24, -1), (24, 2), (24, -2)]
PREC = int(os.environ.get("TSOD_TIMELINE_PREC", "1"))          # 1 = bf16x3, 2 = fp16x2 (tile d128x128k32 only)
if len(sys.argv) > 2:
    SHAPES = [s for s in SHAPES if s[5] in sys.argv[2].split(",")]
for (H, W, Cin, Cout, k, name) in SHAPES:
    x = torch.randn(B, H, W, Cin, device=dev)
    w = torch.randn(Cout, k, k, Cin, device=dev) / (Cin * k * k) ** 0.5
    res = torch.randn(B, H, W, Cout, device=dev)
    fl = 2 * B * H * W * Cout * Cin * k * k
    print(f"--- {name}: B={B} {H}x{W} {Cin}->{Cout} k{k}  ({fl / 1e9:.2f} GFLOP)")
    for tile, split in SCHEDS:
        if PREC == 2:
            wexp = hip_ops.fp16x2_weight_scale_exp(w)
            w2 = hip_ops.pack_conv_weight_fp16x2(w, wexp)
            fn = lambda: hip_ops.conv2d_nhwc(x, w, pad=k // 2, tile=tile, split_k=split, precision=2, residual=res, act=1, slope=0.25,  # noqa: E731
                                             w2=w2, w_scale_exp=wexp)
        else:
            fn = lambda: hip_ops.conv2d_nhwc(x, w, pad=k // 2, tile=tile, split_k=split, precision=1, residual=res, act=1, slope=0.25)  # noqa: E731
        L.tsod_debug_set_dma_stamps(None)
        try:
            fn()
        except TsodError:
            continue
        torch.cuda.synchronize()
        for _ in range(30):
            fn()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        buf.zero_(); L.tsod_debug_set_dma_stamps(buf.data_ptr())
        fn(); torch.cuda.synchronize()
        L.tsod_debug_set_dma_stamps(None)
        s = buf.view(-1, 8).cpu().double()
        s = s[s[:, 1] > 0]
        n = s.shape[0]
        if n == 0:
            print(f"  {TILE_NAMES[tile]:9s} split {split:2d}: {us:6.1f} us/launch {fl / us / 1e6:6.1f} TF/s | (register-staged: no stamps)", flush=True)
            continue
        t0 = s[:, 0].min()
        start, end = (s[:, 0] - t0) / 100.0, (s[:, 1] - t0) / 100.0                  # us (100 MHz)
        life = end - start
        cyc = s[:, 2] + s[:, 3] + s[:, 4]
        clk = (cyc.sum() / (life.sum() * 1e-6) / 1e9).item()                        # GHz (stamped cycles / lifetime)
        tous = lambda c: c / (clk * 1e3)                                             # noqa: E731
        pro, loop, epi = tous(s[:, 2]), tous(s[:, 3]), tous(s[:, 4])
        xcd = s[:, 7].long() & 15
        print(f"  {TILE_NAMES[tile]:9s} split {split:2d}: {us:6.1f} us/launch {fl / us / 1e6:6.1f} TF/s | {n:4d} WGs, start skew {start.max():.1f} us, "
              f"span {end.max():.1f} us, life mean {life.mean():.1f} max {life.max():.1f} | prologue {pro.mean():.1f}/{pro.max():.1f} "
              f"kloop {loop.mean():.1f}/{loop.max():.1f} ({(s[:, 5].mean()):.0f} steps, {tous(s[:, 3].sum() / s[:, 5].sum()) * 1e3:.0f} ns/step) "
              f"epilogue {epi.mean():.1f}/{epi.max():.1f} (p50 {epi.median():.1f}) | clk {clk:.2f} GHz | WGs per XCD {torch.bincount(xcd, minlength=8).tolist()}",
              flush=True)
