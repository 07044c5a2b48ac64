"""Where a K-slice's epilogue spends its time (diagnostic build: make -C two_stage_object_detection_amd/csrc diag).

    TSOD_LIB=two_stage_object_detection_amd/libtsod_diag.so python scripts/epi_timeline.py [B]

Thread 0 of every K-sliced workgroup stamps s_memtime at: epilogue entry, slab stores drained (vmcnt(0)), ticket add returned,
the other slices' slabs (+ residual) in registers, exit.  Printed per (layer, tile, split): microseconds of each leg, mean over
the last arrivers (the workgroups on the launch's critical path) and over the others (which stop after the ticket)."""
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
L.tsod_debug_set_epi_stamps.argtypes = [ctypes.c_void_p]
L.tsod_debug_set_dma_stamps.argtypes = [ctypes.c_void_p]
buf = torch.zeros(8 * 8192, dtype=torch.int64, device=dev)
buf2 = torch.zeros(8 * 8192, dtype=torch.int64, device=dev)
SHAPES = [(50, 84, 256, 256, 3, "layer3.conv2", False), (50, 84, 1024, 256, 1, "layer3.conv1", False), (25, 42, 512, 512, 3, "layer4.conv2", False),
          (25, 42, 512, 2048, 1, "layer4.conv3", True)]
SCHEDS = [(22, 2), (22, 3), (22, 4), (22, -1)]
for (H, W, Cin, Cout, k, name, with_res) in SHAPES:
    x = torch.randn(B, H, W, Cin, device=dev)
    w = torch.randn(Cout, k, k, Cin, device=dev) / (Cin * k * k) ** 0.5
    res = torch.randn(B, H, W, Cout, device=dev) if with_res else None
    w3 = hip_ops.pack_conv_weight_bf16x3(w.reshape(Cout, -1))
    print(f"--- {name}: B={B} {H}x{W} {Cin}->{Cout} k{k} residual={with_res}")
    for tile, split in SCHEDS:
        fn = lambda: hip_ops.conv2d_nhwc(x, w, pad=k // 2, tile=tile, split_k=split, precision=1, residual=res, act=1, slope=0.25, w3=w3)  # noqa: E731
        try:
            fn()
        except TsodError:
            continue
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        buf.zero_(); buf2.zero_()
        L.tsod_debug_set_epi_stamps(buf.data_ptr()); L.tsod_debug_set_dma_stamps(buf2.data_ptr())
        fn(); torch.cuda.synchronize()
        L.tsod_debug_set_epi_stamps(None); L.tsod_debug_set_dma_stamps(None)
        s = buf.view(-1, 8).cpu().double()
        d = buf2.view(-1, 8).cpu().double()
        ok = s[:, 0] > 0
        s, d = s[ok], d[ok]
        life_us = (d[:, 1] - d[:, 0]) / 100.0
        cyc = d[:, 2] + d[:, 3] + d[:, 4]
        clk = (cyc.sum() / (life_us.sum() * 1e-6) / 1e9).item()
        last = s[:, 5] > 0
        us = lambda a, b, m: (((s[m, b] - s[m, a]).mean()) / (clk * 1e3)).item()  # noqa: E731
        print(f"  {TILE_NAMES[tile]:11s} split {split:2d}: {int(ok.sum()):4d} K-sliced WGs, {int(last.sum()):3d} last arrivers, clk {clk:.2f} GHz | "
              f"LAST: stores+drain {us(0, 1, last):.1f}  ticket {us(1, 2, last):.1f}  slab+residual loads {us(2, 3, last):.1f}  sum+act+stores {us(3, 4, last):.1f} us"
              f" | OTHERS: stores+drain {us(0, 1, ~last):.1f}  ticket {us(1, 2, ~last):.1f} us", flush=True)
