"""In-kernel clock of the conv kernel under sustained load (diagnostic build: make -C .../csrc diag, TSOD_LIB=.../libtsod_diag.so).

    TSOD_LIB=two_stage_object_detection_amd/libtsod_diag.so python scripts/conv_clock.py [B]

Every workgroup's thread 0 stamps s_memtime / s_memrealtime around its K loop; after ~1 s of back-to-back launches the
median ratio x 100 MHz is the clock the chip holds under that loop (MI355X_MICROARCH.md, DVFS give-back item 6)."""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_object_detection_amd import hip_ops  # noqa: E402
from two_stage_object_detection_amd._ffi import TILE_NAMES, lib  # noqa: E402

dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
L = lib()
L.tsod_debug_set_clock_buf.argtypes = [ctypes.c_void_p]
buf = torch.zeros(2 * 32768, dtype=torch.int64, device=dev)
for (H, W, Cin, Cout, k) in ((100, 167, 128, 128, 3), (100, 167, 512, 128, 1), (50, 84, 512, 512, 3)):
    x = torch.randn(B, H, W, Cin, device=dev)
    w = torch.randn(Cout, k, k, Cin, device=dev) / (Cin * k * k) ** 0.5
    fl = 2 * B * H * W * Cout * Cin * k * k
    for tile in (8, 10, 14, 5):
        L.tsod_debug_set_clock_buf(None)
        fn = lambda: hip_ops.conv2d_nhwc(x, w, pad=k // 2, tile=tile, split_k=1)   # noqa: E731
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 0
        while time.perf_counter() - t0 < 1.0:
            for _ in range(20):
                fn()
            torch.cuda.synchronize(); n += 20
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        buf.zero_(); L.tsod_debug_set_clock_buf(buf.data_ptr())
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        s = buf.view(-1, 2).cpu().double()
        s = s[s[:, 1] > 50]                                     # loops shorter than 0.5 us give no usable ratio
        clk = (s[:, 0] / s[:, 1] * 100e6).median().item() / 1e9
        tf = fl / ms / 1e9
        print(f"B={B} {H}x{W} {Cin}->{Cout} k{k} {TILE_NAMES[tile]:12s} {tf:6.1f} TF/s  in-kernel clock {clk:.2f} GHz "
              f"-> {tf / (157.3 * clk / 2.4) * 100:.0f}% of the f32-MFMA peak at that clock", flush=True)
