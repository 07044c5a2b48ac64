"""All (LDS-DMA tile, K schedule) times of one conv shape: python scripts/dma_scan.py B H W Cin Cout k"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from two_stage_object_detection_amd import hip_ops, _ffi

B, H, W, Cin, Cout, k = (int(v) for v in sys.argv[1:7])
dev = torch.device("cuda:0")
x = torch.randn(B, H, W, Cin, device=dev)
w = hip_ops.pack_conv_weight(torch.randn(Cout, Cin, k, k, device=dev) / (Cin * k * k) ** 0.5)
flops = 2 * B * H * W * Cout * Cin * k * k


def timeit(fn, reps=20):
    fn(); fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps


for tile in (8, 15) + tuple(_ffi.DMA_TILE_IDS):
    row = []
    for split in (1, -1, -2, 2, 3, 4, 6, 8):
        try:
            t = timeit(lambda: hip_ops.conv2d_nhwc(x, w, pad=k // 2, tile=tile, split_k=split, precision=1))
        except _ffi.TsodError:
            continue
        row.append(f"s{split}: {t * 1e3:6.1f} us ({flops / t / 1e9:5.1f})")
    print(f"{_ffi.TILE_NAMES[tile]:10s} " + "  ".join(row), flush=True)
