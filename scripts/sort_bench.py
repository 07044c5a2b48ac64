"""Time the per-image top-k at the detector's sizes (kernel time: 50 back-to-back launches on preallocated outputs between
two HIP events, host overhead excluded): python scripts/sort_bench.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_object_detection_amd import _ffi  # noqa: E402

dev = torch.device("cuda:0")
L = _ffi.lib()
for B, n, n_pre in ((1, 9450, 3000), (1, 37800, 3000), (8, 9450, 3000), (16, 9450, 3000), (8, 37800, 3000), (1, 37800, 12000)):
    g = torch.Generator().manual_seed(3)
    keys = torch.rand(B, n, generator=g).to(dev) * 0.2 + 0.4        # fg probabilities bunch up around 0.5
    boxes = torch.randn(B, n, 4, generator=g).to(dev)
    counts = torch.empty((B,), dtype=torch.int32, device=dev)
    idx = torch.empty((B, n_pre), dtype=torch.int32, device=dev)
    bs = torch.empty((B, n_pre, 4), device=dev)
    ks = torch.empty((B, n_pre), device=dev)
    res = {}
    for name, with_ws in (("scratch (rank on the chip)", True), ("no scratch (network in one workgroup)", False)):
        wsb = L.tsod_sort_topk_workspace_bytes(B, n, n_pre) if with_ws else 0
        ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
        s = _ffi.stream_ptr()

        def run():
            return L.tsod_sort_topk_desc_ws_f32(_ffi.ptr(keys), _ffi.ptr(boxes), B, n, n_pre, _ffi.ptr(counts), _ffi.ptr(idx), _ffi.ptr(bs),
                                                _ffi.ptr(ks), _ffi.ptr(ws) if wsb else None, wsb, s)
        for _ in range(5):
            _ffi.check(run())
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(50):
            run()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 50 * 1e3
    print(f"B={B:3d} n={n:6d} n_pre={n_pre:6d}  " + "   ".join(f"{k}: {v:7.1f} us" for k, v in res.items()), flush=True)
