"""Experiment: spatial partitioning of the GPU between the requests in flight (CU-masked HIP streams) instead of letting
whole-chip kernels of several streams interleave.  Prints images/s for: 4 plain streams (eager launches), 4 CU-masked
streams of 64 CUs each (eager), and - if replaying works there - graphs on the masked streams.

    python scripts/cu_mask_experiment.py [partitions]"""
import ctypes
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_object_detection_amd import hip_ops  # noqa: E402
from two_stage_object_detection_amd.testing import synthetic_detector  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipExtStreamCreateWithCUMask.restype = ctypes.c_int


def masked_stream(first_cu, n_cu, interleaved=False, total=256):
    words = (ctypes.c_uint32 * (total // 32))()
    for i in range(n_cu):
        cu = (first_cu + i) if not interleaved else ((first_cu // n_cu) + i * (total // n_cu))
        words[cu // 32] |= 1 << (cu % 32)
    s = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), total // 32, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return torch.cuda.ExternalStream(s.value, device=dev)


model, _ = synthetic_detector("resnet50", num_classes=80, seed=0)
model = model.to(dev).eval()
x = torch.rand(1, 3, 800, 1333, generator=torch.Generator().manual_seed(1234)).to(dev)


def run(streams, steps=60, graphs=None, label=""):
    with torch.inference_mode():
        for i in range(2 * len(streams)):
            s = streams[i % len(streams)]
            with torch.cuda.stream(s):
                if graphs:
                    graphs[i % len(streams)]()
                else:
                    o = model(x, slot=i % len(streams)); hip_ops.detections(o[0], o[1], o[2])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            s = streams[i % len(streams)]
            with torch.cuda.stream(s):
                if graphs:
                    graphs[i % len(streams)]()
                else:
                    o = model(x, slot=i % len(streams)); hip_ops.detections(o[0], o[1], o[2])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{label:58s} {steps / dt:7.1f} images/s", flush=True)


with torch.inference_mode():
    for sl in range(P):
        model(x, slot=sl)
    plain = [torch.cuda.Stream(dev) for _ in range(P)]
    ms = [masked_stream(i * (256 // P), 256 // P, interleaved=True) for i in range(P)]
    for sl in range(P):                                  # tune every slot's tiles on its own partition
        with torch.cuda.stream(ms[sl]):
            model.extractor._plan_for(x, sl).autotune()
    gs = [model.make_graphed(x, slot=sl)[0] for sl in range(P)]
    for _ in range(2):
        run(ms, graphs=gs, steps=120, label=f"{P} CU-masked streams ({256 // P} CUs each, strided), tiles tuned per partition, graphs:")
    # the shipped mode: plain streams, whole-chip tiles tuned with two copies in flight, one graph per slot
    model.extractor._plan_for(x, 0).autotune(concurrent=2)
    tiles = model.extractor._plan_for(x, 0).export_tiles()
    for sl in range(1, P):
        model.extractor._plan_for(x, sl).import_tiles(tiles)
    gs = [model.make_graphed(x, slot=sl)[0] for sl in range(P)]
    for _ in range(2):
        run(plain, graphs=gs, steps=120, label=f"{P} plain streams, whole-chip tiles (2-copy objective), graphs (shipped mode):")
