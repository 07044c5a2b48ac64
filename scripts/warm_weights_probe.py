#!/usr/bin/env python3
"""What would a next-layer weight prefetch be worth?  The conv launches of a batch-1 forward, in forward order, each bracketed by HIP
events - (a) as they are (weights as cold as the previous forward's other layers left them), (b) with the layer's weight image read
by a plain device kernel right before the bracket (an upper bound for any prefetch: the weights are as warm as they can be - in the
Infinity Cache and in the L2s of the XCDs that ran the read).  Usage (GPU box): python scripts/warm_weights_probe.py [tiles.json]"""
import json, os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_object_detection_amd._ffi import TILE_NAMES, stream_ptr
from two_stage_object_detection_amd.testing import synthetic_detector

dev = torch.device("cuda:0")
model, _ = synthetic_detector("resnet50", num_classes=80, seed=0)
model = model.to(dev).eval()
x = torch.rand(1, 3, 800, 1333, generator=torch.Generator().manual_seed(1234)).to(dev)
with torch.inference_mode():
    if len(sys.argv) > 1:
        table = json.load(open(sys.argv[1]))
        model.import_tuning(table, x)
    else:
        table = model.tune(x, schedules=("serial",), heads=False)
    model(x)
    plan = model.extractor._plan_for(x)
    steps = plan.gemm_steps
    s = stream_ptr()

    def weights_of(st):
        if hasattr(st, "pc"):                       # ConvStep: the image its arithmetic reads
            pc, pr = st.pc, int(st.desc.precision)
            return pc.w2[0] if pr == 2 else (pc.w3 if pr == 1 else pc.w)
        return None
    wts = [weights_of(st) for st in steps]

    def one_pass(warm):
        ev = []
        for st, w in zip(steps, wts):
            if warm and w is not None:
                w.view(torch.uint8).view(-1)[: w.numel() * w.element_size() // 4 * 4].view(torch.int32).sum()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); st.fn(*st.args, s); e1.record()
            ev.append((e0, e1))
        ev[-1][1].synchronize()
        return [a.elapsed_time(b) * 1e3 for a, b in ev]
    res = {}
    for warm in (False, True, False, True):
        runs = [one_pass(warm) for _ in range(8)][1:]
        res.setdefault(warm, []).append([statistics.median(v) for v in zip(*runs)])
    plan.clear_range_flag()
    cold = [min(a, b) for a, b in zip(*res[False])]
    hot = [min(a, b) for a, b in zip(*res[True])]
    tot_c, tot_h = sum(cold), sum(hot)
    print(f"sum over {len(steps)} launches: as they are {tot_c:.1f} us, weights read right before {tot_h:.1f} us, difference {tot_c - tot_h:.1f} us")
    for st, c, h, w in zip(steps, cold, hot, wts):
        d = st.desc
        tile = TILE_NAMES.get(int(getattr(d, "tile", -1)), "fused")
        mb = 0 if w is None else w.numel() * w.element_size() / 1e6
        print(f"  {st.name:28s} {tile:12s} split {int(getattr(d, 'split_k', 0)):3d} weights {mb:6.2f} MB  {c:6.1f} -> {h:6.1f} us  ({c - h:+.1f})")
