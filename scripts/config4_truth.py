#!/usr/bin/env python3
"""Config 4's box error against a float64 evaluation of the same network: WHO is far from the exact answer?

HarDNet-68, batch 8, 3x800x1333 (the images of tests/test_bench_config_parity.py).  For two images of the batch:
  truth   : the CPU oracle's trunk + RPN convs evaluated in float64, its loc / score tensors rounded ONCE to f32 and handed to the
            reference's f32 proposal layer (the decode, clamp, sort and NMS are the reference's own f32 operations either way)
  oracle  : the CPU oracle as the tests use it (torch CPU f32)
  oracle' : the same with BatchNorm applied as ONE multiply-add by scale / shift folded in f64 (what the HIP epilogue does)
  gpu f32 / fp16x2 / tuned : the HIP path under three arithmetics
Printed: feature-map error of each against truth (relative to the abs-max), and the RoI distance (one-to-one rows, L-inf px) of
every pair.  Run on the GPU box: python scripts/config4_truth.py [image indices, comma list]"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle  # noqa: E402
import oracle.backbones as ob  # noqa: E402
from oracle.box import enumerate_shifted_anchor, generate_basic_anchor, proposal_layer  # noqa: E402
from two_stage_object_detection_amd.testing import synthetic_detector  # noqa: E402

dev = torch.device("cuda:0")
torch.set_num_threads(16)
g = lambda shape, seed: torch.rand(shape, generator=torch.Generator().manual_seed(seed))   # noqa: E731
model, sd = synthetic_detector("hardnet68", num_classes=80, seed=0, conditioned=True)
model = model.to(dev).eval()
x = g((8, 3, 800, 1333), 21)
xg = x.to(dev)
images = tuple(int(v) for v in sys.argv[1].split(",")) if len(sys.argv) > 1 else (0, 5)
sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}


def rois_from(locs, scores, img_size):
    """the reference's f32 proposal path on given RPN outputs [1,A,4], [1,A,2] (oracle.box.rpn_forward without its convs)"""
    fg = F.softmax(scores, dim=-1)[:, :, 1].contiguous().view(1, -1)
    base = generate_basic_anchor()
    anchor = enumerate_shifted_anchor(base, 16, 50, 84)
    return proposal_layer(locs[0], fg[0], anchor, img_size, scale=1.0, mode="training")


def rpn_convs(s, feat):
    n = feat.shape[0]
    locs = F.conv2d(feat, s["rpn.loc.weight"], s["rpn.loc.bias"]).permute(0, 2, 3, 1).contiguous().view(n, -1, 4)
    scores = F.conv2d(feat, s["rpn.score.weight"], s["rpn.score.bias"]).permute(0, 2, 3, 1).contiguous().view(n, -1, 2)
    return locs, scores


def roi_dist(a, b):
    """max over rows of a of the distance to its one-to-one partner in b (positional first, then nearest free row)"""
    d = (a.unsqueeze(1) - b.unsqueeze(0)).abs().amax(-1)
    R = a.shape[0]
    diag = torch.arange(R)
    same = d[diag, diag] <= 2e-3
    taken = same.clone()
    worst = float(d[diag, diag][same].max()) if same.any() else 0.0
    moved = 0
    for i in torch.nonzero(~same).flatten().tolist():
        row = torch.where(taken, torch.full_like(d[i], float("inf")), d[i])
        j = int(row.argmin())
        taken[j] = True
        worst = max(worst, float(row[j]))
        moved += 1
    return worst, moved


_bn_plain = ob._bn


def _bn_folded(s, p, t):
    """BatchNorm as the HIP epilogue applies it: scale / shift folded in f64, rounded once to f32, ONE multiply-add per element"""
    var, mean = s[p + ".running_var"].double(), s[p + ".running_mean"].double()
    scale = s[p + ".weight"].double() / torch.sqrt(var + 1e-5)
    shift = s[p + ".bias"].double() - mean * scale
    return torch.addcmul(shift.to(t.dtype).view(1, -1, 1, 1), t, scale.to(t.dtype).view(1, -1, 1, 1))


out = {}
with torch.inference_mode():
    gpu = {}
    for tag, prec in (("gpu_f32", "f32"), ("gpu_fp16x2", "fp16x2"), ("gpu_bf16x3", "bf16x3")):
        model.extractor.set_conv_precision(prec)
        o = model(xg)
        feat = model(xg, mode="extractor")
        model.raise_if_error()
        gpu[tag] = (o[2].cpu(), feat.cpu())
    model.extractor.set_conv_precision("f32")
    table = model.tune(xg, precisions=(0, 1, 2), schedules=("serial",), reps=2)
    o = model(xg)
    feat = model(xg, mode="extractor")
    model.raise_if_error()
    gpu["gpu_tuned"] = (o[2].cpu(), feat.cpu())
    print("tuned: fp16x2 layers", sum(1 for r in table["serial"] if r[3] == 2), "of", len(table["serial"]), "heads", table["heads"], flush=True)
    for i in images:
        xi = x[i:i + 1]
        img_size = tuple(xi.shape[1:])
        f64 = oracle.hardnet_trunk(sd64, xi.double(), arch=68, prefix="extractor.")
        l64, s64 = rpn_convs(sd64, f64)
        rois = {"truth": rois_from(l64.float(), s64.float(), img_size)}
        feats = {}
        f32 = oracle.hardnet_trunk(sd, xi, arch=68, prefix="extractor.")
        feats["oracle"] = f32
        rois["oracle"] = rois_from(*rpn_convs(sd, f32), img_size)
        ob._bn = _bn_folded
        try:
            f32f = oracle.hardnet_trunk(sd, xi, arch=68, prefix="extractor.")
        finally:
            ob._bn = _bn_plain
        feats["oracle_bn_folded"] = f32f
        rois["oracle_bn_folded"] = rois_from(*rpn_convs(sd, f32f), img_size)
        # the f32 trunk's features through a float64 RPN conv: is it the trunk or the last GEMM?
        l_mix, s_mix = rpn_convs(sd64, f32.double())
        rois["oracle_trunk_f32_rpn_f64"] = rois_from(l_mix.float(), s_mix.float(), img_size)
        for tag, (r, f) in gpu.items():
            rois[tag] = r[i]
            feats[tag] = f[i:i + 1]
        scale = float(f64.abs().max())
        rep = {"feature_abs_max": scale, "feature_err_vs_truth": {k: float((v.double() - f64).abs().max()) / scale for k, v in feats.items()},
               "feature_rms_err_vs_truth": {k: float((v.double() - f64).pow(2).mean().sqrt()) / scale for k, v in feats.items()}}
        names = list(rois)
        rep["roi_dist"] = {}
        for a in names:
            for b in names:
                if a < b:
                    w, moved = roi_dist(rois[a], rois[b])
                    rep["roi_dist"][f"{a} | {b}"] = [round(w, 7), moved]
        out[i] = rep
        print("image", i, json.dumps(rep, indent=1), flush=True)
od = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
if os.path.isdir(od):
    json.dump(out, open(os.path.join(od, "config4_truth.json"), "w"), indent=1)
