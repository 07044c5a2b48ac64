"""Synthetic-weight construction and output comparison used by tests/, bench.py and smoke().

No oracle import here: this module only builds a model and compares two sets of tensors.
"""
from __future__ import annotations

import torch


def synthetic_bn_path(backbone, seed):
    import os
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs", f"synthetic_bn_{backbone}_seed{seed}.npz")


def weights_checksum(sd) -> float:
    """sum of |w| over the extractor's conv weights (f64): ties a file of pre-computed BN statistics to the weights it was made for"""
    return float(sum(v.double().abs().sum() for k, v in sd.items() if k.startswith("extractor.") and k.endswith("weight") and v.dim() == 4))


def synthetic_detector(backbone="resnet50", num_classes=80, seed=0, mode="training", rpn_loc_gain=None,
                       rpn_score_gain=None, head_gain=None, conditioned=False):
    """The detector with seeded random-init weights of the reference architecture:
    ``torch.manual_seed(seed)`` then modules constructed in reference order (ResNet: Kaiming fan_out on
    every conv, BN identity, PReLU 0.25; HarDNet / RPN / head: PyTorch defaults).

    With random init the trunk's activations grow to abs-max ~100 (ResNet-50), which would saturate
    the RPN softmax and exp() of the box decode; the RPN / head weights are therefore scaled by fixed
    gains so that fg probabilities and box offsets are in the range a trained detector produces
    (loc std ~0.3, logit std ~2).  Gains are constants (no data pass), so weights depend on the seed only.
    ``conditioned`` (HarDNet only): load the BatchNorm running statistics pre-computed for these very weights
    (configs/synthetic_bn_<backbone>_seed<seed>.npz, made by scripts/make_synthetic_bn_stats.py: the batch statistics of two
    seeded images flowing through the trunk).  With identity BN a random-init HarDNet collapses every image to a spatially
    constant feature map (~2950 of 3000 RPN scores tie exactly): no workload to time or to check a detector on.
    Returns (model on CPU in eval mode, CPU state_dict with the reference's key names)."""
    from .nets.frcnn import FasterRCNN
    torch.manual_seed(seed)
    model = FasterRCNN(num_classes, mode=mode, backbone=backbone).eval()
    defaults = {"resnet50": (0.03, 0.15, 0.05), "hardnet39": (0.75, 8.0, 2.0), "hardnet68": (0.75, 8.0, 2.0),
                "hardnet85": (0.75, 8.0, 2.0)}[backbone]
    gl = defaults[0] if rpn_loc_gain is None else rpn_loc_gain
    gs = defaults[1] if rpn_score_gain is None else rpn_score_gain
    gh = defaults[2] if head_gain is None else head_gain
    with torch.no_grad():
        for conv, g in ((model.rpn.loc, gl), (model.rpn.score, gs)):
            conv.weight.mul_(g)
            conv.bias.mul_(g)
        for lin in (model.head.cls_loc, model.head.score):
            lin.weight.mul_(gh)
            lin.bias.mul_(gh)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    if conditioned and backbone.startswith("hardnet"):
        import numpy as np
        path = synthetic_bn_path(backbone, seed)
        try:
            data = np.load(path)
        except OSError as e:
            raise RuntimeError(f"no pre-computed BatchNorm statistics for {backbone} seed {seed} ({path}): run "
                               "scripts/make_synthetic_bn_stats.py") from e
        want, have = float(data["__weights_checksum__"]), weights_checksum(sd)
        if abs(want - have) > 1e-6 * abs(want):
            raise RuntimeError(f"{path} was made for other weights (checksum {want} vs {have}): re-run scripts/make_synthetic_bn_stats.py")
        for k in data.files:
            if k != "__weights_checksum__":
                sd[k] = torch.from_numpy(data[k]).clone()
        model.load_state_dict(sd)
        model.eval()
    return model, sd


def cutoff_candidates(rpn_debug, n_post, extra=16):
    """From the oracle's RPN debug record (``oracle.detector_forward(..., return_debug=True)[1]``): per image, what the proposal
    list's cut at ``n_post`` rows looks like from both sides - (the boxes [k,4] and fg scores [k] of the next ``extra`` NMS
    survivors, i.e. the candidates the reference cut off, and the fg scores [n_post] of the rows it kept; None for a padded list).
    Input of ``compare_detector_outputs(..., ref_cutoff=)``.  Plain tensors in, plain tensors out: nothing of the oracle is imported."""
    out = []
    for d in rpn_debug["per_image"]:
        keep_all, roi_s, score_s = d["keep_all"], d["roi_sorted"], d["score_sorted"]
        if int(keep_all.numel()) <= n_post:
            out.append(None)
            continue
        ext = keep_all[n_post:n_post + extra]
        out.append((roi_s[ext].float(), score_s[ext].float(), score_s[keep_all[:n_post]].float()))
    return out


def compare_detector_outputs(got, ref, atol=1e-3, ref_cutoff=None, tie_tol=1e-6):
    """got / ref: (roi_cls_locs [B,R,4n], roi_scores [B,R,n], rois [B,R,4], roi_indices [B]) on CPU.

    Bars (north star): boxes and scores within ``atol`` absolute, arg-max class indices bit-exact.
    Rows are first compared position-wise (``rows_positional_mismatch``).  A discrete decision that
    flips between the two f32 pipelines (two scores closer than their rounding noise swap in the
    sort; an IoU lands within an ulp of the threshold) reorders or shifts the RoI list without changing
    what is computed per RoI, so rows are then matched as MULTISETS per image: a one-to-one pairing
    (every row of either side is used at most once: a row at its own position first, then the nearest
    reference row not taken yet, within ``atol``) in which every pair must meet the bars.
    ``rows_unmatched`` counts the rows left without a partner - by the bijection the same number on
    both sides, so an oracle RoI the GPU replaced by a duplicate of another one is counted - and ``ok``
    allows none.

    ``ref_cutoff`` (``cutoff_candidates`` of the reference's RPN debug record): the one flip the multiset cannot absorb is a
    score tie AT THE CUT of the list - the proposal layer keeps the R best survivors, and when the R-th and the (R+1)-th differ
    by a few ulps of their fg probability (config 4 on bench.py's input: 0.99836999 against 0.99836987, two ulps) either may
    take the last place.  A row of ``got`` without a partner is then paired with one of the candidates the reference cut off if
    it IS that candidate (within ``atol``) and the candidate's score is within ``tie_tol`` of a reference row that is without a
    partner too (the one it displaced).  Such pairs are counted in ``rows_tied_at_cutoff`` (with ``max_tie_score_gap``), leave
    ``rows_unmatched``, and their head outputs are not compared (the reference never computed that RoI's)."""
    g_locs, g_scores, g_rois, g_idx = got
    r_locs, r_scores, r_rois, r_idx = ref
    rep = {"shapes_equal": all(tuple(a.shape) == tuple(b.shape) for a, b in zip(got, ref))}
    if not rep["shapes_equal"]:
        rep["ok"] = False
        rep["shapes"] = [(tuple(a.shape), tuple(b.shape)) for a, b in zip(got, ref)]
        return rep
    B, R, _ = g_rois.shape
    rep["roi_indices_equal"] = bool(torch.equal(g_idx.cpu().long(), r_idx.cpu().long()))
    rep["rows"] = B * R
    rep["rows_positional_mismatch"] = int(((g_rois - r_rois).abs().amax(dim=-1) > atol).sum())
    unmatched, max_roi, max_score, max_loc, cls_bad = 0, 0.0, 0.0, 0.0, 0
    tied, tie_gap = 0, 0.0            # rows explained as score ties at the list's cut (ref_cutoff)
    nearest_unmatched = 0.0            # how far the worst row without a partner is from the nearest free reference row
    inf = float("inf")
    for b in range(B):
        d = (g_rois[b].unsqueeze(1) - r_rois[b].unsqueeze(0)).abs().amax(-1)      # [R,R]
        d = torch.where(torch.isfinite(d), d, torch.full_like(d, inf))            # (a NaN box matches nothing)
        diag = torch.arange(R)
        same = d[diag, diag] <= atol
        partner = torch.where(same, diag, torch.full_like(diag, -1))              # GPU row i -> reference row partner[i]
        taken = same.clone()                                                       # reference rows already paired
        for i in torch.nonzero(~same).flatten().tolist():                          # the few rows that moved
            row = torch.where(taken, torch.full_like(d[i], inf), d[i])
            j = int(row.argmin())
            if float(row[j]) <= atol:
                partner[i] = j
                taken[j] = True
            else:
                nearest_unmatched = max(nearest_unmatched, float(row[j]))
        ok = partner >= 0
        n_un = int((~ok).sum())
        if n_un and ref_cutoff is not None and ref_cutoff[b] is not None:
            ext_roi, ext_sc, kept_sc = ref_cutoff[b]
            free_ref = [j for j in range(R) if not bool(taken[j])]                  # reference rows nobody paired with
            used_ext = set()
            for i in torch.nonzero(~ok).flatten().tolist():
                de = (ext_roi - g_rois[b, i].unsqueeze(0)).abs().amax(-1)
                for c in torch.argsort(de).tolist():
                    if float(de[c]) > atol:
                        break
                    if c in used_ext:
                        continue
                    gaps = [(abs(float(kept_sc[j]) - float(ext_sc[c])), j) for j in free_ref]
                    if gaps and min(gaps)[0] <= tie_tol:
                        gap, j = min(gaps)
                        free_ref.remove(j)
                        used_ext.add(c)
                        tied += 1
                        tie_gap = max(tie_gap, gap)
                        n_un -= 1
                        break
        unmatched += n_un
        if ok.any():
            gi, ri = diag[ok], partner[ok]
            max_roi = max(max_roi, float(d[gi, ri].max()))
            max_score = max(max_score, float((g_scores[b, gi] - r_scores[b, ri]).abs().max()))
            max_loc = max(max_loc, float((g_locs[b, gi] - r_locs[b, ri]).abs().max()))
            cls_bad += int((g_scores[b, gi].argmax(-1) != r_scores[b, ri].argmax(-1)).sum())
    rep.update(rows_unmatched=unmatched, max_abs_roi=max_roi, max_abs_score=max_score, max_abs_cls_loc=max_loc,
               class_mismatch=cls_bad, matching="one-to-one", nearest_unmatched=nearest_unmatched,
               rows_tied_at_cutoff=tied, max_tie_score_gap=tie_gap)
    top2 = r_scores.topk(2, dim=-1).values
    rep["min_top2_logit_gap"] = float((top2[..., 0] - top2[..., 1]).min())
    rep["ok"] = bool(rep["roi_indices_equal"] and unmatched == 0 and cls_bad == 0
                     and max_score <= atol and max_loc <= atol and max_roi <= atol)
    return rep
