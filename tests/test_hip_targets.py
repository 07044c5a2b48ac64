"""GPU parity of the training-side box ops through the C ABI: AnchorTargetCreator /
ProposalTargetCreator (nets/frcnn_training.py:19-177) against vectors the REFERENCE's own classes produced
(tests/golden/targets_*.npz) and against the oracle on larger seeded inputs (full-size anchor grids, many gt boxes,
exact IoU ties).  Bars: labels, kept sets and assignments bit-exact (integer work); offsets within 1e-5 (logf)."""
import ast
import os

import numpy as np
import pytest
import torch

from oracle import targets as oracle_targets
import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def creators():
    from two_stage_object_detection_amd.nets import frcnn_training
    return frcnn_training


def _boxes(g, n, span_x, span_y, lo, hi):
    xy = torch.rand(n, 2, generator=g) * torch.tensor([span_x, span_y])
    wh = torch.rand(n, 2, generator=g) * (hi - lo) + lo
    return torch.cat([xy, xy + wh], dim=1)


def _loc_close(got, ref, tol=1e-5):
    got, ref = got.cpu(), ref.cpu()
    fin = torch.isfinite(ref)
    assert torch.equal(torch.isfinite(got), fin)
    assert (got[fin] - ref[fin]).abs().max().item() <= tol if fin.any() else True


@pytest.mark.parametrize("name", ["default", "dup", "many_pos", "all_pos_ratio", "no_gt"])
def test_anchor_targets_vs_reference_vectors(dev, creators, golden_dir, name):
    z = np.load(os.path.join(golden_dir, "targets_anchor.npz"))
    kw = ast.literal_eval(str(z[f"{name}.kw"]))
    loc, label = creators.AnchorTargetCreator(**kw)(torch.from_numpy(z[f"{name}.bbox"]).to(dev), torch.from_numpy(z["anchor"]).to(dev))
    assert label.dtype == torch.int64 and np.array_equal(label.cpu().numpy(), z[f"{name}.label"])
    _loc_close(loc, torch.from_numpy(z[f"{name}.loc"]))


@pytest.mark.parametrize("name", ["default", "few", "no_gt", "thresholds", "thresholds_gap", "index_error"])
def test_proposal_targets_vs_reference_vectors(dev, creators, golden_dir, name):
    z = np.load(os.path.join(golden_dir, "targets_proposal.npz"))
    kw = ast.literal_eval(str(z[f"{name}.kw"]))
    args = [torch.from_numpy(z[f"{name}.{k}"]).to(dev) for k in ("roi", "bbox", "label")]
    if bool(z[f"{name}.raises"]):
        with pytest.raises(IndexError):
            creators.ProposalTargetCreator(**kw)(*args)
        return
    s_roi, s_loc, s_lab = creators.ProposalTargetCreator(**kw)(*args)
    assert np.array_equal(s_roi.cpu().numpy(), z[f"{name}.sample_roi"])            # gathered rows: exact
    assert s_lab.dtype == torch.int64 and np.array_equal(s_lab.cpu().numpy(), z[f"{name}.gt_roi_label"])
    _loc_close(s_loc, torch.from_numpy(z[f"{name}.gt_roi_loc"]))


@pytest.mark.parametrize("hw,stride,G,seed", [((50, 84), 16, 12, 1), ((25, 42), 32, 40, 2), ((50, 84), 16, 300, 3), ((7, 9), 16, 3, 4)])
def test_anchor_targets_full_size_vs_oracle(dev, creators, hw, stride, G, seed):
    """BASELINE geometries: 37 800 anchors (HarDNet stride 16) and 9 450 (ResNet-50 stride 32) at 800x1333, up to 300 gt
    boxes (more than one LDS chunk of the row arg-max); a quarter of the gt boxes are snapped to the anchor grid so that
    exact IoU ties occur (first-maximum rule in both directions, last-gt-wins override)."""
    from two_stage_object_detection_amd import hip_ops
    g = torch.Generator().manual_seed(seed)
    anchor = oracle.enumerate_shifted_anchor(oracle.generate_basic_anchor(), stride, hw[0], hw[1])
    bbox = _boxes(g, G, hw[1] * stride * 0.9, hw[0] * stride * 0.9, 24, 400)
    k = max(1, G // 4)
    bbox[:k] = anchor[torch.randint(0, anchor.shape[0], (k,), generator=g)]         # IoU exactly 1 with one anchor, ties elsewhere
    if G > 8:
        bbox[k:k + 2] = bbox[:2]                                                      # duplicated gt boxes
    for kw in (dict(), dict(pos_iou_thresh=0.5, neg_iou_thresh=0.4, n_sample=64)):
        ref_loc, ref_label, dbg = oracle_targets.anchor_targets(bbox, anchor, return_debug=True, **kw)
        c = creators.AnchorTargetCreator(**kw)
        loc, label = c(bbox.to(dev), anchor.to(dev))
        assert torch.equal(label.cpu(), ref_label)
        _loc_close(loc, ref_loc)
        _, _, argmax = hip_ops.anchor_targets(bbox.to(dev), anchor.to(dev), int(c.pos_ratio * c.n_sample), c.n_sample,
                                              c.pos_iou_thresh, c.neg_iou_thresh)
        assert torch.equal(argmax.cpu().long(), dbg["argmax_ious"].long())          # assignment incl. the override: exact


@pytest.mark.parametrize("R,G,seed", [(600, 20, 5), (300, 1, 6), (2000, 64, 7), (5, 3, 8)])
def test_proposal_targets_vs_oracle(dev, creators, R, G, seed):
    g = torch.Generator().manual_seed(seed)
    bbox = _boxes(g, G, 1000, 600, 30, 300)
    label = torch.randint(0, 80, (G,), generator=g)
    roi = _boxes(g, R, 1200, 700, 16, 350)
    n_near = min(R // 3, 50)                                                          # near-gt RoIs -> positives, but <= 64 up front
    roi[torch.arange(n_near) * 3] = bbox[torch.arange(n_near) % G] + torch.randn(n_near, 4, generator=g) * 3
    for kw in (dict(), dict(n_sample=96, pos_ratio=0.25, pos_iou_thresh=0.45, neg_iou_thresh_high=0.45)):
        try:
            ref = oracle_targets.proposal_targets(roi, bbox, label, **kw)
        except IndexError:
            with pytest.raises(IndexError):
                creators.ProposalTargetCreator(**kw)(roi.to(dev), bbox.to(dev), label.to(dev))
            continue
        got = creators.ProposalTargetCreator(**kw)(roi.to(dev), bbox.to(dev), label.to(dev))
        assert torch.equal(got[0].cpu(), ref[0]) and torch.equal(got[2].cpu(), ref[2])
        _loc_close(got[1], ref[1])


def test_utils_bbox2loc_surface(dev, golden_dir):
    """utils.loc_bbox_iou.bbox2loc (reference :63-88) through tsod_bbox2loc_f32: against the oracle restatement on random and
    degenerate boxes (zero-width sources are floored at f32 eps like the reference), the reference's own known answer
    loc2bbox(d1, bbox2loc(d1, d2)) == d2 (:103, stored by make_golden.py), empty input, and the [*,4] IndexError."""
    from oracle import targets as otargets
    from two_stage_object_detection_amd.utils.loc_bbox_iou import bbox2loc, loc2bbox
    g = torch.Generator().manual_seed(8)
    xy = torch.rand(5000, 2, generator=g) * 700
    src = torch.cat([xy, xy + torch.rand(5000, 2, generator=g) * 300 + 0.5], dim=1)
    xy2 = torch.rand(5000, 2, generator=g) * 700
    dst = torch.cat([xy2, xy2 + torch.rand(5000, 2, generator=g) * 300 + 0.5], dim=1)
    src[7, 2] = src[7, 0]                                   # zero width: floored at eps
    src[9, 3] = src[9, 1]
    ref = otargets.bbox2loc(src, dst)
    got = bbox2loc(src.to(dev), dst.to(dev)).cpu()
    fin = torch.isfinite(ref)
    assert torch.equal(torch.isfinite(got), fin)
    rel = ((got - ref).abs() / ref.abs().clamp_min(1.0))[fin]
    assert rel.max().item() <= 2e-6, rel.max().item()     # division / log: a few ulp between libm and the device's
    z = np.load(os.path.join(golden_dir, "boxmath.npz"))
    d1 = torch.tensor([[100., 100., 200., 200.]])
    d2 = torch.tensor([[150., 150., 250., 250.]])
    back = loc2bbox(d1.to(dev), bbox2loc(d1.to(dev), d2.to(dev))).cpu()
    assert (back - d2).abs().max().item() <= 1e-4           # the reference's round trip (exact on the CPU; exp/log ulps here)
    assert np.abs(back.numpy() - z["known_roundtrip"]).max() <= 1e-4   # what the REFERENCE's own two functions return
    assert bbox2loc(torch.zeros(0, 4, device=dev), torch.zeros(0, 4, device=dev)).shape == (0, 4)
    with pytest.raises(IndexError):
        bbox2loc(torch.zeros(3, 5, device=dev), torch.zeros(3, 4, device=dev))
