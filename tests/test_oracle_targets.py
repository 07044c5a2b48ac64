"""oracle/targets.py against the reference's own AnchorTargetCreator / ProposalTargetCreator (nets/frcnn_training.py:19-177)
run in the build container by tests/golden/make_golden.py: labels / kept sets bit-exact, offsets bit-exact (same torch CPU
ops in the same order), the IndexError of quirk T2 where the reference raises."""
import ast
import os

import numpy as np
import pytest
import torch

from oracle import targets

ANCHOR_CASES = ["default", "dup", "many_pos", "all_pos_ratio", "no_gt"]
PROPOSAL_CASES = ["default", "few", "no_gt", "thresholds", "thresholds_gap", "index_error"]


@pytest.mark.parametrize("name", ANCHOR_CASES)
def test_anchor_targets_match_the_reference(golden_dir, name):
    z = np.load(os.path.join(golden_dir, "targets_anchor.npz"))
    kw = ast.literal_eval(str(z[f"{name}.kw"]))
    loc, label = targets.anchor_targets(torch.from_numpy(z[f"{name}.bbox"]), torch.from_numpy(z["anchor"]), **kw)
    assert label.dtype == torch.int64 and np.array_equal(label.numpy(), z[f"{name}.label"])
    assert np.array_equal(loc.numpy(), z[f"{name}.loc"])
    if name == "many_pos":
        assert int((label == 1).sum()) == 128                       # the positive cap keeps the first 128 by index
        assert int((label == 0).sum()) > 128                        # T1: negatives are never subsampled
    if name == "all_pos_ratio":
        assert int((label == 0).sum()) == 0 and int((label == 1).sum()) == 16    # T1's other face: n_neg == 0 drops them all
    if name == "no_gt":
        assert (label == 0).all() and (loc == 0).all()


@pytest.mark.parametrize("name", PROPOSAL_CASES)
def test_proposal_targets_match_the_reference(golden_dir, name):
    z = np.load(os.path.join(golden_dir, "targets_proposal.npz"))
    kw = ast.literal_eval(str(z[f"{name}.kw"]))
    args = (torch.from_numpy(z[f"{name}.roi"]), torch.from_numpy(z[f"{name}.bbox"]), torch.from_numpy(z[f"{name}.label"]))
    if bool(z[f"{name}.raises"]):
        with pytest.raises(IndexError):
            targets.proposal_targets(*args, **kw)
        return
    s_roi, s_loc, s_lab = targets.proposal_targets(*args, **kw)
    assert np.array_equal(s_roi.numpy(), z[f"{name}.sample_roi"])
    assert np.array_equal(s_loc.numpy(), z[f"{name}.gt_roi_loc"])
    assert s_lab.dtype == torch.int64 and np.array_equal(s_lab.numpy(), z[f"{name}.gt_roi_label"])


def test_bbox2loc_roundtrip_known_answer():
    """utils/loc_bbox_iou.py:103: loc2bbox(d1, bbox2loc(d1, d2)) == d2 exactly."""
    from oracle import loc2bbox
    d1 = torch.tensor([[100., 100, 200, 200]])
    d2 = torch.tensor([[150., 150, 250, 250]])
    assert torch.equal(loc2bbox(d1, targets.bbox2loc(d1, d2)), d2)
