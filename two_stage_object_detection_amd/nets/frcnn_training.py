"""Training-side box ops with the reference's class surface (nets/frcnn_training.py:19-177) on HIP kernels.

Only the two target creators are provided - SURVEY 8(f) rank 4; the trainer, the losses and the mAP code of that file
are outside this repository's path.  Both classes are deterministic in the reference (they keep "the first n by
index", there is no random sampling) and are reproduced with their indexing quirks (oracle/targets.py T1-T4, pinned by
fixtures the reference's own classes produced):

    AnchorTargetCreator(n_sample, pos_iou_thresh, neg_iou_thresh, pos_ratio)(bbox, anchor) -> (loc [A,4], label [A] int64)
    ProposalTargetCreator(n_sample, pos_ratio, pos_iou_thresh, neg_iou_thresh_high, neg_iou_thresh_low)
        (roi, bbox, label, loc_normalize_std) -> (sample_roi [S,4], gt_roi_loc [S,4], gt_roi_label [S] int64)
"""
from __future__ import annotations

import torch

from .. import hip_ops
from .._ffi import require_cuda


class AnchorTargetCreator:
    def __init__(self, n_sample=256, pos_iou_thresh=0.7, neg_iou_thresh=0.3, pos_ratio=0.5):
        self.n_sample = n_sample
        self.pos_iou_thresh = pos_iou_thresh
        self.neg_iou_thresh = neg_iou_thresh
        self.pos_ratio = pos_ratio

    def __call__(self, bbox, anchor):
        """bbox [G,4] ground truth, anchor [A,4] -> (loc [A,4], label [A]: 1 positive / 0 negative / -1 ignored).
        Four launches (row arg-max, column arg-max, labels + positive cap, offsets), no host synchronisation."""
        require_cuda(anchor, "AnchorTargetCreator")
        loc, label, _ = hip_ops.anchor_targets(bbox, anchor, int(self.pos_ratio * self.n_sample), self.n_sample,
                                               self.pos_iou_thresh, self.neg_iou_thresh)
        return loc, label


class ProposalTargetCreator(object):
    def __init__(self, n_sample=128, pos_ratio=0.5, pos_iou_thresh=0.5, neg_iou_thresh_high=0.5, neg_iou_thresh_low=0):
        self.n_sample = n_sample
        self.pos_ratio = pos_ratio
        self.pos_roi_per_image = int(self.n_sample * self.pos_ratio)
        self.pos_iou_thresh = pos_iou_thresh
        self.neg_iou_thresh_high = neg_iou_thresh_high
        self.neg_iou_thresh_low = neg_iou_thresh_low

    def __call__(self, roi, bbox, label, loc_normalize_std=(0.1, 0.1, 0.2, 0.2)):
        """roi [R,4], bbox [G,4], label [G] -> (sample_roi [S,4], gt_roi_loc [S,4], gt_roi_label [S]), S <= n_sample.
        ``loc_normalize_std`` is accepted and unused, as in the reference (the division is commented out there).
        Two launches; reading S (and the IndexError flag of the reference's quirk T2) is the one host synchronisation."""
        require_cuda(roi, "ProposalTargetCreator")
        sample_roi, gt_roi_loc, gt_roi_label, counts = hip_ops.proposal_targets(
            roi, bbox, label, self.n_sample, self.pos_roi_per_image, self.pos_iou_thresh, self.neg_iou_thresh_high,
            self.neg_iou_thresh_low)
        n_keep, _, _, status = counts.tolist()
        if status:
            raise IndexError("index of a sampled negative is out of bounds for the kept labels "
                             "(the reference raises IndexError at nets/frcnn_training.py:175)")
        return sample_roi[:n_keep], gt_roi_loc[:n_keep], gt_roi_label[:n_keep].to(label.dtype)
