"""oracle/targets.py -- TEST INFRASTRUCTURE ONLY (CPU restatement; never on the product path).

Training-side box ops of the reference: ``AnchorTargetCreator`` (nets/frcnn_training.py:19-103)
and ``ProposalTargetCreator`` (:105-177), restated with torch CPU ops in the reference's op order.  Both are
DETERMINISTIC in the reference (no random sampling: "the first n by index" is kept) and both carry indexing quirks that
are restated as they are, because the fixtures (tests/golden/targets_*.npz, produced by the reference's own classes in
make_golden.py) pin them:

* T1  negatives of the anchor targets are never subsampled: ``if len(neg_index) > n_neg`` (:96) measures the length of the
      TUPLE torch.where returns (= 1), so the branch only fires when n_neg < 1, and then disables ALL negatives.
* T2  ``gt_roi_label[neg_index] = 0`` (:175) indexes the KEPT list (<= n_sample rows) with positions of the ORIGINAL
      roi list: labels are zeroed at kept positions that happen to equal a sampled negative's original index (positives
      included), sampled negatives elsewhere keep ``label[assignment] + 1``, and an original index >= the kept length
      raises IndexError.
* T3  ``loc_normalize_std`` is accepted and unused (the division is commented out at :170).
* T4  torch.max(dim=1) / argmax(dim=0) return the FIRST maximum; the per-gt override loop (:61-63) runs in gt order,
      so the LAST gt claiming an anchor wins.
"""
from __future__ import annotations

import torch

from .box import bbox_iou


def bbox2loc(src_bbox: torch.Tensor, dst_bbox: torch.Tensor) -> torch.Tensor:
    """utils/loc_bbox_iou.py:63-88 (width / height floored at f32 eps)."""
    width = src_bbox[:, 2] - src_bbox[:, 0]
    height = src_bbox[:, 3] - src_bbox[:, 1]
    ctr_x = src_bbox[:, 0] + 0.5 * width
    ctr_y = src_bbox[:, 1] + 0.5 * height
    base_width = dst_bbox[:, 2] - dst_bbox[:, 0]
    base_height = dst_bbox[:, 3] - dst_bbox[:, 1]
    base_ctr_x = dst_bbox[:, 0] + 0.5 * base_width
    base_ctr_y = dst_bbox[:, 1] + 0.5 * base_height
    eps = torch.finfo(height.dtype).eps
    width = torch.maximum(width, torch.tensor(eps))
    height = torch.maximum(height, torch.tensor(eps))
    dx = (base_ctr_x - ctr_x) / width
    dy = (base_ctr_y - ctr_y) / height
    dw = torch.log(base_width / width)
    dh = torch.log(base_height / height)
    return torch.vstack((dx, dy, dw, dh)).T


def anchor_targets(bbox: torch.Tensor, anchor: torch.Tensor, n_sample=256, pos_iou_thresh=0.7, neg_iou_thresh=0.3,
                   pos_ratio=0.5, return_debug=False):
    """AnchorTargetCreator(n_sample, pos_iou_thresh, neg_iou_thresh, pos_ratio)(bbox, anchor)
    (nets/frcnn_training.py:28-41, 43-68, 70-103) -> (loc [A,4] f32, label [A] int64 in {-1, 0, 1})."""
    A = anchor.shape[0]
    # _calc_ious (:43-68)
    ious = bbox_iou(anchor, bbox)
    if bbox.shape[0] == 0:
        argmax_ious = torch.zeros(A, dtype=torch.int32)
        max_ious = torch.zeros(A, dtype=anchor.dtype)
        gt_argmax_ious = torch.zeros(0, dtype=bbox.dtype)
    else:
        max_ious, argmax_ious = torch.max(ious, dim=1)
        gt_argmax_ious = ious.argmax(dim=0)
        for i in range(len(gt_argmax_ious)):                       # T4: later gts override earlier ones
            argmax_ious[gt_argmax_ious[i]] = i
    # _create_label (:70-103)
    label = torch.empty((A,), dtype=torch.int64).fill_(-1)
    label[max_ious < neg_iou_thresh] = 0
    label[max_ious >= pos_iou_thresh] = 1
    if len(gt_argmax_ious) > 0:
        label[gt_argmax_ious] = 1
    n_pos = int(pos_ratio * n_sample)
    pos_index = torch.where(label == 1)
    pos_length = pos_index[0].numel()
    if pos_length > n_pos:
        label[tuple(i[n_pos:] for i in pos_index)] = -1
        pos_length = n_pos
    n_neg = n_sample - pos_length
    neg_index = torch.where(label == 0)
    if len(neg_index) > n_neg:                                     # T1: len of the 1-tuple
        label[tuple(i[n_neg:] for i in neg_index)] = -1
    # __call__ (:28-41)
    if (label > 0).any():
        loc = bbox2loc(anchor, bbox[argmax_ious])
    else:
        loc = torch.zeros_like(anchor)
    if return_debug:
        return loc, label, dict(argmax_ious=argmax_ious, max_ious=max_ious, gt_argmax_ious=gt_argmax_ious)
    return loc, label


def proposal_targets(roi: torch.Tensor, bbox: torch.Tensor, label: torch.Tensor, loc_normalize_std=(0.1, 0.1, 0.2, 0.2),
                     n_sample=128, pos_ratio=0.5, pos_iou_thresh=0.5, neg_iou_thresh_high=0.5, neg_iou_thresh_low=0):
    """ProposalTargetCreator(...)(roi, bbox, label, loc_normalize_std) (nets/frcnn_training.py:123-177) ->
    (sample_roi [S,4], gt_roi_loc [S,4], gt_roi_label [S]); S <= n_sample.  Raises IndexError where the reference does (T2)."""
    pos_roi_per_image = int(n_sample * pos_ratio)
    roi = torch.cat((roi, bbox), dim=0)
    iou = bbox_iou(roi, bbox)
    if len(bbox) == 0:
        gt_assignment = torch.zeros(len(roi), dtype=torch.int32)
        max_iou = torch.zeros(len(roi), dtype=roi.dtype)
        gt_roi_label = torch.zeros(len(roi)).type_as(label)
    else:
        max_iou, gt_assignment = torch.max(iou, dim=1)
        gt_roi_label = label[gt_assignment] + 1
    pos_index = torch.where(max_iou >= pos_iou_thresh)
    pos_length = pos_index[0].numel()
    if pos_length > pos_roi_per_image:
        pos_index = tuple(i[:pos_roi_per_image] for i in pos_index)
        pos_length = pos_index[0].numel()
    neg_index = torch.where((max_iou < neg_iou_thresh_high) & (max_iou >= neg_iou_thresh_low))
    neg_roi_per_this_image = n_sample - pos_length
    neg_length = neg_index[0].numel()
    if neg_length > neg_roi_per_this_image:
        neg_index = tuple(i[:neg_roi_per_this_image] for i in neg_index)
    keep_index = tuple(torch.cat((a, b), dim=0) for a, b in zip(pos_index, neg_index))
    sample_roi = roi[keep_index]
    if len(bbox) == 0:
        return sample_roi, torch.zeros_like(sample_roi), gt_roi_label[keep_index]
    gt_roi_loc = bbox2loc(sample_roi, bbox[gt_assignment[keep_index]])      # T3: no normalisation
    gt_roi_label = gt_roi_label[keep_index]
    gt_roi_label[neg_index] = 0                                              # T2
    return sample_roi, gt_roi_loc, gt_roi_label
