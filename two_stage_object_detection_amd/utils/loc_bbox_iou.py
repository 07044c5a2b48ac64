"""Box arithmetic with the reference's function surface (utils/loc_bbox_iou.py), on HIP kernels.

``bbox_iou`` (reference :4-27) and ``loc2bbox`` (:29-61) are on the inference path; ``bbox2loc`` (:63-88) serves the
training-side target creators and is the same device function they call.  ``xywh2xyxy`` (:91-97, a
list helper of the data loader) is out of this path's scope.
"""
from __future__ import annotations

import torch

from .. import hip_ops


def bbox_iou(bbox_a, bbox_b):
    """Pairwise IoU [n_a, n_b] with +1e-8 in the denominator; IndexError unless both are [*,4]."""
    if bbox_a.shape[1] != 4 or bbox_b.shape[1] != 4:
        raise IndexError
    return hip_ops.bbox_iou(bbox_a, bbox_b, 1e-8)


def loc2bbox(src_bbox, loc):
    """Apply (dx,dy,dw,dh) offsets to xyxy boxes.  Empty input returns a [0,4] tensor."""
    if src_bbox.size()[0] == 0:
        return torch.zeros((0, 4), dtype=loc.dtype, device=src_bbox.device)
    if loc.shape[1] != 4:
        # the reference supports [n, 4k] (k box sets per row); map it onto the [n*k, 4] kernel
        k = loc.shape[1] // 4
        out = hip_ops.loc2bbox(src_bbox.repeat_interleave(k, dim=0), loc.reshape(-1, 4))
        return out.view(loc.shape[0], 4 * k)
    return hip_ops.loc2bbox(src_bbox, loc)


def bbox2loc(src_bbox, dst_bbox):
    """Offsets (dx,dy,dw,dh) that take ``src_bbox`` [n,4] to ``dst_bbox`` [n,4] (widths / heights floored at f32 eps):
    ``loc2bbox(src, bbox2loc(src, dst)) == dst`` (the reference's own known answer, utils/loc_bbox_iou.py:103)."""
    if src_bbox.shape[1] != 4 or dst_bbox.shape[1] != 4:
        raise IndexError
    return hip_ops.bbox2loc(src_bbox, dst_bbox)
