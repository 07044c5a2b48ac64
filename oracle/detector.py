"""oracle/detector.py -- TEST INFRASTRUCTURE ONLY (CPU restatement; never on the product path).

Composition of the detector forward.  ``nets/frcnn.py`` is dead code in the reference
(SURVEY 0.3); the semantics follow the one working wiring, FasterRCNNTrainer
(nets/frcnn_training.py:203-217, 251-260, 289-298), with the decisions of SURVEY 7.0:

* D1  backbone "resnet50" = resnet50(include_top=False) trunk, feat_stride 32, C_f 2048;
      "hardnet39"/"hardnet68" = HarDNetFeatureExtraction(depth_wise=True), stride 16, C_f 512.
* D3  the RPN receives img_size = x.shape[1:] = (C,H,W)  (frcnn_training.py:252, quirk Q1),
      the head receives img_size = x.shape[2:] = (H,W)   (nets/frcnn.py:33,39, quirk Q2).
* Q6  roi_indices = arange(B) (frcnn_training.py:291).
"""
from __future__ import annotations

import torch

from .backbones import resnet_trunk, hardnet_trunk
from .box import rpn_forward, roi_head_forward, loc2bbox

BACKBONES = {"resnet50": (32, 2048), "hardnet39": (16, 512), "hardnet68": (16, 512)}


def extractor_forward(sd, x, backbone="resnet50", prefix="extractor."):
    if backbone == "resnet50":
        return resnet_trunk(sd, x, prefix=prefix)
    if backbone in ("hardnet39", "hardnet68"):
        return hardnet_trunk(sd, x, arch=int(backbone[-2:]), prefix=prefix)
    raise ValueError(backbone)


@torch.inference_mode()
def detector_forward(sd, x, backbone="resnet50", scale=1.0, mode="training",
                     ratios=(0.5, 1, 2), anchor_scales=(8, 16, 32), return_debug=False, roi_op="pool", exact=False):
    """FasterRCNN.forward(x, scale, mode="forward") (nets/frcnn.py:30-40) ->
    (roi_cls_locs [B,R,4*n_class], roi_scores [B,R,n_class], rois [B,R,4], roi_indices [B]).

    ``exact=False`` (default): the reference's arithmetic - torch CPU float32 throughout - bit for bit.
    ``exact=True``: the SAME network evaluated in float64 wherever the reference sums (trunk, RPN convs, the head's mean and two
    Linear layers), every tensor that enters the reference's f32 box code (RPN loc / score, the feature map RoI pooling picks
    from, the head's outputs) rounded ONCE to float32; decode, clamp, sort, NMS, padding and pooling are the reference's own
    float32 operations.  That is the exact value of the reference's math up to one rounding: what a float32 pipeline - the
    reference's own CPU run included - can be measured AGAINST (scripts/config4_truth.py: on HarDNet-68 at 3x800x1333 the
    reference's f32 path is 11-12 RoI ulps from it, as far as any of the HIP arithmetics)."""
    stride, _ = BACKBONES[backbone]
    if exact:
        sd = {k: (v.double() if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in sd.items()}
        x = x.double()
    feat = extractor_forward(sd, x, backbone)
    rpn = rpn_forward(sd, feat, tuple(x.shape[1:]), scale=scale, feat_stride=stride, mode=mode,
                      ratios=ratios, anchor_scales=anchor_scales, prefix="rpn.", return_debug=return_debug)
    dbg = None
    if return_debug:
        rpn, dbg = rpn
    rpn_locs, rpn_scores, rois, anchor = rpn
    roi_indices = torch.arange(x.shape[0], dtype=torch.int32)
    cls_locs, scores = roi_head_forward(sd, feat, rois, roi_indices, tuple(x.shape[2:]), prefix="head.", roi_op=roi_op)
    out = (cls_locs, scores, rois, roi_indices)
    if return_debug:
        dbg = dict(dbg, feat=feat, rpn_locs=rpn_locs, rpn_scores=rpn_scores, anchor=anchor)
        return out, dbg
    return out


def detections_from_outputs(roi_cls_locs, roi_scores, rois):
    """class = argmax over all n_class logits (first max wins), score = that raw
    logit (never softmaxed, Q11: nets/frcnn_training.py:319), box = loc2bbox(roi, loc of
    the arg-max class) as in nets/frcnn_training.py:311-318.
    Returns [B,R,6] f32 rows (x1,y1,x2,y2,score,class)."""
    B, R, n_class = roi_scores.shape
    score, cls = torch.max(roi_scores, dim=2)
    locs = roi_cls_locs.view(B, R, n_class, 4)
    sel = torch.gather(locs, 2, cls.view(B, R, 1, 1).expand(B, R, 1, 4)).squeeze(2)
    boxes = loc2bbox(rois.reshape(-1, 4), sel.reshape(-1, 4)).view(B, R, 4)
    return torch.cat([boxes, score.unsqueeze(-1), cls.to(boxes.dtype).unsqueeze(-1)], dim=-1)


def postprocess(det, iou_threshold=0.1, score_thresh=None, per_class=False, background_class=-1):
    """multi_inference.py:80-87 for a batch of detection records [B,R,6]: per image
    ``keep = nms(boxes_pred, labels_score_pred, iou_threshold)`` (class-agnostic, no score threshold: the defaults).
    ``score_thresh`` / ``background_class`` drop records first; ``per_class`` runs the same nms once per class and
    merges the survivors by score (what torchvision's ``batched_nms`` computes, without its coordinate-offset
    shortcut).  Returns a list of kept record tensors, each in descending-score order (stable: ties keep RoI order)."""
    from .box import nms
    out = []
    for d in det:
        ok = torch.ones(d.shape[0], dtype=torch.bool)
        if score_thresh is not None:
            ok &= d[:, 4] >= score_thresh
        ok &= ~torch.isnan(d[:, 4]) & (d[:, 4] > float("-inf"))
        if background_class >= 0:
            ok &= d[:, 5] != float(background_class)
        d = d[ok]
        order = torch.sort(d[:, 4], descending=True, stable=True).indices
        d = d[order]
        if not per_class:
            keep = nms(d[:, :4], d[:, 4], iou_threshold)
        else:
            keeps = []
            for c in torch.unique(d[:, 5]).tolist():
                rows = torch.nonzero(d[:, 5] == c).squeeze(1)
                keeps.append(rows[nms(d[rows, :4], d[rows, 4], iou_threshold)])
            keep = torch.sort(torch.cat(keeps)).values if keeps else torch.zeros(0, dtype=torch.long)
        out.append(d[keep])
    return out
