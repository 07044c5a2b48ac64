"""oracle/box.py -- TEST INFRASTRUCTURE ONLY (CPU restatement; never on the product path).

Box arithmetic, proposal layer, RPN glue and RoI-head glue of the reference, restated with
torch CPU tensor ops (the arithmetic library the reference itself uses) plus the C
restatement of the two torchvision operators (oracle/box_ops.c via ctypes).

Every function cites the reference lines it follows.  Op ORDER is kept so that every f32
rounding happens where the reference's does.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess

import numpy as np
import torch
import torch.nn.functional as F

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    """Load (building on first use) the C half of the oracle."""
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle_box.so")
        src = os.path.join(_HERE, "box_ops.c")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
            subprocess.check_call(["make", "-s", "-C", _HERE])
        lib = ctypes.CDLL(so)
        i64, f32 = ctypes.c_int64, ctypes.c_float
        p = ctypes.c_void_p
        lib.oracle_argsort_desc.argtypes = [p, i64, p]
        lib.oracle_argsort_desc.restype = ctypes.c_int
        lib.oracle_nms.argtypes = [p, p, i64, f32, p]
        lib.oracle_nms.restype = i64
        lib.oracle_nms_pad.argtypes = [p, i64, f32, i64, p]
        lib.oracle_nms_pad.restype = i64
        lib.oracle_roi_pool.argtypes = [p, i64, i64, i64, i64, p, i64, f32, i64, i64, p]
        lib.oracle_roi_pool.restype = ctypes.c_int
        lib.oracle_roi_align.argtypes = [p, i64, i64, i64, i64, p, i64, f32, i64, i64, i64, ctypes.c_int, p]
        lib.oracle_roi_align.restype = ctypes.c_int
        lib.oracle_bbox_iou.argtypes = [p, i64, p, i64, f32, p]
        lib.oracle_bbox_iou.restype = None
        _LIB = lib
    return _LIB


def _f32c(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to("cpu", torch.float32).contiguous()


# --------------------------------------------------------------------------- anchors
def generate_basic_anchor(base_size=8, ratios=(0.5, 1, 2), anchor_scales=(8, 16, 32)) -> torch.Tensor:
    """utils/basic_anchors.py:11-23.  Row = ratio-major, scale-minor; (-w/2,-h/2,w/2,h/2).

    h = base*scale*sqrt(f32(r)), w = base*scale*sqrt(f32(1/r)) with 1/r taken in double
    first (python float) exactly as the reference writes it.
    """
    rows = []
    for r in ratios:
        sr = torch.sqrt(torch.tensor(r, dtype=torch.float32))
        sir = torch.sqrt(torch.tensor(1.0 / r, dtype=torch.float32))
        for s in anchor_scales:
            h = base_size * s * sr
            w = base_size * s * sir
            rows.append(torch.stack([-w / 2.0, -h / 2.0, w / 2.0, h / 2.0]))
    return torch.stack(rows).to(torch.float32)


def enumerate_shifted_anchor(anchor_base: torch.Tensor, feat_stride: int, height: int, width: int) -> torch.Tensor:
    """utils/basic_anchors.py:27-57.  anchor[(y*W + x)*A + a] = base[a] + (x*s, y*s, x*s, y*s)."""
    base = anchor_base.to(torch.float32)
    sx = (torch.arange(width, dtype=torch.int64) * feat_stride).to(torch.float32)
    sy = (torch.arange(height, dtype=torch.int64) * feat_stride).to(torch.float32)
    gy, gx = torch.meshgrid(sy, sx, indexing="ij")          # [H,W], x fastest when flattened
    shift = torch.stack([gx, gy, gx, gy], dim=-1).reshape(-1, 1, 4)
    return (base.reshape(1, -1, 4) + shift).reshape(-1, 4)


# --------------------------------------------------------------------------- box math
def loc2bbox(src_bbox: torch.Tensor, loc: torch.Tensor) -> torch.Tensor:
    """utils/loc_bbox_iou.py:29-61 (empty-input guard :33-34).  ``loc`` is [n,4] or, as the reference's strided
    slices 0::4 .. 3::4 allow (:42-45, :55-58), [n,4k]: k offset sets per source box, output [n,4k]."""
    if src_bbox.shape[0] == 0:
        return torch.zeros((0, 4), dtype=loc.dtype)
    x1, y1, x2, y2 = (c.unsqueeze(-1) for c in src_bbox.to(loc.dtype).unbind(1))
    w = x2 - x1
    h = y2 - y1
    cx = x1 + 0.5 * w
    cy = y1 + 0.5 * h
    dx, dy, dw, dh = loc[:, 0::4], loc[:, 1::4], loc[:, 2::4], loc[:, 3::4]
    ncx = dx * w + cx
    ncy = dy * h + cy
    nw = torch.exp(dw) * w
    nh = torch.exp(dh) * h
    out = torch.zeros_like(loc)
    out[:, 0::4] = ncx - 0.5 * nw
    out[:, 1::4] = ncy - 0.5 * nh
    out[:, 2::4] = ncx + 0.5 * nw
    out[:, 3::4] = ncy + 0.5 * nh
    return out


def bbox_iou(bbox_a: torch.Tensor, bbox_b: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    """utils/loc_bbox_iou.py:4-27 (IndexError on last dim != 4 at :14-16)."""
    if bbox_a.shape[1] != 4 or bbox_b.shape[1] != 4:
        raise IndexError
    a, b = _f32c(bbox_a), _f32c(bbox_b)
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32)
    _lib().oracle_bbox_iou(a.data_ptr(), a.shape[0], b.data_ptr(), b.shape[0], eps, out.data_ptr())
    return out


# --------------------------------------------------------------------------- torchvision ops
def nms(boxes: torch.Tensor, scores: torch.Tensor, iou_threshold: float) -> torch.Tensor:
    """torchvision.ops.nms as called at nets/rpn.py:63 (C restatement, PARITY UNPINNED)."""
    b, s = _f32c(boxes), _f32c(scores)
    n = b.shape[0]
    keep = torch.empty(max(n, 1), dtype=torch.int64)
    k = _lib().oracle_nms(b.data_ptr(), s.data_ptr(), n, float(iou_threshold), keep.data_ptr())
    if k < 0:
        raise MemoryError
    return keep[:k].clone()


def nms_python(boxes, scores, thr):
    """Pure-Python loop version of ``nms`` (small cases only) used to cross-check the C code."""
    b = np.asarray(boxes, dtype=np.float32)
    s = np.asarray(scores, dtype=np.float32)
    order = sorted(range(len(s)), key=lambda i: (-float(s[i]), i))
    area = ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])).astype(np.float32)
    dead = [False] * len(s)
    keep = []
    thr = np.float32(thr)
    for oi, i in enumerate(order):
        if dead[i]:
            continue
        keep.append(i)
        for j in order[oi + 1:]:
            if dead[j]:
                continue
            w = max(np.float32(0), np.float32(min(b[i, 2], b[j, 2]) - max(b[i, 0], b[j, 0])))
            h = max(np.float32(0), np.float32(min(b[i, 3], b[j, 3]) - max(b[i, 1], b[j, 1])))
            inter = np.float32(w * h)
            with np.errstate(invalid="ignore", divide="ignore"):
                ovr = inter / np.float32(np.float32(area[i] + area[j]) - inter)
            if ovr > thr:
                dead[j] = True
    return keep


def roi_pool(x: torch.Tensor, rois: torch.Tensor, output_size=(7, 7), spatial_scale: float = 1.0) -> torch.Tensor:
    """torchvision.ops.RoIPool(output_size, spatial_scale)(x, rois) as built at
    nets/classify.py:17 and called at :43 (C restatement, PARITY UNPINNED).
    x [B,C,H,W] f32, rois [K,5] = (batch_idx, x1, y1, x2, y2) -> [K,C,PH,PW]."""
    xc, rc = _f32c(x), _f32c(rois)
    B, C, H, W = xc.shape
    K = rc.shape[0]
    PH, PW = output_size
    out = torch.empty((K, C, PH, PW), dtype=torch.float32)
    rcode = _lib().oracle_roi_pool(xc.data_ptr(), B, C, H, W, rc.data_ptr(), K,
                                   float(spatial_scale), PH, PW, out.data_ptr())
    if rcode != 0:
        raise IndexError("roi batch index out of range")
    return out


def roi_align(x: torch.Tensor, rois: torch.Tensor, output_size=(7, 7), spatial_scale: float = 1.0, sampling_ratio: int = 2,
              aligned: bool = False) -> torch.Tensor:
    """torchvision.ops.roi_align(x, rois, output_size, spatial_scale, sampling_ratio, aligned) (C restatement of the
    published CPU kernel, PARITY UNPINNED; checker of the added roi_op="align" option - the reference uses RoIPool)."""
    xc, rc = _f32c(x), _f32c(rois)
    B, C, H, W = xc.shape
    K = rc.shape[0]
    PH, PW = output_size
    out = torch.empty((K, C, PH, PW), dtype=torch.float32)
    if _lib().oracle_roi_align(xc.data_ptr(), B, C, H, W, rc.data_ptr(), K, float(spatial_scale), PH, PW,
                               int(sampling_ratio), 1 if aligned else 0, out.data_ptr()) != 0:
        raise IndexError("roi batch index out of range")
    return out


def roi_pool_python(x, rois, output_size=(7, 7), spatial_scale=1.0):
    """Pure-Python loop version of ``roi_pool`` (small cases only) to cross-check the C code."""
    x = np.asarray(x, dtype=np.float32)
    rois = np.asarray(rois, dtype=np.float32)
    PH, PW = output_size
    _, C, H, W = x.shape
    out = np.zeros((rois.shape[0], C, PH, PW), dtype=np.float32)

    def rnd(v):  # C round(): half away from zero
        v = float(np.float32(v))
        return int(math.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)

    s = np.float32(spatial_scale)
    for k, r in enumerate(rois):
        b = int(r[0])
        sw, sh, ew, eh = rnd(r[1] * s), rnd(r[2] * s), rnd(r[3] * s), rnd(r[4] * s)
        rw, rh = max(ew - sw + 1, 1), max(eh - sh + 1, 1)
        bh, bw = np.float32(rh) / np.float32(PH), np.float32(rw) / np.float32(PW)
        for ph in range(PH):
            hs = min(max(int(math.floor(np.float32(ph) * bh)) + sh, 0), H)
            he = min(max(int(math.ceil(np.float32(ph + 1) * bh)) + sh, 0), H)
            for pw in range(PW):
                ws = min(max(int(math.floor(np.float32(pw) * bw)) + sw, 0), W)
                we = min(max(int(math.ceil(np.float32(pw + 1) * bw)) + sw, 0), W)
                if he <= hs or we <= ws:
                    out[k, :, ph, pw] = 0
                else:
                    out[k, :, ph, pw] = x[b, :, hs:he, ws:we].reshape(C, -1).max(axis=1)
    return out


# --------------------------------------------------------------------------- proposal layer
_PROPOSAL_CFG = {  # nets/rpn.py:18-34 defaults
    "nms_iou": 0.7, "n_train_pre_nms": 12000, "n_train_post_nms": 600,
    "n_test_pre_nms": 3000, "n_test_post_nms": 300, "min_size": 16,
}


def proposal_counts(mode: str):
    """nets/rpn.py:37-42.  Only the literal string "train" selects the train numbers."""
    if mode == "train":
        return _PROPOSAL_CFG["n_train_pre_nms"], _PROPOSAL_CFG["n_train_post_nms"]
    return _PROPOSAL_CFG["n_test_pre_nms"], _PROPOSAL_CFG["n_test_post_nms"]


def proposal_layer(loc, score, anchor, img_size, scale=1.0, mode="training", return_debug=False):
    """ProposalCreator.__call__, nets/rpn.py:36-70, one image.

    loc [A,4], score [A] (fg probability), anchor [A,4]; img_size is indexed at [1] for the
    x clamp and [2] for the y clamp exactly as the reference does.
    Ties in ``score`` sort lower index first."""
    n_pre, n_post = proposal_counts(mode)
    roi = loc2bbox(anchor, loc)
    xs = roi[:, 0::2].clamp(min=0, max=img_size[1])
    ys = roi[:, 1::2].clamp(min=0, max=img_size[2])
    roi = torch.stack([xs[:, 0], ys[:, 0], xs[:, 1], ys[:, 1]], dim=1)
    min_size = _PROPOSAL_CFG["min_size"] * scale
    ok = ((roi[:, 2] - roi[:, 0]) >= min_size) & ((roi[:, 3] - roi[:, 1]) >= min_size)
    valid_idx = torch.nonzero(ok).squeeze(1)
    roi_v, score_v = roi[valid_idx], score[valid_idx]
    order = torch.sort(score_v, descending=True, stable=True).indices
    if n_pre > 0:
        order = order[:n_pre]
    roi_s, score_s = roi_v[order], score_v[order]
    keep = nms(roi_s, score_s, _PROPOSAL_CFG["nms_iou"])
    n_kept = int(keep.numel())
    keep_all = keep                  # (every survivor in score order: what lies just beyond the n_post cut, for the checker's tie rule)
    if n_kept < n_post:
        keep = torch.cat([keep, torch.arange(n_post - n_kept, dtype=keep.dtype)])
    keep = keep[:n_post]
    out = roi_s[keep]            # IndexError here if the pad runs past the candidates (Q4)
    if return_debug:
        return out, {"decoded": roi, "valid": ok, "sorted_src": valid_idx[order], "roi_sorted": roi_s,
                     "score_sorted": score_s, "keep": keep, "n_kept": n_kept, "keep_all": keep_all}
    return out


# --------------------------------------------------------------------------- RPN / head glue
def rpn_forward(sd, feat, img_size, scale=1.0, feat_stride=16, mode="training",
                ratios=(0.5, 1, 2), anchor_scales=(8, 16, 32), prefix="", return_debug=False):
    """RegionProposalNetwork.forward, nets/rpn.py:97-143.  ``sd`` holds
    ``{prefix}loc.weight/bias`` [4A,C,1,1] and ``{prefix}score.weight/bias`` [2A,C,1,1]
    (nets/rpn.py:86-88).  Returns (rpn_locs [B,A,4], rpn_scores [B,A,2], rois [B,n_post,4],
    anchor [1,A,4]) -- the 4-tuple of the working code."""
    n, _, h, w = feat.shape
    # (float32 feature map: the reference's arithmetic, bit for bit.  A float64 feature map - detector_forward(exact=True) - runs
    #  the two convs in float64 and rounds their outputs ONCE to float32; everything behind them is the reference's f32 code)
    wt = lambda k: sd[prefix + k].to(feat.dtype)          # noqa: E731
    locs = F.conv2d(feat, wt("loc.weight"), wt("loc.bias")).float()
    locs = locs.permute(0, 2, 3, 1).contiguous().view(n, -1, 4)
    scores = F.conv2d(feat, wt("score.weight"), wt("score.bias")).float()
    scores = scores.permute(0, 2, 3, 1).contiguous().view(n, -1, 2)
    fg = F.softmax(scores, dim=-1)[:, :, 1].contiguous().view(n, -1)
    base = generate_basic_anchor(ratios=ratios, anchor_scales=anchor_scales)
    anchor = enumerate_shifted_anchor(base, feat_stride, h, w)
    rois, dbg = [], []
    for i in range(n):
        r = proposal_layer(locs[i], fg[i], anchor, img_size, scale=scale, mode=mode, return_debug=return_debug)
        if return_debug:
            r, d = r
            dbg.append(d)
        rois.append(r.unsqueeze(0))
    rois = torch.cat(rois, dim=0).float()
    out = (locs, scores, rois, anchor.unsqueeze(0).float())
    if return_debug:
        return out, {"fg": fg, "per_image": dbg}
    return out


def roi_head_forward(sd, feat, rois, roi_indices, img_size, roi_size=7, spatial_scale=1.0, prefix="", roi_op="pool"):
    """HarNetRoIHead.forward, nets/classify.py:19-56, with the classifier of
    models/hardnet.py:203-212 (mean over the 7x7 bins) inlined.

    x by img_size[1], y by img_size[0]; the image index of row k is
    roi_indices[k // R] for any B, R (the reference hard-codes R=128, batch 1: Q5).
    ``sd``: ``{prefix}cls_loc.weight/bias`` [4*n_class, C], ``{prefix}score.weight/bias`` [n_class, C]."""
    n, _, hf, wf = feat.shape
    R = rois.shape[1]
    flat = rois.reshape(-1, 4)
    fm = torch.zeros_like(flat)
    fm[:, [0, 2]] = flat[:, [0, 2]] / img_size[1] * wf
    fm[:, [1, 3]] = flat[:, [1, 3]] / img_size[0] * hf
    idx = roi_indices.reshape(-1, 1).to(fm.dtype).repeat_interleave(R, dim=0)
    exact = feat.dtype == torch.float64   # detector_forward(exact=True): pooling picks among the feature values rounded once to f32,
    feat32 = feat.float()                 # the mean and the two Linear layers run in float64 and are rounded once
    if roi_op == "align":      # the added option (RoIAlign, sampling_ratio 2, aligned=False); the reference's head is "pool"
        pooled = roi_align(feat32, torch.cat([idx, fm], dim=1), (roi_size, roi_size), spatial_scale, 2, False)
    else:
        pooled = roi_pool(feat32, torch.cat([idx, fm], dim=1), (roi_size, roi_size), spatial_scale)
    if exact:
        pooled = pooled.double()
    fc7 = F.adaptive_avg_pool2d(pooled, (1, 1)).flatten(1)
    wt = lambda k: sd[prefix + k].to(fc7.dtype)           # noqa: E731
    cls_locs = F.linear(fc7, wt("cls_loc.weight"), wt("cls_loc.bias")).float()
    scores = F.linear(fc7, wt("score.weight"), wt("score.bias")).float()
    return cls_locs.view(n, -1, cls_locs.shape[1]), scores.view(n, -1, scores.shape[1])
