"""Region proposal network with the reference's module surface (nets/rpn.py) on HIP kernels.

``ProposalCreator`` and ``RegionProposalNetwork`` keep the reference's constructor / call
signatures, the 4-tuple return of the working code (quirk Q6), the "train" vs "training" mode-string
behaviour (Q3), the img_size[1] / img_size[2] clamp indexing (Q1) and the duplicate-padding rule
after NMS (Q4).  The per-image Python loop of the reference (nets/rpn.py:129-137) is replaced by
batched kernels:

    ONE 1x1 conv GEMM for loc + score (N = 4A + 2A = 54 padded to 56: wide epilogue stores, NHWC out)
    ->  tsod_rpn_decode_f32  ->  tsod_sort_topk_desc_f32  ->  tsod_nms_f32 (mask + wave scan + pad)

Nothing in the chain synchronises with the host.  The one error the reference can raise here
(IndexError when the pad needs more candidates than exist) is recorded in a device status word:
``strict=True`` checks it right away (one sync), otherwise call ``raise_if_error()`` when convenient.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import _ffi, hip_ops
from .._ffi import ACT_NONE, TsodError, require_cuda
from ..engine import PackedConv, PlanOwner, weights_bf16x3, weights_fp16x2
from ..utils._config import load_config
from ..utils.basic_anchors import generate_basic_anchor

config = load_config()
device = config["device"]          # kept for surface parity; forward uses x.device (quirk Q12)


class ProposalCreator:
    def __init__(self, mode, nms_iou=0.7, n_train_pre_nms=12000, n_train_post_nms=600, n_test_pre_nms=3000,
                 n_test_post_nms=300, min_size=16):
        self.mode = mode
        self.nms_iou = nms_iou
        self.n_train_pre_nms = n_train_pre_nms
        self.n_train_post_nms = n_train_post_nms
        self.n_test_pre_nms = n_test_pre_nms
        self.n_test_post_nms = n_test_post_nms
        self.min_size = min_size
        self.strict = True
        self._status = {}

    def counts(self):
        """(n_pre_nms, n_post_nms): only the literal "train" selects the train numbers."""
        if self.mode == "train":
            return self.n_train_pre_nms, self.n_train_post_nms
        return self.n_test_pre_nms, self.n_test_post_nms

    def select(self, boxes, keys, strict=None):
        """boxes [B,n,4] decoded+clamped, keys [B,n] (fg score, -inf = filtered) -> rois [B,n_post,4]."""
        n_pre, n_post = self.counts()
        if n_pre <= 0:
            n_pre = min(boxes.shape[1], 16384)
        counts, _, bs, _ = hip_ops.sort_topk_desc(keys, boxes, n_pre)
        # sticky device-side error word: allocated (zeroed) once per device, only ever OR-ed into by the NMS kernel,
        # cleared by raise_if_error() - no per-forward memset launch
        status = self._status.get(boxes.device)
        if status is None:
            status = self._status[boxes.device] = torch.zeros((1,), dtype=torch.int32, device=boxes.device)
        _, rois, _, status = hip_ops.nms_sorted(bs, counts, self.nms_iou, n_post, status=status)
        self.last_status = status
        if self.strict if strict is None else strict:
            self.raise_if_error()
        return rois

    def raise_if_error(self):
        st = getattr(self, "last_status", None)
        if st is not None and int(st.item()) & 1:
            with torch.inference_mode():            # the word may have been allocated under inference mode
                st.zero_()
            raise IndexError("proposal padding needs more candidates than survive the min-size filter "
                             "(the reference raises IndexError at nets/rpn.py:69)")

    def __call__(self, loc, score, anchor, img_size, scale=1.):
        """One image: loc [A,4], score [A] fg probabilities, anchor [A,4] -> roi [n_post,4]."""
        require_cuda(loc, "ProposalCreator")
        boxes, keys = hip_ops.proposal_decode(anchor, loc, score, img_size[1], img_size[2], self.min_size * scale)
        return self.select(boxes.unsqueeze(0), keys.unsqueeze(0))[0]


class RegionProposalNetwork(PlanOwner, nn.Module):
    def __init__(self, in_channels=512, ratios=[0.5, 1, 2], anchor_scales=[8, 16, 32], feat_stride=16,
                 mode="training"):
        super().__init__()
        if isinstance(ratios, int):
            # legacy call shape RegionProposalNetwork(512, 512, ratios=...) of nets/frcnn.py:16-22 (quirk Q7)
            raise TypeError("RegionProposalNetwork() got multiple values for argument 'ratios'")
        self.anchor_base = generate_basic_anchor(anchor_scales=anchor_scales, ratios=ratios)
        n_anchor = self.anchor_base.shape[0]
        self.score = nn.Conv2d(in_channels, n_anchor * 2, 1, 1, 0)
        self.loc = nn.Conv2d(in_channels, n_anchor * 4, 1, 1, 0)
        self.feat_stride = feat_stride
        self.proposal_layer = ProposalCreator(mode)
        self.proposal_layer.strict = False
        self._init_plan_owner()

    def _pack(self, dev):
        """(fused conv, base anchors, 4A, 2A) on ``dev``: the loc and score convs (nets/rpn.py:86-88) stacked into one
        [4A + 2A (+ pad to a multiple of 4), C] weight: rows [0,4A) = loc, [4A,6A) = score, zero rows after."""
        ent = self._packed_cache.get(("rpn", dev))
        if ent is None:
            n_loc, n_sc = self.loc.out_channels, self.score.out_channels
            cout = (n_loc + n_sc + 3) // 4 * 4
            w = torch.zeros((cout,) + tuple(self.loc.weight.shape[1:]), dtype=torch.float32)
            b = torch.zeros(cout, dtype=torch.float32)
            w[:n_loc], w[n_loc:n_loc + n_sc] = self.loc.weight.detach().float().cpu(), self.score.weight.detach().float().cpu()
            b[:n_loc], b[n_loc:n_loc + n_sc] = self.loc.bias.detach().float().cpu(), self.score.bias.detach().float().cpu()
            ent = self._packed_cache[("rpn", dev)] = (
                PackedConv(w.to(dev), dev, bias=b, act=ACT_NONE),
                torch.as_tensor(self.anchor_base, dtype=torch.float32).to(dev).contiguous(), n_loc, n_sc)
        return ent

    def _conv_kw(self, pc, prec, feat_amax, range_flag, feat=None):
        """Arguments of the fused conv that depend on the arithmetic: the pre-split weight image kept beside the f32 weights
        (never re-split per call: a constant of a captured graph) and, for fp16x2, the feature map's range words - the
        backbone plan's (``feat_amax``) inside the detector forward, or, for a feature map somebody else produced (the staged
        API: ``forward`` / ``forward_nhwc`` / ``FasterRCNN.forward(mode="rpn")`` on a geometry whose tuned choice is fp16x2), a
        temporary set filled by one tsod_absmax_f32 pass over ``feat`` in front of the GEMM."""
        if prec == _ffi.PREC_BF16X3:
            return {"w3": weights_bf16x3(pc)}
        if prec == _ffi.PREC_FP16X2:
            if not feat_amax:
                if feat is None:
                    raise TsodError("RegionProposalNetwork: the fp16x2 arithmetic needs the feature map's range words")
                feat_amax = hip_ops.absmax(feat, hip_ops.new_amax_words(feat.device))
            w2, e = weights_fp16x2(pc)
            return {"w2": w2, "w_scale_exp": e, "amax_in": feat_amax, "range_flag": range_flag}
        return {}

    def propose(self, feat: torch.Tensor, img_size, scale=1., want_anchors=False, feat_amax=None, range_flag=None):
        """feat NHWC [n,h,w,C] -> (fused conv output [n*h*w, pad4(6A)] with loc in columns [0,4A) and score in
        [4A,6A), rois [n,n_post,4], anchors [h*w*A,4] or None).  Four launches, no host sync.  ``feat_amax``: the range words
        of ``feat`` (engine.Plan.output_amax) - what an fp16x2 choice of the fused conv takes its activation scale from."""
        require_cuda(feat, "RegionProposalNetwork")
        n, h, w, _ = feat.shape
        pc, base, n_loc, n_sc = self._pack(feat.device)
        tile, split, prec = self.__dict__.get("_gemm_choice", {}).get((n, h, w), (0, 0, 0))
        fused = hip_ops.conv2d_nhwc(feat, pc.w, shift=pc.shift, tile=tile, split_k=split, precision=prec,
                                    **self._conv_kw(pc, prec, feat_amax, range_flag, feat)).view(n * h * w, pc.cout)
        boxes, _, keys, anchor = hip_ops.rpn_decode(fused[:, :n_loc], fused[:, n_loc:n_loc + n_sc], base, n, h, w,
                                                    self.feat_stride, img_size[1], img_size[2],
                                                    self.proposal_layer.min_size * scale, want_anchors=want_anchors)
        return fused, self.proposal_layer.select(boxes, keys), anchor

    def autotune(self, feat: torch.Tensor, feat_amax=None, range_flag=None):
        """Pin the fastest (tile, K-slice schedule, arithmetic) of the fused loc + score GEMM for this feature geometry
        (fp16x2 among the candidates when the feature map's range words are given)."""
        n, h, w, _ = feat.shape
        pc = self._pack(feat.device)[0]
        kw = {"w3": weights_bf16x3(pc)}
        precisions = (0, 1)
        if feat_amax:
            kw.update(self._conv_kw(pc, _ffi.PREC_FP16X2, feat_amax, range_flag))
            precisions = (0, 1, 2)
        self.__dict__.setdefault("_gemm_choice", {})[(n, h, w)] = hip_ops.tune_conv(feat, pc.w, shift=pc.shift, precisions=precisions, **kw)
        return self._gemm_choice[(n, h, w)]

    def forward_nhwc(self, feat: torch.Tensor, img_size, scale=1.):
        """feat NHWC [n,h,w,C] -> (rpn_locs [n,h*w*A,4], rpn_scores [n,h*w*A,2], rois [n,n_post,4], anchor [1,h*w*A,4])."""
        n = feat.shape[0]
        fused, rois, anchor = self.propose(feat, img_size, scale, want_anchors=True)
        _, _, n_loc, n_sc = self._pack(feat.device)
        # the reference returns contiguous [n, h*w*A, 4] / [n, h*w*A, 2] tensors (its permute + view, nets/rpn.py:108,113):
        # two strided copies out of the fused buffer (memory plumbing; the detector forward never asks for them)
        locs = fused[:, :n_loc].contiguous().view(n, -1, 4)
        scores = fused[:, n_loc:n_loc + n_sc].contiguous().view(n, -1, 2)
        return locs, scores, rois, anchor.unsqueeze(0)

    def forward(self, x, img_size, scale=1.):
        """x NCHW [n,C,h,w] (the reference's layout)."""
        require_cuda(x, "RegionProposalNetwork")
        return self.forward_nhwc(hip_ops.nchw_to_nhwc(x), img_size, scale)

    def raise_if_error(self):
        self.proposal_layer.raise_if_error()
