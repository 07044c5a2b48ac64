"""RoI head with the reference's module surface (nets/classify.py) on HIP kernels.

RoI rescale + image index + RoIPool 7x7 + the classifier's 7x7 mean are one kernel
(tsod_roi_pool_avg_f32: the [K,C,7,7] pooled tensor is never materialised); the two nn.Linear layers run as ONE
f32 MFMA GEMM (N = 4*n_class + n_class = 405 padded to 408: wide epilogue stores), whose two column ranges are
returned as views.  Works for any batch size and any RoIs/image (the reference hard-codes 128
RoIs and batch 1: quirk Q5).  ``in_channels`` generalises the reference's hard-coded 512.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import _ffi, hip_ops
from .._ffi import TsodError, require_cuda
from ..engine import PlanOwner
from ..models.hardnet import HarNetClassifier


class RoIPool(nn.Module):
    """Stand-in for torchvision.ops.RoIPool (same ctor; NCHW in, [K,C,PH,PW] out) on the HIP kernel."""

    def __init__(self, output_size, spatial_scale):
        super().__init__()
        self.output_size = tuple(output_size) if isinstance(output_size, (tuple, list)) else (output_size, output_size)
        self.spatial_scale = spatial_scale

    def forward(self, x, rois):
        require_cuda(x, "RoIPool")
        return hip_ops.roi_pool_nhwc(hip_ops.nchw_to_nhwc(x), rois, self.output_size, self.spatial_scale)


class RoIAlign(nn.Module):
    """Stand-in for torchvision.ops.RoIAlign (same ctor; NCHW in, [K,C,PH,PW] out) on the HIP kernel."""

    def __init__(self, output_size, spatial_scale, sampling_ratio, aligned=False):
        super().__init__()
        self.output_size = tuple(output_size) if isinstance(output_size, (tuple, list)) else (output_size, output_size)
        self.spatial_scale, self.sampling_ratio, self.aligned = spatial_scale, sampling_ratio, aligned

    def forward(self, x, rois):
        require_cuda(x, "RoIAlign")
        return hip_ops.roi_align_nhwc(hip_ops.nchw_to_nhwc(x), rois, self.output_size, self.spatial_scale,
                                      self.sampling_ratio, self.aligned)


class HarNetRoIHead(PlanOwner, nn.Module):
    def __init__(self, n_class, roi_size, spatial_scale, classifier, in_channels=512, roi_op="pool"):
        """``roi_op`` (added, non-breaking): "pool" = the reference's RoIPool (nets/classify.py:17), "align" =
        torchvision-style RoIAlign (sampling_ratio 2, aligned=False; change ``self.roi``'s attributes for others)."""
        super().__init__()
        self.classifier = classifier
        self.cls_loc = nn.Linear(in_channels, n_class * 4)
        self.score = nn.Linear(in_channels, n_class)
        if roi_op == "pool":
            self.roi = RoIPool((roi_size, roi_size), spatial_scale)
        elif roi_op == "align":
            self.roi = RoIAlign((roi_size, roi_size), spatial_scale, sampling_ratio=2, aligned=False)
        else:
            raise ValueError(f"roi_op must be 'pool' or 'align', got {roi_op!r}")
        self.roi_op = roi_op
        self._init_plan_owner()

    def _pack(self, dev):
        """(weight [pad4(5*n_class), C], bias, 4*n_class, n_class) on ``dev``: cls_loc rows, then score rows
        (nets/classify.py:13,15), zero rows up to a multiple of 4."""
        ent = self._packed_cache.get(("head", dev))
        if ent is None:
            n_loc, n_sc = self.cls_loc.out_features, self.score.out_features
            n_pad = (n_loc + n_sc + 3) // 4 * 4
            w = torch.zeros((n_pad, self.cls_loc.in_features), dtype=torch.float32)
            b = torch.zeros(n_pad, dtype=torch.float32)
            w[:n_loc], w[n_loc:n_loc + n_sc] = self.cls_loc.weight.detach().float().cpu(), self.score.weight.detach().float().cpu()
            b[:n_loc], b[n_loc:n_loc + n_sc] = self.cls_loc.bias.detach().float().cpu(), self.score.bias.detach().float().cpu()
            ent = self._packed_cache[("head", dev)] = (w.to(dev).contiguous(), b.to(dev), n_loc, n_sc)
        return ent

    def _w3(self, dev):
        """The pre-split bf16x3 image of the fused weight, made once and kept beside it (a constant of the forward: it must
        not be re-split - and baked into a captured graph - on every call)."""
        w3 = self._packed_cache.get(("head.w3", dev))
        if w3 is None:
            w3 = self._packed_cache[("head.w3", dev)] = hip_ops.pack_conv_weight_bf16x3(self._pack(dev)[0])
            if not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream(dev).synchronize()       # complete before another slot's stream uses it
        return w3

    def _w2(self, dev):
        """(fp16x2 image, exponent) of the fused weight, made once and kept beside it like the bf16x3 image."""
        w2 = self._packed_cache.get(("head.w2", dev))
        if w2 is None:
            w = self._pack(dev)[0]
            e = hip_ops.fp16x2_weight_scale_exp(w)
            w2 = self._packed_cache[("head.w2", dev)] = (hip_ops.pack_conv_weight_fp16x2(w.view(w.shape[0], 1, 1, w.shape[1]), e), e)
            if not torch.cuda.is_current_stream_capturing():
                torch.cuda.current_stream(dev).synchronize()
        return w2

    def _gemm_kw(self, dev, prec, feat_amax, range_flag, fc7=None):
        """What the fused GEMM's arithmetic needs beside the f32 weights.  fp16x2 scales its input with range words: the
        backbone plan's (``feat_amax``) inside the detector forward; for the staged API (``forward`` / ``forward_nhwc`` /
        ``FasterRCNN.forward(mode="head")`` on a RoI count whose tuned choice is fp16x2) a temporary set filled by one
        tsod_absmax_f32 pass over the pooled features in front of the GEMM."""
        if prec == _ffi.PREC_BF16X3:
            return {"w3": self._w3(dev)}
        if prec == _ffi.PREC_FP16X2:
            if not feat_amax:
                if fc7 is None:
                    raise TsodError("HarNetRoIHead: the fp16x2 arithmetic needs range words for its input")
                feat_amax = hip_ops.absmax(fc7, hip_ops.new_amax_words(fc7.device))
            w2, e = self._w2(dev)
            # the pooled means are bounded by the feature map's abs-max: its words give a safe scale
            return {"w2": w2, "w_scale_exp": e, "amax_in": feat_amax, "range_flag": range_flag}
        return {}

    def pooled(self, feat, rois, roi_indices, img_size):
        """RoI rescale + RoIPool / RoIAlign 7x7 + the classifier's mean: [n*R, C] (one launch)."""
        n = feat.shape[0]
        rois = rois.reshape(n, -1, 4)
        if isinstance(self.roi, RoIAlign):
            return hip_ops.roi_align_avg_nhwc(feat, rois, roi_indices, img_size[0], img_size[1], self.roi.output_size,
                                              self.roi.spatial_scale, self.roi.sampling_ratio, self.roi.aligned)
        return hip_ops.roi_pool_avg_nhwc(feat, rois, roi_indices, img_size[0], img_size[1], self.roi.output_size,
                                         self.roi.spatial_scale)

    def forward_nhwc(self, feat, rois, roi_indices, img_size, feat_amax=None, range_flag=None):
        """feat NHWC [n,Hf,Wf,C]; rois [n,R,4] image coords; roi_indices [n]; img_size (H,W) (quirk Q2).  ``feat_amax``: the
        range words of ``feat`` (what an fp16x2 choice of the fused GEMM scales its input with)."""
        require_cuda(feat, "HarNetRoIHead")
        if not isinstance(self.classifier, HarNetClassifier):
            raise TsodError("only the reference's HarNetClassifier (mean over the 7x7 bins) has a HIP path")
        n = feat.shape[0]
        fc7 = self.pooled(feat, rois, roi_indices, img_size)
        w, b, n_loc, n_sc = self._pack(feat.device)
        # the fused Linear as a 1x1 "conv" over M = n*R rows: same GEMM kernel as tsod_linear_f32, with the tile / K-slice /
        # arithmetic choice of autotune() when there is one
        M, K = fc7.shape
        tile, split, prec = self.__dict__.get("_gemm_choice", {}).get(M, (0, 0, 0))
        both = hip_ops.conv2d_nhwc(fc7.view(1, 1, M, K), w.view(w.shape[0], 1, 1, K), shift=b, tile=tile, split_k=split,
                                   precision=prec, **self._gemm_kw(feat.device, prec, feat_amax, range_flag, fc7)
                                   ).view(M, w.shape[0])                           # [n*R, pad4(5*n_class)]
        # views into the fused output (row pitch 408 for 81 classes): same values and shapes as the reference's two
        # Linear outputs; .contiguous() them if a consumer needs dense storage
        return both[:, :n_loc].view(n, -1, n_loc), both[:, n_loc:n_loc + n_sc].view(n, -1, n_sc)

    def autotune(self, fc7: torch.Tensor, feat_amax=None, range_flag=None):
        """Pin the fastest (tile, K-slice schedule, arithmetic) of the fused cls_loc + score GEMM for M = fc7.shape[0] RoIs
        (``fc7``: pooled RoI features of a real forward; fp16x2 among the candidates when the feature map's range words are given)."""
        w, b, _, _ = self._pack(fc7.device)
        M, K = fc7.shape
        kw = {"w3": self._w3(fc7.device)}
        precisions = (0, 1)
        if feat_amax:
            kw.update(self._gemm_kw(fc7.device, _ffi.PREC_FP16X2, feat_amax, range_flag))
            precisions = (0, 1, 2)
        self.__dict__.setdefault("_gemm_choice", {})[M] = hip_ops.tune_conv(fc7.view(1, 1, M, K), w.view(w.shape[0], 1, 1, K), shift=b,
                                                                             precisions=precisions, **kw)
        return self._gemm_choice[M]

    def forward(self, x, rois, roi_indices, img_size):
        """x NCHW [n,C,Hf,Wf] (the reference's layout)."""
        require_cuda(x, "HarNetRoIHead")
        return self.forward_nhwc(hip_ops.nchw_to_nhwc(x), rois, roi_indices, img_size)
