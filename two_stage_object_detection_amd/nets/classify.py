"""RoI head with the reference's module surface (nets/classify.py) on HIP kernels.

RoI rescale + image index + RoIPool 7x7 + the classifier's 7x7 mean are one kernel
(tsod_roi_pool_avg_f32: the [K,C,7,7] pooled tensor is never materialised); the two nn.Linear layers
run on the f32 MFMA GEMM.  Works for any batch size and any RoIs/image (the reference hard-codes 128
RoIs and batch 1: quirk Q5).  ``in_channels`` generalises the reference's hard-coded 512.
"""
from __future__ import annotations

import torch
from torch import nn

from .. import hip_ops
from .._ffi import TsodError, require_cuda
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


class HarNetRoIHead(nn.Module):
    def __init__(self, n_class, roi_size, spatial_scale, classifier, in_channels=512):
        super().__init__()
        self.classifier = classifier
        self.cls_loc = nn.Linear(in_channels, n_class * 4)
        self.score = nn.Linear(in_channels, n_class)
        self.roi = RoIPool((roi_size, roi_size), spatial_scale)

    def forward_nhwc(self, feat, rois, roi_indices, img_size):
        """feat NHWC [n,Hf,Wf,C]; rois [n,R,4] image coords; roi_indices [n]; img_size (H,W) (quirk Q2)."""
        require_cuda(feat, "HarNetRoIHead")
        if not isinstance(self.classifier, HarNetClassifier):
            raise TsodError("only the reference's HarNetClassifier (mean over the 7x7 bins) has a HIP path")
        n = feat.shape[0]
        rois = rois.reshape(n, -1, 4)
        fc7 = hip_ops.roi_pool_avg_nhwc(feat, rois, roi_indices, img_size[0], img_size[1], self.roi.output_size,
                                        self.roi.spatial_scale)
        roi_cls_locs = hip_ops.linear(fc7, self.cls_loc.weight, self.cls_loc.bias)
        roi_scores = hip_ops.linear(fc7, self.score.weight, self.score.bias)
        return roi_cls_locs.view(n, -1, roi_cls_locs.size(1)), roi_scores.view(n, -1, roi_scores.size(1))

    def forward(self, x, rois, roi_indices, img_size):
        """x NCHW [n,C,Hf,Wf] (the reference's layout)."""
        require_cuda(x, "HarNetRoIHead")
        return self.forward_nhwc(hip_ops.nchw_to_nhwc(x), rois, roi_indices, img_size)
