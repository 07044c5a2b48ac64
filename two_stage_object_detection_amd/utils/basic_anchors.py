"""Anchor generation with the reference's function surface (utils/basic_anchors.py).

``generate_basic_anchor`` produces the 9 base anchors (a constructor-time constant: computed on the
host in f32 exactly as the reference's expressions round, then placed on ``device``);
``enumerate_shifted_anchor`` is a HIP kernel (tsod_enumerate_anchors_f32).  Inside the detector the
shifted anchors are never materialised: tsod_rpn_decode_f32 regenerates them from the index.
"""
from __future__ import annotations

import numpy as np
import torch

from ._config import load_config
from .. import hip_ops

config = load_config()
device = config["device"]          # patchable module global, like the reference's


def _anchor_device():
    if torch.cuda.is_available():
        return torch.device(device)
    return torch.device("cpu")      # constants only; every compute entry point refuses CPU tensors


def generate_basic_anchor(base_size=8, ratios=[0.5, 1, 2], anchor_scales=[8, 16, 32]):
    """[len(ratios)*len(scales), 4] f32 rows (-w/2, -h/2, w/2, h/2), ratio-major / scale-minor;
    h = base*scale*sqrt(r), w = base*scale*sqrt(1/r) (reference :11-23, quirk Q8)."""
    out = np.zeros((len(ratios) * len(anchor_scales), 4), dtype=np.float32)
    for i, r in enumerate(ratios):
        sr = np.sqrt(np.float32(r))                 # f32 sqrt of the f32 ratio
        sir = np.sqrt(np.float32(1.0 / r))          # 1/r in double first, as the reference writes it
        for j, s in enumerate(anchor_scales):
            side = np.float32(base_size * s)        # python ints multiply exactly
            h, w = np.float32(side * sr), np.float32(side * sir)
            out[i * len(anchor_scales) + j] = (-w / np.float32(2), -h / np.float32(2), w / np.float32(2), h / np.float32(2))
    return torch.from_numpy(out).to(_anchor_device())


def enumerate_shifted_anchor(anchor_base, feat_stride, height, width):
    """anchor[(y*W + x)*A + a] = base[a] + (x*s, y*s, x*s, y*s)  (reference :27-57, x fastest)."""
    base = torch.as_tensor(anchor_base, dtype=torch.float32)
    return hip_ops.enumerate_anchors(base, feat_stride, height, width)
