"""oracle/transform.py -- TEST INFRASTRUCTURE ONLY (CPU restatement; never on the product path).

The evaluation-time input transform of the reference: ``eval_transform`` (dataset/transform.py:14-17 =
``Resize((600, 600))`` + ``ToTensor``) applied to ``tv_tensors.Image(PIL image, dtype=torch.float32)``
(dataset/dataloader.py:35-44, multi_inference.py:65-76).

torchvision is neither vendored in /root/reference nor installed in the image, so the transform itself cannot be
run here (PARITY UNPINNED for the torchvision glue).  What it does on a float tensor image is, however, a single torch
call - ``torchvision.transforms.v2.functional.resize_image`` hands float tensors to
``torch.nn.functional.interpolate(..., mode="bilinear", align_corners=False, antialias=True)`` - and torch is the
arithmetic library the reference runs on, so the oracle is that call on the CPU.  ``ToTensor`` is a pass-through for
tensors: values stay in 0..255.  Boxes (XYXY) are scaled by (new_w / w, new_h / h) like v2's ``resize_bounding_boxes``.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def eval_transform(img_u8_hwc: torch.Tensor, size=(600, 600), boxes=None):
    """u8 [H,W,3] -> f32 [3,size[0],size[1]] (values 0..255) and, if given, the resized XYXY boxes."""
    x = img_u8_hwc.permute(2, 0, 1).unsqueeze(0).to(torch.float32)       # tv_tensors.Image(img, dtype=float32)
    y = F.interpolate(x, size=tuple(size), mode="bilinear", align_corners=False, antialias=True)[0]
    if boxes is None:
        return y
    H, W = img_u8_hwc.shape[:2]
    ratio = torch.tensor([size[1] / W, size[0] / H, size[1] / W, size[0] / H], dtype=torch.float32)
    return y, torch.as_tensor(boxes, dtype=torch.float32) * ratio
