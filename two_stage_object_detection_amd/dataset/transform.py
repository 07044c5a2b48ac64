"""dataset/transform.py -- the evaluation-time input transform on the GPU.

Reference: ``eval_transform = T.Compose([T.Resize((600, 600)), T.ToTensor()])`` (dataset/transform.py:14-17) applied to
``{"image": tv_tensors.Image(PIL image, dtype=float32), "boxes": ..., "labels": ...}`` (dataset/dataloader.py:35-44,
multi_inference.py:65-76).  On a float tensor torchvision's v2 ``Resize`` is ATen's antialiased bilinear interpolation
(``align_corners=False``); ``ToTensor`` passes tensors through, so the detector sees f32 CHW values in 0..255 (the
reference never divides by 255) and XYXY boxes scaled by (600/W, 600/H).

Here the decoded image stays u8 HWC, goes to the GPU as it is (a third of the f32 bytes), and one HIP kernel
(``tsod_resize_bilinear_aa_u8_f32``) produces the resized f32 image directly in the layout asked for: NCHW like the
reference's tensor (``EvalTransform.__call__``), or the NHWC(4) buffer the first conv reads (``EvalTransform.batch``).
The training-time augmentations (``transform``: photometric distort, flip, scale jitter) are outside the path.
"""
from __future__ import annotations

import torch

from .. import hip_ops
from .._ffi import NHWC4Images, TsodError


class EvalTransform:
    """``EvalTransform(size)(sample)`` mirrors ``eval_transform(sample)`` for ``sample["image"]`` = u8 [H,W,3] CUDA tensor.

    ``mul`` scales the output (1.0 = the reference: values stay 0..255; 1/255 gives [0,1])."""

    def __init__(self, size=(600, 600), mul: float = 1.0):
        self.size = (int(size[0]), int(size[1]))
        self.mul = float(mul)

    def __call__(self, sample):
        if isinstance(sample, dict):
            img = sample["image"]
            out = dict(sample)
            out["image"] = self.image(img)
            if sample.get("boxes") is not None:
                H, W = img.shape[0], img.shape[1]
                b = torch.as_tensor(sample["boxes"], dtype=torch.float32)
                # v2 Resize on XYXY boxes: x * (new_w / w), y * (new_h / h)
                ratio = torch.tensor([self.size[1] / W, self.size[0] / H, self.size[1] / W, self.size[0] / H],
                                     dtype=torch.float32, device=b.device)
                out["boxes"] = b * ratio
            return out
        return self.image(sample)

    def image(self, img_u8_hwc: torch.Tensor) -> torch.Tensor:
        """u8 [H,W,3] on the GPU -> f32 [3,OH,OW] (the tensor the reference's transform returns)."""
        return hip_ops.resize_bilinear_aa(img_u8_hwc, self.size[0], self.size[1], layout="nchw", mul=self.mul)

    def batch(self, images, out: NHWC4Images | None = None) -> NHWC4Images:
        """A list of u8 [H_i,W_i,3] CUDA images (sizes may differ) -> one ``NHWC4Images`` [B,OH,OW,4] batch, one launch per
        image.  ``out`` = ``model.extractor.input_buffer(B, OH, OW, device, slot)`` writes straight into the backbone's
        input buffer."""
        if len(images) == 0:
            raise TsodError("EvalTransform.batch: empty image list")
        OH, OW = self.size
        dev = images[0].device
        if out is None:
            out = NHWC4Images(torch.empty((len(images), OH, OW, 4), dtype=torch.float32, device=dev))
        if tuple(out.data.shape) != (len(images), OH, OW, 4):
            raise TsodError(f"EvalTransform.batch: out is {tuple(out.data.shape)}, expected {(len(images), OH, OW, 4)}")
        for b, img in enumerate(images):
            hip_ops.resize_bilinear_aa(img, OH, OW, layout="nhwc4", mul=self.mul, out=out.data[b])
        return out


eval_transform = EvalTransform((600, 600))      # the reference's instance (dataset/transform.py:14)
