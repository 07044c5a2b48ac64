"""Runs the HBM-bound kernels the detector forward of bench.py does not exercise (or only at small sizes) at full size, a few
launches each, as the target of rocprofv3 --pmc passes (scripts/collect_box_profiles.sh):
    bbox_iou_kernel            utils/loc_bbox_iou.py:4-27      3000 x 3000 boxes (the size BASELINE.md quotes for the CPU)
    targets.hip kernels        nets/frcnn_training.py:19-177   37 800 anchors x 64 gt boxes; 2 000 RoIs x 64 gt boxes
    dwconv3x3 / gconv1x1_pair  models/hardnet.py:21-36,193-196 HarDNet-68 maps at 3x800x1333 (batch 1)
    maxpool, layout, resize    models/resnet.py:139, module boundary, dataset/transform.py:14-17
Writes the algorithmic bytes of every case to the JSON given as argv[1] (input + output, each byte once)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from two_stage_object_detection_amd import hip_ops  # noqa: E402

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
spec = {}
REPS = 6


def boxes(n, span=800.0):
    xy = torch.rand(n, 2, generator=g) * span
    return torch.cat([xy, xy + torch.rand(n, 2, generator=g) * 200 + 4], dim=1).to(dev)


# ---- pairwise IoU 3000 x 3000: 2 * 48 KB in, 36 MB out
a, b = boxes(3000), boxes(3000)
for _ in range(REPS):
    hip_ops.bbox_iou(a, b)
spec["bbox_iou_kernel"] = {"algorithmic_MB": (3000 * 16 * 2 + 3000 * 3000 * 4) / 1e6, "case": "3000 x 3000 boxes"}

# ---- training-side target creators
anchor = hip_ops.enumerate_anchors(torch.tensor([[-45.25, -22.63, 45.25, 22.63]] * 9, device=dev), 16, 50, 84)
gt = boxes(64, 700.0)
for _ in range(REPS):
    hip_ops.anchor_targets(gt, anchor, 128, 256, 0.7, 0.3)
A, G = anchor.shape[0], 64
for k in ("rowmax_kernel", "colargmax_kernel", "anchor_label_kernel", "anchor_loc_kernel"):
    spec[k] = {"case": f"{A} anchors x {G} gt boxes"}
spec["rowmax_kernel"]["algorithmic_MB"] = (A * 16 + G * 16 + A * 8) / 1e6
spec["colargmax_kernel"]["algorithmic_MB"] = (A * 16 + G * 16 + G * 8) / 1e6
spec["anchor_label_kernel"]["algorithmic_MB"] = (A * 8 + A * 8) / 1e6
spec["anchor_loc_kernel"]["algorithmic_MB"] = (A * 16 + A * 4 + A * 16) / 1e6
roi = boxes(2000)
lab = torch.randint(1, 81, (64,), generator=g)
for _ in range(REPS):
    hip_ops.proposal_targets(roi, gt, lab, 128, 32, 0.5, 0.5, 0.0)
spec["proposal_select_kernel"] = {"algorithmic_MB": ((2000 + 64) * (16 + 8) + 128 * 40) / 1e6, "case": "2000 RoIs + 64 gt boxes -> 128 samples"}

# ---- HarDNet-68 depthwise layers at 3x800x1333, batch 1 (stem s2, block layers at 200x334 / 100x167 / 50x84, tail)
for (H, W, C, stride, name) in ((400, 667, 64, 2, "stem dw s2"), (200, 334, 40, 1, "block 1 layer"), (200, 334, 128, 1, "transition 1"),
                                (100, 167, 160, 1, "block 3 layer"), (50, 84, 640, 1, "transition 4"), (50, 84, 1024, 2, "tail dw s2")):
    x = torch.randn(1, H, W, C, device=dev)
    w = torch.randn(3, 3, C, device=dev)
    sc, sh = torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    for _ in range(REPS):
        y = hip_ops.dwconv3x3_nhwc(x, w, sc, sh, stride=stride)
    spec.setdefault("dwconv3x3_kernel", {"cases": []})["cases"].append(
        {"case": f"{name}: {H}x{W}x{C} s{stride}", "algorithmic_MB": (x.numel() + y.numel()) * 4 / 1e6})
x = torch.randn(1, 13, 21, 1024, device=dev)
wg = torch.randn(512, 2, device=dev)
for _ in range(REPS):
    hip_ops.gconv1x1_pair_nhwc(x, wg, torch.randn(512, device=dev))
spec["gconv1x1_pair_kernel"] = {"algorithmic_MB": (x.numel() + x.numel() // 2) * 4 / 1e6, "case": "13x21x1024 -> 512"}

# ---- ResNet stem pool, module-boundary layout, input resize
x = torch.randn(1, 400, 667, 64, device=dev)
for _ in range(REPS):
    hip_ops.maxpool3x3s2_nhwc(x)
spec["maxpool3x3s2_kernel"] = {"algorithmic_MB": 85.4, "case": "400x667x64"}
img = torch.rand(1, 3, 800, 1333, device=dev)
for _ in range(REPS):
    hip_ops.nchw_to_nhwc(img, 4)
spec["nchw_to_nhwc"] = {"algorithmic_MB": 29.9, "case": "3x800x1333 -> NHWC4"}
u8 = (torch.rand(1080, 1920, 3, device=dev) * 255).to(torch.uint8)
for _ in range(REPS):
    hip_ops.resize_bilinear_aa(u8, 800, 1333)
spec["resize_aa"] = {"algorithmic_MB": (1080 * 1920 * 3 + 800 * 1333 * 16) / 1e6, "case": "1080p u8 -> 800x1333 NHWC4 f32"}
torch.cuda.synchronize()
if len(sys.argv) > 1:
    json.dump(spec, open(sys.argv[1], "w"), indent=1)
