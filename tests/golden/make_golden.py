#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own Python modules on CPU.

Run in the build container only (needs /root/reference; it is never shipped):

    python tests/golden/make_golden.py

What is pinned and how
----------------------
* utils/basic_anchors.py, utils/loc_bbox_iou.py, models/resnet.py, models/hardnet.py import
  and run as they are (module global ``device`` patched from "cuda:0" to "cpu").
* nets/rpn.py and nets/classify.py import ``torchvision.ops`` which is not installed in this
  image.  To run the reference's own glue code (reshape order, softmax channel, clamp quirk,
  sort/top-k, pad rule, RoI rescale, cat/view order) a stand-in ``torchvision.ops`` exposing
  ``nms`` and ``RoIPool`` is registered, backed by oracle/box_ops.c.  Those two operators are
  therefore NOT pinned by these vectors ("parity unpinned"); everything
  around them is.

Only inputs, weights (small cases), seeds and outputs are stored -- no reference source text.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("TSOD_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

import oracle  # noqa: E402  (stand-in ops only)

# ---- stand-in torchvision.ops (see docstring) -------------------------------------------
tv = types.ModuleType("torchvision")
tv_ops = types.ModuleType("torchvision.ops")


class _RoIPool(torch.nn.Module):
    def __init__(self, output_size, spatial_scale):
        super().__init__()
        self.output_size, self.spatial_scale = output_size, spatial_scale

    def forward(self, x, rois):
        return oracle.roi_pool(x, rois, self.output_size, self.spatial_scale)


tv_ops.nms = oracle.nms
tv_ops.RoIPool = _RoIPool
tv.ops = tv_ops
sys.modules["torchvision"] = tv
sys.modules["torchvision.ops"] = tv_ops

import utils.basic_anchors as ref_anchors  # noqa: E402
import utils.loc_bbox_iou as ref_box  # noqa: E402
import models.resnet as ref_resnet  # noqa: E402
import models.hardnet as ref_hardnet  # noqa: E402
import nets.rpn as ref_rpn  # noqa: E402
import nets.classify as ref_classify  # noqa: E402

ref_anchors.device = "cpu"
ref_rpn.device = "cpu"


def npd(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


def randomize_bn(module, gen):
    """Give BatchNorm layers non-trivial eval statistics so the fold is actually exercised."""
    for m in module.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.copy_(torch.randn(m.num_features, generator=gen) * 0.2)
            m.running_var.copy_(torch.rand(m.num_features, generator=gen) * 1.5 + 0.25)
            m.weight.data.copy_(torch.rand(m.num_features, generator=gen) + 0.5)
            m.bias.data.copy_(torch.randn(m.num_features, generator=gen) * 0.1)
        if isinstance(m, torch.nn.PReLU):
            m.weight.data.fill_(0.1 + 0.3 * float(torch.rand(1, generator=gen)))


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path) / 1024:.1f} KiB")


@torch.inference_mode()
def extras_round2():
    """Fixtures added in round 2 (own seeds, own files: the round-1 files above are left byte-identical)."""
    g = torch.Generator().manual_seed(20261004)
    # loc2bbox with k box sets per row ([n,4k] locs, utils/loc_bbox_iou.py:42-57: the 0::4 / 1::4 / 2::4 / 3::4 strides)
    src = torch.rand(40, 4, generator=g) * 500
    src[:, 2:] = src[:, :2] + torch.rand(40, 2, generator=g) * 250 + 1
    loc_k = torch.randn(40, 12, generator=g) * 0.4
    save("boxmath_k.npz", src=src.numpy(), loc_k3=loc_k.numpy(), loc2bbox_k3=ref_box.loc2bbox(src, loc_k).numpy(),
         empty=ref_box.loc2bbox(torch.zeros(0, 4), torch.zeros(0, 4)).numpy())

    # ---- training-side box ops: the reference's own target creators, nets/frcnn_training.py:19-177.
    # The module imports torchvision.ops.nms (stand-in above) and reads its device from configs/config.json ("cuda:0"):
    # patched to "cpu" after import, like nets.rpn.  Both classes are deterministic (no random sampling).
    import nets.frcnn_training as ref_train
    ref_train.device = "cpu"

    def boxes(n, span_x, span_y, wh_lo, wh_hi):
        xy = torch.rand(n, 2, generator=g) * torch.tensor([span_x, span_y])
        wh = torch.rand(n, 2, generator=g) * (wh_hi - wh_lo) + wh_lo
        return torch.cat([xy, xy + wh], dim=1)

    base = ref_anchors.generate_basic_anchor()
    anchor = ref_anchors.enumerate_shifted_anchor(base, 16, 20, 28)                 # 5040 anchors of a 320x448 image
    cases = {}
    gt5 = boxes(5, 300, 200, 40, 160)
    cases["default"] = (dict(), gt5)
    gt_dup = torch.cat([gt5[:3], gt5[1:2], boxes(2, 300, 200, 60, 200)])            # a duplicated gt: ties + override order (T4)
    cases["dup"] = (dict(), gt_dup)
    cases["many_pos"] = (dict(pos_iou_thresh=0.35, neg_iou_thresh=0.2), boxes(8, 250, 150, 80, 220))   # > 128 positives: the cap
    cases["all_pos_ratio"] = (dict(n_sample=16, pos_ratio=1.0, pos_iou_thresh=0.4), gt5)   # n_neg == 0: T1 disables all negatives
    cases["no_gt"] = (dict(), torch.zeros(0, 4))
    arrs = {"anchor": anchor.numpy()}
    for name, (kw, gt) in cases.items():
        loc, label = ref_train.AnchorTargetCreator(**kw)(gt, anchor)
        arrs[f"{name}.bbox"] = gt.numpy()
        arrs[f"{name}.loc"] = loc.numpy()
        arrs[f"{name}.label"] = label.numpy()
        arrs[f"{name}.kw"] = np.array(repr(kw))
    save("targets_anchor.npz", **arrs)

    arrs = {}
    gt6 = boxes(6, 300, 200, 40, 160)
    lab6 = torch.randint(0, 20, (6,), generator=g)
    roi = boxes(300, 380, 260, 16, 200)
    roi[::7] = gt6[torch.arange(0, 43) % 6] + torch.randn(43, 4, generator=g) * 4   # some RoIs near a gt -> positives
    pcases = {"default": (dict(), roi, gt6, lab6),
              "few": (dict(), roi[:60], gt6[:2], lab6[:2]),                          # fewer candidates than n_sample
              "no_gt": (dict(), roi[:200], torch.zeros(0, 4), torch.zeros(0, dtype=torch.int64)),
              "thresholds": (dict(n_sample=100, pos_ratio=0.3, pos_iou_thresh=0.4, neg_iou_thresh_high=0.4), roi, gt6, lab6),
              # a gap between the thresholds: sampled negatives sit at original indices beyond the kept length -> T2 raises
              "thresholds_gap": (dict(n_sample=64, pos_ratio=0.25, pos_iou_thresh=0.6, neg_iou_thresh_high=0.4,
                                      neg_iou_thresh_low=0.05), roi, gt6, lab6)}
    crowded = roi.clone()
    crowded[:100] = gt6[torch.arange(0, 100) % 6] + torch.randn(100, 4, generator=g) * 2   # > 64 positives up front: T2 raises
    pcases["index_error"] = (dict(), crowded, gt6, lab6)
    for name, (kw, r, gt, lab) in pcases.items():
        arrs[f"{name}.roi"], arrs[f"{name}.bbox"], arrs[f"{name}.label"] = r.numpy(), gt.numpy(), lab.numpy()
        arrs[f"{name}.kw"] = np.array(repr(kw))
        try:
            s_roi, s_loc, s_lab = ref_train.ProposalTargetCreator(**kw)(r, gt, lab)
            arrs[f"{name}.raises"] = np.array(False)
            arrs[f"{name}.sample_roi"], arrs[f"{name}.gt_roi_loc"], arrs[f"{name}.gt_roi_label"] = \
                s_roi.numpy(), s_loc.numpy(), s_lab.numpy()
        except IndexError:
            arrs[f"{name}.raises"] = np.array(True)
    save("targets_proposal.npz", **arrs)


@torch.inference_mode()
def main():
    g = torch.Generator().manual_seed(20251003)

    # ---- anchors --------------------------------------------------------------------
    base = ref_anchors.generate_basic_anchor()
    base_alt = ref_anchors.generate_basic_anchor(base_size=16, ratios=[0.5, 1, 2, 3], anchor_scales=[4, 8])
    save("anchors.npz",
         base=base.numpy(), base_alt=base_alt.numpy(),
         shifted_s16_h3_w5=ref_anchors.enumerate_shifted_anchor(base, 16, 3, 5).numpy(),
         shifted_s32_h2_w3=ref_anchors.enumerate_shifted_anchor(base, 32, 2, 3).numpy(),
         shifted_s16_h50_w84_rows=ref_anchors.enumerate_shifted_anchor(base, 16, 50, 84)[[0, 9, 755, 756, 37799]].numpy())

    # ---- box math -------------------------------------------------------------------
    src = torch.rand(64, 4, generator=g) * 400
    src[:, 2:] = src[:, :2] + torch.rand(64, 2, generator=g) * 300 + 1
    loc = torch.randn(64, 4, generator=g) * 0.5
    a = torch.rand(7, 4, generator=g) * 100
    a[:, 2:] += a[:, :2]
    b = torch.rand(5, 4, generator=g) * 100
    b[:, 2:] += b[:, :2]
    d1 = torch.tensor([[100, 100, 200, 200]], dtype=torch.float32)
    d2 = torch.tensor([[150, 150, 250, 250]], dtype=torch.float32)
    save("boxmath.npz", src=src.numpy(), loc=loc.numpy(), loc2bbox=ref_box.loc2bbox(src, loc).numpy(),
         iou_a=a.numpy(), iou_b=b.numpy(), iou=ref_box.bbox_iou(a, b).numpy(),
         known_iou=ref_box.bbox_iou(d1, d2).numpy(),
         known_roundtrip=ref_box.loc2bbox(d1, ref_box.bbox2loc(d1, d2)).numpy())

    # ---- single residual blocks (weights stored) ------------------------------------
    torch.manual_seed(11)
    ds = torch.nn.Sequential(torch.nn.Conv2d(32, 32, 1, 2, bias=False), torch.nn.BatchNorm2d(32))
    bott = ref_resnet.Bottleneck(32, 8, stride=2, downsample=ds).eval()
    randomize_bn(bott, g)
    xb = torch.randn(2, 32, 9, 11, generator=g)
    save("resnet_bottleneck.npz", x=xb.numpy(), y=bott(xb).numpy(), **{"sd." + k: v for k, v in npd(bott.state_dict()).items()})
    basic = ref_resnet.BasicBlock(16, 16).eval()
    randomize_bn(basic, g)
    xc = torch.randn(2, 16, 7, 10, generator=g)
    save("resnet_basicblock.npz", x=xc.numpy(), y=basic(xc).numpy(), **{"sd." + k: v for k, v in npd(basic.state_dict()).items()})

    # ---- one HarDBlock + transition (weights stored) --------------------------------
    torch.manual_seed(12)
    hb = ref_hardnet.HarDBlock(16, 6, 1.7, 8, dwconv=True).eval()
    randomize_bn(hb, g)
    xh = torch.randn(2, 16, 9, 12, generator=g)
    save("hardnet_block.npz", x=xh.numpy(), y=hb(xh).numpy(), out_ch=np.int64(hb.get_out_ch()),
         **{"sd." + k: v for k, v in npd(hb.state_dict()).items()})

    # ---- whole backbones from seeds (weights NOT stored: checksums guard the RNG) ------
    for name, ctor, shape in (
        ("resnet50", lambda: ref_resnet.resnet50(include_top=False), (1, 3, 64, 96)),
        ("hardnet39", lambda: ref_hardnet.HarDNetFeatureExtraction(depth_wise=True, arch=39), (1, 3, 64, 96)),
        ("hardnet68", lambda: ref_hardnet.HarDNetFeatureExtraction(depth_wise=True, arch=68), (1, 3, 64, 96)),
    ):
        torch.manual_seed(0)
        m = ctor().eval()
        x = torch.rand(shape, generator=torch.Generator().manual_seed(1234))
        y = m(x)
        sd = m.state_dict()
        keys = sorted(sd.keys())
        wsum = np.array([float(sd[k].double().sum()) for k in keys])
        save(f"{name}_seeded.npz", seed=np.int64(0), x_seed=np.int64(1234), x_shape=np.array(shape),
             y=y.numpy(), keys=np.array(keys), key_shapes=np.array([str(tuple(sd[k].shape)) for k in keys]),
             weight_sums=wsum, n_params=np.int64(sum(p.numel() for p in m.parameters())))

    # ---- reference RPN glue (nets/rpn.py) around the stand-in nms ----------------------
    torch.manual_seed(13)
    rpn = ref_rpn.RegionProposalNetwork(16, feat_stride=16).eval()       # default mode "training" -> test numbers (Q3)
    rpn.loc.weight.data.mul_(0.3)
    feat = torch.randn(2, 16, 10, 12, generator=g)
    img_size = (3, 160, 192)
    locs, scores, rois, anchor = rpn.forward(feat, img_size, 1.0)
    save("rpn_ref.npz", feat=feat.numpy(), img_size=np.array(img_size), rpn_locs=locs.numpy(),
         rpn_scores=scores.numpy(), rois=rois.numpy(), anchor=anchor.numpy(),
         **{"sd." + k: v for k, v in npd(rpn.state_dict()).items()})
    rpn_t = ref_rpn.RegionProposalNetwork(16, feat_stride=16, mode="train").eval()
    rpn_t.load_state_dict(rpn.state_dict())
    feat_t = torch.randn(1, 16, 30, 36, generator=g)
    out_t = rpn_t.forward(feat_t, (3, 480, 576), 1.0)
    save("rpn_ref_train.npz", feat=feat_t.numpy(), img_size=np.array((3, 480, 576)), rois=out_t[2].numpy(),
         **{"sd." + k: v for k, v in npd(rpn.state_dict()).items()})

    # ---- reference RoI head glue (nets/classify.py) around the stand-in RoIPool ---------
    torch.manual_seed(14)
    head = ref_classify.HarNetRoIHead(n_class=5, roi_size=7, spatial_scale=1,
                                      classifier=ref_hardnet.HarNetClassifier()).eval()
    fh = torch.randn(1, 512, 9, 13, generator=g)
    r = torch.rand(1, 128, 4, generator=g)
    r[..., 0] *= 150; r[..., 1] *= 100
    r[..., 2] = r[..., 0] + r[..., 2] * 60 + 1
    r[..., 3] = r[..., 1] + r[..., 3] * 50 + 1
    idx = torch.zeros(1, dtype=torch.int32)
    cl, sc = head.forward(fh, r, idx, (144, 208))
    save("head_ref.npz", feat=fh.numpy(), rois=r.numpy(), img_size=np.array((144, 208)),
         roi_cls_locs=cl.numpy(), roi_scores=sc.numpy(),
         **{"sd." + k: v for k, v in npd(head.state_dict()).items()})


if __name__ == "__main__":
    if "--extras-only" not in sys.argv:
        main()
    extras_round2()
