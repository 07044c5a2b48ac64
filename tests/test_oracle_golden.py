"""The CPU oracle against vectors produced by the reference's own Python modules
(tests/golden/make_golden.py, run in the build container) and its known answers."""
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import backbones


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _sd(z):
    return {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")}


def test_base_anchors_known_answers(golden_dir):
    z = _load(golden_dir, "anchors.npz")
    base = oracle.generate_basic_anchor()
    assert np.array_equal(base.numpy(), z["base"])                       # bit-exact vs reference
    # SURVEY 8(c) known answers (utils/basic_anchors.py:60-63 prints these)
    assert np.allclose(base[0].numpy(), [-45.2548, -22.6274, 45.2548, 22.6274], atol=1e-4)
    assert np.allclose(base[4].numpy(), [-64, -64, 64, 64])
    assert np.allclose(base[8].numpy(), [-90.5097, -181.0193, 90.5097, 181.0193], atol=1e-4)
    alt = oracle.generate_basic_anchor(base_size=16, ratios=[0.5, 1, 2, 3], anchor_scales=[4, 8])
    assert np.array_equal(alt.numpy(), z["base_alt"])


def test_shifted_anchors(golden_dir):
    z = _load(golden_dir, "anchors.npz")
    base = torch.from_numpy(z["base"])
    assert np.array_equal(oracle.enumerate_shifted_anchor(base, 16, 3, 5).numpy(), z["shifted_s16_h3_w5"])
    assert np.array_equal(oracle.enumerate_shifted_anchor(base, 32, 2, 3).numpy(), z["shifted_s32_h2_w3"])
    full = oracle.enumerate_shifted_anchor(base, 16, 50, 84)
    assert full.shape == (37800, 4)
    assert np.array_equal(full[[0, 9, 755, 756, 37799]].numpy(), z["shifted_s16_h50_w84_rows"])
    # anchor #9 at stride 16 = base[0] + (16,0,16,0): x runs fastest
    assert np.allclose(full[9].numpy(), [-29.2548, -22.6274, 61.2548, 22.6274], atol=1e-4)


def test_loc2bbox_and_iou(golden_dir):
    z = _load(golden_dir, "boxmath.npz")
    got = oracle.loc2bbox(torch.from_numpy(z["src"]), torch.from_numpy(z["loc"]))
    assert np.array_equal(got.numpy(), z["loc2bbox"])
    iou = oracle.bbox_iou(torch.from_numpy(z["iou_a"]), torch.from_numpy(z["iou_b"]))
    assert np.allclose(iou.numpy(), z["iou"], rtol=0, atol=1e-7)
    # reference's own __main__ known answers (utils/loc_bbox_iou.py:99-103)
    d1 = torch.tensor([[100., 100, 200, 200]])
    d2 = torch.tensor([[150., 150, 250, 250]])
    assert abs(float(oracle.bbox_iou(d1, d2)) - 2500 / 17500) < 1e-7
    assert np.allclose(z["known_iou"], 0.142857, atol=1e-6)
    assert np.array_equal(z["known_roundtrip"], d2.numpy())
    assert oracle.loc2bbox(torch.zeros(0, 4), torch.zeros(0, 4)).shape == (0, 4)
    zk = _load(golden_dir, "boxmath_k.npz")                 # [n,4k] locs: k offset sets per source box
    got = oracle.loc2bbox(torch.from_numpy(zk["src"]), torch.from_numpy(zk["loc_k3"]))
    assert got.shape == (40, 12) and np.array_equal(got.numpy(), zk["loc2bbox_k3"])
    assert zk["empty"].shape == (0, 4)
    with pytest.raises(IndexError):
        oracle.bbox_iou(torch.zeros(2, 3), torch.zeros(2, 4))


@pytest.mark.parametrize("name", ["resnet_bottleneck.npz", "resnet_basicblock.npz"])
def test_residual_blocks(golden_dir, name):
    z = _load(golden_dir, name)
    sd = {"blk." + k: v for k, v in _sd(z).items()}
    stride = 2 if "bottleneck" in name else 1
    y = backbones._res_block(sd, "blk", torch.from_numpy(z["x"]), stride)
    assert np.allclose(y.numpy(), z["y"], rtol=0, atol=1e-6)


def test_hardblock(golden_dir):
    z = _load(golden_dir, "hardnet_block.npz")
    sd = {"b." + k: v for k, v in _sd(z).items()}
    y = backbones._hard_block(sd, "b", torch.from_numpy(z["x"]), 8)
    assert y.shape[1] == int(z["out_ch"])
    assert np.allclose(y.numpy(), z["y"], rtol=0, atol=1e-6)
    assert backbones.hard_links(8) == [7, 6, 4, 0] and backbones.hard_links(3) == [2] and backbones.hard_links(12) == [11, 10, 8]


def test_rpn_glue_matches_reference(golden_dir):
    z = _load(golden_dir, "rpn_ref.npz")
    locs, scores, rois, anchor = oracle.rpn_forward(_sd(z), torch.from_numpy(z["feat"]), tuple(int(v) for v in z["img_size"]),
                                                    scale=1.0, feat_stride=16, mode="training")
    assert np.array_equal(locs.numpy(), z["rpn_locs"])
    assert np.array_equal(scores.numpy(), z["rpn_scores"])
    assert np.array_equal(anchor.numpy(), z["anchor"])
    assert rois.shape == (2, 300, 4)
    assert np.array_equal(rois.numpy(), z["rois"])


def test_rpn_glue_train_mode(golden_dir):
    z = _load(golden_dir, "rpn_ref_train.npz")
    out = oracle.rpn_forward(_sd(z), torch.from_numpy(z["feat"]), tuple(int(v) for v in z["img_size"]),
                             scale=1.0, feat_stride=16, mode="train")
    assert out[2].shape == (1, 600, 4)
    assert np.array_equal(out[2].numpy(), z["rois"])


def test_head_glue_matches_reference(golden_dir):
    z = _load(golden_dir, "head_ref.npz")
    cl, sc = oracle.roi_head_forward(_sd(z), torch.from_numpy(z["feat"]), torch.from_numpy(z["rois"]),
                                     torch.zeros(1, dtype=torch.int32), tuple(int(v) for v in z["img_size"]))
    assert np.allclose(cl.numpy(), z["roi_cls_locs"], rtol=0, atol=1e-6)
    assert np.allclose(sc.numpy(), z["roi_scores"], rtol=0, atol=1e-6)


@pytest.mark.parametrize("name,backbone", [("resnet50_seeded.npz", "resnet50"), ("hardnet39_seeded.npz", "hardnet39"),
                                           ("hardnet68_seeded.npz", "hardnet68")])
def test_whole_trunks_from_seeds(golden_dir, name, backbone):
    """Whole-trunk outputs of the reference's own modules (models/resnet.py:135-151, models/hardnet.py:198-201) built
    under ``torch.manual_seed(0)``: the package's module surface must reproduce the reference's seeded weights (key
    names, shapes, per-tensor checksums = the checkpoint + RNG contract) and the oracle's trunk composition (stem,
    max pool, stage strides, HarDNet transitions / tail) must reproduce the reference's feature map bit for bit."""
    z = _load(golden_dir, name)
    if backbone == "resnet50":
        from two_stage_object_detection_amd.models.resnet import resnet50
        torch.manual_seed(int(z["seed"]))
        m = resnet50(include_top=False).eval()
        trunk = oracle.resnet_trunk
        kw = {}
    else:
        from two_stage_object_detection_amd.models.hardnet import HarDNetFeatureExtraction
        torch.manual_seed(int(z["seed"]))
        m = HarDNetFeatureExtraction(depth_wise=True, arch=int(backbone[-2:])).eval()
        trunk = oracle.hardnet_trunk
        kw = {"arch": int(backbone[-2:])}
    sd = m.state_dict()
    keys = sorted(sd.keys())
    assert keys == [str(k) for k in z["keys"]]
    assert [str(tuple(sd[k].shape)) for k in keys] == [str(s) for s in z["key_shapes"]]
    assert sum(p.numel() for p in m.parameters()) == int(z["n_params"])
    sums = np.array([float(sd[k].double().sum()) for k in keys])
    assert np.array_equal(sums, z["weight_sums"])                         # same RNG stream, same init order
    x = torch.rand(tuple(int(v) for v in z["x_shape"]), generator=torch.Generator().manual_seed(int(z["x_seed"])))
    with torch.inference_mode():
        y = trunk({k: v.clone() for k, v in sd.items()}, x, **kw)
    assert y.shape == z["y"].shape
    assert np.array_equal(y.numpy(), z["y"])                              # bit-exact (same torch CPU kernels, same op order)
