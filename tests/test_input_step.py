"""the input step in front of the path (dataset/transform.py:14-17 on the GPU).

CPU part: the library's HOST tap tables evaluated separably in numpy f32 against the oracle (torch's antialiased bilinear
interpolate).  GPU part: the fused u8 -> resized f32 kernel against the same oracle, both output layouts, and the hand-over
to the detector.  Tolerance: 1e-3 absolute on values in 0..255 (observed <= 5e-5: f32 sums of <= 17 taps in the same order)."""
import numpy as np
import pytest
import torch

import oracle
from two_stage_object_detection_amd._ffi import lib

TOL = 1e-3
SIZES = [((97, 131), (60, 60)), ((50, 40), (60, 60)), ((480, 640), (600, 600)), ((233, 517), (100, 77)),
         ((1, 1), (5, 7)), ((7, 5), (1, 1)), ((1080, 1920), (600, 600)), ((600, 600), (600, 600)),
         ((540, 960), (30, 30)), ((31, 1000), (31, 40)), ((33, 47), (257, 129))]   # 18x down-scale: the untiled kernel


def _image(shape, seed=0, C=3):
    return torch.randint(0, 256, shape + (C,), generator=torch.Generator().manual_seed(seed), dtype=torch.uint8)


def _tables(n_in, n_out):
    L = lib()
    taps = L.tsod_resize_aa_taps(n_in, n_out)
    first, count = np.zeros(n_out, np.int32), np.zeros(n_out, np.int32)
    w = np.zeros((n_out, taps), np.float32)
    assert L.tsod_resize_aa_tables_f32(n_in, n_out, first.ctypes.data, count.ctypes.data, w.ctypes.data) == 0
    return first, count, w


def _separable(img, OH, OW):
    H, W, C = img.shape
    yf, yc, yw = _tables(H, OH)
    xf, xc, xw = _tables(W, OW)
    src = img.astype(np.float32)
    tmp = np.zeros((H, OW, C), np.float32)
    for ox in range(OW):
        for j in range(xc[ox]):
            v = src[:, xf[ox] + j, :] * xw[ox, j]
            tmp[:, ox, :] = v if j == 0 else tmp[:, ox, :] + v
    out = np.zeros((OH, OW, C), np.float32)
    for oy in range(OH):
        for j in range(yc[oy]):
            v = tmp[yf[oy] + j] * yw[oy, j]
            out[oy] = v if j == 0 else out[oy] + v
    return out


@pytest.mark.parametrize("src,dst", SIZES)
def test_host_tables_reproduce_the_oracle(src, dst):
    img = _image(src, seed=1)
    ref = oracle.eval_transform(img, dst).permute(1, 2, 0).numpy()
    got = _separable(img.numpy(), *dst)
    assert np.abs(got - ref).max() <= TOL


def test_tables_are_normalised_and_in_range():
    for n_in, n_out in ((1333, 600), (600, 1333), (37, 37), (4000, 600), (3, 600)):
        first, count, w = _tables(n_in, n_out)
        assert w.shape[1] == lib().tsod_resize_aa_taps(n_in, n_out)
        assert (first >= 0).all() and (count >= 1).all() and (first + count <= n_in).all()
        assert np.allclose(w.sum(1), 1.0, atol=1e-6) and (w >= 0).all()
        for i in range(n_out):
            assert (w[i, count[i]:] == 0).all()
    assert lib().tsod_resize_aa_tables_f32(0, 5, None, None, None) < 0


# ------------------------------------------------------------------------------------------- GPU
@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("src,dst", SIZES)
def test_resize_kernel_matches_the_oracle(dev, src, dst):
    from two_stage_object_detection_amd import hip_ops
    img = _image(src, seed=2)
    ref = oracle.eval_transform(img, dst)
    nchw = hip_ops.resize_bilinear_aa(img.to(dev), dst[0], dst[1], layout="nchw").cpu()
    assert nchw.shape == ref.shape and (nchw - ref).abs().max().item() <= TOL
    nhwc = hip_ops.resize_bilinear_aa(img.to(dev), dst[0], dst[1], layout="nhwc4", mul=1.0 / 255).cpu()
    assert nhwc.shape == (dst[0], dst[1], 4)
    assert (nhwc[..., :3] * 255 - ref.permute(1, 2, 0)).abs().max().item() <= TOL
    assert (nhwc[..., 3] == 0).all()


@pytest.mark.gpu
def test_resize_other_channel_counts_and_row_pitch(dev):
    from two_stage_object_detection_amd import hip_ops
    for C in (1, 4):
        img = _image((45, 83), seed=3, C=C)
        ref = torch.nn.functional.interpolate(img.permute(2, 0, 1)[None].float(), size=(32, 48), mode="bilinear",
                                              align_corners=False, antialias=True)[0]
        got = hip_ops.resize_bilinear_aa(img.to(dev), 32, 48, layout="nchw").cpu()
        assert (got - ref).abs().max().item() <= TOL
    wide = _image((40, 100), seed=4).to(dev)
    crop = wide[:, 10:70]                                             # row pitch 300 bytes, 60 pixels used
    ref = oracle.eval_transform(crop.cpu(), (25, 25))
    got = hip_ops.resize_bilinear_aa(crop, 25, 25, layout="nchw").cpu()
    assert (got - ref).abs().max().item() <= TOL
    with pytest.raises(Exception):
        hip_ops.resize_bilinear_aa(wide.float(), 8, 8)


@pytest.mark.gpu
def test_eval_transform_mirrors_the_reference_call(dev):
    from two_stage_object_detection_amd.dataset.transform import EvalTransform, eval_transform
    img = _image((375, 500), seed=5)
    boxes = torch.tensor([[10., 20., 200., 300.], [0., 0., 500., 375.]])
    ref_img, ref_boxes = oracle.eval_transform(img, (600, 600), boxes)
    out = eval_transform({"image": img.to(dev), "boxes": boxes.to(dev), "labels": torch.tensor([3, 7])})
    assert out["image"].shape == (3, 600, 600) and (out["image"].cpu() - ref_img).abs().max().item() <= TOL
    assert torch.equal(out["boxes"].cpu(), ref_boxes) and out["labels"].tolist() == [3, 7]
    assert EvalTransform((64, 96))(img.to(dev)).shape == (3, 64, 96)


@pytest.mark.gpu
def test_batch_feeds_the_detector_without_a_layout_pass(dev):
    """Images of different sizes -> one NHWC(4) batch written straight into the backbone's input buffer; the forward
    from it equals the forward from the same pixels handed over as an NCHW tensor (same kernels, same data)."""
    from two_stage_object_detection_amd.dataset.transform import EvalTransform
    from two_stage_object_detection_amd.testing import synthetic_detector
    model, _ = synthetic_detector("resnet50", num_classes=20, seed=0)
    model = model.to(dev).eval()
    tf = EvalTransform((224, 288), mul=1.0 / 255)
    imgs = [_image(s, seed=6 + i).to(dev) for i, s in enumerate(((300, 400), (240, 320)))]
    x_nchw = torch.stack([tf.image(i) for i in imgs])
    with torch.inference_mode():
        ref = [o.clone() for o in model(x_nchw)]
        staged = tf.batch(imgs, out=model.extractor.input_buffer(2, 224, 288, dev))
        assert staged.data.data_ptr() == model.extractor._plan_for(x_nchw).input_nhwc.data_ptr()
        assert tuple(staged.shape) == (2, 3, 224, 288)
        got = model(staged)
        loose = model(tf.batch(imgs))                                  # a free-standing batch is copied in
        model.raise_if_error()
    for a, b, c in zip(got, ref, loose):
        assert torch.equal(a, b) and torch.equal(c, b)
