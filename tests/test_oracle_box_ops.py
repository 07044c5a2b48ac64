"""Hand-derived known-answer cases for the two torchvision operators the oracle restates
(PARITY UNPINNED: torchvision is not in /root/reference nor in the image), plus C-vs-Python
cross-checks of the restatement."""
import numpy as np
import pytest
import torch
from hypothesis import given, settings, strategies as st

import oracle
from oracle.box import nms_python, roi_pool_python


def test_nms_hand_cases():
    # two boxes with IoU = 0.5 exactly: kept at thr 0.5 (strict >), suppressed at thr 0.49
    b = torch.tensor([[0., 0, 10, 10], [0., 0, 10, 5]])
    s = torch.tensor([0.9, 0.8])
    assert oracle.nms(b, s, 0.5).tolist() == [0, 1]
    assert oracle.nms(b, s, 0.49).tolist() == [0]
    # order of the result follows the scores, ties -> lower index first
    b3 = torch.tensor([[0., 0, 1, 1], [10., 10, 11, 11], [20., 20, 21, 21]])
    assert oracle.nms(b3, torch.tensor([0.1, 0.9, 0.5]), 0.7).tolist() == [1, 2, 0]
    assert oracle.nms(b3, torch.tensor([0.5, 0.5, 0.5]), 0.7).tolist() == [0, 1, 2]
    # touching boxes do not overlap (no +1 convention)
    t = torch.tensor([[0., 0, 10, 10], [10., 0, 20, 10]])
    assert oracle.nms(t, torch.tensor([1., 0.5]), 0.0).tolist() == [0, 1]
    # degenerate zero-area duplicates: 0/0 = NaN never suppresses
    z = torch.tensor([[5., 5, 5, 5], [5., 5, 5, 5]])
    assert oracle.nms(z, torch.tensor([1., 0.5]), 0.1).tolist() == [0, 1]
    # chain: 0 suppresses 1, 1 would have suppressed 2 but is dead -> 2 survives
    c = torch.tensor([[0., 0, 10, 10], [4., 0, 14, 10], [8., 0, 18, 10]])
    assert oracle.nms(c, torch.tensor([0.9, 0.8, 0.7]), 0.4).tolist() == [0, 2]
    assert oracle.nms(torch.zeros(0, 4), torch.zeros(0), 0.5).tolist() == []


def test_roi_pool_hand_cases():
    x = torch.arange(36, dtype=torch.float32).view(1, 1, 6, 6)
    # whole map, 2x2 bins of 3x3 -> maxima of each quadrant
    out = oracle.roi_pool(x, torch.tensor([[0., 0, 0, 5, 5]]), (2, 2), 1.0)
    assert out.view(-1).tolist() == [14., 17., 32., 35.]
    # single pixel roi -> every bin sees that pixel
    out = oracle.roi_pool(x, torch.tensor([[0., 2, 3, 2, 3]]), (2, 2), 1.0)
    assert out.view(-1).tolist() == [20.] * 4
    # round half away from zero: 0.5 -> 1, 2.5 -> 3  => window rows/cols 1..3
    out = oracle.roi_pool(x, torch.tensor([[0., 0.5, 0.5, 2.5, 2.5]]), (1, 1), 1.0)
    assert out.item() == 21.0
    # roi completely outside the map -> empty bins -> 0
    out = oracle.roi_pool(x, torch.tensor([[0., 50, 50, 60, 60]]), (2, 2), 1.0)
    assert out.abs().sum().item() == 0.0
    # negative coordinates are clamped to the map
    out = oracle.roi_pool(x, torch.tensor([[0., -4, -4, 1, 1]]), (1, 1), 1.0)
    assert out.item() == 7.0
    # spatial_scale multiplies before rounding
    out = oracle.roi_pool(x, torch.tensor([[0., 0, 0, 20, 20]]), (1, 1), 0.25)
    assert out.item() == 35.0
    # inverted roi (x2 < x1): extent clamps to 1
    out = oracle.roi_pool(x, torch.tensor([[0., 3, 3, 1, 1]]), (1, 1), 1.0)
    assert out.item() == 21.0
    with pytest.raises(IndexError):
        oracle.roi_pool(x, torch.tensor([[1., 0, 0, 1, 1]]), (1, 1), 1.0)


def test_nms_pad_rule():
    import ctypes
    from oracle.box import _lib
    boxes = torch.tensor([[0., 0, 10, 10], [1., 0, 11, 10], [50., 50, 60, 60]])
    out = torch.empty(5, dtype=torch.int64)
    kept = _lib().oracle_nms_pad(boxes.data_ptr(), 3, 0.5, 5, out.data_ptr())
    assert kept == 2 and out.tolist() == [0, 2, 0, 1, 2]
    out7 = torch.empty(7, dtype=torch.int64)
    assert _lib().oracle_nms_pad(boxes.data_ptr(), 3, 0.5, 7, out7.data_ptr()) == -2   # IndexError in the reference


def _boxes(rng, n, span=100.0):
    xy = rng.random((n, 2), dtype=np.float32) * span
    wh = rng.random((n, 2), dtype=np.float32) * span * 0.6
    return np.concatenate([xy, xy + wh], axis=1).astype(np.float32)


@settings(max_examples=40, deadline=None)
@given(seed=st.integers(0, 10_000), n=st.integers(1, 60), thr=st.sampled_from([0.0, 0.3, 0.5, 0.7, 1.0]),
       quant=st.booleans())
def test_nms_c_equals_python(seed, n, thr, quant):
    rng = np.random.default_rng(seed)
    b = _boxes(rng, n)
    s = rng.random(n, dtype=np.float32)
    if quant:                      # force ties and exact-threshold IoUs
        b = np.round(b / 10) * 10
        s = np.round(s * 4) / 4
    got = oracle.nms(torch.from_numpy(b), torch.from_numpy(s), thr).tolist()
    assert got == nms_python(b, s, thr)


@settings(max_examples=30, deadline=None)
@given(seed=st.integers(0, 10_000), k=st.integers(1, 6), ph=st.integers(1, 7), pw=st.integers(1, 7),
       scale=st.sampled_from([1.0, 0.5, 0.0625]))
def test_roi_pool_c_equals_python(seed, k, ph, pw, scale):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((2, 3, 9, 11)).astype(np.float32)
    r = _boxes(rng, k, span=14.0 / scale) - np.float32(2.0 / scale)
    rois = np.concatenate([rng.integers(0, 2, (k, 1)).astype(np.float32), r], axis=1)
    got = oracle.roi_pool(torch.from_numpy(x), torch.from_numpy(rois), (ph, pw), scale).numpy()
    assert np.array_equal(got, roi_pool_python(x, rois, (ph, pw), scale))


def test_proposal_layer_small():
    g = torch.Generator().manual_seed(3)
    base = oracle.generate_basic_anchor()
    anchor = oracle.enumerate_shifted_anchor(base, 16, 12, 14)
    loc = torch.randn(anchor.shape[0], 4, generator=g) * 0.3
    score = torch.rand(anchor.shape[0], generator=g)
    rois, dbg = oracle.proposal_layer(loc, score, anchor, (3, 192, 224), return_debug=True)
    assert rois.shape == (300, 4)
    assert (rois[:, 0::2] >= 0).all() and (rois[:, 0::2] <= 192).all()     # x clamped to img_size[1] (Q1)
    assert (rois[:, 1::2] <= 224).all()
    s = dbg["score_sorted"]
    assert (s[:-1] >= s[1:]).all()
    # fewer candidates than the pad needs -> IndexError like the reference (Q4)
    with pytest.raises(IndexError):
        oracle.proposal_layer(loc[:50], score[:50], anchor[:50], (3, 192, 224))


def test_postprocess_per_class_equals_one_nms_per_class():
    """oracle.postprocess(per_class=True) against the definition: filter, sort, nms inside each class, merge by score."""
    import oracle
    g = torch.Generator().manual_seed(5)
    xy = torch.rand(1, 120, 2, generator=g) * 100
    det = torch.cat([xy, xy + torch.rand(1, 120, 2, generator=g) * 80 + 2, torch.randn(1, 120, 1, generator=g),
                     torch.randint(0, 4, (1, 120, 1), generator=g).float()], dim=-1)
    out = oracle.postprocess(det, 0.3, score_thresh=-0.2, per_class=True, background_class=0)[0]
    d = det[0]
    d = d[(d[:, 4] >= -0.2) & (d[:, 5] != 0)]
    expect = []
    for c in (1.0, 2.0, 3.0):
        dc = d[d[:, 5] == c]
        keep = nms_python(dc[:, :4].numpy(), dc[:, 4].numpy(), 0.3)
        expect.append(dc[keep])
    expect = torch.cat(expect)
    expect = expect[torch.sort(expect[:, 4], descending=True, stable=True).indices]
    assert out.shape == expect.shape and out.shape[0] > 3
    assert torch.equal(torch.sort(out[:, 4], descending=True).values, expect[:, 4])
    assert {tuple(r.tolist()) for r in out} == {tuple(r.tolist()) for r in expect}
    # class-agnostic default = one nms over everything
    out0 = oracle.postprocess(det, 0.3)[0]
    keep0 = nms_python(det[0, :, :4].numpy(), det[0, :, 4].numpy(), 0.3)
    assert torch.equal(out0, det[0][keep0])


# ----------------------------------------------------------------------------- roi_align (added option; parity unpinned)
def _roi_align_python(x, rois, out_size, scale, sampling_ratio, aligned):
    """Independent pure-Python restatement of torchvision.ops.roi_align's published definition (bilinear samples on a
    regular grid per bin, averaged) in float64 - small cases only - to cross-check the C code's structure."""
    import math
    x = np.asarray(x, dtype=np.float64)
    K, (PH, PW) = len(rois), out_size
    _, C, H, W = x.shape
    out = np.zeros((K, C, PH, PW))

    def bil(plane, y, xx):
        if y < -1 or y > H or xx < -1 or xx > W:
            return 0.0
        y, xx = max(y, 0.0), max(xx, 0.0)
        yl, xl = int(y), int(xx)
        if yl >= H - 1:
            yh = yl = H - 1; y = float(yl)
        else:
            yh = yl + 1
        if xl >= W - 1:
            xh = xl = W - 1; xx = float(xl)
        else:
            xh = xl + 1
        ly, lx = y - yl, xx - xl
        return (1 - ly) * (1 - lx) * plane[yl, xl] + (1 - ly) * lx * plane[yl, xh] + ly * (1 - lx) * plane[yh, xl] + ly * lx * plane[yh, xh]
    for k, r in enumerate(np.asarray(rois, dtype=np.float64)):
        b = int(r[0])
        off = 0.5 if aligned else 0.0
        sw, sh, ew, eh = r[1] * scale - off, r[2] * scale - off, r[3] * scale - off, r[4] * scale - off
        rw, rh = ew - sw, eh - sh
        if not aligned:
            rw, rh = max(rw, 1.0), max(rh, 1.0)
        bh, bw = rh / PH, rw / PW
        gh = sampling_ratio if sampling_ratio > 0 else math.ceil(rh / PH)
        gw = sampling_ratio if sampling_ratio > 0 else math.ceil(rw / PW)
        cnt = max(gh * gw, 1)
        for c in range(C):
            for ph in range(PH):
                for pw in range(PW):
                    acc = 0.0
                    for iy in range(gh):
                        for ix in range(gw):
                            acc += bil(x[b, c], sh + ph * bh + (iy + .5) * bh / gh, sw + pw * bw + (ix + .5) * bw / gw)
                    out[k, c, ph, pw] = acc / cnt
    return out


@pytest.mark.parametrize("sampling_ratio,aligned", [(2, False), (0, False), (2, True), (3, True)])
def test_roi_align_c_matches_python_definition(sampling_ratio, aligned):
    g = torch.Generator().manual_seed(40 + sampling_ratio)
    x = torch.randn(2, 3, 9, 11, generator=g)
    rois = torch.tensor([[0, 1.2, 0.7, 7.9, 6.4], [1, -3.0, -2.0, 4.0, 3.0], [0, 8.5, 6.0, 14.0, 12.0], [1, 2.0, 2.0, 2.2, 2.1],
                         [0, 0.0, 0.0, 10.0, 8.0], [1, 30.0, 30.0, 40.0, 40.0]])
    got = oracle.roi_align(x, rois, (3, 4), 0.9, sampling_ratio, aligned)
    ref = _roi_align_python(x.numpy(), rois.numpy(), (3, 4), 0.9, sampling_ratio, aligned)
    assert np.abs(got.numpy() - ref).max() < 1e-5
    assert (got[5] == 0).all()                                    # an RoI entirely outside the map samples nothing


def test_roi_align_known_answers():
    """Hand-derived: on a linear ramp f(y, x) = 10 y + x bilinear sampling is exact, so every bin's value is the ramp at the
    bin centre; a 1x1 output over the whole map is the map's mean of the sampled points."""
    H, W = 6, 8
    ramp = (10.0 * torch.arange(H).view(H, 1) + torch.arange(W).view(1, W)).view(1, 1, H, W).float()
    rois = torch.tensor([[0, 1.0, 1.0, 5.0, 4.0]])
    out = oracle.roi_align(ramp, rois, (3, 2), 1.0, 2, False)[0, 0]
    # bins: height 3 / 3 = 1 starting at y = 1, width 4 / 2 = 2 starting at x = 1 -> centres (1.5 + ph, 2 + 2 pw)
    ref = torch.tensor([[10 * (1.5 + ph) + (2.0 + 2 * pw) for pw in range(2)] for ph in range(3)])
    assert torch.allclose(out, ref, atol=1e-5)
    const = torch.full((1, 2, 5, 5), 3.25)
    assert torch.allclose(oracle.roi_align(const, torch.tensor([[0, 0.3, 0.4, 3.7, 4.1]]), (7, 7), 1.0, 0, True),
                          torch.full((1, 2, 7, 7), 3.25))
