"""GPU parity: every HIP entry point (through the C ABI) against the CPU oracle on seeded inputs.

Bars: bit-exact for index / integer results (sort order, NMS keep lists, RoI max pooling which is
pure gather+compare); <= 1e-3 absolute for f32 box/score arithmetic (in practice ~1e-5);
convolutions against an f64 CPU convolution with a tolerance scaled by sqrt(K)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import oracle
from oracle.box import _lib as oracle_lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ops():
    from two_stage_object_detection_amd import hip_ops
    return hip_ops


def _rand_boxes(g, n, span=800.0, wh=300.0):
    xy = torch.rand(n, 2, generator=g) * span
    return torch.cat([xy, xy + torch.rand(n, 2, generator=g) * wh + 1.0], dim=1)


# ----------------------------------------------------------------------------- layout
@pytest.mark.parametrize("shape,cpad", [((2, 3, 17, 23), 4), ((1, 3, 800, 1333), 4), ((2, 70, 9, 11), 72), ((1, 64, 5, 7), 64)])
def test_layout_roundtrip(ops, dev, shape, cpad):
    x = torch.randn(shape, generator=torch.Generator().manual_seed(1))
    y = ops.nchw_to_nhwc(x.to(dev), cpad)
    ref = torch.zeros(shape[0], shape[2], shape[3], cpad)
    ref[..., :shape[1]] = x.permute(0, 2, 3, 1)
    assert torch.equal(y.cpu(), ref)
    back = ops.nhwc_to_nchw(y, C=shape[1])
    assert torch.equal(back.cpu(), x)


# ----------------------------------------------------------------------------- conv
CONV_CASES = [
    # N, H, W, Cin, Cout, k, stride, pad
    (1, 20, 27, 64, 64, 1, 1, 0),
    (2, 20, 27, 64, 256, 1, 1, 0),
    (1, 33, 41, 128, 128, 3, 1, 1),
    (1, 33, 41, 128, 96, 3, 2, 1),
    (2, 25, 42, 256, 512, 1, 2, 0),
    (1, 25, 42, 512, 54, 1, 1, 0),       # RPN-like narrow N
    (1, 7, 9, 2048, 512, 1, 1, 0),        # long K, tiny M
    (1, 13, 21, 512, 512, 3, 1, 1),       # K = 4608
    (1, 9, 12, 24, 40, 3, 1, 1),          # Cin not a multiple of 32 (K tail inside a step)
]


def _conv_ref(x, w, stride, pad):
    return F.conv2d(x.double(), w.double(), None, stride, pad).float()


_CONV_REF_CACHE = {}


def _conv_case(case, dev, ops):
    """(reference f64 conv on the CPU, NHWC input on the GPU, packed weight on the GPU), cached per case: the same
    operands go through every tile variant and K-slice schedule."""
    if case not in _CONV_REF_CACHE:
        N, H, W, Cin, Cout, k, stride, pad = case
        g = torch.Generator().manual_seed(hash(case) % 1000)
        x = torch.randn(N, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
        _CONV_REF_CACHE[case] = (_conv_ref(x, w, stride, pad), ops.nchw_to_nhwc(x.to(dev)), ops.pack_conv_weight(w.to(dev)))
    return _CONV_REF_CACHE[case]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("tile", [0] + list(__import__("two_stage_object_detection_amd._ffi", fromlist=["TILE_IDS"]).TILE_IDS))
def test_conv_matches_cpu(ops, dev, case, tile):
    """Every tile variant of include/tsod.h (TILE_IDS: 1..15, 0 = the cost model's pick) under every K-slice schedule the
    autotuner may pin (1 = whole tiles, -1 = hybrid, S = S slices: bench.py's tables hold 3, 4, 6, 8 and 12)."""
    N, H, W, Cin, Cout, k, stride, pad = case
    ref, xn, wp = _conv_case(case, dev, ops)
    ksteps = (Cin * k * k + 31) // 32
    tol = 3e-6 * math.sqrt(Cin * k * k) + 1e-5
    for split in [0, 1, -1] + [s for s in (2, 3, 4, 6, 8, 12, 16) if ksteps // s >= 2]:
        y = ops.conv2d_nhwc(xn, wp, stride=stride, pad=pad, tile=tile, split_k=split)
        got = ops.nhwc_to_nchw(y).cpu()
        assert (got - ref).abs().max().item() <= tol, (case, tile, split)


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("tile", [0] + list(__import__("two_stage_object_detection_amd._ffi", fromlist=["BF16X3_TILE_IDS"]).BF16X3_TILE_IDS))
def test_conv_bf16x3_matches_cpu(ops, dev, case, tile):
    """desc.precision = TSOD_PREC_BF16X3 (the reduced-precision conv path, gated by parity): every
    operand cut exactly into three bf16 pieces, six piece products per k on the bf16 matrix pipes, f32 accumulation.
    Same bar as the f32 MFMA path against the f64 CPU convolution - it has to be f32-accurate to be usable at all - and it
    must actually be a different arithmetic (not bit-equal to the f32 path on every case)."""
    from two_stage_object_detection_amd._ffi import DMA_TILE_IDS, TsodError
    N, H, W, Cin, Cout, k, stride, pad = case
    ref, xn, wp = _conv_case(case, dev, ops)
    ksteps = (Cin * k * k + 31) // 32
    tol = 3e-6 * math.sqrt(Cin * k * k) + 1e-5
    if tile in DMA_TILE_IDS and Cin % 32:
        # the LDS-DMA tiles (a K stage inside ONE filter tap) refuse a channel count that is not whole stages - loudly
        with pytest.raises(TsodError, match="unsupported|UNSUPPORTED"):
            ops.conv2d_nhwc(xn, wp, stride=stride, pad=pad, tile=tile, split_k=1, precision=1)
        return
    f32 = ops.nhwc_to_nchw(ops.conv2d_nhwc(xn, wp, stride=stride, pad=pad, tile=3, split_k=1)).cpu()
    if tile not in DMA_TILE_IDS and tile != 0:                # the balanced schedule exists for the LDS-DMA tiles only
        with pytest.raises(TsodError, match="unsupported|UNSUPPORTED"):
            ops.conv2d_nhwc(xn, wp, stride=stride, pad=pad, tile=tile, split_k=-2, precision=1)
    for split in [0, 1, -1] + [s for s in (2, 3, 6) if ksteps // s >= 2] + ([-2] if tile in DMA_TILE_IDS else []):
        y = ops.conv2d_nhwc(xn, wp, stride=stride, pad=pad, tile=tile, split_k=split, precision=1)
        got = ops.nhwc_to_nchw(y).cpu()
        assert (got - ref).abs().max().item() <= tol, (case, tile, split)
        assert (got - ref).abs().max().item() <= 2.0 * (f32 - ref).abs().max().item() + 1e-6, "not f32-accurate"
    if Cin * k * k >= 512:
        assert not torch.equal(got, f32)


def test_conv_bf16x3_exact_on_bf16_representable_inputs(ops, dev):
    """With operands that ARE bf16 values the mid / lo pieces are zero and the six-product sum is the plain product: the
    bf16x3 path must then agree with the f32 path to accumulation-order noise; with operands carrying all 24 significand
    bits it must still agree (that is the point of the three pieces) - a single-piece (plain bf16) GEMM would be off by ~1e-2."""
    g = torch.Generator().manual_seed(3)
    x = torch.randn(1, 256, 20, 24, generator=g)
    w = torch.randn(128, 256, 3, 3, generator=g) / 48
    ref = _conv_ref(x, w, 1, 1)
    xn, wp = ops.nchw_to_nhwc(x.to(dev)), ops.pack_conv_weight(w.to(dev))
    got = ops.nhwc_to_nchw(ops.conv2d_nhwc(xn, wp, pad=1, precision=1)).cpu()
    err = (got - ref).abs().max().item()
    plain_bf16 = _conv_ref(x.bfloat16().float(), w.bfloat16().float(), 1, 1)
    assert err < 2e-5 and (plain_bf16 - ref).abs().max().item() > 100 * err
    # unsupported tile for this precision is refused, not silently run in f32
    from two_stage_object_detection_amd._ffi import TsodError
    with pytest.raises(TsodError, match="no kernel|unsupported|UNSUPPORTED"):
        ops.conv2d_nhwc(xn, wp, pad=1, tile=1, precision=1)


def test_conv_kslice_reduce_is_deterministic_under_load(ops, dev):
    """K-sliced tiles: partial slabs stored write-through by the slice workgroups, summed in slice order by the slice
    that arrives last at the tile's ticket (in-launch combine across XCDs).  Hundreds of back-to-back launches of several
    slice configurations, each compared word for word with its first result (fixed summation order => bit-reproducible)
    and against the CPU.  Slabs and tickets are re-used across launches, so a stale (previous-launch) slab line read
    through some L2 / L1, or a ticket not left at zero, would show up as a mismatch."""
    g = torch.Generator().manual_seed(21)
    for (H, W, Cin, Cout, k) in ((25, 42, 512, 512, 3), (50, 84, 256, 256, 3), (13, 17, 1024, 96, 1)):
        x = torch.randn(1, Cin, H, W, generator=g)
        w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
        ref = _conv_ref(x, w, 1, k // 2)
        xn, wp = ops.nchw_to_nhwc(x.to(dev)), ops.pack_conv_weight(w.to(dev))
        noise = torch.randn(64 << 20, device=dev)                 # evicts L2 between some launches
        for tile, split in ((3, 8), (3, -1), (1, 4), (6, 16), (4, -1)):
            first = None
            for it in range(60):
                y = ops.conv2d_nhwc(xn, wp, pad=k // 2, tile=tile, split_k=split)
                if it % 7 == 3:
                    noise.mul_(1.0001)
                if first is None:
                    first = y.clone()
                    assert (ops.nhwc_to_nchw(y).cpu() - ref).abs().max().item() <= 3e-6 * math.sqrt(Cin * k * k) + 1e-5
                else:
                    assert torch.equal(y, first), (H, W, Cin, Cout, k, tile, split, it)


def test_conv_epilogue_bn_residual_prelu(ops, dev):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 64, 19, 23, generator=g)
    w = torch.randn(96, 64, 3, 3, generator=g) / 24
    scale = torch.rand(96, generator=g) + 0.5
    shift = torch.randn(96, generator=g)
    res = torch.randn(2, 96, 19, 23, generator=g)
    conv = F.conv2d(x, w, None, 1, 1)
    pre = conv * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res
    xn, rn, wp = ops.nchw_to_nhwc(x.to(dev)), ops.nchw_to_nhwc(res.to(dev)), ops.pack_conv_weight(w.to(dev))
    from two_stage_object_detection_amd._ffi import ACT_PRELU, ACT_RELU6, ACT_RELU, ACT_NONE
    for act, fn in ((ACT_PRELU, lambda v: F.prelu(v, torch.tensor([0.2]))), (ACT_RELU6, F.relu6), (ACT_RELU, F.relu),
                    (ACT_NONE, lambda v: v)):
        for split in (1, 2):
            y = ops.conv2d_nhwc(xn, wp, stride=1, pad=1, scale=scale.to(dev), shift=shift.to(dev), residual=rn, act=act,
                                slope=0.2, split_k=split)
            assert (ops.nhwc_to_nchw(y).cpu() - fn(pre)).abs().max().item() < 1e-4


def test_conv_stem_7x7_as_7x8(ops, dev):
    """The ResNet stem: 3 input channels padded to 4, 7x7 taps padded to 7x8 (zero weights)."""
    g = torch.Generator().manual_seed(6)
    x = torch.rand(2, 3, 61, 77, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) / 12
    ref = _conv_ref(x, w, 2, 3)
    xn = ops.nchw_to_nhwc(x.to(dev), 4)
    wp = ops.pack_conv_weight(w.to(dev), cin_pad=4, kw_pad=8)
    y = ops.conv2d_nhwc(xn, wp, stride=2, pad=3, kw_logical=7)
    assert y.shape == (2, 31, 39, 64)
    assert (ops.nhwc_to_nchw(y).cpu() - ref).abs().max().item() < 5e-5


def test_conv_channel_segments_and_offsets(ops, dev):
    """Input gathered from 3 channel slices of a wider pixel (newest-first concat order of a
    HarDBlock) and output written at a channel offset of a wider buffer."""
    g = torch.Generator().manual_seed(7)
    wide = torch.randn(1, 15, 18, 80, generator=g)               # NHWC, pitch 80
    segs = [(40, 16), (8, 24), (64, 12)]
    cat = torch.cat([wide[..., o:o + l] for o, l in segs], dim=-1).permute(0, 3, 1, 2).contiguous()
    w = torch.randn(20, 52, 1, 1, generator=g) / 7
    ref = _conv_ref(cat, w, 1, 0)
    out = torch.full((1, 15, 18, 48), 7.0, device=dev)
    ops.conv2d_nhwc(wide.to(dev), ops.pack_conv_weight(w.to(dev)), segs=segs, out=out, out_off=12)
    o = out.cpu()
    assert (o[..., 12:32].permute(0, 3, 1, 2) - ref).abs().max().item() < 2e-5
    assert (o[..., :12] == 7.0).all() and (o[..., 32:] == 7.0).all()


def test_linear(ops, dev):
    g = torch.Generator().manual_seed(8)
    x = torch.randn(300, 2048, generator=g)
    w = torch.randn(405, 2048, generator=g) / 45
    b = torch.randn(405, generator=g)
    ref = F.linear(x.double(), w.double(), b.double()).float()
    got = ops.linear(x.to(dev), w.to(dev), b.to(dev)).cpu()
    assert (got - ref).abs().max().item() < 2e-4


def test_conv_rejects_bad_arguments(ops, dev):
    from two_stage_object_detection_amd._ffi import TsodError
    x = torch.zeros(1, 4, 4, 6, device=dev)           # pitch 6: not a multiple of 4
    w = torch.zeros(8, 1, 1, 4, device=dev)
    with pytest.raises(TsodError, match="align"):
        ops.conv2d_nhwc(x, w, segs=[(0, 4)])


# ----------------------------------------------------------------------------- HBM-bound layers
@pytest.mark.parametrize("shape", [(2, 64, 37, 51), (1, 64, 400, 667), (1, 64, 401, 666), (3, 8, 5, 4)])
def test_maxpool(ops, dev, shape):
    """nn.MaxPool2d(3, 2, 1) incl. the stem's 400x667 map, odd and even extents; pure compare work: bit-exact.
    (A form with 4 x 2 outputs per thread marching down its input rows - 5.6 loads per output instead of 9 - was measured at
    35 us against 21 us for this one-output-per-thread form on that map: the re-reads are L2 hits, parallelism matters more.)"""
    x = torch.randn(shape, generator=torch.Generator().manual_seed(9))
    y = ops.maxpool3x3s2_nhwc(ops.nchw_to_nhwc(x.to(dev)))
    assert torch.equal(ops.nhwc_to_nchw(y).cpu(), F.max_pool2d(x, 3, 2, 1))


@pytest.mark.parametrize("stride,relu", [(1, False), (2, True)])
def test_dwconv(ops, dev, stride, relu):
    g = torch.Generator().manual_seed(10)
    x = torch.randn(2, 24, 21, 30, generator=g)
    w = torch.randn(24, 1, 3, 3, generator=g)
    scale, shift = torch.rand(24, generator=g) + 0.5, torch.randn(24, generator=g)
    ref = F.conv2d(x, w, None, stride, 1, 1, 24) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if relu:
        ref = F.relu(ref)
    w33c = w.view(24, 9).t().contiguous().view(3, 3, 24)
    y = ops.dwconv3x3_nhwc(ops.nchw_to_nhwc(x.to(dev)), w33c.to(dev), scale.to(dev), shift.to(dev), stride, relu)
    assert (ops.nhwc_to_nchw(y).cpu() - ref).abs().max().item() < 1e-5


def test_gconv_pair(ops, dev):
    g = torch.Generator().manual_seed(11)
    x = torch.randn(1, 64, 6, 9, generator=g)
    w = torch.randn(32, 2, 1, 1, generator=g)
    b = torch.randn(32, generator=g)
    ref = F.conv2d(x, w, b, 1, 0, 1, 32)
    y = ops.gconv1x1_pair_nhwc(ops.nchw_to_nhwc(x.to(dev)), w.view(32, 2).contiguous().to(dev), b.to(dev))
    assert (ops.nhwc_to_nchw(y).cpu() - ref).abs().max().item() < 1e-5


# ----------------------------------------------------------------------------- proposal path
@pytest.mark.parametrize("B,Hf,Wf,stride", [(1, 25, 42, 32), (2, 50, 84, 16), (3, 5, 7, 16)])
def test_rpn_decode(ops, dev, B, Hf, Wf, stride):
    g = torch.Generator().manual_seed(12)
    base = oracle.generate_basic_anchor()
    A = base.shape[0]
    locs = torch.randn(B * Hf * Wf, 4 * A, generator=g) * 0.4
    scores = torch.randn(B * Hf * Wf, 2 * A, generator=g) * 2
    H, W = Hf * stride, Wf * stride
    boxes, fg, keys, anchors = ops.rpn_decode(locs.to(dev), scores.to(dev), base.to(dev), B, Hf, Wf, stride,
                                              clamp_x=H, clamp_y=W, min_size=16.0, want_anchors=True)
    anchor = oracle.enumerate_shifted_anchor(base, stride, Hf, Wf)
    assert torch.equal(anchors.cpu(), anchor)                                   # exact f32 adds
    fg_ref = F.softmax(scores.view(B, -1, 2), dim=-1)[:, :, 1]
    assert (fg.cpu() - fg_ref).abs().max().item() < 1e-6
    n_border = 0
    for b in range(B):
        roi = oracle.loc2bbox(anchor, locs.view(B, -1, 4)[b])
        roi[:, 0::2] = roi[:, 0::2].clamp(0, H)
        roi[:, 1::2] = roi[:, 1::2].clamp(0, W)
        assert (boxes[b].cpu() - roi).abs().max().item() < 1e-3
        ok = ((roi[:, 2] - roi[:, 0]) >= 16) & ((roi[:, 3] - roi[:, 1]) >= 16)
        ok_gpu = torch.isfinite(keys[b].cpu())
        diff = ok != ok_gpu
        # the mask may differ only where a side is within f32 noise of the threshold
        side = torch.minimum(roi[:, 2] - roi[:, 0], roi[:, 3] - roi[:, 1])
        assert ((side[diff] - 16).abs() < 1e-3).all()
        n_border += int(diff.sum())
        assert torch.equal(keys[b].cpu()[ok_gpu], fg[b].cpu()[ok_gpu])
    assert n_border <= 2


@pytest.mark.parametrize("B,n,n_pre", [(1, 9450, 3000), (2, 37800, 3000), (1, 37800, 12000), (3, 500, 3000), (2, 4096, 64),
                                      (1, 9450, 2000), (2, 20000, 6000), (1, 70000, 16384),
                                      (1, 200000, 3000), (2, 100000, 12000)])     # > 81920 keys: the streaming kernel
def test_sort_topk_exact(ops, dev, B, n, n_pre):
    g = torch.Generator().manual_seed(13)
    keys = torch.rand(B, n, generator=g)
    keys = (keys * 4096).round() / 4096                 # many exact ties -> exercises the stable rule
    keys[torch.rand(B, n, generator=g) < 0.3] = float("-inf")
    keys[0, : n // 2] = 1.0 if B > 1 else keys[0, : n // 2]
    boxes = torch.randn(B, n, 4, generator=g)
    counts, idx, bs, ks = ops.sort_topk_desc(keys.to(dev), boxes.to(dev), n_pre)
    for b in range(B):
        valid = torch.nonzero(torch.isfinite(keys[b])).squeeze(1)
        order = torch.sort(keys[b][valid], descending=True, stable=True).indices[:n_pre]
        ref = valid[order]
        c = int(counts[b])
        assert c == ref.numel()
        assert torch.equal(idx[b, :c].cpu().long(), ref)
        assert (idx[b, c:].cpu() == -1).all()
        assert torch.equal(bs[b, :c].cpu(), boxes[b][ref])
        assert torch.equal(ks[b, :c].cpu(), keys[b][ref])


@pytest.mark.parametrize("B,n,n_pre", [(2, 37800, 3000), (1, 9450, 3000), (4, 9450, 3000), (1, 40, 300)])
def test_sort_topk_rank_paths_agree_with_the_network(ops, dev, B, n, n_pre):
    """Three ways to the same order: the full rank sort (small B*n^2), select + rank on the chip (scratch given), select +
    bitonic network in one workgroup (no scratch: tsod_sort_topk_desc_f32).  Heavy ties, +0 / -0, everything-equal rows and
    rows shorter than n_pre: index work, so all outputs must be identical, neutral rows included."""
    from two_stage_object_detection_amd import _ffi
    g = torch.Generator().manual_seed(17)
    keys = (torch.rand(B, n, generator=g) * 64).round() / 64
    keys[torch.rand(B, n, generator=g) < 0.2] = float("-inf")
    keys[0, ::7] = 0.0
    keys[0, 3::7] = -0.0                                # ties with +0 (torch compares them equal): lower index first
    if B > 1:
        keys[1] = 0.25                                  # one huge tie group straddling the cut
    boxes = torch.randn(B, n, 4, generator=g)
    kd, bd = keys.to(dev), boxes.to(dev)
    got = ops.sort_topk_desc(kd, bd, n_pre)             # wrapper: brings scratch when the library asks for it
    L = _ffi.lib()
    counts = torch.empty((B,), dtype=torch.int32, device=dev)
    idx = torch.empty((B, n_pre), dtype=torch.int32, device=dev)
    bs = torch.empty((B, n_pre, 4), device=dev)
    ks = torch.empty((B, n_pre), device=dev)
    _ffi.check(L.tsod_sort_topk_desc_f32(_ffi.ptr(kd), _ffi.ptr(bd), B, n, n_pre, _ffi.ptr(counts), _ffi.ptr(idx), _ffi.ptr(bs),
                                         _ffi.ptr(ks), _ffi.stream_ptr()))
    for a, b_ in zip(got, (counts, idx, bs, ks)):
        assert torch.equal(a, b_)
    for b in range(B):                                  # and both equal torch's stable sort of the finite keys
        valid = torch.nonzero(torch.isfinite(keys[b])).squeeze(1)
        ref = valid[torch.sort(keys[b][valid], descending=True, stable=True).indices[:n_pre]]
        c = int(counts[b])
        assert c == ref.numel() and torch.equal(idx[b, :c].cpu().long(), ref)
        assert (idx[b, c:] == -1).all() and (bs[b, c:] == 0).all() and torch.isinf(ks[b, c:]).all() and (ks[b, c:] < 0).all()


def test_sort_topk_all_filtered(ops, dev):
    keys = torch.full((2, 100), float("-inf"))
    keys[1, 7] = 0.5
    counts, idx, _, _ = ops.sort_topk_desc(keys.to(dev), torch.zeros(2, 100, 4, device=dev), 16)
    assert counts.tolist() == [0, 1] and idx[1, 0].item() == 7 and (idx[0] == -1).all()


def _nms_pad_ref(boxes_sorted, n, thr, n_post):
    out = torch.empty(n_post, dtype=torch.int64)
    b = boxes_sorted[:n].contiguous()
    kept = oracle_lib().oracle_nms_pad(b.data_ptr(), n, thr, n_post, out.data_ptr())
    return kept, out


@pytest.mark.parametrize("n_max,n_post,span", [(3000, 300, 800.0), (3000, 300, 150.0), (12000, 600, 400.0), (200, 300, 50.0)])
def test_nms_exact(ops, dev, n_max, n_post, span):
    g = torch.Generator().manual_seed(14)
    B = 3
    boxes = torch.stack([_rand_boxes(g, n_max, span=span, wh=span * 0.4) for _ in range(B)])
    boxes[1] = (boxes[1] / 8).round() * 8                      # exact-threshold / duplicate boxes
    counts = torch.tensor([n_max, n_max - 37, max(n_max // 3, 150)], dtype=torch.int32)
    keep, rois, n_kept, status = ops.nms_sorted(boxes.to(dev), counts.to(dev), 0.7, n_post)
    for b in range(B):
        n = int(counts[b])
        kept, ref = _nms_pad_ref(boxes[b], n, 0.7, n_post)
        if kept == -2:
            assert status.item() & 1
            continue
        assert int(n_kept[b]) == min(kept, n_post)
        assert torch.equal(keep[b].cpu().long(), ref), f"image {b}"
        assert torch.equal(rois[b].cpu(), boxes[b][ref])


def test_nms_pad_overflow_sets_status(ops, dev):
    boxes = _rand_boxes(torch.Generator().manual_seed(15), 10).unsqueeze(0)
    keep, rois, n_kept, status = ops.nms_sorted(boxes.to(dev), torch.tensor([10], dtype=torch.int32, device=dev), 0.7, 300)
    assert status.item() == 1                     # the reference raises IndexError here (Q4)


def test_bbox_iou(ops, dev):
    g = torch.Generator().manual_seed(16)
    a, b = _rand_boxes(g, 333), _rand_boxes(g, 1201)
    got = ops.bbox_iou(a.to(dev), b.to(dev)).cpu()
    assert torch.equal(got, oracle.bbox_iou(a, b))              # same f32 expression, correctly rounded divide
    d1 = torch.tensor([[100., 100, 200, 200]], device=dev)
    d2 = torch.tensor([[150., 150, 250, 250]], device=dev)
    assert abs(ops.bbox_iou(d1, d2).item() - 2500 / 17500) < 1e-7   # reference known answer
    with pytest.raises(IndexError):
        ops.bbox_iou(torch.zeros(2, 3, device=dev), torch.zeros(2, 4, device=dev))


# ----------------------------------------------------------------------------- RoI head
@pytest.mark.parametrize("C,Hf,Wf", [(512, 50, 84), (2048, 25, 42), (8, 6, 6)])
def test_roi_pool_exact(ops, dev, C, Hf, Wf):
    g = torch.Generator().manual_seed(17)
    B, K = 2, 64
    feat = torch.randn(B, C, Hf, Wf, generator=g)
    r = _rand_boxes(g, K, span=float(Wf), wh=float(Wf) * 0.5) - 2.0
    r[:8] = (r[:8] * 2).round() / 2                          # .5 coordinates: round-half-away cases
    r[8] = torch.tensor([-30., -30, -20, -20])               # outside -> empty bins
    rois5 = torch.cat([torch.randint(0, B, (K, 1), generator=g).float(), r], dim=1)
    ref = oracle.roi_pool(feat, rois5, (7, 7), 1.0)
    got = ops.roi_pool_nhwc(ops.nchw_to_nhwc(feat.to(dev)), rois5.to(dev), (7, 7), 1.0).cpu()
    assert torch.equal(got, ref)


def test_roi_pool_avg_fused(ops, dev):
    g = torch.Generator().manual_seed(18)
    B, R, C, Hf, Wf, H, W = 2, 300, 2048, 25, 42, 800, 1333
    feat = torch.randn(B, C, Hf, Wf, generator=g)
    rois = torch.stack([_rand_boxes(g, R, span=900.0, wh=400.0) for _ in range(B)])
    idx = torch.tensor([1, 0], dtype=torch.int32)            # permuted on purpose
    flat = rois.view(-1, 4)
    fm = torch.zeros_like(flat)
    fm[:, [0, 2]] = flat[:, [0, 2]] / W * Wf
    fm[:, [1, 3]] = flat[:, [1, 3]] / H * Hf
    rois5 = torch.cat([idx.float().repeat_interleave(R).view(-1, 1), fm], dim=1)
    ref = F.adaptive_avg_pool2d(oracle.roi_pool(feat, rois5, (7, 7), 1.0), 1).flatten(1)
    got = ops.roi_pool_avg_nhwc(ops.nchw_to_nhwc(feat.to(dev)), rois.to(dev), idx.to(dev), H, W).cpu()
    assert (got - ref).abs().max().item() < 1e-5


def test_detections(ops, dev):
    g = torch.Generator().manual_seed(19)
    B, R, n_class = 2, 300, 81
    scores = torch.randn(B, R, n_class, generator=g)
    scores[0, 0, 5] = scores[0, 0, 70] = 9.0                  # tie: first maximum wins
    locs = torch.randn(B, R, 4 * n_class, generator=g) * 0.2
    rois = torch.stack([_rand_boxes(g, R) for _ in range(B)])
    ref = oracle.detections_from_outputs(locs, scores, rois)
    got = ops.detections(locs.to(dev), scores.to(dev), rois.to(dev)).cpu()
    assert torch.equal(got[..., 5], ref[..., 5])              # class indices bit-exact
    assert torch.equal(got[..., 4], ref[..., 4])
    assert (got[..., :4] - ref[..., :4]).abs().max().item() < 1e-3
    assert got[0, 0, 5].item() == 5.0


# ----------------------------------------------------------------------------- stand-alone utils surface vs the reference's vectors
def test_utils_loc2bbox_matches_the_references_vectors(dev, golden_dir):
    """utils.loc_bbox_iou.loc2bbox (reference :29-61) through tsod_loc2bbox_f32, both call shapes: [n,4] and the
    [n,4k] form of the strided slices (:42-45), against outputs of the reference's own function (make_golden.py)."""
    import os
    from two_stage_object_detection_amd.utils.loc_bbox_iou import loc2bbox, bbox_iou
    z = np.load(os.path.join(golden_dir, "boxmath.npz"))
    got = loc2bbox(torch.from_numpy(z["src"]).to(dev), torch.from_numpy(z["loc"]).to(dev)).cpu().numpy()
    assert got.shape == z["loc2bbox"].shape
    assert np.abs(got - z["loc2bbox"]).max() <= 1e-3                      # the north star's box bar; in practice ~1 ulp (exp)
    assert np.abs(got - z["loc2bbox"]).max() <= 2e-4
    zk = np.load(os.path.join(golden_dir, "boxmath_k.npz"))
    gk = loc2bbox(torch.from_numpy(zk["src"]).to(dev), torch.from_numpy(zk["loc_k3"]).to(dev)).cpu().numpy()
    assert gk.shape == (40, 12)
    assert np.abs(gk - zk["loc2bbox_k3"]).max() <= 2e-4
    empty = loc2bbox(torch.zeros(0, 4, device=dev), torch.zeros(0, 4, device=dev))
    assert tuple(empty.shape) == (0, 4)
    iou = bbox_iou(torch.from_numpy(z["iou_a"]).to(dev), torch.from_numpy(z["iou_b"]).to(dev)).cpu().numpy()
    assert np.abs(iou - z["iou"]).max() <= 1e-7
    d1 = torch.tensor([[100., 100, 200, 200]], device=dev)
    d2 = torch.tensor([[150., 150, 250, 250]], device=dev)
    assert np.allclose(bbox_iou(d1, d2).cpu().numpy(), z["known_iou"], atol=1e-7)


def test_utils_anchor_functions_match_the_references_vectors(dev, golden_dir):
    """utils.basic_anchors.generate_basic_anchor / enumerate_shifted_anchor (reference :11-23, :27-57) through
    tsod_enumerate_anchors_f32 against the reference's own outputs: exact f32 adds -> bit-exact."""
    import os
    from two_stage_object_detection_amd.utils.basic_anchors import enumerate_shifted_anchor, generate_basic_anchor
    z = np.load(os.path.join(golden_dir, "anchors.npz"))
    base = generate_basic_anchor()
    assert np.array_equal(base.cpu().numpy(), z["base"])
    alt = generate_basic_anchor(base_size=16, ratios=[0.5, 1, 2, 3], anchor_scales=[4, 8])
    assert np.array_equal(alt.cpu().numpy(), z["base_alt"])
    base = base.to(dev)
    assert np.array_equal(enumerate_shifted_anchor(base, 16, 3, 5).cpu().numpy(), z["shifted_s16_h3_w5"])
    assert np.array_equal(enumerate_shifted_anchor(base, 32, 2, 3).cpu().numpy(), z["shifted_s32_h2_w3"])
    full = enumerate_shifted_anchor(base, 16, 50, 84).cpu()
    assert full.shape == (37800, 4)
    assert np.array_equal(full[[0, 9, 755, 756, 37799]].numpy(), z["shifted_s16_h50_w84_rows"])
    assert torch.equal(full, oracle.enumerate_shifted_anchor(base.cpu(), 16, 50, 84))


# ----------------------------------------------------------------------------- RoIAlign (added roi_op="align"; parity unpinned)
@pytest.mark.parametrize("C,Hf,Wf,sampling_ratio,aligned", [(512, 50, 84, 2, False), (2048, 25, 42, 0, False), (8, 6, 6, 2, True),
                                                          (64, 13, 17, 3, True)])
def test_roi_align_matches_oracle(ops, dev, C, Hf, Wf, sampling_ratio, aligned):
    g = torch.Generator().manual_seed(23)
    B, K = 2, 48
    feat = torch.randn(B, C, Hf, Wf, generator=g)
    r = _rand_boxes(g, K, span=float(Wf), wh=float(Wf) * 0.5) - 2.0
    r[0] = torch.tensor([-30., -30, -20, -20])                  # entirely outside: samples nothing
    r[1] = torch.tensor([0.2, 0.3, 0.25, 0.31])                 # tiny: extents floored at 1 unless aligned
    rois5 = torch.cat([torch.randint(0, B, (K, 1), generator=g).float(), r], dim=1)
    ref = oracle.roi_align(feat, rois5, (7, 7), 0.75, sampling_ratio, aligned)
    got = ops.roi_align_nhwc(ops.nchw_to_nhwc(feat.to(dev)), rois5.to(dev), (7, 7), 0.75, sampling_ratio, aligned).cpu()
    assert (got - ref).abs().max().item() <= 1e-5               # same f32 expression order; in practice bit-equal or 1 ulp
    assert (got[0] == 0).all()


def test_roi_align_avg_fused(ops, dev):
    g = torch.Generator().manual_seed(24)
    B, R, C, Hf, Wf, H, W = 2, 300, 2048, 25, 42, 800, 1333
    feat = torch.randn(B, C, Hf, Wf, generator=g)
    rois = torch.stack([_rand_boxes(g, R, span=900.0, wh=400.0) for _ in range(B)])
    idx = torch.tensor([1, 0], dtype=torch.int32)
    flat = rois.view(-1, 4)
    fm = torch.zeros_like(flat)
    fm[:, [0, 2]] = flat[:, [0, 2]] / W * Wf
    fm[:, [1, 3]] = flat[:, [1, 3]] / H * Hf
    rois5 = torch.cat([idx.float().repeat_interleave(R).view(-1, 1), fm], dim=1)
    ref = F.adaptive_avg_pool2d(oracle.roi_align(feat, rois5, (7, 7), 1.0, 2, False), 1).flatten(1)
    got = ops.roi_align_avg_nhwc(ops.nchw_to_nhwc(feat.to(dev)), rois.to(dev), idx.to(dev), H, W).cpu()
    assert (got - ref).abs().max().item() < 1e-5


@pytest.mark.parametrize("C,groups,stride", [(128, 32, 1), (256, 32, 2), (64, 4, 1), (1024, 32, 1)])
def test_grouped_conv3x3(ops, dev, C, groups, stride):
    """ResNeXt's conv2 (models/resnet.py:46-47): grouped 3x3 + folded BN + PReLU against torch's CPU grouped conv."""
    from two_stage_object_detection_amd._ffi import ACT_PRELU
    g = torch.Generator().manual_seed(30)
    x = torch.randn(2, C, 13, 17, generator=g)
    w = torch.randn(C, C // groups, 3, 3, generator=g) / math.sqrt(9 * C // groups)
    scale, shift = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1
    ref = F.prelu(F.conv2d(x, w, None, stride, 1, 1, groups) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1), torch.tensor([0.2]))
    y = ops.gconv3x3_nhwc(ops.nchw_to_nhwc(x.to(dev)), w.permute(0, 2, 3, 1).contiguous().to(dev), groups, scale.to(dev),
                          shift.to(dev), stride, ACT_PRELU, 0.2)
    assert (ops.nhwc_to_nchw(y).cpu() - ref).abs().max().item() < 2e-5


@pytest.mark.parametrize("C1,C2,Cout,stride2,prec", [(64, 64, 256, 1, 0), (128, 256, 512, 2, 0), (256, 512, 1024, 2, 1), (512, 1024, 2048, 2, 1),
                                                    (64, 64, 256, 1, 1)])
def test_conv_dual_source_is_the_sum_of_two_convs(ops, dev, C1, C2, Cout, stride2, prec):
    """tsod_conv2d_dual_f32: a 1x1 conv over y plus a strided 1x1 tap of x as ONE stacked-K GEMM (a bottleneck's conv3 +
    projection shortcut, models/resnet.py:70-76 + :114-116) against the two f64 CPU convolutions added up; every tile that
    can hold it, with and without K-slices cutting through the seam between the sources."""
    from two_stage_object_detection_amd._ffi import BF16X3_TILE_IDS, TILE_IDS, ACT_PRELU, TsodError
    g = torch.Generator().manual_seed(33)
    H2, W2 = 14, 18
    OH, OW = (H2 - 1) // stride2 + 1, (W2 - 1) // stride2 + 1
    y = torch.randn(2, C1, OH, OW, generator=g)
    x = torch.randn(2, C2, H2, W2, generator=g)
    w3 = torch.randn(Cout, C1, 1, 1, generator=g) / math.sqrt(C1)
    wd = torch.randn(Cout, C2, 1, 1, generator=g) / math.sqrt(C2)
    shift = torch.randn(Cout, generator=g) * 0.1
    ref = F.prelu((F.conv2d(y.double(), w3.double()) + F.conv2d(x.double(), wd.double(), None, stride2)
                   + shift.double().view(1, -1, 1, 1)).float(), torch.tensor([0.25]))
    yn, xn = ops.nchw_to_nhwc(y.to(dev)), ops.nchw_to_nhwc(x.to(dev))
    w = torch.cat([w3.flatten(1), wd.flatten(1)], dim=1).contiguous().to(dev)
    tol = 3e-6 * math.sqrt(C1 + C2) + 1e-5
    ran = 0
    from two_stage_object_detection_amd._ffi import DMA_TILE_IDS
    for tile in [0] + list(BF16X3_TILE_IDS if prec else TILE_IDS):
        for split in (0, 1, -1, 2, 3) + ((-2,) if prec and tile in DMA_TILE_IDS else ()):
            out = ops.conv2d_nhwc(yn, w, segs=[(0, C1)], shift=shift.to(dev), act=ACT_PRELU, slope=0.25, tile=tile, split_k=split,
                                  precision=prec, x2=xn, stride2=stride2)
            ran += 1
            assert (ops.nhwc_to_nchw(out).cpu() - ref).abs().max().item() <= tol, (tile, split)
    assert ran >= 30
    with pytest.raises(TsodError):                        # a second source through the single-source entry point is refused
        from ctypes import byref
        from two_stage_object_detection_amd import _ffi
        d = _ffi.make_conv_desc(N=2, H=OH, W=OW, in_pitch=C1, segs=[(0, C1)], Cout=Cout, out_pitch=Cout, src2=(C2, C2, 0, stride2, H2, W2))
        _ffi.check(_ffi.lib().tsod_conv2d_f32(byref(d), _ffi.ptr(yn), _ffi.ptr(w), None, None, None, _ffi.ptr(out), None, 0, None))


@pytest.mark.parametrize("N,Cin,Cout,H,W,k,res", [(1, 256, 256, 50, 84, 3, False), (2, 1024, 256, 25, 21, 1, False), (1, 512, 2048, 13, 21, 1, True),
                                                  (1, 64, 64, 37, 41, 3, True), (1, 128, 96, 19, 23, 3, False)])
def test_conv_fp16x2_matches_the_f64_convolution(ops, dev, N, Cin, Cout, H, W, k, res):
    """TSOD_PREC_FP16X2 (tile d128x128k32 and the register-staged family): two fp16 pieces of 16 x per operand, three piece products per f32
    product.  Held to the bf16x3 / f32 bar against the f64 CPU convolution under every K schedule, and to <= 2x the error of the
    bf16x3 form of the same launch (measured: below it)."""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, Cin, H, W, generator=g)
    x = torch.maximum(x, 0.25 * x)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(k * k * Cin)
    r = torch.randn(N, Cout, H, W, generator=g) if res else None
    ref = F.conv2d(x.double(), w.double(), padding=k // 2)
    if res:
        ref = ref + r.double()
    ref = torch.where(ref >= 0, ref, 0.25 * ref).float()
    xn, wp = ops.nchw_to_nhwc(x.to(dev)), ops.pack_conv_weight(w.to(dev))
    rn = ops.nchw_to_nhwc(r.to(dev)) if res else None
    tol = 3e-6 * math.sqrt(k * k * Cin) + 1e-5
    from two_stage_object_detection_amd._ffi import DMA_TILE_IDS, FP16X2_TILE_IDS
    e3 = (ops.nhwc_to_nchw(ops.conv2d_nhwc(xn, wp, pad=k // 2, tile=22, split_k=1, precision=1, residual=rn, act=1, slope=0.25)).cpu() - ref).abs().max().item()
    for tile in FP16X2_TILE_IDS:                                 # the LDS-DMA tile and the register-staged family
        for split in (1, 3, -1) + ((-2,) if tile in DMA_TILE_IDS else ()):
            out = ops.conv2d_nhwc(xn, wp, pad=k // 2, tile=tile, split_k=split, precision=2, residual=rn, act=1, slope=0.25, a_scale_exp=4)
            e = (ops.nhwc_to_nchw(out).cpu() - ref).abs().max().item()
            assert e <= tol and e <= 2 * e3 + 1e-7, (tile, split, e, e3, tol)


def test_conv_fp16x2_second_source_range_and_gates(ops, dev):
    """The rest of the fp16x2 contract: the stacked-K form with a second source (a bottleneck's conv3 + projection shortcut)
    matches the two f64 convolutions added up; the register-staged bf16x3 tiles and d128x128k32 take the arithmetic (others:
    TSOD_ERR_UNSUPPORTED, TSOD_TILE_AUTO resolves among them), concatenated channel segments included; what happens beyond the range (|x| >= 65504 / 2^a_scale_exp) is pinned down as it IS: the
    pieces overflow to +-inf, their products cancel to NaN, and the branch-free activation of the epilogue maps NaN to 0 - the
    outputs the value feeds come out 0, everything else is untouched - and the launch raises desc.range_flag (every workgroup
    checks its accumulators behind the K loop: a finite tile proves its inputs were in range), under every K schedule; a
    smaller exponent brings the value back into range and leaves the flag alone."""
    from two_stage_object_detection_amd._ffi import TsodError
    g = torch.Generator().manual_seed(36)
    C1, C2, Cout, H2, W2 = 64, 96, 128, 11, 13
    y = torch.randn(2, C1, H2, W2, generator=g)
    x = torch.randn(2, C2, H2, W2, generator=g)
    w3 = torch.randn(Cout, C1, 1, 1, generator=g) / math.sqrt(C1)
    wd = torch.randn(Cout, C2, 1, 1, generator=g) / math.sqrt(C2)
    ref = (F.conv2d(y.double(), w3.double()) + F.conv2d(x.double(), wd.double())).float()
    yn, xn = ops.nchw_to_nhwc(y.to(dev)), ops.nchw_to_nhwc(x.to(dev))
    w = torch.cat([w3.flatten(1), wd.flatten(1)], dim=1).contiguous().to(dev)
    tol = 3e-6 * math.sqrt(C1 + C2) + 1e-5
    for split in (1, -1, 2):
        out = ops.conv2d_nhwc(yn, w, segs=[(0, C1)], tile=22, split_k=split, precision=2, x2=xn)
        assert (ops.nhwc_to_nchw(out).cpu() - ref).abs().max().item() <= tol, split
    out = ops.conv2d_nhwc(yn, w, segs=[(0, C1)], precision=2, x2=xn)                  # AUTO
    assert (ops.nhwc_to_nchw(out).cpu() - ref).abs().max().item() <= tol
    for tile in (1, 2, 12, 18, 20):                                                # f32-only tiles and the 64-row LDS-DMA shapes
        with pytest.raises(TsodError, match="unsupported|UNSUPPORTED"):
            ops.conv2d_nhwc(yn, w, segs=[(0, C1)], tile=tile, split_k=1, precision=2, x2=xn)
    for tile in (3, 14, 16):                                                       # the register-staged family takes the second source too
        out = ops.conv2d_nhwc(yn, w, segs=[(0, C1)], tile=tile, split_k=1, precision=2, x2=xn)
        assert (ops.nhwc_to_nchw(out).cpu() - ref).abs().max().item() <= tol, tile
    # concatenated channel segments and a K that is no multiple of the K-step (HarDNet's layers): register-staged tiles only
    g2 = torch.Generator().manual_seed(37)
    xs = torch.randn(2, 60, 9, 11, generator=g2)
    ws = torch.randn(20, 44, 1, 1, generator=g2) / math.sqrt(44)
    segs = [(36, 24), (4, 20)]
    refs = F.conv2d(torch.cat([xs[:, 36:60], xs[:, 4:24]], dim=1).double(), ws.double()).float()
    xsn, wsp = ops.nchw_to_nhwc(xs.to(dev)), ops.pack_conv_weight(ws.to(dev))
    for tile in (3, 8, 14):
        out = ops.conv2d_nhwc(xsn, wsp, segs=segs, tile=tile, split_k=1, precision=2)
        assert (ops.nhwc_to_nchw(out).cpu() - refs).abs().max().item() <= 3e-6 * math.sqrt(44) + 1e-5, tile
    # range: one activation of 5000 with a_scale_exp = 4 (16 * 5000 > 65504)
    xb = torch.randn(1, 64, 9, 9, generator=g)
    xb[0, 3, 4, 4] = 5000.0
    wb = torch.randn(32, 64, 3, 3, generator=g) / 24.0
    refb = F.conv2d(xb.double(), wb.double(), padding=1).float()
    xbn, wbp = ops.nchw_to_nhwc(xb.to(dev)), ops.pack_conv_weight(wb.to(dev))
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    bad = ops.nhwc_to_nchw(ops.conv2d_nhwc(xbn, wbp, pad=1, tile=22, split_k=1, precision=2, a_scale_exp=4, range_flag=flag)).cpu()
    assert int(flag.item()) == 1                                                       # ... and the launch SAYS so
    touched = torch.zeros(1, 1, 9, 9, dtype=torch.bool)
    touched[0, 0, 3:6, 3:6] = True
    hit = bad[touched.expand_as(bad)]
    assert ((hit == 0) | ~torch.isfinite(hit)).all()                                  # every output the value feeds: 0 (from NaN) or inf
    assert (bad[~touched.expand_as(bad)] - refb[~touched.expand_as(bad)]).abs().max().item() <= 1e-4
    flag.zero_()
    good = ops.nhwc_to_nchw(ops.conv2d_nhwc(xbn, wbp, pad=1, tile=22, split_k=1, precision=2, a_scale_exp=2, range_flag=flag)).cpu()
    assert int(flag.item()) == 0
    for split in (3, -1, -2):                                                          # K-slices report from their partial sums
        flag.zero_()
        ops.conv2d_nhwc(xbn, wbp, pad=1, tile=22, split_k=split, precision=2, a_scale_exp=4, range_flag=flag)
        assert int(flag.item()) == 1, split
    assert (good - refb).abs().max().item() <= 2e-3 * 1.0                            # (values ~100: 5000 * w; f32-level relative error)


def test_conv_dma_tiles_stage_table_limits_and_step_order(ops, dev):
    """The LDS-DMA tiles keep one table entry per K-step of a workgroup's K range (640 entries): a K that needs more steps than
    that at a tile's stage size is REFUSED for that tile when named and never picked by TSOD_TILE_AUTO, while the tiles with
    longer stages still take it.  Same layer: a 3x3 filter on these tiles runs its K-steps in (channel block, tap) order -
    whole tiles, uniform K-slices (slices = channel-block ranges) and the balanced ranges must all match the f64 convolution."""
    from two_stage_object_detection_amd._ffi import TsodError
    g = torch.Generator().manual_seed(77)
    Cin, Cout, H, W = 1280, 128, 9, 11                                       # K = 11520: 720 steps of 16, 360 of 32
    x = torch.randn(1, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(9 * Cin)
    ref = F.conv2d(x.double(), w.double(), padding=1).float()
    xn = ops.nchw_to_nhwc(x.to(dev))
    wp = ops.pack_conv_weight(w.to(dev))
    tol = 3e-6 * math.sqrt(9 * Cin) + 1e-5
    for tile in (17, 19, 21):                                                # 16-float stages: 720 + 8 > 640
        with pytest.raises(TsodError, match="unsupported|UNSUPPORTED"):
            ops.conv2d_nhwc(xn, wp, pad=1, tile=tile, split_k=1, precision=1)
    for tile in (18, 20, 22):                                                # 32-float stages: 360 + 8 entries
        for split in (1, 3, -1, -2):
            out = ops.conv2d_nhwc(xn, wp, pad=1, tile=tile, split_k=split, precision=1)
            assert (ops.nhwc_to_nchw(out).cpu() - ref).abs().max().item() <= tol, (tile, split)
    out = ops.conv2d_nhwc(xn, wp, pad=1, precision=1)                        # AUTO resolves to something that runs
    assert (ops.nhwc_to_nchw(out).cpu() - ref).abs().max().item() <= tol


@pytest.mark.parametrize("C1,C2", [(64, 48), (64, 32), (32, 96)])
def test_conv_dual_source_second_source_must_be_whole_ksteps(ops, dev, C1, C2):
    """The stacked-K GEMM's K-steps must not run past the second source's c2 channels (the uniform-tap loader has no k < K
    mask): a tile whose K-step does not divide c2 is REFUSED when named (TSOD_ERR_UNSUPPORTED) and never picked by
    TSOD_TILE_AUTO; every tile that does divide it matches the two f64 CPU convolutions added up (c2 = 48 only fits the
    16-float stages of the LDS-DMA tiles; c2 = 32 fits the 32-float tiles but not the _K64 ones)."""
    from two_stage_object_detection_amd._ffi import BF16X3_TILE_IDS, TILE_IDS, DMA_TILE_IDS, TsodError
    g = torch.Generator().manual_seed(35)
    H2, W2, Cout = 11, 13, 96
    y = torch.randn(2, C1, H2, W2, generator=g)
    x = torch.randn(2, C2, H2, W2, generator=g)
    w3 = torch.randn(Cout, C1, 1, 1, generator=g) / math.sqrt(C1)
    wd = torch.randn(Cout, C2, 1, 1, generator=g) / math.sqrt(C2)
    ref = (F.conv2d(y.double(), w3.double()) + F.conv2d(x.double(), wd.double())).float()
    yn, xn = ops.nchw_to_nhwc(y.to(dev)), ops.nchw_to_nhwc(x.to(dev))
    w = torch.cat([w3.flatten(1), wd.flatten(1)], dim=1).contiguous().to(dev)
    tol = 3e-6 * math.sqrt(C1 + C2) + 1e-5
    bk_of = {10: 64, 11: 64, 17: 16, 19: 16, 21: 16}                       # K-step of a tile (floats); 32 for the others
    ran = refused = 0
    for prec, tiles in ((0, TILE_IDS), (1, BF16X3_TILE_IDS)):
        for tile in tiles:
            bk = bk_of.get(tile, 32)
            fits = C1 % bk == 0 and C2 % bk == 0
            for split in (1, -1, 2):
                if fits:
                    out = ops.conv2d_nhwc(yn, w, segs=[(0, C1)], tile=tile, split_k=split, precision=prec, x2=xn)
                    assert (ops.nhwc_to_nchw(out).cpu() - ref).abs().max().item() <= tol, (tile, split, prec)
                    ran += 1
                else:
                    with pytest.raises(TsodError, match="unsupported|UNSUPPORTED"):
                        ops.conv2d_nhwc(yn, w, segs=[(0, C1)], tile=tile, split_k=split, precision=prec, x2=xn)
                    refused += 1
        any_fits = any(C1 % bk_of.get(t, 32) == 0 and C2 % bk_of.get(t, 32) == 0 for t in tiles)
        if any_fits:                                                         # AUTO only considers tiles that fit
            out = ops.conv2d_nhwc(yn, w, segs=[(0, C1)], precision=prec, x2=xn)
            assert (ops.nhwc_to_nchw(out).cpu() - ref).abs().max().item() <= tol, ("auto", prec)
        else:
            with pytest.raises(TsodError, match="unsupported|UNSUPPORTED"):
                ops.conv2d_nhwc(yn, w, segs=[(0, C1)], precision=prec, x2=xn)
    assert ran > 0 and refused > 0


def test_conv_xcd_run_orders_agree(dev):
    """The order in which a launch's work items are dealt to the XCDs (runs that share activation rows / runs that share an
    output-channel tile and K-slice: conv_igemm_f32.hip work_item) is a placement choice only: with the order pinned either
    way (TSOD_XCD_NMAJOR, read once per process - hence child processes) every tile / K-slice schedule must produce the
    same bits, and match the f64 CPU convolution."""
    import os
    import subprocess
    import sys
    code = r"""
import math, sys, torch
import torch.nn.functional as F
from two_stage_object_detection_amd import hip_ops as ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(5)
outs = []
for (H, W, Cin, Cout, k, prec, tiles) in ((25, 42, 512, 512, 3, 1, (22, 18, 17, 8)), (13, 21, 1024, 256, 1, 1, (22, 18, 3)), (25, 42, 256, 512, 1, 0, (3, 1, 6))):
    x = torch.randn(1, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    ref = F.conv2d(x.double(), w.double(), None, 1, k // 2).float()
    xn, wp = ops.nchw_to_nhwc(x.to(dev)), ops.pack_conv_weight(w.to(dev))
    for tile in tiles:
        for split in (1, 2, 3, 4):
            y = ops.conv2d_nhwc(xn, wp, pad=k // 2, tile=tile, split_k=split, precision=prec)
            assert (ops.nhwc_to_nchw(y).cpu() - ref).abs().max().item() <= 3e-6 * math.sqrt(Cin * k * k) + 1e-5, (tile, split)
            outs.append(y.cpu())
torch.save(outs, sys.argv[1])
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for mode in ("0", "1"):
        path = os.path.join("/tmp", f"tsod_xcd_{os.getpid()}_{mode}.pt")
        env = dict(os.environ, TSOD_XCD_NMAJOR=mode, PYTHONPATH=root)
        r = subprocess.run([sys.executable, "-c", code, path], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        res.append(torch.load(path))
        os.remove(path)
    assert len(res[0]) == len(res[1]) >= 36
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("shape", [(8, 50, 84, 256, 256, 3), (8, 25, 42, 2048, 512, 1), (1, 100, 167, 128, 128, 3)])
def test_dma_conv_tiles_at_full_layer_sizes(ops, dev, shape):
    """The LDS-DMA tiles (conv_dma_kernel) at layer sizes of BASELINE configs[1] / [4] (too large for an f64 CPU reference in
    a unit test), through properties that do not depend on size: (1) every tile and K schedule, balanced ranges included,
    agrees with the f32-MFMA kernel of the same layer at the accumulation-noise bar; (2) linearity in a power of two is
    EXACT - the three bf16 pieces of 4x are 4 times the pieces of x, every product and partial sum scales exactly - so
    conv(4x) must equal 4 conv(x) bit for bit, whatever the schedule; (3) a zero input gives exact zeros (the null / padding
    DMAs write zeros, never stale LDS)."""
    from two_stage_object_detection_amd._ffi import DMA_TILE_IDS
    B, H, W, Cin, Cout, k = shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, H, W, Cin, generator=g).to(dev)
    w = ops.pack_conv_weight((torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)).to(dev))
    ref = ops.conv2d_nhwc(x, w, pad=k // 2, tile=3, split_k=1)                       # f32 MFMA, whole tiles
    tol = 3e-6 * math.sqrt(Cin * k * k) * 2 + 2e-5
    for tile in DMA_TILE_IDS:
        prec = 2 if tile in (23, 24) else 1              # (d192x128, d64x128k64 exist in fp16x2 only)
        for split in (1, -1, -2, 3):
            y = ops.conv2d_nhwc(x, w, pad=k // 2, tile=tile, split_k=split, precision=prec)
            assert (y - ref).abs().max().item() <= tol, (tile, split)
            y4 = ops.conv2d_nhwc(x * 4.0, w, pad=k // 2, tile=tile, split_k=split, precision=prec)
            if prec == 1:
                assert torch.equal(y4, y * 4.0), (tile, split, "linearity in 4 must be exact")
            else:       # fp16x2 at a static exponent: the low pieces of small elements sit in fp16's subnormals, where 4 x rounds differently
                assert (y4 - y * 4.0).abs().max().item() <= 4 * tol, (tile, split)
    z = ops.conv2d_nhwc(torch.zeros_like(x), w, pad=k // 2, tile=DMA_TILE_IDS[0], split_k=-2, precision=1)
    assert torch.count_nonzero(z).item() == 0


def test_range_words_producers_hold_the_abs_max(ops, dev):
    """Range words (include/tsod.h): every producing kernel adds the abs-max of what it STORES to its tensor's 64 words - both conv
    kernel families (whole tiles, K-slices' last arrivers, the balanced schedule, vector and scalar epilogues), the layout
    kernels, the depthwise / grouped kernels and tsod_absmax_f32.  Bit-exact: the largest word IS the output's abs-max."""
    import ctypes
    from two_stage_object_detection_amd import _ffi
    g = torch.Generator().manual_seed(91)
    x = torch.randn(2, 64, 23, 29, generator=g)
    x[1, 5, 7, 11] = -37.5
    words = ops.new_amax_words(dev, 8)
    xn = ops.nchw_to_nhwc(x.to(dev))
    L, s = _ffi.lib(), _ffi.stream_ptr
    # layout (small-C and tiled) + absmax
    x3 = torch.randn(2, 3, 17, 19, generator=g) * 3
    o4 = torch.empty(2, 17, 19, 4, device=dev)
    _ffi.check(L.tsod_nchw_to_nhwc_amax_f32(_ffi.ptr(x3.to(dev)), 2, 3, 17, 19, _ffi.ptr(o4), 4, 4, words[0].data_ptr(), s()))
    assert ops.amax_value(words[0]) == float(x3.abs().max())
    o64 = torch.empty(2, 23, 29, 64, device=dev)
    _ffi.check(L.tsod_nchw_to_nhwc_amax_f32(_ffi.ptr(x.to(dev)), 2, 64, 23, 29, _ffi.ptr(o64), 64, 64, words[1].data_ptr(), s()))
    assert ops.amax_value(words[1]) == 37.5 and torch.equal(o64, xn)
    ops.absmax(xn, words[2])
    assert ops.amax_value(words[2]) == 37.5
    odd = torch.randn(1001, generator=g).to(dev)                      # (no float4 path)
    ops.absmax(odd, words[3])
    assert ops.amax_value(words[3]) == float(odd.abs().max())
    # conv: every arithmetic / kernel family / K schedule; the words accumulate (max) over launches, so one slot per launch
    w = torch.randn(96, 64, 3, 3, generator=g) / 24.0
    wp = ops.pack_conv_weight(w.to(dev))
    res = ops.nchw_to_nhwc(torch.randn(2, 96, 23, 29, generator=g).to(dev))
    for prec, tile, split in ((0, 3, 1), (0, 1, 3), (0, 8, -1), (1, 14, 1), (1, 22, 3), (1, 19, -2), (2, 22, 1), (2, 17, -2), (2, 8, 2), (2, 15, -1)):
        slot = ops.new_amax_words(dev, 1)
        out = ops.conv2d_nhwc(xn, wp, pad=1, residual=res, act=1, slope=0.25, tile=tile, split_k=split, precision=prec, amax_out=slot)
        assert ops.amax_value(slot) == float(out.abs().max()), (prec, tile, split)
    # scalar epilogue (Cout not a multiple of 4) and a channel slice of a wider output
    w54 = torch.randn(54, 64, 1, 1, generator=g) / 8.0
    slot = ops.new_amax_words(dev, 1)
    out = ops.conv2d_nhwc(xn, ops.pack_conv_weight(w54.to(dev)), amax_out=slot)
    assert ops.amax_value(slot) == float(out.abs().max())
    # depthwise, pair conv, grouped conv
    wd = torch.randn(3, 3, 64, generator=g).to(dev)
    od = torch.zeros(2, 23, 29, 64, device=dev)
    _ffi.check(L.tsod_dwconv3x3_amax_f32(_ffi.ptr(xn), 2, 23, 29, 64, 64, 0, _ffi.ptr(wd), None, None, 1, 0, _ffi.ptr(od), 64, 0, words[4].data_ptr(), s()))
    assert ops.amax_value(words[4]) == float(od.abs().max()) > 0
    wpair = torch.randn(32, 2, generator=g).to(dev)
    op = torch.zeros(2 * 23 * 29, 32, device=dev)
    _ffi.check(L.tsod_gconv1x1_pair_amax_f32(_ffi.ptr(xn), 2 * 23 * 29, 32, 64, _ffi.ptr(wpair), None, _ffi.ptr(op), 32, words[5].data_ptr(), s()))
    assert ops.amax_value(words[5]) == float(op.abs().max()) > 0
    wg = torch.randn(64, 3, 3, 8, generator=g).to(dev)
    og = torch.zeros(2, 23, 29, 64, device=dev)
    _ffi.check(L.tsod_gconv3x3_amax_f32(_ffi.ptr(xn), 2, 23, 29, 64, 64, 8, _ffi.ptr(wg), None, None, 1, 1, ctypes.c_float(0.25), _ffi.ptr(og), 64,
                                        words[6].data_ptr(), s()))
    assert ops.amax_value(words[6]) == float(og.abs().max()) > 0
    # reset
    _ffi.check(L.tsod_amax_reset(words.data_ptr(), 8, s()))
    assert int(words.abs().max()) == 0
    # a misaligned word pointer is refused
    with pytest.raises(_ffi.TsodError):
        ops.conv2d_nhwc(xn, wp, pad=1, amax_out=words[0].data_ptr() + 4)


@pytest.mark.parametrize("scale", [1.0, 300.0, 1e-3, 3e4, 2e-7])
def test_conv_fp16x2_scale_follows_the_tensor(ops, dev, scale):
    """fp16x2 with range words: the activation exponent comes from the abs-max the input's producer left - per launch, in the
    kernel - so the SAME f32 accuracy holds whatever the tensor's range (here 2e-7 ... 3e4 times a PReLU-shaped normal tensor:
    far outside the static 2^4 exponent's |x| < 4094 on one side and deep in fp16's subnormals on the other), in both kernel
    families, under K-slices and the balanced schedule, and for the second source (the larger of the two tensors decides).  The
    range flag stays down; it fires for non-finite input only."""
    g = torch.Generator().manual_seed(77)
    N, Cin, Cout, H, W, k = 1, 128, 128, 27, 31, 3
    x = torch.randn(N, Cin, H, W, generator=g)
    x = torch.maximum(x, 0.25 * x) * scale
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(k * k * Cin)
    ref = F.conv2d(x.double(), w.double(), padding=1).float()
    xn, wp = ops.nchw_to_nhwc(x.to(dev)), ops.pack_conv_weight(w.to(dev))
    words = ops.absmax(xn, ops.new_amax_words(dev, 1))
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    tol = (3e-6 * math.sqrt(k * k * Cin) + 1e-5) * scale
    from two_stage_object_detection_amd._ffi import DMA_TILE_IDS, FP16X2_TILE_IDS
    for tile in FP16X2_TILE_IDS:
        for split in (1, 3) + ((-2,) if tile in DMA_TILE_IDS else ()):
            out = ops.conv2d_nhwc(xn, wp, pad=1, tile=tile, split_k=split, precision=2, amax_in=words, range_flag=flag)
            e = (ops.nhwc_to_nchw(out).cpu() - ref).abs().max().item()
            assert e <= tol, (tile, split, e, tol)
    assert int(flag.item()) == 0
    # second source: the stacked-K GEMM with x2 1000 times larger than x - the exponent must follow the larger tensor
    C2 = 64
    x2 = torch.randn(N, C2, H, W, generator=g) * scale * 1000.0
    wd = torch.randn(Cout, C2, 1, 1, generator=g) / math.sqrt(C2)
    w1 = torch.randn(Cout, Cin, 1, 1, generator=g) / math.sqrt(Cin)
    ref2 = (F.conv2d(x.double(), w1.double()) + F.conv2d(x2.double(), wd.double())).float()
    x2n = ops.nchw_to_nhwc(x2.to(dev))
    words2 = ops.absmax(x2n, ops.new_amax_words(dev, 1))
    wst = torch.cat([w1.flatten(1), wd.flatten(1)], dim=1).contiguous().to(dev)
    for tile in (22, 14, 24):
        out = ops.conv2d_nhwc(xn, wst, segs=[(0, Cin)], tile=tile, split_k=1, precision=2, x2=x2n, amax_in=words, amax_in2=words2, range_flag=flag)
        assert (ops.nhwc_to_nchw(out).cpu() - ref2).abs().max().item() <= 3e-6 * math.sqrt(Cin + C2) * scale * 1000.0 + 1e-5 * scale, tile
    assert int(flag.item()) == 0
    # non-finite input: the words cannot help, the flag must say so
    xbad = xn.clone()
    xbad[0, 3, 4, 5] = float("inf")
    wbad = ops.absmax(xbad, ops.new_amax_words(dev, 1))
    for tile in (22, 8):
        flag.zero_()
        ops.conv2d_nhwc(xbad, wp, pad=1, tile=tile, split_k=1, precision=2, amax_in=wbad, range_flag=flag)
        assert int(flag.item()) == 1, tile


def _bottleneck_reference(x, w1, w2, w3, bn, slope):
    """f64 CPU: out = PReLU(BN3(conv1x1(PReLU(BN2(conv3x3(PReLU(BN1(conv1x1(x)))))))) + x) (models/resnet.py:57-76, identity shortcut)"""
    act = lambda t: torch.where(t >= 0, t, slope * t)      # noqa: E731
    s1, b1, s2, b2, s3, b3 = (v.double().view(1, -1, 1, 1) for v in bn)
    y = act(F.conv2d(x.double(), w1.double()) * s1 + b1)
    y = act(F.conv2d(y, w2.double(), padding=1) * s2 + b2)
    return act(F.conv2d(y, w3.double()) * s3 + b3 + x.double()).float()


@pytest.mark.parametrize("N,H,W,Cin,Cout,gain,xgain", [(1, 10, 16, 64, 256, 1.0, 1.0), (2, 23, 37, 64, 256, 1.0, 1.0), (1, 31, 20, 128, 192, 1.0, 1.0),
                                                        (1, 12, 50, 64, 256, 500.0, 1.0), (1, 9, 9, 64, 128, 1e-4, 1.0), (1, 17, 21, 64, 256, 1.0, 300.0)])
def test_bottleneck_fused_with_projection_shortcut_matches_the_f64_block(ops, dev, N, H, W, Cin, Cout, gain, xgain):
    """tsod_bottleneck_fp16x2, desc.projection = 1 (layer1's first block, models/resnet.py:114-116): conv3 and the 1x1 projection
    shortcut as one stacked-K GEMM inside the one-launch bottleneck - y2 chunks from LDS, x chunks from L2, the accumulators
    changing scale between them - against the f64 CPU block; tile edges, inputs 500x larger / 10^4 x smaller, a shortcut operand
    300x larger than the main path's (the scale switch), Cin != 64, and the abs-max left for the consumer."""
    g = torch.Generator().manual_seed(321 + H)
    x = torch.randn(N, Cin, H, W, generator=g)
    x = torch.maximum(x, 0.25 * x) * gain
    w1 = torch.randn(64, Cin, 1, 1, generator=g) / math.sqrt(Cin)
    w2 = torch.randn(64, 64, 3, 3, generator=g) / math.sqrt(576)
    w3 = torch.randn(Cout, 64, 1, 1, generator=g) / 8.0
    wd = torch.randn(Cout, Cin, 1, 1, generator=g) / math.sqrt(Cin) * xgain
    bn = [torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1 * gain, torch.rand(64, generator=g) + 0.5,
          torch.randn(64, generator=g) * 0.1 * gain, torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1 * gain]
    sd_, bd_ = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1 * gain
    slope = 0.25
    act = lambda t: torch.where(t >= 0, t, slope * t)      # noqa: E731
    v4 = lambda v: v.double().view(1, -1, 1, 1)             # noqa: E731
    y = act(F.conv2d(x.double(), w1.double()) * v4(bn[0]) + v4(bn[1]))
    y = act(F.conv2d(y, w2.double(), padding=1) * v4(bn[2]) + v4(bn[3]))
    ref = act(F.conv2d(y, w3.double()) * v4(bn[4]) + v4(bn[5]) + F.conv2d(x.double(), wd.double()) * v4(sd_) + v4(bd_)).float()
    xn = ops.nchw_to_nhwc(x.to(dev))
    w2p = ops.pack_conv_weight(w2.to(dev))
    stacked = torch.cat([w3.double().flatten(1) * bn[4].double().view(-1, 1), wd.double().flatten(1) * sd_.double().view(-1, 1)], dim=1).float()
    stream, exps = ops.pack_bottleneck_wstream(w1.view(64, Cin).to(dev), w2p, stacked.to(dev), projection=True)
    bnv = torch.cat(bn[:4] + [torch.ones(Cout), (bn[5].double() + bd_.double()).float()]).to(dev)
    words = ops.absmax(xn, ops.new_amax_words(dev, 1))
    wout = ops.new_amax_words(dev, 1)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    out = ops.bottleneck_fused(xn, stream, exps, bnv, Cout, slope, amax_in=words, amax_out=wout, range_flag=flag, cin=Cin)
    got = ops.nhwc_to_nchw(out).cpu()
    tol = (3e-6 * math.sqrt(576) + 1e-5) * float(ref.abs().max()) / 4.0 + 1e-6 * gain
    err = (got - ref).abs().max().item()
    assert err <= tol, (err, tol)
    assert int(flag.item()) == 0
    assert ops.amax_value(wout) == float(out.abs().max())
    # a non-finite pixel must be reported (PReLU's max / min would turn the NaN into a plausible number)
    xb = xn.clone()
    xb[0, H // 2, W // 2, 3] = float("nan")
    ops.bottleneck_fused(xb, stream, exps, bnv, Cout, slope, amax_in=ops.absmax(xb, ops.new_amax_words(dev, 1)), range_flag=flag, cin=Cin)
    assert int(flag.item()) == 1


@pytest.mark.parametrize("N,H,W,C,gain", [(1, 10, 16, 256, 1.0), (2, 23, 37, 256, 1.0), (1, 31, 20, 64, 1.0), (1, 12, 50, 256, 500.0), (1, 9, 9, 128, 1e-4)])
def test_bottleneck_fused_matches_the_f64_block(ops, dev, N, H, W, C, gain):
    """tsod_bottleneck_fp16x2: a whole identity bottleneck in one launch (both 64-channel intermediates in LDS) against the f64
    CPU block - tile edges (H, W not multiples of the 10 x 16 tile; one-tile and multi-tile images), the zero padding of the 3x3
    (y1 is padded, not x: a border pixel's halo must be exact zeros), the tile-local scales of the intermediates and the range
    words of x (inputs 500x larger / 10^4 x smaller), and the abs-max it leaves for its consumer."""
    g = torch.Generator().manual_seed(123 + H)
    x = torch.randn(N, C, H, W, generator=g)
    x = torch.maximum(x, 0.25 * x) * gain
    w1 = torch.randn(64, C, 1, 1, generator=g) / math.sqrt(C)
    w2 = torch.randn(64, 64, 3, 3, generator=g) / math.sqrt(576)
    w3 = torch.randn(C, 64, 1, 1, generator=g) / 8.0
    bn = [torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.1 * gain, torch.rand(64, generator=g) + 0.5,
          torch.randn(64, generator=g) * 0.1 * gain, torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.1 * gain]
    slope = 0.25
    ref = _bottleneck_reference(x, w1, w2, w3, bn, slope)
    xn = ops.nchw_to_nhwc(x.to(dev))
    w2p = ops.pack_conv_weight(w2.to(dev))                              # [64, 3, 3, 64]
    stream, exps = ops.pack_bottleneck_wstream(w1.view(64, C).to(dev), w2p, w3.view(C, 64).to(dev))
    bnv = torch.cat(bn).to(dev)
    words = ops.absmax(xn, ops.new_amax_words(dev, 1))
    wout = ops.new_amax_words(dev, 1)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    out = ops.bottleneck_fused(xn, stream, exps, bnv, C, slope, amax_in=words, amax_out=wout, range_flag=flag)
    got = ops.nhwc_to_nchw(out).cpu()
    tol = (3e-6 * math.sqrt(576) + 1e-5) * float(ref.abs().max()) / 4.0 + 1e-6 * gain
    err = (got - ref).abs().max().item()
    assert err <= tol, (err, tol)
    assert int(flag.item()) == 0
    assert ops.amax_value(wout) == float(out.abs().max())
    # the same block as three launches of the conv kernel (fp16x2, f32 intermediates in HBM): same arithmetic, other summation order
    sc = lambda v: v.to(dev)      # noqa: E731
    y1 = ops.conv2d_nhwc(xn, ops.pack_conv_weight(w1.to(dev)), scale=sc(bn[0]), shift=sc(bn[1]), act=1, slope=slope, precision=2, amax_in=words)
    w_y1 = ops.absmax(y1, ops.new_amax_words(dev, 1))
    y2 = ops.conv2d_nhwc(y1, w2p, pad=1, scale=sc(bn[2]), shift=sc(bn[3]), act=1, slope=slope, precision=2, amax_in=w_y1)
    w_y2 = ops.absmax(y2, ops.new_amax_words(dev, 1))
    o3 = ops.conv2d_nhwc(y2, ops.pack_conv_weight(w3.to(dev)), scale=sc(bn[4]), shift=sc(bn[5]), residual=xn, act=1, slope=slope, precision=2,
                         amax_in=w_y2)
    assert (ops.nhwc_to_nchw(o3).cpu() - got).abs().max().item() <= tol
    if gain == 1.0 and C == 256:
        # static exponent path (no range words) and the refusals
        out2 = ops.bottleneck_fused(xn, stream, exps, bnv, C, slope, a_scale_exp=8)
        assert (ops.nhwc_to_nchw(out2).cpu() - ref).abs().max().item() <= tol
        xbad = xn.clone()
        xbad[0, 2, 3, 5] = float("nan")
        ops.bottleneck_fused(xbad, stream, exps, bnv, C, slope, a_scale_exp=8, range_flag=flag)
        assert int(flag.item()) == 1
        from two_stage_object_detection_amd._ffi import TsodError
        with pytest.raises(TsodError):                                 # channel counts must be multiples of 64
            ops.bottleneck_fused(xn, stream, exps, bnv, 96, slope)


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W,gain", [(1, 37, 53, 1.0), (2, 64, 130, 1.0), (1, 131, 260, 300.0), (1, 200, 336, 1e-3), (1, 9, 11, 1.0)])
def test_stem_fused_matches_the_f64_stem(ops, dev, N, H, W, gain):
    """tsod_stem_fp16x2: conv 7x7/2 + BN + PReLU + max pool 3x3/2 in one launch against the f64 CPU stem, from NCHW images and from
    NHWC4 images: tile edges (sizes that are not multiples of the 4 x 16 pooled tile, one-tile and multi-tile images, odd conv
    sizes: the pool's -inf padding), the conv's zero padding, the tile-local pixel scale (inputs 300x larger / 1000x smaller than
    unit range), the abs-max left for the consumer, and non-finite input raising the flag."""
    from two_stage_object_detection_amd._ffi import NHWC4Images
    g = torch.Generator().manual_seed(77 + H)
    x = torch.randn(N, 3, H, W, generator=g) * gain
    w = torch.randn(64, 3, 7, 7, generator=g) / math.sqrt(147)
    scale, shift = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.2 * gain
    slope = 0.25
    y = torch.nn.functional.conv2d(x.double(), w.double(), stride=2, padding=3) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    y = torch.clamp(y, min=0) + slope * torch.clamp(y, max=0)
    ref = torch.nn.functional.max_pool2d(y, 3, 2, 1).float()
    wfrag, e = ops.pack_stem_wfrag(w.to(dev))
    bn = torch.cat([scale, shift]).to(dev)
    words = ops.new_amax_words(dev, 2)
    flag = torch.zeros(1, dtype=torch.int32, device=dev)
    xd = x.to(dev)
    out = ops.stem_fused(xd, wfrag, e, bn, slope, amax_out=words[0], range_flag=flag)
    assert tuple(out.shape) == (N, ref.shape[2], ref.shape[3], 64)
    got = ops.nhwc_to_nchw(out).cpu()
    tol = (3e-6 * math.sqrt(147) + 1e-5) * float(y.abs().max()) + 1e-6 * gain
    err = (got - ref).abs().max().item()
    assert err <= tol, (err, tol)
    assert int(flag.item()) == 0
    assert ops.amax_value(words[0]) == float(out.abs().max())
    # the same images as the input step writes them (NHWC, 4 floats per pixel; whatever sits in channel 3 is ignored): bit-equal
    x4 = torch.full((N, H, W, 4), 7.0, device=dev)
    x4[..., :3] = xd.permute(0, 2, 3, 1)
    out4 = ops.stem_fused(NHWC4Images(x4), wfrag, e, bn, slope, amax_out=words[1], range_flag=flag)
    assert torch.equal(out4, out) and int(flag.item()) == 0
    # three launches of the existing kernels (layout, fp16x2 implicit GEMM with range words, max pool): same arithmetic, other summation order
    if gain == 1.0 and H > 16:
        xn = ops.nchw_to_nhwc(xd, 4)
        wp = ops.pack_conv_weight(w.to(dev), cin_pad=4, kw_pad=8)
        y3 = ops.conv2d_nhwc(xn, wp, stride=2, pad=3, kw_logical=7, scale=scale.to(dev), shift=shift.to(dev), act=1, slope=slope, precision=2,
                             amax_in=ops.absmax(xn, ops.new_amax_words(dev, 1)))
        p3 = ops.maxpool3x3s2_nhwc(y3)
        assert (p3 - out).abs().max().item() <= tol
    xbad = xd.clone()
    xbad[0, 1, H // 2, W // 3] = float("nan")
    ops.stem_fused(xbad, wfrag, e, bn, slope, range_flag=flag)
    assert int(flag.item()) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("N,H,W,segs,Cout", [
    (2, 9, 11, [(36, 24), (4, 20)], 20),                                # K = 44: three 16-k steps, the last with 12 live channels
    (1, 37, 53, [(0, 64), (200, 40), (96, 68), (320, 16)], 136),        # K = 188 over four slices of a 340-channel buffer, two column tiles
    (2, 50, 84, [(8, 320), (400, 196), (700, 116), (900, 68), (1000, 40)], 336),   # HarDNet-68's base.11.layers.15 shape: K = 740
])
def test_conv_fp16x2_lds_dma_over_channel_segments(ops, dev, N, H, W, segs, Cout):
    """conv_dma_kernel<..., CHAN>: a 1x1 conv over SEVERAL channel segments of a wider NHWC buffer (HarDNet's concatenated inputs) on
    the LDS-DMA tiles in fp16x2 - every tile, whole tiles / K-slices / hybrid, K no multiple of the K-step, range words - against the
    f64 CPU conv, and the same bits as the register-staged kernel's K order allows (same bar).  The balanced schedule has no such
    instantiation and is refused; bf16x3 keeps these layers off the LDS-DMA tiles."""
    from two_stage_object_detection_amd._ffi import TsodError
    P = max(o + n for o, n in segs) + 4
    K = sum(n for _, n in segs)
    g = torch.Generator().manual_seed(91 + K)
    x = torch.randn(N, P, H, W, generator=g)
    x = torch.maximum(x, 0.25 * x)
    w = torch.randn(Cout, K, 1, 1, generator=g) / math.sqrt(K)
    res = torch.randn(N, Cout, H, W, generator=g)
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    y = F.conv2d(torch.cat([x[:, o:o + n] for o, n in segs], dim=1).double(), w.double()) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
    y = y + res.double()
    ref = (torch.clamp(y, min=0) + 0.25 * torch.clamp(y, max=0)).float()
    xn, wp, rn = ops.nchw_to_nhwc(x.to(dev)), ops.pack_conv_weight(w.to(dev)), ops.nchw_to_nhwc(res.to(dev))
    words = ops.absmax(xn, ops.new_amax_words(dev, 1))
    tol = (3e-6 * math.sqrt(K) + 1e-5) * max(1.0, float(ref.abs().max()) / 4.0)
    kw = dict(segs=segs, scale=scale.to(dev), shift=shift.to(dev), residual=rn, act=1, slope=0.25, precision=2)
    for tile in (17, 19, 21, 22, 23, 24):
        for split in (1, 2, -1):
            out = ops.conv2d_nhwc(xn, wp, tile=tile, split_k=split, amax_in=words, **kw)
            err = (ops.nhwc_to_nchw(out).cpu() - ref).abs().max().item()
            assert err <= tol, (tile, split, err, tol)
    out = ops.conv2d_nhwc(xn, wp, tile=22, split_k=1, a_scale_exp=6, **kw)          # static exponent
    assert (ops.nhwc_to_nchw(out).cpu() - ref).abs().max().item() <= tol
    with pytest.raises(TsodError, match="unsupported|UNSUPPORTED"):
        ops.conv2d_nhwc(xn, wp, tile=22, split_k=-2, amax_in=words, **kw)
    with pytest.raises(TsodError, match="unsupported|UNSUPPORTED"):
        ops.conv2d_nhwc(xn, wp, tile=22, split_k=1, **{**kw, "precision": 1})
