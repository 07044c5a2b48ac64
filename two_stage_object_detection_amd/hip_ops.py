"""Tensor-level wrappers over the C ABI (one per entry point of include/tsod.h).

Every function takes CUDA/ROCm f32 tensors, allocates outputs with torch (device memory plumbing
only) and launches the HIP kernel on torch's current stream.  No fallback of any kind.
"""
from __future__ import annotations

import contextlib
import math
import threading
from ctypes import byref, c_int32

import torch

from . import _ffi
from ._ffi import ACT_NONE, check, lib, make_conv_desc, ptr, require_cuda, stream_ptr


# ----------------------------------------------------------------------------- workspace arena
class _Arena:
    """Growable scratch tensors (split-K slabs of the RPN / head GEMMs, NMS masks).

    Keyed by an explicit OWNER when one is in scope (``with ARENA.scope(owner)``: the detector enters one per
    (model, in-flight slot), so two slots or two detectors never share scratch whatever streams their graphs are
    replayed on), else by (device, current stream): stand-alone eager calls on one stream are ordered by that stream,
    and torch handing the same handle out twice means it IS the same HIP stream.
    A buffer that has been handed out is never freed or replaced under a HIP graph that may have its pointer baked in:
    outgrown buffers are retired, not released (``release(owner)`` drops an owner's buffers explicitly)."""

    def __init__(self, zero: bool = False):
        self._buf = {}
        self._retired = {}
        self._zero = zero            # conv workspaces start with arrival tickets that must be zero when first used

    def _key(self, device):
        owner = getattr(_OWNER_TLS, "owner", None)
        if owner is not None:
            return (device, "owner", owner)
        return (device, "stream", torch.cuda.current_stream(device).cuda_stream)

    def get(self, device, nbytes: int) -> torch.Tensor:
        nbytes = max(int(nbytes), 256)
        key = self._key(device)
        cur = self._buf.get(key)
        if cur is None or cur.numel() < nbytes:
            if cur is not None:
                self._retired.setdefault(key, []).append(cur)
            alloc = torch.zeros if self._zero else torch.empty
            cur = alloc(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
            self._buf[key] = cur
        return cur

    def release(self, owner) -> None:
        """Drop every buffer of ``owner`` (only once no graph captured under that owner will be replayed again)."""
        for key in [k for k in self._buf if k[1] == "owner" and k[2] == owner]:
            self._buf.pop(key, None)
            self._retired.pop(key, None)

    def scope(self, owner):                                   # both arenas share one owner scope
        return _owner_scope(owner)


_OWNER_TLS = threading.local()


@contextlib.contextmanager
def _owner_scope(owner):
    prev = getattr(_OWNER_TLS, "owner", None)
    _OWNER_TLS.owner = owner
    try:
        yield
    finally:
        _OWNER_TLS.owner = prev


ARENA = _Arena()
# K-sliced GEMM workspaces (tsod_conv2d_f32 / tsod_linear_f32): [arrival tickets | partial slabs].  The tickets must be
# zero when a launch starts and every launch leaves them zero, so these buffers are zero-filled once at allocation and
# never lent to any other kernel (the NMS masks live in ARENA).
CONV_ARENA = _Arena(zero=True)
TOPK_ARENA = _Arena()           # the selection handed from the top-k's select pass to its rank pass


# ----------------------------------------------------------------------------- layout
def nchw_to_nhwc(x: torch.Tensor, c_pad: int | None = None) -> torch.Tensor:
    """[N,C,H,W] -> [N,H,W,c_pad] (channels >= C zero-filled)."""
    require_cuda(x, "nchw_to_nhwc")
    x = x.contiguous()
    N, C, H, W = x.shape
    c_pad = C if c_pad is None else c_pad
    out = torch.empty((N, H, W, c_pad), dtype=torch.float32, device=x.device)
    check(lib().tsod_nchw_to_nhwc_f32(ptr(x), N, C, H, W, ptr(out), c_pad, c_pad, stream_ptr()), "nchw_to_nhwc")
    return out


def nhwc_to_nchw(x: torch.Tensor, C: int | None = None, c_off: int = 0) -> torch.Tensor:
    """[N,H,W,P] (channel slice [c_off, c_off+C)) -> [N,C,H,W]."""
    require_cuda(x, "nhwc_to_nchw")
    assert x.is_contiguous()
    N, H, W, P = x.shape
    C = P - c_off if C is None else C
    out = torch.empty((N, C, H, W), dtype=torch.float32, device=x.device)
    check(lib().tsod_nhwc_to_nchw_f32(ptr(x), N, C, H, W, P, c_off, ptr(out), stream_ptr()), "nhwc_to_nchw")
    return out


# ----------------------------------------------------------------------------- conv / linear
def pack_conv_weight(w: torch.Tensor, cin_pad: int | None = None, kw_pad: int | None = None) -> torch.Tensor:
    """torch [Cout,Cin,KH,KW] -> packed [Cout,KH,KW_pad,Cin_pad] on the device (zero-filled padding)."""
    require_cuda(w, "pack_conv_weight")
    w = w.detach().contiguous()
    Cout, Cin, KH, KW = w.shape
    cin_pad = Cin if cin_pad is None else cin_pad
    kw_pad = KW if kw_pad is None else kw_pad
    out = torch.empty((Cout, KH, kw_pad, cin_pad), dtype=torch.float32, device=w.device)
    check(lib().tsod_pack_conv_weight_f32(ptr(w), Cout, Cin, KH, KW, cin_pad, kw_pad, ptr(out), stream_ptr()),
          "pack_conv_weight")
    return out


def pack_conv_weight_bf16x3(w_packed: torch.Tensor) -> torch.Tensor:
    """f32 packed weights [Cout,KH,KW,Cin] -> the pre-split bf16x3 image tsod_conv2d_f32 reads with precision = bf16x3
    (uint8 tensor of tsod_conv_weight_bf16x3_bytes: [Cout][ceil(K/8)][hi|mid|lo][8] bf16)."""
    require_cuda(w_packed, "pack_conv_weight_bf16x3")
    w_packed = w_packed.contiguous()
    cout = w_packed.shape[0]
    K = w_packed.numel() // cout
    out = torch.empty(lib().tsod_conv_weight_bf16x3_bytes(cout, K), dtype=torch.uint8, device=w_packed.device)
    check(lib().tsod_pack_conv_weight_bf16x3(ptr(w_packed), cout, K, ptr(out), stream_ptr()), "pack_conv_weight_bf16x3")
    return out


def fp16x2_weight_scale_exp(w_packed: torch.Tensor) -> int:
    """The power of two that brings max |w| just below 2^14 (fp16's largest finite value is 65504; the low pieces of weights
    scaled like this stay out of its subnormals): the ``w_scale_exp`` of pack_conv_weight_fp16x2 and of the conv descriptor."""
    import math
    m = float(w_packed.abs().max())
    return 0 if m == 0.0 or not math.isfinite(m) else max(-40, min(40, int(math.floor(math.log2(16384.0 / m)))))


def pack_conv_weight_fp16x2(w_packed: torch.Tensor, w_scale_exp: int) -> torch.Tensor:
    """f32 packed weights [Cout,KH,KW,Cin] -> the fp16x2 image tsod_conv2d_f32 reads with precision = fp16x2
    (uint8 tensor of tsod_conv_weight_fp16x2_bytes: [Cout][ceil(K/8)][hi|lo][8] fp16 of 2^w_scale_exp * w)."""
    require_cuda(w_packed, "pack_conv_weight_fp16x2")
    w_packed = w_packed.contiguous()
    cout = w_packed.shape[0]
    K = w_packed.numel() // cout
    out = torch.empty(lib().tsod_conv_weight_fp16x2_bytes(cout, K), dtype=torch.uint8, device=w_packed.device)
    check(lib().tsod_pack_conv_weight_fp16x2(ptr(w_packed), cout, K, int(w_scale_exp), ptr(out), stream_ptr()), "pack_conv_weight_fp16x2")
    return out


_W3_CACHE: dict = {}


def _w3_for(w_packed: torch.Tensor) -> torch.Tensor:
    """Pre-split image of a weight tensor for the tensor-level wrapper when the caller did not bring one (``w3=`` of
    conv2d_nhwc; the engine, the RPN and the RoI head keep theirs beside the packed f32 weights).
    Cached per (storage address, shape, version counter); the entry holds the tensor's BASE strongly, so the address cannot be
    handed to another tensor while the entry lives, and a fresh ``w.view(...)`` of the same weights (a new Python object on
    every call) hits.  Inference tensors carry no version counter - an in-place edit would be invisible - so they are
    never cached: packed on every call (correct, slower; bring ``w3=`` on a hot path)."""
    if w_packed.is_inference():
        return pack_conv_weight_bf16x3(w_packed)
    key = (w_packed.data_ptr(), tuple(w_packed.shape), w_packed._version)
    hit = _W3_CACHE.get(key)
    if hit is None:
        if len(_W3_CACHE) >= 64:
            _W3_CACHE.clear()
        base = w_packed._base if w_packed._base is not None else w_packed
        w3 = pack_conv_weight_bf16x3(w_packed)
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(w3.device).synchronize()      # complete before another stream can hit the entry
        hit = _W3_CACHE[key] = (base, w3)
    return hit[1]


def conv2d_nhwc(x: torch.Tensor, w_packed: torch.Tensor, *, stride=1, pad=0, kw_logical=None, scale=None, shift=None,
                residual=None, act=ACT_NONE, slope=0.0, segs=None, out=None, out_off=0, tile=0, split_k=0,
                precision=0, x2=None, stride2=1, x2_off=0, w3=None, a_scale_exp=4, w2=None, w_scale_exp=None,
                range_flag=None, amax_in=None, amax_in2=None, amax_out=None) -> torch.Tensor:
    """Implicit-GEMM convolution on an NHWC tensor [N,H,W,P].  ``w_packed`` is [Cout,KH,KW,Cin]
    (see pack_conv_weight); ``kw_logical`` is the filter width before zero-tap padding (it fixes OW).
    ``segs`` = [(channel offset, length), ...] inside the P-wide pixel (default: the first Cin
    channels).  Returns / fills an NHWC output [N,OH,OW,Pout].
    ``x2`` [N,H2,W2,P2] is the optional second source of tsod_conv2d_dual_f32: ``w_packed`` is then [Cout, K1 + C2] with the
    last C2 columns contracting channels [x2_off, x2_off + C2) of pixel (oh*stride2, ow*stride2) of ``x2`` (``kernel`` =
    (KH, KW, Cin) of the first source must be given through ``segs`` / a 1x1 filter: only 1x1 first sources here).
    ``w3``: the pre-split bf16x3 image of ``w_packed`` (pack_conv_weight_bf16x3) when the caller keeps one; used with
    ``precision`` = bf16x3 instead of the wrapper's own cache.  ``precision`` = fp16x2: activations are split as
    2^``a_scale_exp`` * x (|that| must stay below 65504), ``w2`` / ``w_scale_exp`` = the fp16x2 weight image and the exponent it
    was packed with (made on the spot from ``w_packed`` when not given); ``range_flag``: an int32 device word the launch sets to 1
    when an activation left that range.  ``amax_out`` / ``amax_in`` / ``amax_in2``: range words (``new_amax_words``; a tensor or
    a raw device pointer): the launch adds its outputs' abs-max to ``amax_out``; an fp16x2 launch takes its activation exponent
    from ``amax_in`` (and ``amax_in2`` for ``x2``) instead of ``a_scale_exp``."""
    require_cuda(x, "conv2d")
    assert x.is_contiguous() and w_packed.is_contiguous()
    N, H, W, P = x.shape
    src2 = None
    if x2 is not None:                         # stacked [Cout, Cin1 + C2] weights of a 1x1 conv + a strided 1x1 tap of x2
        assert w_packed.dim() == 2 and x2.is_contiguous() and segs is not None and len(segs) == 1
        Cout, KH, KW, Cin = w_packed.shape[0], 1, 1, segs[0][1]
        src2 = (w_packed.shape[1] - Cin, x2.shape[3], x2_off, stride2, x2.shape[1], x2.shape[2])
    else:
        Cout, KH, KW, Cin = w_packed.shape
    segs = [(0, Cin)] if segs is None else segs
    assert sum(s[1] for s in segs) == Cin, "segments must add up to the packed Cin"
    OH = (H + 2 * pad - KH) // stride + 1
    OW = (W + 2 * pad - (KW if kw_logical is None else kw_logical)) // stride + 1
    if out is None:
        out = torch.empty((N, OH, OW, Cout), dtype=torch.float32, device=x.device)
    assert out.is_contiguous() and out.shape[:3] == (N, OH, OW)
    d = make_conv_desc(N=N, H=H, W=W, in_pitch=P, segs=segs, Cout=Cout, out_pitch=out.shape[3], out_off=out_off,
                       KH=KH, KW=KW, stride=stride, pad_h=pad, pad_w=pad, OH=OH, OW=OW, act=act, slope=slope,
                       res_pitch=0 if residual is None else residual.shape[-1], res_off=0, tile=tile, split_k=split_k,
                       precision=precision, src2=src2)
    if precision == _ffi.PREC_FP16X2:
        if w2 is None:
            w_scale_exp = fp16x2_weight_scale_exp(w_packed)
            w2 = pack_conv_weight_fp16x2(w_packed, w_scale_exp)
        d.a_scale_exp, d.w_scale_exp = int(a_scale_exp), int(w_scale_exp)
        d.range_flag = ptr(range_flag)                            # optional int32 [1] device tensor: 1 = an activation left the range
        d.amax_in, d.amax_in2 = _word_ptr(amax_in), _word_ptr(amax_in2)
    d.amax_out = _word_ptr(amax_out)
    ws_bytes = lib().tsod_conv2d_workspace_bytes(byref(d))
    ws = CONV_ARENA.get(x.device, ws_bytes) if ws_bytes else None
    # bf16x3 / fp16x2 read their pre-split weight images
    w_arg = (w3 if w3 is not None else _w3_for(w_packed)) if precision == _ffi.PREC_BF16X3 else (w2 if precision == _ffi.PREC_FP16X2 else w_packed)
    check(lib().tsod_conv2d_dual_f32(byref(d), ptr(x), ptr(x2), ptr(w_arg), ptr(scale), ptr(shift), ptr(residual), ptr(out),
                                     ptr(ws), ws_bytes, stream_ptr()), "conv2d")
    return out


def pack_bottleneck_wstream(w1: torch.Tensor, w2: torch.Tensor, w3: torch.Tensor, projection: bool = False):
    """The weight stream of tsod_bottleneck_fp16x2 (include/tsod.h): w1 [64, Cin], w2 [64, 3, 3, 64] (packed conv layout:
    [Cout][KH][KW][Cin]), w3 [Cout, 64], f32 -> (uint8 tensor of tsod_bottleneck_wstream_bytes, (e1, e2, e3)).
    ``projection``: w3 is the stacked [Cout, 64 + Cin] matrix [W3 s3 | Wd sd] of desc.projection == 1 (2 + Cin / 32 steps per 64
    output channels, one exponent for the whole matrix).
    Pure index arithmetic on the three matrices (done once per model; the fp16 roundings are torch's round-to-nearest-even,
    the same bits as the device's v_cvt_pk_f16_f32).  A step (64 output channels x 32 k) is stored as the MFMA fragments the
    kernel's lanes load: [channel block cb (2)][lane (64) = 32 hh + j][chunk c (2)][hi | lo][8 k] fp16, where lane (j, hh) holds
    output channel 32 cb + pi(j) and k = 16 c + 8 hh .. + 7."""
    dev = w1.device
    cin, cout = w1.shape[1], w3.shape[0]
    i = torch.arange(32)
    pi = 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3)
    rows = torch.cat([pi, 32 + pi]).to(dev)                                 # fragment row 32 cb + j  <-  channel 32 cb + pi(j)

    def steps(w2d, e):
        """w2d [n_rows (multiple of 64), K (multiple of 32)] -> [n_rows/64 * K/32 steps, 8192 bytes], row-block-major then k"""
        sc = w2d.float() * (2.0 ** e)
        hi = sc.half()
        lo = (sc - hi.float()).half()
        nb, ks = sc.shape[0] // 64, sc.shape[1] // 32
        # [nb, cb, j, ks, c, hh, 8] for each plane
        pl = torch.stack([t.view(nb, 64, ks, 2, 2, 8)[:, rows].view(nb, 2, 32, ks, 2, 2, 8) for t in (hi, lo)], dim=0)   # [plane, nb, cb, j, ks, c, hh, 8]
        # -> [nb, ks, cb, hh, j, c, plane, 8]
        out = pl.permute(1, 4, 2, 6, 3, 5, 0, 7).contiguous()
        return out.view(nb * ks, 64 * 64).view(torch.uint8)                  # 4096 halves = 8192 bytes per step

    e1, e2, e3 = (fp16x2_weight_scale_exp(w) for w in (w1, w2, w3))
    s1 = steps(w1.reshape(64, cin), e1)
    # conv2: step = (tap, channel half): K order of the packed layout is (kh, kw, ci), so k = 32 * (2 tap + half) already
    s2 = steps(w2.reshape(64, 9 * 64), e2)
    s3 = steps(w3.reshape(cout, 64 + cin if projection else 64), e3)         # row blocks of 64 output channels, 2 (+ Cin / 32) k-steps each
    stream = torch.cat([s1, s2, s3], dim=0).contiguous().view(-1)
    assert stream.numel() == (lib().tsod_bottleneck_proj_wstream_bytes if projection else lib().tsod_bottleneck_wstream_bytes)(cin, cout)
    return stream, (e1, e2, e3)


def bottleneck_fused(x: torch.Tensor, wstream: torch.Tensor, w_exps, bn: torch.Tensor, cout: int, slope: float, *, out=None,
                     a_scale_exp=4, amax_in=None, amax_out=None, range_flag=None, cin=None) -> torch.Tensor:
    """tsod_bottleneck_fp16x2 on an NHWC tensor x [N,H,W,P] (channels [0, cout) are the block's input; ``cin``: the projection
    form, channels [0, cin) in, cout out): see include/tsod.h."""
    require_cuda(x, "bottleneck_fused")
    N, H, W, P = x.shape
    if out is None:
        out = torch.empty((N, H, W, cout), dtype=torch.float32, device=x.device)
    d = _ffi.BottleneckDesc()
    d.N, d.H, d.W, d.Cin, d.in_pitch, d.Cmid, d.Cout, d.out_pitch = N, H, W, cout if cin is None else cin, P, 64, cout, out.shape[3]
    d.projection = 0 if cin is None else 1
    d.slope = float(slope)
    for k in range(3):
        d.w_exp[k] = int(w_exps[k])
    d.a_scale_exp = int(a_scale_exp)
    d.range_flag, d.amax_in, d.amax_out = ptr(range_flag) or None, _word_ptr(amax_in), _word_ptr(amax_out)
    check(lib().tsod_bottleneck_fp16x2(byref(d), ptr(x), ptr(wstream), ptr(bn), ptr(out), stream_ptr()), "bottleneck_fused")
    return out


def pack_stem_wfrag(w: torch.Tensor):
    """The weight fragments of tsod_stem_fp16x2 (include/tsod.h): conv1's weight [64, 3, 7, 7] f32 -> (uint8 tensor of
    tsod_stem_wfrag_bytes, w_exp).  K = (kh, kw padded to 8, ci padded to 4) = 224; [channel block cb (2)][chunk c (14)][hi | lo]
    [lane (64) = 32 hh + j][8 k] fp16, lane (j, hh) holding output channel 32 cb + pi(j) and k = 16 c + 8 hh .. + 7.  Index
    arithmetic only; the fp16 roundings are torch's round-to-nearest-even (the bits of the device's conversions)."""
    assert tuple(w.shape) == (64, 3, 7, 7), tuple(w.shape)
    dev = w.device
    e = fp16x2_weight_scale_exp(w)
    wk = torch.zeros((64, 7, 8, 4), dtype=torch.float32, device=dev)
    wk[:, :, :7, :3] = w.detach().float().permute(0, 2, 3, 1)
    sc = wk.reshape(64, 224) * (2.0 ** e)
    hi = sc.half()
    lo = (sc - hi.float()).half()
    i = torch.arange(32)
    pi = (16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3)).to(dev)
    pl = torch.stack([hi, lo], dim=0).view(2, 2, 32, 14, 2, 8)[:, :, pi]     # [plane, cb, j, c, hh, 8]: row j <- channel 32 cb + pi(j)
    out = pl.permute(1, 3, 0, 4, 2, 5).contiguous().view(-1).view(torch.uint8)   # [cb, c, plane, hh, j, 8]
    assert out.numel() == lib().tsod_stem_wfrag_bytes()
    return out, e


def stem_fused(x, wfrag: torch.Tensor, w_exp: int, bn: torch.Tensor, slope: float, *, out=None, amax_out=None, range_flag=None):
    """tsod_stem_fp16x2: conv1 7x7/2 + BN + PReLU + max pool 3x3/2 of ResNet in one launch.  ``x``: an NCHW tensor [N,3,H,W] or
    ``NHWC4Images``; returns the pooled NHWC map [N,PH,PW,64]."""
    nhwc4 = isinstance(x, _ffi.NHWC4Images)
    t = x.data if nhwc4 else x.contiguous()
    require_cuda(t, "stem_fused")
    if nhwc4:
        N, H, W, _ = t.shape
    else:
        N, _, H, W = t.shape
    oh, ow = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    ph, pw = (oh - 1) // 2 + 1, (ow - 1) // 2 + 1
    if out is None:
        out = torch.empty((N, ph, pw, 64), dtype=torch.float32, device=t.device)
    d = _ffi.StemDesc()
    d.N, d.H, d.W, d.in_layout, d.out_pitch = N, H, W, (_ffi.STEM_NHWC4 if nhwc4 else _ffi.STEM_NCHW), out.shape[3]
    d.slope, d.w_exp = float(slope), int(w_exp)
    d.range_flag, d.amax_out = ptr(range_flag) or None, _word_ptr(amax_out)
    check(lib().tsod_stem_fp16x2(byref(d), ptr(t), ptr(wfrag), ptr(bn), ptr(out), stream_ptr()), "stem_fused")
    return out


def _word_ptr(w):
    """None / a raw device pointer / a tensor of range words -> what the descriptor takes."""
    if w is None:
        return None
    return (int(w) if not isinstance(w, torch.Tensor) else w.data_ptr()) or None


def new_amax_words(device, n: int = 1) -> torch.Tensor:
    """Zeroed range words for ``n`` tensors (include/tsod.h "Range words"): int32 [n, AMAX_BYTES / 4]; row i is one tensor's."""
    return torch.zeros((n, _ffi.AMAX_BYTES // 4), dtype=torch.int32, device=device)


def amax_value(words: torch.Tensor) -> float:
    """The abs-max a set of range words holds (host read; tests and diagnostics)."""
    w = words.reshape(-1)[:: _ffi.AMAX_STRIDE // 4][: _ffi.AMAX_WORDS]
    return float(w.max().view(1).view(torch.float32))


def absmax(x: torch.Tensor, words: torch.Tensor) -> torch.Tensor:
    """Add the abs-max of ``x`` to ``words`` (tsod_absmax_f32)."""
    require_cuda(x, "absmax")
    check(lib().tsod_absmax_f32(ptr(x), x.numel(), ptr(words), stream_ptr()), "absmax")
    return words


def tune_conv(x: torch.Tensor, w_packed: torch.Tensor, reps: int = 5, precisions=(0, 1), **kw) -> tuple[int, int, int]:
    """Time every (tile, K-slice schedule, arithmetic) of ONE conv2d_nhwc call on its real operands with HIP events and return
    the fastest as (tile, split_k, precision) - for the few GEMMs outside a backbone plan (the fused RPN conv, the fused head
    GEMM).  A speed choice only: every candidate is f32-accurate.  Precision 2 (fp16x2) among ``precisions`` needs ``w2`` /
    ``w_scale_exp`` and - for a range-proof scale - ``amax_in`` in ``kw`` (they are ignored by the other arithmetics)."""
    K = w_packed.numel() // w_packed.shape[0]
    ksteps = (K + 31) // 32
    best = None
    for prec in precisions:
        for tile in (_ffi.BF16X3_TILE_IDS if prec == _ffi.PREC_BF16X3 else (_ffi.FP16X2_TILE_IDS if prec == _ffi.PREC_FP16X2 else _ffi.TILE_IDS)):
            for split in (1, -1, -2, 2, 3, 4, 6, 8, 12, 16, 24, 32):
                if split > 1 and ksteps // split < 2:
                    continue
                try:
                    conv2d_nhwc(x, w_packed, tile=tile, split_k=split, precision=prec, **kw)          # warm (and validity)
                except _ffi.TsodError:
                    continue
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    conv2d_nhwc(x, w_packed, tile=tile, split_k=split, precision=prec, **kw)
                e1.record()
                e1.synchronize()
                t = e0.elapsed_time(e1) / reps
                if best is None or t < best[0]:
                    best = (t, tile, split, prec)
    return best[1], best[2], best[3]


def conv2d_resolve(d) -> tuple[int, int]:
    t, s = c_int32(0), c_int32(0)
    check(lib().tsod_conv2d_resolve(byref(d), byref(t), byref(s)), "conv2d_resolve")
    return t.value, s.value


def linear(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor | None = None) -> torch.Tensor:
    """nn.Linear: x [M,K] @ weight[N,K]^T + bias."""
    require_cuda(x, "linear")
    x = x.contiguous()
    weight = weight.detach().contiguous()
    M, K = x.shape
    N = weight.shape[0]
    out = torch.empty((M, N), dtype=torch.float32, device=x.device)
    ws_bytes = lib().tsod_linear_workspace_bytes(M, K, N)
    ws = CONV_ARENA.get(x.device, ws_bytes) if ws_bytes else None
    check(lib().tsod_linear_f32(ptr(x), M, K, K, ptr(weight), ptr(bias), N, ptr(out), N, ptr(ws), ws_bytes,
                                stream_ptr()), "linear")
    return out


# ----------------------------------------------------------------------------- HBM-bound layers
def maxpool3x3s2_nhwc(x: torch.Tensor) -> torch.Tensor:
    require_cuda(x, "maxpool")
    N, H, W, C = x.shape
    out = torch.empty((N, (H - 1) // 2 + 1, (W - 1) // 2 + 1, C), dtype=torch.float32, device=x.device)
    check(lib().tsod_maxpool3x3s2_f32(ptr(x), N, H, W, C, C, ptr(out), C, stream_ptr()), "maxpool")
    return out


def dwconv3x3_nhwc(x: torch.Tensor, w33c: torch.Tensor, scale=None, shift=None, stride=1, relu=False, C=None,
                   in_off=0, out=None, out_off=0) -> torch.Tensor:
    """Depthwise 3x3 pad 1 on channels [in_off, in_off+C) of x [N,H,W,P]; w33c is [3,3,C]."""
    require_cuda(x, "dwconv3x3")
    N, H, W, P = x.shape
    C = w33c.shape[2] if C is None else C
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    if out is None:
        out = torch.empty((N, OH, OW, C), dtype=torch.float32, device=x.device)
    check(lib().tsod_dwconv3x3_f32(ptr(x), N, H, W, C, P, in_off, ptr(w33c), ptr(scale), ptr(shift), stride,
                                   1 if relu else 0, ptr(out), out.shape[3], out_off, stream_ptr()), "dwconv3x3")
    return out


def gconv3x3_nhwc(x: torch.Tensor, w_packed: torch.Tensor, groups: int, scale=None, shift=None, stride=1, act=ACT_NONE,
                  slope=0.0) -> torch.Tensor:
    """Grouped 3x3 pad-1 conv on NHWC [N,H,W,C] -> [N,OH,OW,C]; w_packed is [C,3,3,C/groups] (ResNeXt's conv2)."""
    require_cuda(x, "gconv3x3")
    N, H, W, C = x.shape
    OH, OW = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = torch.empty((N, OH, OW, C), dtype=torch.float32, device=x.device)
    check(lib().tsod_gconv3x3_f32(ptr(x), N, H, W, C, C, int(groups), ptr(w_packed.contiguous()), ptr(scale), ptr(shift),
                                  int(stride), int(act), float(slope), ptr(out), C, stream_ptr()), "gconv3x3")
    return out


def gconv1x1_pair_nhwc(x: torch.Tensor, w_g2: torch.Tensor, bias=None) -> torch.Tensor:
    require_cuda(x, "gconv1x1_pair")
    N, H, W, P = x.shape
    G = w_g2.shape[0]
    out = torch.empty((N, H, W, G), dtype=torch.float32, device=x.device)
    check(lib().tsod_gconv1x1_pair_f32(ptr(x), N * H * W, G, P, ptr(w_g2), ptr(bias), ptr(out), G, stream_ptr()),
          "gconv1x1_pair")
    return out


# ----------------------------------------------------------------------------- proposal path
def rpn_decode(locs: torch.Tensor, scores: torch.Tensor, anchor_base: torch.Tensor, B, Hf, Wf, feat_stride,
               clamp_x, clamp_y, min_size, want_anchors=False):
    """locs [B*Hf*Wf, 4A] , scores [B*Hf*Wf, 2A] (rows may be slices of a wider buffer: row pitch = stride(0)) ->
    boxes [B,Hf*Wf*A,4], fg [B,n], keys [B,n] (+ anchors [n,4])."""
    require_cuda(locs, "rpn_decode")
    A = anchor_base.shape[0]
    assert locs.stride(1) == 1 and scores.stride(1) == 1 and locs.shape[1] == 4 * A and scores.shape[1] == 2 * A
    n = Hf * Wf * A
    dev = locs.device
    boxes = torch.empty((B, n, 4), dtype=torch.float32, device=dev)
    fg = torch.empty((B, n), dtype=torch.float32, device=dev)
    keys = torch.empty((B, n), dtype=torch.float32, device=dev)
    anchors = torch.empty((n, 4), dtype=torch.float32, device=dev) if want_anchors else None
    check(lib().tsod_rpn_decode_f32(ptr(locs), locs.stride(0), ptr(scores), scores.stride(0), ptr(anchor_base), A, B,
                                    Hf, Wf, feat_stride, float(clamp_x), float(clamp_y), float(min_size), ptr(boxes),
                                    ptr(fg), ptr(keys), ptr(anchors), stream_ptr()), "rpn_decode")
    return boxes, fg, keys, anchors


def proposal_decode(anchor: torch.Tensor, loc: torch.Tensor, score: torch.Tensor, clamp_x, clamp_y, min_size):
    """anchor [n,4], loc [n,4], score [n] -> boxes [n,4] (decoded + clamped), keys [n] (-inf = too small)."""
    require_cuda(loc, "proposal_decode")
    anchor, loc, score = anchor.float().contiguous(), loc.contiguous(), score.contiguous()
    n = loc.shape[0]
    boxes = torch.empty((n, 4), dtype=torch.float32, device=loc.device)
    keys = torch.empty((n,), dtype=torch.float32, device=loc.device)
    check(lib().tsod_proposal_decode_f32(ptr(anchor), ptr(loc), ptr(score), n, float(clamp_x), float(clamp_y),
                                         float(min_size), ptr(boxes), ptr(keys), stream_ptr()), "proposal_decode")
    return boxes, keys


def enumerate_anchors(anchor_base: torch.Tensor, feat_stride: int, height: int, width: int) -> torch.Tensor:
    require_cuda(anchor_base, "enumerate_anchors")
    base = anchor_base.contiguous()
    out = torch.empty((height * width * base.shape[0], 4), dtype=torch.float32, device=base.device)
    check(lib().tsod_enumerate_anchors_f32(ptr(base), base.shape[0], height, width, int(feat_stride), ptr(out),
                                           stream_ptr()), "enumerate_anchors")
    return out


def loc2bbox(src: torch.Tensor, loc: torch.Tensor) -> torch.Tensor:
    require_cuda(loc, "loc2bbox")
    src, loc = src.to(loc.dtype).contiguous(), loc.contiguous()
    out = torch.empty_like(loc)
    if loc.shape[0]:
        check(lib().tsod_loc2bbox_f32(ptr(src), ptr(loc), loc.shape[0], ptr(out), stream_ptr()), "loc2bbox")
    return out


def bbox2loc(src: torch.Tensor, dst: torch.Tensor) -> torch.Tensor:
    """src [n,4], dst [n,4] xyxy -> offsets [n,4] (tsod_bbox2loc_f32; utils/loc_bbox_iou.py:63-88)."""
    require_cuda(src, "bbox2loc")
    src, dst = src.float().contiguous(), dst.to(src.device, torch.float32).contiguous()
    out = torch.empty_like(src)
    if src.shape[0]:
        check(lib().tsod_bbox2loc_f32(ptr(src), ptr(dst), src.shape[0], ptr(out), stream_ptr()), "bbox2loc")
    return out


def sort_topk_desc(keys: torch.Tensor, boxes: torch.Tensor | None, n_pre: int):
    """keys [B,n] (-inf = filtered), boxes [B,n,4] -> counts [B] i32, idx [B,n_pre] i32,
    boxes_sorted [B,n_pre,4], keys_sorted [B,n_pre]."""
    require_cuda(keys, "sort_topk")
    B, n = keys.shape
    dev = keys.device
    counts = torch.empty((B,), dtype=torch.int32, device=dev)
    idx = torch.empty((B, n_pre), dtype=torch.int32, device=dev)
    bs = torch.empty((B, n_pre, 4), dtype=torch.float32, device=dev) if boxes is not None else None
    ks = torch.empty((B, n_pre), dtype=torch.float32, device=dev)
    ws_bytes = lib().tsod_sort_topk_workspace_bytes(B, n, n_pre)
    ws = TOPK_ARENA.get(dev, ws_bytes) if ws_bytes else None          # its own arena: the NMS mask of the same step lives in ARENA
    check(lib().tsod_sort_topk_desc_ws_f32(ptr(keys), ptr(boxes), B, n, n_pre, ptr(counts), ptr(idx), ptr(bs), ptr(ks),
                                           ptr(ws), ws_bytes, stream_ptr()), "sort_topk")
    return counts, idx, bs, ks


def nms_sorted(boxes_sorted: torch.Tensor, counts: torch.Tensor, iou_thr: float, n_post: int, status=None):
    """boxes_sorted [B,n_max,4] in descending-score order, counts [B] i32 ->
    keep_idx [B,n_post] i32, rois [B,n_post,4], n_kept [B] i32, status [1] i32."""
    require_cuda(boxes_sorted, "nms")
    B, n_max, _ = boxes_sorted.shape
    dev = boxes_sorted.device
    keep = torch.empty((B, n_post), dtype=torch.int32, device=dev)
    rois = torch.empty((B, n_post, 4), dtype=torch.float32, device=dev)
    n_kept = torch.empty((B,), dtype=torch.int32, device=dev)
    if status is None:
        status = torch.zeros((1,), dtype=torch.int32, device=dev)
    ws_bytes = lib().tsod_nms_workspace_bytes(B, n_max)
    ws = ARENA.get(dev, ws_bytes)
    check(lib().tsod_nms_f32(ptr(boxes_sorted), ptr(counts), B, n_max, float(iou_thr), n_post, ptr(keep), ptr(rois),
                             ptr(n_kept), ptr(status), ptr(ws), ws_bytes, stream_ptr()), "nms")
    return keep, rois, n_kept, status


def bbox_iou(a: torch.Tensor, b: torch.Tensor, eps: float = 1e-8) -> torch.Tensor:
    require_cuda(a, "bbox_iou")
    if a.shape[1] != 4 or b.shape[1] != 4:
        raise IndexError
    a, b = a.contiguous(), b.contiguous()
    out = torch.empty((a.shape[0], b.shape[0]), dtype=torch.float32, device=a.device)
    if out.numel():
        check(lib().tsod_bbox_iou_f32(ptr(a), a.shape[0], ptr(b), b.shape[0], float(eps), ptr(out), stream_ptr()),
              "bbox_iou")
    return out


# ----------------------------------------------------------------------------- RoI head
def roi_pool_nhwc(feat: torch.Tensor, rois5: torch.Tensor, output_size=(7, 7), spatial_scale=1.0) -> torch.Tensor:
    """feat [B,Hf,Wf,C] NHWC, rois5 [K,5] -> [K,C,PH,PW] (torchvision layout)."""
    require_cuda(feat, "roi_pool")
    B, Hf, Wf, C = feat.shape
    K = rois5.shape[0]
    PH, PW = output_size
    out = torch.empty((K, C, PH, PW), dtype=torch.float32, device=feat.device)
    check(lib().tsod_roi_pool_f32(ptr(feat), B, Hf, Wf, C, C, ptr(rois5.contiguous()), K, float(spatial_scale), PH, PW,
                                  ptr(out), stream_ptr()), "roi_pool")
    return out


def roi_pool_avg_nhwc(feat: torch.Tensor, rois: torch.Tensor, roi_indices: torch.Tensor, img_h, img_w,
                      output_size=(7, 7), spatial_scale=1.0) -> torch.Tensor:
    """feat [B,Hf,Wf,C], rois [B,R,4] image coords, roi_indices [B] i32 -> [B*R, C]."""
    require_cuda(feat, "roi_pool_avg")
    B, Hf, Wf, C = feat.shape
    R = rois.shape[1]
    PH, PW = output_size
    out = torch.empty((rois.shape[0] * R, C), dtype=torch.float32, device=feat.device)
    check(lib().tsod_roi_pool_avg_f32(ptr(feat), B, Hf, Wf, C, C, ptr(rois.contiguous()),
                                      ptr(roi_indices.to(torch.int32).contiguous()), R, float(img_h), float(img_w),
                                      float(spatial_scale), PH, PW, ptr(out), C, stream_ptr()), "roi_pool_avg")
    return out


def _row_pitch(t: torch.Tensor, width: int) -> int | None:
    """Pitch (floats) of the [B*R, width] row matrix behind a [B,R,width] tensor whose rows may be slices of a wider
    buffer (the fused head GEMM writes both outputs into one [B*R, 408] matrix); None if it is not such a matrix."""
    B, R, w = t.shape
    if w != width or t.stride(2) != 1 or (B > 1 and t.stride(0) != R * t.stride(1)) or t.stride(1) < width:
        return None
    return t.stride(1)


def roi_align_nhwc(feat: torch.Tensor, rois5: torch.Tensor, output_size=(7, 7), spatial_scale=1.0, sampling_ratio=2,
                   aligned=False) -> torch.Tensor:
    """feat [B,Hf,Wf,C] NHWC, rois5 [K,5] -> [K,C,PH,PW] (torchvision.ops.roi_align semantics and layout)."""
    require_cuda(feat, "roi_align")
    B, Hf, Wf, C = feat.shape
    K = rois5.shape[0]
    PH, PW = output_size
    out = torch.empty((K, C, PH, PW), dtype=torch.float32, device=feat.device)
    check(lib().tsod_roi_align_f32(ptr(feat), B, Hf, Wf, C, C, ptr(rois5.contiguous()), K, float(spatial_scale), PH, PW,
                                   int(sampling_ratio), 1 if aligned else 0, ptr(out), stream_ptr()), "roi_align")
    return out


def roi_align_avg_nhwc(feat: torch.Tensor, rois: torch.Tensor, roi_indices: torch.Tensor, img_h, img_w, output_size=(7, 7),
                       spatial_scale=1.0, sampling_ratio=2, aligned=False) -> torch.Tensor:
    """feat [B,Hf,Wf,C], rois [B,R,4] image coords, roi_indices [B] i32 -> [B*R, C] (rescale + RoIAlign + mean over bins)."""
    require_cuda(feat, "roi_align_avg")
    B, Hf, Wf, C = feat.shape
    R = rois.shape[1]
    PH, PW = output_size
    out = torch.empty((rois.shape[0] * R, C), dtype=torch.float32, device=feat.device)
    check(lib().tsod_roi_align_avg_f32(ptr(feat), B, Hf, Wf, C, C, ptr(rois.contiguous()),
                                       ptr(roi_indices.to(torch.int32).contiguous()), R, float(img_h), float(img_w),
                                       float(spatial_scale), PH, PW, int(sampling_ratio), 1 if aligned else 0, ptr(out), C,
                                       stream_ptr()), "roi_align_avg")
    return out


def detections(cls_locs: torch.Tensor, scores: torch.Tensor, rois: torch.Tensor) -> torch.Tensor:
    """[B,R,4*n_class], [B,R,n_class], [B,R,4] -> [B,R,6] (x1,y1,x2,y2,score,class)."""
    require_cuda(scores, "detections")
    B, R, n_class = scores.shape
    lp, sp = _row_pitch(cls_locs, 4 * n_class), _row_pitch(scores, n_class)
    if lp is None:
        cls_locs, lp = cls_locs.contiguous(), 4 * n_class
    if sp is None:
        scores, sp = scores.contiguous(), n_class
    out = torch.empty((B, R, 6), dtype=torch.float32, device=scores.device)
    check(lib().tsod_detections_f32(ptr(cls_locs), lp, ptr(scores), sp, ptr(rois.contiguous()), B * R, n_class, ptr(out),
                                    stream_ptr()), "detections")
    return out


def filter_detections(det: torch.Tensor, iou_thr: float = 0.1, score_thresh: float | None = None, per_class: bool = False,
                      background_class: int = -1):
    """Inference-time filtering of detection records [B,R,6] (multi_inference.py:80-87 + the two deployment switches):
    drop records below ``score_thresh`` / of ``background_class``, order by descending score (stable), greedy NMS
    (class-agnostic like the reference's demo, or per class).  Returns (det_sorted [B,R,6], keep [B,R] i32 rows of
    det_sorted in score order with -1 after n_kept[b], n_kept [B] i32)."""
    require_cuda(det, "filter_detections")
    det = det.contiguous()
    B, R, six = det.shape
    if six != 6:
        raise ValueError("detection records are [B,R,6]")
    dev = det.device
    L = lib()
    keys = torch.empty((B, R), dtype=torch.float32, device=dev)
    thr = float("-inf") if score_thresh is None else float(score_thresh)
    check(L.tsod_detection_keys_f32(ptr(det), B * R, thr, int(background_class), ptr(keys), stream_ptr()), "detection_keys")
    counts = torch.empty((B,), dtype=torch.int32, device=dev)
    idx = torch.empty((B, R), dtype=torch.int32, device=dev)
    ws_bytes = L.tsod_sort_topk_workspace_bytes(B, R, R)
    ws_sort = TOPK_ARENA.get(dev, ws_bytes) if ws_bytes else None
    check(L.tsod_sort_topk_desc_ws_f32(ptr(keys), None, B, R, R, ptr(counts), ptr(idx), None, None, ptr(ws_sort), ws_bytes,
                                       stream_ptr()), "sort_topk")
    det_sorted = torch.empty_like(det)
    check(L.tsod_gather_rows_f32(ptr(det), ptr(idx), B, R, R, 6, ptr(det_sorted), stream_ptr()), "gather_rows")
    keep = torch.empty((B, R), dtype=torch.int32, device=dev)
    n_kept = torch.empty((B,), dtype=torch.int32, device=dev)
    ws_bytes = L.tsod_nms_workspace_bytes(B, R)
    ws = ARENA.get(dev, ws_bytes)
    check(L.tsod_detection_nms_f32(ptr(det_sorted), ptr(counts), B, R, float(iou_thr), 1 if per_class else 0, ptr(keep),
                                   ptr(n_kept), ptr(ws), ws_bytes, stream_ptr()), "detection_nms")
    return det_sorted, keep, n_kept


# ----------------------------------------------------------------------------- training-side box ops
def anchor_targets(bbox: torch.Tensor, anchor: torch.Tensor, n_pos: int, n_sample: int, pos_iou_thresh: float,
                   neg_iou_thresh: float):
    """anchor [A,4], bbox [G,4] -> (loc [A,4] f32, label [A] int64, argmax [A] int32) (tsod_anchor_targets_f32)."""
    require_cuda(anchor, "anchor_targets")
    anchor, bbox = anchor.contiguous(), bbox.to(anchor.device, torch.float32).contiguous()
    A, G = anchor.shape[0], bbox.shape[0]
    dev = anchor.device
    loc = torch.empty((A, 4), dtype=torch.float32, device=dev)
    label = torch.empty((A,), dtype=torch.int64, device=dev)
    argmax = torch.empty((A,), dtype=torch.int32, device=dev)
    ws_bytes = lib().tsod_anchor_targets_workspace_bytes(A, G)
    ws = ARENA.get(dev, ws_bytes)
    check(lib().tsod_anchor_targets_f32(ptr(anchor), A, ptr(bbox) if G else None, G, float(pos_iou_thresh),
                                        float(neg_iou_thresh), int(n_pos), int(n_sample), ptr(loc), ptr(label), ptr(argmax),
                                        ptr(ws), ws_bytes, stream_ptr()), "anchor_targets")
    return loc, label, argmax


def proposal_targets(roi: torch.Tensor, bbox: torch.Tensor, label: torch.Tensor, n_sample: int, pos_per_image: int,
                     pos_iou_thresh: float, neg_iou_thresh_high: float, neg_iou_thresh_low: float):
    """roi [R,4], bbox [G,4], label [G] int64 -> (sample_roi [n_sample,4], gt_roi_loc [n_sample,4], gt_roi_label [n_sample]
    int64, counts [4] int32 = (rows kept, positives, negatives, status)) (tsod_proposal_targets_f32)."""
    require_cuda(roi, "proposal_targets")
    dev = roi.device
    roi, bbox = roi.contiguous(), bbox.to(dev, torch.float32).contiguous()
    label = label.to(dev, torch.int64).contiguous()
    R, G = roi.shape[0], bbox.shape[0]
    sample_roi = torch.empty((n_sample, 4), dtype=torch.float32, device=dev)
    gt_roi_loc = torch.empty((n_sample, 4), dtype=torch.float32, device=dev)
    gt_roi_label = torch.empty((n_sample,), dtype=torch.int64, device=dev)
    counts = torch.empty((4,), dtype=torch.int32, device=dev)
    ws_bytes = lib().tsod_proposal_targets_workspace_bytes(R, G, n_sample)
    ws = ARENA.get(dev, ws_bytes)
    check(lib().tsod_proposal_targets_f32(ptr(roi) if R else None, R, ptr(bbox) if G else None, G, ptr(label) if G else None,
                                          int(n_sample), int(pos_per_image), float(pos_iou_thresh), float(neg_iou_thresh_high),
                                          float(neg_iou_thresh_low), ptr(sample_roi), ptr(gt_roi_loc), ptr(gt_roi_label),
                                          ptr(counts), ptr(ws), ws_bytes, stream_ptr()), "proposal_targets")
    return sample_roi, gt_roi_loc, gt_roi_label, counts


# ----------------------------------------------------------------------------- input step
_RESIZE_TABLES: dict = {}


def resize_tables(n_in: int, n_out: int, device):
    """(first [n_out] i32, count [n_out] i32, weights [n_out,taps] f32) of one axis on ``device``; computed by the
    library's host function once per (n_in, n_out, device)."""
    import numpy as np
    key = (int(n_in), int(n_out), torch.device(device))
    t = _RESIZE_TABLES.get(key)
    if t is None:
        L = lib()
        taps = L.tsod_resize_aa_taps(n_in, n_out)
        first = np.zeros(n_out, np.int32)
        count = np.zeros(n_out, np.int32)
        w = np.zeros((n_out, taps), np.float32)
        check(L.tsod_resize_aa_tables_f32(n_in, n_out, first.ctypes.data, count.ctypes.data, w.ctypes.data), "resize_tables")
        t = _RESIZE_TABLES[key] = tuple(torch.from_numpy(a).to(device) for a in (first, count, w))
    return t


def resize_bilinear_aa(img: torch.Tensor, OH: int, OW: int, layout: str = "nhwc4", mul: float = 1.0, out=None):
    """u8 [H,W,C<=4] CUDA image -> antialiased-bilinear resized f32 image: ``layout="nhwc4"`` -> [OH,OW,4] (extra channels
    zero), ``"nchw"`` -> [C,OH,OW]  (dataset/transform.py:14-17 on the GPU)."""
    if not isinstance(img, torch.Tensor) or not img.is_cuda or img.dtype != torch.uint8 or img.dim() != 3:
        raise TsodError("resize_bilinear_aa: a u8 [H,W,C] CUDA/ROCm tensor is required")
    H, W, C = img.shape
    if not 1 <= C <= 4 or img.stride(2) != 1 or img.stride(1) != C:
        raise TsodError("resize_bilinear_aa: pixels must be interleaved and contiguous along a row")
    dev = img.device
    yf, yc, yw = resize_tables(H, OH, dev)
    xf, xc, xw = resize_tables(W, OW, dev)
    if layout == "nhwc4":
        shape, strides, c_out = (OH, OW, 4), (4 * OW, 4, 1), 4
    elif layout == "nchw":
        shape, strides, c_out = (C, OH, OW), (OW, 1, OH * OW), C
    else:
        raise ValueError(layout)
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=dev)
    if tuple(out.shape) != shape or out.dtype != torch.float32 or not out.is_contiguous() or out.device != dev:
        raise TsodError(f"resize_bilinear_aa: out must be a contiguous f32 {shape} tensor on {dev}")
    check(lib().tsod_resize_bilinear_aa_u8_f32(ptr(img), H, W, C, img.stride(0), ptr(yf), ptr(yc), ptr(yw), ptr(xf), ptr(xc),
                                               ptr(xw), OH, OW, float(mul), ptr(out), strides[0], strides[1], strides[2],
                                               c_out, stream_ptr()), "resize_bilinear_aa")
    return out
