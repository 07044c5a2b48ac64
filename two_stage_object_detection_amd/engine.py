"""Host-side execution engine: weight folding/packing, activation buffers, and a static launch plan.

A *plan* is the list of C-ABI calls (pre-bound ctypes arguments) that one forward of a module makes
for one input geometry.  Building it does all host work once (shape arithmetic, descriptor structs,
buffer assignment, tile selection); running it is a loop of ``libtsod`` launches on the current
HIP stream with no host synchronisation, so a plan can be captured into a HIP graph
(``Plan.capture`` uses ``torch.cuda.CUDAGraph`` purely as the stream-capture plumbing).

Data layout in HBM: every activation is NHWC f32, channel pitch a multiple of 4, 16-byte aligned.
"""
from __future__ import annotations

import os
import re

from collections import OrderedDict
from ctypes import byref, c_int32
from typing import Callable, Sequence

import torch

from . import _ffi
from ._ffi import (ACT_NONE, ACT_PRELU, ACT_RELU, ACT_RELU6, TILE_NAMES, ConvDesc, TsodError, check, lib,
                   make_conv_desc, ptr, stream_ptr)


# --------------------------------------------------------------------------- weights
def fold_bn(bn: torch.nn.BatchNorm2d):
    """eval-mode BatchNorm as y = x*scale + shift (f64 on the host, rounded once to f32)."""
    var = bn.running_var.detach().double().cpu()
    mean = bn.running_mean.detach().double().cpu()
    gamma = bn.weight.detach().double().cpu() if bn.affine else torch.ones_like(var)
    beta = bn.bias.detach().double().cpu() if bn.affine else torch.zeros_like(var)
    scale = gamma / torch.sqrt(var + bn.eps)
    shift = beta - mean * scale
    return scale.float(), shift.float()


class PackedConv:
    """A dense conv (+ folded BN or bias, + activation) in the layout tsod_conv2d_f32 consumes."""

    def __init__(self, weight: torch.Tensor, device, *, bn=None, bias=None, stride=1, pad=0, act=ACT_NONE, slope=0.0,
                 cin_pad=None, kw_pad=None, cout_pad=None):
        from . import hip_ops
        w = weight.detach().to(device=device, dtype=torch.float32)
        self.cout, self.cin_src, self.kh, self.kw_logical = w.shape
        self.w = hip_ops.pack_conv_weight(w, cin_pad, kw_pad)
        self.cin, self.kw = self.w.shape[3], self.w.shape[2]
        self.stride, self.pad, self.act, self.slope = stride, pad, act, float(slope)
        scale = shift = None
        if bn is not None:
            scale, shift = fold_bn(bn)
        elif bias is not None:
            shift = bias.detach().float().cpu()
        self.scale = None if scale is None else scale.to(device)
        self.shift = None if shift is None else shift.to(device)

    def out_hw(self, H, W):
        return ((H + 2 * self.pad - self.kh) // self.stride + 1, (W + 2 * self.pad - self.kw_logical) // self.stride + 1)


class FusedShortcutConv:
    """The last 1x1 conv of a residual block and its projection shortcut as ONE stacked-K GEMM (tsod_conv2d_dual_f32):
        out = act( [y | x(strided)] . [W3 * s3 | Wd * sd]^T + (b3 + bd) )
    (reference: out = bn3(conv3(y)); identity = bn_d(conv_d(x)); out += identity; relu - models/resnet.py:70-76 with the
    downsample of :114-116).  Both BN scales are folded into the stacked weights in f64 and rounded once to f32 (the separate
    form rounds the two scaled sums and their add instead: differences ~1e-7 relative, inside the feature bar)."""

    def __init__(self, conv3_w, bn3, ds_w, ds_bn, ds_stride, device, act, slope):
        s3, b3 = fold_bn(bn3)
        sd, bd = fold_bn(ds_bn)
        w3 = conv3_w.detach().double().cpu().flatten(1) * s3.double().view(-1, 1)          # [Cout, C1]
        wd = ds_w.detach().double().cpu().flatten(1) * sd.double().view(-1, 1)             # [Cout, C2]
        self.w = torch.cat([w3, wd], dim=1).float().contiguous().to(device)                # [Cout, C1 + C2]
        self.cout, self.cin, self.c2 = self.w.shape[0], w3.shape[1], wd.shape[1]
        self.cin_src, self.kh, self.kw, self.kw_logical = self.cin + self.c2, 1, 1, 1      # (FLOP accounting: both GEMMs)
        self.stride, self.pad, self.act, self.slope, self.stride2 = 1, 0, act, float(slope), int(ds_stride)
        self.scale = None
        self.shift = (b3.double() + bd.double()).float().to(device)

    def out_hw(self, H, W):
        return H, W


class FusedBottleneckWeights:
    """A bottleneck with 64 mid channels at stride 1 (models/resnet.py:57-76) packed for tsod_bottleneck_fp16x2: the three convs'
    weights as one stream in consumption order, the three folded BatchNorms as one vector, the shared PReLU slope.  With a
    projection shortcut (``blk.downsample``: :114-116, layer1's first block) conv3 and the shortcut are ONE stacked-K matrix
    [W3 s3 | Wd sd] (both BatchNorm scales folded in f64, rounded once - as FusedShortcutConv does for the three-launch form),
    the vector's s3 is all ones and its b3 the two shifts added up."""

    def __init__(self, blk, device):
        from . import hip_ops
        c1, c2, c3 = blk.conv1, blk.conv2, blk.conv3
        self.cin, self.cmid, self.cout = c1.in_channels, c1.out_channels, c3.out_channels
        self.projection = blk.downsample is not None
        w1 = c1.weight.detach().float().to(device).view(self.cmid, self.cin)
        w2 = hip_ops.pack_conv_weight(c2.weight.detach().float().to(device))            # [64, 3, 3, 64]
        bn = []
        for m in (blk.bn1, blk.bn2):
            bn.extend(fold_bn(m))
        s3, b3 = fold_bn(blk.bn3)
        if self.projection:
            sd, bd = fold_bn(blk.downsample[1])
            w3 = c3.weight.detach().double().cpu().flatten(1) * s3.double().view(-1, 1)                     # [Cout, 64]
            wd = blk.downsample[0].weight.detach().double().cpu().flatten(1) * sd.double().view(-1, 1)      # [Cout, Cin]
            w3 = torch.cat([w3, wd], dim=1).float().contiguous().to(device)
            bn.extend([torch.ones_like(s3), (b3.double() + bd.double()).float()])
        else:
            w3 = c3.weight.detach().float().to(device).view(self.cout, self.cmid)
            bn.extend([s3, b3])
        self.stream, self.w_exps = hip_ops.pack_bottleneck_wstream(w1, w2, w3, projection=self.projection)
        self.bn = torch.cat(bn).to(device)
        self.slope = prelu_slope(blk.relu)
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(self.stream.device).synchronize()


class FusedStemWeights:
    """ResNet's conv1 + bn1 + PReLU (models/resnet.py:136-138) packed for tsod_stem_fp16x2: the 7x7 weights as per-lane MFMA
    fragments of two fp16 pieces, the folded BatchNorm as one vector, the PReLU slope."""

    def __init__(self, conv, bn, relu, device):
        from . import hip_ops
        self.wfrag, self.w_exp = hip_ops.pack_stem_wfrag(conv.weight.detach().float().to(device))
        self.bn = torch.cat(fold_bn(bn)).to(device)
        self.slope = prelu_slope(relu)
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(self.wfrag.device).synchronize()


class FusedStep:
    """One tsod_bottleneck_fp16x2 launch of a plan, with what the timing / roofline code asks of a ConvStep."""
    precision = _ffi.PREC_FP16X2

    def __init__(self, name, fn, args, desc, flops, algorithmic_bytes, x=None):
        self.name, self.fn, self.args, self.desc, self.flops, self.algorithmic_bytes = name, fn, args, desc, flops, algorithmic_bytes
        self.x, self.x2 = x, None            # the activation input (None: the stem, whose pixel scale is each tile's own)


def step_precision(st) -> int:
    """Arithmetic of a matrix launch of a plan (ConvStep or FusedStep): _ffi.PREC_*"""
    return int(st.precision) if isinstance(st, FusedStep) else int(st.desc.precision)


def weights_bf16x3(pc) -> torch.Tensor:
    """The pre-split bf16x3 image of a packed layer's weights (PackedConv or a compatible holder), made on first use and
    kept beside the f32 weights."""
    w3 = getattr(pc, "w3", None)
    if w3 is None:
        from . import hip_ops
        w3 = pc.w3 = hip_ops.pack_conv_weight_bf16x3(pc.w)
        # the image is shared by every plan, in-flight slot and stream of the owner: it must be complete before a consumer on
        # ANOTHER stream can see the attribute (one host wait per layer, at build time; illegal - and never needed, the
        # warm-up forwards of a capture run first - while the stream is being captured)
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(w3.device).synchronize()
    return w3


FP16X2_A_SCALE_EXP = 4     # static default (no range words): activations are split as 2^4 * x: |x| < 4094 (65504 / 16) is its range


def new_range_flag(device) -> torch.Tensor:
    """One int32 in device memory: what desc.range_flag points at.  (A word in pinned host memory, readable without a device round
    trip, was tried and dropped: every reporting wave's atomic then crosses PCIe, and tuning launches on stale buffers - thousands
    of reporting waves per launch - ran 100x slower and took the fp16x2 candidates out of the table.)"""
    return torch.zeros(1, dtype=torch.int32, device=device)


def fp16x2_activation_exp(absmax: float, headroom_bits: int = 4) -> int:
    """The fp16x2 activation exponent e for a tensor of that abs-max: 2^e * absmax <= 65504 / 2^headroom_bits, clamped to
    [-24, 8] (8 for an all-zero or non-finite measurement: the guard decides at run time)."""
    import math
    if absmax == 0.0 or not math.isfinite(absmax):
        return 8
    return max(-24, min(8, int(math.floor(math.log2(65504.0 / (absmax * (1 << headroom_bits)))))))


def weights_fp16x2(pc):
    """(fp16x2 image, w_scale_exp) of a packed layer's weights (include/tsod.h TSOD_PREC_FP16X2),
    made on first use and kept beside the f32 weights; the exponent brings max |w| just below 2^14."""
    hit = getattr(pc, "w2", None)
    if hit is None:
        from . import hip_ops
        e = hip_ops.fp16x2_weight_scale_exp(pc.w)
        hit = pc.w2 = (hip_ops.pack_conv_weight_fp16x2(pc.w, e), e)
        if not torch.cuda.is_current_stream_capturing():
            torch.cuda.current_stream(pc.w.device).synchronize()
    return hit


def prelu_slope(m: torch.nn.PReLU) -> float:
    if m.weight.numel() != 1:
        raise TsodError("only single-parameter nn.PReLU is supported (what the reference uses)")
    return float(m.weight.detach().cpu().item())


# --------------------------------------------------------------------------- buffers
class BufferPool:
    """Activation buffers reused across layers of one plan (liveness is explicit: alloc / release)."""

    def __init__(self, device):
        self.device = device
        self._free: dict[int, list[torch.Tensor]] = {}
        self.total_bytes = 0
        self.on_alloc = None                 # called with every tensor handed out (Plan: a fresh set of range words per tensor)

    def alloc(self, shape: Sequence[int], dtype=torch.float32) -> torch.Tensor:
        n = 1
        for s in shape:
            n *= int(s)
        key = (n, dtype)
        lst = self._free.get(key)
        if lst:
            t = lst.pop().view(*shape)
        else:
            self.total_bytes += n * (4 if dtype in (torch.float32, torch.int32) else 8)
            t = torch.empty(n, dtype=dtype, device=self.device).view(*shape)
        if self.on_alloc is not None:
            self.on_alloc(t)
        return t

    def release(self, t: torch.Tensor) -> None:
        self._free.setdefault((t.numel(), t.dtype), []).append(t.reshape(-1))


def _merge_adjacent(segs):
    """Fuse channel segments that are contiguous in the pixel (same K order, fewer descriptor entries)."""
    out = []
    for off, ln in segs:
        if out and out[-1][0] + out[-1][1] == off:
            out[-1] = (out[-1][0], out[-1][1] + ln)
        else:
            out.append((off, ln))
    return out


# --------------------------------------------------------------------------- plan
class ConvStep:
    __slots__ = ("desc", "args", "name", "flops", "ws_bytes", "pc", "fn", "w_index", "ws_index", "range_flag", "x", "x2", "exps")

    def choose(self, tile: int, split_k: int, precision: int):
        """Pin (tile, K-slice schedule, arithmetic); the weight argument follows the arithmetic (f32 or pre-split bf16x3)."""
        d = self.desc
        d.tile, d.split_k, d.precision = int(tile), int(split_k), int(precision)
        if precision == _ffi.PREC_FP16X2:
            w2, e = weights_fp16x2(self.pc)
            d.a_scale_exp, d.w_scale_exp = int(self.exps.get(self.name, FP16X2_A_SCALE_EXP)), int(e)
            d.range_flag = ptr(self.range_flag)                  # the owner's word: PlanOwner.raise_if_error reads it
            self.args[self.w_index] = ptr(w2)
        else:
            self.args[self.w_index] = ptr(weights_bf16x3(self.pc)) if precision == _ffi.PREC_BF16X3 else ptr(self.pc.w)


class Plan:
    def __init__(self, device, packed: dict | None = None):
        self.device = torch.device(device)
        self.pool = BufferPool(self.device)
        self._packed = packed if packed is not None else {}   # the owner's packed-weight cache (shared by all its plans)
        self.precision = 0                   # default arithmetic of the plan's dense convs (_ffi.PREC_F32 / PREC_BF16X3)
        self._retired: list = []             # outgrown workspaces: graphs captured earlier still hold their pointers
        self.steps: list[list] = []          # [cfunc, [args...]]
        self.conv_steps: list[ConvStep] = []
        self.fused_steps: list[FusedStep] = []                   # whole-bottleneck launches (tsod_bottleneck_fp16x2)
        self.gemm_steps: list = []           # every matrix launch in forward order: ConvStep | FusedStep (timing, roofline)
        self.stem_step = None                # the one-launch stem (tsod_stem_fp16x2), whose input pointer stage_input binds per forward
        self._bound_input = None
        self.keep: list = []                 # keeps descriptors / tensors alive
        self._ws_slots: list[tuple[list, int, int, int]] = []   # (args, ptr index, size index, bytes)
        self.workspace: torch.Tensor | None = None
        self.graph = None
        # fp16x2 layers OR 1 into this word when a launch ends with non-finite accumulators (include/tsod.h: range_flag);
        # a PlanOwner hands its plans one word per (device, in-flight slot) out of a tensor of its own (see _cached_plan): the word
        # survives plan eviction, and checking all of an owner's words costs one small read per device
        self.range_flag = new_range_flag(self.device)
        self.a_exps: dict = {}               # layer name -> fp16x2 activation exponent (calibrate_fp16x2); the owner shares ONE dict among its plans
        self.flops = 0
        self.on_calibrated = None
        # Range words (include/tsod.h): every tensor the pool hands out gets a fresh set; producers add their outputs' abs-max,
        # fp16x2 convs take their activation scale from the words of their input(s) - per forward, inside the launches.
        self.dynamic_scale = bool(self.DEFAULT_DYNAMIC_SCALE)
        self.amax = torch.zeros(self.AMAX_SLOTS * _ffi.AMAX_BYTES // 4, dtype=torch.int32, device=self.device)
        self._amax_slot: dict[int, int] = {}                     # storage pointer of a tensor -> its current slot
        self._amax_used = 0
        self.pool.on_alloc = self._new_amax_slot

    # -- range words ------------------------------------------------------------------------
    # False: plans built from now on use the static fp16x2 exponents (calibrate_fp16x2).  TSOD_NO_RANGE_WORDS=1 switches the words off
    # for A/B timing (scripts/ab_env.sh); the static 2^4 exponent covers the synthetic detector's activations
    DEFAULT_DYNAMIC_SCALE = os.environ.get("TSOD_NO_RANGE_WORDS", "0") in ("", "0")
    AMAX_SLOTS = 320                         # tensors per forward (ResNet-101: ~110, HarDNet-85: ~200); 4 KB each

    def _new_amax_slot(self, t: torch.Tensor) -> None:
        if self._amax_used >= self.AMAX_SLOTS:
            raise TsodError("plan: more activation tensors than range-word slots")
        self._amax_slot[t.untyped_storage().data_ptr()] = self._amax_used
        self._amax_used += 1

    def amax_ptr(self, t) -> int:
        """Device pointer of the range words of the tensor that lives in ``t``'s storage right now (0: none / switched off)."""
        if t is None or not self.dynamic_scale:
            return 0
        slot = self._amax_slot.get(t.untyped_storage().data_ptr())
        return 0 if slot is None else self.amax.data_ptr() + slot * _ffi.AMAX_BYTES

    def alias_amax(self, t: torch.Tensor, src: torch.Tensor) -> None:
        """``t`` is a max / mean pooling of ``src``: its abs-max is bounded by src's, it shares src's words."""
        slot = self._amax_slot.get(src.untyped_storage().data_ptr())
        if slot is not None:
            self._amax_slot[t.untyped_storage().data_ptr()] = slot

    def clear_range_flag(self) -> None:
        """Forget what launches issued so far reported (tuning / timing launches run on whatever the pooled buffers hold); waits
        for launches on other streams first."""
        torch.cuda.synchronize(self.device)
        with torch.inference_mode():
            self.range_flag.zero_()

    def _refresh_amax_for(self, st) -> None:
        """Isolated timing of one layer (autotune's first look): the pooled buffer its input lives in may hold a LATER tensor of
        the forward by now, or what earlier candidates left - make the input's range words describe the bytes that are there, so
        that an fp16x2 candidate is timed on in-range operands like the ones it will meet (non-finite accumulators cost power
        and reports, and would pick the table for the wrong reasons)."""
        for t, a in ((st.x, getattr(st.desc, "amax_in", None)), (st.x2, getattr(st.desc, "amax_in2", None))):
            if t is not None and a:
                check(lib().tsod_amax_reset(a, 1, stream_ptr()), "amax_reset")
                check(lib().tsod_absmax_f32(ptr(t), t.numel(), a, stream_ptr()), "absmax")

    def reset_amax(self) -> None:
        """Zero the words in use (stream-ordered, capturable): first thing of every forward, before the input is staged."""
        if self.dynamic_scale and self._amax_used:
            check(lib().tsod_amax_reset(ptr(self.amax), self._amax_used, stream_ptr()), "amax_reset")

    # -- building ---------------------------------------------------------------------------
    def packed(self, key, make: Callable):
        """The packed form of a layer's weights, built once per (layer, device) and shared by every plan (input
        geometry, in-flight slot) of the owning module."""
        key = (key, self.device)
        obj = self._packed.get(key)
        if obj is None:
            obj = self._packed[key] = make()
        return obj

    def call(self, fn, *args, keep=()):
        self.steps.append([fn, list(args)])
        self.keep.extend(keep)
        return self.steps[-1]

    # the conv kernel addresses every tensor through 32-bit byte offsets (buffer descriptors): one launch may
    # touch at most this many bytes of any single tensor; larger batches are cut into image groups
    MAX_TENSOR_BYTES = 0xF0000000

    def conv(self, pc: PackedConv, x: torch.Tensor, out: torch.Tensor, *, segs=None, out_off=0, residual=None,
             name="conv", tile=0, split_k=0, precision=None, x2=None, stride2=1):
        """x [N,H,W,P] -> out [N,OH,OW,Pout] (channel slice [out_off, out_off+Cout)).  ``x2`` [N,H2,W2,P2]: second source of
        a stacked-weight 1x1 conv (``pc`` = FusedShortcutConv): its last ``pc.c2`` K columns read pixel (oh*stride2, ow*stride2)."""
        N, H, W, P = x.shape
        OH, OW = pc.out_hw(H, W)
        assert tuple(out.shape[:3]) == (N, OH, OW), (out.shape, (N, OH, OW))
        precision = self.precision if precision is None else precision
        per_img = max(H * W * P, OH * OW * out.shape[3], 0 if residual is None else OH * OW * residual.shape[3],
                      0 if x2 is None else x2.shape[1] * x2.shape[2] * x2.shape[3]) * 4
        if N > 1 and per_img * N >= self.MAX_TENSOR_BYTES:
            group = max(1, self.MAX_TENSOR_BYTES // per_img)
            for n0 in range(0, N, group):
                n1 = min(N, n0 + group)
                self.conv(pc, x[n0:n1], out[n0:n1], segs=segs, out_off=out_off,
                          residual=None if residual is None else residual[n0:n1], name=f"{name}[{n0}:{n1}]", tile=tile,
                          split_k=split_k, precision=precision, x2=None if x2 is None else x2[n0:n1], stride2=stride2)
            return out
        segs = [(0, pc.cin)] if segs is None else _merge_adjacent(segs)
        d = make_conv_desc(N=N, H=H, W=W, in_pitch=P, segs=segs, Cout=pc.cout, out_pitch=out.shape[3], out_off=out_off,
                           KH=pc.kh, KW=pc.kw, stride=pc.stride, pad_h=pc.pad, pad_w=pc.pad, OH=OH, OW=OW, act=pc.act,
                           slope=pc.slope, res_pitch=0 if residual is None else residual.shape[3], res_off=0,
                           tile=tile, split_k=split_k, precision=precision,
                           src2=None if x2 is None else (pc.c2, x2.shape[3], 0, stride2, x2.shape[1], x2.shape[2]))
        d.amax_in, d.amax_in2, d.amax_out = self.amax_ptr(x) or None, self.amax_ptr(x2) or None, self.amax_ptr(out) or None
        args = [byref(d), ptr(x), ptr(weights_bf16x3(pc)) if precision == _ffi.PREC_BF16X3 else ptr(pc.w), ptr(pc.scale),
                ptr(pc.shift), ptr(residual), ptr(out), 0, 0]
        if x2 is not None:
            args.insert(2, ptr(x2))                               # tsod_conv2d_dual_f32(desc, in, in2, w, ...)
        self.steps.append([lib().tsod_conv2d_dual_f32 if x2 is not None else lib().tsod_conv2d_f32, args])
        st = ConvStep()
        st.desc, st.args, st.name, st.pc = d, args, name, pc
        st.range_flag = self.range_flag
        st.x, st.x2, st.exps = x, x2, self.a_exps                 # (the inputs: Plan.calibrate_fp16x2 measures their range)
        st.fn = self.steps[-1][0]
        st.w_index, st.ws_index = (3, 8) if x2 is not None else (2, 7)
        if precision == _ffi.PREC_FP16X2:                         # (weight image, scale exponents, range flag: choose() sets them)
            st.choose(tile, split_k, precision)
        st.flops = 2 * N * OH * OW * getattr(pc, "cout_real", pc.cout) * pc.kh * pc.kw_logical * pc.cin_src   # algorithmic
        st.ws_bytes = 0
        self.conv_steps.append(st)
        self.gemm_steps.append(st)
        self.flops += st.flops
        self.keep.extend([d, x, out, pc, residual])
        return out

    def bottleneck(self, fb: "FusedBottleneckWeights", x: torch.Tensor, out: torch.Tensor, name: str):
        """x [N,H,W,Cin] -> out [N,H,W,Cout]: conv1 + conv2 + conv3 + identity of a bottleneck as ONE launch (fp16x2; the input's
        scale from its range words when the plan keeps them, the intermediates' from each tile's own abs-max)."""
        N, H, W, P = x.shape
        proj = bool(getattr(fb, "projection", False))
        assert tuple(out.shape[:3]) == (N, H, W) and (proj or fb.cin == fb.cout)
        d = _ffi.BottleneckDesc()
        d.N, d.H, d.W, d.Cin, d.in_pitch, d.Cmid, d.Cout, d.out_pitch = N, H, W, fb.cin, P, fb.cmid, fb.cout, out.shape[3]
        d.projection = 1 if proj else 0
        d.slope = float(fb.slope)
        for k in range(3):
            d.w_exp[k] = int(fb.w_exps[k])
        d.a_scale_exp = int(self.a_exps.get(name, FP16X2_A_SCALE_EXP))
        d.range_flag = ptr(self.range_flag)
        d.amax_in, d.amax_out = self.amax_ptr(x) or None, self.amax_ptr(out) or None
        args = [byref(d), ptr(x), ptr(fb.stream), ptr(fb.bn), ptr(out)]
        self.steps.append([lib().tsod_bottleneck_fp16x2, args])
        px = N * H * W
        flops = 2 * px * (fb.cin * fb.cmid + 9 * fb.cmid * fb.cmid + fb.cmid * fb.cout + (fb.cin * fb.cout if proj else 0))
        # what the block must move however it is computed: x in, out out, the weights (the residual / shortcut input is x again: counted once)
        alg = 4 * (px * fb.cin + px * fb.cout + fb.cin * fb.cmid + 9 * fb.cmid * fb.cmid + fb.cmid * fb.cout + (fb.cin * fb.cout if proj else 0))
        st = FusedStep(name, self.steps[-1][0], args, d, flops, alg, x=x)
        self.fused_steps.append(st)
        self.gemm_steps.append(st)
        self.flops += flops
        self.keep.extend([d, x, out, fb])
        return out

    def stem(self, fs: "FusedStemWeights", N: int, H: int, W: int, out: torch.Tensor, name: str = "stem"):
        """images [N,3,H,W] (NCHW) or NHWC4Images -> out [N,PH,PW,64]: conv1 + bn1 + PReLU + max pool as ONE launch
        (tsod_stem_fp16x2; the pixel scale is each tile's own).  The images are whatever ``stage_input`` bound last: the launch
        reads them where the caller holds them - no layout pass, no copy."""
        d = _ffi.StemDesc()
        d.N, d.H, d.W, d.in_layout, d.out_pitch = N, H, W, _ffi.STEM_NHWC4, out.shape[3]
        d.slope, d.w_exp = float(fs.slope), int(fs.w_exp)
        d.range_flag = ptr(self.range_flag)
        d.amax_out = self.amax_ptr(out) or None
        args = [byref(d), ptr(self.input_nhwc), ptr(fs.wfrag), ptr(fs.bn), ptr(out)]
        self.steps.append([lib().tsod_stem_fp16x2, args])
        oh, ow = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        flops = 2 * N * oh * ow * 64 * 147
        alg = 4 * (N * H * W * 3 + out.shape[0] * out.shape[1] * out.shape[2] * 64 + 64 * 147)
        st = FusedStep(name, self.steps[-1][0], args, d, flops, alg)
        self.stem_step = st
        self.fused_steps.append(st)
        self.gemm_steps.append(st)
        self.flops += flops
        self.keep.extend([d, out, fs])
        return out

    def bind_input(self, x) -> None:
        """Point the one-launch stem at the images of the next forward (an NCHW tensor or NHWC4Images; kept alive until the next
        call).  A HIP graph captured from the plan replays the pointer it was captured with."""
        st = self.stem_step
        nhwc4 = isinstance(x, _ffi.NHWC4Images)
        t = x.data if nhwc4 else x
        d = st.desc
        want = (d.N, d.H, d.W, 4) if nhwc4 else (d.N, 3, d.H, d.W)
        if t.dtype != torch.float32 or tuple(t.shape) != want or t.device != self.device:
            raise TsodError(f"the one-launch stem expects float32 {want} on {self.device}, got {t.dtype} {tuple(t.shape)} on {t.device}")
        t = t.contiguous()                   # (a copy only for a strided view; a batch slice x[i:i+1] is contiguous as it is)
        if t.data_ptr() % (16 if nhwc4 else 4):
            t = t.clone()
        st.desc.in_layout = _ffi.STEM_NHWC4 if nhwc4 else _ffi.STEM_NCHW
        st.args[1] = ptr(t)
        self._bound_input = t

    def finalize(self):
        """Size the shared K-slice workspace (stream order makes sharing safe) and bind it.  Its head holds the arrival
        tickets of the K-sliced tiles, which every launch expects to find zero and leaves zero: zero-filled once here."""
        need = 256
        for st in self.conv_steps:
            st.ws_bytes = lib().tsod_conv2d_workspace_bytes(byref(st.desc))
            need = max(need, st.ws_bytes)
        if self.workspace is None or self.workspace.numel() < need:
            if self.workspace is not None:
                # a HIP graph captured from this plan earlier has the old pointer baked in: the buffer must outlive it
                self._retired.append(self.workspace)
            self.workspace = torch.zeros(need, dtype=torch.uint8, device=self.device)
        for st in self.conv_steps:
            st.args[st.ws_index] = ptr(self.workspace)
            st.args[st.ws_index + 1] = self.workspace.numel()
        self.graph = None
        return self

    # -- the fp16x2 arithmetic's activation scale -----------------------------------------------
    def calibrate_fp16x2(self, x, headroom_bits: int = 4):
        """Set every conv step's fp16x2 activation exponent from the range its input REALLY has on ``x`` (one forward, launch by
        launch, abs-max of each conv's input(s) read right before it runs): 2^e * absmax <= 65504 / 2^headroom_bits, e clamped
        to [-24, 8].  Without it every layer uses 2^4 (|x| < 4094); with it a model whose activations are larger (or much
        smaller) than the synthetic detector's gets exponents that fit, with 16x headroom for other inputs - and the range guard
        (include/tsod.h: range_flag) still watches every launch.  Returns {layer name: (absmax, exponent)}."""
        import math
        stage_input(self, x)
        by_args = {id(st.args): st for st in self.conv_steps}
        by_args.update({id(st.args): st for st in self.fused_steps if st.x is not None})     # (one-launch bottlenecks: the static path's exponent of x)
        s = stream_ptr()
        seen = {}
        for fn, args in self.steps:
            st = by_args.get(id(args))
            if st is not None:
                m = float(st.x.abs().max())
                if st.x2 is not None:
                    m = max(m, float(st.x2.abs().max()))
                e = fp16x2_activation_exp(m, headroom_bits)
                e = min(e, self.a_exps.get(st.name, e)) if st.name in seen else e      # (image groups of one layer: the smallest)
                self.a_exps[st.name] = e
                if step_precision(st) == _ffi.PREC_FP16X2:
                    st.desc.a_scale_exp = e
                seen[st.name] = (m, e)
            rc = fn(*args, s)
            if rc != 0:
                check(rc, getattr(fn, "__name__", "tsod call"))
        self.clear_range_flag()
        self.graph = None                                        # (a captured graph has the old exponents baked in)
        if self.on_calibrated is not None:
            self.on_calibrated()                                 # (the owner: graphs captured at detector level are stale too)
        return seen

    # -- running ----------------------------------------------------------------------------
    def launch(self):
        s = stream_ptr()
        for fn, args in self.steps:
            rc = fn(*args, s)
            if rc != 0:
                check(rc, getattr(fn, "__name__", "tsod call"))

    def run(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            self.launch()

    def sequence_time(self, reps: int = 10) -> float:
        """HIP-event time (ms) of ONE pass over the plan's matrix launches (convs and fused bottlenecks) in forward order, back to
        back on the current stream, averaged over ``reps`` passes: every layer finds its input where the previous launch left it
        and its weights as cold as a forward leaves them (what bench.py's roofline and FasterRCNN.tune's structure choice use)."""
        s = stream_ptr()
        for st in self.gemm_steps:
            st.fn(*st.args, s)
        # the pooled buffers now hold what a pass leaves; inputs that no matrix launch produces (a max pool's or a depthwise conv's
        # output) may be a LATER tensor of the forward: make every input's range words an upper bound of the bytes that are there
        # (the producers of the pass add their own abs-max on top), so that no fp16x2 launch is timed on out-of-range operands
        if self.dynamic_scale:
            for st in self.gemm_steps:
                self._refresh_amax_for(st)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            for st in self.gemm_steps:
                st.fn(*st.args, s)
        e1.record()
        e1.synchronize()
        self.clear_range_flag()      # (these launches ran on whatever the pooled buffers held, not on a forward's activations)
        return e0.elapsed_time(e1) / reps

    def forward_time(self, reps: int = 5) -> float:
        """HIP-event time (ms) of ONE pass over EVERY launch of the plan (``launch()``: range-word reset, layout pass, max pool and
        depthwise launches included) on the input bound last, averaged over ``reps`` passes issued back to back: real activations
        at every layer, which a pass over the matrix launches alone does not have.  What FasterRCNN.tune compares structures by."""
        self.launch()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            self.launch()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / reps

    def capture(self):
        """Record the plan into a HIP graph (one graph launch per forward afterwards)."""
        torch.cuda.synchronize(self.device)
        side = torch.cuda.Stream(self.device)
        with torch.cuda.stream(side):
            self.launch()                      # warm-up outside capture
        side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
            self.launch()
        self.graph = g
        return self

    # -- tile autotuning ----------------------------------------------------------------------
    def autotune(self, reps: int = 3, verbose: bool = False, splits=None, concurrent: int = 1, precisions=None,
                 in_sequence: int = 0, keep_shortlist: int = 0):
        """Measure every (tile, split_k) candidate of every conv step on the real buffers with HIP
        events and keep the fastest.  Purely a speed choice: every candidate computes the same sums
        in the same k order per slab; only slab boundaries move.
        ``concurrent`` > 1 times each candidate as that many copies in flight on separate streams
        (per-copy time = elapsed / copies): the objective of a server that overlaps requests, where a
        schedule that fills the whole chip for one launch is not automatically the cheapest.
        ``precisions``: the arithmetics to choose from per layer (default: only what each step has now; (0, 1) lets the
        f32-MFMA and the bf16x3 form of a layer compete - both are f32-accurate, see include/tsod.h).
        ``in_sequence`` = n > 0 (serial objective only): a second look at the n fastest candidates of every layer INSIDE the
        forward - the whole conv sequence is launched in order and only the layer under test is bracketed by HIP events, so
        the candidate runs on the cache state a forward leaves (input just written by its producer, weights not touched since
        the previous forward) instead of on operands kept hot by its own repetitions, which flatters the tiles that re-read
        their weights most (measured: a layer3 1x1 conv chosen at 25 us isolated runs 33 us in the forward)."""
        self.graph = None
        concurrent = max(1, int(concurrent))
        bigs = [torch.zeros(512 << 20, dtype=torch.uint8, device=self.device) for _ in range(concurrent)]   # zero tickets
        big = bigs[0]
        side = [torch.cuda.Stream(self.device) for _ in range(concurrent - 1)]
        results = []
        shortlist = []
        best_by_prec = []                    # per layer: {arithmetic: (tile, split)} = its fastest candidate of every arithmetic (first look)
        for st in self.conv_steps:
            d = st.desc
            K = d.KH * d.KW * sum(d.seg_len[i] for i in range(d.n_seg)) + max(0, int(d.c2))
            ksteps = (K + 31) // 32
            M = d.N * d.OH * d.OW
            self._refresh_amax_for(st)
            cands = []
            for prec in (precisions if precisions is not None else (int(d.precision),)):
                for tile in (_ffi.BF16X3_TILE_IDS if prec == _ffi.PREC_BF16X3 else (_ffi.FP16X2_TILE_IDS if prec == _ffi.PREC_FP16X2
                                                                                    else _ffi.TILE_IDS)):
                    for split in (splits or (1, -1, -2, 2, 3, 4, 6, 8, 12, 16, 24, 32)):
                        if split > 1 and ksteps // split < 2:
                            continue
                        st.choose(tile, split, prec)
                        if lib().tsod_conv2d_workspace_bytes(byref(d)) > big.numel():   # exact need of THIS candidate
                            continue
                        cands.append((tile, split, prec))
            def time_candidate(tile, split, prec, n_reps):
                """elapsed ms per launch of this candidate (None: the library refuses it)"""
                st.choose(tile, split, prec)
                args = list(st.args)
                args[st.ws_index], args[st.ws_index + 1] = ptr(big), big.numel()
                s = stream_ptr()
                conv_fn = st.fn
                rc = conv_fn(*args, s)          # warm
                if rc != 0:
                    return None                               # a candidate the library refuses is skipped, not fatal
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if concurrent == 1:
                    e0.record()
                    for _ in range(n_reps):
                        rc |= conv_fn(*args, s)
                    e1.record()
                else:
                    cur = torch.cuda.current_stream(self.device)
                    for st2 in side:
                        st2.wait_stream(cur)
                    e0.record()
                    for st2 in side:
                        st2.wait_stream(cur)
                    for _ in range(n_reps):
                        rc |= conv_fn(*args, s)
                        for ci, st2 in enumerate(side):
                            a2 = list(args)
                            a2[st.ws_index] = ptr(bigs[ci + 1])
                            rc |= conv_fn(*a2, st2.cuda_stream)
                    for st2 in side:
                        cur.wait_stream(st2)
                    e1.record()
                e1.synchronize()
                if rc != 0:
                    return None
                return e0.elapsed_time(e1) / (n_reps * concurrent)

            timed = []
            for tile, split, prec in cands:
                t = time_candidate(tile, split, prec, reps)
                if t is not None:
                    timed.append((t, tile, split, prec))
            # second look at the few fastest with four times the repetitions: at 25-50 us per launch a 3-repetition sample
            # is noisy enough to pick a 3-5 % slower schedule now and then
            timed.sort()
            best, second_look = None, []
            # the four fastest of the first look, plus the fastest candidate of every arithmetic that is not among them (the
            # arithmetics are timed one after the other, so a clock step in between can push a whole family out of the top four:
            # seen as 21 instead of 30 fp16x2 layers and 1.62 instead of 1.40 ms of conv launches on one of three runs)
            look = list(timed[:4])
            for pr in sorted({c[3] for c in timed}):
                if all(c[3] != pr for c in look):
                    look.append(next(c for c in timed if c[3] == pr))
            for t0, tile, split, prec in look:
                t = time_candidate(tile, split, prec, 4 * reps) if len(timed) > 1 else t0
                if t is not None:
                    second_look.append((t, tile, split, prec))
                    if best is None or t < best[0]:
                        best = (t, tile, split, prec)
            if best is None:
                raise TsodError(f"autotune: no runnable (tile, split) candidate for {st.name}")
            # a tie on the clock (within 2 %) goes to the candidate that moves fewer bytes beyond the L2s: the K-slice slabs
            # (written and read back) plus the weights, which every XCD fetches for itself under the uniform schedules and every
            # WORKGROUP under the balanced one (its workgroups walk K out of step, so a weight block is in nobody else's L2 when
            # it is wanted: measured 1.4 GB for a 14 MB layer4 3x3 at batch 8, 2 % faster than the hybrid schedule's 0.28 GB)
            def beyond_l2(tile, split, prec):
                st.choose(tile, split, prec)
                slabs = max(0, int(lib().tsod_conv2d_workspace_bytes(byref(d))) - (256 << 10))
                wbytes = d.Cout * K * (6 if prec == _ffi.PREC_BF16X3 else 4)   # (fp16x2: 4)
                m = re.search(r"(\d+)x(\d+)", TILE_NAMES[tile])
                tiles_m = -(-M // int(m.group(1))) if m else 8
                return 2 * slabs + wbytes * (tiles_m if split == -2 else 8)
            close = [c for c in second_look if c[0] <= best[0] * 1.02]
            if len(close) > 1:
                best = min(close, key=lambda c: (beyond_l2(c[1], c[2], c[3]), c[0]))
            st.choose(best[1], best[2], best[3])
            n_short = max(1, int(in_sequence), int(keep_shortlist))
            if n_short > 1:
                # the shortlist of the second looks: the fastest candidates of the first look, at most TWO schedules per (tile,
                # arithmetic) - a tile whose repetitions keep its operands hot takes every place with its K schedules otherwise, and
                # the look inside the sequence (cold weights, the input where its producer left it) then has nothing else to choose
                # from (round 5: d64x128k64, 18.5 us isolated / 26.4 in sequence on layer3's 1x1 convs, had pushed d128x128k32 out
                # of the list on 16 layers) - plus every arithmetic's best
                short, per = [], {}
                for c in timed:
                    if per.get((c[1], c[3]), 0) < 2:
                        short.append(c)
                        per[(c[1], c[3])] = per.get((c[1], c[3]), 0) + 1
                    if len(short) >= n_short:
                        break
                for pr in sorted({c[3] for c in timed}):
                    if all(c[3] != pr for c in short):
                        short.append(next(c for c in timed if c[3] == pr))
            else:
                short = list(timed[:1])
            shortlist.append([(tile, split, prec) for _, tile, split, prec in short])
            by = {}
            for t_, tile, split, prec in timed:                      # (sorted: the first of an arithmetic is its fastest)
                by.setdefault(prec, (tile, split))
            best_by_prec.append(by)
            results.append((st.name, best[0], best[1], best[2], st.flops, best[3]))
            if verbose:
                print(f"  {st.name:34s} {TILE_NAMES[best[1]]:8s} split {best[2]:3d} {_ffi.PREC_NAMES[best[3]]:6s} {best[0] * 1e3:8.1f} us "
                      f"{st.flops / best[0] / 1e9:7.1f} TF/s")
        self.last_shortlist = shortlist if keep_shortlist > 0 else None      # (``keep_shortlist`` fastest per layer: refine_in_flight)
        if in_sequence > 0 and concurrent == 1 and len(self.conv_steps) > 1:
            results = self._refine_in_sequence(shortlist, results, big, reps=5, verbose=verbose)
        del big, bigs
        self.finalize()
        if concurrent == 1 and len(self.conv_steps) > 1 and len({p for by in best_by_prec for p in by}) > 1:
            results = self._whole_table_check(best_by_prec, results, verbose)
        self.clear_range_flag()      # (timing launches ran on whatever the pooled buffers held: not a forward's verdict)
        return results

    def _whole_table_check(self, best_by_prec, results, verbose):
        """Guard of the serial table against a transient during the per-layer looks (the arithmetics of a layer are timed one after
        the other; on a box that has just started, or right after a CPU-heavy phase, one family of kernels has been seen to time
        10-20 % slow for a third of the layers, and the table then keeps 20 bf16x3 layers and 0.2 ms it need not have): ONE pass
        over the conv sequence is timed for the table as chosen and for the tables in which every layer takes its own fastest
        candidate of ONE arithmetic (where it has one); the fastest table as a whole is pinned."""
        chosen = self.export_tiles()
        t_best, best_table, best_name = self.sequence_time(), chosen, "as tuned"
        for prec in sorted({p for by in best_by_prec for p in by}):
            table = [(row[0],) + (by[prec] + (prec,) if prec in by else tuple(row[1:])) for row, by in zip(chosen, best_by_prec)]
            if table == chosen:
                continue
            self.import_tiles(table)
            t = self.sequence_time()
            if t < t_best * 0.99:
                t_best, best_table, best_name = t, table, f"every layer on {_ffi.PREC_NAMES[prec]}"
        self.import_tiles(best_table)
        if verbose or best_table is not chosen:
            print(f"  whole-table check: {best_name} ({t_best * 1e3:.1f} us per pass)")
        if best_table is not chosen:
            results = [(r[0], r[1], row[1], row[2], r[4], row[3]) for r, row in zip(results, best_table)]
        return results

    def _refine_in_sequence(self, shortlist, results, big, reps, verbose):
        """One sweep over the layers: each shortlisted candidate of layer i is timed as launch i of the whole conv sequence
        (HIP events around that one launch, median of ``reps`` passes); the other layers run their current choice."""
        import statistics
        s = stream_ptr()

        def args_of(st):
            a = list(st.args)
            a[st.ws_index], a[st.ws_index + 1] = ptr(big), big.numel()
            return a
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        out = []
        for i, st in enumerate(self.conv_steps):
            cands = shortlist[i]
            timed = []
            for tile, split, prec in cands:
                st.choose(tile, split, prec)
                all_args = [args_of(t) for t in self.conv_steps]
                ts = []
                rc = 0
                for _ in range(reps + 1):
                    for j, t in enumerate(self.conv_steps):
                        if j == i:
                            e0.record()
                        rc |= t.fn(*all_args[j], s)
                        if j == i:
                            e1.record()
                    e1.synchronize()
                    ts.append(e0.elapsed_time(e1))
                if rc == 0:
                    timed.append((statistics.median(ts[1:]), tile, split, prec))
            timed.sort()
            best = timed[0] if timed else (results[i][1], results[i][2], results[i][3], results[i][5])
            st.choose(best[1], best[2], best[3])
            out.append((st.name, best[0], best[1], best[2], st.flops, best[3]))
            if verbose:
                was = results[i]
                note = "" if (was[2], was[3], was[5]) == (best[1], best[2], best[3]) else f"   (isolated pick: {TILE_NAMES[was[2]]} split {was[3]})"
                print(f"  {st.name:34s} {TILE_NAMES[best[1]]:11s} split {best[2]:3d} {_ffi.PREC_NAMES[best[3]]:6s} {best[0] * 1e3:8.1f} us in sequence "
                      f"{st.flops / best[0] / 1e9:7.1f} TF/s{note}")
        return out

    def export_tiles(self):
        """[(name, tile, split_k, precision), ...] as currently pinned in the descriptors (tile 0 / split 0 = heuristic)."""
        return [(st.name, int(st.desc.tile), int(st.desc.split_k), int(st.desc.precision)) for st in self.conv_steps]

    def import_tiles(self, tiles):
        """Pin (tile, split_k[, precision]) choices saved by export_tiles (same plan geometry; 3-tuples = f32)."""
        if len(tiles) != len(self.conv_steps):
            raise TsodError("tile table does not match this plan")
        for st, row in zip(self.conv_steps, tiles):
            name, tile, split = row[0], row[1], row[2]
            if name != st.name:
                raise TsodError(f"tile table mismatch: {name} vs {st.name}")
            st.choose(tile, split, int(row[3]) if len(row) > 3 else 0)
        return self.finalize()

    def import_tiles_by_name(self, tiles):
        """Pin the rows of a table whose layer names this plan has (a table tuned on the same model with ANOTHER launch
        structure, e.g. before some bottlenecks were fused); layers the table does not name keep their choice."""
        by_name = {row[0]: row for row in tiles}
        for st in self.conv_steps:
            row = by_name.get(st.name)
            if row is not None:
                st.choose(row[1], row[2], int(row[3]) if len(row) > 3 else 0)
        return self.finalize()

    def tile_choices(self):
        out = []
        for st in self.conv_steps:
            t, s = c_int32(0), c_int32(0)
            lib().tsod_conv2d_resolve(byref(st.desc), byref(t), byref(s))
            out.append((st.name, t.value, s.value))
        return out


class InFlightMeter:
    """The figure of merit of a tile table under SEVERAL requests in flight: every slot's stream runs the whole conv sequence of its
    plan, the slots staggered around it (slot s starts s / n of the way in, wrapping), ``rounds`` passes; ``measure`` = the median
    of three such timings (ms).  Stale activations are read where the rotation puts a consumer ahead of its producer - timing only."""

    def __init__(self, plans, rounds: int = 4):
        self.plans, self.rounds = plans, rounds
        self.n = len(plans)
        dev = plans[0].device
        self.L = len(plans[0].conv_steps)
        self.streams = [torch.cuda.Stream(dev) for _ in range(self.n)]
        self.offs = [(s * self.L) // self.n for s in range(self.n)]
        self.cur = torch.cuda.current_stream(dev)
        self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.bigs = [torch.zeros(512 << 20, dtype=torch.uint8, device=dev) for _ in range(self.n)]   # zero tickets, any candidate's slabs

    def args(self):
        """the launches of every plan as they stand now (call again after a choose()), on the meter's own workspaces"""
        out = []
        for pl, big in zip(self.plans, self.bigs):
            rows = []
            for st in pl.conv_steps:
                a = list(st.args)
                a[st.ws_index], a[st.ws_index + 1] = ptr(big), big.numel()
                rows.append((st.fn, a))
            out.append(rows)
        return out

    def measure(self, args=None):
        import statistics
        args = self.args() if args is None else args
        streams, cur, e0, e1, L = self.streams, self.cur, self.e0, self.e1, self.L
        ts = []
        for _ in range(3):
            for st_ in streams:
                st_.wait_stream(cur)
            e0.record(cur)
            for st_ in streams:
                st_.wait_event(e0)
            rc = 0
            for _r in range(self.rounds):
                for j in range(L):
                    for s_, st_ in enumerate(streams):
                        fn, a = args[s_][(j + self.offs[s_]) % L]
                        rc |= fn(*a, st_.cuda_stream)
            for st_ in streams:
                cur.wait_stream(st_)
            e1.record(cur)
            e1.synchronize()
            if rc != 0:
                return None
            ts.append(e0.elapsed_time(e1))
        return statistics.median(ts)


def best_table_in_flight(plans, tables: dict, rounds: int = 4, turns: int = 3):
    """Which of several whole tile tables (name -> export_tiles rows of the plans' structure) serves ``len(plans)`` requests in
    flight fastest: each is pinned in every plan and timed with the staggered-streams measure (InFlightMeter), the tables in turn,
    the best of ``turns`` turns each.  The fastest stays pinned.  Returns (its name, {name: us of conv time per forward})."""
    meter = InFlightMeter(plans, rounds)
    best = {k: float("inf") for k in tables}
    for _ in range(turns):
        for k, rows in tables.items():
            for pl in plans:
                pl.import_tiles(rows)
            t = meter.measure()
            if t is not None:
                best[k] = min(best[k], t)
    name = min(best, key=best.get)
    del meter
    for pl in plans:
        pl.import_tiles(tables[name])
        pl.clear_range_flag()
    return name, {k: v / rounds / len(plans) * 1e3 for k, v in best.items()}


def refine_in_flight(plans, shortlist, rounds: int = 4, verbose: bool = False):
    """Second look at a tile table whose objective is SEVERAL requests in flight (serving.InFlightDetector): ``plans`` are the
    backbone plans of the server's slots (same geometry, own buffers and workspaces), ``shortlist[i]`` the candidates of conv
    layer i (``Plan.autotune(..., keep_shortlist=k)``: its k fastest by the first look, which times copies of ONE layer side by
    side on hot operands).  Here every slot's stream runs the whole conv sequence, the slots STAGGERED around it (slot s starts
    s / n of the way in, wrapping), so that at any moment the chip holds launches of different layers, as a pipelined server
    does; the time of ``rounds`` such passes is the figure of merit.  One sweep over the layers: the candidate that makes the
    passes fastest is pinned in every plan.  Stale activations are read where the rotation puts a consumer ahead of its
    producer - timing only; the plans are finalized (workspaces re-bound) on return.  Returns the table (export_tiles)."""
    meter = InFlightMeter(plans, rounds)
    n, L = meter.n, meter.L
    all_args, measure = meter.args, meter.measure

    base = measure(all_args())
    changed = 0
    for i in range(L):
        keep = (int(plans[0].conv_steps[i].desc.tile), int(plans[0].conv_steps[i].desc.split_k), int(plans[0].conv_steps[i].desc.precision))
        best_t, best_c = base, keep
        for cand in shortlist[i]:
            if tuple(cand) == keep:
                continue
            for pl in plans:
                pl.conv_steps[i].choose(*cand)
            t = measure(all_args())
            if t is not None and t < best_t * 0.997:                # (0.3 %: below that the passes' own noise decides)
                best_t, best_c = t, tuple(cand)
        for pl in plans:
            pl.conv_steps[i].choose(*best_c)
        if best_c != keep:
            changed += 1
            if verbose:
                print(f"  in flight: {plans[0].conv_steps[i].name:34s} {TILE_NAMES[keep[0]]} split {keep[1]} -> {TILE_NAMES[best_c[0]]} split {best_c[1]}"
                      f"   ({base / rounds / n * 1e3:.1f} -> {best_t / rounds / n * 1e3:.1f} us of conv time per forward)")
        base = best_t
    del meter
    for pl in plans:
        pl.finalize()
    if verbose:
        print(f"  in flight: {changed} of {L} picks changed; conv time per forward with {n} staggered streams {base / rounds / n * 1e3:.1f} us")
    for pl in plans:
        pl.clear_range_flag()                                        # (see Plan.autotune)
    return plans[0].export_tiles()


def _invalidate_after_load(module, incompatible_keys):
    module.invalidate_packed()


class PlanOwner:
    """Mixin of the modules that own launch plans and packed weights (the backbones, the RPN, the RoI head).

    * ``_plans``: LRU-bounded cache of plans keyed by (input shape, device, slot) - a stream of differently sized
      inputs cannot grow HBM without bound (``max_plans``; a plan evicted here stays alive for as long as a HIP graph's
      ``run`` closure references it).
    * ``_packed_cache``: folded / packed weights per (layer, device), shared by all plans and in-flight slots.
    * Both are dropped whenever the weights may have changed: ``.to()`` / ``.cuda()`` (``_apply``) and EVERY
      ``load_state_dict`` that reaches this module, also through a parent (``nn.Module.load_state_dict`` recurses via
      ``_load_from_state_dict`` and never calls a child's ``load_state_dict``; the post-hook registered here does fire).
      In-place edits of parameters (``p.data.mul_()``, an optimiser step) are invisible: call ``invalidate_packed()``.
    * ``weights_version`` counts invalidations: a graph captured by ``FasterRCNN.make_graphed`` refuses to replay once it
      is stale (it would run the old folded weights)."""
    max_plans = 8
    conv_precision = "f32"       # "f32" | "bf16x3" | "fp16x2": default arithmetic of the dense convs of plans built from now on
    fuse_bottleneck = False      # ResNet: identity bottlenecks with 64 mid channels (layer1.1, layer1.2) as ONE launch each
                                 # (tsod_bottleneck_fp16x2; FasterRCNN.tune switches it on where it measures faster)
    fuse_stem = False            # ResNet: conv1 + bn1 + PReLU + max pool (and the NCHW -> NHWC pass) as ONE launch (tsod_stem_fp16x2)
    fuse_shortcut = True         # ResNet: a bottleneck's last 1x1 conv + its projection shortcut as one stacked-K GEMM
                                 # (set False + invalidate_packed() for the one-launch-per-conv plan, e.g. to pin a tile
                                 # table recorded from it)

    fuse_projection = False      # ... and (with fuse_bottleneck) the block with the 1x1 projection shortcut at stride 1 (layer1.0) too

    def set_fuse_bottleneck(self, on: bool, projection: bool = False):
        """Switch the one-launch bottlenecks on / off for plans built from now on (existing plans are dropped, packed weights stay).
        ``projection``: the block whose shortcut is a 1x1 projection (layer1.0) as one launch too (only with ``on``)."""
        projection = bool(on) and bool(projection)
        if bool(on) != bool(self.fuse_bottleneck) or projection != bool(self.fuse_projection):
            self.fuse_bottleneck, self.fuse_projection = bool(on), projection
            self.__dict__["_plans"] = OrderedDict()
        return self

    def set_fuse_stem(self, on: bool):
        """Switch the one-launch stem on / off for plans built from now on (existing plans are dropped, packed weights stay)."""
        if bool(on) != bool(self.fuse_stem):
            self.fuse_stem = bool(on)
            self.__dict__["_plans"] = OrderedDict()
        return self

    def set_structure(self, table) -> "PlanOwner":
        """The launch structure a tuning table was made for (``FasterRCNN.tune``: "fuse_bottleneck", "fuse_projection", "fuse_stem"; absent = off)."""
        table = table or {}
        self.set_fuse_bottleneck(bool(table.get("fuse_bottleneck", False)), bool(table.get("fuse_projection", False)))   # (tables of round 4: identity blocks only)
        self.set_fuse_stem(bool(table.get("fuse_stem", False)))
        return self

    def set_conv_precision(self, precision: str):
        """Arithmetic of the dense conv GEMMs: "f32" (v_mfma_f32_32x32x2_f32), "bf16x3" (three exact bf16 pieces per
        operand, six bf16 MFMAs per 16 k: f32-accurate, less matrix-pipe time) or "fp16x2" (two fp16 pieces of 2^e x per operand,
        three fp16 MFMAs per 16 k: f32-accurate for every finite input - e follows each tensor's abs-max per forward through its
        range words; only with the words switched off (TSOD_NO_RANGE_WORDS / dynamic_scale False) is the static 2^4 and its
        |x| < 4094 range in force; raise_if_error() reports non-finite accumulators either way).  Existing plans are dropped."""
        if precision not in ("f32", "bf16x3", "fp16x2"):
            raise ValueError(precision)
        self.conv_precision = precision
        self.__dict__["_plans"] = OrderedDict()
        return self

    RANGE_WORDS = 64             # range words per device: one per in-flight slot (slot % 64)

    def _range_word(self, device, slot: int = 0) -> torch.Tensor:
        """The int32 word the fp16x2 launches of this owner's plans of (device, slot) report into (include/tsod.h: range_flag): a
        one-element view of ONE int32[64] tensor per device, which belongs to the owner (it survives plan eviction), so every
        in-flight slot has a word of its own (serving.result(ticket) blames the request at fault, not whichever is collected
        first) and a module that holds plans on two devices never hands a launch a pointer into the other device's memory."""
        words = self.__dict__.setdefault("_range_words", {})
        device = torch.device(device)
        t = words.get(device)
        if t is None:
            t = words[device] = torch.zeros(self.RANGE_WORDS, dtype=torch.int32, device=device)
        i = int(slot) % self.RANGE_WORDS
        return t[i:i + 1]

    def _range_mirror(self, device):
        """(page-locked host int32[64], its device-side address) - the host's copy of this device's range words, one per in-flight
        slot, written by ``publish_range_word`` at the end of a forward: ``range_flag_raised(slot, host=True)`` then costs a read
        of host memory instead of a device-to-host copy per request.  None where the memory cannot be mapped into the device."""
        mirrors = self.__dict__.setdefault("_range_mirrors", {})
        device = torch.device(device)
        if device not in mirrors:
            from ctypes import c_void_p
            with torch.inference_mode(False):                      # (written from outside inference mode too: raise_if_error)
                host = torch.zeros(self.RANGE_WORDS, dtype=torch.int32).pin_memory()
            dptr = c_void_p(0)
            with torch.cuda.device(device):
                rc = lib().tsod_host_mapped_pointer(host.data_ptr(), byref(dptr))
            mirrors[device] = (host, int(dptr.value)) if rc == 0 and dptr.value else None
        return mirrors[device]

    def publish_range_word(self, plan) -> None:
        """Copy the plan's range word to the host mirror (one thread, stream-ordered, capturable): last launch of a detector forward."""
        m = self._range_mirror(plan.device)
        if m is not None:
            slot = int(getattr(plan, "slot", 0)) % self.RANGE_WORDS
            check(lib().tsod_word_publish_i32(ptr(plan.range_flag), m[1] + 4 * slot, stream_ptr()), "word_publish")

    def raise_if_error(self, slot=None):
        """Surface what the fp16x2 launches of ANY plan of this owner (``slot``: of that in-flight slot only) reported since the
        last call (evicted plans included: the words belong to the owner): a launch that ended with non-finite accumulators -
        non-finite input, or, for a conv without range words, an activation beyond the static exponent's range; its outputs are
        garbage.  One 256-byte device read per device (a sync)."""
        bad = False
        for t in self.__dict__.get("_range_words", {}).values():
            v = t if slot is None else t[int(slot) % self.RANGE_WORDS:int(slot) % self.RANGE_WORDS + 1]
            if bool(v.cpu().any()):
                bad = True
                with torch.inference_mode():
                    v.zero_()
        if bad:
            for m in self.__dict__.get("_range_mirrors", {}).values():     # (the host's copies follow)
                if m is not None:
                    if slot is None:
                        m[0].zero_()
                    else:
                        m[0][int(slot) % self.RANGE_WORDS] = 0
            raise TsodError("fp16x2: a conv layer ended with non-finite accumulators - non-finite input (or, without range words, "
                            f"an activation beyond +-{65504 // (1 << FP16X2_A_SCALE_EXP)}); its outputs are garbage")

    def range_flag_raised(self, slot=None, host=False) -> bool:
        """The words as they are now (one small device read): True once a launch that has COMPLETED reported (serving.result()).
        ``host`` = True (with ``slot``): read the host mirror instead - what the slot's last COMPLETED forward published at its end
        (``publish_range_word``): no device call at all; falls back to the device read where there is no mirror."""
        if host and slot is not None:
            mirrors = self.__dict__.get("_range_mirrors", {})
            if mirrors and all(m is not None for m in mirrors.values()):
                return any(int(m[0][int(slot) % self.RANGE_WORDS]) != 0 for m in mirrors.values())
        for t in self.__dict__.get("_range_words", {}).values():
            v = t if slot is None else t[int(slot) % self.RANGE_WORDS:int(slot) % self.RANGE_WORDS + 1]
            if bool(v.cpu().any()):
                return True
        return False

    def _init_plan_owner(self):
        self.__dict__["_plans"] = OrderedDict()
        self.__dict__["_packed_cache"] = {}
        self.__dict__["weights_version"] = 0
        self.register_load_state_dict_post_hook(_invalidate_after_load)

    def invalidate_packed(self):
        """Drop compiled plans and packed weights (call after changing weights in place).  The fp16x2 activation exponents
        calibrated for the old weights go with them (in place: plans that are still referenced see the change)."""
        self.__dict__["_plans"] = OrderedDict()
        self.__dict__["_packed_cache"] = {}
        self.__dict__.setdefault("_a_exps", {}).clear()
        self.__dict__["_range_words"] = {}       # (per device; the plans that pointed at the old words are gone with them)
        # the host mirrors STAY: a graph somebody still holds has the mirror's mapped address baked into its last launch, and page-
        # locked memory handed back to the allocator could be anybody's by the time that graph is replayed
        for m in self.__dict__.get("_range_mirrors", {}).values():
            if m is not None:
                m[0].zero_()
        self.__dict__["weights_version"] = self.__dict__.get("weights_version", 0) + 1

    def _bump_version(self):
        self.__dict__["weights_version"] = self.__dict__.get("weights_version", 0) + 1

    def _apply(self, fn, *a, **k):
        self.invalidate_packed()
        return super()._apply(fn, *a, **k)

    def __getstate__(self):                       # copy.deepcopy / pickling: plans hold ctypes objects and raw pointers
        st = self.__dict__.copy()
        st["_plans"], st["_packed_cache"], st["_range_words"], st["_range_mirrors"] = OrderedDict(), {}, {}, {}
        return st

    def _cached_plan(self, key, build: Callable):
        plans = self._plans
        plan = plans.get(key)
        if plan is None:
            plan = plans[key] = build()
            shared = self.__dict__.setdefault("_a_exps", {})      # fp16x2 activation exponents by layer name, for every plan of this owner
            if isinstance(plan, Plan):
                # the owner's range word of (the plan's device, its in-flight slot): survives plan eviction
                slot = key[2] if isinstance(key, tuple) and len(key) > 2 and isinstance(key[2], int) else 0
                flag = self._range_word(plan.device, slot)
                assert flag.device == plan.device
                plan.slot = slot
                plan.a_exps = shared
                plan.range_flag = flag
                plan.on_calibrated = self._bump_version           # (a graph captured at detector level holds the old exponents)
                for st in plan.conv_steps:
                    st.exps = shared
                    st.range_flag = flag
                    if int(st.desc.precision) == _ffi.PREC_FP16X2:    # (built in that arithmetic: choose() ran before the dict was shared)
                        st.desc.a_scale_exp = int(shared.get(st.name, FP16X2_A_SCALE_EXP))
                        st.desc.range_flag = ptr(flag)
                for st in plan.fused_steps:
                    st.desc.range_flag = ptr(flag)
            while len(plans) > max(1, int(self.max_plans)):
                plans.popitem(last=False)
        else:
            plans.move_to_end(key)
        return plan

    # -- the backbones' plan lookup (build_plan(N, H, W, device) is theirs) ---------------------
    def _plan_for_shape(self, shape, device, slot: int = 0) -> "Plan":
        device = torch.device(device)
        if device.type != "cuda":
            raise TsodError("a CUDA/ROCm device is required")
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        shape = tuple(int(v) for v in shape)

        def build():
            if self.training:
                raise TsodError("the HIP path implements the inference forward only: call .eval() first")
            return self.build_plan(shape[0], shape[2], shape[3], device)
        # slot: independent buffer sets for forwards in flight concurrently (the packed weights are shared)
        return self._cached_plan((shape, device, slot), build)

    def _plan_for(self, x, slot: int = 0) -> "Plan":
        _ffi.require_cuda(x, type(self).__name__ + ".forward")
        if x.dim() != 4 or x.shape[1] != 3:
            raise TsodError(f"expected [N,3,H,W], got {tuple(x.shape)}")
        return self._plan_for_shape(x.shape, x.device, slot)

    def forward_nhwc(self, x, slot: int = 0) -> torch.Tensor:
        """[N,3,H,W] (or NHWC4Images) -> NHWC feature map (plan-owned buffer, valid until the next forward)."""
        plan = self._plan_for(x, slot)
        stage_input(plan, x)
        plan.run()
        return plan.output_nhwc

    def input_buffer(self, N, H, W, device, slot: int = 0):
        """The plan's own input buffer for [N,3,H,W] images as ``NHWC4Images``: an input pipeline that writes there
        (dataset.transform.EvalTransform.batch(..., out=...)) hands its result to the first conv without any copy."""
        return _ffi.NHWC4Images(self._plan_for_shape((N, 3, H, W), device, slot).input_nhwc)

    def drop_plan(self, shape=None, slot=None):
        """Forget the plans of an input shape and / or slot (their buffers are freed once no graph references them)."""
        for key in [k for k in self._plans if (shape is None or k[0] == tuple(shape)) and (slot is None or k[2] == slot)]:
            del self._plans[key]


def stage_input(plan: "Plan", x) -> None:
    """Put the images of one forward into the plan's NHWC(4) input buffer: NCHW tensors go through the layout kernel
    (tsod_nchw_to_nhwc_f32), ``NHWC4Images`` are already in layout (no launch at all when they were written straight
    into ``plan.input_nhwc``, see ``input_buffer`` of the backbones)."""
    from ._ffi import NHWC4Images
    plan.reset_amax()                                   # the range words of this forward start from zero
    if plan.stem_step is not None:                      # the one-launch stem reads the images where they are
        plan.bind_input(x)
        return
    a_in = plan.amax_ptr(plan.input_nhwc)
    if isinstance(x, NHWC4Images):
        if x.data.data_ptr() != plan.input_nhwc.data_ptr():
            with torch.inference_mode():                # the plan's buffers may have been allocated under inference mode
                plan.input_nhwc.copy_(x.data)
        if a_in:                                        # (nobody of ours produced these pixels: one pass for their range)
            check(lib().tsod_absmax_f32(ptr(plan.input_nhwc), plan.input_nhwc.numel(), a_in, stream_ptr()), "absmax")
        return
    x = x.contiguous()
    N, _, H, W = x.shape
    check(lib().tsod_nchw_to_nhwc_amax_f32(ptr(x), N, 3, H, W, ptr(plan.input_nhwc), 4, 4, a_in or None, stream_ptr()), "nchw_to_nhwc")
