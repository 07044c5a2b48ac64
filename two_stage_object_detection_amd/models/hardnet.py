"""HarDNet backbones with the reference's module surface (models/hardnet.py of the reference) and a
HIP execution path (depth_wise=True variants: the only usable ones).

Same class names, sub-module names (``base.N.conv/norm/relu``, ``base.N.dwconv/norm``,
``base.N.layers.L.layer1/layer2``) and parameter shapes as the reference, created in the same order,
so checkpoints load with ``strict=True`` and a seed reproduces the same initial weights.

Execution (engine.Plan), all NHWC f32:
  * every HarDBlock owns ONE wide pixel-major buffer: slice 0 is the block input, slice i the output
    of layer i (each slice padded to a multiple of 4 channels, pad channels hold exact zeros).
    A layer's input "torch.cat(linked layers)" (reference :99-110) is never built: the 1x1 implicit
    GEMM gathers its K dimension from the linked slices (channel segments of the conv descriptor),
    and the producer of every tensor writes straight into its slice, so both concats of the
    reference (:108, :120) are free.
  * 1x1 conv + BN + ReLU6 -> f32 MFMA GEMM with fused epilogue; depthwise 3x3 + BN -> streaming
    stencil kernel writing into the block buffer; tail = 2 depthwise s2 + grouped-pair 1x1.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import hip_ops
from .._ffi import ACT_NONE, ACT_RELU6, TsodError, lib, ptr, require_cuda
from ..engine import PackedConv, Plan, PlanOwner, fold_bn


def _pad4(c: int) -> int:
    return (c + 3) // 4 * 4


class Flatten(nn.Module):
    def forward(self, x):
        return x.view(x.size(0), -1)


class ConvLayer(nn.Sequential):
    """conv(k, stride, pad k//2, no bias) + BN + ReLU6 (reference :38-55; ``dropout`` is unused there too)."""

    def __init__(self, in_channels, out_channels, kernel=3, stride=1, dropout=0.1, bias=False):
        super().__init__()
        self.add_module("conv", nn.Conv2d(in_channels, out_channels, kernel_size=kernel, stride=stride,
                                          padding=kernel // 2, groups=1, bias=bias))
        self.add_module("norm", nn.BatchNorm2d(out_channels))
        self.add_module("relu", nn.ReLU6(True))


class DWConvLayer(nn.Sequential):
    """depthwise 3x3 (pad 1) + BN, no activation (reference :21-36)."""

    def __init__(self, in_channels, stride=1, bias=False):
        super().__init__()
        self.add_module("dwconv", nn.Conv2d(in_channels, in_channels, kernel_size=3, stride=stride, padding=1,
                                            groups=in_channels, bias=bias))
        self.add_module("norm", nn.BatchNorm2d(in_channels))


class CombConvLayer(nn.Sequential):
    def __init__(self, in_channels, out_channels, kernel=1, stride=1):
        super().__init__()
        self.add_module("layer1", ConvLayer(in_channels, out_channels, kernel))
        self.add_module("layer2", DWConvLayer(out_channels, stride=stride))


def hard_block_links(layer: int):
    """Inputs of HarDBlock layer ``layer`` >= 1: layer - 2^i for every i with layer % 2^i == 0 (newest first)."""
    return [layer - (1 << i) for i in range(10) if (1 << i) <= layer and layer % (1 << i) == 0]


class HarDBlock(nn.Module):
    def get_link(self, layer, base_ch, growth_rate, grmul):
        if layer == 0:
            return base_ch, 0, []
        link = hard_block_links(layer)
        out_channels = growth_rate
        for _ in range(len(link) - 1):
            out_channels *= grmul
        out_channels = int(int(out_channels + 1) / 2) * 2
        in_channels = sum(self.get_link(k, base_ch, growth_rate, grmul)[0] for k in link)
        return out_channels, in_channels, link

    def get_out_ch(self):
        return self.out_channels

    def __init__(self, in_channels, growth_rate, grmul, n_layers, keepBase=False, dwconv=False):
        super().__init__()
        self.keepBase = keepBase
        self.in_channels = in_channels
        self.links, self.layer_out = [], []
        self.out_channels = 0
        layers_ = []
        for i in range(n_layers):
            outch, inch, link = self.get_link(i + 1, in_channels, growth_rate, grmul)
            self.links.append(link)
            self.layer_out.append(outch)
            layers_.append(CombConvLayer(inch, outch) if dwconv else ConvLayer(inch, outch))
            if (i % 2 == 0) or (i == n_layers - 1):
                self.out_channels += outch
        self.layers = nn.ModuleList(layers_)
        self.dwconv = dwconv

    # slices of the block buffer: index 0 = block input, i = output of layer i
    def slice_table(self):
        real = [self.in_channels] + list(self.layer_out)
        offs, o = [], 0
        for c in real:
            offs.append(o)
            o += _pad4(c)
        return real, offs, o

    def output_slices(self):
        t = len(self.layers) + 1
        return [i for i in range(t) if (i == 0 and self.keepBase) or i == t - 1 or i % 2 == 1]


_ARCH = {
    68: dict(first_ch=(32, 64), grmul=1.7, gr=(14, 16, 20, 40, 160), n_layers=(8, 16, 16, 16, 4),
             ch_list=(128, 256, 320, 640, 1024), downSamp=(1, 0, 1, 1, 0)),
    85: dict(first_ch=(48, 96), grmul=1.7, gr=(24, 24, 28, 36, 48, 256), n_layers=(8, 16, 16, 16, 16, 4),
             ch_list=(192, 256, 320, 480, 720, 1024), downSamp=(1, 0, 1, 0, 1, 0)),
    39: dict(first_ch=(24, 48), grmul=1.6, gr=(16, 20, 64, 160), n_layers=(4, 16, 8, 4),
             ch_list=(96, 320, 640, 1024), downSamp=(1, 1, 1, 0)),
}


def _gathered_weight(w: torch.Tensor, src_real, cout_pad):
    """[Cout, sum(src_real), 1, 1] -> [cout_pad, 1, 1, sum(pad4(src_real))]: zero columns at the pad
    channels of every gathered slice, zero rows for the padded output channels."""
    cout = w.shape[0]
    w2 = w.detach().float().cpu().view(cout, -1)
    cols, o = [], 0
    for c in src_real:
        blk = torch.zeros(cout_pad, _pad4(c))
        blk[:cout, :c] = w2[:, o:o + c]
        cols.append(blk)
        o += c
    return torch.cat(cols, dim=1).view(cout_pad, 1, 1, -1).contiguous()


def _padded(v: torch.Tensor, n: int, fill=0.0):
    out = torch.full((n,), fill, dtype=torch.float32)
    out[:v.numel()] = v.float().cpu()
    return out


class _RawConv:
    """PackedConv-compatible holder for a pre-gathered 1x1 weight."""

    def __init__(self, w_packed, scale, shift, device, act, cin_real=None, cout_real=None):
        self.w = w_packed.to(device)
        self.cout, self.kh, self.kw, self.cin = self.w.shape
        self.kw_logical, self.cin_src = self.kw, (self.cin if cin_real is None else cin_real)   # for FLOP accounting
        self.cout_real = self.cout if cout_real is None else cout_real
        self.stride, self.pad, self.act, self.slope = 1, 0, act, 0.0
        self.scale = None if scale is None else scale.to(device)
        self.shift = None if shift is None else shift.to(device)

    def out_hw(self, H, W):
        return H, W


class HarDNetFeatureExtraction(PlanOwner, nn.Module):
    def __init__(self, depth_wise=True, arch=39):
        super().__init__()
        cfg = _ARCH[arch if arch in (39, 85) else 68]          # any other value = HarDNet-68, like the reference
        self.arch, self.depth_wise = arch, depth_wise
        first_ch, ch_list, gr = cfg["first_ch"], cfg["ch_list"], cfg["gr"]
        n_layers, down, grmul = cfg["n_layers"], cfg["downSamp"], cfg["grmul"]
        second_kernel, max_pool = (1, False) if depth_wise else (3, True)
        self.base = nn.ModuleList([])
        self.base.append(ConvLayer(in_channels=3, out_channels=first_ch[0], kernel=3, stride=2, bias=False))
        self.base.append(ConvLayer(first_ch[0], first_ch[1], kernel=second_kernel))
        self.base.append(nn.MaxPool2d(kernel_size=3, stride=2, padding=1) if max_pool
                         else DWConvLayer(first_ch[1], stride=2))
        ch = first_ch[1]
        blks = len(n_layers)
        for i in range(blks):
            blk = HarDBlock(ch, gr[i], grmul, n_layers[i], dwconv=depth_wise)
            ch = blk.get_out_ch()
            self.base.append(blk)
            if i == blks - 1 and arch == 85:
                self.base.append(nn.Dropout(0.1))
            self.base.append(ConvLayer(ch, ch_list[i], kernel=1))
            ch = ch_list[i]
            if down[i] == 1:
                self.base.append(nn.MaxPool2d(kernel_size=2, stride=2) if max_pool else DWConvLayer(ch, stride=1))
        self.base.append(nn.Conv2d(ch_list[-1], ch_list[-1], 3, 2, 1, groups=ch_list[-1]))
        self.base.append(nn.ReLU())
        self.base.append(nn.Conv2d(ch_list[-1], ch_list[-1], 3, 2, 1, groups=ch_list[-1]))
        self.base.append(nn.Conv2d(ch_list[-1], 512, 1, groups=512))
        self._init_plan_owner()
        self.out_channels = 512

    # -- plan (cache, invalidation, lookup: engine.PlanOwner) -----------------------------------
    @staticmethod
    def _dw_params(conv: nn.Conv2d, bn, device):
        """depthwise weights as [3][3][C_pad] + per-channel scale/shift (folded BN, or the conv bias)."""
        C = conv.weight.shape[0]
        cp = _pad4(C)
        w = torch.zeros(3, 3, cp)
        w[:, :, :C] = conv.weight.detach().float().cpu().view(C, 9).t().reshape(3, 3, C)
        if bn is not None:
            scale, shift = fold_bn(bn)
            scale, shift = _padded(scale, cp, 0.0), _padded(shift, cp, 0.0)
        else:
            scale = None
            shift = _padded(conv.bias.detach(), cp) if conv.bias is not None else None
        return (w.to(device), None if scale is None else scale.to(device), None if shift is None else shift.to(device), cp)

    def build_plan(self, N, H, W, device) -> Plan:
        if not self.depth_wise:
            raise TsodError("depth_wise=False HarDNet (max-pool variant) has no HIP path; the reference only "
                            "uses depth_wise=True")
        plan = Plan(device, self._packed_cache)
        plan.precision = {"f32": 0, "bf16x3": 1, "fp16x2": 2}[self.conv_precision]
        L = lib()
        mods = list(self.base)
        x4 = plan.pool.alloc((N, H, W, 4))
        plan.input_nhwc = x4

        def dest_for(next_idx, C, h, w):
            """Where the tensor feeding module ``next_idx`` must be written: slice 0 of the next
            HarDBlock's buffer, or a fresh tensor."""
            nxt = mods[next_idx] if next_idx < len(mods) else None
            if isinstance(nxt, HarDBlock):
                _, _, P = nxt.slice_table()
                return plan.pool.alloc((N, h, w, P)), 0
            return plan.pool.alloc((N, h, w, _pad4(C))), 0

        def emit_dw(src, src_off, C, conv, bn, stride, relu, dst, dst_off, name):
            w33, scale, shift, cp = plan.packed(name, lambda: self._dw_params(conv, bn, device))
            n, h, w_, P = src.shape
            plan.call(L.tsod_dwconv3x3_amax_f32, ptr(src), n, h, w_, cp, P, src_off, ptr(w33), ptr(scale), ptr(shift), stride,
                      1 if relu else 0, ptr(dst), dst.shape[3], dst_off, plan.amax_ptr(dst) or None, keep=(src, dst, w33, scale, shift))

        # --- stem: 3x3 s2 conv (3 -> c0, input padded to 4 channels), 1x1 conv, dw3x3 s2
        m0, m1, m2 = mods[0], mods[1], mods[2]
        pc0 = plan.packed("base.0", lambda: PackedConv(m0.conv.weight, device, bn=m0.norm, stride=2, pad=1, act=ACT_RELU6,
                                                       cin_pad=4))
        h, w = pc0.out_hw(H, W)
        t0 = plan.conv(pc0, x4, plan.pool.alloc((N, h, w, pc0.cout)), name="base.0")
        pc1 = plan.packed("base.1", lambda: PackedConv(m1.conv.weight, device, bn=m1.norm, act=ACT_RELU6))
        t1 = plan.conv(pc1, t0, plan.pool.alloc((N, h, w, pc1.cout)), name="base.1")
        plan.pool.release(t0)
        h2, w2 = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        cur, cur_off = dest_for(3, pc1.cout, h2, w2)
        emit_dw(t1, 0, pc1.cout, m2.dwconv, m2.norm, 2, False, cur, cur_off, "base.2")
        plan.pool.release(t1)
        cur_C = pc1.cout
        h, w = h2, w2

        i = 3
        while i < len(mods):
            m = mods[i]
            if isinstance(m, HarDBlock):
                real, offs, P = m.slice_table()
                assert cur.shape[3] == P and cur_C == real[0]
                buf = cur
                for li, comb in enumerate(m.layers, start=1):
                    link = m.links[li - 1]
                    segs = [(offs[k], _pad4(real[k])) for k in link]
                    cout, cp = real[li], _pad4(real[li])
                    def make_layer1(comb=comb, link=link, cp=cp, cout=cout):
                        wg = _gathered_weight(comb.layer1.conv.weight, [real[k] for k in link], cp)
                        sc, sh = fold_bn(comb.layer1.norm)
                        return _RawConv(wg, _padded(sc, cp), _padded(sh, cp), device, ACT_RELU6,
                                        cin_real=sum(real[k] for k in link), cout_real=cout)
                    rc = plan.packed(f"base.{i}.layers.{li - 1}.layer1", make_layer1)
                    tmp = plan.pool.alloc((N, h, w, cp))
                    plan.conv(rc, buf, tmp, segs=segs, name=f"base.{i}.layers.{li - 1}.layer1")
                    emit_dw(tmp, 0, cout, comb.layer2.dwconv, comb.layer2.norm, 1, False, buf, offs[li],
                            f"base.{i}.layers.{li - 1}.layer2")
                    plan.pool.release(tmp)
                # transition 1x1 conv gathers the block's output slices (oldest first)
                outs = m.output_slices()
                i += 1
                if isinstance(mods[i], nn.Dropout):
                    i += 1
                tr = mods[i]
                def make_transition(tr=tr, outs=outs):
                    wg = _gathered_weight(tr.conv.weight, [real[k] for k in outs], tr.conv.weight.shape[0])
                    sc, sh = fold_bn(tr.norm)
                    return _RawConv(wg, sc, sh, device, ACT_RELU6, cin_real=sum(real[k] for k in outs))
                rc = plan.packed(f"base.{i}", make_transition)
                dst, dst_off = dest_for(i + 1, rc.cout, h, w)
                if isinstance(mods[i + 1], DWConvLayer):          # "downsample" dw3x3 at stride 1 follows
                    plan.pool.release(dst)
                    dst, dst_off = plan.pool.alloc((N, h, w, rc.cout)), 0
                plan.conv(rc, buf, dst, segs=[(offs[k], _pad4(real[k])) for k in outs], out_off=dst_off, name=f"base.{i}")
                plan.pool.release(buf)
                cur, cur_off, cur_C = dst, dst_off, rc.cout
                i += 1
            elif isinstance(m, DWConvLayer):
                dst, dst_off = dest_for(i + 1, cur_C, h, w)
                emit_dw(cur, cur_off, cur_C, m.dwconv, m.norm, m.dwconv.stride[0], False, dst, dst_off, f"base.{i}")
                plan.pool.release(cur)
                cur, cur_off = dst, dst_off
                i += 1
            elif isinstance(m, nn.Conv2d) and m.groups == m.in_channels and m.kernel_size == (3, 3):
                relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
                s = m.stride[0]
                nh, nw = (h - 1) // s + 1, (w - 1) // s + 1
                dst = plan.pool.alloc((N, nh, nw, _pad4(cur_C)))
                emit_dw(cur, cur_off, cur_C, m, None, s, relu, dst, 0, f"base.{i}")
                plan.pool.release(cur)
                cur, cur_off, h, w = dst, 0, nh, nw
                i += 2 if relu else 1
            elif isinstance(m, nn.Conv2d) and m.kernel_size == (1, 1) and m.groups == m.out_channels \
                    and m.in_channels == 2 * m.out_channels:
                G = m.out_channels
                wg, bias = plan.packed(f"base.{i}", lambda m=m: (
                    m.weight.detach().float().view(G, 2).contiguous().to(device),
                    None if m.bias is None else m.bias.detach().float().to(device)))
                dst = plan.pool.alloc((N, h, w, G))
                plan.call(L.tsod_gconv1x1_pair_amax_f32, ptr(cur), N * h * w, G, cur.shape[3], ptr(wg), ptr(bias), ptr(dst), G,
                          plan.amax_ptr(dst) or None, keep=(cur, dst, wg, bias))
                plan.pool.release(cur)
                cur, cur_off, cur_C = dst, 0, G
                i += 1
            else:
                raise TsodError(f"no HIP lowering for base.{i}: {type(m).__name__}")
        plan.output_nhwc = cur
        plan.output_amax = plan.amax_ptr(cur)
        return plan.finalize()

    def forward(self, x):
        return hip_ops.nhwc_to_nchw(self.forward_nhwc(x))


class HarNetClassifier(nn.Module):
    """AdaptiveAvgPool2d(1) + Flatten (reference :203-212; attribute name ``clssifier`` kept).  Inside the
    detector it is fused into the RoI pooling kernel; stand-alone it averages an NCHW tensor's H*W."""

    def __init__(self):
        super().__init__()
        self.clssifier = nn.Sequential(nn.AdaptiveAvgPool2d((1, 1)), nn.Flatten())

    def forward(self, x):
        require_cuda(x, "HarNetClassifier")
        n, c, h, w = x.shape
        # mean over H*W as a GEMM against a constant 1/(H*W) row on the f32 MFMA path
        hw = h * w
        flat = torch.nn.functional.pad(x.reshape(n * c, hw), (0, _pad4(hw) - hw)).contiguous()   # K % 4 == 0 for the GEMM
        ones = torch.zeros((4, _pad4(hw)), dtype=torch.float32, device=x.device)
        ones[:, :hw] = 1.0 / hw
        return hip_ops.linear(flat, ones)[:, 0].reshape(n, c)
