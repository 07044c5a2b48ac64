"""ResNet backbones with the reference's module surface (models/resnet.py of the reference) and a
HIP execution path.

Same class names, constructor signatures, sub-module names and parameter shapes as the reference
(``conv1/bn1/relu/maxpool/layer{1..4}[/avgpool/fc]``, blocks ``conv{1,2,3}/bn{1,2,3}/relu/downsample``),
so its checkpoints load with ``strict=True`` and ``torch.manual_seed(s)`` reproduces the same
initial weights (sub-modules are created, and the Kaiming fan_out pass applied, in the same order:
reference models/resnet.py:94-109).

The nn.Conv2d / nn.BatchNorm2d / nn.PReLU children are parameter containers only.  ``forward``
compiles a launch plan of fused conv+BN+PReLU(+residual) implicit-GEMM kernels (engine.Plan) for the
input geometry and runs it on the GPU; there is no eager / CPU path.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import hip_ops
from .._ffi import ACT_NONE, ACT_PRELU, TsodError, lib, ptr, require_cuda
from ..engine import FusedBottleneckWeights, FusedShortcutConv, FusedStemWeights, PackedConv, Plan, PlanOwner, prelu_slope


def _conv(cin, cout, k, stride=1, pad=0, groups=1):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=pad, groups=groups, bias=False)


class _ResidualBlock(nn.Module):
    """Shared machinery of the two block types: a list of (conv, bn) stages, one shared PReLU."""
    expansion = 1
    _stage_names: tuple = ()

    def _emit(self, plan: Plan, x: torch.Tensor, name: str) -> torch.Tensor:
        dev = plan.device
        slope = prelu_slope(self.relu)
        identity = x
        # the whole block as ONE launch (tsod_bottleneck_fp16x2): 64 mid channels, stride 1 - layer1, the HBM-bound stage of the trunk
        # (the 64-channel intermediates stay in LDS): the identity blocks layer1.1 / layer1.2 and (round 5) the block in front of
        # them, whose shortcut is a 1x1 projection at stride 1 (layer1.0): conv3 + shortcut as one stacked-K GEMM inside the launch
        if getattr(plan, "fuse_bottleneck", False) and len(self._stage_names) == 3:
            ident = self.downsample is None and self.conv1.in_channels == self.conv3.out_channels
            proj = (getattr(plan, "fuse_projection", False) and self.downsample is not None and len(self.downsample) == 2 and isinstance(self.downsample[0], nn.Conv2d)
                    and self.downsample[0].kernel_size == (1, 1) and self.downsample[0].stride == (1, 1) and self.downsample[0].groups == 1)
            if ((ident or proj) and self.conv1.out_channels == 64 and self.conv2.groups == 1 and self.conv2.stride == (1, 1)
                    and self.conv1.in_channels % 64 == 0 and self.conv3.out_channels % 64 == 0):
                fb = plan.packed(f"{name}.fused", lambda: FusedBottleneckWeights(self, dev))
                out = plan.pool.alloc((x.shape[0], x.shape[1], x.shape[2], fb.cout))
                return plan.bottleneck(fb, x, out, name=f"{name}.fused")
        # projection shortcut + last 1x1 conv as ONE stacked-K GEMM (engine.FusedShortcutConv): one launch, no shortcut tensor
        # written and re-read as residual.  Needs a 1x1 last conv (Bottleneck) and channel counts the K-steps divide.
        last_name, last_bn = self._stage_names[-1]
        last_conv = getattr(self, last_name)
        fuse = (self.downsample is not None and getattr(plan, "fuse_shortcut", True) and last_conv.kernel_size == (1, 1)
                and last_conv.groups == 1 and last_conv.in_channels % 32 == 0 and self.downsample[0].in_channels % 32 == 0
                and self.downsample[0].kernel_size == (1, 1))
        if self.downsample is not None and not fuse:
            ds_conv, ds_bn = self.downsample[0], self.downsample[1]
            pc = plan.packed(f"{name}.downsample", lambda: PackedConv(ds_conv.weight, dev, bn=ds_bn, stride=ds_conv.stride[0],
                                                                      act=ACT_NONE))
            oh, ow = pc.out_hw(x.shape[1], x.shape[2])
            identity = plan.conv(pc, x, plan.pool.alloc((x.shape[0], oh, ow, pc.cout)), name=f"{name}.downsample")
        cur = x
        last = len(self._stage_names) - 1
        for i, (cname, bname) in enumerate(self._stage_names):
            conv, bn = getattr(self, cname), getattr(self, bname)
            if conv.groups != 1:                      # ResNeXt's grouped 3x3 (models/resnet.py:46-47): direct kernel, no MFMA
                cur = self._emit_grouped(plan, conv, bn, cur, x, slope, f"{name}.{cname}")
                continue
            if i == last and fuse:
                ds_conv, ds_bn = self.downsample[0], self.downsample[1]
                pc = plan.packed(f"{name}.{cname}+downsample", lambda conv=conv, bn=bn: FusedShortcutConv(
                    conv.weight, bn, ds_conv.weight, ds_bn, ds_conv.stride[0], dev, ACT_PRELU, slope))
                out = plan.pool.alloc((cur.shape[0], cur.shape[1], cur.shape[2], pc.cout))
                plan.conv(pc, cur, out, segs=[(0, pc.cin)], name=f"{name}.{cname}+downsample", x2=x, stride2=pc.stride2)
                if cur is not x:
                    plan.pool.release(cur)
                cur = out
                continue
            pc = plan.packed(f"{name}.{cname}", lambda conv=conv, bn=bn: PackedConv(
                conv.weight, dev, bn=bn, stride=conv.stride[0], pad=conv.padding[0], act=ACT_PRELU, slope=slope))
            oh, ow = pc.out_hw(cur.shape[1], cur.shape[2])
            out = plan.pool.alloc((cur.shape[0], oh, ow, pc.cout))
            plan.conv(pc, cur, out, residual=identity if i == last else None, name=f"{name}.{cname}")
            if cur is not x:
                plan.pool.release(cur)
            cur = out
        if identity is not x:
            plan.pool.release(identity)
        return cur

    def _emit_grouped(self, plan: Plan, conv, bn, cur, x, slope, name):
        from ..engine import fold_bn
        C, groups, stride = conv.out_channels, conv.groups, conv.stride[0]
        if conv.kernel_size != (3, 3) or conv.in_channels != C or (C // groups) % 4:
            raise TsodError(f"{name}: only 3x3 grouped convs with as many output as input channels and a multiple of 4 "
                            "channels per group have a HIP kernel (what resnext50_32x4d uses)")

        def make():
            w = conv.weight.detach().float().permute(0, 2, 3, 1).contiguous().to(plan.device)     # [C][3][3][C/groups]
            sc, sh = fold_bn(bn)
            return w, sc.to(plan.device), sh.to(plan.device)
        w, sc, sh = plan.packed(name, make)
        N, H, W, _ = cur.shape
        out = plan.pool.alloc((N, (H - 1) // stride + 1, (W - 1) // stride + 1, C))
        plan.call(lib().tsod_gconv3x3_amax_f32, ptr(cur), N, H, W, C, cur.shape[3], groups, ptr(w), ptr(sc), ptr(sh), stride,
                  ACT_PRELU, float(slope), ptr(out), C, plan.amax_ptr(out) or None, keep=(cur, out, w, sc, sh))
        if cur is not x:
            plan.pool.release(cur)
        return out


class BasicBlock(_ResidualBlock):
    expansion = 1
    _stage_names = (("conv1", "bn1"), ("conv2", "bn2"))

    def __init__(self, in_channel, out_channel, stride=1, downsample=None, **kwargs):
        super().__init__()
        self.conv1 = _conv(in_channel, out_channel, 3, stride, 1)
        self.bn1 = nn.BatchNorm2d(out_channel)
        self.relu = nn.PReLU()
        self.conv2 = _conv(out_channel, out_channel, 3, 1, 1)
        self.bn2 = nn.BatchNorm2d(out_channel)
        self.downsample = downsample


class Bottleneck(_ResidualBlock):
    expansion = 4
    _stage_names = (("conv1", "bn1"), ("conv2", "bn2"), ("conv3", "bn3"))

    def __init__(self, in_channel, out_channel, stride=1, downsample=None, groups=1, width_per_group=64):
        super().__init__()
        width = int(out_channel * (width_per_group / 64.)) * groups
        self.conv1 = _conv(in_channel, width, 1)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = _conv(width, width, 3, stride, 1, groups)      # stride on the 3x3 (v1.5)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = _conv(width, out_channel * self.expansion, 1)
        self.bn3 = nn.BatchNorm2d(out_channel * self.expansion)
        self.relu = nn.PReLU()
        self.downsample = downsample


class ResNet(PlanOwner, nn.Module):
    def __init__(self, block, blocks_num, num_classes=25, include_top=True, groups=1, width_per_group=64):
        super().__init__()
        self.include_top = include_top
        self.in_channel = 64
        self.groups = groups
        self.width_per_group = width_per_group
        self.conv1 = _conv(3, 64, 7, 2, 3)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.PReLU()
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        for i, (ch, n, stride) in enumerate(zip((64, 128, 256, 512), blocks_num, (1, 2, 2, 2)), start=1):
            setattr(self, f"layer{i}", self._make_layer(block, ch, n, stride))
        if include_top:
            self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
            self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        self._init_plan_owner()
        self.out_channels = 512 * block.expansion

    def _make_layer(self, block, channel, block_num, stride=1):
        out_ch = channel * block.expansion
        downsample = None
        if stride != 1 or self.in_channel != out_ch:
            downsample = nn.Sequential(_conv(self.in_channel, out_ch, 1, stride), nn.BatchNorm2d(out_ch))
        blocks = [block(self.in_channel, channel, downsample=downsample, stride=stride, groups=self.groups,
                        width_per_group=self.width_per_group)]
        self.in_channel = out_ch
        blocks += [block(out_ch, channel, groups=self.groups, width_per_group=self.width_per_group)
                   for _ in range(1, block_num)]
        return nn.Sequential(*blocks)

    # -- plan (cache, invalidation, lookup: engine.PlanOwner) -----------------------------------
    def build_plan(self, N, H, W, device) -> Plan:
        """Launch plan for a [N,3,H,W] input: NCHW->NHWC4, 7x7 stem as a 7x8x4 implicit GEMM with
        BN+PReLU, 3x3/s2 max pool, then the residual stages."""
        plan = Plan(device, self._packed_cache)
        plan.precision = {"f32": 0, "bf16x3": 1, "fp16x2": 2}[self.conv_precision]
        plan.fuse_shortcut = bool(self.fuse_shortcut)
        plan.fuse_bottleneck = bool(self.fuse_bottleneck)
        plan.fuse_projection = bool(self.fuse_projection)
        x4 = plan.pool.alloc((N, H, W, 4))
        plan.input_nhwc = x4
        oh, ow = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        ph, pw = (oh - 1) // 2 + 1, (ow - 1) // 2 + 1
        if self.fuse_stem and tuple(self.conv1.weight.shape) == (64, 3, 7, 7):
            # conv1 + bn1 + PReLU + max pool as ONE launch that reads the images where stage_input finds them (NCHW or NHWC4):
            # no layout pass, the 64-channel conv output never leaves the CU (tsod_stem_fp16x2)
            fs = plan.packed("conv1.fused", lambda: FusedStemWeights(self.conv1, self.bn1, self.relu, device))
            cur = plan.stem(fs, N, H, W, plan.pool.alloc((N, ph, pw, 64)), name="conv1+maxpool")
        else:
            stem = plan.packed("conv1", lambda: PackedConv(self.conv1.weight, device, bn=self.bn1, stride=2, pad=3, act=ACT_PRELU,
                                                           slope=prelu_slope(self.relu), cin_pad=4, kw_pad=8))
            assert (oh, ow) == tuple(stem.out_hw(H, W))
            s_out = plan.conv(stem, x4, plan.pool.alloc((N, oh, ow, 64)), name="conv1")
            cur = plan.pool.alloc((N, ph, pw, 64))
            plan.call(lib().tsod_maxpool3x3s2_f32, ptr(s_out), N, oh, ow, 64, 64, ptr(cur), 64, keep=(s_out, cur))
            plan.alias_amax(cur, s_out)          # range words: max |pooled| <= max |stem output|
            plan.pool.release(s_out)
        for li in range(1, 5):
            for bi, blk in enumerate(getattr(self, f"layer{li}")):
                nxt = blk._emit(plan, cur, f"layer{li}.{bi}")
                plan.pool.release(cur)
                cur = nxt
        plan.output_nhwc = cur
        plan.output_amax = plan.amax_ptr(cur)
        return plan.finalize()

    def forward(self, x):
        feat = self.forward_nhwc(x)
        if self.include_top:
            raise TsodError("include_top=True (avgpool + fc classifier) is outside the detector forward path; "
                            "build with include_top=False")
        return hip_ops.nhwc_to_nchw(feat)


def resnet34(num_classes=25, include_top=True):
    return ResNet(BasicBlock, [3, 4, 6, 3], num_classes=num_classes, include_top=include_top)


def resnet50(num_classes=25, include_top=True):
    return ResNet(Bottleneck, [3, 4, 6, 3], num_classes=num_classes, include_top=include_top)


def resnet101(num_classes=25, include_top=True):
    return ResNet(Bottleneck, [3, 4, 23, 3], num_classes=num_classes, include_top=include_top)


def resnext50_32x4d(num_classes=25, include_top=True):
    return ResNet(Bottleneck, [3, 4, 6, 3], num_classes=num_classes, include_top=include_top, groups=32,
                  width_per_group=4)
