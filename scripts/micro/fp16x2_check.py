import math, sys, torch
import torch.nn.functional as F
sys.path.insert(0, '/root/repo')
from two_stage_object_detection_amd import hip_ops as ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(5)
ok = True
for (N, Cin, Cout, H, W, k, res) in [(1, 256, 256, 50, 84, 3, False), (2, 1024, 256, 25, 21, 1, False), (1, 512, 2048, 13, 21, 1, True), (1, 64, 64, 37, 41, 3, True), (1, 128, 96, 19, 23, 3, False)]:
    x = torch.randn(N, Cin, H, W, generator=g); x = torch.maximum(x, 0.25 * x)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(k * k * Cin)
    r = torch.randn(N, Cout, H, W, generator=g) if res else None
    ref = F.conv2d(x.double(), w.double(), padding=k // 2)
    if res: ref = ref + r.double()
    ref = torch.where(ref >= 0, ref, 0.25 * ref).float()
    xn = ops.nchw_to_nhwc(x.to(dev)); wp = ops.pack_conv_weight(w.to(dev)); rn = ops.nchw_to_nhwc(r.to(dev)) if res else None
    tol = 3e-6 * math.sqrt(k * k * Cin) + 1e-5
    e3 = (ops.nhwc_to_nchw(ops.conv2d_nhwc(xn, wp, pad=k // 2, tile=22, split_k=1, precision=1, residual=rn, act=1, slope=0.25)).cpu() - ref).abs().max().item()
    for split in (1, 3, -1, -2):
        out = ops.conv2d_nhwc(xn, wp, pad=k // 2, tile=22, split_k=split, precision=2, residual=rn, act=1, slope=0.25, a_scale_exp=4)
        e = (ops.nhwc_to_nchw(out).cpu() - ref).abs().max().item()
        flag = e <= tol
        ok &= flag
        print(f"{Cin}->{Cout} k{k} {H}x{W} split {split:2d}: fp16x2 err {e:.2e} (bf16x3 {e3:.2e}, tol {tol:.1e}) {'ok' if flag else 'FAIL'}", flush=True)
print('ALL OK' if ok else 'FAILED')
