"""Numerics of "fp16x2": every f32 operand as TWO fp16 pieces (hi = fp16(s x), lo = fp16(s x - hi), s a per-tensor power of two),
THREE piece products per f32 product (lo*hi, hi*lo, hi*hi) accumulated in f32 in ONE accumulator, result scaled back - against
the f32 MFMA chain and the shipped bf16x3 (three bf16 pieces, six products).  Pure numpy emulation (exact piece products, f32
accumulation in 16-k chunks like the MFMA), error measured against float64 and quoted relative to sum |a b|.

    python scripts/micro/fp16x2_numerics.py

Finding: with the weights scaled to |w| s_w < 2^14 and the activations by 2^4 .. 2^8, three fp16 products reach
the error of the f32 kernel (6-9e-8) - half the MFMAs of bf16x3 for the same accuracy; fp16's range (65504 / s_a) is the price."""
import numpy as np
rng = np.random.default_rng(0)
def bf16(x):  # round to nearest even bf16, returned as f32
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)
def split_bf16x3(x):
    h = bf16(x); r = (x - h).astype(np.float32); m = bf16(r); r2 = (r - m).astype(np.float32); l = bf16(r2)
    return h, m, l
def split_fp16x2(x, scale=1.0):
    h = x.astype(np.float16).astype(np.float32); r = (x - h).astype(np.float32)
    m = (r * scale).astype(np.float16).astype(np.float32) / scale
    return h, m
def gemm_f32acc(a, b):  # exact products, f32 accumulation emulated by chunks of 16 in f32 (like MFMA k-chunks), then f32 adds
    K = a.shape[1]; acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for k0 in range(0, K, 16):
        acc = (acc + (a[:, k0:k0+16].astype(np.float64) @ b[k0:k0+16].astype(np.float64)).astype(np.float32)).astype(np.float32)
    return acc
for (M, K, N, sa, sw, name) in [(256, 2304, 64, 1.0, 1/48., "3x3x256 unit acts"), (256, 2304, 64, 0.05, 1/48., "small acts 0.05"), (256, 1024, 64, 30.0, 1/32., "large acts 30"), (256, 576, 64, 1.0, 1/24., "3x3x64")]:
    A = (rng.standard_normal((M, K)) * sa).astype(np.float32); A = np.maximum(A, 0.25 * A)  # PReLU-like
    W = (rng.standard_normal((K, N)) * sw).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64)
    den = np.abs(A).astype(np.float64) @ np.abs(W).astype(np.float64)
    e_f32 = np.abs(gemm_f32acc(A, W) - ref)
    ah, am, al = split_bf16x3(A); wh, wm, wl = split_bf16x3(W)
    out = np.zeros_like(ref, dtype=np.float32)
    for (x, y) in [(al, wh), (ah, wl), (am, wm), (am, wh), (ah, wm), (ah, wh)]:
        out = (out + gemm_f32acc(x, y)).astype(np.float32)
    e_b3 = np.abs(out - ref)
    res = {}
    for scale in (1.0, 2048.0):
        ah, am = split_fp16x2(A, scale); wh, wm = split_fp16x2(W, scale)
        out = np.zeros_like(ref, dtype=np.float32)
        for (x, y) in [(am, wh), (ah, wm), (ah, wh)]:
            out = (out + gemm_f32acc(x, y)).astype(np.float32)
        res[scale] = np.abs(out - ref)
    f = lambda e: f"max {e.max():.2e} rel-to-sum|ab| {(e / den).max():.2e}"
    print(f"{name:20s} K={K}: f32 {f(e_f32)} | bf16x3(6) {f(e_b3)} | fp16x2(3) {f(res[1.0])} | fp16x2(3, low piece scaled 2^11) {f(res[2048.0])}")

print("\n--- single accumulator, per-tensor power-of-two scales (x * Sa, w * Sw), result * 2^-(sa+sw) ---")
def split_fp16x2_scaled(x, S):
    xs = (x * S).astype(np.float32)
    h = xs.astype(np.float16).astype(np.float32); r = (xs - h).astype(np.float32)
    m = r.astype(np.float16).astype(np.float32)
    return h, m
for (M, K, N, sa, sw, name) in [(256, 2304, 64, 1.0, 1/48., "3x3x256 unit acts"), (256, 2304, 64, 0.05, 1/48., "small acts 0.05"), (256, 1024, 64, 30.0, 1/32., "large acts 30"), (256, 576, 64, 1.0, 1/24., "3x3x64"), (256, 4608, 64, 1.0, 1/68., "3x3x512")]:
    A = (rng.standard_normal((M, K)) * sa).astype(np.float32); A = np.maximum(A, 0.25 * A)
    W = (rng.standard_normal((K, N)) * sw).astype(np.float32)
    ref = A.astype(np.float64) @ W.astype(np.float64)
    den = np.abs(A).astype(np.float64) @ np.abs(W).astype(np.float64)
    e_f32 = np.abs(gemm_f32acc(A, W) - ref)
    for Sa in (2.0**4, 2.0**6, 2.0**8):
        Sw = 2.0 ** np.floor(np.log2(16384.0 / np.abs(W).max()))
        ah, am = split_fp16x2_scaled(A, Sa); wh, wm = split_fp16x2_scaled(W, Sw)
        out = np.zeros_like(ref, dtype=np.float32)
        for (x, y) in [(am, wh), (ah, wm), (ah, wh)]:
            out = (out + gemm_f32acc(x, y)).astype(np.float32)
        out = out.astype(np.float64) / (Sa * Sw)
        e = np.abs(out - ref)
        print(f"{name:20s} K={K} Sa=2^{int(np.log2(Sa))} Sw=2^{int(np.log2(Sw))}: f32 rel {(e_f32/den).max():.2e} | fp16x2(3 products, one accumulator) rel {(e/den).max():.2e}  max|A*Sa| {np.abs(A*Sa).max():.0f} (fp16 max 65504)  acc max {np.abs(out*Sa*Sw).max():.2e}")
