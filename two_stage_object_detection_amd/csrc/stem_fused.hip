// stem_fused.hip -- the ResNet stem (models/resnet.py:136-139 of the reference: conv 7x7 / 2 pad 3, 3 -> 64, BN, PReLU, max pool
// 3x3 / 2 pad 1) in ONE launch, reading the images as the reference's callers hand them over (NCHW) or as the input step writes
// them (NHWC, 4 floats per pixel).
//
// Why: as three launches (layout kernel, implicit-GEMM conv, max pool) the stem moves the 64-channel conv output (68 MB per image at
// 3x800x1333) out to HBM and back for 17 MB of pooled result, and its three launches take 81 us of a 1.48 ms batch-1 forward
// (profiles/r04_b1_serial_kernel_trace_summary.md: 48.8 + 19.1 + 12.8) / 509 us of 7.3 ms at batch 8.  Here the conv output of a tile
// lives in LDS: pixels in (13 MB per image), pooled map out (17 MB).
//
// Arithmetic: "fp16x2" as in conv_igemm_f32.hip (two fp16 pieces per operand, three products on v_mfma_f32_32x32x16_f16, f32
// accumulation).  The pixel scale is THIS TILE's own: the abs-max of the input patch, taken by the workgroup while it stages the patch
// (a GEMM needs one scale per accumulation, and every accumulation here is over pixels of one patch) - no range words for the image, no
// pass over it.
//
// A workgroup (256 threads = 4 waves, two per CU) makes 4 x 16 pooled pixels x 64 channels:
//   * conv pixels it needs: 9 x 33 = 297 (rows 2 p0 - 1 .. 2 p0 + 7: the pool's halo) = 10 blocks of 32; input patch 23 x 71 pixels.
//   * the patch goes to LDS ONCE, already split: two fp16 planes of [23 rows][72 pixels][4 channels] (8 bytes per pixel and plane;
//     channel 3 and pixel column 71 are zeros).  GEMM K = (kh, kw, c) with kw padded to 8 and c to 4 = 224 = 14 chunks of 16: the 8 k
//     of an MFMA fragment lane are two neighbouring pixels of one filter row = 16 contiguous, 16-byte aligned bytes of a plane, so the
//     im2col gather is ONE per-lane base address (conv pixel, k half) + a compile-time offset per chunk: no address arithmetic in the
//     loop.
//   * MFMA orientation as in bottleneck_fused.hip: first operand = weights (32 output channels, rows permuted by pi), second = pixels:
//     an accumulator lane holds 16 consecutive channels of ONE conv pixel.  Wave (cb, ph) owns channel block cb and the pixel blocks ph,
//     ph + 2, ..: its weight fragments (14 chunks x (hi, lo), pre-packed by the host per lane) come straight from L2 into registers, three
//     chunks ahead of their use; each feeds five pixel fragments.
//   * epilogue: BN + PReLU, -inf outside the conv output (the pool's padding), into LDS as f32 [297][64] over the patch (16-byte slots
//     XOR-ed with the pixel index: conflict-free writes and reads); then the 3x3 / 2 maxima, 256 contiguous bytes per pooled pixel out,
//     and the abs-max of what was stored into the output's range words.
#include "tsod_internal.h"
#include <stdlib.h>
#include <math.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int TPH = 4, TPW = 16;                       // pooled pixels per tile
constexpr int CH = 2 * TPH + 1, CW = 2 * TPW + 1;      // conv pixels the tile's pool windows cover
constexpr int NPX = CH * CW, MB = (NPX + 31) / 32;     // 297 conv pixels = 10 blocks of 32
constexpr int IH = 2 * CH + 5, IW = 2 * CW + 5;        // input patch
constexpr int IWP = IW + 1;                            // patch row pitch in pixels (even: fragment reads stay 16-byte aligned)
constexpr int PLANE = IH * IWP * 8;                    // one fp16 plane of the patch
constexpr int CT_PITCH = 256;                          // conv tile: 64 f32 per pixel
constexpr int CT_BYTES = NPX * CT_PITCH;
constexpr int SCR_OFF = CT_BYTES > 2 * PLANE ? CT_BYTES : 2 * PLANE, LDS_BYTES = SCR_OFF + 64;
constexpr int KCH = 14;                                // 16-k chunks: k = 32 kh + 4 kw + c
constexpr int WF_BYTES = 2 * KCH * 2 * 1024;           // [channel block][chunk][hi | lo][lane][8 k] fp16
constexpr int PF = 4;                                  // weight fragments are requested PF chunks ahead
static_assert(MB % 2 == 0 && 2 * LDS_BYTES <= 160 * 1024 && IWP % 2 == 0, "pixel blocks split over two waves; two workgroups per CU");
constexpr unsigned kOOB = 0xFFFFFFF0u;                 // byte offset beyond any buffer-descriptor extent: loads return 0

struct Params {
    const float *x;
    float *out;
    const unsigned char *wfrag;
    const float *bn;           // [scale(64) | shift(64)]
    int N, H, W, OH, OW, PH, PW, out_pitch, nchw;
    int tiles_x, tiles_y;
    unsigned x_bytes;
    float slope;
    int w_exp;
    unsigned *amax_out;
    int *range_flag;
    int dbg;                   // timing experiments only (TSOD_STEM_DBG): 1 no GEMM, 2 no BN / PReLU / conv-tile writes, 4 no pool, 8 no patch
                               // requests and staging after the first - wrong results by design, never set by the library's callers
};

__device__ __forceinline__ float prelu(float v, float a) { return fmaxf(v, 0.f) + a * fminf(v, 0.f); }

// two fp16 pieces of s * x for a pair of elements (split2_pair of conv_igemm_f32.hip)
__device__ __forceinline__ void split2(float x0, float x1, float sc, unsigned &h, unsigned &l) {
    asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h), "=&v"(l) : "v"(x0), "v"(x1), "s"(sc));
}

// three piece products, smallest first: lo*hi, hi*lo, hi*hi
__device__ __forceinline__ void mfma3(f32x16 &acc, const u32x4 &wh, const u32x4 &wl, const u32x4 &ah, const u32x4 &al) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wl), __builtin_bit_cast(f16x8, ah), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wh), __builtin_bit_cast(f16x8, al), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wh), __builtin_bit_cast(f16x8, ah), acc, 0, 0, 0);
}

__device__ __forceinline__ float block_max(float v, float *scr, int tid) {
    v = tsod_wave_max(v);
    __syncthreads();
    if ((tid & 63) == 0) scr[tid >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(scr[0], scr[1]), fmaxf(scr[2], scr[3]));
}

struct WFrag { u32x4 h, l; };

__global__ void __launch_bounds__(256, 2) stem_kernel(const Params p) {
    __shared__ __align__(16) unsigned char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hh = lane >> 5;             // fragment column (conv pixel) / weight row, k half
    const int cb = wave & 1, ph = wave >> 1;             // channel block of 32, first pixel block (stride 2)
    float *scr = reinterpret_cast<float *>(lds + SCR_OFF);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, (short)0, (int)p.x_bytes, 0x00020000);

    // ---- tiles of this workgroup: round r of the launch covers tiles [r G, (r + 1) G), G = the grid; inside a round XCD x (blocks x,
    // x + 8, ..: one private L2) takes a contiguous run (neighbouring tiles share patch pixels).  The workgroups are PERSISTENT (two per
    // CU): the next tile's patch is requested before this tile's GEMM starts, so that only the first patch of a workgroup is waited for.
    const int tiles_per_img = p.tiles_x * p.tiles_y, n_tiles = p.N * tiles_per_img;
    const int nwg = (int)gridDim.x, per = nwg >> 3, r8 = nwg & 7, xcd = (int)blockIdx.x & 7;
    const int in_round = (xcd < r8 ? xcd * (per + 1) : r8 * (per + 1) + (xcd - r8) * per) + ((int)blockIdx.x >> 3);
    struct Tile { int img, p0y, p0x; };
    auto tile_at = [&](int t) {
        Tile tl;
        tl.img = t / tiles_per_img;
        const int t_in = t - tl.img * tiles_per_img, ty = t_in / p.tiles_x;
        tl.p0y = ty * TPH;
        tl.p0x = (t_in - ty * p.tiles_x) * TPW;
        return tl;
    };

    // ---- this wave's weight fragments: chunk c at wf_base + c * 2048 (hi, then lo 1024 bytes on)
    // (buffer loads: one per-lane offset register, the chunk's offset is a scalar operand - 14 flat addresses would cost 28 registers)
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)p.wfrag, (short)0, WF_BYTES, 0x00020000);
    const unsigned wf_off = (unsigned)(cb * (KCH * 2048) + lane * 16);
    auto wload = [&](int c) {
        WFrag f;
        f.h = __builtin_amdgcn_raw_buffer_load_b128(rs_w, wf_off, c * 2048, 0);
        f.l = __builtin_amdgcn_raw_buffer_load_b128(rs_w, wf_off, c * 2048 + 1024, 0);
        return f;
    };
    WFrag wf[PF + 1];
#pragma unroll
    for (int c = 0; c < PF; ++c) wf[c] = wload(c);

    // ---- the input patch: IH x IWP pixel slots, three channels each, into registers; zeros outside the image and in the pad column
    constexpr int SLOTS = IH * IWP, PER = (SLOTS + 255) / 256;
    float px[PER][3];
    auto request_patch = [&](const Tile &tl) {
        const int i0y = 4 * tl.p0y - 5, i0x = 4 * tl.p0x - 5;    // conv origin 2 p0 - 1 (the pool's padding), input origin 2 c0 - 3
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int q = tid + 256 * i, iy = q / IWP, ix = q - iy * IWP;
            const int gy = i0y + iy, gx = i0x + ix;
            const bool ok = q < SLOTS && ix < IW && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
            if (p.nchw) {
                const unsigned o = (unsigned)((((long)tl.img * 3) * p.H + gy) * p.W + gx) * 4u, plane = (unsigned)(p.H * p.W) * 4u;
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    px[i][c] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_x, ok ? o + (unsigned)c * plane : kOOB, 0, 0));
            } else {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? (unsigned)(((long)tl.img * p.H + gy) * p.W + gx) * 16u : kOOB, 0, 0);
                px[i][0] = __uint_as_float(v.x); px[i][1] = __uint_as_float(v.y); px[i][2] = __uint_as_float(v.z);
            }
        }
    };
    // LDS offset of this lane's conv pixels at tap (0, 0), k half included (the same for every tile)
    constexpr int NB = MB / 2;
    int pa[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        int m = (ph + 2 * b) * 32 + j;
        m = m < NPX ? m : NPX - 1;                       // (rows past the tile compute a duplicate, never stored)
        const int cr = m / CW, cc = m - cr * CW;
        pa[b] = ((2 * cr) * IWP + 2 * cc) * 8 + hh * 16;
    }
    const int ch16 = cb * 32 + 16 * hh;                  // this lane's 16 output channels
    float chk = 0.f, amax = 0.f;

    int t = in_round;
    if (t < n_tiles) request_patch(tile_at(t));
    for (; t < n_tiles; t += nwg) {
        const Tile tl = tile_at(t);
        const int c0y = 2 * tl.p0y - 1, c0x = 2 * tl.p0x - 1;
        // (opaque copies: what the phases below derive from the thread index - slot addresses, BN vectors, pool offsets - is cheap to
        //  recompute and must not be hoisted out of the tile loop into ~100 registers that then spill)
        int tid_l = tid, ch16_l = ch16;
        asm volatile("" : "+v"(tid_l), "+v"(ch16_l));
        // ---- the patch: tile-wide abs-max -> scale, two fp16 pieces into LDS
        float mx = 0.f;
#pragma unroll
        for (int i = 0; i < PER; ++i) mx = fmaxf(mx, fmaxf(fabsf(px[i][0]), fmaxf(fabsf(px[i][1]), fabsf(px[i][2]))));
        // (a NaN pixel is dropped by fmaxf here and poisons its accumulators below: the range flag reports it)
        mx = block_max(mx, scr, tid);                    // (its first barrier: the previous tile's pool reads are done with the LDS)
        const int e_a = tsod_fp16x2_exp_from_bits(__float_as_uint(mx));
        const float a_scale = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((127 + e_a) << 23));
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int q = tid_l + 256 * i;
            if (q < SLOTS) {
                unsigned h0, l0, h1, l1;
                split2(px[i][0], px[i][1], a_scale, h0, l0);
                split2(px[i][2], 0.f, a_scale, h1, l1);
                *reinterpret_cast<u32x2 *>(lds + q * 8) = u32x2{h0, h1};
                *reinterpret_cast<u32x2 *>(lds + PLANE + q * 8) = u32x2{l0, l1};
            }
        }
        if (t + nwg < n_tiles && !(p.dbg & 8)) request_patch(tile_at(t + nwg));   // lands while this tile is computed
        __syncthreads();

        // ---- GEMM: acc[b] = conv pixels of block ph + 2 b (columns) x channels 32 cb + pi(rows), K = 224
        f32x16 acc[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
        struct AFrag { u32x4 h, l; };
        auto aload = [&](int c, int b) {
            // chunk c = 2 t + u: k = 16 c + 8 hh .. = filter row t, pixels 2 (hh + 2 u), + 1
            const int off = (c >> 1) * (IWP * 8) + (c & 1) * 32;
            AFrag f;
            f.h = *reinterpret_cast<const u32x4 *>(lds + pa[b] + off);
            f.l = *reinterpret_cast<const u32x4 *>(lds + pa[b] + off + PLANE);
            return f;
        };
        AFrag af[2];
        af[0] = aload(0, 0);
        if (!(p.dbg & 1))
#pragma unroll
        for (int c = 0; c < KCH; ++c) {
            // (past the last chunk: the first fragments again, for this workgroup's next tile)
            wf[(c + PF) % (PF + 1)] = wload(c + PF < KCH ? c + PF : c + PF - KCH);
            const WFrag &w = wf[c % (PF + 1)];
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int n = c * NB + b;
                if (n + 1 < KCH * NB) af[(n + 1) & 1] = aload((n + 1) / NB, (n + 1) % NB);
                __builtin_amdgcn_sched_barrier(0);       // (the next fragment's reads and the weight requests stay AHEAD of these MFMAs)
                mfma3(acc[b], w.h, w.l, af[n & 1].h, af[n & 1].l);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // (the fragments requested past the last chunk are the next tile's first: into the slots it starts from)
        if constexpr (KCH % (PF + 1) != 0) {
            WFrag nx[PF];
#pragma unroll
            for (int i = 0; i < PF; ++i) nx[i] = wf[(KCH + i) % (PF + 1)];
#pragma unroll
            for (int i = 0; i < PF; ++i) wf[i] = nx[i];
        }
        __syncthreads();                                 // every wave is done with the patch: the conv tile goes over it

        // ---- BN + PReLU, -inf outside the conv output, into the conv tile: lane = pixel m, channels ch16 + 0..15
        if (!(p.dbg & 2)) {
            const float sc = __uint_as_float((unsigned)(127 - e_a - p.w_exp) << 23);
            float sv[16], bv[16];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float4 s4 = *reinterpret_cast<const float4 *>(p.bn + ch16_l + 4 * v), b4 = *reinterpret_cast<const float4 *>(p.bn + 64 + ch16_l + 4 * v);
                sv[4 * v] = s4.x * sc; sv[4 * v + 1] = s4.y * sc; sv[4 * v + 2] = s4.z * sc; sv[4 * v + 3] = s4.w * sc;
                bv[4 * v] = b4.x; bv[4 * v + 1] = b4.y; bv[4 * v + 2] = b4.z; bv[4 * v + 3] = b4.w;
            }
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int m = (ph + 2 * b) * 32 + j, cr = m / CW, cc = m - cr * CW;
                const bool inside = (unsigned)(c0y + cr) < (unsigned)p.OH && (unsigned)(c0x + cc) < (unsigned)p.OW;
                float v[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    chk = fmaf(acc[b][e], 0.f, chk);     // (NaN once any accumulator is inf / NaN: PReLU's max / min would hide it)
                    v[e] = inside ? prelu(fmaf(acc[b][e], sv[e], bv[e]), p.slope) : -INFINITY;
                }
                if (m < NPX) {
                    unsigned char *row = lds + m * CT_PITCH;
                    const int s0 = ch16_l >> 2, sw = m & 15;
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        *reinterpret_cast<float4 *>(row + (((s0 + q) ^ sw) << 4)) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
                }
            }
        }
        __syncthreads();

        // ---- 3x3 / 2 maxima: thread = (pooled column g, channel slot s), the tile's TPH rows in turn
        if (!(p.dbg & 4)) {
            const int s = tid_l & 15, g = tid_l >> 4;
            const int gx = tl.p0x + g;
#pragma unroll
            for (int py = 0; py < TPH; ++py) {
                const int gy = tl.p0y + py;
                float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int m = (2 * py + dy) * CW + 2 * g + dx;
                        const float4 v = *reinterpret_cast<const float4 *>(lds + m * CT_PITCH + ((s ^ (m & 15)) << 4));
                        best.x = fmaxf(best.x, v.x); best.y = fmaxf(best.y, v.y); best.z = fmaxf(best.z, v.z); best.w = fmaxf(best.w, v.w);
                    }
                if (gy < p.PH && gx < p.PW) {
                    *reinterpret_cast<float4 *>(p.out + (((size_t)tl.img * p.PH + gy) * p.PW + gx) * p.out_pitch + 4 * s) = best;
                    amax = fmaxf(amax, fmaxf(fmaxf(fabsf(best.x), fabsf(best.y)), fmaxf(fabsf(best.z), fabsf(best.w))));
                }
            }
        }
    }
    if (p.range_flag != nullptr && __any(!(chk == 0.f)) && lane == 0) atomicOr(p.range_flag, 1);
    if (p.amax_out != nullptr) tsod_amax_commit(p.amax_out, amax, scr, tid, 256);
}

}  // namespace

extern "C" size_t tsod_stem_wfrag_bytes(void) { return WF_BYTES; }

extern "C" int tsod_stem_fp16x2(const tsod_stem_desc *d, const float *x, const void *wfrag, const float *bn, float *out, tsod_stream_t stream) {
    TSOD_REQUIRE(d && x && wfrag && bn && out, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(d->in_layout == TSOD_STEM_NCHW || d->in_layout == TSOD_STEM_NHWC4, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(d->out_pitch >= 64 && (d->out_pitch & 3) == 0, TSOD_ERR_ALIGNMENT);
    // (NCHW images are read with 4-byte loads: a batch slice x[i:i+1] of a caller's tensor is as good as the tensor; NHWC4 pixels are 16-byte loads)
    TSOD_REQUIRE(d->in_layout == TSOD_STEM_NCHW ? (reinterpret_cast<uintptr_t>(x) & 3u) == 0 : tsod_aligned16(x), TSOD_ERR_ALIGNMENT);
    TSOD_REQUIRE(tsod_aligned16(out) && tsod_aligned16(wfrag) && tsod_aligned16(bn), TSOD_ERR_ALIGNMENT);
    TSOD_REQUIRE((reinterpret_cast<uintptr_t>(d->amax_out) & 63u) == 0, TSOD_ERR_ALIGNMENT);
    const uint64_t in_bytes = (uint64_t)d->N * d->H * d->W * (d->in_layout == TSOD_STEM_NCHW ? 12 : 16);
    TSOD_REQUIRE(in_bytes < 0xFFFFFFF0ull, TSOD_ERR_UNSUPPORTED);            // 32-bit byte offsets into x
    Params p;
    p.x = x; p.out = out; p.wfrag = static_cast<const unsigned char *>(wfrag); p.bn = bn;
    p.N = d->N; p.H = d->H; p.W = d->W;
    p.OH = (d->H - 1) / 2 + 1; p.OW = (d->W - 1) / 2 + 1;                    // 7x7 / 2, pad 3
    p.PH = (p.OH - 1) / 2 + 1; p.PW = (p.OW - 1) / 2 + 1;                    // 3x3 / 2, pad 1
    p.out_pitch = d->out_pitch; p.nchw = d->in_layout == TSOD_STEM_NCHW;
    p.tiles_x = (p.PW + TPW - 1) / TPW; p.tiles_y = (p.PH + TPH - 1) / TPH;
    p.x_bytes = (unsigned)in_bytes;
    p.slope = d->slope; p.w_exp = d->w_exp;
    p.amax_out = d->amax_out; p.range_flag = d->range_flag;
    const int64_t tiles = (int64_t)d->N * p.tiles_x * p.tiles_y;
    TSOD_REQUIRE(tiles < 0x7FFFFFFF, TSOD_ERR_UNSUPPORTED);
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
        (void)hipGetLastError();
        cus = 256;
    }
    // persistent workgroups, two per CU; every workgroup gets the same number of tiles when that is possible (a grid of ceil(tiles / rounds))
    const int64_t slots = 2 * (int64_t)cus, rounds = (tiles + slots - 1) / slots;
    const int64_t grid = (tiles + rounds - 1) / rounds;
    static const int dbg = [] { const char *e = getenv("TSOD_STEM_DBG"); return e ? atoi(e) : 0; }();
    p.dbg = dbg;
    hipLaunchKernelGGL(stem_kernel, dim3((unsigned)grid), dim3(256), 0, tsod_stream(stream), p);
    return tsod_launch_status();
}
