// bottleneck_fused.hip -- a whole ResNet bottleneck (models/resnet.py:57-76, stride 1, identity shortcut) in ONE launch:
//
//     y1  = PReLU(BN1(conv1x1(x)))          Cin  -> 64      on the tile's pixels + a one-pixel halo
//     y2  = PReLU(BN2(conv3x3(y1)))         64   -> 64      from LDS (y1 never leaves the CU)
//     out = PReLU(BN3(conv1x1(y2)) + x)     64   -> Cout    (Cout == Cin: the identity blocks of layer1)
// or, PROJ (models/resnet.py:114-116, the first block of layer1: a 1x1 projection shortcut at stride 1),
//     out = PReLU(BN3(conv1x1(y2)) + BNd(conv1x1_d(x)))             as ONE stacked-K GEMM [y2 | x] . [W3 s3 | Wd sd]^T + (b3 + bd):
//     the y2 chunks come from LDS, the x chunks of the tile's own pixels straight from L2 into MFMA fragments (the workgroup read
//     them for conv1 a few microseconds earlier); no residual pass.
//
// Why: the three launches of such a block move 273 MB per image at 3x800x1333 (x read by conv1, y1 written and read, y2 written
// and read, x read again as the residual, out written) for 3.6 GFLOP - layer1 is the one HBM-bound stage of the trunk
// (profiles/r03_b8_pmc_summary.md rows 4-9: 5.5 TB/s).  Here y1 and y2 live in LDS: x (+ halo) in, x again (residual: an L2 / Infinity
// Cache hit, the same workgroup read it a few microseconds earlier), out out.
//
// Arithmetic: "fp16x2" as in conv_igemm_f32.hip (two fp16 pieces per operand, three products on v_mfma_f32_32x32x16_f16, f32
// accumulation): x is split with the scale its range words give (or the static exponent), y1 and y2 with the scale THIS TILE's own
// abs-max calls for (a workgroup-wide max through LDS: a GEMM only needs one scale per accumulation, and every accumulation here
// is over values of one tile) - no range words for tensors that never exist.
//
// Layout of a workgroup (256 threads = 4 waves), output tile TH x TW = 10 x 16 pixels (160 = 5 blocks of 32), halo 12 x 18 =
// 216 -> 224 rows (7 blocks):
//   * MFMA orientation: first operand = WEIGHTS (32 output channels), second = ACTIVATIONS (32 pixels): an accumulator lane then
//     holds 16 values of ONE pixel, and with the weight rows permuted (pi below) they are 16 CONSECUTIVE channels: 64 contiguous
//     bytes per lane for the residual loads and output stores, two ds_write_b128 per plane for y1 / y2.
//   * weights: ALL three convs as one stream of 8 KB steps (64 output channels x 32 k x (hi, lo)), pre-packed by the host in
//     consumption order and in the exact LDS image (row permutation and bank swizzle included): the loader is a plain copy through
//     a two-slot ring.  Cin = 256: 8 + 18 + 8 = 34 steps.
//   * conv1's activations: the x halo tile, one 32-channel K-step at a time through LDS (split into pieces on the way in);
//     conv2's: y1 planes in LDS, the im2col gather is the fragment's row address; conv3's: y2 planes (the y1 region re-used).
//   * a wave owns one 32-channel block of the step's 64 and every second pixel block: ONE weight fragment pair per 16-k chunk
//     feeds up to four activation fragments.
#include "tsod_internal.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Tile: TH x 16 output pixels, TH = 10 (160 pixels = 5 blocks of 32, halo 12 x 18 = 216 -> 224 rows = 7 blocks) or 8 (128 = 4 blocks,
// halo 180 -> 192 rows = 6 blocks).  10 rows make one round of a batch-1 launch (420 tiles on 512 slots); 8 rows split evenly over the
// waves (3 + 3 halo blocks, 2 + 2 tile blocks: the slowest wave issues 456 MFMAs for 128 pixels instead of 660 for 160) and win
// from two images on.  The launcher picks by rounds x MFMAs of the slowest wave.
constexpr int TW = 16, HW_ = TW + 2;
constexpr int CMID = 64;
// y planes: [rows][64 channels] fp16 at a pitch of 144 bytes (9 x 16: the 16 rows of a ds_read_b128 lane group land in 16 distinct
// 16-byte bank groups without an XOR, so a fragment address is ONE per-lane base + an immediate offset - tap shift, channel slot and
// plane are all compile-time in the unrolled loops)
constexpr int Y_PITCH = 144;
constexpr int W_STEP = 8192;                     // one weight step: 64 output channels x 32 k x (hi, lo)
template <int TH> struct Geo {
    static constexpr int HALO = (TH + 2) * (TW + 2);             // halo pixels
    static constexpr int PB1 = (HALO + 31) / 32, PB2 = TH * TW / 32;     // pixel blocks of 32: halo, tile
    static constexpr int Y_PLANE = PB1 * 32 * Y_PITCH;           // one fp16 plane of y1
    static constexpr int A_PLANE = PB1 * 32 * 64, A_STAGE = 2 * A_PLANE;   // conv1's x stage: two fp16 planes of [rows][32 k], XOR-swizzled
    static constexpr int Y_BYTES = 2 * Y_PLANE, SCR_OFF = Y_BYTES, LDS_BYTES = SCR_OFF + 64;
    static_assert(TH * TW % 32 == 0 && 2 * A_STAGE <= Y_BYTES && 2 * LDS_BYTES <= 160 * 1024, "x stages inside the y region; two workgroups per CU");
};
constexpr unsigned kOOB = 0xFFFFFFF0u;           // byte offset beyond any buffer-descriptor extent: loads return 0, stores are dropped

struct Params {
    const float *x;            // [N][H][W][in_pitch]
    float *out;                // [N][H][W][out_pitch]
    const unsigned char *wstream;
    const float *bn;           // [s1(64) | b1(64) | s2(64) | b2(64) | s3(Cout) | b3(Cout)]
    int N, H, W, Cin, in_pitch, Cout, out_pitch;
    int tiles_x, tiles_y;
    unsigned x_bytes, out_bytes;
    float slope;
    int w_exp1, w_exp2, w_exp3;
    float a_scale;             // static 2^a_scale_exp (no range words)
    const unsigned *amax_in;
    unsigned *amax_out;
    int *range_flag;
    int dbg;                   // timing experiments only (TSOD_BN_DBG): 1 no x loads after the first, 2 no residual loads, 4 no weight loads after
                               // the first two steps, 8 no output stores - wrong results by design, never set by the library's callers
};

__device__ __forceinline__ float prelu(float v, float a) { return fmaxf(v, 0.f) + a * fminf(v, 0.f); }

// two fp16 pieces of s * x for a pair of elements (split2_pair of conv_igemm_f32.hip: four mixed-precision FMAs; `sc` wave-uniform)
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

// byte offset of 16-byte slot `slot` of row `row` in the XOR-swizzled 64-byte-row x stage (4 slots per row, 4 rows per bank line)
__device__ __forceinline__ int off64(int row, int slot) { return row * 64 + ((slot ^ ((row >> 2) & 3)) << 4); }

// workgroup-wide max of a non-negative value through LDS (two barriers); every thread gets it
__device__ __forceinline__ float block_max(float v, float *scr, int tid) {
    v = tsod_wave_max(v);
    __syncthreads();
    if ((tid & 63) == 0) scr[tid >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(scr[0], scr[1]), fmaxf(scr[2], scr[3]));
}

__device__ __forceinline__ float4 bload4(__amdgpu_buffer_rsrc_t r, unsigned off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// the weight fragments of one step for one wave: chunk c (16 k): hi, lo
struct WFrag { u32x4 h[2], l[2]; };
struct AFrag { u32x4 h, l; };

template <int TH, bool PROJ = false>
__global__ void __launch_bounds__(256, 2) bottleneck_kernel(const Params p) {
    using G = Geo<TH>;
    constexpr int HALO = G::HALO, PB1 = G::PB1, PB2 = G::PB2, Y_PLANE = G::Y_PLANE, A_PLANE = G::A_PLANE, A_STAGE = G::A_STAGE,
                  SCR_OFF = G::SCR_OFF, LDS_BYTES = G::LDS_BYTES;
    __shared__ __align__(16) unsigned char lds[LDS_BYTES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, hh = lane >> 5;             // fragment column (pixel) / row (weight row) index, k half
    const int cb = wave & 1, pb0 = wave >> 1;            // this wave's channel block of the step's 64, its first pixel block (stride 2)
    // pixel blocks this wave owns (pb0, pb0 + 2, ...): of the halo's PB1, of the tile's PB2
    const int nb1 = (PB1 - pb0 + 1) / 2, nb2 = (PB2 - pb0 + 1) / 2;
    float *scr = reinterpret_cast<float *>(lds + SCR_OFF);

    // ---- the tile.  Blocks b, b + 8, ... share an XCD (private L2; observed round-robin placement - a speed matter only): XCD x takes
    // a contiguous run of the tile order, so that neighbouring tiles - which share halo pixels - are read through ONE L2
    const int tiles_per_img = p.tiles_x * p.tiles_y;
    const int nwg = (int)gridDim.x, per = nwg >> 3, r8 = nwg & 7, xcd = (int)blockIdx.x & 7;
    const int tile_id = (xcd < r8 ? xcd * (per + 1) : r8 * (per + 1) + (xcd - r8) * per) + ((int)blockIdx.x >> 3);
    const int img = tile_id / tiles_per_img, t_in = tile_id - img * tiles_per_img;
    const int ty = t_in / p.tiles_x, tx = t_in - ty * p.tiles_x;
    const int h0 = ty * TH, w0 = tx * TW;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)p.x, (short)0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void *)p.out, (short)0, (int)((p.dbg & 8) ? 0u : p.out_bytes), 0x00020000);

    // ---- x scale: the static exponent, or what the input's range words call for (wave-uniform)
    float a_scale = p.a_scale;
    int e_x;
    if (p.amax_in != nullptr) {
        e_x = tsod_fp16x2_exp_from_bits(tsod_amax_reduce_bits(p.amax_in[lane * TSOD_AMAX_STRIDE_WORDS]));
        a_scale = __uint_as_float((unsigned)(127 + e_x) << 23);
    } else {
        e_x = (int)(__float_as_uint(a_scale) >> 23) - 127;
    }
    a_scale = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(a_scale)));

    // ---- weight stream: step s = 8 KB = [channel block cb][lane][chunk 0 hi | chunk 0 lo | chunk 1 hi | chunk 1 lo]: a lane's four
    // fragments of a step are 64 contiguous bytes, a wave's 4 KB - straight from L2 into registers (no LDS ring: no barrier per
    // step, and the weights are the same 272 KB for every workgroup of the launch: L2-resident), two steps ahead of their use
    // steps of one 64-channel slice of conv3: two over y2 and, PROJ, Cin / 32 more over x (the stacked projection shortcut)
    const int n_steps1 = p.Cin / 32, sps = PROJ ? 2 + n_steps1 : 2, n_steps = n_steps1 + 18 + (p.Cout / 64) * sps;
    const u32x4 *wbase = reinterpret_cast<const u32x4 *>(p.wstream + cb * (W_STEP / 2) + lane * 64);
    auto w_load = [&](int s, WFrag &f) {
        if (s >= n_steps || ((p.dbg & 4) && s > 2)) return;
        const u32x4 *src = wbase + (size_t)s * (W_STEP / 16);
        f.h[0] = src[0]; f.l[0] = src[1]; f.h[1] = src[2]; f.l[1] = src[3];
    };
    WFrag wf[3];                                         // steps s, s + 1, s + 2 rotate through these (compile-time indices only)
    w_load(0, wf[0]);
    w_load(1, wf[1]);

    // ---- conv1's x stage: PB1 * 32 rows x 8 float4 per K-step = PB1 float4 per thread; unit u = tid + 256 i: row = u >> 3, quad = u & 7
    constexpr int NX = PB1;
    unsigned xoff[NX];                                  // byte offset of the unit in x at K-step 0 (kOOB: outside the image / padding row)
    int xdst[NX];                                       // byte offset of the unit's 8 bytes inside an x-stage plane
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int u = tid + 256 * i, row = u >> 3, q = u & 7;
        const int hr = row / HW_, hc = row - hr * HW_;
        const int gh = h0 - 1 + hr, gw = w0 - 1 + hc;
        const bool ok = row < HALO && (unsigned)gh < (unsigned)p.H && (unsigned)gw < (unsigned)p.W;
        xoff[i] = ok ? (unsigned)((((long)img * p.H + gh) * p.W + gw) * p.in_pitch + q * 4) * 4u : kOOB;
        xdst[i] = off64(row, q >> 1) + (q & 1) * 8;
    }
    float4 xr[NX];
    auto x_load = [&](int ks) {
        if (ks >= n_steps1 || ((p.dbg & 1) && ks > 0)) return;
#pragma unroll
        for (int i = 0; i < NX; ++i) xr[i] = bload4(rs_x, xoff[i] != kOOB ? xoff[i] + (unsigned)ks * 128u : kOOB);
    };
    auto x_store = [&](int stage) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            unsigned h0_, l0_, h1_, l1_;
            split2(xr[i].x, xr[i].y, a_scale, h0_, l0_);
            split2(xr[i].z, xr[i].w, a_scale, h1_, l1_);
            *reinterpret_cast<uint2 *>(lds + stage * A_STAGE + xdst[i]) = make_uint2(h0_, h1_);
            *reinterpret_cast<uint2 *>(lds + stage * A_STAGE + A_PLANE + xdst[i]) = make_uint2(l0_, l1_);
        }
    };
    // x fragments of the wave's pixel blocks: row (pb0 + 2 b) * 32 + j, chunk c
    int xfa[4][2];                                       // (LDS byte offsets, not pointers: a pointer into LDS costs two registers)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int c = 0; c < 2; ++c) xfa[b][c] = off64((pb0 + 2 * b) * 32 + j, 2 * c + hh);

    // ================= conv1: y1[224 x 64] = x_halo[224 x Cin] . W1^T =================
    f32x16 acc[4];
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
    x_load(0);
    x_store(0);
    x_load(1);
    __syncthreads();
    // step ks: x stage (ks & 1) -> MFMAs, while stage (ks + 1) & 1 is filled from the registers (loaded one step ago) and the loads of
    // step ks + 2 go out; ONE barrier per step.  The weights of steps ks, ks + 1, ks + 2 sit in wf[0], wf[1], wf[2]: a step ends by
    // moving them down one place (register moves; a rotation by index would need the loop unrolled by six, which the register
    // allocator answers with 250+ spilled registers).
    auto conv1_step = [&](auto STAGE, int ks) {
        constexpr int stage = decltype(STAGE)::value;
        w_load(ks + 2, wf[2]);
        if (ks + 1 < n_steps1) x_store(stage ^ 1);
        x_load(ks + 2);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if (b < nb1) {
                    const u32x4 ah = *reinterpret_cast<const u32x4 *>(lds + xfa[b][c] + stage * A_STAGE);
                    const u32x4 al = *reinterpret_cast<const u32x4 *>(lds + xfa[b][c] + stage * A_STAGE + A_PLANE);
                    mfma3(acc[b], wf[0].h[c], wf[0].l[c], ah, al);
                }
            }
        }
        wf[0] = wf[1];
        wf[1] = wf[2];
        __syncthreads();
    };
    for (int ks = 0; ks < n_steps1; ks += 2) {           // (Cin is a multiple of 64: an even number of steps)
        conv1_step(std::integral_constant<int, 0>{}, ks);
        conv1_step(std::integral_constant<int, 1>{}, ks + 1);
    }
    // (wf[0], wf[1] now hold conv2's first two steps)
    // ---- y1 = PReLU(BN1(.)), zero outside the image (conv2's zero padding pads y1, not x), tile-wide scale, pieces into LDS
    // acc[b][e]: pixel row (pb0 + 2 b) * 32 + j of the halo, channel cb * 32 + 16 hh + e
    const int ch16 = cb * 32 + 16 * hh;
    const float sc1 = __uint_as_float((unsigned)(127 - e_x - p.w_exp1) << 23);
    float mx = 0.f, chk = 0.f;
    {
        float sv[16], bv[16];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float4 s4 = *reinterpret_cast<const float4 *>(p.bn + ch16 + 4 * v), b4 = *reinterpret_cast<const float4 *>(p.bn + 64 + ch16 + 4 * v);
            sv[4 * v] = s4.x * sc1; sv[4 * v + 1] = s4.y * sc1; sv[4 * v + 2] = s4.z * sc1; sv[4 * v + 3] = s4.w * sc1;
            bv[4 * v] = b4.x; bv[4 * v + 1] = b4.y; bv[4 * v + 2] = b4.z; bv[4 * v + 3] = b4.w;
        }
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            if (b >= nb1) continue;
            const int row = (pb0 + 2 * b) * 32 + j, hr = row / HW_, hc = row - hr * HW_;
            const bool ok = row < HALO && (unsigned)(h0 - 1 + hr) < (unsigned)p.H && (unsigned)(w0 - 1 + hc) < (unsigned)p.W;
            const float keep = ok ? 1.f : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                chk = fmaf(acc[b][e], 0.f, chk);                 // (NaN once any accumulator is inf / NaN: PReLU's max / min would hide it)
                const float v = keep * prelu(fmaf(acc[b][e], sv[e], bv[e]), p.slope);
                acc[b][e] = v;
                mx = fmaxf(mx, fabsf(v));
            }
        }
    }
    mx = block_max(mx, scr, tid);                        // (its barriers also order the x stages' last reads before the y1 writes)
    const int e1 = tsod_fp16x2_exp_from_bits(__float_as_uint(mx));
    const float ys1 = __uint_as_float((unsigned)(127 + e1) << 23);
    auto y_store = [&](const f32x16 &v, int row, float ysc) {
        unsigned hq[8], lq[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) split2(v[2 * e], v[2 * e + 1], ysc, hq[e], lq[e]);
        unsigned char *dst = lds + (row * Y_PITCH + (4 * cb + 2 * hh) * 16);
        *reinterpret_cast<u32x4 *>(dst) = u32x4{hq[0], hq[1], hq[2], hq[3]};
        *reinterpret_cast<u32x4 *>(dst + 16) = u32x4{hq[4], hq[5], hq[6], hq[7]};
        *reinterpret_cast<u32x4 *>(dst + Y_PLANE) = u32x4{lq[0], lq[1], lq[2], lq[3]};
        *reinterpret_cast<u32x4 *>(dst + Y_PLANE + 16) = u32x4{lq[4], lq[5], lq[6], lq[7]};
    };
#pragma unroll
    for (int b = 0; b < 4; ++b)
        if (b < nb1) y_store(acc[b], (pb0 + 2 * b) * 32 + j, ys1);
    __syncthreads();

    // ================= conv2: y2[160 x 64] = im2col(y1)[160 x 576] . W2^T (18 steps: tap, channel half) =================
    // No barrier inside: y1 is read-only, the weights come from registers.  The fragments of chunk n + 1 are read while chunk n's MFMAs
    // run (two register sets), the weights of step t + 2 are requested at step t.
    int yfa[3];                                          // this lane's output pixels: LDS offset of the halo row at tap (0, 0), k half included
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        int pidx = (pb0 + 2 * b) * 32 + j;
        pidx = pidx < TH * TW ? pidx : TH * TW - 1;      // (rows of a block past the tile compute a duplicate, never stored)
        yfa[b] = ((pidx / TW) * HW_ + (pidx % TW)) * Y_PITCH + hh * 16;
    }
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
    const int s2base = n_steps1;
    AFrag af[2][3];
    auto y1_frags = [&](int n, AFrag (&f)[3]) {          // chunk n = 2 t + c of conv2: tap t >> 1, channels 32 (t & 1) + 16 c
        const int t = n >> 1, c = n & 1, tap = t >> 1, kh = tap / 3, kw = tap - kh * 3;
        const int imm = (kh * HW_ + kw) * Y_PITCH + (t & 1) * 64 + c * 32;            // an immediate after unrolling
#pragma unroll
        for (int b = 0; b < 3; ++b)
            if (b < nb2) {
                f[b].h = *reinterpret_cast<const u32x4 *>(lds + yfa[b] + imm);
                f[b].l = *reinterpret_cast<const u32x4 *>(lds + yfa[b] + imm + Y_PLANE);
            }
    };
    y1_frags(0, af[0]);
#pragma unroll
    for (int t = 0; t < 18; ++t) {
        w_load(s2base + t + 2, wf[(t + 2) % 3]);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int n = 2 * t + c;
            if (n + 1 < 36) y1_frags(n + 1, af[(n + 1) & 1]);
#pragma unroll
            for (int b = 0; b < 3; ++b)
                if (b < nb2) mfma3(acc[b], wf[t % 3].h[c], wf[t % 3].l[c], af[n & 1][b].h, af[n & 1][b].l);
            __builtin_amdgcn_sched_barrier(0);           // (the unrolled loop is one basic block: keep the scheduler from hoisting every
                                                         //  later chunk's loads to the top - 400+ live registers, scratch spills)
        }
    }
    // (18 steps: conv3's first two steps sit in slots 0 and 1 again)
    __syncthreads();                                     // every wave is done reading y1: the region takes y2
    // ---- y2 = PReLU(BN2(.)) -> pieces into the y region
    const float sc2 = __uint_as_float((unsigned)(127 - e1 - p.w_exp2) << 23);
    mx = 0.f;
    {
        float sv[16], bv[16];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const float4 s4 = *reinterpret_cast<const float4 *>(p.bn + 128 + ch16 + 4 * v), b4 = *reinterpret_cast<const float4 *>(p.bn + 192 + ch16 + 4 * v);
            sv[4 * v] = s4.x * sc2; sv[4 * v + 1] = s4.y * sc2; sv[4 * v + 2] = s4.z * sc2; sv[4 * v + 3] = s4.w * sc2;
            bv[4 * v] = b4.x; bv[4 * v + 1] = b4.y; bv[4 * v + 2] = b4.z; bv[4 * v + 3] = b4.w;
        }
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            if (b >= nb2) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                chk = fmaf(acc[b][e], 0.f, chk);
                const float v = prelu(fmaf(acc[b][e], sv[e], bv[e]), p.slope);
                acc[b][e] = v;
                mx = fmaxf(mx, fabsf(v));
            }
        }
    }
    mx = block_max(mx, scr, tid);
    const int e2 = tsod_fp16x2_exp_from_bits(__float_as_uint(mx));
    const float ys2 = __uint_as_float((unsigned)(127 + e2) << 23);
#pragma unroll
    for (int b = 0; b < 3; ++b)
        if (b < nb2) y_store(acc[b], (pb0 + 2 * b) * 32 + j, ys2);
    __syncthreads();

    // ================= conv3: out[160 x Cout] = y2[160 x 64] . W3^T + x, 64 output channels (two weight steps) at a time ======
    // No barrier inside either (y2 is read-only, weights from registers); y2 fragments are re-read per slice, chunk n + 1 under chunk
    // n's MFMAs like conv2's.
    const float sc3 = __uint_as_float((unsigned)(127 - e2 - p.w_exp3) << 23);
    float amax = 0.f;
    // Epilogue layout.  An accumulator lane holds 16 channels of ONE pixel: stored from there, every 16-byte piece of an instruction
    // lands in a 64-byte segment of its own (measured: the stores cost 15 us of a 73 us tile at batch 1 - four times the L2 requests
    // of a coalesced pass).  So each 32 pixel x 32 channel block goes through a wave-private LDS patch (the y region has two gaps of
    // 9 KB behind y2's 160 rows: two patches of 32 x 144 bytes each) and comes back with 8 lanes per pixel: residual loads and output
    // stores are whole 128-byte row segments, and the BatchNorm vectors are one float4 pair per lane.
    constexpr int P_PITCH = 144, P_BYTES = 32 * P_PITCH;
    static_assert(2 * P_BYTES <= Y_PLANE - TH * TW * Y_PITCH, "two patches per gap of the y region");
    unsigned char *patch = lds + (wave >> 1) * Y_PLANE + TH * TW * Y_PITCH + (wave & 1) * P_BYTES;
    const int ep_px = lane >> 3, ep_q = lane & 7;        // this lane's pixel (+ 8 i) and 4-channel piece of a block in the coalesced layout
    int y2a[3];
    unsigned xpix[3][4], opix[3][4];                     // byte offsets of the lane's piece in x and in out (kOOB: not a pixel of the image)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const int pidx = (pb0 + 2 * b) * 32 + j;
        y2a[b] = (pidx < TH * TW ? pidx : TH * TW - 1) * Y_PITCH + hh * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pe = (pb0 + 2 * b) * 32 + ep_px + 8 * i, pr = pe >> 4, pc = pe & 15;      // (TW == 16)
            const bool ok = b < nb2 && pe < TH * TW && h0 + pr < p.H && w0 + pc < p.W;
            const long gp = ((long)img * p.H + h0 + pr) * p.W + w0 + pc;
            xpix[b][i] = ok ? (unsigned)(gp * p.in_pitch + cb * 32 + ep_q * 4) * 4u : kOOB;
            opix[b][i] = ok ? (unsigned)(gp * p.out_pitch + cb * 32 + ep_q * 4) * 4u : kOOB;
        }
    }
    static_assert(TW == 16 && (PB1 + 1) / 2 <= 4 && (PB2 + 1) / 2 <= 3, "pixel index -> (row, column) by shifts; blocks per wave");
    auto y2_frags = [&](int n, AFrag (&f)[3]) {          // chunk n (16 channels of y2)
#pragma unroll
        for (int b = 0; b < 3; ++b)
            if (b < nb2) {
                f[b].h = *reinterpret_cast<const u32x4 *>(lds + y2a[b] + n * 32);
                f[b].l = *reinterpret_cast<const u32x4 *>(lds + y2a[b] + n * 32 + Y_PLANE);
            }
    };
    const int s3base = n_steps1 + 18;
  if constexpr (PROJ) {
    // ---- stacked-K slices: steps 0, 1 contract y2 (LDS), steps 2 .. sps - 1 contract x (the tile's own pixels, from L2).  The two
    // operands carry different scales (y2: this tile's, x: its tensor's): the accumulators change scale between them (a power of two:
    // exact).  Weights rotate through wf[0 .. 2] by register moves as in conv1 (steps s, s + 1, s + 2).
    unsigned xfo[3];                                     // byte offset in x of this lane's fragment pixel (channel 8 hh), kOOB outside the image
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const int pidx = (pb0 + 2 * b) * 32 + j, pr = pidx >> 4, pc = pidx & 15;
        const bool ok = b < nb2 && pidx < TH * TW && h0 + pr < p.H && w0 + pc < p.W;
        xfo[b] = ok ? (unsigned)((((long)img * p.H + h0 + pr) * p.W + w0 + pc) * p.in_pitch + 8 * hh) * 4u : kOOB;
    }
    const float rescale = __uint_as_float((unsigned)(127 + e_x - e2) << 23);           // from 2^e2 y2 to 2^e_x x
    const float sc3x = __uint_as_float((unsigned)(127 - e_x - p.w_exp3) << 23);
    float4 xraw[2][3][2];                                // the x floats of one step (two chunks), requested one step ahead
    auto x_frag_load = [&](int xs) {                     // x step xs: channels 32 xs + 16 c + 8 hh .. + 7
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                const unsigned o = xfo[b] != kOOB ? xfo[b] + (unsigned)(32 * xs + 16 * c) * 4u : kOOB;
                xraw[c][b][0] = bload4(rs_x, o);
                xraw[c][b][1] = bload4(rs_x, o != kOOB ? o + 16u : kOOB);
            }
    };
    for (int q = 0; q < p.Cout / 64; ++q) {
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
        const unsigned chq = (unsigned)(q * 64) * 4u;
        const int sbase = s3base + q * sps;
        for (int st = 0; st < sps; ++st) {
            w_load(sbase + st + 2, wf[2]);
            if (st < 2) {
                if (st == 1) x_frag_load(0);             // the first x step flies under the second y2 step's MFMAs
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    y2_frags(2 * st + c, af[c]);
#pragma unroll
                    for (int b = 0; b < 3; ++b)
                        if (b < nb2) mfma3(acc[b], wf[0].h[c], wf[0].l[c], af[c][b].h, af[c][b].l);
                }
                if (st == 1) {
#pragma unroll
                    for (int b = 0; b < 3; ++b)
#pragma unroll
                        for (int e = 0; e < 16; ++e) acc[b][e] *= rescale;
                }
            } else {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        unsigned h0_, l0_, h1_, l1_, h2_, l2_, h3_, l3_;
                        split2(xraw[c][b][0].x, xraw[c][b][0].y, a_scale, h0_, l0_);
                        split2(xraw[c][b][0].z, xraw[c][b][0].w, a_scale, h1_, l1_);
                        split2(xraw[c][b][1].x, xraw[c][b][1].y, a_scale, h2_, l2_);
                        split2(xraw[c][b][1].z, xraw[c][b][1].w, a_scale, h3_, l3_);
                        af[c][b].h = u32x4{h0_, h1_, h2_, h3_};
                        af[c][b].l = u32x4{l0_, l1_, l2_, l3_};
                    }
                if (st + 1 < sps) x_frag_load(st - 1);   // the next x step's floats (their registers are free now)
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int b = 0; b < 3; ++b)
                        if (b < nb2) mfma3(acc[b], wf[0].h[c], wf[0].l[c], af[c][b].h, af[c][b].l);
            }
            wf[0] = wf[1];
            wf[1] = wf[2];
        }
        const float4 s4 = *reinterpret_cast<const float4 *>(p.bn + 256 + q * 64 + cb * 32 + ep_q * 4);
        const float4 b4 = *reinterpret_cast<const float4 *>(p.bn + 256 + p.Cout + q * 64 + cb * 32 + ep_q * 4);
        const float4 sv = make_float4(s4.x * sc3x, s4.y * sc3x, s4.z * sc3x, s4.w * sc3x);
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            if (b >= nb2) continue;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                chk = fmaf(acc[b][4 * v], 0.f, fmaf(acc[b][4 * v + 1], 0.f, fmaf(acc[b][4 * v + 2], 0.f, fmaf(acc[b][4 * v + 3], 0.f, chk))));
                *reinterpret_cast<float4 *>(patch + j * P_PITCH + (16 * hh + 4 * v) * 4) =
                    make_float4(acc[b][4 * v], acc[b][4 * v + 1], acc[b][4 * v + 2], acc[b][4 * v + 3]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            float4 t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = *reinterpret_cast<const float4 *>(patch + (ep_px + 8 * i) * P_PITCH + ep_q * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            float m4 = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float o0 = prelu(fmaf(t[i].x, sv.x, b4.x), p.slope);
                const float o1 = prelu(fmaf(t[i].y, sv.y, b4.y), p.slope);
                const float o2 = prelu(fmaf(t[i].z, sv.z, b4.z), p.slope);
                const float o3 = prelu(fmaf(t[i].w, sv.w, b4.w), p.slope);
                const float m = fmaxf(fmaxf(fabsf(o0), fabsf(o1)), fmaxf(fabsf(o2), fabsf(o3)));
                m4 = fmaxf(m4, opix[b][i] != kOOB ? m : 0.f);
                u32x4 o;
                o.x = __float_as_uint(o0); o.y = __float_as_uint(o1); o.z = __float_as_uint(o2); o.w = __float_as_uint(o3);
                __builtin_amdgcn_raw_buffer_store_b128(o, rs_o, opix[b][i] != kOOB ? opix[b][i] + chq : kOOB, 0, 0);
            }
            amax = fmaxf(amax, m4);
        }
    }
  } else {
      // one 64-channel slice per iteration: its two weight steps sit in wf[0], wf[1]; the next slice's first step is requested into wf[2]
      // at the start, its second into wf[0] as soon as that is dead (after two chunks); the slice ends by moving them into place
      for (int q = 0; q < p.Cout / 64; ++q) {
  #pragma unroll
          for (int b = 0; b < 3; ++b)
  #pragma unroll
              for (int e = 0; e < 16; ++e) acc[b][e] = 0.f;
          const unsigned chq = (unsigned)(q * 64) * 4u;                    // byte offset of the slice inside a pixel
          w_load(s3base + 2 * q + 2, wf[2]);                               // (the weights first: in-order returns - they are needed first)
          float4 res[3][4];                                                // the residual of this slice: requested before the MFMAs
  #pragma unroll
          for (int b = 0; b < 3; ++b)
  #pragma unroll
              for (int i = 0; i < 4; ++i) res[b][i] = bload4(rs_x, (xpix[b][i] != kOOB && !(p.dbg & 2)) ? xpix[b][i] + chq : kOOB);
          y2_frags(0, af[0]);
  #pragma unroll
          for (int n = 0; n < 4; ++n) {
              if (n == 2) w_load(s3base + 2 * q + 3, wf[0]);
              if (n + 1 < 4) y2_frags(n + 1, af[(n + 1) & 1]);
  #pragma unroll
              for (int b = 0; b < 3; ++b)
                  if (b < nb2) mfma3(acc[b], wf[n >> 1].h[n & 1], wf[n >> 1].l[n & 1], af[n & 1][b].h, af[n & 1][b].l);
              __builtin_amdgcn_sched_barrier(0);
          }
          const float4 s4 = *reinterpret_cast<const float4 *>(p.bn + 256 + q * 64 + cb * 32 + ep_q * 4);
          const float4 b4 = *reinterpret_cast<const float4 *>(p.bn + 256 + p.Cout + q * 64 + cb * 32 + ep_q * 4);
          const float4 sv = make_float4(s4.x * sc3, s4.y * sc3, s4.z * sc3, s4.w * sc3);
  #pragma unroll
          for (int b = 0; b < 3; ++b) {
              if (b >= nb2) continue;
              // accumulators -> patch: row j, channels 16 hh + 4 v .. + 3 (only this wave touches its patch: in-order LDS + the waits)
  #pragma unroll
              for (int v = 0; v < 4; ++v) {
                  chk = fmaf(acc[b][4 * v], 0.f, fmaf(acc[b][4 * v + 1], 0.f, fmaf(acc[b][4 * v + 2], 0.f, fmaf(acc[b][4 * v + 3], 0.f, chk))));
                  *reinterpret_cast<float4 *>(patch + j * P_PITCH + (16 * hh + 4 * v) * 4) =
                      make_float4(acc[b][4 * v], acc[b][4 * v + 1], acc[b][4 * v + 2], acc[b][4 * v + 3]);
              }
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
              float4 t[4];
  #pragma unroll
              for (int i = 0; i < 4; ++i) t[i] = *reinterpret_cast<const float4 *>(patch + (ep_px + 8 * i) * P_PITCH + ep_q * 16);
              asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // (read before the next block overwrites the patch)
              float m4 = 0.f;
  #pragma unroll
              for (int i = 0; i < 4; ++i) {
                  const float o0 = prelu(fmaf(t[i].x, sv.x, b4.x + res[b][i].x), p.slope);
                  const float o1 = prelu(fmaf(t[i].y, sv.y, b4.y + res[b][i].y), p.slope);
                  const float o2 = prelu(fmaf(t[i].z, sv.z, b4.z + res[b][i].z), p.slope);
                  const float o3 = prelu(fmaf(t[i].w, sv.w, b4.w + res[b][i].w), p.slope);
                  const float m = fmaxf(fmaxf(fabsf(o0), fabsf(o1)), fmaxf(fabsf(o2), fabsf(o3)));
                  m4 = fmaxf(m4, opix[b][i] != kOOB ? m : 0.f);
                  u32x4 o;
                  o.x = __float_as_uint(o0); o.y = __float_as_uint(o1); o.z = __float_as_uint(o2); o.w = __float_as_uint(o3);
                  __builtin_amdgcn_raw_buffer_store_b128(o, rs_o, opix[b][i] != kOOB ? opix[b][i] + chq : kOOB, 0, 0);
              }
              amax = fmaxf(amax, m4);
          }
          wf[1] = wf[0];
          wf[0] = wf[2];
      }
  }
    if (p.range_flag != nullptr && __any(!(chk == 0.f)) && lane == 0) atomicOr(p.range_flag, 1);
    if (p.amax_out != nullptr) tsod_amax_commit(p.amax_out, amax, scr, tid, 256);
}

}  // namespace

extern "C" size_t tsod_bottleneck_wstream_bytes(int32_t Cin, int32_t Cout) {
    if (Cin <= 0 || Cout <= 0 || Cin % 32 || Cout % 64) return 0;
    return (size_t)(Cin / 32 + 18 + (Cout / 64) * 2) * W_STEP;
}

extern "C" size_t tsod_bottleneck_proj_wstream_bytes(int32_t Cin, int32_t Cout) {
    if (Cin <= 0 || Cout <= 0 || Cin % 64 || Cout % 64) return 0;
    return (size_t)(Cin / 32 + 18 + (Cout / 64) * (2 + Cin / 32)) * W_STEP;
}

extern "C" int tsod_bottleneck_fp16x2(const tsod_bottleneck_desc *d, const float *x, const void *wstream, const float *bn,
                                      float *out, tsod_stream_t stream) {
    TSOD_REQUIRE(d && x && wstream && bn && out, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(d->projection == 0 || d->projection == 1, TSOD_ERR_INVALID_ARG);
    const bool proj = d->projection == 1;
    TSOD_REQUIRE(d->Cmid == CMID && d->Cin % 32 == 0 && d->Cin >= 32 && d->Cout % 64 == 0 && (proj || d->Cout == d->Cin), TSOD_ERR_UNSUPPORTED);
    TSOD_REQUIRE(d->in_pitch >= d->Cin && d->out_pitch >= d->Cout && (d->in_pitch & 3) == 0 && (d->out_pitch & 3) == 0, TSOD_ERR_ALIGNMENT);
    TSOD_REQUIRE(tsod_aligned16(x) && tsod_aligned16(out) && tsod_aligned16(wstream) && tsod_aligned16(bn), TSOD_ERR_ALIGNMENT);
    TSOD_REQUIRE((reinterpret_cast<uintptr_t>(d->amax_in) & 63u) == 0 && (reinterpret_cast<uintptr_t>(d->amax_out) & 63u) == 0, TSOD_ERR_ALIGNMENT);
    TSOD_REQUIRE(d->a_scale_exp >= -24 && d->a_scale_exp <= 24, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE((uint64_t)d->N * d->H * d->W * d->in_pitch * 4 < 0xFFFFFFF0ull, TSOD_ERR_UNSUPPORTED);   // 32-bit byte offsets into x
    Params p;
    p.x = x; p.out = out; p.wstream = static_cast<const unsigned char *>(wstream); p.bn = bn;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.in_pitch = d->in_pitch; p.Cout = d->Cout; p.out_pitch = d->out_pitch;
    // tile height: rounds of the launch (two workgroups per CU) x MFMAs the slowest wave issues per tile
    int cus = 0, dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) {
        (void)hipGetLastError();
        cus = 256;
    }
    auto cost = [&](int th, int mfma_slowest) {
        const int64_t tiles = (int64_t)d->N * ((d->W + TW - 1) / TW) * ((d->H + th - 1) / th);
        return (double)((tiles + 2 * cus - 1) / (2 * cus)) * mfma_slowest;
    };
    const int n1 = d->Cin / 32, nq = d->Cout / 64, c3 = proj ? 4 + 2 * n1 : 4;           // chunks of a conv3 slice
    static const int force_th = [] { const char *e = getenv("TSOD_BN_TH"); return e ? atoi(e) : 0; }();
    const int TH = force_th == 8 || force_th == 10 ? force_th
                   : (cost(8, 3 * (3 * 2 * n1 + 2 * 36 + 2 * c3 * nq)) < cost(10, 3 * (4 * 2 * n1 + 3 * 36 + 3 * c3 * nq)) ? 8 : 10);
    p.tiles_x = (d->W + TW - 1) / TW; p.tiles_y = (d->H + TH - 1) / TH;
    TSOD_REQUIRE((uint64_t)d->N * d->H * d->W * d->out_pitch * 4 < 0xFFFFFFF0ull && d->Cin % 64 == 0, TSOD_ERR_UNSUPPORTED);
    p.x_bytes = (unsigned)((uint64_t)d->N * d->H * d->W * d->in_pitch * 4);
    p.out_bytes = (unsigned)((uint64_t)d->N * d->H * d->W * d->out_pitch * 4);
    p.slope = d->slope; p.w_exp1 = d->w_exp[0]; p.w_exp2 = d->w_exp[1]; p.w_exp3 = d->w_exp[2];
    p.a_scale = ldexpf(1.f, d->a_scale_exp);
    p.amax_in = d->amax_in; p.amax_out = d->amax_out; p.range_flag = d->range_flag;
    static const int dbg = [] { const char *e = getenv("TSOD_BN_DBG"); return e ? atoi(e) : 0; }();
    p.dbg = dbg;
    const int64_t grid = (int64_t)d->N * p.tiles_x * p.tiles_y;
    TSOD_REQUIRE(grid < 0x7FFFFFFF, TSOD_ERR_UNSUPPORTED);
    if (proj) {
        if (TH == 8) hipLaunchKernelGGL((bottleneck_kernel<8, true>), dim3((unsigned)grid), dim3(256), 0, tsod_stream(stream), p);
        else hipLaunchKernelGGL((bottleneck_kernel<10, true>), dim3((unsigned)grid), dim3(256), 0, tsod_stream(stream), p);
    } else {
        if (TH == 8) hipLaunchKernelGGL((bottleneck_kernel<8, false>), dim3((unsigned)grid), dim3(256), 0, tsod_stream(stream), p);
        else hipLaunchKernelGGL((bottleneck_kernel<10, false>), dim3((unsigned)grid), dim3(256), 0, tsod_stream(stream), p);
    }
    return tsod_launch_status();
}
