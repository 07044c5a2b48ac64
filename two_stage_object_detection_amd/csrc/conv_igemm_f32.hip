// conv_igemm_f32.hip -- dense convolution / linear as an implicit GEMM on the CDNA4 matrix cores,
// f32 in / f32 accumulate (v_mfma_f32_32x32x2_f32: bit-for-bit a k-ordered fmaf chain, which is
// what lets the detector meet the 1e-3 box/score parity bar; bf16 MFMA does not).
//
// Replaces nn.Conv2d(groups=1) + eval BatchNorm2d + PReLU/ReLU6/ReLU [+ residual]:
//   models/resnet.py:62-74, :21-31, :136-138, :114-116      models/hardnet.py:38-55
//   nets/rpn.py:86-88,107,111                               nets/classify.py:13,15,48,50
//
// GEMM view:  out[m][n] = sum_k A[m][k] * Wp[n][k]
//   m = (image, oh, ow)            -- NHWC activations, so for a fixed filter tap (kh,kw) the
//   k = (kh, kw, ci)                  Cin input channels of a pixel are contiguous in HBM
//   n = output channel             -- Wp is [Cout][KH][KW][Cin]: both operands are "row x K".
//
// Workgroup = 256 threads = 4 waves (2 x 2), tile BM x BN, K-step 32 floats.
//   HBM -> registers : each thread moves 16-byte chunks (4 consecutive k of one row); 8 adjacent
//                      threads cover one row's 128 contiguous bytes (coalesced NHWC reads; the
//                      im2col gather, zero padding and channel-segment concat happen here).
//   registers -> LDS : ds_write_b128 into [row][36] float tiles (row pitch 144 B: the 16-lane
//                      groups of ds_read_b128 then hit 16 distinct 4-bank slots, conflict-free).
//   LDS -> MFMA      : lane l reads ONE ds_read_b128 = 4 consecutive k of row (l & 31) at k-offset
//                      4*(l >> 5); element j feeds MFMA j, so MFMA j contracts k = {j, 4 + j} of the
//                      8-wide sub-step.  A and B use the same permutation, the sum is unchanged.
//   double-buffered LDS, the next K-step's global loads are in flight during the MFMAs,
//   one barrier per K-step; 2 (128-wide tiles) or 4 (64x64) workgroups share a CU.
//   epilogue         : scale/shift (folded BN or bias) + residual + activation on the accumulator,
//                      128-byte row segments per store instruction; or raw partial sums when K is
//                      split across workgroups (deterministic slab reduction in a second kernel).
// Workgroup ids are remapped so that each XCD (private L2) walks a contiguous run of tiles that
// share activation rows.
#include "tsod_internal.h"
#include <limits.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBK = 32;   // floats per K-step
constexpr int kLDK = 36;  // LDS row pitch in floats (32 + 4 pad)

struct ConvParams {
    const float *in, *w, *scale, *shift, *res;
    float *out, *partial;
    int N, H, W, in_pitch;
    int n_seg, seg_off[TSOD_MAX_SEGMENTS], seg_end[TSOD_MAX_SEGMENTS];  // seg_end = cumulative channel count
    int Cin, Cout, out_pitch, out_off;
    int KH, KW, stride, pad_h, pad_w, OH, OW;
    int act;
    float slope;
    int res_pitch, res_off;
    int M, K, ksteps, split_k, ksteps_per_split, tiles_m, tiles_n;
};

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    switch (act) {
        case TSOD_ACT_PRELU: return v > 0.f ? v : v * slope;
        case TSOD_ACT_RELU6: return fminf(fmaxf(v, 0.f), 6.f);
        case TSOD_ACT_RELU: return fmaxf(v, 0.f);
        default: return v;
    }
}

// channel index inside the (concatenated) Cin -> offset inside the input pixel
__device__ __forceinline__ int seg_channel(const ConvParams &p, int ci) {
    if (p.n_seg == 1) return p.seg_off[0] + ci;
    int start = 0;
#pragma unroll
    for (int s = 0; s < TSOD_MAX_SEGMENTS; ++s) {
        if (s < p.n_seg) {
            if (ci < p.seg_end[s]) return p.seg_off[s] + (ci - start);
            start = p.seg_end[s];
        }
    }
    return p.seg_off[0];
}

template <int BM, int BN, int MIN_WAVES>
__global__ void __launch_bounds__(256, MIN_WAVES) conv_igemm_kernel(const ConvParams p) {
    constexpr int TM = BM / 64, TN = BN / 64;          // 32x32 MFMA tiles per wave in m / n
    constexpr int A_ROWS = BM / 32, B_ROWS = BN / 32;  // rows each thread stages per K-step
    constexpr int STAGE = (BM + BN) * kLDK;
    __shared__ __align__(16) float smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // ---- XCD-aware workgroup id: blocks b, b+8, b+16.. share an XCD; give each XCD a contiguous run
    const int nwg = gridDim.x;
    const int q = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int wgid = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (blockIdx.x >> 3);
    const int tn_i = wgid % p.tiles_n;
    const int t2 = wgid / p.tiles_n;
    const int tm_i = t2 % p.tiles_m;
    const int z = t2 / p.tiles_m;
    const int m0 = tm_i * BM, n0 = tn_i * BN;
    const int kt_begin = z * p.ksteps_per_split;
    const int kt_end = min(p.ksteps, kt_begin + p.ksteps_per_split);

    // ---- per-thread staging geometry
    const int c4 = (tid & 7) * 4;  // k offset of this thread's chunk inside the K-step
    const int r0 = tid >> 3;       // 0..31
    long a_base[A_ROWS];
    int a_ih0[A_ROWS], a_iw0[A_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
        const int m = m0 + r0 + 32 * i;
        if (m < p.M) {
            const int ow = m % p.OW;
            const int t = m / p.OW;
            const int oh = t % p.OH;
            const int img = t / p.OH;
            a_ih0[i] = oh * p.stride - p.pad_h;
            a_iw0[i] = ow * p.stride - p.pad_w;
            a_base[i] = (((long)img * p.H + a_ih0[i]) * p.W + a_iw0[i]) * p.in_pitch;
        } else {
            a_ih0[i] = INT_MIN / 2;
            a_iw0[i] = INT_MIN / 2;
            a_base[i] = 0;
        }
    }
    long b_base[B_ROWS];
    bool b_ok[B_ROWS];
#pragma unroll
    for (int i = 0; i < B_ROWS; ++i) {
        const int n = n0 + r0 + 32 * i;
        b_ok[i] = n < p.Cout;
        b_base[i] = (long)n * p.K;
    }

    // running decomposition of this thread's k into (kh, kw, ci)
    int k = kt_begin * kBK + c4;
    int seg = k / p.Cin;
    int ci = k - seg * p.Cin;
    int kh = seg / p.KW;
    int kw = seg - kh * p.KW;

    float4 ra[A_ROWS], rb[B_ROWS];
    auto load_global = [&]() {
        const bool kin = k < p.K;
        const long delta = ((long)kh * p.W + kw) * p.in_pitch + seg_channel(p, ci);
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            const int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;
            const bool ok = kin && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            ra[i] = ok ? *reinterpret_cast<const float4 *>(p.in + a_base[i] + delta) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) {
            rb[i] = (kin && b_ok[i]) ? *reinterpret_cast<const float4 *>(p.w + b_base[i] + k)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // advance to the next K-step
        k += kBK;
        ci += kBK;
        while (ci >= p.Cin) {
            ci -= p.Cin;
            if (++kw == p.KW) { kw = 0; ++kh; }
        }
    };
    auto store_lds = [&](int buf) {
        float *As = smem + buf * STAGE;
        float *Bs = As + BM * kLDK;
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) *reinterpret_cast<float4 *>(As + (r0 + 32 * i) * kLDK + c4) = ra[i];
#pragma unroll
        for (int i = 0; i < B_ROWS; ++i) *reinterpret_cast<float4 *>(Bs + (r0 + 32 * i) * kLDK + c4) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int frag_row = lane & 31;
    const int frag_k = 4 * (lane >> 5);

    if (kt_begin < kt_end) {
        load_global();
        store_lds(0);
        __syncthreads();
        for (int kt = kt_begin; kt < kt_end; ++kt) {
            const int buf = (kt - kt_begin) & 1;
            const bool more = kt + 1 < kt_end;
            if (more) load_global();  // in flight while the MFMAs below run
            const float *As = smem + buf * STAGE + (wm * (BM / 2) + frag_row) * kLDK + frag_k;
            const float *Bs = smem + buf * STAGE + BM * kLDK + (wn * (BN / 2) + frag_row) * kLDK + frag_k;
#pragma unroll
            for (int ks = 0; ks < kBK / 8; ++ks) {
                float4 fa[TM], fb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const float4 *>(As + i * 32 * kLDK + ks * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const float4 *>(Bs + j * 32 * kLDK + ks * 8);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
                    }
            }
            if (more) store_lds(buf ^ 1);
            __syncthreads();
        }
    }

    // ---- epilogue.  acc[i][j][e]: column = lane & 31 (n), row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) (m)
    const int col_in = lane & 31;
    const int row_in = 4 * (lane >> 5);
    if (p.split_k > 1) {
        float *dst = p.partial + (long)z * p.M * p.Cout;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * (BN / 2) + j * 32 + col_in;
            if (n >= p.Cout) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + row_in;
                    if (m < p.M) dst[(long)m * p.Cout + n] = acc[i][j][e];
                }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * (BN / 2) + j * 32 + col_in;
        if (n >= p.Cout) continue;
        const float sc = p.scale ? p.scale[n] : 1.f;
        const float sh = p.shift ? p.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * (BM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + row_in;
                if (m < p.M) {
                    float v = acc[i][j][e] * sc + sh;
                    if (p.res) v += p.res[(long)m * p.res_pitch + p.res_off + n];
                    p.out[(long)m * p.out_pitch + p.out_off + n] = apply_act(v, p.act, p.slope);
                }
            }
    }
}

// Deterministic split-K tail: sum the S partial slabs in slab order, then the same epilogue.
__global__ void __launch_bounds__(256) conv_splitk_reduce_kernel(const ConvParams p) {
    const long total = (long)p.M * p.Cout;
    const long slab = total;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int m = (int)(t / p.Cout);
        const int n = (int)(t - (long)m * p.Cout);
        float v = p.partial[t];
        for (int s = 1; s < p.split_k; ++s) v += p.partial[(long)s * slab + t];
        v = v * (p.scale ? p.scale[n] : 1.f) + (p.shift ? p.shift[n] : 0.f);
        if (p.res) v += p.res[(long)m * p.res_pitch + p.res_off + n];
        p.out[(long)m * p.out_pitch + p.out_off + n] = apply_act(v, p.act, p.slope);
    }
}

// torch [Cout][Cin_src][KH][KW_src] -> [Cout][KH][KW][Cin], zero-filling the added channels / taps
__global__ void __launch_bounds__(256)
pack_weight_kernel(const float *__restrict__ w, int Cout, int Cin_src, int KH, int KW_src, int Cin, int KW,
                   float *__restrict__ out) {
    const long total = (long)Cout * KH * KW * Cin;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int ci = (int)(t % Cin);
        long u = t / Cin;
        const int kw = (int)(u % KW);
        u /= KW;
        const int kh = (int)(u % KH);
        const int co = (int)(u / KH);
        float v = 0.f;
        if (ci < Cin_src && kw < KW_src) v = w[(((long)co * Cin_src + ci) * KH + kh) * KW_src + kw];
        out[t] = v;
    }
}

struct TileInfo { int bm, bn; float cost; };
const TileInfo kTiles[TSOD_TILE_COUNT] = {
    {0, 0, 0.f}, {128, 128, 1.00f}, {128, 64, 1.06f}, {64, 64, 1.15f}, {64, 128, 1.06f}};

int validate(const tsod_conv2d_desc *d) {
    TSOD_REQUIRE(d != nullptr, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(d->N > 0 && d->H > 0 && d->W > 0 && d->Cout > 0 && d->KH > 0 && d->KW > 0, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(d->stride > 0 && d->pad_h >= 0 && d->pad_w >= 0 && d->OH > 0 && d->OW > 0, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(d->n_seg >= 1 && d->n_seg <= TSOD_MAX_SEGMENTS, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE((d->in_pitch & 3) == 0 && d->in_pitch > 0, TSOD_ERR_ALIGNMENT);
    for (int s = 0; s < d->n_seg; ++s) {
        TSOD_REQUIRE(d->seg_len[s] > 0 && d->seg_off[s] >= 0, TSOD_ERR_INVALID_ARG);
        TSOD_REQUIRE((d->seg_len[s] & 3) == 0 && (d->seg_off[s] & 3) == 0, TSOD_ERR_ALIGNMENT);
        TSOD_REQUIRE(d->seg_off[s] + d->seg_len[s] <= d->in_pitch, TSOD_ERR_INVALID_ARG);
    }
    TSOD_REQUIRE(d->out_off >= 0 && d->out_pitch >= d->out_off + d->Cout, TSOD_ERR_INVALID_ARG);
    // the last filter tap of the last output pixel must stay inside the padded input
    TSOD_REQUIRE((d->OH - 1) * d->stride - d->pad_h < d->H && (d->OW - 1) * d->stride - d->pad_w < d->W,
                 TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(d->act >= TSOD_ACT_NONE && d->act <= TSOD_ACT_RELU, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(d->tile >= 0 && d->tile < TSOD_TILE_COUNT && d->split_k >= 0 && d->split_k <= 64, TSOD_ERR_INVALID_ARG);
    const int64_t M = (int64_t)d->N * d->OH * d->OW;
    TSOD_REQUIRE(M < (int64_t)INT_MAX, TSOD_ERR_UNSUPPORTED);
    return TSOD_OK;
}

int desc_cin(const tsod_conv2d_desc *d) {
    int c = 0;
    for (int s = 0; s < d->n_seg; ++s) c += d->seg_len[s];
    return c;
}

int g_cu_count = 0;
int cu_count() {
    if (g_cu_count == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess &&
            hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
            g_cu_count = n;
        else {
            (void)hipGetLastError();
            g_cu_count = 256;  // MI355X
        }
    }
    return g_cu_count;
}

// Cost model, in CU-cycles of the most loaded CU: workgroups are dealt round-robin over the CUs and
// co-resident workgroups share the CU's matrix pipes, so a CU's time is (its workgroups) x (K-steps x
// BM*BN/4 MFMA cycles + a fixed prologue/epilogue), plus the slab round trip when K is split.
void resolve(const tsod_conv2d_desc *d, int *tile_out, int *split_out) {
    const int64_t M = (int64_t)d->N * d->OH * d->OW;
    const int K = d->KH * d->KW * desc_cin(d);
    const int ksteps = (K + kBK - 1) / kBK;
    const int cus = cu_count();
    double best = 1e300;
    int best_tile = TSOD_TILE_128x128, best_split = 1;
    for (int t = 1; t < TSOD_TILE_COUNT; ++t) {
        if (d->tile != TSOD_TILE_AUTO && d->tile != t) continue;
        const int bm = kTiles[t].bm, bn = kTiles[t].bn;
        const int64_t tiles = tsod_cdiv(M, bm) * tsod_cdiv(d->Cout, bn);
        for (int s = 1; s <= 32; s = (d->split_k != 0 ? 64 : s * 2)) {
            if (d->split_k != 0) s = d->split_k < ksteps ? d->split_k : ksteps;  // explicit request: honour it
            else if (s > 1 && ksteps / s < 2) continue;
            const int per = (ksteps + s - 1) / s;
            const int64_t wgs = tiles * s;
            const double per_wg = (double)per * bm * bn / 4.0 * kTiles[t].cost + 2500.0 + (double)bm * bn / 8.0;
            double cost = (double)tsod_cdiv(wgs, cus) * per_wg;
            if (s > 1) cost += 4000.0 + (double)(s + 1) * M * d->Cout * 4.0 / 2000.0;  // ~2 KB / cycle chip-wide
            if (cost < best) { best = cost; best_tile = t; best_split = s; }
        }
    }
    *tile_out = best_tile;
    *split_out = best_split;
}

template <int BM, int BN, int MW>
void launch_tile(const ConvParams &p, hipStream_t s) {
    const int grid = p.tiles_m * p.tiles_n * p.split_k;
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, MW>), dim3(grid), dim3(256), 0, s, p);
}

}  // namespace

extern "C" int tsod_conv2d_resolve(const tsod_conv2d_desc *d, int32_t *tile, int32_t *split_k) {
    const int rc = validate(d);
    if (rc != TSOD_OK) return rc;
    TSOD_REQUIRE(tile && split_k, TSOD_ERR_INVALID_ARG);
    int t, s;
    resolve(d, &t, &s);
    *tile = t;
    *split_k = s;
    return TSOD_OK;
}

extern "C" size_t tsod_conv2d_workspace_bytes(const tsod_conv2d_desc *d) {
    if (validate(d) != TSOD_OK) return 0;
    int t, s;
    resolve(d, &t, &s);
    if (s <= 1) return 0;
    return (size_t)s * (size_t)d->N * d->OH * d->OW * (size_t)d->Cout * sizeof(float);
}

extern "C" int tsod_conv2d_f32(const tsod_conv2d_desc *d, const float *in, const float *w_packed, const float *scale,
                               const float *shift, const float *residual, float *out, void *workspace,
                               size_t workspace_bytes, tsod_stream_t stream) {
    const int rc = validate(d);
    if (rc != TSOD_OK) return rc;
    TSOD_REQUIRE(in && w_packed && out, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(tsod_aligned16(in) && tsod_aligned16(w_packed), TSOD_ERR_ALIGNMENT);
    if (residual) TSOD_REQUIRE(d->res_off >= 0 && d->res_pitch >= d->res_off + d->Cout, TSOD_ERR_INVALID_ARG);

    ConvParams p;
    p.in = in; p.w = w_packed; p.scale = scale; p.shift = shift; p.res = residual; p.out = out;
    p.partial = static_cast<float *>(workspace);
    p.N = d->N; p.H = d->H; p.W = d->W; p.in_pitch = d->in_pitch;
    p.n_seg = d->n_seg;
    int cum = 0;
    for (int s = 0; s < TSOD_MAX_SEGMENTS; ++s) {
        p.seg_off[s] = s < d->n_seg ? d->seg_off[s] : 0;
        if (s < d->n_seg) cum += d->seg_len[s];
        p.seg_end[s] = cum;
    }
    p.Cin = cum; p.Cout = d->Cout; p.out_pitch = d->out_pitch; p.out_off = d->out_off;
    p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad_h = d->pad_h; p.pad_w = d->pad_w;
    p.OH = d->OH; p.OW = d->OW; p.act = d->act; p.slope = d->slope;
    p.res_pitch = d->res_pitch; p.res_off = d->res_off;
    p.M = d->N * d->OH * d->OW;
    p.K = d->KH * d->KW * p.Cin;
    p.ksteps = (p.K + kBK - 1) / kBK;
    int tile, split;
    resolve(d, &tile, &split);
    if (split > p.ksteps) split = p.ksteps;
    p.split_k = split;
    p.ksteps_per_split = (p.ksteps + split - 1) / split;
    p.split_k = (p.ksteps + p.ksteps_per_split - 1) / p.ksteps_per_split;  // no empty slabs
    p.tiles_m = (int)tsod_cdiv(p.M, kTiles[tile].bm);
    p.tiles_n = (int)tsod_cdiv(p.Cout, kTiles[tile].bn);
    if (p.split_k > 1) {
        const size_t need = (size_t)p.split_k * (size_t)p.M * (size_t)p.Cout * sizeof(float);
        TSOD_REQUIRE(workspace != nullptr && workspace_bytes >= need, TSOD_ERR_WORKSPACE);
    }
    hipStream_t s = tsod_stream(stream);
    switch (tile) {
        case TSOD_TILE_128x128: launch_tile<128, 128, 2>(p, s); break;
        case TSOD_TILE_128x64: launch_tile<128, 64, 2>(p, s); break;
        case TSOD_TILE_64x128: launch_tile<64, 128, 2>(p, s); break;
        default: launch_tile<64, 64, 4>(p, s); break;
    }
    if (p.split_k > 1) {
        const long total = (long)p.M * p.Cout;
        const int blocks = (int)(tsod_cdiv(total, 256) < 2048 ? tsod_cdiv(total, 256) : 2048);
        hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, p);
    }
    return tsod_launch_status();
}

extern "C" size_t tsod_linear_workspace_bytes(int32_t M, int32_t K, int32_t N) {
    if (M <= 0 || K <= 0 || N <= 0) return 0;
    tsod_conv2d_desc d = {};
    d.N = 1; d.H = 1; d.W = M; d.in_pitch = K; d.n_seg = 1; d.seg_off[0] = 0; d.seg_len[0] = K;
    d.Cout = N; d.out_pitch = N; d.KH = 1; d.KW = 1; d.stride = 1; d.OH = 1; d.OW = M;
    return tsod_conv2d_workspace_bytes(&d);
}

extern "C" int tsod_linear_f32(const float *in, int32_t M, int32_t K, int32_t in_pitch, const float *w,
                               const float *bias, int32_t N, float *out, int32_t out_pitch, void *workspace,
                               size_t workspace_bytes, tsod_stream_t stream) {
    TSOD_REQUIRE(M > 0 && K > 0 && N > 0, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE((K & 3) == 0, TSOD_ERR_ALIGNMENT);
    tsod_conv2d_desc d = {};
    d.N = 1; d.H = 1; d.W = M; d.in_pitch = in_pitch; d.n_seg = 1; d.seg_off[0] = 0; d.seg_len[0] = K;
    d.Cout = N; d.out_pitch = out_pitch; d.KH = 1; d.KW = 1; d.stride = 1; d.OH = 1; d.OW = M;
    d.act = TSOD_ACT_NONE;
    return tsod_conv2d_f32(&d, in, w, nullptr, bias, nullptr, out, workspace, workspace_bytes, stream);
}

extern "C" int tsod_pack_conv_weight_f32(const float *w_oihw, int32_t Cout, int32_t Cin_src, int32_t KH, int32_t KW_src,
                                         int32_t Cin, int32_t KW, float *w_packed, tsod_stream_t stream) {
    TSOD_REQUIRE(w_oihw && w_packed, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(Cout > 0 && Cin_src > 0 && KH > 0 && KW_src > 0 && Cin >= Cin_src && KW >= KW_src, TSOD_ERR_INVALID_ARG);
    const long total = (long)Cout * KH * KW * Cin;
    const int blocks = (int)(tsod_cdiv(total, 256) < 4096 ? tsod_cdiv(total, 256) : 4096);
    hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, tsod_stream(stream), w_oihw, Cout, Cin_src, KH,
                       KW_src, Cin, KW, w_packed);
    return tsod_launch_status();
}
