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
//
// PREC = 1 ("bf16x3", desc.precision = TSOD_PREC_BF16X3): the same kernel with the f32 operands cut EXACTLY into three
// bf16 pieces (8 + 8 + 8 significand bits by truncation: hi + mid + lo == x) on their way into LDS, and every product
// accumulated in f32 from its six largest piece products (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi; the three dropped
// ones are <= 2^-24 relative) on v_mfma_f32_32x32x16_bf16 - 16x the f32 MFMA rate, 6 instructions per 16 k instead of
// 8 per 16 k at 1/16 the rate: 0.375x the matrix-pipe time at f32-level accuracy (error ~1.3e-7 of sum|a*b|, the f32 fma
// chain's is 1.4-2.5e-7).  Storage, accumulation, epilogue, scheduling and K-slice combine are unchanged.
#include "tsod_internal.h"
#include <limits.h>
#include <stdlib.h>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

constexpr int kPatchLD = 36;  // row pitch (floats) of the epilogue transpose patches
constexpr int kChanTab = 512; // entries of the register-staged kernel's channel-offset table (1x1 convs over concatenated segments: K <= 2048)

struct ConvParams {
    const float *in, *w, *scale, *shift, *res;
    float *out, *partial;
    int *tickets;                 // one arrival counter per K-sliced tile (zero on entry, left zero on exit)
    unsigned part_bytes;          // extent of the slab area (buffer descriptor)
    int N, H, W, in_pitch;
    int n_seg, seg_off[TSOD_MAX_SEGMENTS], seg_end[TSOD_MAX_SEGMENTS];  // seg_end = cumulative channel count
    int Cin, Cout, out_pitch, out_off;
    int KH, KW, stride, pad_h, pad_w, OH, OW;
    int act;
    float slope;
    int res_pitch, res_off;
    int M, K, ksteps, tiles_m, tiles_n;
    // schedule: workgroups [0, dp_tiles) each own one whole output tile; the remaining tiles are cut into
    // `split` K-slices, one workgroup per (tile, slice), partial sums reduced by conv_reduce_kernel
    int dp_tiles, split, ksteps_per_split;
    int sk_q;                     // > 0: balanced schedule (conv_dma_kernel): K-steps per workgroup of the tile-major K-step sequence
    int *range_flag;              // fp16x2 only, optional: set to 1 by any workgroup that ends its K loop with a non-finite accumulator
    float a_scale, acc_scale;     // fp16x2 arithmetic only: activations are split as a_scale * x (a power of two), the accumulators
                                  // are multiplied by acc_scale = 1 / (a_scale * weight scale) before the epilogue
    int ci_wrap;                  // channels a tap's run of K-steps covers before the next tap: Cin (tap-major), 32 (cmajor)
    int cmajor;                   // conv_dma_kernel, KH*KW > 1: K-steps run in (32-channel block, tap) order instead of (tap, channels):
                                  // a pixel's 128-byte line is used by all taps that touch it within KH*KW steps (L2-resident),
                                  // and a K-slice reads only its channel blocks.  Which step of the SAME weight image a K-step is.
    int nmajor;                   // 1: workgroups of one XCD share an output-channel tile (and K-slice): the weights it streams
                                  //    stay in that XCD's L2 and are fetched from beyond it once, not once per XCD (small-M layers,
                                  //    where the weights outweigh the activations); 0: they share activation rows (the default)
    unsigned out_bytes, res_bytes;
    unsigned in_bytes, w_bytes;   // buffer-descriptor extents (hardware bounds check: out of range reads 0)
    int vec_epilogue;             // 1: channels/pitches/offsets are multiples of 4 -> dwordx4 epilogue
    // optional SECOND source (a strided 1x1 tap of another tensor, e.g. the block input under a projection shortcut):
    // k in [K1, K1 + c2) reads channel in2_off + (k - K1) of pixel (oh * stride2, ow * stride2) of in2 [N][H2][W2][in2_pitch]
    const float *in2;
    unsigned in2_bytes;
    int K1, c2, in2_pitch, in2_off, stride2, H2, W2;
    int chan_tab;                 // LDS-DMA tiles: 1 = a 1x1 filter over SEVERAL channel segments (conv_dma_kernel<..., CHAN = true>)
    int pointwise_tab;            // 1: a 1x1 filter (no padding, no second source) over several channel segments (HarDNet's concatenated
                                  //    inputs) with K <= 4 * kChanTab: the channel offset of every 4-k chunk comes from a table the workgroup
                                  //    builds in LDS once, instead of a select chain over the segments + the k -> (tap, channel) arithmetic
    int uniform_tap;              // 1: Cin % K-step == 0 and one channel segment: a K-step lies inside ONE filter tap and one
                                  //    contiguous channel run, so (kh, kw, channel base) are wave-uniform and live on the scalar unit
    float neg_slope, act_hi;      // activation as min(max(v,0) + neg_slope*min(v,0), act_hi)
    float inv_cin, inv_kw;        // reciprocals for the branch-free k -> (kh, kw, ci) split
    // range words (include/tsod.h): amax_out - the abs-max of what this launch stores is added to them (one atomic max per
    // workgroup); amax_in / amax_in2 (fp16x2 only) - the activation scale comes from them instead of a_scale / acc_scale
    const unsigned *amax_in, *amax_in2;
    unsigned *amax_out;
    int w_scale_exp;              // fp16x2: exponent the weight image was packed with (acc_scale = 2^-(activation exponent + this))
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned kOOB = 0xFFFFFFF0u;  // byte offset beyond any descriptor extent -> the load returns zeros

__device__ __forceinline__ float4 buffer_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

template <int AUX>
__device__ __forceinline__ float4 buffer_load4_aux(__amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, AUX);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// Branch-free activation: y = min(max(v,0) + neg_slope * min(v,0), hi) covers NONE (1, +inf), PRELU (a, +inf),
// RELU (0, +inf) and RELU6 (0, 6) exactly (one of the two terms is always an exact zero).
__device__ __forceinline__ float apply_act(float v, float neg_slope, float hi) {
    return fminf(fmaxf(v, 0.f) + neg_slope * fminf(v, 0.f), hi);
}

// x == hi + mid + lo exactly, each piece a bf16 (kept in the upper half of a 32-bit word): truncation leaves <= 16, then <= 8
// significant bits, so both subtractions are exact.
__device__ __forceinline__ void split3(float x, unsigned &h, unsigned &m, unsigned &l) {
    h = __float_as_uint(x) & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(h);
    m = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(m);
    l = __float_as_uint(r2) & 0xFFFF0000u;
}
__device__ __forceinline__ unsigned pack2(unsigned lo_elem, unsigned hi_elem) { return (lo_elem >> 16) | hi_elem; }

// Two operands at once on the hardware converter (v_cvt_pk_bf16_f32, round to nearest even): h, m, l = packed bf16 pairs
// (x0 in the low half = the lower k).  x - bf16(x) is exact (the pieces are within half a bf16 ulp), so h + m + l differs
// from x by at most the rounding of the LAST piece, <= 2^-26 |x|: below the dropped piece products.  5.5 VALU ops per
// element instead of 9.5 for the truncating form.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3_pair(float x0, float x1, unsigned &h, unsigned &m, unsigned &l) {
    f32x2 v = {x0, x1};
    h = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    f32x2 r1 = {x0 - __uint_as_float(h << 16), x1 - __uint_as_float(h & 0xFFFF0000u)};
    m = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2));
    f32x2 r2 = {r1.x - __uint_as_float(m << 16), r1.y - __uint_as_float(m & 0xFFFF0000u)};
    l = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2));
}

// fp16x2 (TSOD_PREC_FP16X2): two elements as fp16 pieces of sc * x: hi = rne_f16(sc x), lo = rne_f16(sc x - hi) (the product by a
// power of two and the subtraction are exact in f32, so each piece is ONE rounding of an exact value).  Four instructions per pair
// on the mixed-precision FMA (v_fma_mixlo/mixhi_f16: f32 x f32 + f16 -> f16, written into one half of the destination): no separate
// scale multiply, no conversion of hi back to f32, no pack - the form with v_cvt_pk_f16_f32 / v_cvt_f32_f16 / v_sub_f32 took eight,
// and the split is what the fp16x2 K loops are short of issue slots for (both kernel families; `sc` wave-uniform).
__device__ __forceinline__ void split2_pair(float x0, float x1, float sc, unsigned &h, unsigned &l) {
    asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
        "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
        "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(h), "=&v"(l) : "v"(x0), "v"(x1), "s"(sc));
}

// channel index inside the (concatenated) Cin -> offset inside the input pixel (select chain, no branches)
__device__ __forceinline__ int seg_channel(const ConvParams &p, int ci) {
    int ch = p.seg_off[0] + ci;
    for (int s = 1; s < p.n_seg; ++s) ch = ci >= p.seg_end[s - 1] ? p.seg_off[s] + (ci - p.seg_end[s - 1]) : ch;
    return ch;
}

// ---- workgroup -> (tile, K-slice) under the uniform schedules.  Blocks b, b + 8, ... share an XCD (private L2; observed
// round-robin placement - a speed matter only, nothing here depends on it for correctness), so the launch's work items are put
// in an order in which neighbours share operands and every XCD takes one CONTIGUOUS run of that order.
//   default: whole tiles first, output-channel tile fastest - an XCD's run shares activation rows; the K-sliced left-over tiles
//            keep the plain round-robin placement (dispatched last, spread over all XCDs);
//   nmajor : (pure schedules only: every tile whole, or every tile cut into `split` slices) order = (output-channel tile,
//            K-slice, row tile) with the row tile fastest - an XCD's run shares ONE block of the weights.
// tile_id = row tile * tiles_n + output-channel tile (the ticket / slab index of a K-sliced tile is tile_id - dp_tiles).
__device__ __forceinline__ void work_item(const ConvParams &p, int &tile_id, int &z) {
    const int b = (int)blockIdx.x;
    if (p.nmajor) {
        const int nwg = (int)gridDim.x;
        const int q = nwg >> 3, r8 = nwg & 7, xcd = b & 7;
        const int L = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (b >> 3);
        const int tm = L % p.tiles_m, rest = L / p.tiles_m;
        const int split = p.dp_tiles > 0 ? 1 : p.split;
        z = p.dp_tiles > 0 ? -1 : rest % split;
        tile_id = tm * p.tiles_n + rest / split;
        return;
    }
    if (b < p.dp_tiles) {
        const int nwg = p.dp_tiles;
        const int q = nwg >> 3, r8 = nwg & 7, xcd = b & 7;
        tile_id = (xcd < r8 ? xcd * (q + 1) : r8 * (q + 1) + (xcd - r8) * q) + (b >> 3);
        z = -1;
    } else {
        const int r = b - p.dp_tiles;
        tile_id = p.dp_tiles + r / p.split;
        z = r % p.split;
    }
}

// ---- epilogue of one BM x BN tile held as 32x32 accumulator blocks, shared by every conv kernel of this file.
// acc[i][j][e]: column = lane & 31 (n), row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) (m) of block (i, j) of wave (wm, wn).
// z < 0: whole tile (BN / PReLU / ReLU6 / residual, then the store); z >= 0: K-slice z (slab + ticket + last-arriver combine).
// `smem`: at least (THREADS / 64) * 32 * kPatchLD floats of LDS that no wave reads any more.
// Which K-slice of its tile a workgroup holds and where every slice's slab is.  Uniform K-slices (desc.split_k = S / -1): tile
// `ticket` of the K-sliced ones has p.split slices, slice s in slab ticket * split + s.  Balanced K ranges (split_k = -2, the
// LDS-DMA kernel): workgroup w owns K-steps [w q, (w+1) q) of the launch's tile-major K-step sequence, so tile t is held by the
// workgroups first .. first + count - 1, and workgroup w keeps the partial tile it STARTS inside in slab 2 w and the one it
// ENDS inside (having started before that tile) in slab 2 w + 1 - at most those two are partial.
struct SliceMap {
    int z, count, ticket;         // this workgroup's slice (-1: it holds the whole tile), slices of the tile, ticket index
    int first;                    // balanced ranges: first workgroup of the tile (-1: uniform K-slices)
    long q, tile_k0;              // balanced ranges: K-steps per workgroup, first K-step of the tile in the launch's sequence
    __device__ __forceinline__ unsigned slab(int s) const {
        if (first < 0) return (unsigned)(ticket * count + s);
        const long w = first + s;
        return (unsigned)(2 * w + (w * q < tile_k0 ? 1 : 0));
    }
};

#ifdef TSOD_CLOCK_DIAG
// diagnostic build: s_memtime stamps of thread 0 inside a K-slice's epilogue, 8 int64 per workgroup (scripts/dma_timeline.py):
//   [0] entry  [1] slab stores drained  [2] ticket add returned  [3] other slices' slabs + residual in registers  [4] exit  [5] 1 = last arriver
__device__ long long *g_epi_stamps = nullptr;
#endif
// G = slabs of OTHER slices a combining thread keeps in flight at once (registers: G * 32 on top of the running sums)
template <int BM, int BN, int WM, int WN, int THREADS, int G = 1>
__device__ __forceinline__ void conv_epilogue(const ConvParams &p, f32x16 (&acc)[WM / 32][WN / 32], float *smem, int tid, int wm, int wn,
                                              int m0, int n0, const SliceMap sm) {
    constexpr int TM = WM / 32, TN = WN / 32;
    const int lane = tid & 63, wave = tid >> 6;
    const int z = sm.z;
    float amax = 0.f;             // largest |value| this thread stores (range words: tsod_amax_commit at every exit that stored)
    // ---- epilogue.  acc[i][j][e]: column = lane & 31 (n), row = (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5) (m).
    // All global accesses go through buffer descriptors: rows / columns outside the problem get the
    // out-of-range offset, so there is no per-element branch (loads return 0, stores are dropped), and the
    // 16 residual loads of a 32x32 tile are all in flight before the first one is consumed.
    const int col_in = lane & 31;
    const int row_in = 4 * (lane >> 5);
    if (z >= 0) {
        // K-slice: the whole BM x BN partial tile (zero rows / columns included) goes to this slice's own slab, then the
        // slice that arrives LAST at the tile's ticket sums the slabs in slice order (fixed order -> bit-reproducible) and
        // applies the epilogue: no second kernel, no launch boundary.  Hand-off between workgroups that may sit on different
        // XCDs (private L2s): slabs are stored WRITE-THROUGH (sc1, 16 bytes per lane: each accumulator block goes through the
        // wave's LDS patch so that a lane owns 4 consecutive columns), every storing wave drains its stores, one lane adds
        // to the ticket at agent scope, and the reducer reads EVERY slab byte with sc1 loads (never through a stale L1/L2 line).
        //
        // Why this is a valid hand-off on gfx950 (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup
        // visibility"; cdna_hip_programming.md Guideline 16).  What can go stale on this chip is (a) a line in the READING
        // CU's vector L1, which no other CU's store ever refreshes, and (b) a line in an XCD's L2 when the producer sits on
        // another XCD (the L2s are not coherent with each other).  The guide's "valid forms" allow sc1 loads IN PLACE OF the
        // consumer's agent-scope acquire when four conditions hold, and they hold here by construction:
        //   (1) EVERY load of the handed-off bytes is a buffer_load ... sc1 to registers (buffer_load4_aux<AUX_SC1> below is
        //       the only reader of the slab area): sc1 loads bypass the L1 and are served at agent scope, so (a) cannot occur;
        //   (2) the producer stored EVERY one of those bytes sc1 (raw_buffer_store_b128 with AUX_SC1 just below): a
        //       write-through store that leaves the producer XCD's L2 for memory and DROPS the line there (guide, stores
        //       table), so no XCD's L2 keeps a private dirty copy, which removes (b);
        //   (3) every storing wave runs `s_waitcnt vmcnt(0)` after its stores (the asm statement below - inline asm, so the
        //       compiler hazard that drops a wait in front of a provably empty scoreboard cannot apply), and the ONE ticket
        //       add that signals for the whole workgroup comes behind the workgroup barrier that follows those waits;
        //   (4) the shape is the guide's first measured row: one lane of each storing workgroup adds to ONE unsharded
        //       agent-scope counter (a returning atomic, executed at the memory side), the consumer is "the workgroup whose
        //       add came last, told by the value its add returned"; the adding wave loads only after its add has returned
        //       (data dependence through s_flag), the other waves after a workgroup barrier that wave joins; hipMalloc
        //       memory (torch's allocator); 16-byte sc1 stores and loads.  One cell differs: that row was measured with one
        //       workgroup per CU and these tiles run 1-6 per CU.  Residency does not enter the argument above (neither L1
        //       bypass nor write-through depends on it); tests/test_hip_ops.py::test_conv_kslice_reduce_is_deterministic_
        //       under_load holds tiles at 2-6 workgroups per CU to bit-identical results over hundreds of launches with L2
        //       eviction in between.  The guide labels this form "measured, not an architectural guarantee"; the
        //       architectural form would add fence(acquire, "agent") (buffer_inv sc1, ~1.7 us) on the last arriver, which
        //       protects plain loads - there are none.
        // The atomic itself is relaxed: ordering against the slab stores comes from (3), not from the atomic's semantics.
        constexpr int AUX_SC1 = 16;
#ifdef TSOD_CLOCK_DIAG
        long long *const eo = (g_epi_stamps != nullptr && tid == 0 && blockIdx.x < 8192) ? g_epi_stamps + 8 * (long)blockIdx.x : nullptr;
        if (eo) eo[0] = __builtin_amdgcn_s_memtime();
#endif
        const __amdgpu_buffer_rsrc_t rs_part = __builtin_amdgcn_make_buffer_rsrc((void *)p.partial, (short)0, (int)p.part_bytes, 0x00020000);
        const int rem = sm.ticket;
        const unsigned my_slab = sm.slab(z) * (unsigned)(BM * BN * 4);                       // byte offset of this slice's slab
        {
            float *patch = smem + wave * (32 * kPatchLD);
            const int pr = lane >> 3, pc = (lane & 7) * 4;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) patch[((e & 3) + 8 * (e >> 2) + row_in) * kPatchLD + col_in] = acc[i][j][e];
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float4 v = *reinterpret_cast<const float4 *>(patch + (pr + 8 * t) * kPatchLD + pc);
                        const int rl = wm * WM + i * 32 + pr + 8 * t, cl = wn * WN + j * 32 + pc;
                        u32x4 o;
                        o.x = __float_as_uint(v.x); o.y = __float_as_uint(v.y); o.z = __float_as_uint(v.z); o.w = __float_as_uint(v.w);
                        __builtin_amdgcn_raw_buffer_store_b128(o, rs_part, my_slab + (unsigned)(rl * BN + cl) * 4u, 0, AUX_SC1);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // patch reads done before the next block overwrites it
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // EVERY storing wave drains its write-through stores
#ifdef TSOD_CLOCK_DIAG
        if (eo) eo[1] = __builtin_amdgcn_s_memtime();
#endif
        __syncthreads();
        int *s_flag = reinterpret_cast<int *>(smem);                       // the one LDS array (patches are idle past the barrier)
        if (tid == 0) {
            const int old = __hip_atomic_fetch_add(p.tickets + rem, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == sm.count - 1;
            if (last) __hip_atomic_store(p.tickets + rem, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // every slice has arrived
            s_flag[0] = last;
#ifdef TSOD_CLOCK_DIAG
            if (eo) { eo[2] = __builtin_amdgcn_s_memtime(); eo[5] = last; }
#endif
        }
        __syncthreads();
        if (s_flag[0] == 0) return;
        // compiler-only ordering (emits no instruction): the sc1 slab loads below must not be hoisted above the barrier that
        // publishes the ticket result; the hardware side of the acquire is condition (1) of the comment at the top
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc((void *)p.out, (short)0, (int)p.out_bytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc((void *)(p.res ? p.res : p.out), (short)0,
                                                                               (int)(p.res ? p.res_bytes : 0u), 0x00020000);
        // The combine is a latency chain (every slab byte comes from beyond this XCD's L2), so a thread keeps ALL the loads of
        // a slice in flight at once - QB quads (16 bytes each) per pass - and pays one round trip per slice and pass, not one
        // per quad; slices are still added in slice order (bit-reproducible).
        constexpr int QPR = BN / 4, QUADS = BM * BN / 4, QPT = QUADS / THREADS, QB = QPT < 8 ? QPT : 8;
        static_assert(QUADS % THREADS == 0 && QPT % QB == 0, "quads per thread");
        const unsigned slab_bytes = (unsigned)(BM * BN * 4);
#pragma unroll 1
        for (int pass = 0; pass < QPT / QB; ++pass) {
            float4 v[QB], rs4[QB];
            unsigned qoff[QB];
            int mq[QB], nq[QB];
#pragma unroll
            for (int u = 0; u < QB; ++u) {
                const int q = tid + THREADS * (pass * QB + u);
                qoff[u] = (unsigned)q * 16u;
                mq[u] = m0 + q / QPR;
                nq[u] = n0 + (q % QPR) * 4;
            }
            // (THREADS is a multiple of the quads per row, so all of a thread's quads sit in ONE column group: its BN scale / shift
            //  are one pair of loads, issued HERE with the residual and the slabs - inside the store loop below they were a
            //  dependent round trip between the last slab and the first store of the launch's critical path)
            static_assert(THREADS % QPR == 0, "a thread's quads share their output channels");
            float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.vec_epilogue && nq[0] < p.Cout) {
                if (p.scale) sc4 = *reinterpret_cast<const float4 *>(p.scale + nq[0]);
                if (p.shift) sh4 = *reinterpret_cast<const float4 *>(p.shift + nq[0]);
            }
            if (p.vec_epilogue) {                                          // the residual rides along with the first group of slices
#pragma unroll
                for (int u = 0; u < QB; ++u) {
                    const bool ok = mq[u] < p.M && nq[u] < p.Cout && p.res != nullptr;
                    rs4[u] = buffer_load4(rs_r, ok ? ((unsigned)mq[u] * (unsigned)p.res_pitch + (unsigned)(p.res_off + nq[u])) * 4u : kOOB);
                }
            }
            // G slices' loads in flight together (one round trip beyond the XCD per group of G, the first group included, not one
            // per slice); the adds stay in slice order: v = slab 0, then + slab 1, + slab 2, ...  The group-size tests are
            // wave-uniform (sm.count is a property of the tile).
            for (int sl = 0; sl < sm.count; sl += G) {
                float4 t[G][QB];
#pragma unroll
                for (int gi = 0; gi < G; ++gi) {
                    if (sl + gi < sm.count) {
                        const unsigned so = sm.slab(sl + gi) * slab_bytes;
#pragma unroll
                        for (int u = 0; u < QB; ++u) t[gi][u] = buffer_load4_aux<AUX_SC1>(rs_part, so + qoff[u]);
                    }
                }
#pragma unroll
                for (int gi = 0; gi < G; ++gi) {
                    if (sl + gi < sm.count) {
                        if (gi == 0 && sl == 0) {
#pragma unroll
                            for (int u = 0; u < QB; ++u) v[u] = t[0][u];
                        } else {
#pragma unroll
                            for (int u = 0; u < QB; ++u) { v[u].x += t[gi][u].x; v[u].y += t[gi][u].y; v[u].z += t[gi][u].z; v[u].w += t[gi][u].w; }
                        }
                    }
                }
            }
#ifdef TSOD_CLOCK_DIAG
            if (eo) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); eo[3] = __builtin_amdgcn_s_memtime(); }
#endif
#pragma unroll
            for (int u = 0; u < QB; ++u) {
                const int m = mq[u], n = nq[u];
                if (m >= p.M || n >= p.Cout) continue;
                float vv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                if (p.vec_epilogue) {                                      // Cout % 4 == 0: all four channels exist
                    const float4 sc = sc4, sh = sh4;
                    const float o0 = apply_act(vv[0] * sc.x + sh.x + rs4[u].x, p.neg_slope, p.act_hi);
                    const float o1 = apply_act(vv[1] * sc.y + sh.y + rs4[u].y, p.neg_slope, p.act_hi);
                    const float o2 = apply_act(vv[2] * sc.z + sh.z + rs4[u].z, p.neg_slope, p.act_hi);
                    const float o3 = apply_act(vv[3] * sc.w + sh.w + rs4[u].w, p.neg_slope, p.act_hi);
                    amax = fmaxf(fmaxf(amax, fmaxf(fabsf(o0), fabsf(o1))), fmaxf(fabsf(o2), fabsf(o3)));
                    u32x4 o;
                    o.x = __float_as_uint(o0); o.y = __float_as_uint(o1); o.z = __float_as_uint(o2); o.w = __float_as_uint(o3);
                    __builtin_amdgcn_raw_buffer_store_b128(o, rs_o, ((unsigned)m * (unsigned)p.out_pitch + (unsigned)(p.out_off + n)) * 4u, 0, 0);
                } else {
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        if (n + c >= p.Cout) continue;
                        float o = vv[c] * (p.scale ? p.scale[n + c] : 1.f) + (p.shift ? p.shift[n + c] : 0.f);
                        if (p.res) o += p.res[(long)m * p.res_pitch + p.res_off + n + c];
                        o = apply_act(o, p.neg_slope, p.act_hi);
                        amax = fmaxf(amax, fabsf(o));
                        p.out[(long)m * p.out_pitch + p.out_off + n + c] = o;
                    }
                }
            }
        }
#ifdef TSOD_CLOCK_DIAG
        if (eo) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); eo[4] = __builtin_amdgcn_s_memtime(); }
#endif
        if (p.amax_out != nullptr) tsod_amax_commit(p.amax_out, amax, smem, tid, THREADS);   // (the last arriver stored the tile)
        return;
    }
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc((void *)p.out, (short)0, (int)p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc((void *)(p.res ? p.res : p.out), (short)0,
                                                                             (int)(p.res ? p.res_bytes : 0u), 0x00020000);
    const bool has_res = p.res != nullptr;
    if (p.vec_epilogue) {
        // Wide epilogue: each 32x32 accumulator block is transposed through a wave-private LDS patch (the staging
        // buffers are idle now) so that a lane owns 4 consecutive output channels: residual loads and output
        // stores become dwordx4, 8 lanes per 128-byte row segment, 4x fewer VMEM instructions than the
        // column-per-lane form.  Only LDS ops of this wave touch the patch: in-order LDS + lgkmcnt(0) orders them.
        float *patch = smem + wave * (32 * kPatchLD);
        const int pr = lane >> 3, pc = (lane & 7) * 4;
        // ALL residual loads of the wave's tile go out before the first transpose (one memory round trip for the tile, under
        // the LDS work, instead of one per 32x32 block); no residual (wave-uniform): no loads at all
        float4 rs_all[TN][TM][4];
        if (has_res) {
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + wn * WN + j * 32 + pc;
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int m = m0 + wm * WM + i * 32 + pr + 8 * t;
                        rs_all[j][i][t] = buffer_load4(rs_res, (m < p.M && n < p.Cout) ? ((unsigned)m * (unsigned)p.res_pitch + (unsigned)(p.res_off + n)) * 4u : kOOB);
                    }
            }
        } else {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int t = 0; t < 4; ++t) rs_all[j][i][t] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + wn * WN + j * 32 + pc;           // this lane's 4 output channels
            const bool n_ok = n < p.Cout;                        // Cout % 4 == 0 on this path: all 4 or none
            float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.scale && n_ok) sc = *reinterpret_cast<const float4 *>(p.scale + n);
            if (p.shift && n_ok) sh = *reinterpret_cast<const float4 *>(p.shift + n);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int e = 0; e < 16; ++e) patch[((e & 3) + 8 * (e >> 2) + row_in) * kPatchLD + col_in] = acc[i][j][e];
                const int mb = m0 + wm * WM + i * 32 + pr;
                float4 v[4];
                const float4 (&rs)[4] = rs_all[j][i];
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = *reinterpret_cast<const float4 *>(patch + (pr + 8 * t) * kPatchLD + pc);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int m = mb + 8 * t;
                    const bool ok = m < p.M && n_ok;
                    const unsigned off = ok ? ((unsigned)m * (unsigned)p.out_pitch + (unsigned)(p.out_off + n)) * 4u : kOOB;
                    const float o0 = apply_act(v[t].x * sc.x + sh.x + rs[t].x, p.neg_slope, p.act_hi);
                    const float o1 = apply_act(v[t].y * sc.y + sh.y + rs[t].y, p.neg_slope, p.act_hi);
                    const float o2 = apply_act(v[t].z * sc.z + sh.z + rs[t].z, p.neg_slope, p.act_hi);
                    const float o3 = apply_act(v[t].w * sc.w + sh.w + rs[t].w, p.neg_slope, p.act_hi);
                    const float m4 = fmaxf(fmaxf(fabsf(o0), fabsf(o1)), fmaxf(fabsf(o2), fabsf(o3)));
                    amax = fmaxf(amax, ok ? m4 : 0.f);               // (rows / columns outside the problem are not stored)
                    u32x4 o;
                    o.x = __float_as_uint(o0); o.y = __float_as_uint(o1); o.z = __float_as_uint(o2); o.w = __float_as_uint(o3);
                    __builtin_amdgcn_raw_buffer_store_b128(o, rs_out, off, 0, 0);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // patch reads done before the next block overwrites it
            }
        }
        if (p.amax_out != nullptr) tsod_amax_commit(p.amax_out, amax, smem, tid, THREADS);
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn * WN + j * 32 + col_in;
        const bool n_ok = n < p.Cout;
        const float sc = (p.scale && n_ok) ? p.scale[n] : 1.f;
        const float sh = (p.shift && n_ok) ? p.shift[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + wm * WM + i * 32 + row_in;
            float r[16];
            if (has_res) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = mb + (e & 3) + 8 * (e >> 2);
                    const unsigned off = (m < p.M && n_ok) ? ((unsigned)m * (unsigned)p.res_pitch + (unsigned)(p.res_off + n)) * 4u : kOOB;
                    r[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs_res, off, 0, 0));
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) r[e] = 0.f;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = mb + (e & 3) + 8 * (e >> 2);
                const bool ok = m < p.M && n_ok;
                const unsigned off = ok ? ((unsigned)m * (unsigned)p.out_pitch + (unsigned)(p.out_off + n)) * 4u : kOOB;
                const float v = apply_act(acc[i][j][e] * sc + sh + r[e], p.neg_slope, p.act_hi);
                amax = fmaxf(amax, ok ? fabsf(v) : 0.f);
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs_out, off, 0, 0);
            }
        }
    }
    if (p.amax_out != nullptr) tsod_amax_commit(p.amax_out, amax, smem, tid, THREADS);
}

// fp16x2 with range words: the activation exponent from the abs-max the producers left (one word per lane), the accumulators'
// way back from it and the weight image's exponent.  Every wave computes the same (wave-uniform) pair for itself.
__device__ __forceinline__ void fp16x2_scales_from_bits(unsigned bits, int w_scale_exp, float &a_scale, float &acc_scale) {
    const int e = tsod_fp16x2_exp_from_bits(bits);
    a_scale = __uint_as_float((unsigned)(127 + e) << 23);
    acc_scale = __uint_as_float((unsigned)(127 - e - w_scale_exp) << 23);
}

// BM x BN workgroup tile, WM x WN per-wave tile: (BM/WM) x (BN/WN) waves of 64 lanes.
// NBUF = 2: double-buffered LDS (the next K-step is staged under the MFMAs); NBUF = 1: one LDS buffer, the next
// K-step waits in registers and is written between two barriers - half the LDS and fewer registers per workgroup,
// so more workgroups share a CU (more waves per SIMD to cover each other's waits, finer-grained chip filling).
// BK = floats per K-step (32 or 64): LDS rows are BK + 4 floats (pitch = 4 mod 64 banks: every 16-lane group of
// ds_read_b128 hits 16 distinct 4-bank slots); a longer step halves the barriers per FLOP of long-K (3x3) layers.
#ifdef TSOD_CLOCK_DIAG
// Diagnostic build only (make diag -> libtsod_diag.so): thread 0 of every workgroup stamps s_memtime (core clock) and
// s_memrealtime (100 MHz) around its K loop; clock = d(memtime) / d(memrealtime) * 100 MHz (scripts/conv_clock.py).
// The stamps go to a buffer of their own and nothing is computed from them.
__device__ long long *g_clock_buf = nullptr;
// conv_dma_kernel timeline (scripts/dma_timeline.py): 8 int64 per workgroup =
//   [0] s_memrealtime at entry  [1] s_memrealtime at exit  [2] cycles in prologues (ring fill -> stage 0 visible)
//   [3] cycles in K loops       [4] cycles in epilogues (incl. slab store, ticket, combine)  [5] K-steps  [6] segments  [7] XCC id
__device__ long long *g_dma_stamps = nullptr;
#endif

template <int BM, int BN, int WM, int WN, int MIN_WAVES, int NBUF = 2, int BK = 32, int PREC = 0>
__global__ void __launch_bounds__(64 * (BM / WM) * (BN / WN), MIN_WAVES) conv_igemm_kernel(const ConvParams p) {
    // f32: rows of BK + 4 floats.  bf16x3: three planes (hi, mid, lo) per operand, rows of BK bf16 with NO padding and the
    // 16-byte slots of a row XOR-swizzled by the row index (kSwz).  Banking on gfx950 (MI355X_MICROARCH.md, LDS): a
    // ds_read_b128 is served in four fixed 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, + 32) against 64 banks
    // (256 bytes); ds_write_b64 / _b128 in contiguous 16- / 8-lane groups against 32 banks (128 bytes).  With an unpadded
    // pitch the staging writes of a group cover whole rows (conflict-free; the padded 80-byte pitch cost 2-way conflicts
    // on every write, a third of the LDS-active cycles), and slot' = slot ^ ((row / rows per 256 B) & (slots - 1)) gives the
    // 16 rows of every read group 16 distinct slots of the 256-byte line.  Row pitch counted in floats here.
    constexpr int kBK = BK, kLDK = PREC ? BK / 2 : BK + 4;
    constexpr int NPL = PREC == 2 ? 2 : 3;             // pieces per operand of the split arithmetics (bf16x3: 3, fp16x2: 2)
    constexpr int kPlanes = PREC ? NPL : 1;
    constexpr int kSlots = BK / 8;                     // 16-byte slots (8 bf16) per bf16x3 row
    constexpr int kRowsPerLine = 256 / (kSlots * 16) > 0 ? 256 / (kSlots * 16) : 1;   // rows sharing one 256-byte bank line
    auto swz = [](int row) { return (row / kRowsPerLine) & (kSlots - 1); };            // slot' = slot ^ swz(row)
    constexpr int TPR = BK / 4;                        // threads per staged row (16-byte chunks)
    constexpr int WAVES_N = BN / WN;
    constexpr int THREADS = 64 * (BM / WM) * WAVES_N;
    constexpr int RPP = THREADS / TPR;                 // rows staged per pass
    constexpr int TM = WM / 32, TN = WN / 32;          // 32x32 MFMA tiles per wave in m / n
    constexpr int A_ROWS = BM / RPP;                   // A rows each thread stages per K-step (one float4 chunk each)
    // B per thread and K-step.  f32: BN / RPP float4 chunks.  bf16x3: the weights arrive PRE-SPLIT, [Cout][K/8][hi|mid|lo][8]
    // bf16 (48 bytes per 8 k), so a thread moves whole 8-k groups: three 16-byte loads, three ds_write_b128, no VALU.
    constexpr int GPR = BK / 8;                        // 8-k groups per row and K-step
    constexpr int B_TASKS = (BN * GPR + THREADS - 1) / THREADS;
    constexpr int B_ROWS = PREC ? NPL * B_TASKS : BN / RPP;   // staging registers (16 bytes each) for B
    constexpr int STAGE = kPlanes * (BM + BN) * kLDK;
    static_assert(BM % RPP == 0 && BN % RPP == 0, "tile rows must be a multiple of the staging pass");
    constexpr int PATCHES = (THREADS / 64) * 32 * kPatchLD;                      // epilogue transpose patches, one per wave
    constexpr int SMEM = NBUF * STAGE > PATCHES ? NBUF * STAGE : PATCHES;
    __shared__ __align__(16) float smem[SMEM];
    __shared__ int chan_tab[kChanTab];                 // pointwise_tab: byte offset inside the pixel of 4-k chunk j (kOOB past K)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    float a_scale = p.a_scale, acc_scale = p.acc_scale;            // fp16x2: static exponents, or (range words) the tensor's own
    if constexpr (PREC == 2) {
        if (p.amax_in != nullptr) {                                  // (wave-uniform; the load flies under the first K-step's loads)
            unsigned bits = p.amax_in[lane * TSOD_AMAX_STRIDE_WORDS];
            if (p.amax_in2 != nullptr) { const unsigned b2 = p.amax_in2[lane * TSOD_AMAX_STRIDE_WORDS]; bits = b2 > bits ? b2 : bits; }
            fp16x2_scales_from_bits(tsod_amax_reduce_bits(bits), p.w_scale_exp, a_scale, acc_scale);
        }
        // (both are wave-uniform: scalar registers, as the kernel arguments they replace were)
        a_scale = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(a_scale)));
        acc_scale = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(acc_scale)));
    }

    // ---- workgroup -> work.  Whole-tile workgroups come first and are remapped so that each XCD (private
    // L2; blocks b, b+8, ... share one) walks a contiguous run of tiles sharing activation rows.  K-slice
    // workgroups keep the round-robin placement: they are dispatched last and spread over all XCDs.
    int tile_id, z;
    work_item(p, tile_id, z);
    const int tn_i = tile_id % p.tiles_n;
    const int tm_i = tile_id / p.tiles_n;
    const int m0 = tm_i * BM, n0 = tn_i * BN;
    const int kt_begin = z < 0 ? 0 : z * p.ksteps_per_split;
    const int kt_end = z < 0 ? p.ksteps : min(p.ksteps, kt_begin + p.ksteps_per_split);

    // ---- per-thread staging geometry (all offsets are 32-bit BYTE offsets into buffer descriptors)
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void *)p.in, (short)0, (int)p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void *)p.w, (short)0, (int)p.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_in2 = __builtin_amdgcn_make_buffer_rsrc((void *)(p.in2 ? p.in2 : p.in), (short)0,
                                                                             (int)(p.in2 ? p.in2_bytes : 0u), 0x00020000);
    const int c4 = (tid % TPR) * 4;  // k offset of this thread's chunk inside the K-step
    const int r0 = tid / TPR;        // 0..RPP-1
    unsigned a_base[A_ROWS], a2_base[A_ROWS];
    int a_ih0[A_ROWS], a_iw0[A_ROWS];
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
        const int m = m0 + r0 + RPP * i;
        if (m < p.M) {
            const int ow = m % p.OW;
            const int t = m / p.OW;
            const int oh = t % p.OH;
            const int img = t / p.OH;
            a_ih0[i] = oh * p.stride - p.pad_h;
            a_iw0[i] = ow * p.stride - p.pad_w;
            // may wrap below zero for border rows; adding a valid tap's delta brings it back in range
            a_base[i] = (unsigned)((((long)img * p.H + a_ih0[i]) * p.W + a_iw0[i]) * p.in_pitch) * 4u;
            a2_base[i] = p.c2 > 0 ? (unsigned)((((long)img * p.H2 + oh * p.stride2) * p.W2 + ow * p.stride2) * p.in2_pitch + p.in2_off) * 4u : kOOB;
        } else {
            a_ih0[i] = INT_MIN / 2;
            a_iw0[i] = INT_MIN / 2;
            a_base[i] = 0;
            a2_base[i] = kOOB;
        }
    }
    constexpr int B_BASES = PREC ? B_TASKS : B_ROWS;
    unsigned b_base[B_BASES];
    const int kgroups = (p.K + 7) / 8;                 // bf16x3: 8-k groups per weight row (the pack pads K to a multiple of 8)
#pragma unroll
    for (int i = 0; i < B_BASES; ++i) {
        if constexpr (PREC == 0) {
            const int n = n0 + r0 + RPP * i;
            b_base[i] = n < p.Cout ? (unsigned)n * (unsigned)p.K * 4u : kOOB;
        } else {
            const int t = tid + THREADS * i, n = n0 + t / GPR;
            b_base[i] = (t < BN * GPR && n < p.Cout) ? (unsigned)n * (unsigned)kgroups * (16u * NPL) + (unsigned)(t % GPR) * (16u * NPL) : kOOB;
        }
    }

    if (p.pointwise_tab) {           // (wave-uniform; the barrier in front of the K loop's first LDS reads publishes the table)
        for (int jj = tid; jj < kChanTab; jj += THREADS) chan_tab[jj] = 4 * jj < p.K ? seg_channel(p, 4 * jj) * 4 : (int)kOOB;
        __syncthreads();
    }
    int kq = kt_begin;               // K-step being loaded
    int k = kt_begin * kBK + c4;     // this thread's k for the A chunk of that K-step

    // Two register staging sets: the loads of K-step kt+2 are issued while step kt computes and are only
    // written to LDS at the end of step kt+1, so a workgroup tolerates ~2 K-steps of memory latency (a lone
    // workgroup per CU is otherwise bound by one L2/HBM round trip per K-step).
    float4 ra0[A_ROWS], rb0[B_ROWS], ra1[A_ROWS], rb1[B_ROWS];
    // uniform-tap fast path: (kh, kw, channel base) of the K-step being loaded, advanced with scalar adds / compares
    int u_kh = 0, u_kw = 0, u_ci = 0;
    if (p.uniform_tap) {
        const int kb = kt_begin * kBK;
        const int seg0 = kb / p.Cin;                  // once per workgroup
        u_ci = kb - seg0 * p.Cin;
        u_kh = seg0 / p.KW;
        u_kw = seg0 - u_kh * p.KW;
    }
    auto load_global = [&](float4(&ra)[A_ROWS], float4(&rb)[B_ROWS]) {
#ifdef TSOD_DIAG_NOLOAD
        // timing diagnostic only (wrong results): after the first two K-steps no global load is issued, the staged registers
        // are re-used - what the K loop costs when memory latency is taken out
        if (kq > kt_begin + 1) { ++kq; k += kBK; return; }
#endif
        if (p.uniform_tap) {
            // the per-thread part of the address is the constant c4; everything else about this K-step is wave-uniform
            if (kq * kBK >= p.K1) {                   // K-steps past the first source's K: the second source (a 1x1 tap, always in range)
                const unsigned d2 = (unsigned)((kq * kBK - p.K1 + c4) * 4);
#pragma unroll
                for (int i = 0; i < A_ROWS; ++i) ra[i] = buffer_load4(rs_in2, a2_base[i] != kOOB ? a2_base[i] + d2 : kOOB);
            } else {
                const unsigned delta = (unsigned)(((u_kh * p.W + u_kw) * p.in_pitch + p.seg_off[0] + u_ci + c4) * 4);
#pragma unroll
                for (int i = 0; i < A_ROWS; ++i) {
                    const int ih = a_ih0[i] + u_kh, iw = a_iw0[i] + u_kw;
                    const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                    ra[i] = buffer_load4(rs_in, ok ? a_base[i] + delta : kOOB);
                }
            }
            if constexpr (PREC == 0) {
#pragma unroll
                for (int i = 0; i < B_ROWS; ++i) rb[i] = buffer_load4(rs_w, b_base[i] != kOOB ? b_base[i] + (unsigned)k * 4u : kOOB);
            } else {
                const unsigned goff = (unsigned)kq * (unsigned)(GPR * 16 * NPL);
#pragma unroll
                for (int i = 0; i < B_TASKS; ++i) {
#pragma unroll
                    for (int q = 0; q < NPL; ++q) rb[NPL * i + q] = buffer_load4(rs_w, b_base[i] != kOOB ? b_base[i] + goff + 16u * q : kOOB);
                }
            }
            u_ci += kBK;
            if (u_ci >= p.Cin) { u_ci = 0; if (++u_kw == p.KW) { u_kw = 0; ++u_kh; } }
            k += kBK;
            ++kq;
            return;
        }
        // k -> (filter tap, channel) without divisions or loops: exact for k < 2^21
        int kh = 0, kw = 0;
        bool kin;
        unsigned delta;
        if (p.pointwise_tab) {       // 1x1 over concatenated segments: tap (0, 0); the chunk's channel offset is one LDS word (kOOB past K)
            delta = (unsigned)chan_tab[k >> 2];
            kin = delta != kOOB;
        } else {
            const int seg = (int)(((float)k + 0.5f) * p.inv_cin);
            const int ci = k - seg * p.Cin;
            kh = (int)(((float)seg + 0.5f) * p.inv_kw);
            kw = seg - kh * p.KW;
            kin = k < p.K;
            delta = (unsigned)(((kh * p.W + kw) * p.in_pitch + seg_channel(p, ci)) * 4);
        }
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) {
            const int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;
            const bool ok = kin && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
            ra[i] = buffer_load4(rs_in, ok ? a_base[i] + delta : kOOB);
        }
        if constexpr (PREC == 0) {
#pragma unroll
            for (int i = 0; i < B_ROWS; ++i) rb[i] = buffer_load4(rs_w, (kin && b_base[i] != kOOB) ? b_base[i] + (unsigned)k * 4u : kOOB);
        } else {
            const unsigned goff = (unsigned)kq * (unsigned)(GPR * 16 * NPL);
#pragma unroll
            for (int i = 0; i < B_TASKS; ++i) {
                const bool ok = b_base[i] != kOOB && kq * GPR + (tid + THREADS * i) % GPR < kgroups;
#pragma unroll
                for (int q = 0; q < NPL; ++q) rb[NPL * i + q] = buffer_load4(rs_w, ok ? b_base[i] + goff + 16u * q : kOOB);
            }
        }
        k += kBK;
        ++kq;
    };
    // one staged A chunk (4 consecutive k of one row) -> LDS; f32 B chunks go the same way (operand 1)
    auto store_chunk = [&](int buf, int operand, int row, const float4 &v) {
        float *base = smem + buf * STAGE + (operand ? kPlanes * BM * kLDK : 0);
        if constexpr (PREC == 0) {
            *reinterpret_cast<float4 *>(base + row * kLDK + c4) = v;
        } else {
            unsigned h[2], m[2], l[2];
            if constexpr (PREC == 2) {
                split2_pair(v.x, v.y, a_scale, h[0], l[0]);
                split2_pair(v.z, v.w, a_scale, h[1], l[1]);
                m[0] = m[1] = 0;
            } else {
                split3_pair(v.x, v.y, h[0], m[0], l[0]);
                split3_pair(v.z, v.w, h[1], m[1], l[1]);
            }
            unsigned char *b = reinterpret_cast<unsigned char *>(base) + row * (kLDK * 4) + (((c4 >> 3) ^ swz(row)) << 4) + (c4 & 4) * 2;
            *reinterpret_cast<uint2 *>(b) = make_uint2(h[0], h[1]);
            if constexpr (PREC == 1) *reinterpret_cast<uint2 *>(b + BM * (kLDK * 4)) = make_uint2(m[0], m[1]);
            *reinterpret_cast<uint2 *>(b + (NPL - 1) * BM * (kLDK * 4)) = make_uint2(l[0], l[1]);
        }
    };
    auto store_b = [&](int buf, const float4(&rb)[B_ROWS]) {
        if constexpr (PREC == 0) {
#pragma unroll
            for (int i = 0; i < B_ROWS; ++i) store_chunk(buf, 1, r0 + RPP * i, rb[i]);
        } else {
            unsigned char *base = reinterpret_cast<unsigned char *>(smem + buf * STAGE + kPlanes * BM * kLDK);
#pragma unroll
            for (int i = 0; i < B_TASKS; ++i) {
                const int t = tid + THREADS * i;
                if (t < BN * GPR) {
#pragma unroll
                    for (int q = 0; q < NPL; ++q)
                        *reinterpret_cast<float4 *>(base + q * BN * (kLDK * 4) + (t / GPR) * (kLDK * 4) +
                                                    (((t % GPR) ^ swz(t / GPR)) << 4)) = rb[NPL * i + q];
                }
            }
        }
    };
    auto store_lds = [&](int buf, const float4(&ra)[A_ROWS], const float4(&rb)[B_ROWS]) {
#ifdef TSOD_DIAG_NOSTORE
        if (kq > kt_begin + 2) return;        // timing diagnostic only (wrong results): no split / LDS writes after the first steps
#endif
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) store_chunk(buf, 0, r0 + RPP * i, ra[i]);
        store_b(buf, rb);
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
    const int a_frag_off = (wm * WM + frag_row) * kLDK + frag_k;
    const int b_frag_off = BM * kLDK + (wn * WN + frag_row) * kLDK + frag_k;

    // One K-step.  Instruction order is chosen so that every latency sits under matrix-core time:
    //   LDS fragment reads of sub-step ks+1 are issued before the MFMAs of sub-step ks (register double buffer),
    //   the global loads of step kt+2 (address VALU + buffer_load) are issued while the first fragments fly,
    //   the LDS writes of step kt+1 are issued behind the first MFMA group and complete under the rest.
    auto kstep = [&](int buf, bool do_load, float4(&lra)[A_ROWS], float4(&lrb)[B_ROWS], bool do_store,
                     const float4(&sra)[A_ROWS], const float4(&srb)[B_ROWS]) {
        if constexpr (PREC != 0) {
            // bf16x3 / fp16x2: per 16-k chunk a lane reads ONE ds_read_b128 per plane and 32-row block = 8 consecutive bf16 of row
            // (lane & 31) at k = 16 * chunk + 8 * (lane >> 5); six MFMAs per (A block, B block), smallest products first
            constexpr int ROW_B = kLDK * 4, CHUNKS = kBK / 16;
            // logical slot of chunk c for this lane = 2c + (lane >> 5); physical = that ^ swz(row), and swz(row) depends only on
            // (lane & 31) because every 32-row block starts at a multiple of the swizzle period: one XOR per chunk
            const int x0 = (lane >> 5) ^ swz(frag_row);
            const unsigned char *As = reinterpret_cast<const unsigned char *>(smem + buf * STAGE) + (wm * WM + frag_row) * ROW_B;
            const unsigned char *Bs = reinterpret_cast<const unsigned char *>(smem + buf * STAGE + kPlanes * BM * kLDK) +
                                      (wn * WN + frag_row) * ROW_B;
            // two fragment sets: chunk c + 1 is read into the other set BEFORE chunk c's MFMAs are issued, so its LDS latency
            // lies under them (with one set the compiler re-reads into registers an MFMA has just consumed and the next MFMA
            // waits for LDS three times per chunk)
            bf16x8 fa[2][TM][NPL], fb[2][TN][NPL];
            auto load_frags = [&](int c, int set) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int q = 0; q < NPL; ++q)
                        fa[set][i][q] = *reinterpret_cast<const bf16x8 *>(As + q * BM * ROW_B + i * 32 * ROW_B + (((2 * c) ^ x0) << 4));
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < NPL; ++q)
                        fb[set][j][q] = *reinterpret_cast<const bf16x8 *>(Bs + q * BN * ROW_B + j * 32 * ROW_B + (((2 * c) ^ x0) << 4));
            };
            load_frags(0, 0);
            if (do_load) load_global(lra, lrb);
#pragma unroll
            for (int c = 0; c < CHUNKS; ++c) {
                const int cur = c & 1;
                if (c + 1 < CHUNKS) load_frags(c + 1, cur ^ 1);
                // six piece products per (A block, B block), smallest first; the product loop is OUTSIDE the block loops so that
                // consecutive MFMAs write different accumulators wherever the wave owns more than one 32x32 block
                // (fp16x2: lo*hi, hi*lo, hi*hi on the fp16 instruction)
                constexpr int NQ = PREC == 2 ? 3 : 6;
                constexpr int PA[6] = {PREC == 2 ? 1 : 2, 0, PREC == 2 ? 0 : 1, 1, 0, 0};          // lo*hi, hi*lo, mid*mid, mid*hi, hi*mid, hi*hi
                constexpr int PB[6] = {0, PREC == 2 ? 1 : 2, PREC == 2 ? 0 : 1, 0, 1, 0};
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) {
                            if constexpr (PREC == 2)
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[cur][i][PA[q]]),
                                                                                   __builtin_bit_cast(f16x8, fb[cur][j][PB[q]]), acc[i][j], 0, 0, 0);
                            else
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][i][PA[q]], fb[cur][j][PB[q]], acc[i][j], 0, 0, 0);
                        }
                if (do_store) {                       // the next step's split + LDS writes ride behind this chunk's MFMAs
                    if (c == 0) {
#pragma unroll
                        for (int q = 0; q < A_ROWS; ++q) store_chunk(buf ^ 1, 0, r0 + RPP * q, sra[q]);
                    }
                    if (c == CHUNKS - 1) store_b(buf ^ 1, srb);
                }
            }
            __syncthreads();
            return;
        }
        const float *As = smem + buf * STAGE + a_frag_off;
        const float *Bs = smem + buf * STAGE + b_frag_off;
        float4 fa[2][TM], fb[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const float4 *>(As + i * 32 * kLDK);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const float4 *>(Bs + j * 32 * kLDK);
        if (do_load) load_global(lra, lrb);
#pragma unroll
        for (int ks = 0; ks < kBK / 8; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < kBK / 8) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[nxt][i] = *reinterpret_cast<const float4 *>(As + i * 32 * kLDK + (ks + 1) * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[nxt][j] = *reinterpret_cast<const float4 *>(Bs + j * 32 * kLDK + (ks + 1) * 8);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].x, fb[cur][j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].y, fb[cur][j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].z, fb[cur][j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i].w, fb[cur][j].w, acc[i][j], 0, 0, 0);
                }
            if (do_store) {
                // the two operands' LDS writes of the next step go out behind different MFMA groups: one 16 KB burst right
                // after the barrier stalls the fragment reads of the other workgroups on the CU (+2-4 % on the two-stage tiles)
                constexpr int SUB = kBK / 8;
                if (ks == 0) {
#pragma unroll
                    for (int q = 0; q < A_ROWS; ++q) store_chunk(buf ^ 1, 0, r0 + RPP * q, sra[q]);
                }
                if (ks == (SUB > 2 ? 2 : SUB - 1)) store_b(buf ^ 1, srb);
            }
        }
        __syncthreads();
    };

    const int nk = kt_end - kt_begin;
#ifdef TSOD_CLOCK_DIAG
    long long diag_c0 = 0, diag_r0 = 0;
    if (g_clock_buf != nullptr && tid == 0) { diag_c0 = __builtin_amdgcn_s_memtime(); diag_r0 = __builtin_amdgcn_s_memrealtime(); }
#endif
    if (NBUF == 1) {
        if (nk > 0) {
            load_global(ra0, rb0);
            store_lds(0, ra0, rb0);
            __syncthreads();
            for (int it = 0; it < nk; ++it) {
                const bool more = it + 1 < nk;
                kstep(0, more, ra0, rb0, false, ra1, rb1);     // loads of step it+1 fly under the MFMAs of step it
                if (more) {
                    store_lds(0, ra0, rb0);                     // every wave is past the barrier that ends kstep
                    __syncthreads();
                }
            }
        }
    } else if (nk > 0) {
        load_global(ra0, rb0);                       // step 0
        if (nk > 1) load_global(ra1, rb1);           // step 1
        store_lds(0, ra0, rb0);
        __syncthreads();
        for (int it = 0; it < nk; it += 2) {
            // even step: LDS buffer 0; set 0 is free -> prefetch step it+2 into it; set 1 (step it+1) -> LDS buffer 1
            kstep(0, it + 2 < nk, ra0, rb0, it + 1 < nk, ra1, rb1);
            if (it + 1 >= nk) break;
            // odd step: LDS buffer 1; prefetch step it+3 into set 1; set 0 (step it+2) -> LDS buffer 0
            kstep(1, it + 3 < nk, ra1, rb1, it + 2 < nk, ra0, rb0);
        }
    }

#ifdef TSOD_CLOCK_DIAG
    if (g_clock_buf != nullptr && tid == 0 && blockIdx.x < 32768) {
        g_clock_buf[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - diag_c0;
        g_clock_buf[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - diag_r0;
    }
#endif
    const SliceMap sm = {z, p.split, tile_id - p.dp_tiles, -1, 0, 0};
    if constexpr (PREC == 2) {                                   // fp16x2: back to x . w, and the range guard (conv_dma_kernel has the same)
        bool bad = false;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    acc[i][j][e] *= acc_scale;
                    bad |= !(fabsf(acc[i][j][e]) <= 3.4028234664e38f);
                }
        if (p.range_flag != nullptr && __any(bad) && lane == 0) atomicOr(p.range_flag, 1);
    }
    // (two slices' slabs in flight per round trip of the combine where a thread holds few quads: 64x64 tiles; one elsewhere - registers)
    conv_epilogue<BM, BN, WM, WN, THREADS, (BM * BN / 4 / THREADS <= 4) ? 2 : 1>(p, acc, smem, tid, wm, wn, m0, n0, sm);
}

// =====================================================================================================================
// bf16x3 convolution fed by LDS-DMA (gfx950 `buffer_load_dwordx4 ... offen lds`): no VGPR staging, no ds_write, every
// instruction of the K loop a pinned asm statement.  What the register-staged kernel above loses at low occupancy is issue
// order (an LDS-read burst, a split burst and a DMA burst in front of the MFMAs of each K-step); here the order is the
// source order:  barrier -> {MFMAs of stage p | LDS reads of stage p+1 | split of stage p+1's A fragment | DMA of stage p+S}.
//   * A (activations, f32) lands in LDS RAW: rows of BK floats, the 16-byte slots of a row XOR-swizzled through the per-lane
//     SOURCE address (an LDS-DMA writes lane i at base + 16 i, so the swizzle can only be applied on the source side).  The
//     im2col gather is that same per-lane source address; padding taps and rows >= M use the out-of-range offset, for which
//     the DMA writes zeros (scripts/micro/lds_dma_oob.hip).  The wave that OWNS a fragment splits it into the three bf16
//     pieces, in the shadow of its own MFMAs: no element is split twice, and there are 44 VALU per 24 MFMAs.
//   * B (weights) is the pre-split image of tsod_pack_conv_weight_bf16x3 and lands in three unpadded planes.
//   * Wave tile 32 x 128 (BN = 128, or 256 with two columns of waves): WAVES_M = BM / 32 waves along M, and for the small tile WAVES_K = 2 waves along K
//     inside a stage (each owns one 16-k chunk; the halves are exchanged through LDS before the epilogue, every wave then
//     finishing 32 x 64).  One phase per stage and wave; an S-deep ring of stages; vmcnt / lgkmcnt counted by hand (the
//     compiler sees none of these memory operations; what it must not do is keep LDS reads of its own pending in the loop).
// Requires one channel segment, Cin % BK == 0 (a stage lies inside one filter tap) and, with a second source, K1 % BK == 0 - or
// (fp16x2, CHAN) a 1x1 filter over several segments, where every 16-byte chunk of a stage has its own place (chunk table).
// Measurements and the instruction-level findings behind this layout: DESIGN.md 4.3, scripts/micro/bf16x3_dma_probe.hip.
typedef int v4i32 __attribute__((ext_vector_type(4)));
#ifdef TSOD_DIAG_NODMA
__device__ int g_nodma_steps = 3;   // (a load the compiler cannot fold: with a constant here it proves the loop's descriptors null)
#endif

__device__ __forceinline__ v4i32 dma_rsrc(const void *ptr, unsigned bytes) {
    v4i32 r; const unsigned long long a = (unsigned long long)ptr;
    r[0] = (int)(unsigned)a; r[1] = (int)((unsigned)(a >> 32) & 0xffff); r[2] = (int)bytes; r[3] = 0x00020000;
    return r;
}
// one wave-instruction: 64 lanes x 16 bytes from per-lane source offsets to LDS [lds_dst, lds_dst + 1024)
#ifndef TSOD_DMA_POLICY_A
#define TSOD_DMA_POLICY_A ""       /* cache policy suffix of the activation / weight DMAs (experiments: " nt", " sc0", " sc1") */
#endif
#ifndef TSOD_DMA_POLICY_B
#define TSOD_DMA_POLICY_B ""
#endif
// (the LDS destination = piece base + ring-slot offset is added INTO m0 by the statement itself: one scalar instruction
//  instead of an add and a move - scalar instructions are what the K loop is short of)
template <int WEIGHTS = 0>
__device__ __forceinline__ void dma16(unsigned voff, v4i32 rsrc, unsigned soff, unsigned lds_dst, unsigned slot_off) {
    if (WEIGHTS)
        asm volatile("s_add_u32 m0, %3, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen" TSOD_DMA_POLICY_B " lds"
                     :: "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst), "s"(slot_off) : "memory", "scc");
    else
        asm volatile("s_add_u32 m0, %3, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen" TSOD_DMA_POLICY_A " lds"
                     :: "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst), "s"(slot_off) : "memory", "scc");
}
// one wave-instruction: 64 lanes x 4 bytes from per-lane source offsets to LDS [lds_dst, lds_dst + 256) (the range words)
__device__ __forceinline__ void dma4(unsigned voff, v4i32 rsrc, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %0, %1, 0 offen lds" :: "v"(voff), "s"(rsrc), "s"(lds_dst) : "memory");
}
// s_waitcnt lgkmcnt(N) that the uses of `x` (an LDS read's destination) cannot be scheduled above
template <int N, typename T> __device__ __forceinline__ void wait_lgkm_for(T &x) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void lds_read4(unsigned &out, unsigned addr) { asm volatile("ds_read_b32 %0, %1" : "=v"(out) : "v"(addr) : "memory"); }
template <int OFF, typename T> __device__ __forceinline__ void lds_read16(T &out, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(out) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void mfma_bf16(f32x16 &c, const bf16x8 &a, const bf16x8 &b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
// One MFMA gap = ONE asm statement (the compiler pads every boundary between two asm statements that share a register with a
// conservative s_nop, so the fewer boundaries the better; inside a statement the order and the spacing are ours):
//   gap_cvt:  MFMA; pk = rne_bf16x2(x0, x1); t0 = f32(pk.lo); t1 = f32(pk.hi)
//   gap_sub:  MFMA; r0 = x0 - t0; r1 = x1 - t1 (exact); one LDS fragment read of the next stage
//   gap_last: MFMA; pk = rne_bf16x2(x0, x1)
// (DO = false only in the `make halfmfma` timing probe: the statement without its MFMA)
template <bool DO = true>
__device__ __forceinline__ void gap_cvt(f32x16 &c, const bf16x8 &a, const bf16x8 &b, float x0, float x1, unsigned &pk, float &t0, float &t1) {
    if constexpr (DO)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\t"
                 "v_cvt_pk_bf16_f32 %1, %6, %7\n\t"
                 "v_lshlrev_b32 %2, 16, %1\n\t"
                 "v_and_b32 %3, 0xffff0000, %1"
                 : "+v"(c), "=&v"(pk), "=&v"(t0), "=&v"(t1) : "v"(a), "v"(b), "v"(x0), "v"(x1));
    else
    asm volatile("v_cvt_pk_bf16_f32 %1, %6, %7\n\t"
                 "v_lshlrev_b32 %2, 16, %1\n\t"
                 "v_and_b32 %3, 0xffff0000, %1"
                 : "+v"(c), "=&v"(pk), "=&v"(t0), "=&v"(t1) : "v"(a), "v"(b), "v"(x0), "v"(x1));
}
template <int OFF, bool DO = true, typename T>
__device__ __forceinline__ void gap_sub(f32x16 &c, const bf16x8 &a, const bf16x8 &b, float &r0, float &r1, float x0, float x1, float t0, float t1,
                                        T &rd, unsigned addr) {
    if constexpr (DO)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\t"
                 "v_sub_f32 %1, %6, %8\n\t"
                 "v_sub_f32 %2, %7, %9\n\t"
                 "ds_read_b128 %3, %10 offset:%11"
                 : "+v"(c), "=&v"(r0), "=&v"(r1), "=&v"(rd) : "v"(a), "v"(b), "v"(x0), "v"(x1), "v"(t0), "v"(t1), "v"(addr), "n"(OFF) : "memory");
    else
    asm volatile("v_sub_f32 %1, %6, %8\n\t"
                 "v_sub_f32 %2, %7, %9\n\t"
                 "ds_read_b128 %3, %10 offset:%11"
                 : "+v"(c), "=&v"(r0), "=&v"(r1), "=&v"(rd) : "v"(a), "v"(b), "v"(x0), "v"(x1), "v"(t0), "v"(t1), "v"(addr), "n"(OFF) : "memory");
}
template <bool DO = true>
__device__ __forceinline__ void gap_last(f32x16 &c, const bf16x8 &a, const bf16x8 &b, float x0, float x1, unsigned &pk) {
    if constexpr (DO)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t"
                 "v_cvt_pk_bf16_f32 %1, %4, %5"
                 : "+v"(c), "=&v"(pk) : "v"(a), "v"(b), "v"(x0), "v"(x1));
    else
    asm volatile("v_cvt_pk_bf16_f32 %1, %4, %5" : "+v"(c), "=&v"(pk) : "v"(a), "v"(b), "v"(x0), "v"(x1));
}
template <int OFF, bool DO = true, typename T>
__device__ __forceinline__ void gap_read(f32x16 &c, const bf16x8 &a, const bf16x8 &b, T &rd, unsigned addr) {
    if constexpr (DO)
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t"
                 "ds_read_b128 %1, %4 offset:%5"
                 : "+v"(c), "=&v"(rd) : "v"(a), "v"(b), "v"(addr), "n"(OFF) : "memory");
    else
    asm volatile("ds_read_b128 %1, %4 offset:%5" : "+v"(c), "=&v"(rd) : "v"(a), "v"(b), "v"(addr), "n"(OFF) : "memory");
}

// ---- "fp16x2" (TSOD_PREC_FP16X2): every operand as TWO fp16 pieces of s * x (hi = rne(s x), lo = rne(s x - hi), s a power of two
// per tensor), THREE piece products per f32 product (lo*hi, hi*lo, hi*hi; the dropped lo*lo is 2^-22) on
// v_mfma_f32_32x32x16_f16, f32 accumulation: the f32 kernel's accuracy with half the MFMAs of bf16x3 while |s x| < 65504.
//   gap2_a:  MFMA; h = rne_f16x2(s * x) for two elements (v_fma_mixlo / mixhi_f16: the product by a power of two is exact)
//   gap2_b:  MFMA; l = rne_f16x2(s * x - h) (the same instructions with h as the f16 addend: exact before the one rounding); one
//            LDS fragment read of the next stage
__device__ __forceinline__ void mfma_f16(f32x16 &c, const bf16x8 &a, const bf16x8 &b) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <int OFF, typename T>
__device__ __forceinline__ void gap2_read(f32x16 &c, const bf16x8 &a, const bf16x8 &b, T &rd, unsigned addr) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\t"
                 "ds_read_b128 %1, %4 offset:%5"
                 : "+v"(c), "=&v"(rd) : "v"(a), "v"(b), "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void gap2_a(f32x16 &c, const bf16x8 &a, const bf16x8 &b, float x0, float x1, float sc, unsigned &h) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %2, %3, %0\n\t"
                 "v_fma_mixlo_f16 %1, %4, %6, 0\n\t"
                 "v_fma_mixhi_f16 %1, %5, %6, 0"
                 : "+v"(c), "=&v"(h) : "v"(a), "v"(b), "v"(x0), "v"(x1), "s"(sc));
}
template <int OFF, typename T>
__device__ __forceinline__ void gap2_b(f32x16 &c, const bf16x8 &a, const bf16x8 &b, float x0, float x1, float sc, unsigned h, unsigned &l,
                                       T &rd, unsigned addr) {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %3, %4, %0\n\t"
                 "v_fma_mixlo_f16 %1, %5, %7, -%8 op_sel_hi:[0,0,1]\n\t"
                 "v_fma_mixhi_f16 %1, %6, %7, -%8 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
                 "ds_read_b128 %2, %9 offset:%10"
                 : "+v"(c), "=&v"(l), "=&v"(rd) : "v"(a), "v"(b), "v"(x0), "v"(x1), "s"(sc), "v"(h), "v"(addr), "n"(OFF) : "memory");
}
#ifdef TSOD_DIAG_MFMA16
// timing probe only (make mfma16; wrong results by design): every v_mfma_f32_32x32x16_bf16 of the K loop replaced by TWO
// v_mfma_f32_16x16x32_bf16 on the same operand registers (the same matrix-pipe cycles and FLOPs, the other shape's register
// traffic and power), everything else of the loop - DMAs, LDS reads, the activation split - unchanged
typedef float f32x4p __attribute__((ext_vector_type(4)));
struct Acc16 { f32x4p lo, hi; };
__device__ __forceinline__ void mfma_bf16(Acc16 &c, const bf16x8 &a, const bf16x8 &b) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %2, %3, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %2, %3, %1" : "+v"(c.lo), "+v"(c.hi) : "v"(a), "v"(b));
}
template <bool DO = true>
__device__ __forceinline__ void gap_cvt(Acc16 &c, const bf16x8 &a, const bf16x8 &b, float x0, float x1, unsigned &pk, float &t0, float &t1) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %5, %6, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %5, %6, %1\n\t"
                 "v_cvt_pk_bf16_f32 %2, %7, %8\n\tv_lshlrev_b32 %3, 16, %2\n\tv_and_b32 %4, 0xffff0000, %2"
                 : "+v"(c.lo), "+v"(c.hi), "=&v"(pk), "=&v"(t0), "=&v"(t1) : "v"(a), "v"(b), "v"(x0), "v"(x1));
}
template <int OFF, bool DO = true, typename T>
__device__ __forceinline__ void gap_sub(Acc16 &c, const bf16x8 &a, const bf16x8 &b, float &r0, float &r1, float x0, float x1, float t0, float t1,
                                        T &rd, unsigned addr) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %5, %6, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %5, %6, %1\n\t"
                 "v_sub_f32 %2, %7, %9\n\tv_sub_f32 %3, %8, %10\n\tds_read_b128 %4, %11 offset:%12"
                 : "+v"(c.lo), "+v"(c.hi), "=&v"(r0), "=&v"(r1), "=&v"(rd) : "v"(a), "v"(b), "v"(x0), "v"(x1), "v"(t0), "v"(t1), "v"(addr), "n"(OFF) : "memory");
}
template <bool DO = true>
__device__ __forceinline__ void gap_last(Acc16 &c, const bf16x8 &a, const bf16x8 &b, float x0, float x1, unsigned &pk) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %3, %4, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %3, %4, %1\n\tv_cvt_pk_bf16_f32 %2, %5, %6"
                 : "+v"(c.lo), "+v"(c.hi), "=&v"(pk) : "v"(a), "v"(b), "v"(x0), "v"(x1));
}
template <int OFF, bool DO = true, typename T>
__device__ __forceinline__ void gap_read(Acc16 &c, const bf16x8 &a, const bf16x8 &b, T &rd, unsigned addr) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %3, %4, %0\n\tv_mfma_f32_16x16x32_bf16 %1, %3, %4, %1\n\tds_read_b128 %2, %5 offset:%6"
                 : "+v"(c.lo), "+v"(c.hi), "=&v"(rd) : "v"(a), "v"(b), "v"(addr), "n"(OFF) : "memory");
}
#endif

constexpr int dma_stage_bytes(int bm, int bk, int bn = 128, int npl = 3) { return bm * bk * 4 + npl * bn * bk * 2; }
// workgroups of an LDS-DMA tile per CU: two where two rings fit the LDS AND two workgroups' waves stay at two per SIMD (the loop
// holds ~214 registers: a third wave on a SIMD does not fit)
constexpr int dma_wgs_per_cu(int bm, int bk, int waves_k, int s, int waves_n, int npl) {
    return (2 * s * dma_stage_bytes(bm, bk, 128 * waves_n, npl) <= 160 * 1024 && 2 * (bm / 32) * waves_k * waves_n <= 8) ? 2 : 1;
}
constexpr int kDmaChanEntries = 640;  // CHAN: 16-byte chunks of one workgroup's K range + ring (K <= 2048: 512 + 8 slots x 8 stages)
constexpr int kDmaTabEntries = 640;   // K-steps of one workgroup's K range + ring depth + 1 (tile_ok_for keeps K / bk + 8 below it)
constexpr unsigned kDmaSecondBit = 0x80000000u;   // validity-mask bit of the second source's 1x1 tap (filter taps use bits 0..30)

// WAVES_N = 2: two columns of waves, BN = 256 - both read (and split) the same activation rows, each its own 128 output
// channels: more FLOP per byte fetched from beyond the CU (the activation stage is shared) at the same 128-row granularity
// NPL = pieces per operand: 3 = bf16x3 (six piece products per k chunk), 2 = fp16x2 (three)
// CHAN: a 1x1 filter over SEVERAL channel segments (HarDNet's concatenated inputs; K = the segments' channels in order, any
// multiple of 4): the 16-byte chunk a lane fetches for a stage lies wherever its segment lies, so besides the stage table there
// is a CHUNK table in LDS - byte offset inside the pixel of every 4-channel chunk of the K range (out-of-range past K: zeros) -
// and a lane reads its chunk's entry beside the stage's (one ds_read_b32 per activation piece and phase).
template <int BM, int BK, int WAVES_K, int S, bool BALANCED, int WAVES_N = 1, int NPL = 3, bool CHAN = false>
__global__ void __launch_bounds__((BM / 32) * WAVES_K * WAVES_N * 64, dma_wgs_per_cu(BM, BK, WAVES_K, S, WAVES_N, NPL))
conv_dma_kernel(const ConvParams p) {
    constexpr int BN = 128 * WAVES_N, TN = 4, WAVES_M = BM / 32, WAVES = WAVES_M * WAVES_K * WAVES_N, THREADS = WAVES * 64;
    static_assert(BK == 16 * WAVES_K, "every wave owns one 16-k chunk of the stage");
    static_assert(WAVES_N == 1 || WAVES_K == 1, "K halves and column halves are not combined");
    constexpr int A_ROW = BK * 4, B_ROW = BK * 2;                    // bytes per LDS row (A raw f32 / one bf16 plane of B)
    constexpr int A_SLOTS = A_ROW / 16, B_SLOTS = B_ROW / 16;        // 16-byte slots per row
    constexpr int A_RPL = 256 / A_ROW, B_RPL = 256 / B_ROW;          // rows per 256-byte bank line: slot ^= (row / RPL) & (SLOTS - 1)
    constexpr int A_BYTES = BM * A_ROW, B_PLANE = BN * B_ROW, STAGE = dma_stage_bytes(BM, BK, BN, NPL);
    constexpr int A_RPP = 1024 / A_ROW, B_RPP = 1024 / B_ROW;        // rows per 1-KiB DMA piece
    constexpr int A_PIECES = BM / A_RPP, B_PIECES = NPL * (BN / B_RPP);
    static_assert(A_PIECES % WAVES == 0, "piece kinds per wave at compile time");
    // this wave's A pieces / all its pieces per stage; when the B pieces do not divide by the waves the last ones are
    // padding (an out-of-range source: zeros into a scratch KiB behind the stage), so that every wave counts the same vmcnt
    constexpr int PA_W = A_PIECES / WAVES, PB_W = (B_PIECES + WAVES - 1) / WAVES, P = PA_W + PB_W;
    constexpr bool B_PAD = B_PIECES % WAVES != 0;
    constexpr int PATCHES = WAVES * 32 * kPatchLD * 4;
    static_assert(S * STAGE >= PATCHES && S * STAGE >= WAVES * (WAVES_K == 4 ? 4 : 2) * 16 * 64 * 4, "epilogue patches / K-half exchange fit the ring");
    // Per-stage operands of the K loop come from a TABLE in LDS, built once per K range: entry j (16 bytes) describes K-step
    // kt_begin + j = {byte offset its filter tap and channel run add to a pixel's base address, byte offset of its k-groups in a
    // row of the weight image, the tap's bit in the rows' validity masks, flags}.  The K loop is bound by the instructions its
    // waves issue, scalar ones above all (measured on the 128x128, 32-k step: 1949 cycles with 69 scalar instructions per wave
    // and phase, 2290 with 104, 1672 with the stage state frozen; the step's MFMAs take 1536), and keeping (step, tap, channel)
    // counters, the tap's address offset and two descriptors up to date on the scalar unit cost ~45 of them; one broadcast
    // ds_read_b128 per phase and a few vector operations on its result cost next to nothing (they issue in MFMA shadows).
    // Tiles whose ring fills the LDS of two workgroups per CU keep the scalar form (TABLE false).
    constexpr int TAB_N = kDmaTabEntries, TAB_BYTES = TAB_N * 16, TAB_OFF = S * STAGE + (B_PAD ? 1024 : 0);
    constexpr int LB_WGS = dma_wgs_per_cu(BM, BK, WAVES_K, S, WAVES_N, NPL);
    constexpr bool TABLE = LB_WGS * (TAB_OFF + TAB_BYTES) <= 160 * 1024;
    // fp16x2: 2 x 256 bytes behind the table for the range words of the two sources (the bf16x3 d128x128 ring fills the LDS of two
    // workgroups per CU exactly and has no use for them)
    constexpr int AMAX_OFF = TAB_OFF + (TABLE ? TAB_BYTES : 0), AMAX_LDS = NPL == 2 ? 512 : 0;
    constexpr int CH_OFF = AMAX_OFF + AMAX_LDS, CH_BYTES = CHAN ? kDmaChanEntries * 4 : 0;
    static_assert(!CHAN || (TABLE && NPL == 2 && !BALANCED && LB_WGS * (CH_OFF + CH_BYTES) <= 160 * 1024), "the chunk table rides beside the stage table");
    __shared__ __align__(16) unsigned char lds[CH_OFF + CH_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WAVES_M, wk = (wave / WAVES_M) % WAVES_K, wn = wave / (WAVES_M * WAVES_K);
    // ---- work of this workgroup: ONE (tile, K range) under the uniform schedules (the map of conv_igemm_kernel), or the K-steps
    // [w q, (w+1) q) of the launch's tile-major K-step sequence under the balanced schedule (p.sk_q > 0): a run of up to two
    // partial tiles with whole tiles between them, every workgroup the same number of K-steps whatever the tile count
    // (BALANCED is a template argument: the loop over segments keeps enough scalar state alive to push the K loop's own
    //  scalars out of the SGPR file - the uniform-schedule instantiation has no such loop)
    constexpr bool balanced = BALANCED;
#ifdef TSOD_CLOCK_DIAG
    long long dg_rt0 = 0, dg_pro = 0, dg_loop = 0, dg_epi = 0, dg_c = 0, dg_steps = 0, dg_segs = 0;
    const bool dg_on = g_dma_stamps != nullptr && tid == 0 && blockIdx.x < 8192;
    if (dg_on) dg_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    long g = 0, g_end = 1;
    if (balanced) {
        g = (long)blockIdx.x * p.sk_q;
        g_end = min(g + (long)p.sk_q, (long)p.tiles_m * p.tiles_n * p.ksteps);
    }
    // fp16x2 activation scale: the descriptor's static exponent, or - range words - the exponent the tensor's abs-max calls for.
    // The words come in by LDS-DMA like everything else (wave 0, BEFORE the first ring fill: the oldest vector-memory operation,
    // so every counted wait of the prologue covers it) and are read behind the prologue's barrier; nothing waits for them alone.
    float a_scale = p.a_scale;
    bool amax_pending = false;
    if constexpr (NPL == 2) {
        if (p.amax_in != nullptr) {
            amax_pending = true;
            if (wave == 0) {
                const unsigned l0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char *)lds + AMAX_OFF;
                dma4((unsigned)lane * TSOD_AMAX_STRIDE, dma_rsrc(p.amax_in, TSOD_AMAX_BYTES), (unsigned)__builtin_amdgcn_readfirstlane((int)l0));
                // (no second source: the same words again, so that the reader below has no branch)
                dma4((unsigned)lane * TSOD_AMAX_STRIDE, dma_rsrc(p.amax_in2 ? p.amax_in2 : p.amax_in, TSOD_AMAX_BYTES),
                     (unsigned)__builtin_amdgcn_readfirstlane((int)(l0 + 256)));
            }
        }
    }
  // Balanced schedule: the segments of the range [g, g_end) are taken LAST FIRST.  A workgroup's range is about one K sweep long and
  // starts somewhere inside a tile: in sequence order it would run that tile's K tail and then the next tile's K head, every
  // workgroup at another K position at any moment - a weight block wanted by one is in nobody's L2 (measured at batch 8:
  // layer4's 3x3 convs moved 802 MB for 43.8 MB of operands, L2 hit rate 17 %, and ran at 5.8 TB/s: bound by their own
  // re-fetches).  Head first, tail second, the K position of EVERY workgroup is close to the time since launch: the workgroups
  // of an XCD sweep K together and share each weight block as it passes, as whole tiles in lock-step do.  Segments are independent
  // (own accumulators, own slab): the order changes no result.
  for (; g < g_end;) {
    int tile_id, kt_begin, kt_end;
    SliceMap sm;
    if (balanced) {
        tile_id = (int)((g_end - 1) / p.ksteps);
        const long tile_k0 = (long)tile_id * p.ksteps;
        const long seg_lo = g > tile_k0 ? g : tile_k0;
        kt_begin = (int)(seg_lo - tile_k0);
        kt_end = (int)(g_end - tile_k0);
        const int first = (int)(tile_k0 / p.sk_q), last = (int)((tile_k0 + p.ksteps - 1) / p.sk_q);
        sm = {first == last ? -1 : (int)blockIdx.x - first, last - first + 1, tile_id, first, (long)p.sk_q, tile_k0};
        g_end = seg_lo;
    } else {
        int z;
        work_item(p, tile_id, z);
        kt_begin = z < 0 ? 0 : z * p.ksteps_per_split;
        kt_end = z < 0 ? p.ksteps : min(p.ksteps, kt_begin + p.ksteps_per_split);
        sm = {z, p.split, tile_id - p.dp_tiles, -1, 0, 0};
        g = g_end;
    }
    const int tn_i = tile_id % p.tiles_n, tm_i = tile_id / p.tiles_n;
    const int m0 = tm_i * BM, n0 = tn_i * BN;
    const int nk = kt_end - kt_begin;
    __syncthreads();                                             // (a previous segment's epilogue is done with the LDS)

    v4i32 rs_in = dma_rsrc(p.in, p.in_bytes), rs_w = dma_rsrc(p.w, p.w_bytes);
    v4i32 rs_in2 = dma_rsrc(p.in2 ? p.in2 : p.in, p.in2 ? p.in2_bytes : 0u);
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char *)lds;
    const int kgroups = (p.K + 7) / 8;                            // (the weight image pads K to whole 8-k groups)

    // ---- this wave's DMA pieces: q = WAVES * i + wave; piece i is an A piece for i < PA_W (1 or 2 for every tile shape; the
    // rows' state in two named structs: an array of them ends up in scratch memory behind the source select below, and a
    // scratch access is a VMEM operation that would sit in the middle of the hand-counted vmcnt)
    static_assert(PA_W == 1 || PA_W == 2, "one or two A pieces per wave and stage");
    struct ARow { unsigned base, base2; int ih0, iw0; unsigned mask; int lslot; };   // mask: bit t = filter tap t lies inside the image for this row; lslot (CHAN): the 16-byte chunk of a stage this lane fetches
    unsigned b_voff[P - PA_W], ldst[P];
    auto make_arow = [&](int i) {
        ARow ar;
        const int q = WAVES * i + wave;
        const int row = q * A_RPP + lane / A_SLOTS, phys = lane % A_SLOTS;
        ar.lslot = phys ^ ((row / A_RPL) & (A_SLOTS - 1));
        // (CHAN: the chunk's place inside the pixel comes from the chunk table, neither the first segment's offset nor the slot's)
        const unsigned lofs = CHAN ? 0u : (unsigned)(ar.lslot * 16);
        const int m = m0 + row;
        if (m < p.M) {
            const int ow = m % p.OW, t = m / p.OW, oh = t % p.OH, img = t / p.OH;
            ar.ih0 = oh * p.stride - p.pad_h;
            ar.iw0 = ow * p.stride - p.pad_w;
            // may wrap below zero for border rows; adding a valid tap's delta brings it back in range
            ar.base = (unsigned)((((long)img * p.H + ar.ih0) * p.W + ar.iw0) * p.in_pitch + (CHAN ? 0 : p.seg_off[0])) * 4u + lofs;
            ar.base2 = p.c2 > 0 ? (unsigned)((((long)img * p.H2 + oh * p.stride2) * p.W2 + ow * p.stride2) * p.in2_pitch + p.in2_off) * 4u + lofs
                                : kOOB;
        } else {
            ar.ih0 = INT_MIN / 2; ar.iw0 = INT_MIN / 2; ar.base = 0; ar.base2 = kOOB;
        }
        ar.mask = 0;
        if (TABLE && m < p.M) {
            ar.mask = kDmaSecondBit;
            for (int kh = 0, t = 0; kh < p.KH; ++kh)
                for (int kw = 0; kw < p.KW; ++kw, ++t)
                    if ((unsigned)(ar.ih0 + kh) < (unsigned)p.H && (unsigned)(ar.iw0 + kw) < (unsigned)p.W) ar.mask |= 1u << t;
        }
        return ar;
    };
    const ARow ar0 = make_arow(0), ar1 = make_arow(PA_W - 1);
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int q = WAVES * i + wave;
        if (i < PA_W) {
            ldst[i] = q * 1024;
        } else {
            const int qb = WAVES * (i - PA_W) + wave, plane = qb / (BN / B_RPP), rb = qb % (BN / B_RPP);
            const int row = rb * B_RPP + lane / B_SLOTS, phys = lane % B_SLOTS, logical = phys ^ ((row / B_RPL) & (B_SLOTS - 1));
            const int n = n0 + row;
            const bool real = qb < B_PIECES;
            b_voff[i - PA_W] = (real && n < p.Cout) ? ((unsigned)n * (unsigned)kgroups + (unsigned)logical) * (16u * NPL) + (unsigned)plane * 16u : kOOB;
            ldst[i] = real ? A_BYTES + plane * B_PLANE + rb * 1024 : S * STAGE;
        }
        ldst[i] = __builtin_amdgcn_readfirstlane(ldst[i] + lds0);
        asm volatile("" : "+s"(ldst[i]));                        // (opaque: dma16 adds a possibly literal slot offset to it in one s_add)
    }
    // State of the stage being ISSUED, all wave-uniform (scalar unit): its K-step, filter tap, channel base, the byte offset
    // that tap adds to a pixel's base address, which source it reads and the two descriptors (zero records once the stage lies
    // past this workgroup's K range: a null DMA, no memory traffic, zeros written, so that the counted waits never change).
    // Step order of the first source: (tap, channels) = the order of k in the weight image, or under p.cmajor (32-channel
    // block, tap, 16-channel half when BK == 16): the same steps in another sequence, so only which k-groups of the image a
    // step fetches changes (u_woff below).  ONE update serves both orders: a tap is done when the channel offset has run
    // p.ci_wrap channels past the block base u_cb (ci_wrap = Cin with u_cb = 0: tap-major; 32 with u_cb moving on after the last
    // tap: block-major).  The K loop is bound by the instructions its two waves per SIMD issue, not by the matrix pipe (one
    // more scalar instruction per stage costs 8 cycles of a 1950-cycle step: measured 1949 -> 2290 cycles with 40 more), so this
    // state is kept to plain scalar integer selects.
#ifdef TSOD_DIAG_NODMA
    const int nodma_steps = __builtin_amdgcn_readfirstlane(*(volatile int *)&g_nodma_steps);
#endif
    // (channel blocks of the block-major order: 32 channels, or a whole stage where a stage is longer than that)
    constexpr int CBLK = BK > 32 ? BK : 32;
    int u_kt = kt_begin, u_kh, u_kw, u_ci, u_cb;
    {
        constexpr int SUB = CBLK / BK;
        const int blk = kt_begin / SUB, sub = kt_begin - blk * SUB, taps = p.KH * p.KW;
        const int cb = blk / taps, tap_c = blk - cb * taps;      // channel-block-major
        const int kb = kt_begin * BK, tap_t = kb / p.Cin;        // tap-major
        const int tap = p.cmajor ? tap_c : tap_t;
        u_cb = p.cmajor ? cb * CBLK : 0;
        u_ci = p.cmajor ? cb * CBLK + sub * BK : kb - tap_t * p.Cin;
        u_kh = tap / p.KW;
        u_kw = tap - u_kh * p.KW;
    }
    unsigned u_delta, u_woff;
    bool u_second;
    v4i32 u_rs_a = rs_in, u_rs_w = rs_w;
    const int k1_steps = CHAN ? 0x7fffffff : p.K1 / BK;          // (K1 % BK == 0: tile_ok_for; CHAN: no second source, K any multiple of 4)
    auto stage_state = [&]() {
#ifdef TSOD_DIAG_NODMA
        const bool live = u_kt < kt_begin + nodma_steps;         // timing probe only (make nodma): null DMAs after the ring's first fill
#else
        const bool live = u_kt < kt_end;
#endif
        u_second = u_kt >= k1_steps;                              // second source: a 1x1 tap, always inside the image
        const int d1 = ((u_kh * p.W + u_kw) * p.in_pitch + u_ci) * 4, d2 = (u_kt - k1_steps) * (BK * 4);
        // byte offset of k in a row of the weight image = (k / 8) * 48 = 6 k (k % 8 == 0); first source k = tap * Cin + channel
        const int w1 = ((u_kh * p.KW + u_kw) * p.Cin + u_ci) * (2 * NPL), w2 = u_kt * (BK * 2 * NPL);
        u_delta = (unsigned)(u_second ? d2 : d1);
        u_woff = (unsigned)(u_second ? w2 : w1);
        u_rs_a[0] = u_second ? rs_in2[0] : rs_in[0];
        u_rs_a[1] = u_second ? rs_in2[1] : rs_in[1];
        u_rs_a[2] = live ? (u_second ? rs_in2[2] : rs_in[2]) : 0;
        u_rs_w[2] = live ? rs_w[2] : 0;
    };
    stage_state();
    auto advance_stage = [&]() {
#ifdef TSOD_DIAG_NOADVANCE
        return;                                                  // timing probe only (make noadvance): every stage fetches the first one again
#endif
        ++u_kt;
        // (selects, not nested updates: written as branches these scalars end up in scratch memory, and a scratch access is a
        //  VMEM operation in the middle of the hand-counted vmcnt)
        const int ci1 = u_ci + BK, kw1 = u_kw + 1, kh1 = u_kh + 1;
        const bool tap_done = ci1 - u_cb >= p.ci_wrap;
        const bool row_done = tap_done & (kw1 == p.KW);
        const bool all_taps = row_done & (kh1 == p.KH);
        u_cb = all_taps ? u_cb + CBLK : u_cb;
        u_ci = tap_done ? u_cb : ci1;
        u_kw = tap_done ? (row_done ? 0 : kw1) : u_kw;
        u_kh = row_done ? (all_taps ? 0 : kh1) : u_kh;
        stage_state();
    };
    // piece I of the stage being issued into ring slot `slot` (branch-free: selects only; I is a compile-time constant so
    // that the per-piece state stays in registers)
    // What the table entry of the stage being issued says (TABLE; set at the top of a phase): the entry itself (vector
    // registers, the same in every lane), the weight offset, which source, and the two descriptors
    u32x4 t_e = {0u, 0u, 0u, 0u};
    unsigned t_woff = 0;
    bool t_second = false;
    v4i32 t_rs_a = rs_in, t_rs_w = rs_w;
    unsigned t_cd[2] = {0u, 0u};                                  // CHAN: the chunk-table entries of this lane's activation pieces
    auto issue_piece = [&](auto I, auto FROM_TABLE, unsigned slot_off) {
        constexpr int i = decltype(I)::value;
        constexpr bool from_table = decltype(FROM_TABLE)::value;
        if constexpr (i < PA_W) {
            const ARow ar = i == 0 ? ar0 : ar1;
            if constexpr (CHAN && from_table) {
                const bool ok = (ar.mask & t_e.z) != 0 && t_cd[i] != kOOB;
                dma16(ok ? ar.base + t_cd[i] : kOOB, t_rs_a, 0u, ldst[i], slot_off);
            } else if constexpr (CHAN) {
                // the ring's first fill (no tables yet): this lane's chunk of stage u_kt straight from the segment list
                const int k = u_kt * BK + 4 * ar.lslot;
                const bool ok = ar.ih0 >= 0 && k < p.K;            // (1x1, no padding: a row < M is inside the image)
                dma16(ok ? ar.base + (unsigned)(seg_channel(p, k) * 4) : kOOB, u_rs_a, 0u, ldst[i], slot_off);
            } else if constexpr (from_table) {
                // a row fetches when the stage's tap lies inside the image for it (rows past M and every row of a stage past the
                // K range have no bit in common with the entry: the out-of-range offset, no memory traffic, zeros written)
                const bool ok = (ar.mask & t_e.z) != 0;
                const unsigned v = (t_second ? ar.base2 : ar.base) + t_e.x;
                dma16(ok ? v : kOOB, t_rs_a, 0u, ldst[i], slot_off);
            } else {
                const int ih = ar.ih0 + u_kh, iw = ar.iw0 + u_kw;
                const bool ok1 = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                // (second source: rows >= M carry the out-of-range base, plus a delta it may wrap into the tensor: such rows read
                //  real bytes instead of zeros, which is harmless - they feed output rows that are never stored)
                const unsigned v1 = ok1 ? ar.base + u_delta : kOOB, v2 = ar.base2 + u_delta;
                dma16(u_second ? v2 : v1, u_rs_a, 0u, ldst[i], slot_off);
            }
        } else {
            // (a padding piece has the out-of-range source everywhere; its destination is the scratch KiB behind the ring, which
            //  the slot offset must not move: ldst - lds0 == S * STAGE marks it)
            const unsigned so = (B_PAD && ldst[i] - lds0 == (unsigned)(S * STAGE)) ? 0u : slot_off;
            if constexpr (from_table) dma16<1>(b_voff[i - PA_W], t_rs_w, t_woff, ldst[i], so);
            else dma16<1>(b_voff[i - PA_W], u_rs_w, u_woff, ldst[i], so);
        }
    };

    float a_scale_s = p.a_scale;                                  // fp16x2: the activation scale as a scalar operand of the loop's statements
    f32x16 acc[1][TN];
#ifdef TSOD_DIAG_MFMA16
    Acc16 pacc[TN];                                               // the probe's accumulators (acc is zeroed behind the loop instead)
#pragma unroll
    for (int j = 0; j < TN; ++j) pacc[j].lo = pacc[j].hi = f32x4p{0.f, 0.f, 0.f, 0.f};
#else
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][j][e] = 0.f;
#endif

    // fragment addresses inside a stage: lane (h, r) holds k = 16 wk + 8 h .. + 7 of row r
    const int h = lane >> 5, r = lane & 31;
    unsigned a_addr[2], b_addr[TN];
    {
        const int row = wm * 32 + r, s0 = wk * 4 + 2 * h, sw = (row / A_RPL) & (A_SLOTS - 1);
        a_addr[0] = lds0 + row * A_ROW + ((s0 ^ sw) * 16);
        a_addr[1] = lds0 + row * A_ROW + (((s0 + 1) ^ sw) * 16);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wn * 128 + j * 32 + r, sl = wk * 2 + h, sw = (row / B_RPL) & (B_SLOTS - 1);
        b_addr[j] = lds0 + A_BYTES + row * B_ROW + ((sl ^ sw) * 16);
    }

    struct Frags { bf16x8 a[NPL], b[TN][NPL]; };
    // product q = A piece PA[q] x B piece PB[q] (smallest first); fp16x2: lo*hi, hi*lo, hi*hi
    constexpr int PA[6] = {NPL == 3 ? 2 : 1, 0, NPL == 3 ? 1 : 0, 1, 0, 0}, PB[6] = {0, NPL == 3 ? 2 : 1, NPL == 3 ? 1 : 0, 0, 1, 0};

    // One phase.  MFMA n = 4 q + j (product q, accumulator j) runs on `cur`; the 14 LDS reads of the next stage go out in the
    // first gaps (A raw first), its A fragment is split behind MFMAs 4..23 (one pair per five MFMAs: h = rne(x); x -= h;
    // m = rne(x); x -= m; l = rne(x); at most 3 VALU per MFMA gap, which is what issues in an MFMA's shadow), and this
    // wave's DMA pieces of the stage being issued go out one per few gaps.  `soff` = byte offset of the ring slot that is read.
    // `tab_ptr` (TABLE): LDS address of the table entry of the stage this phase issues; it moves on by one entry per phase
    unsigned tab_ptr = lds0 + TAB_OFF + S * 16;
    // (CHAN) LDS addresses of this lane's chunk-table entries for the stage a phase issues: one per activation piece, moving on by a
    // stage's A_SLOTS entries per phase
    unsigned ch_ptr0 = lds0 + CH_OFF + (unsigned)((S * A_SLOTS + ar0.lslot) * 4), ch_ptr1 = lds0 + CH_OFF + (unsigned)((S * A_SLOTS + ar1.lslot) * 4);
    auto phase = [&](const Frags &cur, Frags &nxt, unsigned soff, int dma_slot) {
        float4 raw0, raw1;
        unsigned hh[4], mm[4], ll[4];
        float t0, t1, r0, r1, q0, q1;
        const unsigned slot_off = (unsigned)(dma_slot * STAGE);
      if constexpr (NPL == 2) {
        // ---- fp16x2: 12 MFMAs (n = 4 q + j: product q of accumulator j).  The 10 LDS reads of the next stage go out in the first
        // gaps (A raw first, then the hi plane of B) and behind every second MFMA of the rest (lo plane); the A fragment is
        // split behind MFMAs 4..11, one pair of elements per two MFMAs; this wave's DMA pieces one per pair.
#define TSOD_DMA2(I) do { if constexpr ((I) < P) issue_piece(std::integral_constant<int, (I)>{}, std::integral_constant<bool, TABLE>{}, slot_off); } while (0)
#define TSOD_MF2(n) acc[0][(n) & 3], cur.a[PA[(n) >> 2]], cur.b[(n) & 3][PB[(n) >> 2]]
        static_assert(NPL != 2 || P <= 6, "DMA slots of an fp16x2 phase");
        if constexpr (TABLE) {
            lds_read16<0>(t_e, tab_ptr);
            tab_ptr += 16;
            if constexpr (CHAN) {                                // (issued before the raw rows: the counted wait below covers them too)
                lds_read4(t_cd[0], ch_ptr0);
                if constexpr (PA_W == 2) lds_read4(t_cd[1], ch_ptr1);
                ch_ptr0 += A_SLOTS * 4;
                ch_ptr1 += A_SLOTS * 4;
            }
        }
        lds_read16<0>(raw0, a_addr[0] + soff);
        lds_read16<0>(raw1, a_addr[1] + soff);
        gap2_read<0>(TSOD_MF2(0), nxt.b[0][0], b_addr[0] + soff);
        gap2_read<0>(TSOD_MF2(1), nxt.b[1][0], b_addr[1] + soff);
        gap2_read<0>(TSOD_MF2(2), nxt.b[2][0], b_addr[2] + soff);
        gap2_read<0>(TSOD_MF2(3), nxt.b[3][0], b_addr[3] + soff);
        if constexpr (TABLE) {
            wait_lgkm_for<4>(t_e);
            // (CHAN: the chunk entries were read by asm statements too - what is computed from them must follow the wait, and only a
            //  statement the compiler cannot move ties that down)
            if constexpr (CHAN) asm volatile("" : "+v"(t_cd[0]), "+v"(t_cd[1]));
            const unsigned flags = (unsigned)__builtin_amdgcn_readfirstlane((int)t_e.w);
            t_woff = (unsigned)__builtin_amdgcn_readfirstlane((int)t_e.y);
            t_second = (flags & 1u) != 0;
            t_rs_w[0] = __builtin_amdgcn_readfirstlane(rs_w[0]);
            t_rs_w[1] = __builtin_amdgcn_readfirstlane(rs_w[1]);
            t_rs_w[2] = __builtin_amdgcn_readfirstlane((flags & 2u) ? 0 : rs_w[2]);
            t_rs_w[3] = __builtin_amdgcn_readfirstlane(rs_w[3]);
            t_rs_a[0] = __builtin_amdgcn_readfirstlane(t_second ? rs_in2[0] : rs_in[0]);
            t_rs_a[1] = __builtin_amdgcn_readfirstlane(t_second ? rs_in2[1] : rs_in[1]);
            t_rs_a[2] = __builtin_amdgcn_readfirstlane(t_second ? rs_in2[2] : rs_in[2]);
            t_rs_a[3] = __builtin_amdgcn_readfirstlane(rs_in[3]);
            TSOD_DMA2(0);
        } else {
            TSOD_DMA2(0);
            wait_lgkm<4>();
        }
#define TSOD_SPLIT2(N0, X0, X1, G)                                                                        \
        gap2_a(TSOD_MF2(N0), X0, X1, a_scale_s, hh[G]);                                                    \
        gap2_b<B_PLANE>(TSOD_MF2(N0 + 1), X0, X1, a_scale_s, hh[G], ll[G], nxt.b[G][1], b_addr[G] + soff); \
        TSOD_DMA2(1 + G);
        TSOD_SPLIT2(4, raw0.x, raw0.y, 0)
        TSOD_SPLIT2(6, raw0.z, raw0.w, 1)
        TSOD_SPLIT2(8, raw1.x, raw1.y, 2)
        TSOD_SPLIT2(10, raw1.z, raw1.w, 3)
        TSOD_DMA2(5);                                            // (the 64-row tile with four K quarters: six pieces per wave and stage)
#undef TSOD_SPLIT2
#undef TSOD_MF2
#undef TSOD_DMA2
        if constexpr (!TABLE) advance_stage();
        nxt.a[0] = __builtin_bit_cast(bf16x8, (u32x4{hh[0], hh[1], hh[2], hh[3]}));
        nxt.a[1] = __builtin_bit_cast(bf16x8, (u32x4{ll[0], ll[1], ll[2], ll[3]}));
      } else {
#define TSOD_DMA(I) do { if constexpr ((I) < P) issue_piece(std::integral_constant<int, (I)>{}, std::integral_constant<bool, TABLE>{}, slot_off); } while (0)
#ifdef TSOD_DIAG_MFMA16
#define TSOD_MF(n) pacc[(n) & 3], cur.a[PA[(n) >> 2]], cur.b[(n) & 3][PB[(n) >> 2]]
#else
#define TSOD_MF(n) acc[0][(n) & 3], cur.a[PA[(n) >> 2]], cur.b[(n) & 3][PB[(n) >> 2]]
#endif
        if constexpr (TABLE) {
            lds_read16<0>(t_e, tab_ptr);                         // oldest of this phase's LDS reads
            tab_ptr += 16;
        }
        lds_read16<0>(raw0, a_addr[0] + soff);
        lds_read16<0>(raw1, a_addr[1] + soff);
#ifdef TSOD_DIAG_HALFMFMA
#define TSOD_DO(n) (((n) >> 2) % 2 == 0)     /* timing probe only (make halfmfma): three of the six piece products, 12 MFMAs per phase */
#else
#define TSOD_DO(n) true
#endif
        gap_read<0 * B_PLANE, TSOD_DO(0)>(TSOD_MF(0), nxt.b[0][0], b_addr[0] + soff);
        gap_read<0 * B_PLANE, TSOD_DO(1)>(TSOD_MF(1), nxt.b[1][0], b_addr[1] + soff);
        gap_read<0 * B_PLANE, TSOD_DO(2)>(TSOD_MF(2), nxt.b[2][0], b_addr[2] + soff);
        gap_read<0 * B_PLANE, TSOD_DO(3)>(TSOD_MF(3), nxt.b[3][0], b_addr[3] + soff);
        if constexpr (TABLE) {
            wait_lgkm_for<4>(t_e);                               // the entry, raw0, raw1 have landed (four younger reads may be out)
            const unsigned flags = (unsigned)__builtin_amdgcn_readfirstlane((int)t_e.w);
            t_woff = (unsigned)__builtin_amdgcn_readfirstlane((int)t_e.y);
            t_second = (flags & 1u) != 0;
            // (readfirstlane: the divergence analysis loses sight of the uniformity of this state, and a descriptor must sit
            //  in scalar registers; on values that are scalar already it folds away)
            t_rs_w[0] = __builtin_amdgcn_readfirstlane(rs_w[0]);
            t_rs_w[1] = __builtin_amdgcn_readfirstlane(rs_w[1]);
            t_rs_w[2] = __builtin_amdgcn_readfirstlane((flags & 2u) ? 0 : rs_w[2]);   // a stage past the K range: null weight DMAs as well
            t_rs_w[3] = __builtin_amdgcn_readfirstlane(rs_w[3]);
            t_rs_a[0] = __builtin_amdgcn_readfirstlane(t_second ? rs_in2[0] : rs_in[0]);
            t_rs_a[1] = __builtin_amdgcn_readfirstlane(t_second ? rs_in2[1] : rs_in[1]);
            t_rs_a[2] = __builtin_amdgcn_readfirstlane(t_second ? rs_in2[2] : rs_in[2]);
            t_rs_a[3] = __builtin_amdgcn_readfirstlane(rs_in[3]);
            TSOD_DMA(0);
        } else {
            TSOD_DMA(0);
            wait_lgkm<4>();                                     // raw0, raw1 have landed (four younger reads may be out)
        }
#ifdef TSOD_DIAG_NOSPLIT
        // timing probe only (make nosplit; wrong results by design): the same MFMAs, LDS reads and DMAs without the VALU work of
        // the activation split - what a K loop fed with PRE-SPLIT activations could reach
#define TSOD_SPLIT_GROUP(N0, X0, X1, G, PL, J0, J1)                                                        \
        mfma_bf16(TSOD_MF(N0));                                                                            \
        gap_read<PL * B_PLANE>(TSOD_MF(N0 + 1), nxt.b[J0][PL], b_addr[J0] + soff);                         \
        mfma_bf16(TSOD_MF(N0 + 2));                                                                        \
        if constexpr (P > 5) TSOD_DMA(1 + 2 * G);                                                          \
        gap_read<PL * B_PLANE>(TSOD_MF(N0 + 3), nxt.b[J1][PL], b_addr[J1] + soff);                         \
        mfma_bf16(TSOD_MF(N0 + 4));                                                                        \
        hh[G] = __float_as_uint(X0) & 0x3f803f80u; mm[G] = __float_as_uint(X1) & 0x3f803f80u; ll[G] = hh[G];   \
        if constexpr (P > 5) TSOD_DMA(2 + 2 * G); else TSOD_DMA(1 + G);
#else
#define TSOD_SPLIT_GROUP(N0, X0, X1, G, PL, J0, J1)                                                        \
        gap_cvt<TSOD_DO(N0)>(TSOD_MF(N0), X0, X1, hh[G], t0, t1);                                          \
        gap_sub<PL * B_PLANE, TSOD_DO(N0 + 1)>(TSOD_MF(N0 + 1), r0, r1, X0, X1, t0, t1, nxt.b[J0][PL], b_addr[J0] + soff);  \
        gap_cvt<TSOD_DO(N0 + 2)>(TSOD_MF(N0 + 2), r0, r1, mm[G], t0, t1);                                  \
        if constexpr (P > 5) TSOD_DMA(1 + 2 * G);                                                          \
        gap_sub<PL * B_PLANE, TSOD_DO(N0 + 3)>(TSOD_MF(N0 + 3), q0, q1, r0, r1, t0, t1, nxt.b[J1][PL], b_addr[J1] + soff);  \
        gap_last<TSOD_DO(N0 + 4)>(TSOD_MF(N0 + 4), q0, q1, ll[G]);                                         \
        if constexpr (P > 5) TSOD_DMA(2 + 2 * G); else TSOD_DMA(1 + G);
#endif
        TSOD_SPLIT_GROUP(4, raw0.x, raw0.y, 0, 2, 0, 1)
        TSOD_SPLIT_GROUP(9, raw0.z, raw0.w, 1, 2, 2, 3)
        TSOD_SPLIT_GROUP(14, raw1.x, raw1.y, 2, 1, 0, 1)
        TSOD_SPLIT_GROUP(19, raw1.z, raw1.w, 3, 1, 2, 3)
#undef TSOD_MF
#undef TSOD_SPLIT_GROUP
        static_assert(P <= 9, "DMA slots of a phase");
        if constexpr (!TABLE) advance_stage();
        nxt.a[0] = __builtin_bit_cast(bf16x8, (u32x4{hh[0], hh[1], hh[2], hh[3]}));
        nxt.a[1] = __builtin_bit_cast(bf16x8, (u32x4{mm[0], mm[1], mm[2], mm[3]}));
        nxt.a[NPL - 1] = __builtin_bit_cast(bf16x8, (u32x4{ll[0], ll[1], ll[2], ll[3]}));
      }
    };
    // next stage visible to every wave; the ring slot of the stage that is now in registers may be refilled
    auto turn = [&]() {
        wait_vm<(S - 2) * P>();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

#ifdef TSOD_CLOCK_DIAG
    if (dg_on) { dg_c = __builtin_amdgcn_s_memtime(); dg_steps += nk; ++dg_segs; }
#endif
    if (nk > 0) {
        // prologue: every ring slot filled (stages kt_begin .. kt_begin + S - 1), stage 0 visible, its fragments in X
#undef TSOD_DMA
#define TSOD_DMA(I) do { if constexpr ((I) < P) issue_piece(std::integral_constant<int, (I)>{}, std::false_type{}, slot_off); } while (0)
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const unsigned slot_off = (unsigned)(s * STAGE);
            TSOD_DMA(0); TSOD_DMA(1); TSOD_DMA(2); TSOD_DMA(3); TSOD_DMA(4); TSOD_DMA(5); TSOD_DMA(6); TSOD_DMA(7); TSOD_DMA(8);
            advance_stage();
        }
        // the stage table of this K range (entries 0 .. nk + S: the loop issues stages S .. nk - 1 + S, one more when nk is odd;
        // built while the prologue's DMAs are in flight, visible to every wave behind the barrier that ends the prologue)
        if constexpr (TABLE) {
            constexpr int SUB = CBLK / BK;
            const int taps = p.KH * p.KW;
            for (int j = tid; j <= nk + S; j += THREADS) {
#if defined(TSOD_DIAG_NOADVANCE)
                const int step = kt_begin;
#else
                const int step = kt_begin + j;
#endif
#if defined(TSOD_DIAG_NODMA)
                const bool live = j < nodma_steps;
#else
                const bool live = step < kt_end;
#endif
                u32x4 e = {0u, 0u, 0u, 2u};                       // dead: no row fetches, null weight descriptor
                if (live) {
                    if (step >= k1_steps) {
                        e = {(unsigned)((step - k1_steps) * (BK * 4)), (unsigned)(step * (BK * 2 * NPL)), kDmaSecondBit, 1u};
                    } else {
                        int tap, ci;
                        if (p.cmajor) {
                            const int blk = step / SUB, cb = blk / taps;
                            tap = blk - cb * taps;
                            ci = cb * CBLK + (step - blk * SUB) * BK;
                        } else {
                            tap = step * BK / p.Cin;
                            ci = step * BK - tap * p.Cin;
                        }
                        const int kh = tap / p.KW, kw = tap - kh * p.KW;
                        e = {(unsigned)(((kh * p.W + kw) * p.in_pitch + ci) * 4), (unsigned)((tap * p.Cin + ci) * (2 * NPL)), 1u << tap, 0u};
                    }
                }
                *reinterpret_cast<u32x4 *>(lds + TAB_OFF + j * 16) = e;
            }
            if constexpr (CHAN) {
                // the chunk table of the same steps: byte offset inside the pixel of the 4 channels k .. k + 3 (out of range past K or
                // past this K range: zeros land)
                for (int idx = tid; idx < (nk + S + 1) * A_SLOTS; idx += THREADS) {
                    const int step = kt_begin + idx / A_SLOTS, k = step * BK + 4 * (idx % A_SLOTS);
                    *reinterpret_cast<unsigned *>(lds + CH_OFF + idx * 4) = (step < kt_end && k < p.K) ? (unsigned)(seg_channel(p, k) * 4) : kOOB;
                }
            }
        }
        wait_vm<(S - 1) * P>();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifdef TSOD_CLOCK_DIAG
        if (dg_on) { const long long c = __builtin_amdgcn_s_memtime(); dg_pro += c - dg_c; dg_c = c; }
#endif
        if constexpr (NPL == 2) {
            if (amax_pending) {                                  // (wave-uniform; once per workgroup: the scale is the launch's)
                const unsigned *wds = reinterpret_cast<const unsigned *>(lds + AMAX_OFF);
                const unsigned b1 = wds[lane], b2 = wds[64 + lane];
                float acc_unused;
                fp16x2_scales_from_bits(tsod_amax_reduce_bits(b2 > b1 ? b2 : b1), 0, a_scale, acc_unused);
                amax_pending = false;
            }
        }
        a_scale_s = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(a_scale)));
        // every scalar the loop reads must BE in its register here: a kernel-argument load of the compiler's own that is still
        // pending at the loop head costs a conservative lgkmcnt(0) inside the phase, in front of the table entry's first use -
        // which then waits for the four fragment reads issued behind the entry as well (seen in the balanced instantiation once
        // the range words had lengthened the prologue: the descriptors were re-loaded late)
        asm volatile("" : "+s"(rs_w), "+s"(rs_in), "+s"(rs_in2), "+s"(a_scale_s));
        Frags X, Y;
        {
            const float4 raw0 = *reinterpret_cast<const float4 *>(lds + (a_addr[0] - lds0));
            const float4 raw1 = *reinterpret_cast<const float4 *>(lds + (a_addr[1] - lds0));
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) X.b[j][pl] = *reinterpret_cast<const bf16x8 *>(lds + (b_addr[j] - lds0) + pl * B_PLANE);
            unsigned hh[4], mm[4], ll[4];
            if constexpr (NPL == 2) {
                split2_pair(raw0.x, raw0.y, a_scale_s, hh[0], ll[0]);
                split2_pair(raw0.z, raw0.w, a_scale_s, hh[1], ll[1]);
                split2_pair(raw1.x, raw1.y, a_scale_s, hh[2], ll[2]);
                split2_pair(raw1.z, raw1.w, a_scale_s, hh[3], ll[3]);
                mm[0] = mm[1] = mm[2] = mm[3] = 0;
            } else {
                split3_pair(raw0.x, raw0.y, hh[0], mm[0], ll[0]);
                split3_pair(raw0.z, raw0.w, hh[1], mm[1], ll[1]);
                split3_pair(raw1.x, raw1.y, hh[2], mm[2], ll[2]);
                split3_pair(raw1.z, raw1.w, hh[3], mm[3], ll[3]);
            }
            X.a[0] = __builtin_bit_cast(bf16x8, (u32x4{hh[0], hh[1], hh[2], hh[3]}));
            X.a[1] = __builtin_bit_cast(bf16x8, (u32x4{mm[0], mm[1], mm[2], mm[3]}));
            X.a[NPL - 1] = __builtin_bit_cast(bf16x8, (u32x4{ll[0], ll[1], ll[2], ll[3]}));
            // the compiler must finish its own LDS reads HERE: with one pending at the loop head it puts a conservative
            // lgkmcnt(0) in front of the first MFMA of every iteration
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) asm volatile("" : "+v"(X.b[j][pl]));
        }
        // phase i: MFMAs of stage i, LDS reads of stage i+1 (slot (i+1) % S), DMA of stage i+S into slot i % S
        int slot = 0;
        auto next = [&](int s) { return s + 1 == S ? 0 : s + 1; };
        int i = 0;
        for (; i + 1 < nk; i += 2) {
            turn();
            const int s1 = next(slot), s2 = next(s1);
            phase(X, Y, (unsigned)(s1 * STAGE), slot);
            turn();
            phase(Y, X, (unsigned)(s2 * STAGE), s1);
            slot = s2;
        }
        if (i < nk) {                                            // odd count: one more phase (its reads fetch a null stage)
            turn();
            phase(X, Y, (unsigned)(next(slot) * STAGE), slot);
        }
        wait_vm<0>();
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // asm-issued MFMA results -> ordinary reads
    }
#undef TSOD_DMA
    // The epilogue reads its parameters from the kernel-argument segment (the struct is the only argument) instead of from
    // `p`: values that only the epilogue needs then do not occupy scalar registers across the K loop (under the balanced
    // schedule's segment loop they pushed the loop's own scalars out into v_readlane / v_writelane).
    const ConvParams &pe = *(const ConvParams *)__builtin_amdgcn_kernarg_segment_ptr();
    if constexpr (NPL == 2) {                                    // back from (a_scale x) . (weight scale w) to x . w (a power of two: exact)
        if (amax_pending) wait_vm<0>();                           // (a workgroup without K-steps never waited for its range-word DMAs)
        // 1 / (a_scale * 2^w_scale_exp), from the scale that was really used (static or from the range words)
        const float acc_scale = __uint_as_float((unsigned)(254 - (int)(__float_as_uint(a_scale_s) >> 23) - pe.w_scale_exp) << 23);
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[0][j][e] *= acc_scale;
        // the range guard: an activation beyond 65504 / a_scale splits into +-inf pieces whose products make EVERY accumulator
        // it feeds inf or NaN (a zero weight gives 0 * inf = NaN too), so a finite tile proves its inputs were in range; a
        // non-finite one (also from genuinely non-finite input) raises the caller's flag - the epilogue's branch-free
        // activation would otherwise turn the NaNs into plausible zeros
        if (p.range_flag != nullptr) {
            bool bad = false;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) bad |= !(fabsf(acc[0][j][e]) <= 3.4028234664e38f);
            if (__any(bad) && lane == 0) atomicOr(p.range_flag, 1);
        }
    }
#ifdef TSOD_DIAG_MFMA16
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][j][e] = pacc[j].lo[e & 3] + pacc[j].hi[e & 3];
#endif
    __syncthreads();                                             // no wave reads the ring any more: the epilogue may use it
#ifdef TSOD_CLOCK_DIAG
    if (dg_on) { const long long c = __builtin_amdgcn_s_memtime(); dg_loop += c - dg_c; dg_c = c; }
#endif
    float *smem = reinterpret_cast<float *>(lds);
    if constexpr (WAVES_K == 2) {
        // the two K halves of a 32 x 128 stripe sit in waves (wm, 0) and (wm, 1): each keeps the 64 columns [64 wk, 64 wk + 64),
        // hands the other 64 over through LDS ([wave][block][e][lane], lane-contiguous) and adds what its partner handed over
        f32x16 acc2[1][2];
        const int partner = wm + WAVES_M * (1 - wk);
        if (wk == 0) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) smem[((wave * 2 + b) * 16 + e) * 64 + lane] = acc[0][2 + b][e];
        } else {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) smem[((wave * 2 + b) * 16 + e) * 64 + lane] = acc[0][b][e];
        }
        __syncthreads();
        if (wk == 0) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc2[0][b][e] = acc[0][b][e] + smem[((partner * 2 + b) * 16 + e) * 64 + lane];
        } else {
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc2[0][b][e] = smem[((partner * 2 + b) * 16 + e) * 64 + lane] + acc[0][2 + b][e];
        }
        __syncthreads();
        conv_epilogue<BM, BN, 32, 64, THREADS, BALANCED ? 3 : 4>(pe, acc2, smem, tid, wm, wk, m0, n0, sm);
    } else if constexpr (WAVES_K == 4) {
        // the four K quarters of a 32 x 128 stripe sit in waves (wm, 0 .. 3): every wave puts its four 32 x 32 blocks into LDS
        // ([wave][block][e][lane], lane-contiguous) and wave (wm, wk) finishes block wk = columns [32 wk, 32 wk + 32) as the sum of the
        // four quarters IN K ORDER (quarter 0 first, whoever adds: bit-reproducible).  The own quarter goes through LDS too - picking it
        // from the registers would index the accumulators by a run-time wave number.
        f32x16 acc1[1][1];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) smem[((wave * 4 + b) * 16 + e) * 64 + lane] = acc[0][b][e];
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float v = smem[(((wm + WAVES_M * 0) * 4 + wk) * 16 + e) * 64 + lane];
            v += smem[(((wm + WAVES_M * 1) * 4 + wk) * 16 + e) * 64 + lane];
            v += smem[(((wm + WAVES_M * 2) * 4 + wk) * 16 + e) * 64 + lane];
            v += smem[(((wm + WAVES_M * 3) * 4 + wk) * 16 + e) * 64 + lane];
            acc1[0][0][e] = v;
        }
        __syncthreads();
        conv_epilogue<BM, BN, 32, 32, THREADS, BALANCED ? 3 : 4>(pe, acc1, smem, tid, wm, wk, m0, n0, sm);
    } else {
        conv_epilogue<BM, BN, 32, 128, THREADS, BALANCED ? 3 : 4>(pe, acc, smem, tid, wm, wn, m0, n0, sm);
    }
#ifdef TSOD_CLOCK_DIAG
    if (dg_on) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (diag only: the epilogue's stores have left)
        dg_epi += __builtin_amdgcn_s_memtime() - dg_c;
    }
#endif
  }
#ifdef TSOD_CLOCK_DIAG
    if (dg_on) {
        long long *o = g_dma_stamps + 8 * (long)blockIdx.x;
        o[0] = dg_rt0; o[1] = __builtin_amdgcn_s_memrealtime(); o[2] = dg_pro; o[3] = dg_loop; o[4] = dg_epi; o[5] = dg_steps; o[6] = dg_segs;
        o[7] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);   // HW_REG_XCC_ID (id 20), bits [3:0]
    }
#endif
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

// f32 packed weights [Cout][K] -> bf16x3 [Cout][ceil(K/8)][hi|mid|lo][8]: each weight cut EXACTLY into three bf16 pieces by
// truncation (hi + mid + lo == w); k beyond K is zero.  One thread per 8-k group.
__global__ void __launch_bounds__(256)
pack_weight_bf16x3_kernel(const float *__restrict__ w, int Cout, int K, unsigned *__restrict__ out) {
    const int groups = (K + 7) / 8;
    const long total = (long)Cout * groups;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int g = (int)(t % groups);
        const long n = t / groups;
        unsigned h[8], m[8], l[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = g * 8 + e;
            split3(k < K ? w[n * K + k] : 0.f, h[e], m[e], l[e]);
        }
        unsigned *o = out + t * 12;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            o[e] = pack2(h[2 * e], h[2 * e + 1]);
            o[4 + e] = pack2(m[2 * e], m[2 * e + 1]);
            o[8 + e] = pack2(l[2 * e], l[2 * e + 1]);
        }
    }
}

// f32 packed weights [Cout][K] -> fp16x2 [Cout][ceil(K/8)][hi|lo][8] fp16 of scale * w (scale a power of two): hi = rne(s w),
// lo = rne(s w - hi); k beyond K is zero.  One thread per 8-k group.
__global__ void __launch_bounds__(256)
pack_weight_fp16x2_kernel(const float *__restrict__ w, int Cout, int K, float scale, unsigned *__restrict__ out) {
    const int groups = (K + 7) / 8;
    const long total = (long)Cout * groups;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int g = (int)(t % groups);
        const long n = t / groups;
        unsigned *o = out + t * 8;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = g * 8 + 2 * e;
            unsigned h, l;
            split2_pair(k < K ? w[n * K + k] : 0.f, k + 1 < K ? w[n * K + k + 1] : 0.f, scale, h, l);
            o[e] = h;
            o[4 + e] = l;
        }
    }
}

// resident = workgroups per CU (LDS / VGPR bound); dma = 1: conv_dma_kernel (bf16x3 only, nbuf = ring stages)
struct TileInfo { int bm, bn, threads, resident; float cost; int bk, nbuf, bf16x3, dma; };
const TileInfo kTiles[TSOD_TILE_COUNT] = {
    {0, 0, 0, 0, 0.f, 32, 2, 0},        {128, 128, 256, 2, 1.00f, 32, 2, 0}, {128, 64, 256, 2, 1.06f, 32, 2, 0}, {64, 64, 256, 4, 1.15f, 32, 2, 1},
    {64, 128, 256, 2, 1.06f, 32, 2, 0}, {128, 128, 512, 2, 1.00f, 32, 2, 0}, {128, 64, 512, 2, 1.06f, 32, 2, 0}, {256, 128, 512, 1, 0.98f, 32, 2, 0},
    {64, 64, 256, 6, 1.20f, 32, 1, 1},  {128, 64, 512, 3, 1.10f, 32, 1, 1},  {64, 64, 256, 4, 1.10f, 64, 1, 1},  {128, 64, 512, 2, 1.05f, 64, 1, 0},
    {64, 64, 64, 8, 1.40f, 32, 1, 0},   {128, 64, 128, 4, 1.35f, 32, 1, 0}, {128, 64, 256, 4, 1.12f, 32, 1, 1}, {64, 128, 256, 4, 1.12f, 32, 1, 1},
    {128, 128, 256, 2, 1.02f, 32, 1, 1},
    {128, 128, 256, 2, 0.80f, 16, 4, 1, 1}, {64, 128, 256, 1, 0.95f, 32, 3, 1, 1}, {256, 128, 512, 1, 0.72f, 16, 4, 1, 1},
    {64, 128, 256, 2, 0.98f, 32, 2, 1, 1}, {128, 256, 512, 1, 0.74f, 16, 4, 1, 1}, {128, 128, 512, 1, 0.70f, 32, 3, 1, 1},
    {192, 128, 384, 1, 0.74f, 16, 4, 1, 1}, {64, 128, 512, 1, 0.80f, 64, 3, 1, 1}};
// bf16x3 = 1: the tile also exists as a bf16x3 variant (three bf16 planes per operand fit the 64 KB of static LDS)

// workgroups per CU: the f32 figure (VGPR / LDS bound), for bf16x3 additionally capped by its larger LDS footprint
// tiles that exist in the fp16x2 arithmetic: the register-staged bf16x3 tiles and d128x128k32
bool fp16x2_tile(int t) {   // (the 64-row LDS-DMA tiles would need six DMA slots in a 12-MFMA phase: not built)
    return t == TSOD_TILE_D128x128_K32 || t == TSOD_TILE_D128x128 || t == TSOD_TILE_D256x128 || t == TSOD_TILE_D128x256 || t == TSOD_TILE_D192x128 ||
           t == TSOD_TILE_D64x128_K64 || (kTiles[t].bf16x3 && !kTiles[t].dma);
}

int residency(int tile, int prec) {
    const TileInfo &t = kTiles[tile];
    if (!prec) return t.resident;
    const int lds = t.dma ? t.nbuf * dma_stage_bytes(t.bm, t.bk, t.bn) : t.nbuf * 3 * (t.bm + t.bn) * (t.bk / 2) * 4;
    const int fit = 160 * 1024 / lds;
    return fit < t.resident ? (fit < 1 ? 1 : fit) : t.resident;
}

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
    TSOD_REQUIRE(d->tile >= 0 && d->tile < TSOD_TILE_COUNT && d->split_k >= -2 && d->split_k <= 64, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(d->precision == TSOD_PREC_F32 || d->precision == TSOD_PREC_BF16X3 || d->precision == TSOD_PREC_FP16X2, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(d->precision == TSOD_PREC_F32 || d->tile == TSOD_TILE_AUTO || kTiles[d->tile].bf16x3, TSOD_ERR_UNSUPPORTED);
    TSOD_REQUIRE((d->tile != TSOD_TILE_D192x128 && d->tile != TSOD_TILE_D64x128_K64) || d->precision == TSOD_PREC_FP16X2, TSOD_ERR_UNSUPPORTED);   // (fp16x2 only)
    TSOD_REQUIRE(d->precision != TSOD_PREC_F32 || d->tile == TSOD_TILE_AUTO || !kTiles[d->tile].dma, TSOD_ERR_UNSUPPORTED);
    if (d->precision == TSOD_PREC_FP16X2) {                       // (the register-staged bf16x3 tiles and the 128x128 / 32-k LDS-DMA tile)
        TSOD_REQUIRE(d->tile == TSOD_TILE_AUTO || fp16x2_tile(d->tile), TSOD_ERR_UNSUPPORTED);
        TSOD_REQUIRE(d->a_scale_exp >= -24 && d->a_scale_exp <= 24 && d->w_scale_exp >= -40 && d->w_scale_exp <= 40, TSOD_ERR_INVALID_ARG);
    }
    const int64_t M = (int64_t)d->N * d->OH * d->OW;
    TSOD_REQUIRE(M < (int64_t)INT_MAX, TSOD_ERR_UNSUPPORTED);
    if (d->c2 != 0) {                                   // second source: a strided 1x1 tap of another tensor
        TSOD_REQUIRE(d->c2 > 0 && (d->c2 & 3) == 0 && d->stride2 > 0 && d->H2 > 0 && d->W2 > 0, TSOD_ERR_INVALID_ARG);
        TSOD_REQUIRE((d->in2_pitch & 3) == 0 && (d->in2_off & 3) == 0, TSOD_ERR_ALIGNMENT);
        TSOD_REQUIRE(d->in2_off >= 0 && d->in2_off + d->c2 <= d->in2_pitch, TSOD_ERR_INVALID_ARG);
        TSOD_REQUIRE((d->OH - 1) * d->stride2 < d->H2 && (d->OW - 1) * d->stride2 < d->W2, TSOD_ERR_INVALID_ARG);
        // K-steps must not straddle the two sources, and the fast (wave-uniform tap) loader is the one that knows about them
        TSOD_REQUIRE(d->n_seg == 1 && (d->KH * d->KW * d->seg_len[0]) % 32 == 0 && d->seg_len[0] % 32 == 0, TSOD_ERR_UNSUPPORTED);
    }
    return TSOD_OK;
}

int desc_cin(const tsod_conv2d_desc *d) {
    int c = 0;
    for (int s = 0; s < d->n_seg; ++s) c += d->seg_len[s];
    return c;
}

// contraction length: the filter taps over the first source + the second source's channels
int desc_k(const tsod_conv2d_desc *d) { return d->KH * d->KW * desc_cin(d) + (d->c2 > 0 ? d->c2 : 0); }

// a tile's K-step must divide the channel count (uniform taps), K1 AND the second source's channel count when there is a
// second source (the uniform-tap loader has no k < K mask: a K-step that ran past in2's c2 channels would read the next
// pixel's channels and the next weight row); the LDS-DMA tiles need that always (a stage lies inside one filter tap of one
// channel segment, K is whole stages)
// conv_dma_kernel<..., CHAN = true>: a 1x1 filter (stride 1, no padding, no second source) over several channel segments, fp16x2
bool dma_chan_case(const tsod_conv2d_desc *d) {
    return d->precision == TSOD_PREC_FP16X2 && d->n_seg > 1 && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad_h == 0 && d->pad_w == 0 &&
           d->c2 <= 0;
}

bool tile_ok_for(const tsod_conv2d_desc *d, int tile) {
    const int bk = kTiles[tile].bk;
    if (kTiles[tile].dma) {
        if (dma_chan_case(d))                    // 1x1 over several segments: the fp16x2 instantiations carry a chunk table (K <= 2048)
            return desc_k(d) <= 2048 && desc_k(d) / bk + 9 <= kDmaTabEntries;
        return d->n_seg == 1 && desc_cin(d) % bk == 0 && (d->c2 <= 0 || d->c2 % bk == 0) &&
               d->KH * d->KW <= 31 && desc_k(d) / bk + 8 <= kDmaTabEntries;   // (the stage table: one mask bit per tap, K range + ring)
    }
    if (d->c2 <= 0) return true;
    return desc_cin(d) % bk == 0 && (d->KH * d->KW * desc_cin(d)) % bk == 0 && d->c2 % bk == 0;
}

// TSOD_XCD_NMAJOR = 0 / 1 pins the XCD run order of the uniform schedules for experiments (unset: the byte estimate decides)
int xcd_map_override() {
    static const int v = [] {
        const char *e = getenv("TSOD_XCD_NMAJOR");
        return e == nullptr || *e == 0 ? -1 : (atoi(e) != 0 ? 1 : 0);
    }();
    return v;
}

// TSOD_KSTEP_TAPMAJOR = 1 keeps the LDS-DMA tiles on the weight image's own K-step order, (tap, channels), for experiments
// (unset: channel blocks outermost wherever a filter has more than one tap, ConvParams::cmajor)
int kstep_order_override() {
    static const int v = [] {
        const char *e = getenv("TSOD_KSTEP_TAPMAJOR");
        return e != nullptr && atoi(e) != 0 ? 0 : 1;
    }();
    return v;
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

constexpr size_t kTicketBytes = 256 * 1024;   // tickets for up to 65536 K-sliced tiles per launch

struct Sched {
    int tile, bm, bn, tiles_m, tiles_n, tiles, dp_tiles, rem_tiles, split, ksteps_per_split, grid;
    int sk_q;                     // balanced schedule: K-steps per workgroup (0: a uniform schedule)
    int nmajor;                   // XCD runs share an output-channel tile instead of activation rows (work_item)
    size_t ws_bytes, ticket_bytes;
    double cost;
};

// desc.split_k:  1 = every tile whole (no K split);  S > 1 = every tile cut into S K-slices;
//               -1 = hybrid: as many full chip-waves of whole tiles as fit, the left-over tiles cut into
//                    K-slices so that the last wave also fills the chip;  0 = let the cost model choose;
//               -2 = balanced (LDS-DMA tiles only): one workgroup per CU slot, each the same number of K-steps of the
//                    tile-major K-step sequence (a tile held by several workgroups is combined like K-slices).
Sched make_sched(const tsod_conv2d_desc *d, int tile, int mode) {
    Sched s;
    const int64_t M = (int64_t)d->N * d->OH * d->OW;
    const int K = desc_k(d);
    const int bk = kTiles[tile].bk;
    const int ksteps = (K + bk - 1) / bk;
    s.tile = tile; s.bm = kTiles[tile].bm; s.bn = kTiles[tile].bn;
    s.tiles_m = (int)tsod_cdiv(M, s.bm); s.tiles_n = (int)tsod_cdiv(d->Cout, s.bn);
    s.tiles = s.tiles_m * s.tiles_n;
    const int slots = cu_count() * residency(tile, d->precision);
    s.sk_q = 0;
    s.nmajor = 0;
    if (mode == -2) {
        const int64_t total = (int64_t)s.tiles * ksteps;
        int64_t q = tsod_cdiv(total, slots);
        if (q < 4) q = 4;                                         // a workgroup's prologue / epilogue need some K loop to pay for
        const int64_t grid = tsod_cdiv(total, q);
        // (the multi-segment 1x1 form of the LDS-DMA kernel has no balanced instantiation)
        if (kTiles[tile].dma && !dma_chan_case(d) && q < INT_MAX && (size_t)s.tiles * sizeof(int) <= kTicketBytes) {
            s.sk_q = (int)q; s.grid = (int)grid;
            s.split = 1; s.ksteps_per_split = ksteps; s.dp_tiles = 0; s.rem_tiles = s.tiles;
            s.ticket_bytes = kTicketBytes;
            s.ws_bytes = s.ticket_bytes + (size_t)grid * 2 * s.bm * s.bn * sizeof(float);
            const double step = (double)s.bm * s.bn / 4.0 * kTiles[tile].cost * (bk / 32.0) * 0.55;
            const double fixed = 3000.0 + (double)s.bm * s.bn / 8.0;
            s.cost = (double)q * step + fixed * (1.0 + (double)q / ksteps) + 2500.0 + 2.0 * s.bm * s.bn * 4.0 / 60.0;
            return s;
        }
        s.cost = 1e300;                                           // not a tile that knows this schedule
        s.split = 1; s.ksteps_per_split = ksteps; s.dp_tiles = s.tiles; s.rem_tiles = 0; s.grid = s.tiles; s.ws_bytes = 0; s.ticket_bytes = 0;
        return s;
    }
    int split = 1, dp = s.tiles;
    if (mode > 1) {
        split = mode < ksteps ? mode : ksteps;
        dp = 0;
    } else if (mode == -1) {
        const int full = s.tiles / slots * slots;
        const int rem = s.tiles - full;
        if (rem > 0) {
            split = slots / rem;
            if (split > ksteps / 2) split = ksteps / 2;      // at least 2 K-steps per slice
            if (split < 1) split = 1;
            if (split > 1) dp = full;
        }
    }
    int kps = (ksteps + split - 1) / split;
    split = (ksteps + kps - 1) / kps;                         // no empty slices
    if (split <= 1 || (size_t)(s.tiles - dp) * sizeof(int) > kTicketBytes) {   // (more K-sliced tiles than tickets: whole tiles)
        split = 1; dp = s.tiles; kps = ksteps;
    }
    s.split = split; s.ksteps_per_split = kps; s.dp_tiles = dp; s.rem_tiles = s.tiles - dp;
    s.grid = dp + s.rem_tiles * split;
    // XCD locality: what the 8 private L2s fetch from beyond them.  Runs that share activation rows: every XCD streams ALL the
    // weights (8 W) and its own rows (A).  Runs that share an output-channel tile: an XCD streams the rows of all row tiles
    // (g A, g = min(tiles_n, 8) groups) and 1/g of the weights (8 W / g).  The second form wins where the weights outweigh the
    // activations: the small-M layers of a batch-1 forward.  Pure schedules only (work_item).
    if ((dp == s.tiles || dp == 0) && s.tiles_n >= 2 && s.grid >= 16) {
        const double wbytes = (double)d->Cout * K * (d->precision ? 6.0 : 4.0);
        const double abytes = (double)d->N * d->H * d->W * desc_cin(d) * 4.0 + (d->c2 > 0 ? (double)M * d->c2 * 4.0 : 0.0);
        const double g = s.tiles_n < 8 ? s.tiles_n : 8;
        const int force = xcd_map_override();
        s.nmajor = force >= 0 ? force : (8.0 * wbytes * (1.0 - 1.0 / g) > abytes * (g - 1.0) * 1.25 ? 1 : 0);
    }
    // workspace = [kTicketBytes of arrival tickets, one int per K-sliced tile][rem_tiles * split slabs of BM x BN floats].
    // The ticket area has ONE size for every launch: launches that share a workspace must never see another launch's
    // slab bytes where they expect zeroed tickets.
    s.ticket_bytes = s.rem_tiles > 0 ? kTicketBytes : 0;
    s.ws_bytes = s.ticket_bytes + (size_t)s.rem_tiles * split * s.bm * s.bn * sizeof(float);

    // cost, in cycles of the most loaded CU: co-resident workgroups share the CU's matrix pipes, so a wave of
    // workgroups costs (workgroups per CU) x (K-steps x BM*BN/4 MFMA cycles + fixed prologue/epilogue)
    const double step = (double)s.bm * s.bn / 4.0 * kTiles[tile].cost * (bk / 32.0) * (d->precision ? 0.55 : 1.0);
    const double fixed = 3000.0 + (double)s.bm * s.bn / 8.0;
    const int cus = cu_count();
    double c = (double)tsod_cdiv(dp, cus) * (ksteps * step + fixed);
    if (s.rem_tiles > 0)
        c += (double)tsod_cdiv((int64_t)s.rem_tiles * split, cus) * (kps * step + fixed) + 2500.0 +
             (double)split * s.bm * s.bn * 4.0 / 60.0;     // last arriver: serial read of `split` slabs at ~60 B/cycle
    s.cost = c;
    return s;
}

Sched resolve(const tsod_conv2d_desc *d) {
    Sched best;
    best.cost = 1e300;
    for (int t = 1; t < TSOD_TILE_COUNT; ++t) {
        if (d->tile != TSOD_TILE_AUTO && d->tile != t) continue;
        if (d->precision && !kTiles[t].bf16x3) continue;
        if (!d->precision && kTiles[t].dma) continue;
        if ((t == TSOD_TILE_D192x128 || t == TSOD_TILE_D64x128_K64) && d->precision != TSOD_PREC_FP16X2) continue;
        if (d->precision == TSOD_PREC_FP16X2 && !fp16x2_tile(t)) continue;
        if (!tile_ok_for(d, t)) continue;        // (also an explicitly named tile: the caller gets TSOD_ERR_UNSUPPORTED)
        if (d->split_k != 0) {
            const Sched s = make_sched(d, t, d->split_k);
            if (s.cost < best.cost) best = s;
            continue;
        }
        const int K = desc_k(d);
        const int ksteps = (K + kTiles[t].bk - 1) / kTiles[t].bk;
        for (int mode : {1, -1, 2, 4, 8, 16, -2}) {
            if (mode > 1 && ksteps / mode < 2) continue;
            if (mode == -2 && !kTiles[t].dma) continue;
            const Sched s = make_sched(d, t, mode);
            if (s.cost < best.cost) best = s;
        }
    }
    return best;
}

template <int BM, int BK, int WAVES_K, int S, int WAVES_N = 1, int NPL = 3>
void launch_dma_tile(const ConvParams &p, int grid, hipStream_t s) {
    if constexpr (NPL == 2) {
        if (p.chan_tab) {                                         // 1x1 over several channel segments (never balanced: make_sched)
            hipLaunchKernelGGL((conv_dma_kernel<BM, BK, WAVES_K, S, false, WAVES_N, NPL, true>), dim3(grid), dim3((BM / 32) * WAVES_K * WAVES_N * 64), 0, s, p);
            return;
        }
    }
    if (p.sk_q > 0)
        hipLaunchKernelGGL((conv_dma_kernel<BM, BK, WAVES_K, S, true, WAVES_N, NPL>), dim3(grid), dim3((BM / 32) * WAVES_K * WAVES_N * 64), 0, s, p);
    else
        hipLaunchKernelGGL((conv_dma_kernel<BM, BK, WAVES_K, S, false, WAVES_N, NPL>), dim3(grid), dim3((BM / 32) * WAVES_K * WAVES_N * 64), 0, s, p);
}

template <int BM, int BN, int WM, int WN, int MW, int NBUF = 2, int BK = 32, int PREC = 0>
void launch_tile(const ConvParams &p, int grid, hipStream_t s) {
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, MW, NBUF, BK, PREC>), dim3(grid), dim3(64 * (BM / WM) * (BN / WN)), 0, s, p);
}

}  // namespace

extern "C" int tsod_conv2d_resolve(const tsod_conv2d_desc *d, int32_t *tile, int32_t *split_k) {
    const int rc = validate(d);
    if (rc != TSOD_OK) return rc;
    TSOD_REQUIRE(tile && split_k, TSOD_ERR_INVALID_ARG);
    const Sched s = resolve(d);
    TSOD_REQUIRE(s.cost < 1e299, TSOD_ERR_UNSUPPORTED);
    *tile = s.tile;
    *split_k = s.sk_q > 0 ? -2 : (s.rem_tiles == 0 ? 1 : (s.dp_tiles == 0 ? s.split : -1));
    return TSOD_OK;
}

extern "C" size_t tsod_conv2d_workspace_bytes(const tsod_conv2d_desc *d) {
    if (validate(d) != TSOD_OK) return 0;
    const Sched s = resolve(d);
    return s.cost < 1e299 ? s.ws_bytes : 0;
}

extern "C" int tsod_conv2d_f32(const tsod_conv2d_desc *d, const float *in, const float *w_packed, const float *scale,
                               const float *shift, const float *residual, float *out, void *workspace,
                               size_t workspace_bytes, tsod_stream_t stream) {
    if (d != nullptr && d->c2 != 0) return TSOD_ERR_INVALID_ARG;       // a second source needs tsod_conv2d_dual_f32
    return tsod_conv2d_dual_f32(d, in, nullptr, w_packed, scale, shift, residual, out, workspace, workspace_bytes, stream);
}

extern "C" int tsod_conv2d_dual_f32(const tsod_conv2d_desc *d, const float *in, const float *in2, const float *w_packed,
                                    const float *scale, const float *shift, const float *residual, float *out,
                                    void *workspace, size_t workspace_bytes, tsod_stream_t stream) {
    const int rc = validate(d);
    if (rc != TSOD_OK) return rc;
    TSOD_REQUIRE(in && w_packed && out, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE((d->c2 > 0) == (in2 != nullptr) && (!in2 || tsod_aligned16(in2)), TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(tsod_aligned16(in) && tsod_aligned16(w_packed), TSOD_ERR_ALIGNMENT);
    if (residual) TSOD_REQUIRE(d->res_off >= 0 && d->res_pitch >= d->res_off + d->Cout, TSOD_ERR_INVALID_ARG);

    ConvParams p;
    p.in = in; p.w = w_packed; p.scale = scale; p.shift = shift; p.res = residual; p.out = out;
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
    p.K1 = d->KH * d->KW * p.Cin;
    p.K = desc_k(d);
    p.in2 = in2; p.c2 = d->c2 > 0 ? d->c2 : 0; p.in2_pitch = d->in2_pitch; p.in2_off = d->in2_off;
    p.stride2 = d->stride2; p.H2 = d->H2; p.W2 = d->W2;
    {
        const uint64_t in2_bytes = in2 ? (uint64_t)d->N * d->H2 * d->W2 * d->in2_pitch * sizeof(float) : 0;
        TSOD_REQUIRE(in2_bytes < 0xFFFFFFF0ull, TSOD_ERR_UNSUPPORTED);
        p.in2_bytes = (unsigned)in2_bytes;
        const uint64_t in_bytes = (uint64_t)d->N * d->H * d->W * d->in_pitch * sizeof(float);
        // bf16x3 weights are pre-split: [Cout][ceil(K/8)][hi|mid|lo][8] bf16 = 48 bytes per 8 k (tsod_pack_conv_weight_bf16x3)
        const uint64_t w_bytes = d->precision == TSOD_PREC_BF16X3 ? (uint64_t)d->Cout * ((p.K + 7) / 8) * 48
                                 : d->precision == TSOD_PREC_FP16X2 ? (uint64_t)d->Cout * ((p.K + 7) / 8) * 32
                                                                    : (uint64_t)d->Cout * p.K * sizeof(float);
        const uint64_t out_bytes = (uint64_t)p.M * d->out_pitch * sizeof(float);
        const uint64_t res_bytes = residual ? (uint64_t)p.M * d->res_pitch * sizeof(float) : 0;
        // 32-bit buffer offsets: one activation tensor must stay below 4 GiB (shard the batch otherwise)
        TSOD_REQUIRE(in_bytes < 0xFFFFFFF0ull && w_bytes < 0xFFFFFFF0ull && out_bytes < 0xFFFFFFF0ull &&
                         res_bytes < 0xFFFFFFF0ull && p.K < (1 << 21),
                     TSOD_ERR_UNSUPPORTED);
        p.in_bytes = (unsigned)in_bytes;
        p.w_bytes = (unsigned)w_bytes;
        p.out_bytes = (unsigned)out_bytes;
        p.res_bytes = (unsigned)res_bytes;
        p.neg_slope = d->act == TSOD_ACT_NONE ? 1.f : (d->act == TSOD_ACT_PRELU ? d->slope : 0.f);
        p.act_hi = d->act == TSOD_ACT_RELU6 ? 6.f : __builtin_huge_valf();
        p.vec_epilogue = ((d->Cout | d->out_pitch | d->out_off) & 3) == 0 && tsod_aligned16(out) &&
                         (!residual || (((d->res_pitch | d->res_off) & 3) == 0 && tsod_aligned16(residual))) &&
                         (!scale || tsod_aligned16(scale)) && (!shift || tsod_aligned16(shift));
        p.inv_cin = 1.0f / (float)p.Cin;
        p.inv_kw = 1.0f / (float)d->KW;
    }
    const Sched sc = resolve(d);
    TSOD_REQUIRE(sc.cost < 1e299, TSOD_ERR_UNSUPPORTED);        // the named tile cannot run this problem (LDS-DMA tiles: tile_ok_for)
    p.uniform_tap = (d->n_seg == 1 && p.Cin % kTiles[sc.tile].bk == 0) ? 1 : 0;
    static const bool no_chan_tab = getenv("TSOD_NO_CHAN_TAB") != nullptr;     // diagnostic: the arithmetic loader for every layer
    p.chan_tab = (kTiles[sc.tile].dma && dma_chan_case(d)) ? 1 : 0;
    p.pointwise_tab = (!no_chan_tab && !p.uniform_tap && !kTiles[sc.tile].dma && d->KH == 1 && d->KW == 1 && d->pad_h == 0 && d->pad_w == 0 && d->stride == 1 &&
                       p.c2 == 0 && p.K <= 4 * kChanTab) ? 1 : 0;
    TSOD_REQUIRE(p.c2 == 0 || (p.uniform_tap && p.K1 % kTiles[sc.tile].bk == 0 && p.c2 % kTiles[sc.tile].bk == 0), TSOD_ERR_UNSUPPORTED);
    p.ksteps = (p.K + kTiles[sc.tile].bk - 1) / kTiles[sc.tile].bk;
    p.tiles_m = sc.tiles_m; p.tiles_n = sc.tiles_n;
    p.dp_tiles = sc.dp_tiles; p.split = sc.split; p.ksteps_per_split = sc.ksteps_per_split; p.sk_q = sc.sk_q;
    p.nmajor = sc.nmajor;
    const int cblk = kTiles[sc.tile].bk > 32 ? kTiles[sc.tile].bk : 32;      // channel block of the block-major K-step order (conv_dma_kernel: CBLK)
    p.cmajor = (kTiles[sc.tile].dma && d->KH * d->KW > 1 && p.Cin % cblk == 0 && kstep_order_override() != 0) ? 1 : 0;
    p.ci_wrap = p.cmajor ? cblk : p.Cin;
    if (sc.rem_tiles > 0)
        TSOD_REQUIRE(workspace != nullptr && workspace_bytes >= sc.ws_bytes && tsod_aligned16(workspace), TSOD_ERR_WORKSPACE);
    p.tickets = static_cast<int *>(workspace);
    p.partial = reinterpret_cast<float *>(static_cast<char *>(workspace) + sc.ticket_bytes);
    p.part_bytes = (unsigned)(sc.ws_bytes - sc.ticket_bytes);
    TSOD_REQUIRE(sc.ws_bytes < 0xFFFFFFF0ull, TSOD_ERR_UNSUPPORTED);
    hipStream_t s = tsod_stream(stream);
    p.a_scale = 1.f; p.acc_scale = 1.f; p.range_flag = nullptr;
    p.amax_in = nullptr; p.amax_in2 = nullptr; p.w_scale_exp = 0;
    p.amax_out = d->amax_out;
    TSOD_REQUIRE((reinterpret_cast<uintptr_t>(d->amax_out) & 63u) == 0 && (reinterpret_cast<uintptr_t>(d->amax_in) & 63u) == 0 &&
                     (reinterpret_cast<uintptr_t>(d->amax_in2) & 63u) == 0, TSOD_ERR_ALIGNMENT);
    if (d->precision == TSOD_PREC_FP16X2) {
        TSOD_REQUIRE(fp16x2_tile(sc.tile), TSOD_ERR_UNSUPPORTED);
        p.range_flag = d->range_flag;
        p.amax_in = d->amax_in;                                   // (NULL: the static a_scale_exp)
        p.amax_in2 = (d->amax_in != nullptr && p.c2 > 0) ? d->amax_in2 : nullptr;
        p.w_scale_exp = d->w_scale_exp;
        p.a_scale = ldexpf(1.f, d->a_scale_exp);
        p.acc_scale = ldexpf(1.f, -(d->a_scale_exp + d->w_scale_exp));
        switch (sc.tile) {
            case TSOD_TILE_64x64_S1: launch_tile<64, 64, 32, 32, 5, 1, 32, 2>(p, sc.grid, s); break;
            case TSOD_TILE_128x64_W8_S1: launch_tile<128, 64, 32, 32, 3, 1, 32, 2>(p, sc.grid, s); break;
            case TSOD_TILE_64x64_S1_K64: launch_tile<64, 64, 32, 32, 2, 1, 64, 2>(p, sc.grid, s); break;
            case TSOD_TILE_128x64_S1: launch_tile<128, 64, 64, 32, 4, 1, 32, 2>(p, sc.grid, s); break;
            case TSOD_TILE_64x128_S1: launch_tile<64, 128, 32, 64, 4, 1, 32, 2>(p, sc.grid, s); break;
            case TSOD_TILE_128x128_S1: launch_tile<128, 128, 64, 64, 2, 1, 32, 2>(p, sc.grid, s); break;
            case TSOD_TILE_D128x128_K32: launch_dma_tile<128, 32, 2, 3, 1, 2>(p, sc.grid, s); break;
            case TSOD_TILE_D128x128: launch_dma_tile<128, 16, 1, 4, 1, 2>(p, sc.grid, s); break;
            case TSOD_TILE_D256x128: launch_dma_tile<256, 16, 1, 4, 1, 2>(p, sc.grid, s); break;
            case TSOD_TILE_D128x256: launch_dma_tile<128, 16, 1, 4, 2, 2>(p, sc.grid, s); break;
            case TSOD_TILE_D192x128: launch_dma_tile<192, 16, 1, 4, 1, 2>(p, sc.grid, s); break;
            case TSOD_TILE_D64x128_K64: launch_dma_tile<64, 64, 4, 3, 1, 2>(p, sc.grid, s); break;
            default: launch_tile<64, 64, 32, 32, 2, 2, 32, 2>(p, sc.grid, s); break;      // TSOD_TILE_64x64 (two LDS stages)
        }
        return tsod_launch_status();
    }
    if (d->precision == TSOD_PREC_BF16X3) {
        switch (sc.tile) {
            case TSOD_TILE_64x64_S1: launch_tile<64, 64, 32, 32, 5, 1, 32, 1>(p, sc.grid, s); break;
            case TSOD_TILE_128x64_W8_S1: launch_tile<128, 64, 32, 32, 3, 1, 32, 1>(p, sc.grid, s); break;
            case TSOD_TILE_64x64_S1_K64: launch_tile<64, 64, 32, 32, 2, 1, 64, 1>(p, sc.grid, s); break;
            case TSOD_TILE_128x64_S1: launch_tile<128, 64, 64, 32, 4, 1, 32, 1>(p, sc.grid, s); break;
            case TSOD_TILE_64x128_S1: launch_tile<64, 128, 32, 64, 4, 1, 32, 1>(p, sc.grid, s); break;
            case TSOD_TILE_128x128_S1: launch_tile<128, 128, 64, 64, 2, 1, 32, 1>(p, sc.grid, s); break;
            case TSOD_TILE_D128x128: launch_dma_tile<128, 16, 1, 4>(p, sc.grid, s); break;
            case TSOD_TILE_D64x128: launch_dma_tile<64, 32, 2, 3>(p, sc.grid, s); break;
            case TSOD_TILE_D256x128: launch_dma_tile<256, 16, 1, 4>(p, sc.grid, s); break;
            case TSOD_TILE_D64x128_S2: launch_dma_tile<64, 32, 2, 2>(p, sc.grid, s); break;
            case TSOD_TILE_D128x256: launch_dma_tile<128, 16, 1, 4, 2>(p, sc.grid, s); break;
            case TSOD_TILE_D128x128_K32: launch_dma_tile<128, 32, 2, 3>(p, sc.grid, s); break;
            default: launch_tile<64, 64, 32, 32, 2, 2, 32, 1>(p, sc.grid, s); break;      // TSOD_TILE_64x64 (two LDS stages)
        }
        return tsod_launch_status();
    }
    switch (sc.tile) {
        case TSOD_TILE_128x128: launch_tile<128, 128, 64, 64, 2>(p, sc.grid, s); break;
        case TSOD_TILE_128x64: launch_tile<128, 64, 64, 32, 2>(p, sc.grid, s); break;
        case TSOD_TILE_64x128: launch_tile<64, 128, 32, 64, 2>(p, sc.grid, s); break;
        case TSOD_TILE_128x128_W8: launch_tile<128, 128, 64, 32, 4>(p, sc.grid, s); break;
        case TSOD_TILE_128x64_W8: launch_tile<128, 64, 32, 32, 4>(p, sc.grid, s); break;
        case TSOD_TILE_256x128_W8: launch_tile<256, 128, 64, 64, 2>(p, sc.grid, s); break;
        case TSOD_TILE_64x64_S1: launch_tile<64, 64, 32, 32, 6, 1>(p, sc.grid, s); break;
        case TSOD_TILE_128x64_W8_S1: launch_tile<128, 64, 32, 32, 6, 1>(p, sc.grid, s); break;
        case TSOD_TILE_64x64_S1_K64: launch_tile<64, 64, 32, 32, 4, 1, 64>(p, sc.grid, s); break;
        case TSOD_TILE_128x64_W8_S1_K64: launch_tile<128, 64, 32, 32, 4, 1, 64>(p, sc.grid, s); break;
        case TSOD_TILE_64x64_W1_S1: launch_tile<64, 64, 64, 64, 2, 1>(p, sc.grid, s); break;
        case TSOD_TILE_128x64_W2_S1: launch_tile<128, 64, 64, 64, 2, 1>(p, sc.grid, s); break;
        case TSOD_TILE_128x64_S1: launch_tile<128, 64, 64, 32, 4, 1>(p, sc.grid, s); break;
        case TSOD_TILE_64x128_S1: launch_tile<64, 128, 32, 64, 4, 1>(p, sc.grid, s); break;
        case TSOD_TILE_128x128_S1: launch_tile<128, 128, 64, 64, 2, 1>(p, sc.grid, s); break;
        default: launch_tile<64, 64, 32, 32, 4>(p, sc.grid, s); break;
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

extern "C" size_t tsod_conv_weight_bf16x3_bytes(int32_t Cout, int32_t K) {
    if (Cout <= 0 || K <= 0) return 0;
    return (size_t)Cout * ((K + 7) / 8) * 48;
}

extern "C" int tsod_pack_conv_weight_bf16x3(const float *w_packed, int32_t Cout, int32_t K, void *w_bf16x3, tsod_stream_t stream) {
    TSOD_REQUIRE(w_packed && w_bf16x3, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(Cout > 0 && K > 0, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(tsod_aligned16(w_bf16x3), TSOD_ERR_ALIGNMENT);
    const long total = (long)Cout * ((K + 7) / 8);
    const int blocks = (int)(tsod_cdiv(total, 256) < 4096 ? tsod_cdiv(total, 256) : 4096);
    hipLaunchKernelGGL(pack_weight_bf16x3_kernel, dim3(blocks), dim3(256), 0, tsod_stream(stream), w_packed, Cout, K,
                       static_cast<unsigned *>(w_bf16x3));
    return tsod_launch_status();
}

extern "C" size_t tsod_conv_weight_fp16x2_bytes(int32_t Cout, int32_t K) {
    if (Cout <= 0 || K <= 0) return 0;
    return (size_t)Cout * ((K + 7) / 8) * 32;
}

extern "C" int tsod_pack_conv_weight_fp16x2(const float *w_packed, int32_t Cout, int32_t K, int32_t w_scale_exp, void *w_fp16x2,
                                            tsod_stream_t stream) {
    TSOD_REQUIRE(w_packed && w_fp16x2, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(Cout > 0 && K > 0 && w_scale_exp >= -40 && w_scale_exp <= 40, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(tsod_aligned16(w_fp16x2), TSOD_ERR_ALIGNMENT);
    const long total = (long)Cout * ((K + 7) / 8);
    const int blocks = (int)(tsod_cdiv(total, 256) < 4096 ? tsod_cdiv(total, 256) : 4096);
    hipLaunchKernelGGL(pack_weight_fp16x2_kernel, dim3(blocks), dim3(256), 0, tsod_stream(stream), w_packed, Cout, K,
                       ldexpf(1.f, w_scale_exp), static_cast<unsigned *>(w_fp16x2));
    return tsod_launch_status();
}

#ifdef TSOD_CLOCK_DIAG
extern "C" int tsod_debug_set_clock_buf(long long *buf /* device, 2 * 32768 int64, or NULL */) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_clock_buf), &buf, sizeof(buf)) == hipSuccess ? TSOD_OK : TSOD_ERR_LAUNCH;
}
extern "C" int tsod_debug_set_dma_stamps(long long *buf /* device, 8 * 8192 int64, or NULL */) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_dma_stamps), &buf, sizeof(buf)) == hipSuccess ? TSOD_OK : TSOD_ERR_LAUNCH;
}
extern "C" int tsod_debug_set_epi_stamps(long long *buf /* device, 8 * 8192 int64, or NULL */) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_epi_stamps), &buf, sizeof(buf)) == hipSuccess ? TSOD_OK : TSOD_ERR_LAUNCH;
}
#endif
