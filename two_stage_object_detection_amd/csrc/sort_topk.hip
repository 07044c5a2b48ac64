// sort_topk.hip -- per-image stable descending top-k of the RPN scores.
// Replaces torch.argsort(score, descending=True)[:n_pre_nms] + gathers (nets/rpn.py:56-61) and the
// min-size compaction before it (nets/rpn.py:52-54: filtered entries arrive as key = -inf).
//
// One 1024-thread workgroup per image, everything in LDS after one pass over the keys:
//   1. map f32 -> u32 "descending-orderable" d (smaller d = larger score),
//   2. 4x8-bit MSB-first radix select of the n_sel-th smallest d (histograms in LDS),
//   3. compaction of every d < T plus the first `need` entries with d == T in index order
//      (that is the stable tie rule: lower index first),
//   4. bitonic sort of the <= n_pre composite keys (d << 32 | index) in LDS,
//   5. gather boxes / keys in sorted order.
// Integer/index work: bit-exact against the oracle by construction.  No MFMA.
#include "tsod_internal.h"
#include <math.h>

namespace {

constexpr int kThreads = 1024;
constexpr unsigned kDNegInf = 0xFF800000u;  // d(-inf)

__device__ __forceinline__ unsigned desc_key(float f) {
    f = f + 0.0f;  // -0 -> +0 so that they tie like torch's comparison does
    unsigned u = __float_as_uint(f);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending-orderable
    return ~u;                                        // descending-orderable
}

__global__ void __launch_bounds__(kThreads)
sort_topk_kernel(const float *__restrict__ keys, const float *__restrict__ boxes, int n, int n_pre, int P,
                 int *__restrict__ counts, int *__restrict__ idx_out, float *__restrict__ boxes_out,
                 float *__restrict__ keys_out) {
    extern __shared__ __align__(16) unsigned long long sm[];  // P composite keys
    __shared__ unsigned hist[256];
    __shared__ unsigned scan[256];
    __shared__ unsigned wave_tot[kThreads / 64];
    __shared__ unsigned s_prefix, s_need, s_nvalid, s_lt, s_eq_base;

    const int tid = threadIdx.x;
    const int b = blockIdx.x;
    const float *k = keys + (long)b * n;

    if (tid == 0) { s_nvalid = 0; s_lt = 0; s_eq_base = 0; s_prefix = 0; }
    __syncthreads();
    {
        unsigned local = 0;
        for (int i = tid; i < n; i += kThreads) local += desc_key(k[i]) < kDNegInf ? 1u : 0u;
        for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off);
        if ((tid & 63) == 0 && local) atomicAdd(&s_nvalid, local);
    }
    __syncthreads();
    const int n_sel = min((int)s_nvalid, n_pre);
    if (tid == 0) { counts[b] = n_sel; s_need = (unsigned)n_sel; }
    __syncthreads();

    if (n_sel > 0) {
        // ---- radix select: after the 4 passes s_prefix is the d of rank n_sel (1-based) and
        //      s_need the number of entries equal to it that belong to the selection.
        unsigned mask = 0;
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const unsigned prefix = s_prefix;
            for (int i = tid; i < n; i += kThreads) {
                const unsigned d = desc_key(k[i]);
                if ((d & mask) == prefix) atomicAdd(&hist[(d >> shift) & 255u], 1u);
            }
            __syncthreads();
            if (tid < 256) scan[tid] = hist[tid];
            __syncthreads();
            for (int off = 1; off < 256; off <<= 1) {  // inclusive Hillis-Steele scan over 256 bins
                unsigned v = 0;
                if (tid < 256 && tid >= off) v = scan[tid - off];
                __syncthreads();
                if (tid < 256) scan[tid] += v;
                __syncthreads();
            }
            const unsigned need = s_need;
            __syncthreads();
            if (tid < 256) {
                const unsigned incl = scan[tid];
                const unsigned excl = incl - hist[tid];
                if (excl < need && need <= incl) {  // exactly one bin satisfies this
                    s_prefix = prefix | ((unsigned)tid << shift);
                    s_need = need - excl;
                }
            }
            mask |= 0xFFu << shift;
            __syncthreads();
        }
        const unsigned T = s_prefix;
        const unsigned need_eq = s_need;
        const unsigned count_lt = (unsigned)n_sel - need_eq;

        // ---- compaction.  d < T: any slot in [0, count_lt) (the sort fixes the order);
        //      d == T: slot count_lt + rank-among-equals, rank taken in index order.
        for (int base = 0; base < n; base += kThreads) {
            const int i = base + tid;
            const unsigned d = i < n ? desc_key(k[i]) : 0xFFFFFFFFu;
            if (i < n && d < T) {
                const unsigned pos = atomicAdd(&s_lt, 1u);
                sm[pos] = ((unsigned long long)d << 32) | (unsigned)i;
            }
            const bool is_eq = (i < n) && (d == T);
            const unsigned long long bal = __ballot(is_eq);
            const unsigned in_wave = __popcll(bal & ((1ull << (tid & 63)) - 1ull));
            if ((tid & 63) == 0) wave_tot[tid >> 6] = (unsigned)__popcll(bal);
            __syncthreads();
            unsigned before = s_eq_base, total = 0;
            for (int w = 0; w < kThreads / 64; ++w) {
                const unsigned t = wave_tot[w];
                if (w < (tid >> 6)) before += t;
                total += t;
            }
            const unsigned rank = before + in_wave;
            if (is_eq && rank < need_eq) sm[count_lt + rank] = ((unsigned long long)d << 32) | (unsigned)i;
            __syncthreads();
            if (tid == 0) s_eq_base += total;
            __syncthreads();
        }
        for (int i = n_sel + tid; i < P; i += kThreads) sm[i] = ~0ull;
        __syncthreads();

        // ---- bitonic sort, ascending in the composite key.
        for (int kk = 2; kk <= P; kk <<= 1) {
            for (int j = kk >> 1; j > 0; j >>= 1) {
                for (int t = tid; t < (P >> 1); t += kThreads) {
                    const int lo = ((t / j) * (j << 1)) + (t % j);
                    const int hi = lo + j;
                    const unsigned long long a = sm[lo], c = sm[hi];
                    const bool up = (lo & kk) == 0;
                    if ((a > c) == up) { sm[lo] = c; sm[hi] = a; }
                }
                __syncthreads();
            }
        }
    }

    // ---- gather in sorted order; rows beyond n_sel are neutral.
    for (int r = tid; r < n_pre; r += kThreads) {
        int src = -1;
        if (r < n_sel) src = (int)(unsigned)(sm[r] & 0xFFFFFFFFull);
        idx_out[(long)b * n_pre + r] = src;
        if (boxes_out != nullptr) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (src >= 0) v = reinterpret_cast<const float4 *>(boxes)[(long)b * n + src];
            reinterpret_cast<float4 *>(boxes_out)[(long)b * n_pre + r] = v;
        }
        if (keys_out != nullptr) keys_out[(long)b * n_pre + r] = src >= 0 ? k[src] : -INFINITY;
    }
}

}  // namespace

extern "C" int tsod_sort_topk_desc_f32(const float *keys, const float *boxes, int32_t B, int32_t n, int32_t n_pre,
                                       int32_t *counts, int32_t *idx, float *boxes_out, float *keys_out,
                                       tsod_stream_t stream) {
    TSOD_REQUIRE(keys && counts && idx, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(B > 0 && n > 0 && n_pre > 0, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(n_pre <= 16384, TSOD_ERR_UNSUPPORTED);
    TSOD_REQUIRE(boxes_out == nullptr || boxes != nullptr, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE((boxes == nullptr || tsod_aligned16(boxes)) && (boxes_out == nullptr || tsod_aligned16(boxes_out)),
                 TSOD_ERR_ALIGNMENT);
    int P = 2;
    while (P < n_pre) P <<= 1;
    const size_t lds = (size_t)P * sizeof(unsigned long long);
    if (lds > 48 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(sort_topk_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return TSOD_ERR_UNSUPPORTED;
        }
    }
    hipLaunchKernelGGL(sort_topk_kernel, dim3(B), dim3(kThreads), lds, tsod_stream(stream), keys, boxes, n, n_pre, P,
                       counts, idx, boxes_out, keys_out);
    return tsod_launch_status();
}
