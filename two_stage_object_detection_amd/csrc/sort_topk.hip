// sort_topk.hip -- per-image stable descending top-k of the RPN scores.
// Replaces torch.argsort(score, descending=True)[:n_pre_nms] + gathers (nets/rpn.py:56-61) and the
// min-size compaction before it (nets/rpn.py:52-54: filtered entries arrive as key = -inf).
//
// Small problems (B * n^2 <= 3e8): ONE launch that ranks every key against every key of its image on the whole chip
// (topk_rank_kernel<true>).  Larger ones: a selection pass per image (below, steps 1-3), then the same rank kernel over the
// selection when the caller brought scratch for it (tsod_sort_topk_desc_ws_f32), else steps 4-5 in the selecting workgroup.
// The selection pass: one 1024-thread workgroup per image, everything in LDS:
//   1. map f32 -> u32 "descending-orderable" d (smaller d = larger score),
//   2. 4x8-bit MSB-first radix select of the n_sel-th smallest d (histograms in LDS),
//   3. compaction of every d < T plus the first `need` entries with d == T in index order
//      (that is the stable tie rule: lower index first),
//   4. bitonic sort of the <= n_pre composite keys (d << 32 | index): registers + lane shuffles, LDS only for the
//      cross-wave distances,
//   5. gather boxes / keys in sorted order.
// Integer/index work: bit-exact against the oracle by construction.  No MFMA.
#include "tsod_internal.h"
#include <math.h>

namespace {

constexpr int kThreads = 1024;              // 16 waves: the bitonic stages are LDS-latency bound, more waves hide it
constexpr unsigned kDNegInf = 0xFF800000u;  // d(-inf)

__device__ __forceinline__ unsigned desc_key_bits(unsigned u) {
    u = u == 0x80000000u ? 0u : u;                    // -0 -> +0 so that they tie like torch's comparison does
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending-orderable
    return ~u;                                        // descending-orderable
}
__device__ __forceinline__ unsigned desc_key(float f) { return desc_key_bits(__float_as_uint(f)); }

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int lane_mask) {
    const unsigned lo = __shfl_xor((unsigned)v, lane_mask);
    const unsigned hi = __shfl_xor((unsigned)(v >> 32), lane_mask);
    return ((unsigned long long)hi << 32) | lo;
}

// Ascending bitonic sort of S = E * kThreads composite keys held in LDS.  Thread t keeps elements [t*E, t*E+E) in
// registers for the whole network: exchanges at distance < E are register swaps, at distance < 64*E lane shuffles
// inside the wave, and only the few longest distances of each merge go through LDS (two barriers each).  For
// S = 4096 that is 10 LDS stages out of 78.
template <int E>
__device__ __noinline__ void block_bitonic(unsigned long long *sm, int tid) {
    constexpr int S = E * kThreads;
    constexpr int LOG_E = E == 1 ? 0 : E == 2 ? 1 : E == 4 ? 2 : E == 8 ? 3 : 4;
    unsigned long long v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = sm[tid * E + e];
    for (int kk = 2; kk <= S; kk <<= 1) {
        int j = kk >> 1;
        for (; j >= 64 * E; j >>= 1) {            // partner lives in another wave
            __syncthreads();
#pragma unroll
            for (int e = 0; e < E; ++e) sm[tid * E + e] = v[e];
            __syncthreads();
            const int pbase = (tid ^ (j >> LOG_E)) * E;
            const bool is_lo = ((tid * E) & j) == 0;
            const bool up = ((tid * E) & kk) == 0;       // kk > j >= E: the same for all E elements
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const unsigned long long o = sm[pbase + e];
                const bool take_min = is_lo == up;
                v[e] = take_min ? (o < v[e] ? o : v[e]) : (o > v[e] ? o : v[e]);
            }
        }
        for (; j >= E; j >>= 1) {                 // partner is a lane of this wave
            const bool is_lo = ((tid * E) & j) == 0;
            const bool up = ((tid * E) & kk) == 0;      // j >= E implies kk > E
            const bool take_min = is_lo == up;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const unsigned long long o = shfl_xor_u64(v[e], j >> LOG_E);
                v[e] = take_min ? (o < v[e] ? o : v[e]) : (o > v[e] ? o : v[e]);
            }
        }
#pragma unroll
        for (int jj = E / 2; jj > 0; jj >>= 1) {  // partner is another register of this thread
            if (jj > (kk >> 1)) continue;
#pragma unroll
            for (int e = 0; e < E; ++e) {
                if (e & jj) continue;
                const bool up = ((tid * E + e) & kk) == 0;
                const unsigned long long a = v[e], c = v[e | jj];
                if ((a > c) == up) { v[e] = c; v[e | jj] = a; }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) sm[tid * E + e] = v[e];
    __syncthreads();
}

// sel_out != nullptr: SELECT ONLY - the n_sel composite keys (d << 32 | index) of the selection are written, unordered, to
// sel_out[b][0 .. n_sel) and the kernel ends there; topk_rank_kernel orders them on the whole chip.
template <int KPT>  // keys per thread held in registers: every pass after the first runs without touching memory
__global__ void __launch_bounds__(kThreads)
sort_topk_kernel(const float *__restrict__ keys, const float *__restrict__ boxes, int n, int n_pre, int P,
                 int *__restrict__ counts, int *__restrict__ idx_out, float *__restrict__ boxes_out,
                 float *__restrict__ keys_out, unsigned long long *__restrict__ sel_out) {
    extern __shared__ __align__(16) unsigned long long sm[];  // P composite keys
    __shared__ unsigned hist[256];
    __shared__ unsigned wave_tot[kThreads / 64];
    __shared__ unsigned s_prefix, s_need, s_nvalid, s_cnt, s_eq_base;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.x;
    const float *k = keys + (long)b * n;

    if (tid == 0) { s_nvalid = 0; s_cnt = 0; s_eq_base = 0; s_prefix = 0; }
    // one pass over memory: all of this thread's loads are in flight together (a load per loop iteration in each
    // of the six passes below was one exposed L2 round trip each, ~50 us in total)
    unsigned dk[KPT];
#pragma unroll
    for (int q = 0; q < KPT; ++q) {
        const int i = q * kThreads + tid;
        dk[q] = i < n ? desc_key(k[i]) : 0xFFFFFFFFu;     // 0xFFFFFFFF (> d(-inf)) never passes a filter below
    }
    __syncthreads();
    {
        unsigned local = 0;
#pragma unroll
        for (int q = 0; q < KPT; ++q) local += dk[q] < kDNegInf ? 1u : 0u;
        for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off);
        if (lane == 0 && local) atomicAdd(&s_nvalid, local);
    }
    __syncthreads();
    const int n_sel = min((int)s_nvalid, n_pre);
    if (tid == 0) { counts[b] = n_sel; s_need = (unsigned)n_sel; }
    __syncthreads();

    if (n_sel > 0) {
        // ---- radix select, MSB first: after the 4 passes s_prefix is the d of rank n_sel (1-based), s_need the
        //      number of entries equal to it that belong to the selection, count_eq how many equal it in total.
        unsigned mask = 0, count_eq = 0;
        for (int shift = 24; shift >= 0; shift -= 8) {
            if (tid < 256) hist[tid] = 0;
            __syncthreads();
            const unsigned prefix = s_prefix;
#pragma unroll
            for (int q = 0; q < KPT; ++q) {
                if (q * kThreads >= n) break;
                const int i = q * kThreads + tid;
                const unsigned d = dk[q];
                bool pending = i < n && (d & mask) == prefix;
                const unsigned bin = (d >> shift) & 255u;
                // scores share sign/exponent bits, so a leading digit can put a whole wave into one bin: count the
                // first pending lane's bin once per wave (ballot + popcount, one LDS atomic) instead of up to 64
                // serialised same-address atomics; lanes in other bins use plain LDS atomics (spread digits rarely
                // collide).  The lane index comes from a ballot, so it is wave-uniform: v_readlane, no LDS shuffle.
                const unsigned long long live = __ballot(pending);
                if (live != 0ull) {
                    const int leader = __ffsll((long long)live) - 1;
                    const unsigned lb = (unsigned)__builtin_amdgcn_readlane((int)bin, leader);
                    const bool same = pending && bin == lb;
                    const unsigned long long grp = __ballot(same);
                    if (lane == leader) atomicAdd(&hist[lb], (unsigned)__popcll(grp));
                    if (pending && !same) atomicAdd(&hist[bin], 1u);
                }
            }
            __syncthreads();
            // inclusive scan of the 256 bins by the first 4 waves: shuffles inside a wave, wave totals through LDS
            const unsigned mine = tid < 256 ? hist[tid] : 0u;
            unsigned incl = mine;
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned v = __shfl_up(incl, off);
                if (lane >= off) incl += v;
            }
            if (lane == 63 && wid < 4) wave_tot[wid] = incl;
            __syncthreads();
            for (int w = 0; w < wid && w < 4; ++w) incl += wave_tot[w];
            const unsigned need = s_need;
            const unsigned excl = incl - mine;
            __syncthreads();
            if (tid < 256 && excl < need && need <= incl) {  // exactly one bin satisfies this
                s_prefix = prefix | ((unsigned)tid << shift);
                s_need = need - excl;
                if (shift == 0) s_eq_base = mine;   // entries equal to the threshold value
            }
            mask |= 0xFFu << shift;
            __syncthreads();
        }
        const unsigned T = s_prefix;
        const unsigned need_eq = s_need;
        count_eq = s_eq_base;
        const unsigned count_lt = (unsigned)n_sel - need_eq;
        __syncthreads();
        if (tid == 0) s_eq_base = 0;
        __syncthreads();

        if (need_eq == count_eq) {
            // common case: the whole tie group at the threshold is selected -> any slot will do, the sort
            // below orders by (d, index)
#pragma unroll
            for (int q = 0; q < KPT; ++q) {
                const int i = q * kThreads + tid;
                const unsigned d = dk[q];
                if (i < n && d <= T) sm[atomicAdd(&s_cnt, 1u)] = ((unsigned long long)d << 32) | (unsigned)i;
            }
        } else {
            // the tie group straddles the cut: take its first need_eq members in index order (stable rule)
#pragma unroll
            for (int q = 0; q < KPT; ++q) {
                if (q * kThreads >= n) break;
                const int i = q * kThreads + tid;
                const unsigned d = dk[q];
                if (i < n && d < T) sm[atomicAdd(&s_cnt, 1u)] = ((unsigned long long)d << 32) | (unsigned)i;
                const bool is_eq = (i < n) && (d == T);
                const unsigned long long bal = __ballot(is_eq);
                const unsigned in_wave = __popcll(bal & ((1ull << lane) - 1ull));
                if (lane == 0) wave_tot[wid] = (unsigned)__popcll(bal);
                __syncthreads();
                unsigned before = s_eq_base, total = 0;
                for (int w = 0; w < kThreads / 64; ++w) {
                    const unsigned t = wave_tot[w];
                    if (w < wid) before += t;
                    total += t;
                }
                const unsigned rank = before + in_wave;
                if (is_eq && rank < need_eq) sm[count_lt + rank] = ((unsigned long long)d << 32) | (unsigned)i;
                __syncthreads();
                if (tid == 0) s_eq_base += total;
                __syncthreads();
            }
        }
        if (sel_out != nullptr) {             // select only: hand the unordered selection to the rank kernel
            __syncthreads();
            for (int i = tid; i < n_sel; i += kThreads) sel_out[(long)b * P + i] = sm[i];
            return;
        }
        int S = kThreads;                 // sort only the power of two that covers the selection (>= one key per thread)
        while (S < n_sel) S <<= 1;
        for (int i = n_sel + tid; i < S; i += kThreads) sm[i] = ~0ull;
        __syncthreads();
        switch (S / kThreads) {
            case 1: block_bitonic<1>(sm, tid); break;
            case 2: block_bitonic<2>(sm, tid); break;
            case 4: block_bitonic<4>(sm, tid); break;
            case 8: block_bitonic<8>(sm, tid); break;
            default: block_bitonic<16>(sm, tid); break;
        }
    }
    if (sel_out != nullptr) return;       // (n_sel == 0: the rank kernel writes the neutral rows)

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

// ---- ordering by RANK on the whole chip.  The composite keys (d << 32 | index) are all distinct, so the position of a key
// in the sorted order is simply the number of keys smaller than it: no network, no barriers between stages, and it
// parallelises over every CU instead of living in ONE workgroup (the bitonic network of 4096 keys took ~25 us of the 50).
// A workgroup = 64 candidate keys (one per lane) x 16 waves, each wave counting over its own 16th of the candidates; the
// "other" key of an iteration is the same for all lanes of a wave, so it arrives through the scalar unit (s_load) and the
// loop body is two vector instructions (64-bit compare + add).
//   FULL  = true : candidates = ALL n keys of the image (no selection pass at all): rank r < min(n_valid, n_pre) -> output row
//                  r; the filtered keys (d >= d(-inf)) sort last and write the neutral rows.  Work B * n^2: small problems.
//   FULL  = false: candidates = the n_sel keys sort_topk_kernel selected (sel[b][0 .. counts[b])), work B * n_sel^2.
constexpr int kRankParts = 16;
constexpr int kRankMaxPer = 512;                   // candidates one wave ranks against per pass (its private LDS strip)
template <bool FULL>
__global__ void __launch_bounds__(64 * kRankParts)
topk_rank_kernel(const float *__restrict__ keys, const float *__restrict__ boxes, const unsigned long long *__restrict__ sel, int P,
                 int n, int n_pre, int *__restrict__ counts, int *__restrict__ idx_out, float *__restrict__ boxes_out,
                 float *__restrict__ keys_out) {
    // per wave a private strip of "other" keys: d in s_d, the source index (only when it is not the position) in s_i
    __shared__ __align__(16) unsigned s_d[kRankParts][kRankMaxPer];
    __shared__ __align__(16) unsigned s_i[FULL ? 1 : kRankParts][FULL ? 4 : kRankMaxPer];
    __shared__ unsigned s_rank[kRankParts][64];
    __shared__ unsigned s_valid[kRankParts];
    const int tid = threadIdx.x, lane = tid & 63;
    const int part = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y, chunk = blockIdx.x;
    const float *k = keys + (long)b * n;
    const unsigned *kbits = reinterpret_cast<const unsigned *>(k);
    const unsigned long long *sb = sel + (long)b * P;
    const int n_cand = FULL ? n : __builtin_amdgcn_readfirstlane(counts[b]);
    const int ci = chunk * 64 + lane;                         // this lane's candidate (its POSITION in the candidate list)
    unsigned md = 0xFFFFFFFFu, mi = 0xFFFFFFFFu;              // its key: d and source index
    if (ci < n_cand) {
        if (FULL) { md = desc_key_bits(kbits[ci]); mi = (unsigned)ci; }
        else { const unsigned long long o = sb[ci]; md = (unsigned)(o >> 32); mi = (unsigned)o; }
    }
    // the candidate list is cut into kRankParts strips of whole 64-blocks; wave `part` counts, for every lane's key, the keys
    // of its strip that sort before it.  Order = (d, source index).  FULL: the list position IS the source index, so a
    // 64-block of others lies entirely before this workgroup's block (then "d_j <= d_mine" decides), entirely after it
    // ("d_j < d_mine"), or is the block itself (full comparison): one 32-bit compare + one add per pair in the first two cases.
    const int blocks = (n_cand + 63) / 64;
    const int bpp = (blocks + kRankParts - 1) / kRankParts;   // 64-blocks per strip
    const int lo = min(n_cand, part * bpp * 64), hi = min(n_cand, lo + bpp * 64);
    unsigned less = 0, valid = 0, eq = 0;
    for (int p0 = lo; p0 < hi; p0 += kRankMaxPer) {            // (one pass unless a strip exceeds the LDS strip)
        const int p1 = min(hi, p0 + kRankMaxPer);
        for (int j = p0 + lane; j < p0 + ((p1 - p0 + 63) & ~63); j += 64) {   // whole 64-blocks: the tail is padding
            unsigned dj = 0xFFFFFFFFu, ij = 0xFFFFFFFFu;      // padding: sorts after every real key
            if (j < p1) {
                if (FULL) { dj = desc_key_bits(kbits[j]); }
                else { const unsigned long long o = sb[j]; dj = (unsigned)(o >> 32); ij = (unsigned)o; }
            }
            if (FULL) valid += (unsigned)__popcll(__ballot(j < p1 && dj < kDNegInf));
            s_d[part][j - p0] = dj;
            if (!FULL) s_i[part][j - p0] = ij;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // wave-private strip: in-order LDS, no barrier needed
        const int nblk = (p1 - p0 + 63) / 64;
        for (int bl = 0; bl < nblk; ++bl) {
            const int jb = p0 + bl * 64;                        // first list position of this block of others
            const uint4 *dq = reinterpret_cast<const uint4 *>(&s_d[part][bl * 64]);
            if (FULL && jb + 63 < chunk * 64) {                 // (wave-uniform) wholly before my block: ties go to them
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const uint4 d4 = dq[q];                      // same address in every lane: an LDS broadcast
                    less += (d4.x <= md) + (d4.y <= md) + (d4.z <= md) + (d4.w <= md);
                }
            } else if (FULL && jb >= chunk * 64 + 64) {         // wholly after my block: ties go to me
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const uint4 d4 = dq[q];
                    less += (d4.x < md) + (d4.y < md) + (d4.z < md) + (d4.w < md);
                }
            } else if (FULL) {                                   // my own block: the full (d, index) comparison
#pragma unroll 4
                for (int q = 0; q < 64; ++q) {
                    const unsigned dj = s_d[part][bl * 64 + q];
                    less += (dj < md || (dj == md && (unsigned)(jb + q) < mi)) ? 1u : 0u;
                }
            } else {                                             // a selection (indices in any order): strict count + equal count
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const uint4 d4 = dq[q];
                    less += (d4.x < md) + (d4.y < md) + (d4.z < md) + (d4.w < md);
                    eq += (d4.x == md) + (d4.y == md) + (d4.z == md) + (d4.w == md);
                }
            }
        }
        if (!FULL) {
            // ties: keys of this strip equal to a lane's key (its own position excluded) are ordered by source index.  Scores are
            // f32 probabilities, so this pass is rare; it walks the strip only when some lane of the wave has such a tie.
            const bool own_here = ci >= p0 && ci < p1;
            if (__ballot(ci < n_cand && eq > (own_here ? 1u : 0u)) != 0ull) {
                for (int q = 0; q < p1 - p0; ++q) {
                    const unsigned dj = s_d[part][q], ij = s_i[part][q];
                    less += (dj == md && ij < mi) ? 1u : 0u;
                }
            }
            eq = 0;
        }
    }
    s_rank[part][lane] = less;
    if (lane == 0) s_valid[part] = valid;
    __syncthreads();
    if (part != 0) return;
    unsigned rank = 0, n_valid = 0;
#pragma unroll
    for (int q = 0; q < kRankParts; ++q) { rank += s_rank[q][lane]; n_valid += s_valid[q]; }
    if (!FULL) n_valid = (unsigned)n_cand;
    const int n_sel = min((int)n_valid, n_pre);
    if (FULL && chunk == 0 && lane == 0) counts[b] = n_sel;
    auto write_row = [&](int r, int src) {
        idx_out[(long)b * n_pre + r] = src;
        if (boxes_out != nullptr) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (src >= 0) v = reinterpret_cast<const float4 *>(boxes)[(long)b * n + src];
            reinterpret_cast<float4 *>(boxes_out)[(long)b * n_pre + r] = v;
        }
        if (keys_out != nullptr) keys_out[(long)b * n_pre + r] = src >= 0 ? k[src] : -INFINITY;
    };
    if (ci < n_cand && (int)rank < n_pre) write_row((int)rank, (int)rank < n_sel ? (int)mi : -1);
    // rows no candidate owns: beyond the candidates (FULL: r >= n; else r >= n_sel), neutral
    if (FULL) {
        if (chunk == 0)
            for (int r = n + lane; r < n_pre; r += 64) write_row(r, -1);
    } else if (ci >= n_cand && ci < n_pre) {
        write_row(ci, -1);
    }
}

// Same selection for key rows too long to hold in registers (n > 80 * 1024): every pass re-reads the keys from global
// memory (six sweeps of an L2-resident row), plain LDS histogram atomics, ties always compacted in index order.
__global__ void __launch_bounds__(kThreads)
sort_topk_stream_kernel(const float *__restrict__ keys, const float *__restrict__ boxes, int n, int n_pre,
                        int *__restrict__ counts, int *__restrict__ idx_out, float *__restrict__ boxes_out,
                        float *__restrict__ keys_out) {
    extern __shared__ __align__(16) unsigned long long sm[];
    __shared__ unsigned hist[256];
    __shared__ unsigned wave_tot[kThreads / 64];
    __shared__ unsigned s_prefix, s_need, s_nvalid, s_lt, s_eq_base;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int b = blockIdx.x;
    const float *k = keys + (long)b * n;
    if (tid == 0) { s_nvalid = 0; s_lt = 0; s_eq_base = 0; s_prefix = 0; }
    __syncthreads();
    {
        unsigned local = 0;
        for (int i = tid; i < n; i += kThreads) local += desc_key(k[i]) < kDNegInf ? 1u : 0u;
        for (int off = 32; off > 0; off >>= 1) local += __shfl_xor(local, off);
        if (lane == 0 && local) atomicAdd(&s_nvalid, local);
    }
    __syncthreads();
    const int n_sel = min((int)s_nvalid, n_pre);
    if (tid == 0) { counts[b] = n_sel; s_need = (unsigned)n_sel; }
    __syncthreads();
    if (n_sel > 0) {
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
            const unsigned mine = tid < 256 ? hist[tid] : 0u;
            unsigned incl = mine;
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned v = __shfl_up(incl, off);
                if (lane >= off) incl += v;
            }
            if (lane == 63 && wid < 4) wave_tot[wid] = incl;
            __syncthreads();
            for (int w = 0; w < wid && w < 4; ++w) incl += wave_tot[w];
            const unsigned need = s_need;
            const unsigned excl = incl - mine;
            __syncthreads();
            if (tid < 256 && excl < need && need <= incl) {
                s_prefix = prefix | ((unsigned)tid << shift);
                s_need = need - excl;
            }
            mask |= 0xFFu << shift;
            __syncthreads();
        }
        const unsigned T = s_prefix;
        const unsigned need_eq = s_need;
        const unsigned count_lt = (unsigned)n_sel - need_eq;
        // d < T: any slot in [0, count_lt) (the sort fixes the order); d == T: slot count_lt + rank among equals in index order
        for (int base = 0; base < n; base += kThreads) {
            const int i = base + tid;
            const unsigned d = i < n ? desc_key(k[i]) : 0xFFFFFFFFu;
            if (i < n && d < T) sm[atomicAdd(&s_lt, 1u)] = ((unsigned long long)d << 32) | (unsigned)i;
            const bool is_eq = (i < n) && (d == T);
            const unsigned long long bal = __ballot(is_eq);
            if (lane == 0) wave_tot[wid] = (unsigned)__popcll(bal);
            __syncthreads();
            unsigned before = s_eq_base, total = 0;
            for (int w = 0; w < kThreads / 64; ++w) {
                const unsigned t = wave_tot[w];
                if (w < wid) before += t;
                total += t;
            }
            const unsigned rank = before + __popcll(bal & ((1ull << lane) - 1ull));
            if (is_eq && rank < need_eq) sm[count_lt + rank] = ((unsigned long long)d << 32) | (unsigned)i;
            __syncthreads();
            if (tid == 0) s_eq_base += total;
            __syncthreads();
        }
        int S = kThreads;
        while (S < n_sel) S <<= 1;
        for (int i = n_sel + tid; i < S; i += kThreads) sm[i] = ~0ull;
        __syncthreads();
        switch (S / kThreads) {
            case 1: block_bitonic<1>(sm, tid); break;
            case 2: block_bitonic<2>(sm, tid); break;
            case 4: block_bitonic<4>(sm, tid); break;
            case 8: block_bitonic<8>(sm, tid); break;
            default: block_bitonic<16>(sm, tid); break;
        }
    }
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

// full rank sort (no selection pass) while B * n^2 compares stay cheap on 256 CUs; beyond that select first, rank the selection
constexpr double kFullRankWork = 3.0e8;

extern "C" size_t tsod_sort_topk_workspace_bytes(int32_t B, int32_t n, int32_t n_pre) {
    if (B <= 0 || n <= 0 || n_pre <= 0 || n_pre > 16384) return 0;
    if ((double)B * n * n <= kFullRankWork || n > 80 * kThreads) return 0;
    int P = kThreads;
    while (P < n_pre) P <<= 1;
    return (size_t)B * P * sizeof(unsigned long long);
}

extern "C" int tsod_sort_topk_desc_ws_f32(const float *keys, const float *boxes, int32_t B, int32_t n, int32_t n_pre,
                                          int32_t *counts, int32_t *idx, float *boxes_out, float *keys_out, void *workspace,
                                          size_t workspace_bytes, tsod_stream_t stream) {
    TSOD_REQUIRE(keys && counts && idx, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(B > 0 && n > 0 && n_pre > 0, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(n_pre <= 16384, TSOD_ERR_UNSUPPORTED);
    TSOD_REQUIRE(boxes_out == nullptr || boxes != nullptr, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE((boxes == nullptr || tsod_aligned16(boxes)) && (boxes_out == nullptr || tsod_aligned16(boxes_out)),
                 TSOD_ERR_ALIGNMENT);
    int P = kThreads;                       // the sort network runs on >= one key per thread
    while (P < n_pre) P <<= 1;
    const size_t lds = (size_t)P * sizeof(unsigned long long);
    hipStream_t st = tsod_stream(stream);
    if ((double)B * n * n <= kFullRankWork && B <= 65535) {
        // small problem: every key ranked against every key of its image, ONE launch on (n / 64) x B workgroups
        hipLaunchKernelGGL(topk_rank_kernel<true>, dim3((n + 63) / 64, B), dim3(64 * kRankParts), 0, st, keys, boxes,
                           (const unsigned long long *)nullptr, P, n, n_pre, counts, idx, boxes_out, keys_out);
        return tsod_launch_status();
    }
    const size_t need = tsod_sort_topk_workspace_bytes(B, n, n_pre);
    // select per image, then rank the selection on the whole chip - when the caller brought the scratch for it
    unsigned long long *sel = (workspace != nullptr && need > 0 && workspace_bytes >= need && tsod_aligned16(workspace) && B <= 65535)
                                  ? static_cast<unsigned long long *>(workspace) : nullptr;
#define TSOD_SORT(KPT)                                                                                                 \
    do {                                                                                                               \
        if (lds > 48 * 1024 &&                                                                                         \
            hipFuncSetAttribute(reinterpret_cast<const void *>(sort_topk_kernel<KPT>),                                 \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {                 \
            (void)hipGetLastError();                                                                                   \
            return TSOD_ERR_UNSUPPORTED;                                                                               \
        }                                                                                                              \
        hipLaunchKernelGGL(sort_topk_kernel<KPT>, dim3(B), dim3(kThreads), lds, st, keys, boxes, n, n_pre, P, counts,  \
                           idx, boxes_out, keys_out, sel);                                                             \
    } while (0)
    if (n > 80 * kThreads) {
        if (lds > 48 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void *>(sort_topk_stream_kernel),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            (void)hipGetLastError();
            return TSOD_ERR_UNSUPPORTED;
        }
        hipLaunchKernelGGL(sort_topk_stream_kernel, dim3(B), dim3(kThreads), lds, st, keys, boxes, n, n_pre, counts, idx,
                           boxes_out, keys_out);
        return tsod_launch_status();
    }
    if (n <= 10 * kThreads) TSOD_SORT(10);
    else if (n <= 20 * kThreads) TSOD_SORT(20);
    else if (n <= 40 * kThreads) TSOD_SORT(40);
    else TSOD_SORT(80);
#undef TSOD_SORT
    if (sel != nullptr) {
        const int rc = tsod_launch_status();
        if (rc != TSOD_OK) return rc;
        hipLaunchKernelGGL(topk_rank_kernel<false>, dim3((n_pre + 63) / 64, B), dim3(64 * kRankParts), 0, st, keys, boxes, sel, P,
                           n, n_pre, counts, idx, boxes_out, keys_out);
    }
    return tsod_launch_status();
}

extern "C" int tsod_sort_topk_desc_f32(const float *keys, const float *boxes, int32_t B, int32_t n, int32_t n_pre,
                                       int32_t *counts, int32_t *idx, float *boxes_out, float *keys_out,
                                       tsod_stream_t stream) {
    return tsod_sort_topk_desc_ws_f32(keys, boxes, B, n, n_pre, counts, idx, boxes_out, keys_out, nullptr, 0, stream);
}
