// Probe for a bf16x3 GEMM loop fed entirely by LDS-DMA (`buffer_load_dwordx4 ... offen lds`): no VGPR staging, no ds_write.
//   A [M][K] f32 goes to LDS RAW (16-B slots XOR-swizzled through the per-lane SOURCE address); the wave that owns a
//   fragment splits it into the three bf16 pieces in registers, in the shadow of its MFMAs (no element is split twice);
//   B is the pre-split weight image of the conv path ([N][K/8][hi|mid|lo][8] bf16) and lands in three unpadded LDS planes.
// Wave tile 32 x 128 (24 MFMAs per 16 k for 44 VALU of split), BN = 128.  Waves along M (WAVES_M) and, for small tiles,
// along K inside the stage (WAVES_K = 2: the two halves are added through LDS at the end).  One phase per stage and wave:
//   barrier (stage p+1 visible, slot of stage p free) -> LDS reads of stage p+1 into the other register set, MFMAs of
//   stage p, split of stage p+1's A fragment, DMA of stage p+S -- every instruction of the loop is a pinned asm statement,
//   so the issue order is the source order.  Counted vmcnt / lgkmcnt by hand (the compiler sees none of these operations).
// C[m][n] = sum_k A[m][k] B[n][k].  Prints TFLOP/s (f32-equivalent) and the error against an f64 reference.
// Build: hipcc -w -O3 --offload-arch=gfx950 bf16x3_dma_probe.hip -o bf16x3_dma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4i make_rsrc(const void *p, unsigned bytes) {
    v4i r; const unsigned long long a = (unsigned long long)p;
    r[0] = (int)(unsigned)a; r[1] = (int)((unsigned)(a >> 32) & 0xffff); r[2] = (int)bytes; r[3] = 0x00020000;
    return r;
}
// ---- the pinned micro-operations (volatile asm keeps their relative order) ----
// one wave-instruction: 64 lanes x 16 B from per-lane source offsets to LDS [lds_dst, lds_dst + 1024)
__device__ __forceinline__ void dma16(unsigned voff, v4i rsrc, unsigned soff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :: "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" :: "n"(N) : "memory"); }
template <int OFF, typename T> __device__ __forceinline__ void lds_read16(T &out, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(out) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void mfma(f32x16 &c, const bf16x8 &a, const bf16x8 &b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
__device__ __forceinline__ void v_cvt_pk(unsigned &d, float x0, float x1) { asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(x0), "v"(x1)); }
__device__ __forceinline__ void v_lo_f32(float &d, unsigned pk) { asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(d) : "v"(pk)); }
__device__ __forceinline__ void v_hi_f32(float &d, unsigned pk) { asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(d) : "v"(pk)); }
__device__ __forceinline__ void v_sub(float &d, float a, float b) { asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b)); }

// x == hi + mid + lo exactly when |x| is not tiny (3 x 8 significand bits); prologue only (compiler-scheduled)
__device__ __forceinline__ void split3_pair(f32x2 x, unsigned &h, unsigned &m, unsigned &l) {
    const bf16x2 hb = __builtin_convertvector(x, bf16x2);
    const f32x2 r1 = x - __builtin_convertvector(hb, f32x2);
    const bf16x2 mb = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(mb, f32x2);
    const bf16x2 lb = __builtin_convertvector(r2, bf16x2);
    h = __builtin_bit_cast(unsigned, hb); m = __builtin_bit_cast(unsigned, mb); l = __builtin_bit_cast(unsigned, lb);
}

template <int BM, int BK, int WAVES_K, int S, int MINB, bool STAMP, int FLAGS = 0>
__global__ void __launch_bounds__((BM / 32) * WAVES_K * 64, MINB)
gemm_dma(const float *__restrict__ A, const unsigned short *__restrict__ W3, float *__restrict__ C, int M, int N, int K,
         long long *__restrict__ stamps) {
    constexpr int BN = 128, TN = 4, WAVES_M = BM / 32, WAVES = WAVES_M * WAVES_K;
    static_assert(BK == 16 * WAVES_K, "every wave owns one 16-k chunk of the stage");
    constexpr int A_ROW = BK * 4, B_ROW = BK * 2;                    // bytes per LDS row (A raw f32 / one bf16 plane of B)
    constexpr int A_SLOTS = A_ROW / 16, B_SLOTS = B_ROW / 16;        // 16-B slots per row
    constexpr int A_RPL = 256 / A_ROW, B_RPL = 256 / B_ROW;          // rows per 256-B bank line: slot ^= (row / RPL) & (SLOTS - 1)
    constexpr int A_BYTES = BM * A_ROW, B_PLANE = BN * B_ROW, STAGE = A_BYTES + 3 * B_PLANE;
    constexpr int A_RPP = 1024 / A_ROW, B_RPP = 1024 / B_ROW;        // rows per 1-KiB DMA piece
    constexpr int A_PIECES = BM / A_RPP, B_PIECES = 3 * (BN / B_RPP), PIECES = A_PIECES + B_PIECES;
    static_assert(A_PIECES % WAVES == 0 && B_PIECES % WAVES == 0, "piece kinds per wave at compile time");
    constexpr int PA_W = A_PIECES / WAVES, P = PIECES / WAVES;       // this wave's A pieces / all pieces per stage
    extern __shared__ __align__(16) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WAVES_M, wk = wave / WAVES_M;
    const int tiles_n = N / BN;
    const int nwg = gridDim.x, qq = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int tile = (xcd < r8 ? xcd * (qq + 1) : r8 * (qq + 1) + (xcd - r8) * qq) + (blockIdx.x >> 3);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int KG = K / 8, nk = K / BK;
    const v4i rsA = make_rsrc(A, (unsigned)((size_t)M * K * 4)), rsB = make_rsrc(W3, (unsigned)((size_t)N * KG * 48));
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char *)lds;

    // this wave's DMA pieces: q = WAVES * i + wave; piece i is an A piece for i < PA_W
    unsigned voff[P], ldst[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int q = WAVES * i + wave;
        if (i < PA_W) {
            const int row = q * A_RPP + lane / A_SLOTS, phys = lane % A_SLOTS, logical = phys ^ ((row / A_RPL) & (A_SLOTS - 1));
            voff[i] = (unsigned)(((size_t)(m0 + row) * K) * 4 + logical * 16);
            ldst[i] = q * 1024;
        } else {
            const int qb = q - A_PIECES, plane = qb / (BN / B_RPP), rb = qb % (BN / B_RPP);
            const int row = rb * B_RPP + lane / B_SLOTS, phys = lane % B_SLOTS, logical = phys ^ ((row / B_RPL) & (B_SLOTS - 1));
            voff[i] = (unsigned)(((size_t)(n0 + row) * KG + logical) * 48 + plane * 16);
            ldst[i] = A_BYTES + plane * B_PLANE + rb * 1024;
        }
        ldst[i] = __builtin_amdgcn_readfirstlane(ldst[i] + lds0);
    }
    // piece i of stage kt into ring slot `slot`; a stage past the end is a null DMA (zero records: no traffic, zeros written)
    auto issue_piece = [&](int i, int kt, int slot) {
        v4i rs = i < PA_W ? rsA : rsB;
        if (kt >= nk) rs[2] = 0;
        dma16(voff[i], rs, (unsigned)(i < PA_W ? kt * BK * 4 : kt * (BK / 8) * 48), ldst[i] + slot * STAGE);
    };

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    // fragment addresses inside a stage: lane (h, r) holds k = 16*wk + 8h .. +7 of row r
    const int h = lane >> 5, r = lane & 31;
    unsigned a_addr[2], b_addr[TN];
    {
        const int row = wm * 32 + r, s0 = wk * 4 + 2 * h, sw = (row / A_RPL) & (A_SLOTS - 1);
        a_addr[0] = lds0 + row * A_ROW + ((s0 ^ sw) * 16);
        a_addr[1] = lds0 + row * A_ROW + (((s0 + 1) ^ sw) * 16);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = j * 32 + r, sl = wk * 2 + h, sw = (row / B_RPL) & (B_SLOTS - 1);
        b_addr[j] = lds0 + A_BYTES + row * B_ROW + ((sl ^ sw) * 16);
    }

    struct Frags { bf16x8 a[3], b[TN][3]; };
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};   // product q = A piece PA[q] x B piece PB[q]

    // One phase.  MFMA n = 4q + j (product q, accumulator j) runs on `cur`; the 14 LDS reads of the next stage go out in the
    // first gaps (A raw first), its A fragment is split behind MFMAs 4..23 (one pair per five MFMAs), this wave's DMA
    // pieces of stage `dma_kt` go out one per few gaps.  `soff` = LDS byte offset of the slot that is read.
    auto phase = [&](const Frags &cur, Frags &nxt, unsigned soff, int dma_kt, int dma_slot) {
        float4 raw0, raw1;
        unsigned hh[4], mm[4], ll[4];
        float t0, t1, r0, r1;
        auto MF = [&](int n) { mfma(acc[n & 3], cur.a[PA[n >> 2]], cur.b[n & 3][PB[n >> 2]]); };
        int dma_i = 0;
        auto DMA = [&]() { if (!(FLAGS & 2) && dma_i < P) issue_piece(dma_i, dma_kt, dma_slot); ++dma_i; };
        lds_read16<0>(raw0, a_addr[0] + soff);
        lds_read16<0>(raw1, a_addr[1] + soff);
        if (FLAGS & 4) { MF(0); MF(1); MF(2); MF(3); nxt = cur; } else {
        MF(0); lds_read16<0 * B_PLANE>(nxt.b[0][0], b_addr[0] + soff);
        MF(1); lds_read16<0 * B_PLANE>(nxt.b[1][0], b_addr[1] + soff);
        MF(2); lds_read16<0 * B_PLANE>(nxt.b[2][0], b_addr[2] + soff);
        MF(3); lds_read16<0 * B_PLANE>(nxt.b[3][0], b_addr[3] + soff); }
        DMA();
        if (FLAGS & 4) wait_lgkm<0>(); else wait_lgkm<4>();      // raw0, raw1 have landed (four younger reads may be out)
#define SPLIT_GROUP(N0, X0, X1, G, RD0, RD1)                                                              \
        if (FLAGS & 1) { MF(N0); MF(N0 + 1); if (!(FLAGS & 4)) { RD0; } MF(N0 + 2); MF(N0 + 3); if (!(FLAGS & 4)) { RD1; } MF(N0 + 4); DMA(); \
                         hh[G] = __builtin_bit_cast(unsigned, X0); mm[G] = hh[G]; ll[G] = hh[G]; } else {      \
        MF(N0);     v_cvt_pk(hh[G], X0, X1); v_lo_f32(t0, hh[G]); v_hi_f32(t1, hh[G]);                    \
        MF(N0 + 1); v_sub(r0, X0, t0); v_sub(r1, X1, t1); if (!(FLAGS & 4)) { RD0; }                       \
        MF(N0 + 2); v_cvt_pk(mm[G], r0, r1); v_lo_f32(t0, mm[G]); v_hi_f32(t1, mm[G]);                    \
        MF(N0 + 3); v_sub(r0, r0, t0); v_sub(r1, r1, t1); if (!(FLAGS & 4)) { RD1; }                       \
        MF(N0 + 4); v_cvt_pk(ll[G], r0, r1); DMA(); }
        SPLIT_GROUP(4, raw0.x, raw0.y, 0, lds_read16<2 * B_PLANE>(nxt.b[0][2], b_addr[0] + soff), lds_read16<2 * B_PLANE>(nxt.b[1][2], b_addr[1] + soff))
        SPLIT_GROUP(9, raw0.z, raw0.w, 1, lds_read16<2 * B_PLANE>(nxt.b[2][2], b_addr[2] + soff), lds_read16<2 * B_PLANE>(nxt.b[3][2], b_addr[3] + soff))
        SPLIT_GROUP(14, raw1.x, raw1.y, 2, lds_read16<1 * B_PLANE>(nxt.b[0][1], b_addr[0] + soff), lds_read16<1 * B_PLANE>(nxt.b[1][1], b_addr[1] + soff))
        SPLIT_GROUP(19, raw1.z, raw1.w, 3, lds_read16<1 * B_PLANE>(nxt.b[2][1], b_addr[2] + soff), lds_read16<1 * B_PLANE>(nxt.b[3][1], b_addr[3] + soff))
#undef SPLIT_GROUP
#pragma unroll
        for (; dma_i < P;) DMA();
        nxt.a[0] = __builtin_bit_cast(bf16x8, (u32x4{hh[0], hh[1], hh[2], hh[3]}));
        nxt.a[1] = __builtin_bit_cast(bf16x8, (u32x4{mm[0], mm[1], mm[2], mm[3]}));
        nxt.a[2] = __builtin_bit_cast(bf16x8, (u32x4{ll[0], ll[1], ll[2], ll[3]}));
    };

    long long t_wait = 0, t_prev = 0, t_comp = 0;
    auto turn = [&]() {                                      // next stage visible; the slot of the stage now in registers is free
        long long t0 = 0;
        if (STAMP) { t0 = __builtin_amdgcn_s_memtime(); if (t_prev) t_comp += t0 - t_prev; }
        if (!(FLAGS & 2)) wait_vm<(S - 2) * P>();
        if (FLAGS & 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // also: VALU-written fragments are long done
        if (STAMP) { t_prev = __builtin_amdgcn_s_memtime(); t_wait += t_prev - t0; }
    };
    // prologue: every ring slot filled, stage 0 visible, its fragments in X
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int i = 0; i < P; ++i) issue_piece(i, s, s);
    wait_vm<(S - 1) * P>();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    Frags X, Y;
    {
        const float4 raw0 = *reinterpret_cast<const float4 *>(lds + (a_addr[0] - lds0));
        const float4 raw1 = *reinterpret_cast<const float4 *>(lds + (a_addr[1] - lds0));
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) X.b[j][p] = *reinterpret_cast<const bf16x8 *>(lds + (b_addr[j] - lds0) + p * B_PLANE);
        unsigned hh[4], mm[4], ll[4];
        split3_pair(f32x2{raw0.x, raw0.y}, hh[0], mm[0], ll[0]);
        split3_pair(f32x2{raw0.z, raw0.w}, hh[1], mm[1], ll[1]);
        split3_pair(f32x2{raw1.x, raw1.y}, hh[2], mm[2], ll[2]);
        split3_pair(f32x2{raw1.z, raw1.w}, hh[3], mm[3], ll[3]);
        X.a[0] = __builtin_bit_cast(bf16x8, (u32x4{hh[0], hh[1], hh[2], hh[3]}));
        X.a[1] = __builtin_bit_cast(bf16x8, (u32x4{mm[0], mm[1], mm[2], mm[3]}));
        X.a[2] = __builtin_bit_cast(bf16x8, (u32x4{ll[0], ll[1], ll[2], ll[3]}));
        // make the compiler finish its own LDS reads here: in the loop it must have nothing pending (else it puts a
        // conservative lgkmcnt(0) in front of the first MFMA of every iteration)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p) asm volatile("" : "+v"(X.b[j][p]));
    }
    // phase p: MFMAs of stage p, reads of stage p+1 (slot (p+1) % S), DMA of stage p+S into slot p % S
    int slot = 0;
    auto next = [&](int s) { return s + 1 == S ? 0 : s + 1; };
    int p = 0;
    for (; p + 1 < nk; p += 2) {
        turn();
        const int s1 = next(slot), s2 = next(s1);
        phase(X, Y, (unsigned)(s1 * STAGE), p + S, slot);
        turn();
        phase(Y, X, (unsigned)(s2 * STAGE), p + 1 + S, s1);
        slot = s2;
    }
    if (p < nk) {                                            // odd count: one more phase; its reads fetch a null stage
        turn();
        phase(X, Y, (unsigned)(next(slot) * STAGE), p + S, slot);
    }
    wait_vm<0>();
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // asm-issued MFMA results -> epilogue reads
    if (STAMP && tid == 0) {
        stamps[blockIdx.x * 4 + 0] = t_wait; stamps[blockIdx.x * 4 + 1] = 0; stamps[blockIdx.x * 4 + 2] = t_comp;
        stamps[blockIdx.x * 4 + 3] = nk - 1;
    }
    if (WAVES_K == 2) {                                      // add the two K halves: waves wk = 1 hand their tile over through LDS
        __syncthreads();
        float *part = reinterpret_cast<float *>(lds) + (wm * 64 + lane) * 68;    // 64 floats per lane, padded
        if (wk == 1) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) part[j * 16 + e] = acc[j][e];
        }
        __syncthreads();
        if (wk == 1) return;
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] += part[j * 16 + e];
    }
    // C layout of the 32x32 MFMA: lane (h, r): column r, rows 8*g + 4*h + e for acc[4*g + e]
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int m = m0 + wm * 32 + 8 * g + 4 * h + e, n = n0 + j * 32 + r;
                C[(size_t)m * N + n] = acc[j][4 * g + e];
            }
}

static void split3_host(float x, unsigned short &h, unsigned short &m, unsigned short &l) {
    auto rne = [](float v) { unsigned b; memcpy(&b, &v, 4); b += 0x7fffu + ((b >> 16) & 1); return (unsigned short)(b >> 16); };
    auto up = [](unsigned short s) { unsigned b = (unsigned)s << 16; float f; memcpy(&f, &b, 4); return f; };
    h = rne(x); const float r1 = x - up(h); m = rne(r1); const float r2 = r1 - up(m); l = rne(r2);
}

template <int BM, int BK, int WAVES_K, int S, int MINB, int FLAGS = 0>
static void run(const char *name, int M, int N, int K, const float *dA, const unsigned short *dW3, float *dC,
                const std::vector<float> &hA, const std::vector<float> &hB) {
    constexpr int STAGE = BM * BK * 4 + 3 * 128 * BK * 2;
    if (M % BM || N % 128 || K % BK) { printf("%-30s skipped (shape)\n", name); return; }
    const int lds_bytes = S * STAGE, threads = (BM / 32) * WAVES_K * 64, grid = (M / BM) * (N / 128);
    auto kern = gemm_dma<BM, BK, WAVES_K, S, MINB, false, FLAGS>;
    auto kern_s = gemm_dma<BM, BK, WAVES_K, S, MINB, true, FLAGS>;
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    (void)hipFuncSetAttribute((const void *)kern_s, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) kern<<<grid, threads, lds_bytes>>>(dA, dW3, dC, M, N, K, nullptr);
    const int reps = 20;
    (void)hipEventRecord(e0);
    for (int w = 0; w < reps; ++w) kern<<<grid, threads, lds_bytes>>>(dA, dW3, dC, M, N, K, nullptr);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    std::vector<float> hC((size_t)M * N);
    (void)hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int t = 0; t < 4000; ++t) {
        const int m = (int)((1103515245u * (unsigned)(t + 1) + 12345u) % (unsigned)M), n = (int)((2654435761u * (unsigned)(t + 7)) % (unsigned)N);
        double ref = 0, mag = 0;
        for (int k = 0; k < K; ++k) { const double pr = (double)hA[(size_t)m * K + k] * hB[(size_t)n * K + k]; ref += pr; mag += fabs(pr); }
        const double err = fabs(hC[(size_t)m * N + n] - ref) / mag;
        if (err > worst) worst = err;
    }
    long long *dS; (void)hipMalloc(&dS, (size_t)grid * 4 * 8);
    kern_s<<<grid, threads, lds_bytes>>>(dA, dW3, dC, M, N, K, dS);
    std::vector<long long> hS((size_t)grid * 4);
    (void)hipMemcpy(hS.data(), dS, hS.size() * 8, hipMemcpyDeviceToHost); (void)hipFree(dS);
    double sw = 0, sc = 0, sn = 0;
    for (int g = 0; g < grid; ++g) { sw += hS[4 * g]; sc += hS[4 * g + 2]; sn += hS[4 * g + 3]; }
    printf("%-30s M=%6d N=%5d K=%5d grid %5d lds %6d : %8.1f us  %7.1f TF/s-eq  err %.1e | cycles per phase (24 MFMA = 768): wait+barrier %4.0f  phase %5.0f\n",
           name, M, N, K, grid, lds_bytes, ms * 1e3, 2.0 * M * N * K / ms * 1e-9, worst, sw / sn, sc / sn);
}

int main(int argc, char **argv) {
    struct Shape { int M, N, K; };
    const Shape shapes[] = {{65536, 256, 2304}, {4224, 256, 2304}};
    for (const Shape &sh : shapes) {
        const int M = sh.M, N = sh.N, K = sh.K, KG = K / 8;
        std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
        unsigned s = 12345u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
        for (auto &v : hA) v = rnd();
        for (auto &v : hB) v = rnd();
        std::vector<unsigned short> hW3((size_t)N * KG * 24);
        for (int n = 0; n < N; ++n)
            for (int k = 0; k < K; ++k) {
                unsigned short hh, mm, ll; split3_host(hB[(size_t)n * K + k], hh, mm, ll);
                const size_t base = ((size_t)n * KG + k / 8) * 24 + (k & 7);
                hW3[base] = hh; hW3[base + 8] = mm; hW3[base + 16] = ll;
            }
        float *dA, *dC; unsigned short *dW3;
        (void)hipMalloc(&dA, hA.size() * 4); (void)hipMalloc(&dW3, hW3.size() * 2); (void)hipMalloc(&dC, (size_t)M * N * 4);
        (void)hipMemcpy(dA, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(dW3, hW3.data(), hW3.size() * 2, hipMemcpyHostToDevice);
        run<128, 16, 1, 4, 2>("128x128 BK16 S4 (2/CU)", M, N, K, dA, dW3, dC, hA, hB);
        run<128, 16, 1, 4, 2, 1>("  - no split VALU", M, N, K, dA, dW3, dC, hA, hB);
        run<128, 16, 1, 4, 2, 2>("  - no DMA", M, N, K, dA, dW3, dC, hA, hB);
        run<128, 16, 1, 4, 2, 4>("  - no B reads", M, N, K, dA, dW3, dC, hA, hB);
        run<128, 16, 1, 4, 2, 8>("  - no barrier", M, N, K, dA, dW3, dC, hA, hB);
        run<128, 16, 1, 4, 2, 7>("  - MFMA + A reads + barrier", M, N, K, dA, dW3, dC, hA, hB);
        run<128, 16, 1, 4, 2, 15>("  - MFMA + A reads only", M, N, K, dA, dW3, dC, hA, hB);
        run<128, 16, 1, 6, 1>("128x128 BK16 S6 (1/CU)", M, N, K, dA, dW3, dC, hA, hB);
        run<128, 16, 1, 6, 1, 1>("  - no split VALU", M, N, K, dA, dW3, dC, hA, hB);
        run<128, 16, 1, 6, 1, 2>("  - no DMA", M, N, K, dA, dW3, dC, hA, hB);
        run<128, 16, 1, 6, 1, 4>("  - no B reads", M, N, K, dA, dW3, dC, hA, hB);
        run<128, 16, 1, 6, 1, 8>("  - no barrier", M, N, K, dA, dW3, dC, hA, hB);
        run<128, 16, 1, 6, 1, 15>("  - MFMA + A reads only", M, N, K, dA, dW3, dC, hA, hB);
        (void)hipFree(dA); (void)hipFree(dW3); (void)hipFree(dC);
    }
    return 0;
}
