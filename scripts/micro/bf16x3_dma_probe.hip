// Probe for a bf16x3 GEMM loop fed entirely by LDS-DMA (`buffer_load_dwordx4 ... offen lds`), no VGPR staging, no ds_write:
//   A [M][K] f32 goes to LDS RAW (128-B rows, 16-B slots XOR-swizzled through the per-lane SOURCE address) and every wave
//   splits the fragment it read into the three bf16 pieces in registers;
//   B is the pre-split weight image of the conv path ([N][K/8][hi|mid|lo][8] bf16) and lands in three unpadded LDS planes.
// An S-deep ring of stages, one barrier per K-step, counted vmcnt.  C[m][n] = sum_k A[m][k] B[n][k].
// Prints TFLOP/s (f32-equivalent) and the error against an f64 reference.
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
// one wave-instruction: 64 lanes x 16 B from per-lane source offsets to LDS [lds_dst, lds_dst + 1024)
__device__ __forceinline__ void dma16(unsigned voff, v4i rsrc, unsigned soff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
                 :: "v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

// x == hi + mid + lo exactly when |x| is not tiny (3 x 8 significand bits); pairs, so that v_cvt_pk_bf16_f32 is used
__device__ __forceinline__ void split3_pair(f32x2 x, unsigned &h, unsigned &m, unsigned &l) {
    const bf16x2 hb = __builtin_convertvector(x, bf16x2);
    const f32x2 r1 = x - __builtin_convertvector(hb, f32x2);
    const bf16x2 mb = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(mb, f32x2);
    const bf16x2 lb = __builtin_convertvector(r2, bf16x2);
    h = __builtin_bit_cast(unsigned, hb); m = __builtin_bit_cast(unsigned, mb); l = __builtin_bit_cast(unsigned, lb);
}

template <int BM, int BN, int WM, int WN, int S, int MINB, bool STAMP>
__global__ void __launch_bounds__((BM / WM) * (BN / WN) * 64, MINB)
gemm_dma(const float *__restrict__ A, const unsigned short *__restrict__ W3, float *__restrict__ C, int M, int N, int K,
         long long *__restrict__ stamps) {
    constexpr int BK = 32, WAVES = (BM / WM) * (BN / WN), TM = WM / 32, TN = WN / 32;
    constexpr int A_BYTES = BM * 128, B_PLANE = BN * 64, STAGE = A_BYTES + 3 * B_PLANE;
    constexpr int A_PIECES = BM / 8, B_PIECES = 3 * (BN / 16), PIECES = A_PIECES + B_PIECES;
    static_assert(PIECES % WAVES == 0, "pieces per wave");
    constexpr int P = PIECES / WAVES;                       // DMA instructions per wave per stage
    extern __shared__ __align__(16) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / (BN / WN), wn = wave % (BN / WN);
    const int tiles_n = N / BN;
    const int nwg = gridDim.x, qq = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int tile = (xcd < r8 ? xcd * (qq + 1) : r8 * (qq + 1) + (xcd - r8) * qq) + (blockIdx.x >> 3);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int KG = K / 8;
    const v4i rsA = make_rsrc(A, (unsigned)((size_t)M * K * 4)), rsB = make_rsrc(W3, (unsigned)((size_t)N * KG * 48));
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char *)lds;

    // this wave's pieces: q = WAVES * i + wave, so that piece i is an A piece for i < A_PIECES / WAVES at compile time
    static_assert(A_PIECES % WAVES == 0 && B_PIECES % WAVES == 0, "piece kinds per wave");
    constexpr int PA_W = A_PIECES / WAVES;
    unsigned voff[P], ldst[P];
#pragma unroll
    for (int i = 0; i < P; ++i) {
        const int q = WAVES * i + wave;
        if (i < PA_W) {
            const int row = q * 8 + (lane >> 3), phys = lane & 7, logical = phys ^ ((row >> 1) & 7);
            voff[i] = (unsigned)(((size_t)(m0 + row) * K) * 4 + logical * 16);
            ldst[i] = q * 1024;
        } else {
            const int qb = q - A_PIECES, plane = qb / (BN / 16), rb = qb % (BN / 16);
            const int row = rb * 16 + (lane >> 2), phys = lane & 3, logical = phys ^ ((row >> 2) & 3);
            voff[i] = (unsigned)(((size_t)(n0 + row) * KG + logical) * 48 + plane * 16);
            ldst[i] = A_BYTES + plane * B_PLANE + rb * 1024;
        }
        ldst[i] = __builtin_amdgcn_readfirstlane(ldst[i] + lds0);
    }
    const int nk = K / BK;
    auto issue = [&](int kt, int slot) {                    // stage kt into ring slot `slot`; a stage past the end is a null DMA
        v4i ra = rsA, rb = rsB;
        if (kt >= nk) { ra[2] = 0; rb[2] = 0; }
#pragma unroll
        for (int i = 0; i < P; ++i) {
            if (i < PA_W) dma16(voff[i], ra, (unsigned)(kt * BK * 4), ldst[i] + slot * STAGE);
            else dma16(voff[i], rb, (unsigned)(kt * (BK / 8) * 48), ldst[i] + slot * STAGE);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int h = lane >> 5, r = lane & 31;
    int a_off[TM], a_swz[TM], b_off[TN], b_swz[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) { const int row = wm * WM + i * 32 + r; a_off[i] = row * 128; a_swz[i] = (row >> 1) & 7; }
#pragma unroll
    for (int j = 0; j < TN; ++j) { const int row = wn * WN + j * 32 + r; b_off[j] = A_BYTES + row * 64; b_swz[j] = (row >> 2) & 3; }

#pragma unroll
    for (int s = 0; s < S - 1; ++s) issue(s, s);

    // software pipeline over 16-k chunks: a phase issues the LDS reads of chunk x+1, runs the MFMAs of chunk x out of
    // registers and splits chunk x+1's A fragment in their shadow.  Two register sets (X, Y) alternate, no copies.
    struct Frags { bf16x8 a[TM][3], b[TN][3]; };
    auto read_chunk = [&](const unsigned char *st, int c, float4 (&raw)[TM][2], Frags &f) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int s0 = c * 4 + 2 * h;
            raw[i][0] = *reinterpret_cast<const float4 *>(st + a_off[i] + ((s0 ^ a_swz[i]) * 16));
            raw[i][1] = *reinterpret_cast<const float4 *>(st + a_off[i] + (((s0 + 1) ^ a_swz[i]) * 16));
        }
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int p = 0; p < 3; ++p)
                f.b[j][p] = *reinterpret_cast<const bf16x8 *>(st + b_off[j] + p * B_PLANE + (((c * 2 + h) ^ b_swz[j]) * 16));
    };
    auto split_pair_of = [&](const float4 (&raw)[TM][2], int g, unsigned (&hh)[4 * TM], unsigned (&mm)[4 * TM], unsigned (&ll)[4 * TM]) {
        const int i = g >> 2, e = g & 3;                    // pair e of fragment i: elements 2e, 2e+1 of the 8
        const float4 v = raw[i][e >> 1];
        split3_pair((e & 1) ? f32x2{v.z, v.w} : f32x2{v.x, v.y}, hh[g], mm[g], ll[g]);
    };
    auto pack_frags = [&](Frags &f, const unsigned (&hh)[4 * TM], const unsigned (&mm)[4 * TM], const unsigned (&ll)[4 * TM]) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            f.a[i][0] = __builtin_bit_cast(bf16x8, (u32x4{hh[4 * i], hh[4 * i + 1], hh[4 * i + 2], hh[4 * i + 3]}));
            f.a[i][1] = __builtin_bit_cast(bf16x8, (u32x4{mm[4 * i], mm[4 * i + 1], mm[4 * i + 2], mm[4 * i + 3]}));
            f.a[i][2] = __builtin_bit_cast(bf16x8, (u32x4{ll[4 * i], ll[4 * i + 1], ll[4 * i + 2], ll[4 * i + 3]}));
        }
    };
    constexpr int NMFMA = 6 * TM * TN, NP = 4 * TM;
    auto mfma_n = [&](const Frags &f, int n) {
        constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
        const int q = n / (TM * TN), i = (n / TN) % TM, j = n % TN;
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[i][PA[q]], f.b[j][PB[q]], acc[i][j], 0, 0, 0);
    };
    // one phase: LDS reads of the next chunk, then NP groups of {this chunk's MFMAs, one pair of the next chunk's split}
    auto phase = [&](const Frags &cur, Frags &nxt, const unsigned char *st, int c) {
        float4 raw[TM][2];
        read_chunk(st, c, raw, nxt);
        __builtin_amdgcn_sched_barrier(0);
        unsigned hh[NP], mm[NP], ll[NP];
#pragma unroll
        for (int g = 0; g < NP; ++g) {
#pragma unroll
            for (int n = g * NMFMA / NP; n < (g + 1) * NMFMA / NP; ++n) mfma_n(cur, n);
            split_pair_of(raw, g, hh, mm, ll);
            __builtin_amdgcn_sched_barrier(0);
        }
        pack_frags(nxt, hh, mm, ll);
    };
    int slot_rd = 0, slot_wr = S - 1;                       // ring slots of the stage being read / the next one to fill
    long long t_wait = 0, t_issue = 0, t_comp = 0, t_prev = 0;
    auto turn = [&](int kt_next_issue) {                     // next stage visible, the finished stage's slot refilled
        long long t0 = 0, t1 = 0;
        if (STAMP) { t0 = __builtin_amdgcn_s_memtime(); if (t_prev) t_comp += t0 - t_prev; }
        wait_vm<(S - 2) * P>();
        __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0), as a builtin so that the compiler's scoreboard knows
        __builtin_amdgcn_s_barrier();
        if (STAMP) { t1 = __builtin_amdgcn_s_memtime(); t_wait += t1 - t0; }
        issue(kt_next_issue, slot_wr);
        slot_wr = slot_wr + 1 == S ? 0 : slot_wr + 1;
        if (STAMP) { t_prev = __builtin_amdgcn_s_memtime(); t_issue += t_prev - t1; }
    };
    Frags X, Y;
    turn(S - 1);                                            // stage 0 visible
    {
        float4 raw[TM][2];
        unsigned hh[NP], mm[NP], ll[NP];
        read_chunk(lds, 0, raw, X);
#pragma unroll
        for (int g = 0; g < NP; ++g) split_pair_of(raw, g, hh, mm, ll);
        pack_frags(X, hh, mm, ll);
    }
    phase(X, Y, lds, 1);
    for (int kt = 0; kt < nk - 1; ++kt) {                   // back edge right behind the turn: nothing the compiler must guess
        turn(kt + S);
        slot_rd = slot_rd + 1 == S ? 0 : slot_rd + 1;
        phase(Y, X, lds + slot_rd * STAGE, 0);
        phase(X, Y, lds + slot_rd * STAGE, 1);
    }
#pragma unroll
    for (int n = 0; n < NMFMA; ++n) mfma_n(Y, n);
    wait_vm<0>();
    if (STAMP && tid == 0) {
        stamps[blockIdx.x * 4 + 0] = t_wait; stamps[blockIdx.x * 4 + 1] = t_issue; stamps[blockIdx.x * 4 + 2] = t_comp;
        stamps[blockIdx.x * 4 + 3] = nk - 1;
    }
    // C layout of the 32x32 MFMA: lane (h, r): column r, rows 8*g + 4*h + e for acc[4*g + e]
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int m = m0 + wm * WM + i * 32 + 8 * g + 4 * h + e, n = n0 + wn * WN + j * 32 + r;
                    C[(size_t)m * N + n] = acc[i][j][4 * g + e];
                }
}

static void split3_host(float x, unsigned short &h, unsigned short &m, unsigned short &l) {
    auto rne = [](float v) { unsigned b; memcpy(&b, &v, 4); b += 0x7fffu + ((b >> 16) & 1); return (unsigned short)(b >> 16); };
    auto up = [](unsigned short s) { unsigned b = (unsigned)s << 16; float f; memcpy(&f, &b, 4); return f; };
    h = rne(x); const float r1 = x - up(h); m = rne(r1); const float r2 = r1 - up(m); l = rne(r2);
}

template <int BM, int BN, int WM, int WN, int S, int MINB>
static void run(const char *name, int M, int N, int K, const float *dA, const unsigned short *dW3, float *dC,
                const std::vector<float> &hA, const std::vector<float> &hB) {
    constexpr int STAGE = BM * 128 + 3 * BN * 64;
    const int lds_bytes = S * STAGE, threads = (BM / WM) * (BN / WN) * 64, grid = (M / BM) * (N / BN);
    auto kern = gemm_dma<BM, BN, WM, WN, S, MINB, false>;
    auto kern_s = gemm_dma<BM, BN, WM, WN, S, MINB, true>;
    (void)hipFuncSetAttribute((const void *)kern_s, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) kern<<<grid, threads, lds_bytes>>>(dA, dW3, dC, M, N, K, nullptr);
    const int reps = 20;
    (void)hipEventRecord(e0);
    for (int w = 0; w < reps; ++w) kern<<<grid, threads, lds_bytes>>>(dA, dW3, dC, M, N, K, nullptr);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    std::vector<float> hC((size_t)M * N);
    (void)hipMemcpy(hC.data(), dC, hC.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0, scale = 0;
    for (int t = 0; t < 4000; ++t) {
        const int m = (int)((1103515245u * (unsigned)(t + 1) + 12345u) % (unsigned)M), n = (int)((2654435761u * (unsigned)(t + 7)) % (unsigned)N);
        double ref = 0, mag = 0;
        for (int k = 0; k < K; ++k) { const double p = (double)hA[(size_t)m * K + k] * hB[(size_t)n * K + k]; ref += p; mag += fabs(p); }
        const double err = fabs(hC[(size_t)m * N + n] - ref) / mag;
        if (err > worst) worst = err;
        scale += 1;
    }
    long long *dS; (void)hipMalloc(&dS, (size_t)grid * 4 * 8);
    kern_s<<<grid, threads, lds_bytes>>>(dA, dW3, dC, M, N, K, dS);
    std::vector<long long> hS((size_t)grid * 4);
    (void)hipMemcpy(hS.data(), dS, hS.size() * 8, hipMemcpyDeviceToHost); (void)hipFree(dS);
    double sw = 0, si = 0, sc = 0, sn = 0;
    for (int g = 0; g < grid; ++g) { sw += hS[4 * g]; si += hS[4 * g + 1]; sc += hS[4 * g + 2]; sn += hS[4 * g + 3]; }
    printf("%-26s M=%6d N=%5d K=%5d grid %5d lds %6d : %8.1f us  %7.1f TF/s-eq   max err/|terms| %.2e  | stamped cycles per K-step: wait+barrier %5.0f  issue %4.0f  phases %5.0f\n", name, M, N, K, grid, lds_bytes,
           ms * 1e3, 2.0 * M * N * K / ms * 1e-9, worst, sw / sn, si / sn, sc / sn);
}

int main(int argc, char **argv) {
    struct Shape { int M, N, K; };
    const Shape shapes[] = {{67200 / 128 * 128, 256, 2304}, {16768, 2048, 512}, {4224, 256, 2304}, {4224, 1024, 256}, {1024, 512, 4608}};
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
        run<64, 128, 32, 64, 2, 2>("64x128 w32x64 S2", M, N, K, dA, dW3, dC, hA, hB);
        run<64, 128, 32, 64, 3, 1>("64x128 w32x64 S3", M, N, K, dA, dW3, dC, hA, hB);
        run<64, 128, 32, 64, 4, 1>("64x128 w32x64 S4", M, N, K, dA, dW3, dC, hA, hB);
        run<128, 128, 64, 64, 2, 1>("128x128 w64x64 S2", M, N, K, dA, dW3, dC, hA, hB);
        run<128, 128, 64, 64, 3, 1>("128x128 w64x64 S3", M, N, K, dA, dW3, dC, hA, hB);
        run<128, 64, 32, 64, 3, 1>("128x64 w32x64(4x1) S3", M, N, K, dA, dW3, dC, hA, hB);
        run<128, 64, 32, 64, 2, 2>("128x64 w32x64(4x1) S2", M, N, K, dA, dW3, dC, hA, hB);
        run<64, 64, 32, 32, 3, 2>("64x64 w32x32 S3", M, N, K, dA, dW3, dC, hA, hB);
        run<64, 64, 32, 32, 4, 2>("64x64 w32x32 S4", M, N, K, dA, dW3, dC, hA, hB);
        (void)hipFree(dA); (void)hipFree(dW3); (void)hipFree(dC);
    }
    return 0;
}
