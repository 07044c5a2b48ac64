// Where does the f32-MFMA K loop lose time?  A stripped 64x64-tile GEMM (C[m][n] = sum_k A[m][k] B[n][k], both row x
// contiguous K like the conv kernel's operands) with switches that remove one ingredient at a time:
//   bit 0: no global loads in the loop (registers keep the first K-step)
//   bit 1: no LDS writes in the loop (LDS keeps the first K-step; the barrier pair goes too)
//   bit 2: no LDS fragment reads in the loop (fragments stay in registers)
//   bit 3: no barriers around the LDS writes (racy on purpose: prices the barriers alone)
//   bit 4: the B operand is not written to LDS in the loop (upper bound for feeding B fragments straight from global memory)
// WAVES = 1 (one wave owns the 64x64 tile) or 4 (four waves, 32x32 each).  Results are meaningless with a switch on;
// only the timing matters.  Build: hipcc -O3 --offload-arch=gfx950 gemm_loop_probe.hip -o probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 64, BN = 64, BK = 32, LDK = BK + 4;

template <int WAVES, int FLAGS>
__global__ void __launch_bounds__(64 * WAVES, WAVES == 1 ? 2 : 6)
probe(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, int M, int N, int K,
      long long *__restrict__ stamps = nullptr) {
    long long c0 = 0, r0s = 0;
    if (stamps && threadIdx.x == 0) { c0 = __builtin_amdgcn_s_memtime(); r0s = __builtin_amdgcn_s_memrealtime(); }
    constexpr int THREADS = 64 * WAVES, TPR = BK / 4, RPP = THREADS / TPR, ROWS = BM / RPP;
    constexpr int WM = WAVES == 1 ? 64 : 32, WN = WM, TM = WM / 32, TN = WN / 32;
    __shared__ __align__(16) float smem[(BM + BN) * LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = WAVES == 1 ? 0 : wave >> 1, wn = WAVES == 1 ? 0 : wave & 1;
    const int tiles_n = N / BN;
    const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
    const int c4 = (tid % TPR) * 4, r0 = tid / TPR;
    const float *ap[ROWS], *bp[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        ap[i] = A + (long)(m0 + r0 + RPP * i) * K + c4;
        bp[i] = B + (long)(n0 + r0 + RPP * i) * K + c4;
    }
    float4 ra[ROWS], rb[ROWS];
    auto load_global = [&](int k) {
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            ra[i] = *reinterpret_cast<const float4 *>(ap[i] + k);
            rb[i] = *reinterpret_cast<const float4 *>(bp[i] + k);
        }
    };
    auto store_lds = [&]() {
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            *reinterpret_cast<float4 *>(smem + (r0 + RPP * i) * LDK + c4) = ra[i];
            if (!(FLAGS & 16)) *reinterpret_cast<float4 *>(smem + BM * LDK + (r0 + RPP * i) * LDK + c4) = rb[i];
        }
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const float *As = smem + (wm * WM + (lane & 31)) * LDK + 4 * (lane >> 5);
    const float *Bs = smem + BM * LDK + (wn * WN + (lane & 31)) * LDK + 4 * (lane >> 5);

    load_global(0);
    store_lds();
    __syncthreads();
    float4 fa[2][TM], fb[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[0][i] = fa[1][i] = *reinterpret_cast<const float4 *>(As + i * 32 * LDK);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[0][j] = fb[1][j] = *reinterpret_cast<const float4 *>(Bs + j * 32 * LDK);
    const int nk = K / BK;
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (!(FLAGS & 4)) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const float4 *>(As + i * 32 * LDK);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const float4 *>(Bs + j * 32 * LDK);
        }
        if (more && !(FLAGS & 1)) load_global((kt + 1) * BK);
#pragma unroll
        for (int ks = 0; ks < BK / 8; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < BK / 8 && !(FLAGS & 4)) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[nxt][i] = *reinterpret_cast<const float4 *>(As + i * 32 * LDK + (ks + 1) * 8);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[nxt][j] = *reinterpret_cast<const float4 *>(Bs + j * 32 * LDK + (ks + 1) * 8);
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
        }
        if (!(FLAGS & 2)) {
            if (!(FLAGS & 8)) __syncthreads();
            if (more) store_lds();
            if (!(FLAGS & 8)) __syncthreads();
        }
    }
    if (stamps && threadIdx.x == 0) {       // in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz
        stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
        stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0s;
    }
    // plain column-per-lane store (not part of the question)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * WM + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const int col = n0 + wn * WN + j * 32 + (lane & 31);
                C[(long)row * N + col] = acc[i][j][e];
            }
}


// The same tile fed by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write): NBUF LDS buffers of
// unpadded 128-byte rows, XOR-swizzled through the per-lane SOURCE address (a wave-instruction writes 1 KiB = 8 rows
// lane-linearly), one barrier per K-step, the DMA of step k+NBUF-1 in flight while step k computes.
template <int NBUF>
__global__ void __launch_bounds__(256)
probe_glds(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, int M, int N, int K) {
    constexpr int STAGE = (BM + BN) * BK;                 // floats per buffer, no padding
    __shared__ __align__(1024) float smem[NBUF * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = N / BN;
    // XCD remap (blocks b, b+8, ... share an L2): each XCD walks a contiguous run of tiles
    const int nwg = gridDim.x, qq = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int tile = (xcd < r8 ? xcd * (qq + 1) : r8 * (qq + 1) + (xcd - r8) * qq) + (blockIdx.x >> 3);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    // this wave's 4 DMA pieces per K-step: pieces 2*wave, 2*wave+1 of A (8 rows each) and the same of B
    const float *src[4];
    int dst_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool isB = i >= 2;
        const int piece = 2 * wave + (i & 1);
        const int row = piece * 8 + (lane >> 3);           // tile row this lane's 16 bytes belong to
        const int q = (lane & 7) ^ ((row >> 1) & 7);       // source chunk that must land at LDS chunk position lane&7
        src[i] = (isB ? B + (long)(n0 + row) * K : A + (long)(m0 + row) * K) + 4 * q;
        dst_off[i] = (isB ? BM * BK : 0) + piece * 8 * BK;
    }
    auto issue = [&](int kt, int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src[i] + kt * BK),
                                             (__attribute__((address_space(3))) void *)(smem + buf * STAGE + dst_off[i]), 16, 0, 0);
    };
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const int ra = wm * 32 + (lane & 31), rb = wn * 32 + (lane & 31), h = lane >> 5;
    const int nk = K / BK;
#pragma unroll
    for (int s = 0; s < NBUF - 1; ++s)
        if (s < nk) issue(s, s);
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt % NBUF;
        // retire the DMA of step kt (leave the NBUF-2 younger steps in flight), then publish it
        if (NBUF == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (NBUF == 3) { if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        else { if (kt + 2 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();
        if (kt + NBUF - 1 < nk) issue(kt + NBUF - 1, (kt + NBUF - 1) % NBUF);   // overwrites the buffer read in step kt-1
        const float *As = smem + buf * STAGE, *Bs = As + BM * BK;
        float4 fa[4], fb[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            fa[ks] = *reinterpret_cast<const float4 *>(As + ra * BK + 4 * ((2 * ks + h) ^ ((ra >> 1) & 7)));
            fb[ks] = *reinterpret_cast<const float4 *>(Bs + rb * BK + 4 * ((2 * ks + h) ^ ((rb >> 1) & 7)));
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ks].x, fb[ks].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ks].y, fb[ks].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ks].z, fb[ks].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ks].w, fb[ks].w, acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        const int col = n0 + wn * 32 + (lane & 31);
        C[(long)row * N + col] = acc[e];
    }
}

template <int NBUF>
void run_glds(const float *A, const float *B, float *C, int M, int N, int K) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = (M / BM) * (N / BN);
    probe_glds<NBUF><<<grid, 256>>>(A, B, C, M, N, K);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) probe_glds<NBUF><<<grid, 256>>>(A, B, C, M, N, K);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("glds, %d LDS buffers: %7.3f ms  %6.1f TF/s\n", NBUF, ms, 2.0 * M * N * K / ms / 1e9);
}

// B fragments straight from global memory: B is pre-packed in MFMA fragment order,
//   Bf[n/32][k/32][ks (4)][lane (64)][4] = B[n = 32*(n/32) + lane%32][k = 32*(k/32) + 8*ks + 4*(lane/32) + 0..3],
// so one wave-load is 1 KiB contiguous and B never touches LDS (no ds_write, no ds_read for it); A is staged through a
// single LDS stage as in probe<4,0>.  FLAGS bit 0: no global loads for A in the loop (prices the B stream alone).
template <int FLAGS>
__global__ void __launch_bounds__(256, 6)
probe_bdirect(const float *__restrict__ A, const float *__restrict__ Bf, float *__restrict__ C, int M, int N, int K) {
    constexpr int TPR = BK / 4, RPP = 256 / TPR, ROWS = BM / RPP;
    __shared__ __align__(16) float smem[BM * LDK];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = N / BN;
    const int nwg = gridDim.x, qq = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int tile = (xcd < r8 ? xcd * (qq + 1) : r8 * (qq + 1) + (xcd - r8) * qq) + (blockIdx.x >> 3);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int c4 = (tid % TPR) * 4, r0 = tid / TPR;
    const float *ap[ROWS];
#pragma unroll
    for (int i = 0; i < ROWS; ++i) ap[i] = A + (long)(m0 + r0 + RPP * i) * K + c4;
    const int nk = K / BK;
    // this wave's B stream: block row (n0 + wn*32)/32, K-steps consecutive, 4 KiB each
    const float4 *bf = reinterpret_cast<const float4 *>(Bf) + ((long)((n0 >> 5) + wn) * nk) * 256 + lane;
    float4 ra[ROWS];
    auto load_a = [&](int k) {
#pragma unroll
        for (int i = 0; i < ROWS; ++i) ra[i] = *reinterpret_cast<const float4 *>(ap[i] + k);
    };
    auto store_a = [&]() {
#pragma unroll
        for (int i = 0; i < ROWS; ++i) *reinterpret_cast<float4 *>(smem + (r0 + RPP * i) * LDK + c4) = ra[i];
    };
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const float *As = smem + (wm * 32 + (lane & 31)) * LDK + 4 * (lane >> 5);
    float4 fb[2][4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) fb[0][ks] = bf[ks * 64];
    load_a(0);
    store_a();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        const int cur = kt & 1, nxt = cur ^ 1;
        float4 fa[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) fa[ks] = *reinterpret_cast<const float4 *>(As + ks * 8);
        if (more) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) fb[nxt][ks] = bf[(long)(kt + 1) * 256 + ks * 64];
            if (!(FLAGS & 1)) load_a((kt + 1) * BK);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ks].x, fb[cur][ks].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ks].y, fb[cur][ks].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ks].z, fb[cur][ks].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ks].w, fb[cur][ks].w, acc, 0, 0, 0);
        }
        __syncthreads();
        if (more) store_a();
        __syncthreads();
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
        const int col = n0 + wn * 32 + (lane & 31);
        C[(long)row * N + col] = acc[e];
    }
}

template <int FLAGS>
void run_bdirect(const float *A, const float *Bf, float *C, int M, int N, int K) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = (M / BM) * (N / BN);
    probe_bdirect<FLAGS><<<grid, 256>>>(A, Bf, C, M, N, K);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) probe_bdirect<FLAGS><<<grid, 256>>>(A, Bf, C, M, N, K);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("B fragments from global, no-A-global %d: %7.3f ms  %6.1f TF/s\n", FLAGS & 1, ms, 2.0 * M * N * K / ms / 1e9);
}

template <int WAVES, int FLAGS>
void run(const float *A, const float *B, float *C, int M, int N, int K) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = (M / BM) * (N / BN);
    probe<WAVES, FLAGS><<<grid, 64 * WAVES>>>(A, B, C, M, N, K);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) probe<WAVES, FLAGS><<<grid, 64 * WAVES>>>(A, B, C, M, N, K);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("waves %d  no-global %d no-ldswrite %d no-ldsread %d no-barrier %d no-B-write %d : %7.3f ms  %6.1f TF/s\n", WAVES, FLAGS & 1,
           (FLAGS >> 1) & 1, (FLAGS >> 2) & 1, (FLAGS >> 3) & 1, (FLAGS >> 4) & 1, ms, 2.0 * M * N * K / ms / 1e9);
}

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 16384, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 1024;
    std::vector<float> h((size_t)M * K);
    srand(3);
    for (auto &v : h) v = (float)rand() / RAND_MAX - 0.5f;
    float *A, *B, *C;
    (void)hipMalloc(&A, (size_t)M * K * 4); (void)hipMalloc(&B, (size_t)N * K * 4); (void)hipMalloc(&C, (size_t)M * N * 4);
    (void)hipMemcpy(A, h.data(), (size_t)M * K * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(B, h.data(), (size_t)N * K * 4, hipMemcpyHostToDevice);
    printf("M %d N %d K %d (64x64 tiles: %d)\n", M, N, K, (M / 64) * (N / 64));
    run<1, 0>(A, B, C, M, N, K); run<1, 1>(A, B, C, M, N, K); run<1, 3>(A, B, C, M, N, K); run<1, 7>(A, B, C, M, N, K);
    run<1, 4>(A, B, C, M, N, K); run<1, 2>(A, B, C, M, N, K);
    run<4, 0>(A, B, C, M, N, K); run<4, 1>(A, B, C, M, N, K); run<4, 3>(A, B, C, M, N, K); run<4, 7>(A, B, C, M, N, K);
    run<4, 4>(A, B, C, M, N, K); run<4, 2>(A, B, C, M, N, K); run<4, 9>(A, B, C, M, N, K); run<4, 8>(A, B, C, M, N, K); run<4, 17>(A, B, C, M, N, K);
    run_glds<2>(A, B, C, M, N, K); run_glds<3>(A, B, C, M, N, K); run_glds<4>(A, B, C, M, N, K);
    {   // in-kernel clock of three loop bodies after ~1 s of back-to-back launches each
        const int grid = (M / BM) * (N / BN);
        long long *st;
        (void)hipMalloc(&st, (size_t)grid * 16);
        std::vector<long long> h((size_t)grid * 2);
        auto clock_of = [&](const char *what, auto launch) {
            for (int r = 0; r < 600; ++r) launch();
            (void)hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
            std::vector<double> c;
            for (int i = 0; i < grid; ++i) if (h[2 * i + 1] > 0) c.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100e6);
            std::sort(c.begin(), c.end());
            printf("in-kernel clock, %-46s %.2f GHz (median over %zu workgroups)\n", what, c[c.size() / 2] / 1e9, c.size());
        };
        clock_of("MFMA only (no global, no LDS write/read):", [&] { probe<4, 7><<<grid, 256>>>(A, B, C, M, N, K, st); });
        clock_of("MFMA + LDS fragment reads:", [&] { probe<4, 3><<<grid, 256>>>(A, B, C, M, N, K, st); });
        clock_of("MFMA + LDS reads + writes + barriers:", [&] { probe<4, 1><<<grid, 256>>>(A, B, C, M, N, K, st); });
        clock_of("everything (global loads too):", [&] { probe<4, 0><<<grid, 256>>>(A, B, C, M, N, K, st); });
    }
    {   // B in fragment order
        std::vector<float> hb((size_t)N * K), hf((size_t)N * K);
        (void)hipMemcpy(hb.data(), B, hb.size() * 4, hipMemcpyDeviceToHost);
        const int nkk = K / 32;
        for (int nb = 0; nb < N / 32; ++nb)
            for (int kt = 0; kt < nkk; ++kt)
                for (int ks = 0; ks < 4; ++ks)
                    for (int l = 0; l < 64; ++l)
                        for (int e = 0; e < 4; ++e)
                            hf[((((size_t)nb * nkk + kt) * 4 + ks) * 64 + l) * 4 + e] =
                                hb[(size_t)(nb * 32 + (l & 31)) * K + kt * 32 + ks * 8 + 4 * (l >> 5) + e];
        float *Bf;
        (void)hipMalloc(&Bf, hf.size() * 4);
        (void)hipMemcpy(Bf, hf.data(), hf.size() * 4, hipMemcpyHostToDevice);
        run_bdirect<0>(A, Bf, C, M, N, K); run_bdirect<1>(A, Bf, C, M, N, K);
        std::vector<float> c0((size_t)M * N), c1((size_t)M * N);
        probe<4, 0><<<(M / BM) * (N / BN), 256>>>(A, B, C, M, N, K);
        (void)hipMemcpy(c0.data(), C, c0.size() * 4, hipMemcpyDeviceToHost);
        (void)hipMemset(C, 0, c0.size() * 4);
        probe_bdirect<0><<<(M / BM) * (N / BN), 256>>>(A, Bf, C, M, N, K);
        (void)hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < c0.size(); ++i) bad += c0[i] != c1[i];
        printf("B-direct vs register-staged: %zu of %zu elements differ\n", bad, c0.size());
    }
    // the DMA path must reproduce the register-staged result bit for bit (same k order per MFMA chain)
    std::vector<float> c0((size_t)M * N), c1((size_t)M * N);
    probe<4, 0><<<(M / BM) * (N / BN), 256>>>(A, B, C, M, N, K);
    (void)hipMemcpy(c0.data(), C, c0.size() * 4, hipMemcpyDeviceToHost);
    for (int nb = 2; nb <= 4; ++nb) {
        (void)hipMemset(C, 0, c0.size() * 4);
        if (nb == 2) probe_glds<2><<<(M / BM) * (N / BN), 256>>>(A, B, C, M, N, K);
        if (nb == 3) probe_glds<3><<<(M / BM) * (N / BN), 256>>>(A, B, C, M, N, K);
        if (nb == 4) probe_glds<4><<<(M / BM) * (N / BN), 256>>>(A, B, C, M, N, K);
        (void)hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (size_t i = 0; i < c0.size(); ++i) bad += c0[i] != c1[i];
        printf("glds<%d> vs register-staged: %zu of %zu elements differ\n", nb, bad, c0.size());
    }
    return 0;
}
