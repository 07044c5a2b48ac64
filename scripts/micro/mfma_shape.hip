// f32 MFMA on RANDOM operands: FLOP/s and in-kernel clock of the 32x32x2 and 16x16x4 shapes at the same 64x64 output tile
// per wave (MI355X_MICROARCH.md, DVFS give-back items 6 and 7).  Build: hipcc -O3 --offload-arch=gfx950 mfma_shape.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// SHAPE 32: 4 accumulators of 32x32 (2x2 blocks), per k-pair 2 A regs x 2 B regs -> 4 MFMAs of 4096 FLOP
// SHAPE 16: 16 accumulators of 16x16 (4x4 blocks), per k-quad 4 A regs x 4 B regs -> 16 MFMAs of 2048 FLOP
template <int SHAPE>
__global__ void __launch_bounds__(256) k(const float *__restrict__ src, float *out, long long *stamps, int iters) {
    constexpr int NB = SHAPE == 32 ? 2 : 4;
    float a[8][NB], b[8][NB];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            a[u][i] = src[(t * 64 + u * 8 + i) & 0xFFFFF];
            b[u][i] = src[(t * 64 + u * 8 + 4 + i) & 0xFFFFF];
        }
    f32x16 acc32[2][2];
    f32x4 acc16[4][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc32[i][j][e] = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc16[i][j][e] = 0.f;
    long long c0 = 0, r0 = 0;
    if (threadIdx.x == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (SHAPE == 32) {
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc32[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u][i], b[u][j], acc32[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][i], b[u][j], acc16[i][j], 0, 0, 0);
            }
        }
    }
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
        stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    float s = 0.f;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int e = 0; e < 16; ++e) s += acc32[i][j][e];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int e = 0; e < 4; ++e) s += acc16[i][j][e];
    out[t] = s;
}

template <int SHAPE>
void run(const char *what, const float *src, float *out, long long *stamps, int waves_per_simd) {
    const int blocks = 256 * waves_per_simd, iters = 20000;   // 256-thread blocks: one wave per SIMD each
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0.f;
    for (int rep = 0; rep < 40; ++rep) {                      // ~2 s of back-to-back launches before the reading
        hipEventRecord(e0);
        k<SHAPE><<<blocks, 256>>>(src, out, stamps, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<long long> h(2 * blocks);
    hipMemcpy(h.data(), stamps, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    std::vector<double> clk;
    for (int i = 0; i < blocks; ++i) clk.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100e6);
    std::sort(clk.begin(), clk.end());
    const double flops = (double)iters * 8 * (SHAPE == 32 ? 4 * 4096.0 : 16 * 2048.0) * 4 * blocks;
    printf("%-6s shape %dx%d waves/SIMD %d: %8.2f ms  %6.1f TF/s  in-kernel clock %.2f GHz\n", what, SHAPE, SHAPE,
           waves_per_simd, ms, flops / ms / 1e9, clk[clk.size() / 2] / 1e9);
}

int main() {
    const size_t n = 1 << 20;
    std::vector<float> h(n);
    srand(7);
    for (auto &v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    float *src, *zero, *out; long long *stamps;
    hipMalloc(&src, n * 4); hipMalloc(&zero, n * 4); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&stamps, 4096 * 16);
    hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(zero, 0, n * 4);
    run<32>("random", src, out, stamps, 1);
    run<16>("random", src, out, stamps, 1);
    run<32>("random", src, out, stamps, 2);
    run<16>("random", src, out, stamps, 2);
    run<32>("zero", zero, out, stamps, 1);
    run<16>("zero", zero, out, stamps, 1);
    return 0;
}
