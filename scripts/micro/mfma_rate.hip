// MFMA f32 32x32x2 issue-rate microbenchmark: CHAINS independent accumulators per wave, WAVES waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int CHAINS>
__global__ void __launch_bounds__(1024) k(float *out, int iters, float a0, float b0) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) for (int e = 0; e < 16; ++e) s += acc[c][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS>
void run(int waves_per_simd, float *out) {
    const int iters = 2000, threads = 64 * 4 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<CHAINS><<<256, threads>>>(out, 10, 1.f, 1.f);
    hipEventRecord(e0);
    k<CHAINS><<<256, threads>>>(out, iters, 1.f, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma_per_simd = (double)iters * 16 * CHAINS * waves_per_simd;
    const double flops = n_mfma_per_simd * 4096.0 * 4 * 256;
    printf("chains %d waves/SIMD %d: %.3f ms  %.1f TF/s  %.1f cycles/MFMA/SIMD @2.4GHz\n", CHAINS, waves_per_simd, ms,
           flops / ms / 1e9, ms * 1e-3 * 2.4e9 / n_mfma_per_simd);
}
int main() {
    float *out; hipMalloc(&out, 256 * 1024 * 4);
    run<1>(1, out); run<2>(1, out); run<4>(1, out); run<1>(2, out); run<1>(4, out); run<2>(2, out);
    return 0;
}
