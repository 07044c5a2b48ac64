// v_mfma_f32_32x32x16_bf16 issue-rate microbenchmark (the instruction of the bf16x3 conv path): CHAINS independent
// accumulators per wave, W waves per SIMD, no memory traffic at all.  What it prints is the MFMA rate this part sustains
// under its power limit, to be read beside the nominal 2516.8 TFLOP/s of the micro-architecture guide.
// Build + run on the GPU box: hipcc -O3 --offload-arch=gfx950 mfma_rate_bf16.hip -o mfma_rate_bf16 && ./mfma_rate_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int CHAINS>
__global__ void __launch_bounds__(1024) k(float *out, int iters, float a0) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(a0 + threadIdx.x * 1e-3f + e); b[e] = (__bf16)(1.f + e); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    }
    float s = 0.f;
    for (int c = 0; c < CHAINS; ++c) for (int e = 0; e < 16; ++e) s += acc[c][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS>
void run(int waves_per_simd, float *out, int iters) {
    const int threads = 64 * 4 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<CHAINS><<<256, threads>>>(out, 10, 1.f);
    hipEventRecord(e0);
    k<CHAINS><<<256, threads>>>(out, iters, 1.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double n_mfma_per_simd = (double)iters * 16 * CHAINS * waves_per_simd;
    const double flops = n_mfma_per_simd * 32768.0 * 4 * 256;
    printf("chains %d  waves/SIMD %d  iters %6d : %8.3f ms  %7.1f TFLOP/s  (%.1f cycles/MFMA at 2.4 GHz)\n", CHAINS, waves_per_simd,
           iters, ms, flops / ms * 1e-9, ms * 1e-3 * 2.4e9 / n_mfma_per_simd);
}
int main() {
    float *out; hipMalloc(&out, 256 * 1024 * sizeof(float));
    for (int w : {1, 2, 4}) { run<1>(w, out, 4000 / w); run<2>(w, out, 2000 / w); run<4>(w, out, 1000 / w); }
    // a long run, to see the sustained (power-limited) rate rather than the first milliseconds
    run<4>(2, out, 40000); run<4>(2, out, 40000);
    hipFree(out); return 0;
}
