// Do VALU instructions of the SAME wave issue in the shadow of its MFMAs?  One wave per SIMD, a pinned asm loop of
// {1 v_mfma_f32_32x32x16_bf16, V independent VALU ops}; prints cycles per MFMA at 2.4 GHz nominal for V = 0..10.
// Build: hipcc -w -O3 --offload-arch=gfx950 mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int V, int KIND>
__global__ void __launch_bounds__(256) k(float *out, int iters, float a0) {
    f32x16 c0, c1;
    for (int e = 0; e < 16; ++e) { c0[e] = 0.f; c1[e] = 0.f; }
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(a0 + threadIdx.x * 1e-3f + e); b[e] = (__bf16)(1.f + e); }
    float x0 = a0 + threadIdx.x, x1 = a0 * 3.f, y = 0.f;
    unsigned u = threadIdx.x, w = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep) {
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %2, %3, %0\n\t" : "+v"(c0), "+v"(c1) : "v"(a), "v"(b));
#pragma unroll
            for (int v = 0; v < V; ++v) {
                if (KIND == 0) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(y) : "v"(x0), "v"(x1));
                if (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(x0), "v"(x1));
                if (KIND == 2) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(w) : "v"(u));
            }
            asm volatile("v_mfma_f32_32x32x16_bf16 %1, %2, %3, %1\n\t" : "+v"(c0), "+v"(c1) : "v"(a), "v"(b));
#pragma unroll
            for (int v = 0; v < V; ++v) {
                if (KIND == 0) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(y) : "v"(x0), "v"(x1));
                if (KIND == 1) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(w) : "v"(x0), "v"(x1));
                if (KIND == 2) asm volatile("v_and_b32 %0, 0xffff0000, %1" : "=v"(w) : "v"(u));
            }
        }
    }
    float s = y + (float)w;
    for (int e = 0; e < 16; ++e) s += c0[e] + c1[e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int V, int KIND>
void run(float *out) {
    const int iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<V, KIND><<<256, 256>>>(out, 10, 1.f);
    (void)hipEventRecord(e0);
    k<V, KIND><<<256, 256>>>(out, iters, 1.f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const char *names[] = {"v_sub_f32", "v_cvt_pk_bf16_f32", "v_and_b32 literal"};
    printf("%-20s V=%2d per MFMA: %6.1f cycles per MFMA (2.4 GHz nominal)\n", names[KIND], V, ms * 1e-3 * 2.4e9 / (iters * 16.0));
}
int main() {
    float *out; (void)hipMalloc(&out, 256 * 256 * sizeof(float));
    run<0, 0>(out); run<2, 0>(out); run<4, 0>(out); run<6, 0>(out); run<8, 0>(out); run<10, 0>(out);
    run<2, 1>(out); run<4, 1>(out); run<6, 1>(out); run<8, 1>(out);
    run<2, 2>(out); run<4, 2>(out); run<6, 2>(out); run<8, 2>(out);
    (void)hipFree(out); return 0;
}
