// bf16 MFMA on RANDOM operands: wall-clock FLOP/s and in-kernel clock of v_mfma_f32_32x32x16_bf16 vs v_mfma_f32_16x16x32_bf16 at
// the SAME output tile per wave (32 x 128) and the same bf16x3 product structure (3 planes per operand, 6 products), operands in
// registers (MI355X_MICROARCH.md, DVFS give-back item 7: "the chip can hold a higher clock on one shape than on the other").
// Build: hipcc -O3 --offload-arch=gfx950 mfma_shape_bf16.hip -o mfma_shape_bf16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// one "chunk" = 32 k of a 32 x 128 wave tile x 6 piece products:
//   SHAPE 32: 2 k-halves x 4 col blocks x 6 products = 48 MFMAs 32x32x16 (32 cycles each)
//   SHAPE 16: 2 row blocks x 8 col blocks x 6 products = 96 MFMAs 16x16x32 (16 cycles each)
template <int SHAPE, int WAVES>
__global__ void __launch_bounds__(64 * WAVES) k(const unsigned *__restrict__ src, float *out, long long *stamps, int iters) {
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    bf16x8 a[2][3], b[8][3];                       // SHAPE 32 uses a[2 k-halves][3], b[4 cols x 2 k-halves][3]; SHAPE 16: a[2 rows], b[8 cols]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            u32x4 v;
            for (int e = 0; e < 4; ++e) v[e] = src[(t * 97 + i * 31 + q * 7 + e) & 0xFFFFF];
            a[i][q] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            u32x4 v;
            for (int e = 0; e < 4; ++e) v[e] = src[(t * 131 + j * 29 + q * 11 + e + 4096) & 0xFFFFF];
            b[j][q] = __builtin_bit_cast(bf16x8, v);
        }
    f32x16 acc32[4];
    f32x4 acc16[2][8];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc32[j][e] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc16[i][j][e] = 0.f;
    long long c0 = 0, r0 = 0;
    if (stamps && threadIdx.x == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
    for (int it = 0; it < iters; ++it) {
        if (SHAPE == 32) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc32[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[h][PA[q]], b[j * 2 + h][PB[q]], acc32[j], 0, 0, 0);
        } else {
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i][PA[q]], b[j][PB[q]], acc16[i][j], 0, 0, 0);
        }
    }
    if (stamps && threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - c0;
        stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
    float s = 0.f;
    if (SHAPE == 32) { for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc32[j][e]; }
    else { for (int i = 0; i < 2; ++i) for (int j = 0; j < 8; ++j) for (int e = 0; e < 4; ++e) s += acc16[i][j][e]; }
    out[t] = s;
}

template <int SHAPE, int WAVES>
static void run(const char *name, const unsigned *dsrc, float *dout, long long *dst, int wgs) {
    const int iters = 4000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 40; ++w) k<SHAPE, WAVES><<<wgs, 64 * WAVES>>>(dsrc, dout, nullptr, iters);      // ~1 s of sustained load
    (void)hipEventRecord(e0);
    for (int w = 0; w < 10; ++w) k<SHAPE, WAVES><<<wgs, 64 * WAVES>>>(dsrc, dout, nullptr, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    k<SHAPE, WAVES><<<wgs, 64 * WAVES>>>(dsrc, dout, dst, iters);
    std::vector<long long> h((size_t)wgs * 2);
    (void)hipMemcpy(h.data(), dst, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> clk;
    for (int g = 0; g < wgs; ++g) if (h[2 * g + 1] > 0) clk.push_back((double)h[2 * g] / h[2 * g + 1] * 100e6);
    std::sort(clk.begin(), clk.end());
    const double flop = (double)wgs * WAVES * iters * 32.0 * 128.0 * 32.0 * 2.0 * 6.0;     // executed bf16 FLOPs
    const double cyc_per_chunk = clk.empty() ? 0 : (double)h[0] / iters;
    printf("%-34s %8.1f us  %7.1f TFLOP/s bf16 executed = %6.1f TF/s f32-equivalent | in-kernel clock %.2f GHz, %6.0f cycles per 32-k chunk (ideal 1536 per wave)\n",
           name, ms * 1e3, flop / ms * 1e-9, flop / 6.0 / ms * 1e-9, clk.empty() ? 0.0 : clk[clk.size() / 2] / 1e9, cyc_per_chunk);
}

int main() {
    std::vector<unsigned> hsrc(1 << 20);
    unsigned s = 777u;
    for (auto &v : hsrc) {                          // two random bf16 in [-1, 1) per word
        s = s * 1664525u + 1013904223u; const float f0 = ((s >> 8) & 0xffff) / 32768.f - 1.f;
        s = s * 1664525u + 1013904223u; const float f1 = ((s >> 8) & 0xffff) / 32768.f - 1.f;
        unsigned b0, b1; memcpy(&b0, &f0, 4); memcpy(&b1, &f1, 4);
        v = (b0 >> 16) | (b1 & 0xffff0000u);
    }
    unsigned *dsrc; float *dout; long long *dst;
    (void)hipMalloc(&dsrc, hsrc.size() * 4); (void)hipMalloc(&dout, 4 << 20); (void)hipMalloc(&dst, 8192 * 16);
    (void)hipMemcpy(dsrc, hsrc.data(), hsrc.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        run<32, 4>("32x32x16, 1 wave/SIMD (256 CUs)", dsrc, dout, dst, 256);
        run<16, 4>("16x16x32, 1 wave/SIMD (256 CUs)", dsrc, dout, dst, 256);
        run<32, 8>("32x32x16, 2 waves/SIMD", dsrc, dout, dst, 256);
        run<16, 8>("16x16x32, 2 waves/SIMD", dsrc, dout, dst, 256);
    }
    return 0;
}
