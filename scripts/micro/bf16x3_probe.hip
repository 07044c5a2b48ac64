// What would an f32-accurate GEMM on the bf16 matrix pipes give?  ("bf16x3 split conv path".)
// Every f32 operand is cut EXACTLY into three bf16 pieces by truncation (8 + 8 + 8 significand bits: hi + mid + lo == x),
// and a*b is accumulated in f32 from the six largest piece products (hi*hi, hi*mid, mid*hi, mid*mid, hi*lo, lo*hi; the
// dropped ones are <= 2^-24 relative), on v_mfma_f32_32x32x16_bf16 (16x the f32 MFMA rate, so 16/6 = 2.7x per product).
// Probe only: C[m][n] = sum_k A[m][k] B[n][k], 128x128 workgroup tile, four waves of 64x64, one LDS stage holding the
// three planes of both operands, operands split in registers on their way to LDS.  Prints TFLOP/s (f32-equivalent) and
// the error against an f64 reference next to the error of a plain f32 summation.
// Build: hipcc -O3 --offload-arch=gfx950 bf16x3_probe.hip -o bf16x3_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int ROW_B = 80;                       // bytes per LDS row of one plane: 32 bf16 + 16 pad (conflict-free b128 reads)
constexpr int PLANE_B = 128 * ROW_B;            // one plane of one operand
constexpr int LDS_B = 6 * PLANE_B;              // A hi/mid/lo, B hi/mid/lo

__device__ __forceinline__ void split3(float x, unsigned &h, unsigned &m, unsigned &l) {
    const unsigned xb = __float_as_uint(x);
    h = xb & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(h);                    // exact
    m = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(m);                   // exact, <= 8 significant bits left
    l = __float_as_uint(r2) & 0xFFFF0000u;
}
__device__ __forceinline__ unsigned pack2(unsigned lo_elem, unsigned hi_elem) { return (lo_elem >> 16) | hi_elem; }

__global__ void __launch_bounds__(256, 2)
gemm_bf16x3(const float *__restrict__ A, const float *__restrict__ B, float *__restrict__ C, int M, int N, int K) {
    extern __shared__ __align__(16) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = N / BN;
    const int nwg = gridDim.x, qq = nwg >> 3, r8 = nwg & 7, xcd = blockIdx.x & 7;
    const int tile = (xcd < r8 ? xcd * (qq + 1) : r8 * (qq + 1) + (xcd - r8) * qq) + (blockIdx.x >> 3);
    const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
    const int c4 = (tid & 7) * 4, r0 = tid >> 3;               // this thread stages k = c4..c4+3 of rows r0 + 32*i
    const float *ap[4], *bp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ap[i] = A + (long)(m0 + r0 + 32 * i) * K + c4;
        bp[i] = B + (long)(n0 + r0 + 32 * i) * K + c4;
    }
    float4 ra[4], rb[4];
    auto load_global = [&](int k) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ra[i] = *reinterpret_cast<const float4 *>(ap[i] + k);
            rb[i] = *reinterpret_cast<const float4 *>(bp[i] + k);
        }
    };
    auto store_split = [&](const float4 &v, int operand, int row) {
        unsigned h[4], m[4], l[4];
        split3(v.x, h[0], m[0], l[0]); split3(v.y, h[1], m[1], l[1]);
        split3(v.z, h[2], m[2], l[2]); split3(v.w, h[3], m[3], l[3]);
        unsigned char *base = lds + operand * 3 * PLANE_B + row * ROW_B + c4 * 2;
        *reinterpret_cast<uint2 *>(base) = make_uint2(pack2(h[0], h[1]), pack2(h[2], h[3]));
        *reinterpret_cast<uint2 *>(base + PLANE_B) = make_uint2(pack2(m[0], m[1]), pack2(m[2], m[3]));
        *reinterpret_cast<uint2 *>(base + 2 * PLANE_B) = make_uint2(pack2(l[0], l[1]), pack2(l[2], l[3]));
    };
    auto store_lds = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            store_split(ra[i], 0, r0 + 32 * i);
            store_split(rb[i], 1, r0 + 32 * i);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // fragment of plane p, 32-row block at row `row`, 16-k chunk c: lane reads 8 bf16 at k = 16*c + 8*(lane/32)
    const int frag_off = (lane & 31) * ROW_B + (lane >> 5) * 16;
    auto frag = [&](int operand, int plane, int row, int chunk) {
        return *reinterpret_cast<const bf16x8 *>(lds + (operand * 3 + plane) * PLANE_B + row * ROW_B + frag_off + chunk * 32);
    };
    const int nk = K / BK;
    load_global(0);
    store_lds();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more) load_global((kt + 1) * BK);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            bf16x8 a[2][3], b[2][3];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    a[i][p] = frag(0, p, wm * 64 + i * 32, c);
                    b[i][p] = frag(1, p, wn * 64 + i * 32, c);
                }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    f32x16 t = acc[i][j];
                    t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][2], b[j][0], t, 0, 0, 0);   // lo*hi
                    t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][2], t, 0, 0, 0);   // hi*lo
                    t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][1], t, 0, 0, 0);   // mid*mid
                    t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][1], b[j][0], t, 0, 0, 0);   // mid*hi
                    t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][1], t, 0, 0, 0);   // hi*mid
                    t = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][0], b[j][0], t, 0, 0, 0);   // hi*hi
                    acc[i][j] = t;
                }
        }
        __syncthreads();
        if (more) store_lds();
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wm * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                const int col = n0 + wn * 64 + j * 32 + (lane & 31);
                C[(long)row * N + col] = acc[i][j][e];
            }
}

int main(int argc, char **argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 16384, N = argc > 2 ? atoi(argv[2]) : 4096, K = argc > 3 ? atoi(argv[3]) : 1024;
    std::vector<float> ha((size_t)M * K), hb((size_t)N * K);
    srand(5);
    for (auto &v : ha) v = (float)rand() / RAND_MAX - 0.5f;
    for (auto &v : hb) v = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
    float *A, *B, *C;
    (void)hipMalloc(&A, ha.size() * 4); (void)hipMalloc(&B, hb.size() * 4); (void)hipMalloc(&C, (size_t)M * N * 4);
    (void)hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_bf16x3), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_B);
    const int grid = (M / BM) * (N / BN);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    gemm_bf16x3<<<grid, 256, LDS_B>>>(A, B, C, M, N, K);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) gemm_bf16x3<<<grid, 256, LDS_B>>>(A, B, C, M, N, K);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("M %d N %d K %d: bf16x3 (6 products) %7.3f ms  %6.1f TFLOP/s f32-equivalent (f32 MFMA peak 157.3)\n", M, N, K, ms,
           2.0 * M * N * K / ms / 1e9);
    std::vector<float> hc((size_t)M * N);
    (void)hipMemcpy(hc.data(), C, hc.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0, worst_f32 = 0, scale = 0;
    for (int s = 0; s < 4000; ++s) {
        const int m = rand() % M, n = rand() % N;
        double ref = 0, mag = 0; float f = 0.f;
        for (int k = 0; k < K; ++k) {
            const double p = (double)ha[(size_t)m * K + k] * (double)hb[(size_t)n * K + k];
            ref += p; mag += std::fabs(p);
            f = fmaf(ha[(size_t)m * K + k], hb[(size_t)n * K + k], f);
        }
        worst = std::fmax(worst, std::fabs(hc[(size_t)m * N + n] - ref) / mag);
        worst_f32 = std::fmax(worst_f32, std::fabs((double)f - ref) / mag);
        scale = std::fmax(scale, mag);
    }
    printf("max |err| / sum|a*b| over 4000 sampled outputs: bf16x3 %.3e   plain f32 fmaf chain %.3e\n", worst, worst_f32);
    return 0;
}
