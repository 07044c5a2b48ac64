// Cost probe for DESIGN 4.7 (per-forward abs-max words): G workgroups (256 threads) each stream some bytes, then ONE lane does a
// no-return agent-scope atomicMax on word[(block % W) * stride].  Question: what do W words at a given byte stride cost a launch of
// ~1000 (batch 1) / ~16000 (batch 8) workgroups that all end together, against the same kernel without the atomic?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void __launch_bounds__(256) k(const float4 *src, unsigned *words, int W, int stride_words, int do_atomic, float4 *sink) {
    const int t = threadIdx.x;
    float4 v = src[(size_t)blockIdx.x * 256 + t];
    float m = fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
    for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float s[4];
    if ((t & 63) == 0) s[t >> 6] = m;
    __syncthreads();
    if (t == 0) {
        m = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
        if (do_atomic == 1) atomicMax(words + (size_t)(blockIdx.x % W) * stride_words, __float_as_uint(m));
        else if (do_atomic == 2) { if (m > 1e30f) sink[0] = v; }
    }
}
int main() {
    const int GMAX = 16384;
    float4 *src, *sink; unsigned *words;
    hipMalloc(&src, (size_t)GMAX * 256 * 16); hipMalloc(&sink, 64); hipMalloc(&words, 64 << 20);
    hipMemset(src, 0, (size_t)GMAX * 256 * 16); hipMemset(words, 0, 64 << 20);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int G : {1024, 4096, 16384}) {
        auto run = [&](int W, int stride_words, int mode) {
            for (int i = 0; i < 3; ++i) k<<<G, 256>>>(src, words, W, stride_words, mode, sink);
            hipEventRecord(e0);
            const int R = 50;
            for (int i = 0; i < R; ++i) k<<<G, 256>>>(src, words, W, stride_words, mode, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            return ms / R * 1e3f;
        };
        printf("G = %5d workgroups: no atomic %.2f us\n", G, run(1, 1, 2));
        for (int W : {1, 8, 32, 64})
            for (int sb : {4, 64, 256, 4096, 4352}) {
                if ((size_t)W * sb > (64u << 20)) continue;
                printf("  W = %2d words, stride %5d B: %.2f us\n", W, sb, run(W, sb / 4, 1));
            }
    }
    return 0;
}
