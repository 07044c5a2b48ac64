// tsod_internal.h -- helpers shared by the HIP translation units of libtsod.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/tsod.h"

#define TSOD_WAVE 64

static inline hipStream_t tsod_stream(tsod_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

static inline int tsod_launch_status() {
    return hipGetLastError() == hipSuccess ? TSOD_OK : TSOD_ERR_LAUNCH;
}

static inline bool tsod_aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

static inline int64_t tsod_cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

#define TSOD_REQUIRE(cond, code) \
    do {                         \
        if (!(cond)) return (code); \
    } while (0)


// ---- range words (include/tsod.h "Range words"): the abs-max of a tensor as TSOD_AMAX_WORDS u32 words, TSOD_AMAX_STRIDE bytes
// apart (the bit pattern of a non-negative float orders like the float); the tensor's abs-max is the largest word.  A producer
// adds ONE no-return agent-scope atomicMax per workgroup, to word (block % TSOD_AMAX_WORDS): measured on MI355X
// (scripts/micro/amax_atomics.hip) 64 words at a 64-byte stride cost a launch of 1 024 / 4 096 / 16 384 workgroups that all end
// together -0.1 / +0.6 / +0.6 us, ONE word +10.6 / +45 / +178 us (11.5 ns per serialised atomic), 4-byte stride +3.8 / +18 / +68 us.
#ifdef __HIPCC__
#define TSOD_AMAX_STRIDE_WORDS (TSOD_AMAX_STRIDE / 4)
__device__ __forceinline__ float tsod_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
// Every thread of the workgroup calls this (uniformly) with the largest |value| it stored (>= 0; fmaxf drops NaN, a NaN output
// shows up in the consumer's range flag instead).  `smem`: >= blockDim.x / 64 floats of LDS; other waves may still be using
// OTHER parts of the array it belongs to - the first barrier makes the words free, the second publishes them.
__device__ __forceinline__ void tsod_amax_commit(unsigned *amax, float mx, float *smem, int tid, int nthreads) {
    mx = tsod_wave_max(mx);
    __syncthreads();
    if ((tid & 63) == 0) smem[tid >> 6] = mx;
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < nthreads / 64; ++w) mx = fmaxf(mx, smem[w]);
        atomicMax(amax + (blockIdx.x % TSOD_AMAX_WORDS) * TSOD_AMAX_STRIDE_WORDS, __float_as_uint(mx));
    }
}
// wave-uniform largest word (bits of the tensor's abs-max so far); every lane of a full wave calls it
__device__ __forceinline__ unsigned tsod_amax_reduce_bits(unsigned mine) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const unsigned other = (unsigned)__shfl_xor((int)mine, o); mine = other > mine ? other : mine; }
    return (unsigned)__builtin_amdgcn_readfirstlane((int)mine);
}
static_assert(TSOD_AMAX_WORDS == 64, "one word per lane of a wave");
// fp16x2 activation exponent for a tensor whose abs-max has these bits: 2^e * absmax < 2^15 (fp16 ends at 65504), e in [-24, 24]
// (zero / subnormal abs-max: 24; inf: -24 - the range flag of the launch then reports the non-finite input)
__device__ __forceinline__ int tsod_fp16x2_exp_from_bits(unsigned bits) {
    const int e = 141 - (int)(bits >> 23);                 // 14 - (biased exponent - 127)
    return e < -24 ? -24 : (e > 24 ? 24 : e);
}
#endif
