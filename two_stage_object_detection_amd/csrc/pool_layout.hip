// pool_layout.hip -- HBM-bound NHWC layer kernels: 3x3/s2 max pool, depthwise 3x3, the grouped 1x1
// "pair" conv of the HarDNet tail, and the NCHW <-> NHWC changes at the module boundary.
// All are streaming kernels: 16-byte (float4 over channels) accesses, no LDS reuse needed except
// the tiled transposes, no MFMA.
//
//   nn.MaxPool2d(3, 2, 1)                      models/resnet.py:98,139
//   DWConvLayer (dw3x3 + BN), tail dw3x3 s2    models/hardnet.py:21-36, :193-195
//   nn.Conv2d(1024, 512, 1, groups=512)        models/hardnet.py:196
#include "tsod_internal.h"
#include <math.h>

namespace {

__global__ void __launch_bounds__(256)
maxpool3x3s2_kernel(const float *__restrict__ in, int N, int H, int W, int C4, int in_pitch, int OH, int OW,
                    float *__restrict__ out, int out_pitch) {
    const long total = (long)N * OH * OW * C4;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(t % C4);
        long u = t / C4;
        const int ow = (int)(u % OW);
        u /= OW;
        const int oh = (int)(u % OH);
        const int n = (int)(u / OH);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const int ih = oh * 2 - 1 + dh;
            if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const int iw = ow * 2 - 1 + dw;
                if ((unsigned)iw >= (unsigned)W) continue;
                const float4 v = *reinterpret_cast<const float4 *>(in + (((long)n * H + ih) * W + iw) * in_pitch + 4 * c4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        *reinterpret_cast<float4 *>(out + (((long)n * OH + oh) * OW + ow) * out_pitch + 4 * c4) = m;
    }
}

// Depthwise 3x3, pad 1.  Each thread owns one channel quad of OUTS adjacent output pixels of a row and walks the 3
// input rows once: the (OUTS-1)*STRIDE + 3 input columns of a row are loaded once and shared by the OUTS outputs
// (stride 1: 6 loads feed 12 taps), halving the L1/L2 traffic of the one-pixel-per-thread form.  Taps are accumulated
// in the same (dh, dw) order per output as a plain loop, so results do not depend on OUTS.
template <int STRIDE, int OUTS>
__global__ void __launch_bounds__(256)
dwconv3x3_kernel(const float *__restrict__ in, int N, int H, int W, int C4, int in_pitch, int in_off,
                 const float *__restrict__ w, const float *__restrict__ scale, const float *__restrict__ shift,
                 int relu, int OH, int OW, float *__restrict__ out, int out_pitch, int out_off) {
    constexpr int COLS = (OUTS - 1) * STRIDE + 3;
    const int OWG = (OW + OUTS - 1) / OUTS;
    const long total = (long)N * OH * OWG * C4;
    const int C = C4 * 4;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(t % C4);
        long u = t / C4;
        const int og = (int)(u % OWG);
        u /= OWG;
        const int oh = (int)(u % OH);
        const int n = (int)(u / OH);
        const int ow0 = og * OUTS;
        const int iw0 = ow0 * STRIDE - 1;
        float4 acc[OUTS];
#pragma unroll
        for (int o = 0; o < OUTS; ++o) acc[o] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int dh = 0; dh < 3; ++dh) {
            const int ih = oh * STRIDE - 1 + dh;
            if ((unsigned)ih >= (unsigned)H) continue;
            const float *rowp = in + (((long)n * H + ih) * W) * in_pitch + in_off + 4 * c4;
            float4 v[COLS];
#pragma unroll
            for (int cidx = 0; cidx < COLS; ++cidx) {
                const int iw = iw0 + cidx;
                v[cidx] = (unsigned)iw < (unsigned)W ? *reinterpret_cast<const float4 *>(rowp + (long)iw * in_pitch)
                                                     : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const float4 k = *reinterpret_cast<const float4 *>(w + (dh * 3 + dw) * C + 4 * c4);
#pragma unroll
                for (int o = 0; o < OUTS; ++o) {
                    const int iw = iw0 + o * STRIDE + dw;
                    if ((unsigned)iw >= (unsigned)W) continue;      // padded taps are skipped, exactly like the plain loop
                    const float4 x = v[o * STRIDE + dw];
                    acc[o].x += x.x * k.x; acc[o].y += x.y * k.y; acc[o].z += x.z * k.z; acc[o].w += x.w * k.w;
                }
            }
        }
        float4 s = make_float4(1.f, 1.f, 1.f, 1.f), b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (scale) s = *reinterpret_cast<const float4 *>(scale + 4 * c4);
        if (shift) b = *reinterpret_cast<const float4 *>(shift + 4 * c4);
#pragma unroll
        for (int o = 0; o < OUTS; ++o) {
            const int ow = ow0 + o;
            if (ow >= OW) continue;
            float4 r = make_float4(acc[o].x * s.x + b.x, acc[o].y * s.y + b.y, acc[o].z * s.z + b.z, acc[o].w * s.w + b.w);
            if (relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
            *reinterpret_cast<float4 *>(out + (((long)n * OH + oh) * OW + ow) * out_pitch + out_off + 4 * c4) = r;
        }
    }
}

__global__ void __launch_bounds__(256)
gconv1x1_pair_kernel(const float *__restrict__ in, long pixels, int G, int in_pitch, const float *__restrict__ w,
                     const float *__restrict__ bias, float *__restrict__ out, int out_pitch) {
    const long total = pixels * G;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int g = (int)(t % G);
        const long px = t / G;
        const float2 v = *reinterpret_cast<const float2 *>(in + px * in_pitch + 2 * g);
        const float2 k = *reinterpret_cast<const float2 *>(w + 2 * g);
        float o = v.x * k.x + v.y * k.y;
        if (bias) o += bias[g];
        out[px * out_pitch + g] = o;
    }
}

// C <= 4 images (the RGB input): one thread per pixel, planes read coalesced along w.
__global__ void __launch_bounds__(256)
nchw_to_nhwc_small_kernel(const float *__restrict__ in, int N, int C, long HW, float *__restrict__ out, int out_pitch,
                          int C_pad) {
    const long total = (long)N * HW;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const long n = t / HW, px = t - n * HW;
        float v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = c < C ? in[(n * C + c) * HW + px] : 0.f;
        float *o = out + t * out_pitch;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < C_pad) o[c] = v[c];
    }
}

// generic 32x32 tiled transposes through LDS (pitch 33: conflict-free column reads)
__global__ void __launch_bounds__(256)
nchw_to_nhwc_tile_kernel(const float *__restrict__ in, int C, long HW, float *__restrict__ out, int out_pitch,
                         int C_pad) {
    __shared__ float tile[32][33];
    const long n = blockIdx.z;
    const long px0 = (long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r;
        const long px = px0 + tx;
        tile[r][tx] = (c < C && px < HW) ? in[(n * C + c) * HW + px] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const long px = px0 + r;
        const int c = c0 + tx;
        if (px < HW && c < C_pad) out[(n * HW + px) * out_pitch + c] = tile[tx][r];
    }
}

__global__ void __launch_bounds__(256)
nhwc_to_nchw_tile_kernel(const float *__restrict__ in, int C, long HW, int in_pitch, int in_off,
                         float *__restrict__ out) {
    __shared__ float tile[32][33];
    const long n = blockIdx.z;
    const long px0 = (long)blockIdx.x * 32;
    const int c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const long px = px0 + r;
        const int c = c0 + tx;
        tile[r][tx] = (px < HW && c < C) ? in[(n * HW + px) * in_pitch + in_off + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r;
        const long px = px0 + tx;
        if (c < C && px < HW) out[(n * C + c) * HW + px] = tile[tx][r];
    }
}

inline int grid_for(long total, int threads, int cap) {
    const long b = (total + threads - 1) / threads;
    return (int)(b < cap ? b : cap);
}

}  // namespace

extern "C" int tsod_maxpool3x3s2_f32(const float *in, int32_t N, int32_t H, int32_t W, int32_t C, int32_t in_pitch,
                                     float *out, int32_t out_pitch, tsod_stream_t stream) {
    TSOD_REQUIRE(in && out, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE((C & 3) == 0 && (in_pitch & 3) == 0 && (out_pitch & 3) == 0 && in_pitch >= C && out_pitch >= C,
                 TSOD_ERR_ALIGNMENT);
    TSOD_REQUIRE(tsod_aligned16(in) && tsod_aligned16(out), TSOD_ERR_ALIGNMENT);
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    const long total = (long)N * OH * OW * (C / 4);
    hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, tsod_stream(stream), in, N,
                       H, W, C / 4, in_pitch, OH, OW, out, out_pitch);
    return tsod_launch_status();
}

extern "C" int tsod_dwconv3x3_f32(const float *in, int32_t N, int32_t H, int32_t W, int32_t C, int32_t in_pitch,
                                  int32_t in_off, const float *w, const float *scale, const float *shift,
                                  int32_t stride, int32_t relu, float *out, int32_t out_pitch, int32_t out_off,
                                  tsod_stream_t stream) {
    TSOD_REQUIRE(in && w && out, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && (stride == 1 || stride == 2), TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE((C & 3) == 0 && (in_pitch & 3) == 0 && (out_pitch & 3) == 0 && (in_off & 3) == 0 && (out_off & 3) == 0,
                 TSOD_ERR_ALIGNMENT);
    TSOD_REQUIRE(in_off >= 0 && out_off >= 0 && in_pitch >= in_off + C && out_pitch >= out_off + C, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(tsod_aligned16(in) && tsod_aligned16(out) && tsod_aligned16(w), TSOD_ERR_ALIGNMENT);
    TSOD_REQUIRE((!scale || tsod_aligned16(scale)) && (!shift || tsod_aligned16(shift)), TSOD_ERR_ALIGNMENT);
    const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
    if (stride == 1) {
        const long total = (long)N * OH * ((OW + 3) / 4) * (C / 4);
        hipLaunchKernelGGL((dwconv3x3_kernel<1, 4>), dim3(grid_for(total, 256, 16384)), dim3(256), 0, tsod_stream(stream), in,
                           N, H, W, C / 4, in_pitch, in_off, w, scale, shift, relu, OH, OW, out, out_pitch, out_off);
    } else {
        const long total = (long)N * OH * ((OW + 1) / 2) * (C / 4);
        hipLaunchKernelGGL((dwconv3x3_kernel<2, 2>), dim3(grid_for(total, 256, 16384)), dim3(256), 0, tsod_stream(stream), in,
                           N, H, W, C / 4, in_pitch, in_off, w, scale, shift, relu, OH, OW, out, out_pitch, out_off);
    }
    return tsod_launch_status();
}

extern "C" int tsod_gconv1x1_pair_f32(const float *in, int64_t pixels, int32_t G, int32_t in_pitch, const float *w,
                                      const float *bias, float *out, int32_t out_pitch, tsod_stream_t stream) {
    TSOD_REQUIRE(in && w && out, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(pixels > 0 && G > 0 && in_pitch >= 2 * G && out_pitch >= G, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE((in_pitch & 1) == 0 && (reinterpret_cast<uintptr_t>(in) & 7u) == 0 &&
                     (reinterpret_cast<uintptr_t>(w) & 7u) == 0,
                 TSOD_ERR_ALIGNMENT);
    hipLaunchKernelGGL(gconv1x1_pair_kernel, dim3(grid_for(pixels * G, 256, 8192)), dim3(256), 0, tsod_stream(stream), in,
                       (long)pixels, G, in_pitch, w, bias, out, out_pitch);
    return tsod_launch_status();
}

extern "C" int tsod_nchw_to_nhwc_f32(const float *in, int32_t N, int32_t C, int32_t H, int32_t W, float *out,
                                     int32_t out_pitch, int32_t C_pad, tsod_stream_t stream) {
    TSOD_REQUIRE(in && out, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && C_pad >= C && out_pitch >= C_pad, TSOD_ERR_INVALID_ARG);
    const long HW = (long)H * W;
    if (C_pad <= 4) {
        hipLaunchKernelGGL(nchw_to_nhwc_small_kernel, dim3(grid_for((long)N * HW, 256, 8192)), dim3(256), 0,
                           tsod_stream(stream), in, N, C, HW, out, out_pitch, C_pad);
    } else {
        TSOD_REQUIRE(N <= 65535 && (C_pad + 31) / 32 <= 65535, TSOD_ERR_UNSUPPORTED);
        hipLaunchKernelGGL(nchw_to_nhwc_tile_kernel, dim3((unsigned)((HW + 31) / 32), (C_pad + 31) / 32, N), dim3(256), 0,
                           tsod_stream(stream), in, C, HW, out, out_pitch, C_pad);
    }
    return tsod_launch_status();
}

extern "C" int tsod_nhwc_to_nchw_f32(const float *in, int32_t N, int32_t C, int32_t H, int32_t W, int32_t in_pitch,
                                     int32_t in_off, float *out, tsod_stream_t stream) {
    TSOD_REQUIRE(in && out, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && in_off >= 0 && in_pitch >= in_off + C, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(N <= 65535 && (C + 31) / 32 <= 65535, TSOD_ERR_UNSUPPORTED);
    const long HW = (long)H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_tile_kernel, dim3((unsigned)((HW + 31) / 32), (C + 31) / 32, N), dim3(256), 0,
                       tsod_stream(stream), in, C, HW, in_pitch, in_off, out);
    return tsod_launch_status();
}
