// resize.hip -- the input step in front of the path: antialiased bilinear resize of a decoded u8 HWC image to the
// detector's fixed input size, fused with the u8 -> f32 conversion and the NHWC(4-channel) / NCHW layout the
// backbone reads.  HBM-bound gather + a handful of FMAs per output value; no LDS, no MFMA.
//
//   dataset/dataloader.py:35-44   PIL RGB -> tv_tensors.Image(img, dtype=float32)  (f32 CHW, values 0..255, quirk: never /255)
//   dataset/transform.py:14-17    eval_transform = Resize((600, 600)) + ToTensor (a pass-through for tensors)
//   torchvision v2 Resize on a float tensor = torch.nn.functional.interpolate(mode="bilinear", antialias=True,
//   align_corners=False): separable triangle filter whose support grows with the down-scale factor
//   (ATen UpSampleKernel.cpp, HelperInterpLinear::aa_filter + _compute_indices_min_size_weights_aa); restated here:
//   the tap tables on the host in f32 with ATen's own expression order, the two passes (horizontal, then vertical) in
//   one kernel with the horizontal sums kept in f32 like ATen's intermediate image.  Compiled with -ffp-contract=off.
#include "tsod_internal.h"
#include <math.h>

namespace {

// out[oy][ox][c] = sum_j wy[oy][j] * (sum_i wx[ox][i] * src[y0+j][x0+i][c]) * mul
__global__ void __launch_bounds__(256)
resize_aa_kernel(const unsigned char *__restrict__ src, int H, int W, int C, long src_row_bytes,
                 const int *__restrict__ yfirst, const int *__restrict__ ycount, const float *__restrict__ ywt, int ytaps,
                 const int *__restrict__ xfirst, const int *__restrict__ xcount, const float *__restrict__ xwt, int xtaps,
                 int OH, int OW, float mul, float *__restrict__ out, long stride_y, long stride_x, long stride_c,
                 int C_out) {
    const long total = (long)OH * OW;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(t % OW);
        const int oy = (int)(t / OW);
        const int x0 = xfirst[ox], nx = xcount[ox];
        const int y0 = yfirst[oy], ny = ycount[oy];
        const float *wx = xwt + (long)ox * xtaps;
        const float *wy = ywt + (long)oy * ytaps;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int j = 0; j < ny; ++j) {
            const unsigned char *row = src + (long)(y0 + j) * src_row_bytes + (long)x0 * C;
            float h[4] = {0.f, 0.f, 0.f, 0.f};
            for (int i = 0; i < nx; ++i) {
                const float w = wx[i];
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (c < C) {
                        const float v = (float)row[i * C + c] * w;
                        h[c] = i == 0 ? v : h[c] + v;
                    }
            }
            const float w = wy[j];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float v = h[c] * w;
                acc[c] = j == 0 ? v : acc[c] + v;
            }
        }
        float *o = out + oy * stride_y + ox * stride_x;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c < C_out) o[c * stride_c] = c < C ? acc[c] * mul : 0.f;
    }
}

// Tiled form: a 256-thread workgroup owns TY x TX output pixels, stages the u8 input region those pixels touch into
// LDS with coalesced aligned dword loads (every input byte is fetched from global memory once per tile instead of once
// per tap), then every thread runs the same horizontal-then-vertical f32 sums as above out of LDS.  The region of a
// tile is [first[o0], first[oL] + count[oL]) per axis (both are non-decreasing in the output index).
constexpr int kTY = 8, kTX = 32;

__global__ void __launch_bounds__(kTY * kTX)
resize_aa_tile_kernel(const unsigned char *__restrict__ src, int H, int W, int C, long src_row_bytes,
                      const int *__restrict__ yfirst, const int *__restrict__ ycount, const float *__restrict__ ywt,
                      int ytaps, const int *__restrict__ xfirst, const int *__restrict__ xcount,
                      const float *__restrict__ xwt, int xtaps, int OH, int OW, float mul, float *__restrict__ out,
                      long stride_y, long stride_x, long stride_c, int C_out, int cap_rows, int cap_row_bytes) {
    extern __shared__ __align__(16) unsigned char region[];          // cap_rows x cap_row_bytes, then the tile's weights
    const int tiles_x = (OW + kTX - 1) / kTX;
    const int ty = blockIdx.x / tiles_x, tx = blockIdx.x % tiles_x;
    const int oy0 = ty * kTY, ox0 = tx * kTX;
    const int oyL = min(oy0 + kTY, OH) - 1, oxL = min(ox0 + kTX, OW) - 1;
    const int ry0 = yfirst[oy0], rx0 = xfirst[ox0];
    const int rows = min(yfirst[oyL] + ycount[oyL] - ry0, cap_rows);
    const int row_bytes = (xfirst[oxL] + xcount[oxL] - rx0) * C;
    // row r of the region starts at global byte address g_r = src + (ry0 + r) * pitch + rx0 * C; LDS holds the aligned
    // dwords from g_r - (g_r & 3) on, so a consumer adds that misalignment back
    const unsigned long long base = reinterpret_cast<unsigned long long>(src) + (unsigned long long)rx0 * C;
    const int words_cap = cap_row_bytes >> 2;
    for (int r = threadIdx.x / 32; r < rows; r += (kTY * kTX) / 32) {
        const unsigned long long g = base + (unsigned long long)(ry0 + r) * src_row_bytes;
        const int mis = (int)(g & 3ull);
        const int words = min((row_bytes + mis + 3) >> 2, words_cap);
        const unsigned *gw = reinterpret_cast<const unsigned *>(g - mis);
        unsigned *lw = reinterpret_cast<unsigned *>(region + (long)r * cap_row_bytes);
        for (int w = threadIdx.x % 32; w < words; w += 32) lw[w] = gw[w];
    }
    // the tile's weight rows go to LDS as well: the tap loops below then touch no global memory at all
    float *s_wx = reinterpret_cast<float *>(region + (long)cap_rows * cap_row_bytes);
    float *s_wy = s_wx + kTX * xtaps;
    for (int t = threadIdx.x; t < kTX * xtaps; t += kTY * kTX) {
        const int o = ox0 + t / xtaps;
        s_wx[t] = o < OW ? xwt[(long)o * xtaps + t % xtaps] : 0.f;
    }
    for (int t = threadIdx.x; t < kTY * ytaps; t += kTY * kTX) {
        const int o = oy0 + t / ytaps;
        s_wy[t] = o < OH ? ywt[(long)o * ytaps + t % ytaps] : 0.f;
    }
    __syncthreads();
    const int ox = ox0 + (threadIdx.x % kTX), oy = oy0 + (threadIdx.x / kTX);
    if (ox >= OW || oy >= OH) return;
    const int x0 = xfirst[ox], nx = xcount[ox];
    const int y0 = yfirst[oy], ny = ycount[oy];
    const float *wx = s_wx + (threadIdx.x % kTX) * xtaps;
    const float *wy = s_wy + (threadIdx.x / kTX) * ytaps;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < ny; ++j) {
        const int r = y0 + j - ry0;
        const int mis = (int)((base + (unsigned long long)(y0 + j) * src_row_bytes) & 3ull);
        const unsigned char *row = region + (long)min(r, cap_rows - 1) * cap_row_bytes + mis + (x0 - rx0) * C;
        float h[4] = {0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < nx; ++i) {
            const float w = wx[i];
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < C) {
                    const float v = (float)row[i * C + c] * w;
                    h[c] = i == 0 ? v : h[c] + v;
                }
        }
        const float w = wy[j];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float v = h[c] * w;
            acc[c] = j == 0 ? v : acc[c] + v;
        }
    }
    float *o = out + oy * stride_y + ox * stride_x;
    if (stride_c == 1 && C_out == 4 && ((stride_x | stride_y) & 3) == 0 && (reinterpret_cast<uintptr_t>(out) & 15u) == 0) {
        // NHWC(4): one 16-byte store per pixel
        *reinterpret_cast<float4 *>(o) = make_float4(acc[0] * mul, C > 1 ? acc[1] * mul : 0.f, C > 2 ? acc[2] * mul : 0.f,
                                                     C > 3 ? acc[3] * mul : 0.f);
        return;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
        if (c < C_out) o[c * stride_c] = c < C ? acc[c] * mul : 0.f;
}

}  // namespace

extern "C" int32_t tsod_resize_aa_taps(int32_t in_size, int32_t out_size) {
    if (in_size <= 0 || out_size <= 0) return 0;
    const float scale = (float)in_size / (float)out_size;
    const float support = scale >= 1.0f ? scale : 1.0f;
    return (int32_t)ceilf(support) * 2 + 1;
}

extern "C" int tsod_resize_aa_tables_f32(int32_t in_size, int32_t out_size, int32_t *first, int32_t *count,
                                         float *weights) {
    TSOD_REQUIRE(first && count && weights && in_size > 0 && out_size > 0, TSOD_ERR_INVALID_ARG);
    const int taps = tsod_resize_aa_taps(in_size, out_size);
    // f32 throughout, in ATen's expression order (area_pixel_compute_scale + _compute_indices_min_size_weights_aa)
    const float scale = (float)in_size / (float)out_size;
    const float support = scale >= 1.0f ? scale : 1.0f;
    const float invscale = scale >= 1.0f ? 1.0f / scale : 1.0f;
    for (int i = 0; i < out_size; ++i) {
        const float center = scale * ((float)i + 0.5f);
        // ATen adds the 0.5 as a double to the f32 difference / sum before truncating
        long lo = (long)((double)(center - support) + 0.5);
        if (lo < 0) lo = 0;
        long hi = (long)((double)(center + support) + 0.5);
        if (hi > in_size) hi = in_size;
        long n = hi - lo;
        if (n < 0) n = 0;
        if (n > taps) n = taps;
        float *w = weights + (long)i * taps;
        float total = 0.f;
        for (long j = 0; j < n; ++j) {
            float x = (float)(((double)((float)(j + lo) - center) + 0.5) * (double)invscale);
            if (x < 0.f) x = -x;
            w[j] = x < 1.0f ? 1.0f - x : 0.f;
            total += w[j];
        }
        if (total != 0.f)
            for (long j = 0; j < n; ++j) w[j] /= total;
        for (long j = n; j < taps; ++j) w[j] = 0.f;
        first[i] = (int32_t)lo;
        count[i] = (int32_t)n;
    }
    return TSOD_OK;
}

extern "C" int tsod_resize_bilinear_aa_u8_f32(const uint8_t *src, int32_t H, int32_t W, int32_t C, int64_t src_row_bytes,
                                              const int32_t *yfirst, const int32_t *ycount, const float *ywt,
                                              const int32_t *xfirst, const int32_t *xcount, const float *xwt,
                                              int32_t OH, int32_t OW, float mul, float *out, int64_t stride_y,
                                              int64_t stride_x, int64_t stride_c, int32_t C_out, tsod_stream_t stream) {
    TSOD_REQUIRE(src && yfirst && ycount && ywt && xfirst && xcount && xwt && out, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(H > 0 && W > 0 && OH > 0 && OW > 0 && C >= 1 && C <= 4 && C_out >= C && C_out <= 4, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(src_row_bytes >= (int64_t)W * C, TSOD_ERR_INVALID_ARG);
    const int ytaps = tsod_resize_aa_taps(H, OH), xtaps = tsod_resize_aa_taps(W, OW);
    // LDS the tiled form needs for the largest region a tile can touch: (T-1)*scale + taps + 2 input rows / columns
    const float sy = (float)H / (float)OH, sx = (float)W / (float)OW;
    const int cap_rows = (int)ceilf((kTY - 1) * sy) + ytaps + 2;
    const int cap_cols = (int)ceilf((kTX - 1) * sx) + xtaps + 2;
    const int cap_row_bytes = ((cap_cols * C + 3 + 3) / 4) * 4;
    const size_t lds = (size_t)cap_rows * cap_row_bytes + (size_t)(kTX * xtaps + kTY * ytaps) * sizeof(float);
    if (lds <= 48 * 1024) {
        const int tiles = ((OH + kTY - 1) / kTY) * ((OW + kTX - 1) / kTX);
        hipLaunchKernelGGL(resize_aa_tile_kernel, dim3(tiles), dim3(kTY * kTX), lds, tsod_stream(stream), src, H, W, C,
                           (long)src_row_bytes, yfirst, ycount, ywt, ytaps, xfirst, xcount, xwt, xtaps, OH, OW, mul, out,
                           (long)stride_y, (long)stride_x, (long)stride_c, C_out, cap_rows, cap_row_bytes);
        return tsod_launch_status();
    }
    const long total = (long)OH * OW;                       // very large down-scales: every tap straight from global memory
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(resize_aa_kernel, dim3(blocks), dim3(256), 0, tsod_stream(stream), src, H, W, C, (long)src_row_bytes,
                       yfirst, ycount, ywt, ytaps, xfirst, xcount, xwt, xtaps, OH, OW, mul, out, (long)stride_y,
                       (long)stride_x, (long)stride_c, C_out);
    return tsod_launch_status();
}
