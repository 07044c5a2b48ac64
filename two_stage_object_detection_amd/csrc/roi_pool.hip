// roi_pool.hip -- RoI max pooling on an NHWC feature map, plain and fused with the RoI rescale and
// the 7x7 mean that the reference's "classifier" applies.  Gather + compare, L2/HBM-bound, no MFMA.
//
//   torchvision.ops.RoIPool (built nets/classify.py:17, called :43; algorithm = torchvision
//     ops/cpu/roi_pool_kernel.cpp): round() half away from zero, +1 extents, float bin sizes,
//     floor/ceil bin edges, clamp to the map, empty bin -> 0, max from -FLT_MAX with strict '>'.
//   nets/classify.py:29-38   rois_fm.x = x / img_w * Wf, rois_fm.y = y / img_h * Hf, row index
//   models/hardnet.py:203-212  AdaptiveAvgPool2d(1) + Flatten  == mean over the PH*PW bins
//
// One 256-thread workgroup per RoI; lanes run over channel quads (float4), so every window read is
// a coalesced sweep over the pixel's channels.  Compiled with -ffp-contract=off.
#include "tsod_internal.h"
#include <float.h>
#include <math.h>

namespace {

struct RoiGeom {
    int b, sw, sh, rw, rh;
};

__device__ __forceinline__ RoiGeom roi_geom(float bidx, float x1, float y1, float x2, float y2, float scale) {
    RoiGeom g;
    g.b = (int)bidx;
    g.sw = (int)roundf(x1 * scale);
    g.sh = (int)roundf(y1 * scale);
    const int ew = (int)roundf(x2 * scale);
    const int eh = (int)roundf(y2 * scale);
    g.rw = max(ew - g.sw + 1, 1);
    g.rh = max(eh - g.sh + 1, 1);
    return g;
}

__device__ __forceinline__ void bin_range(int p, float bin, int start, int limit, int &lo, int &hi) {
    lo = (int)floorf((float)p * bin) + start;
    hi = (int)ceilf((float)(p + 1) * bin) + start;
    lo = min(max(lo, 0), limit);
    hi = min(max(hi, 0), limit);
}

__device__ __forceinline__ float4 max4(float4 m, const float4 v) {
    if (v.x > m.x) m.x = v.x;
    if (v.y > m.y) m.y = v.y;
    if (v.z > m.z) m.z = v.z;
    if (v.w > m.w) m.w = v.w;
    return m;
}

__device__ __forceinline__ float4 bin_max(const float *__restrict__ fmap, int Wf, int pitch, int c4, int hs, int he,
                                          int ws, int we) {
    if (he <= hs || we <= ws) return make_float4(0.f, 0.f, 0.f, 0.f);
    float4 m = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
    for (int h = hs; h < he; ++h) {
        const float *rowp = fmap + ((long)h * Wf) * pitch + 4 * c4;
        for (int w = ws; w < we; ++w) m = max4(m, *reinterpret_cast<const float4 *>(rowp + (long)w * pitch));
    }
    return m;
}

// out [K][C][PH][PW]
__global__ void __launch_bounds__(256)
roi_pool_kernel(const float *__restrict__ feat, int B, int Hf, int Wf, int C, int pitch,
                const float *__restrict__ rois5, float scale, int PH, int PW, float *__restrict__ out) {
    const int k = blockIdx.x;
    const float *r = rois5 + 5l * k;
    const RoiGeom g = roi_geom(r[0], r[1], r[2], r[3], r[4], scale);
    if (g.b < 0 || g.b >= B) return;
    const float bin_h = (float)g.rh / (float)PH;
    const float bin_w = (float)g.rw / (float)PW;
    const float *fmap = feat + (long)g.b * Hf * Wf * pitch;
    const int bins = PH * PW;
    for (int c4 = threadIdx.x; c4 < (C >> 2); c4 += blockDim.x) {
        for (int ph = 0; ph < PH; ++ph) {
            int hs, he;
            bin_range(ph, bin_h, g.sh, Hf, hs, he);
            for (int pw = 0; pw < PW; ++pw) {
                int ws, we;
                bin_range(pw, bin_w, g.sw, Wf, ws, we);
                const float4 m = bin_max(fmap, Wf, pitch, c4, hs, he, ws, we);
                float *o = out + ((long)k * C + 4 * c4) * bins + ph * PW + pw;
                o[0] = m.x; o[bins] = m.y; o[2 * bins] = m.z; o[3 * bins] = m.w;
            }
        }
    }
}

// out [B*R][out_pitch]: mean over the PH*PW bin maxima, RoIs given in image coordinates.
// A workgroup = kQuads channel quads x PH bin rows of one RoI: thread (q, ph) sums the PW bin maxima of its row (the
// per-thread chain is PW windows instead of PH*PW), the PH row sums meet in LDS and are added in row order.
constexpr int kQuads = 32;

__global__ void __launch_bounds__(256)
roi_pool_avg_kernel(const float *__restrict__ feat, int B, int Hf, int Wf, int C, int pitch,
                    const float *__restrict__ rois, const int *__restrict__ roi_indices, int R,
                    float img_h, float img_w, float scale, int PH, int PW, float *__restrict__ out, int out_pitch, int roi_in_y) {
    __shared__ float4 rowsum[8][kQuads];
    // blockIdx.x = channel group, blockIdx.y = RoI: workgroups go to the XCDs round-robin in linear order, so with the channel group
    // fastest XCD x only ever reads channel groups x, x + 8, ... of the feature map - an eighth of it (2.1 MB per image at 50 x 84 x
    // 1024) stays in that XCD's 4 MB L2 across the RoIs of an image, instead of all eight L2s each streaming the whole map
    // (grids beyond 65535 RoIs keep the RoI in x: roi_in_y = 0)
    const int k = roi_in_y ? blockIdx.y : blockIdx.x;
    const float4 rr = reinterpret_cast<const float4 *>(rois)[k];
    // nets/classify.py:35-36: divide by the image side, then multiply by the map side
    const float fx1 = rr.x / img_w * (float)Wf;
    const float fy1 = rr.y / img_h * (float)Hf;
    const float fx2 = rr.z / img_w * (float)Wf;
    const float fy2 = rr.w / img_h * (float)Hf;
    const RoiGeom g = roi_geom((float)roi_indices[k / R], fx1, fy1, fx2, fy2, scale);
    if (g.b < 0 || g.b >= B) return;
    const float bin_h = (float)g.rh / (float)PH;
    const float bin_w = (float)g.rw / (float)PW;
    const float *fmap = feat + (long)g.b * Hf * Wf * pitch;
    const float nb = (float)(PH * PW);
    const int q = threadIdx.x % kQuads, slot = threadIdx.x / kQuads;      // slot < 8
    const int c4 = (roi_in_y ? blockIdx.x : blockIdx.y) * kQuads + q;
    const bool live = c4 < (C >> 2);
    float4 total = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ph0 = 0; ph0 < PH; ph0 += 8) {                                  // PH <= 8: one pass
        const int ph = ph0 + slot;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live && ph < PH) {
            int hs, he;
            bin_range(ph, bin_h, g.sh, Hf, hs, he);
            // Neighbouring bins of a row share at most ONE pixel column (floor / ceil edges of a float bin width): its column maximum is
            // kept from the bin before instead of being read again - a window pixel is read once per bin ROW it belongs to, not once
            // per bin (a 7-bin row read 1.2-2x its width before).  A maximum taken in another order is the same maximum: bit-exact.
            int kept_x = -1;
            float4 kept = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
            for (int pw = 0; pw < PW; ++pw) {
                int ws, we;
                bin_range(pw, bin_w, g.sw, Wf, ws, we);
                float4 m = make_float4(0.f, 0.f, 0.f, 0.f);                   // empty bin
                if (he > hs && we > ws) {
                    m = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
                    for (int w = ws; w < we; ++w) {
                        float4 col = kept;
                        if (w != kept_x) {
                            col = make_float4(-FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX);
                            const float *colp = fmap + ((long)hs * Wf + w) * pitch + 4 * c4;
                            for (int h = hs; h < he; ++h, colp += (long)Wf * pitch) col = max4(col, *reinterpret_cast<const float4 *>(colp));
                        }
                        m = max4(m, col);
                        if (w == we - 1) { kept = col; kept_x = w; }
                    }
                }
                acc.x += m.x; acc.y += m.y; acc.z += m.z; acc.w += m.w;
            }
        }
        rowsum[slot][q] = acc;
        __syncthreads();
        if (slot == 0) {
            const int rows = min(8, PH - ph0);
            for (int r = 0; r < rows; ++r) {
                const float4 v = rowsum[r][q];
                total.x += v.x; total.y += v.y; total.z += v.z; total.w += v.w;
            }
        }
        __syncthreads();
    }
    if (slot == 0 && live)
        *reinterpret_cast<float4 *>(out + (long)k * out_pitch + 4 * c4) =
            make_float4(total.x / nb, total.y / nb, total.z / nb, total.w / nb);
}

// ---- RoIAlign (the north star's wording; the reference's head is built with RoIPool, nets/classify.py:17, so this is the
// added `roi_op="align"` option of SURVEY 8(b)).  Algorithm = torchvision.ops.roi_align (ops/cpu/roi_align_kernel.cpp +
// roi_align_common.h), restated in oracle/box_ops.c: PARITY UNPINNED like RoIPool (torchvision is not in the image).
//   offset = aligned ? 0.5 : 0; start = coord * scale - offset; extent = end - start (>= 1 unless aligned);
//   bin = extent / P; grid = sampling_ratio > 0 ? sampling_ratio : ceil(extent / P); count = max(grid_h * grid_w, 1)
//   sample (iy, ix) of bin (ph, pw): y = start_h + ph * bin_h + (iy + .5) * bin_h / grid_h (x likewise);
//   outside [-1, H] x [-1, W] -> 0; clamp at 0; low = (int), high = low + 1 (both H-1 at the border, y = low there);
//   value = hy*hx*v1 + hy*lx*v2 + ly*hx*v3 + ly*lx*v4; bin output = sum over the grid (iy outer, ix inner) / count.
struct AlignGeom {
    int b, grid_h, grid_w;
    float start_h, start_w, bin_h, bin_w, count;
};

__device__ __forceinline__ AlignGeom align_geom(float bidx, float x1, float y1, float x2, float y2, float scale, int PH,
                                                int PW, int sampling_ratio, int aligned) {
    AlignGeom g;
    g.b = (int)bidx;
    const float offset = aligned ? 0.5f : 0.f;
    g.start_w = x1 * scale - offset;
    g.start_h = y1 * scale - offset;
    const float end_w = x2 * scale - offset, end_h = y2 * scale - offset;
    float rw = end_w - g.start_w, rh = end_h - g.start_h;
    if (!aligned) { rw = fmaxf(rw, 1.f); rh = fmaxf(rh, 1.f); }
    g.bin_h = rh / (float)PH;
    g.bin_w = rw / (float)PW;
    g.grid_h = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rh / (float)PH);
    g.grid_w = sampling_ratio > 0 ? sampling_ratio : (int)ceilf(rw / (float)PW);
    g.count = (float)max(g.grid_h * g.grid_w, 1);
    return g;
}

// sum of the grid samples of bin (ph, pw) for one channel quad (not yet divided by count)
__device__ __forceinline__ float4 align_bin_sum(const float *__restrict__ fmap, int Hf, int Wf, int pitch, int c4,
                                                const AlignGeom &g, int ph, int pw) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int iy = 0; iy < g.grid_h; ++iy) {
        const float yy = g.start_h + (float)ph * g.bin_h + ((float)iy + .5f) * g.bin_h / (float)g.grid_h;
        for (int ix = 0; ix < g.grid_w; ++ix) {
            const float xx = g.start_w + (float)pw * g.bin_w + ((float)ix + .5f) * g.bin_w / (float)g.grid_w;
            float y = yy, x = xx;
            if (y < -1.f || y > (float)Hf || x < -1.f || x > (float)Wf) continue;      // contributes 0
            if (y <= 0.f) y = 0.f;
            if (x <= 0.f) x = 0.f;
            int y_low = (int)y, x_low = (int)x, y_high, x_high;
            if (y_low >= Hf - 1) { y_high = y_low = Hf - 1; y = (float)y_low; } else y_high = y_low + 1;
            if (x_low >= Wf - 1) { x_high = x_low = Wf - 1; x = (float)x_low; } else x_high = x_low + 1;
            const float ly = y - (float)y_low, lx = x - (float)x_low, hy = 1.f - ly, hx = 1.f - lx;
            const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
            const float4 v1 = *reinterpret_cast<const float4 *>(fmap + ((long)y_low * Wf + x_low) * pitch + 4 * c4);
            const float4 v2 = *reinterpret_cast<const float4 *>(fmap + ((long)y_low * Wf + x_high) * pitch + 4 * c4);
            const float4 v3 = *reinterpret_cast<const float4 *>(fmap + ((long)y_high * Wf + x_low) * pitch + 4 * c4);
            const float4 v4 = *reinterpret_cast<const float4 *>(fmap + ((long)y_high * Wf + x_high) * pitch + 4 * c4);
            acc.x += w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x;
            acc.y += w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y;
            acc.z += w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z;
            acc.w += w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w;
        }
    }
    return acc;
}

// out [K][C][PH][PW] (torchvision's layout): thread = (channel quad, bin)
__global__ void __launch_bounds__(256)
roi_align_kernel(const float *__restrict__ feat, int B, int Hf, int Wf, int C, int pitch, const float *__restrict__ rois5,
                 float scale, int PH, int PW, int sampling_ratio, int aligned, float *__restrict__ out) {
    const int k = blockIdx.x;
    const float *r = rois5 + 5l * k;
    const AlignGeom g = align_geom(r[0], r[1], r[2], r[3], r[4], scale, PH, PW, sampling_ratio, aligned);
    if (g.b < 0 || g.b >= B) return;
    const float *fmap = feat + (long)g.b * Hf * Wf * pitch;
    const int bins = PH * PW, quads = C >> 2;
    for (int t = threadIdx.x; t < quads * bins; t += blockDim.x) {
        const int c4 = t % quads, bin = t / quads;                   // lanes over channel quads: coalesced sample reads
        const float4 sum = align_bin_sum(fmap, Hf, Wf, pitch, c4, g, bin / PW, bin % PW);
        float *o = out + ((long)k * C + 4 * c4) * bins + bin;
        o[0] = sum.x / g.count; o[bins] = sum.y / g.count; o[2 * bins] = sum.z / g.count; o[3 * bins] = sum.w / g.count;
    }
}

// fused with the RoI rescale (nets/classify.py:29-38) and the classifier's mean over the PH*PW bins: out [B*R][out_pitch]
__global__ void __launch_bounds__(256)
roi_align_avg_kernel(const float *__restrict__ feat, int B, int Hf, int Wf, int C, int pitch, const float *__restrict__ rois,
                     const int *__restrict__ roi_indices, int R, float img_h, float img_w, float scale, int PH, int PW,
                     int sampling_ratio, int aligned, float *__restrict__ out, int out_pitch, int roi_in_y) {
    __shared__ float4 rowsum[8][kQuads];
    const int k = roi_in_y ? blockIdx.y : blockIdx.x;      // (channel group fastest: see roi_pool_avg_kernel)
    const float4 rr = reinterpret_cast<const float4 *>(rois)[k];
    const float fx1 = rr.x / img_w * (float)Wf, fy1 = rr.y / img_h * (float)Hf;
    const float fx2 = rr.z / img_w * (float)Wf, fy2 = rr.w / img_h * (float)Hf;
    const AlignGeom g = align_geom((float)roi_indices[k / R], fx1, fy1, fx2, fy2, scale, PH, PW, sampling_ratio, aligned);
    if (g.b < 0 || g.b >= B) return;
    const float *fmap = feat + (long)g.b * Hf * Wf * pitch;
    const float nb = (float)(PH * PW);
    const int q = threadIdx.x % kQuads, slot = threadIdx.x / kQuads;
    const int c4 = (roi_in_y ? blockIdx.x : blockIdx.y) * kQuads + q;
    const bool live = c4 < (C >> 2);
    float4 total = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ph0 = 0; ph0 < PH; ph0 += 8) {
        const int ph = ph0 + slot;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live && ph < PH) {
            for (int pw = 0; pw < PW; ++pw) {
                const float4 s4 = align_bin_sum(fmap, Hf, Wf, pitch, c4, g, ph, pw);
                acc.x += s4.x / g.count; acc.y += s4.y / g.count; acc.z += s4.z / g.count; acc.w += s4.w / g.count;
            }
        }
        rowsum[slot][q] = acc;
        __syncthreads();
        if (slot == 0) {
            const int rows = min(8, PH - ph0);
            for (int r = 0; r < rows; ++r) {
                const float4 v = rowsum[r][q];
                total.x += v.x; total.y += v.y; total.z += v.z; total.w += v.w;
            }
        }
        __syncthreads();
    }
    if (slot == 0 && live)
        *reinterpret_cast<float4 *>(out + (long)k * out_pitch + 4 * c4) =
            make_float4(total.x / nb, total.y / nb, total.z / nb, total.w / nb);
}

}  // namespace

extern "C" int tsod_roi_align_f32(const float *feat, int32_t B, int32_t Hf, int32_t Wf, int32_t C, int32_t feat_pitch,
                                  const float *rois5, int32_t K, float spatial_scale, int32_t PH, int32_t PW,
                                  int32_t sampling_ratio, int32_t aligned, float *out, tsod_stream_t stream) {
    TSOD_REQUIRE(feat && rois5 && out, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(B > 0 && Hf > 0 && Wf > 0 && C > 0 && K > 0 && PH > 0 && PW > 0 && sampling_ratio >= 0, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE((C & 3) == 0 && (feat_pitch & 3) == 0 && feat_pitch >= C && tsod_aligned16(feat), TSOD_ERR_ALIGNMENT);
    hipLaunchKernelGGL(roi_align_kernel, dim3(K), dim3(256), 0, tsod_stream(stream), feat, B, Hf, Wf, C, feat_pitch, rois5,
                       spatial_scale, PH, PW, sampling_ratio, aligned ? 1 : 0, out);
    return tsod_launch_status();
}

extern "C" int tsod_roi_align_avg_f32(const float *feat, int32_t B, int32_t Hf, int32_t Wf, int32_t C, int32_t feat_pitch,
                                      const float *rois, const int32_t *roi_indices, int32_t R, float img_h, float img_w,
                                      float spatial_scale, int32_t PH, int32_t PW, int32_t sampling_ratio, int32_t aligned,
                                      float *out, int32_t out_pitch, tsod_stream_t stream) {
    TSOD_REQUIRE(feat && rois && roi_indices && out, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(B > 0 && Hf > 0 && Wf > 0 && C > 0 && R > 0 && PH > 0 && PW > 0 && PH <= 64 && sampling_ratio >= 0,
                 TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(img_h > 0.f && img_w > 0.f, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE((C & 3) == 0 && (feat_pitch & 3) == 0 && feat_pitch >= C && (out_pitch & 3) == 0 && out_pitch >= C,
                 TSOD_ERR_ALIGNMENT);
    TSOD_REQUIRE(tsod_aligned16(feat) && tsod_aligned16(rois) && tsod_aligned16(out), TSOD_ERR_ALIGNMENT);
    const int quads = C / 4;
    const int groups = (quads + kQuads - 1) / kQuads, roi_in_y = (long)B * R <= 65535 ? 1 : 0;
    hipLaunchKernelGGL(roi_align_avg_kernel, roi_in_y ? dim3(groups, B * R) : dim3(B * R, groups), dim3(256), 0, tsod_stream(stream),
                       feat, B, Hf, Wf, C, feat_pitch, rois, roi_indices, R, img_h, img_w, spatial_scale, PH, PW,
                       sampling_ratio, aligned ? 1 : 0, out, out_pitch, roi_in_y);
    return tsod_launch_status();
}

extern "C" int tsod_roi_pool_f32(const float *feat, int32_t B, int32_t Hf, int32_t Wf, int32_t C, int32_t feat_pitch,
                                 const float *rois5, int32_t K, float spatial_scale, int32_t PH, int32_t PW,
                                 float *out, tsod_stream_t stream) {
    TSOD_REQUIRE(feat && rois5 && out, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(B > 0 && Hf > 0 && Wf > 0 && C > 0 && K > 0 && PH > 0 && PW > 0, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE((C & 3) == 0 && (feat_pitch & 3) == 0 && feat_pitch >= C && tsod_aligned16(feat), TSOD_ERR_ALIGNMENT);
    hipLaunchKernelGGL(roi_pool_kernel, dim3(K), dim3(256), 0, tsod_stream(stream), feat, B, Hf, Wf, C, feat_pitch,
                       rois5, spatial_scale, PH, PW, out);
    return tsod_launch_status();
}

extern "C" int tsod_roi_pool_avg_f32(const float *feat, int32_t B, int32_t Hf, int32_t Wf, int32_t C,
                                     int32_t feat_pitch, const float *rois, const int32_t *roi_indices, int32_t R,
                                     float img_h, float img_w, float spatial_scale, int32_t PH, int32_t PW, float *out,
                                     int32_t out_pitch, tsod_stream_t stream) {
    TSOD_REQUIRE(feat && rois && roi_indices && out, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(B > 0 && Hf > 0 && Wf > 0 && C > 0 && R > 0 && PH > 0 && PW > 0, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(img_h > 0.f && img_w > 0.f, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE((C & 3) == 0 && (feat_pitch & 3) == 0 && feat_pitch >= C && (out_pitch & 3) == 0 && out_pitch >= C,
                 TSOD_ERR_ALIGNMENT);
    TSOD_REQUIRE(tsod_aligned16(feat) && tsod_aligned16(rois) && tsod_aligned16(out), TSOD_ERR_ALIGNMENT);
    const int quads = C / 4;
    const int groups = (quads + kQuads - 1) / kQuads, roi_in_y = (long)B * R <= 65535 ? 1 : 0;
    hipLaunchKernelGGL(roi_pool_avg_kernel, roi_in_y ? dim3(groups, B * R) : dim3(B * R, groups), dim3(256), 0, tsod_stream(stream), feat, B, Hf,
                       Wf, C, feat_pitch, rois, roi_indices, R, img_h, img_w, spatial_scale, PH, PW, out, out_pitch, roi_in_y);
    return tsod_launch_status();
}
