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
                    float img_h, float img_w, float scale, int PH, int PW, float *__restrict__ out, int out_pitch) {
    __shared__ float4 rowsum[8][kQuads];
    const int k = blockIdx.x;
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
    const int c4 = blockIdx.y * kQuads + q;
    const bool live = c4 < (C >> 2);
    float4 total = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int ph0 = 0; ph0 < PH; ph0 += 8) {                                  // PH <= 8: one pass
        const int ph = ph0 + slot;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live && ph < PH) {
            int hs, he;
            bin_range(ph, bin_h, g.sh, Hf, hs, he);
            for (int pw = 0; pw < PW; ++pw) {
                int ws, we;
                bin_range(pw, bin_w, g.sw, Wf, ws, we);
                const float4 m = bin_max(fmap, Wf, pitch, c4, hs, he, ws, we);
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

}  // namespace

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
    hipLaunchKernelGGL(roi_pool_avg_kernel, dim3(B * R, (quads + kQuads - 1) / kQuads), dim3(256), 0, tsod_stream(stream), feat, B, Hf, Wf, C,
                       feat_pitch, rois, roi_indices, R, img_h, img_w, spatial_scale, PH, PW, out, out_pitch);
    return tsod_launch_status();
}
