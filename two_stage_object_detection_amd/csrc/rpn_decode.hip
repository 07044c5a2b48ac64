// rpn_decode.hip -- fused anchor shift + fg softmax + loc2bbox + clamp + min-size test, and the
// final per-RoI detection record kernel.  HBM/latency-bound elementwise work, no MFMA.
//
// Follows (op for op, so every f32 rounding sits where the reference's does; this file is
// compiled with -ffp-contract=off):
//   utils/basic_anchors.py:27-57   anchor[(y*W+x)*A+a] = base[a] + (x*s, y*s, x*s, y*s)
//   nets/rpn.py:115-118            softmax over the (bg, fg) logit pair, take fg
//   utils/loc_bbox_iou.py:29-61    loc2bbox
//   nets/rpn.py:47-54              clamp x to [0,img_size[1]], y to [0,img_size[2]]; min-size keep
#include "tsod_internal.h"
#include <math.h>

namespace {

struct Box { float x1, y1, x2, y2; };

__device__ __forceinline__ Box decode_box(float ax1, float ay1, float ax2, float ay2,
                                          float dx, float dy, float dw, float dh) {
    const float w = ax2 - ax1;
    const float h = ay2 - ay1;
    const float cx = ax1 + 0.5f * w;
    const float cy = ay1 + 0.5f * h;
    const float ncx = dx * w + cx;
    const float ncy = dy * h + cy;
    const float nw = expf(dw) * w;
    const float nh = expf(dh) * h;
    Box o;
    o.x1 = ncx - 0.5f * nw;
    o.y1 = ncy - 0.5f * nh;
    o.x2 = ncx + 0.5f * nw;
    o.y2 = ncy + 0.5f * nh;
    return o;
}

__device__ __forceinline__ float clampf(float v, float lo, float hi) { return fminf(fmaxf(v, lo), hi); }

__global__ void __launch_bounds__(256)
rpn_decode_kernel(const float *__restrict__ locs, int loc_pitch, const float *__restrict__ scores, int score_pitch,
                  const float *__restrict__ anchor_base, int A, int B, int Hf, int Wf, int feat_stride,
                  float clamp_x, float clamp_y, float min_size,
                  float *__restrict__ boxes, float *__restrict__ fg, float *__restrict__ keys,
                  float *__restrict__ anchors_out) {
    const int HW = Hf * Wf;
    const long per_img = (long)HW * A;
    const long total = per_img * B;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int b = (int)(t / per_img);
        const int r = (int)(t - (long)b * per_img);
        const int pix = r / A;
        const int a = r - pix * A;
        const int y = pix / Wf;
        const int x = pix - y * Wf;
        const float sx = (float)(x * feat_stride);
        const float sy = (float)(y * feat_stride);
        const float ax1 = anchor_base[4 * a + 0] + sx;
        const float ay1 = anchor_base[4 * a + 1] + sy;
        const float ax2 = anchor_base[4 * a + 2] + sx;
        const float ay2 = anchor_base[4 * a + 3] + sy;
        if (anchors_out != nullptr && b == 0) {
            reinterpret_cast<float4 *>(anchors_out)[r] = make_float4(ax1, ay1, ax2, ay2);
        }
        const float *l = locs + ((long)b * HW + pix) * loc_pitch + 4 * a;
        Box o = decode_box(ax1, ay1, ax2, ay2, l[0], l[1], l[2], l[3]);
        o.x1 = clampf(o.x1, 0.f, clamp_x);
        o.x2 = clampf(o.x2, 0.f, clamp_x);
        o.y1 = clampf(o.y1, 0.f, clamp_y);
        o.y2 = clampf(o.y2, 0.f, clamp_y);
        const float *s = scores + ((long)b * HW + pix) * score_pitch + 2 * a;
        const float s0 = s[0], s1 = s[1];
        const float m = fmaxf(s0, s1);
        const float e0 = expf(s0 - m), e1 = expf(s1 - m);
        const float p = e1 / (e0 + e1);
        const bool ok = ((o.x2 - o.x1) >= min_size) && ((o.y2 - o.y1) >= min_size);
        reinterpret_cast<float4 *>(boxes)[t] = make_float4(o.x1, o.y1, o.x2, o.y2);
        fg[t] = p;
        keys[t] = ok ? p : -INFINITY;
    }
}

__global__ void __launch_bounds__(256)
enumerate_anchors_kernel(const float *__restrict__ anchor_base, int A, int Hf, int Wf, int feat_stride,
                         float *__restrict__ out) {
    const long total = (long)Hf * Wf * A;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int pix = (int)(t / A);
        const int a = (int)(t - (long)pix * A);
        const int y = pix / Wf, x = pix - y * Wf;
        const float sx = (float)(x * feat_stride), sy = (float)(y * feat_stride);
        reinterpret_cast<float4 *>(out)[t] = make_float4(anchor_base[4 * a] + sx, anchor_base[4 * a + 1] + sy,
                                                         anchor_base[4 * a + 2] + sx, anchor_base[4 * a + 3] + sy);
    }
}

__global__ void __launch_bounds__(256)
loc2bbox_kernel(const float *__restrict__ src, const float *__restrict__ loc, long n, float *__restrict__ out) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
        const float4 s = reinterpret_cast<const float4 *>(src)[t];
        const float4 l = reinterpret_cast<const float4 *>(loc)[t];
        const Box o = decode_box(s.x, s.y, s.z, s.w, l.x, l.y, l.z, l.w);
        reinterpret_cast<float4 *>(out)[t] = make_float4(o.x1, o.y1, o.x2, o.y2);
    }
}

__global__ void __launch_bounds__(256)
proposal_decode_kernel(const float *__restrict__ anchor, const float *__restrict__ loc, const float *__restrict__ score,
                       long n, float clamp_x, float clamp_y, float min_size, float *__restrict__ boxes,
                       float *__restrict__ keys) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
        const float4 a = reinterpret_cast<const float4 *>(anchor)[t];
        const float4 l = reinterpret_cast<const float4 *>(loc)[t];
        Box o = decode_box(a.x, a.y, a.z, a.w, l.x, l.y, l.z, l.w);
        o.x1 = clampf(o.x1, 0.f, clamp_x);
        o.x2 = clampf(o.x2, 0.f, clamp_x);
        o.y1 = clampf(o.y1, 0.f, clamp_y);
        o.y2 = clampf(o.y2, 0.f, clamp_y);
        const bool ok = ((o.x2 - o.x1) >= min_size) && ((o.y2 - o.y1) >= min_size);
        reinterpret_cast<float4 *>(boxes)[t] = make_float4(o.x1, o.y1, o.x2, o.y2);
        keys[t] = ok ? score[t] : -INFINITY;
    }
}

// One wave per RoI: arg-max over n_class logits (first maximum wins), then loc2bbox with the
// 4 offsets of that class (nets/frcnn_training.py:311-319).
__global__ void __launch_bounds__(256)
detections_kernel(const float *__restrict__ cls_locs, int loc_pitch, const float *__restrict__ scores, int score_pitch,
                  const float *__restrict__ rois, int K, int n_class, float *__restrict__ det) {
    const int lane = threadIdx.x & 63;
    const int k = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (k >= K) return;
    const float *s = scores + (long)k * score_pitch;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < n_class; c += 64) {
        const float v = s[c];
        if (bi == 0x7fffffff || v > best) { best = v; bi = c; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off);
        const int oi = __shfl_xor(bi, off);
        if (oi != 0x7fffffff && (bi == 0x7fffffff || ov > best || (ov == best && oi < bi))) { best = ov; bi = oi; }
    }
    if (lane == 0) {
        const float *l = cls_locs + (long)k * loc_pitch + bi * 4;
        const float *r = rois + (long)k * 4;
        Box o = decode_box(r[0], r[1], r[2], r[3], l[0], l[1], l[2], l[3]);
        float *d = det + (long)k * 6;
        d[0] = o.x1; d[1] = o.y1; d[2] = o.x2; d[3] = o.y2; d[4] = best; d[5] = (float)bi;
    }
}

// Sort keys of the final decode: the record's score where it passes the threshold and is not the background class,
// -inf (= dropped by tsod_sort_topk_desc_f32) otherwise.  NaN scores fail the comparison and are dropped.
__global__ void __launch_bounds__(256)
detection_keys_kernel(const float *__restrict__ det, long n, float score_thresh, int background_class,
                      float *__restrict__ keys) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x) {
        const float s = det[t * 6 + 4];
        const float c = det[t * 6 + 5];
        const bool ok = s >= score_thresh && (background_class < 0 || c != (float)background_class);
        keys[t] = ok ? s : -INFINITY;
    }
}

// out[b][r][:] = src[b][idx[b][r]][:], zero rows where idx < 0.
__global__ void __launch_bounds__(256)
gather_rows_kernel(const float *__restrict__ src, const int *__restrict__ idx, int B, int n, int m, int C,
                   float *__restrict__ out) {
    const long total = (long)B * m * C;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long)gridDim.x * blockDim.x) {
        const int c = (int)(t % C);
        const long br = t / C;
        const int b = (int)(br / m);
        const int i = idx[br];
        out[t] = (i >= 0 && i < n) ? src[((long)b * n + i) * C + c] : 0.f;
    }
}

}  // namespace

extern "C" int tsod_rpn_decode_f32(const float *locs, int32_t loc_pitch, const float *scores, int32_t score_pitch,
                                   const float *anchor_base, int32_t A, int32_t B, int32_t Hf, int32_t Wf,
                                   int32_t feat_stride, float clamp_x, float clamp_y, float min_size,
                                   float *boxes, float *fg, float *keys, float *anchors_out, tsod_stream_t stream) {
    TSOD_REQUIRE(locs && scores && anchor_base && boxes && fg && keys, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(A > 0 && B > 0 && Hf > 0 && Wf > 0 && feat_stride > 0, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(loc_pitch >= 4 * A && score_pitch >= 2 * A, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(tsod_aligned16(boxes) && (anchors_out == nullptr || tsod_aligned16(anchors_out)), TSOD_ERR_ALIGNMENT);
    const long total = (long)B * Hf * Wf * A;
    const int threads = 256;
    const int blocks = (int)((total + threads - 1) / threads < 4096 ? (total + threads - 1) / threads : 4096);
    hipLaunchKernelGGL(rpn_decode_kernel, dim3(blocks), dim3(threads), 0, tsod_stream(stream), locs, loc_pitch, scores,
                       score_pitch, anchor_base, A, B, Hf, Wf, feat_stride, clamp_x, clamp_y, min_size, boxes, fg, keys,
                       anchors_out);
    return tsod_launch_status();
}

extern "C" int tsod_detections_f32(const float *cls_locs, int32_t loc_pitch, const float *scores, int32_t score_pitch,
                                   const float *rois, int32_t K, int32_t n_class, float *det, tsod_stream_t stream) {
    TSOD_REQUIRE(cls_locs && scores && rois && det, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(K > 0 && n_class > 0 && loc_pitch >= 4 * n_class && score_pitch >= n_class, TSOD_ERR_INVALID_ARG);
    const int waves_per_block = 4;
    hipLaunchKernelGGL(detections_kernel, dim3((K + waves_per_block - 1) / waves_per_block), dim3(64 * waves_per_block),
                       0, tsod_stream(stream), cls_locs, loc_pitch, scores, score_pitch, rois, K, n_class, det);
    return tsod_launch_status();
}

extern "C" int tsod_enumerate_anchors_f32(const float *anchor_base, int32_t A, int32_t Hf, int32_t Wf,
                                          int32_t feat_stride, float *out, tsod_stream_t stream) {
    TSOD_REQUIRE(anchor_base && out, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(A > 0 && Hf > 0 && Wf > 0 && feat_stride > 0, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(tsod_aligned16(out), TSOD_ERR_ALIGNMENT);
    const long total = (long)Hf * Wf * A;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(enumerate_anchors_kernel, dim3(blocks), dim3(256), 0, tsod_stream(stream), anchor_base, A, Hf, Wf,
                       feat_stride, out);
    return tsod_launch_status();
}

extern "C" int tsod_loc2bbox_f32(const float *src, const float *loc, int64_t n, float *out, tsod_stream_t stream) {
    TSOD_REQUIRE(src && loc && out && n > 0, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(tsod_aligned16(src) && tsod_aligned16(loc) && tsod_aligned16(out), TSOD_ERR_ALIGNMENT);
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(loc2bbox_kernel, dim3(blocks), dim3(256), 0, tsod_stream(stream), src, loc, (long)n, out);
    return tsod_launch_status();
}

extern "C" int tsod_proposal_decode_f32(const float *anchor, const float *loc, const float *score, int64_t n,
                                        float clamp_x, float clamp_y, float min_size, float *boxes, float *keys,
                                        tsod_stream_t stream) {
    TSOD_REQUIRE(anchor && loc && score && boxes && keys && n > 0, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(tsod_aligned16(anchor) && tsod_aligned16(loc) && tsod_aligned16(boxes), TSOD_ERR_ALIGNMENT);
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(proposal_decode_kernel, dim3(blocks), dim3(256), 0, tsod_stream(stream), anchor, loc, score,
                       (long)n, clamp_x, clamp_y, min_size, boxes, keys);
    return tsod_launch_status();
}

extern "C" int tsod_detection_keys_f32(const float *det, int64_t n, float score_thresh, int32_t background_class,
                                       float *keys, tsod_stream_t stream) {
    TSOD_REQUIRE(det && keys && n > 0, TSOD_ERR_INVALID_ARG);
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(detection_keys_kernel, dim3(blocks), dim3(256), 0, tsod_stream(stream), det, (long)n, score_thresh,
                       background_class, keys);
    return tsod_launch_status();
}

extern "C" int tsod_gather_rows_f32(const float *src, const int32_t *idx, int32_t B, int32_t n, int32_t m, int32_t C,
                                    float *out, tsod_stream_t stream) {
    TSOD_REQUIRE(src && idx && out, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(B > 0 && n > 0 && m > 0 && C > 0, TSOD_ERR_INVALID_ARG);
    const long total = (long)B * m * C;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(gather_rows_kernel, dim3(blocks), dim3(256), 0, tsod_stream(stream), src, idx, B, n, m, C, out);
    return tsod_launch_status();
}
