/*
 * oracle/box_ops.c  --  TEST INFRASTRUCTURE ONLY (never imported by the product path).
 *
 * CPU restatement, in plain C, of the two third-party operators the reference's
 * hot path calls but does not contain:
 *
 *   torchvision.ops.nms      called at /root/reference/nets/rpn.py:63
 *   torchvision.ops.RoIPool  built  at /root/reference/nets/classify.py:17,
 *                            called at /root/reference/nets/classify.py:43
 *
 * torchvision is NOT vendored in /root/reference and is not installed in this image;
 * the reference pins no version (no requirements/lock file).  The functions below
 * restate the published torchvision CPU algorithms (ops/cpu/nms_kernel.cpp and
 * ops/cpu/roi_pool_kernel.cpp).  The reference has no test or golden vector at this
 * boundary, so for these two operators the oracle is "PARITY UNPINNED": it is pinned
 * only by the hand-derived known-answer cases in tests/test_oracle_box_ops.py.
 *
 * Also here: bbox_iou (/root/reference/utils/loc_bbox_iou.py:4-27), pinned by the
 * reference's own known answer 0.142857 (loc_bbox_iou.py:100-102), and the
 * ProposalCreator tail (/root/reference/nets/rpn.py:63-69).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; no FMA contraction so every
 * product/sum rounds exactly as the scalar f32 expression is written).
 */
#include <math.h>
#include <float.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- stable descending argsort (torch.sort(stable=True, descending=True)) ---- */
typedef struct { float key; int64_t idx; } kv_t;

static void merge_sort_desc(kv_t *a, kv_t *tmp, int64_t n)
{
    if (n < 2) return;
    int64_t h = n / 2;
    merge_sort_desc(a, tmp, h);
    merge_sort_desc(a + h, tmp, n - h);
    int64_t i = 0, j = h, k = 0;
    while (i < h && j < n) {
        /* take from the right run only when strictly greater: keeps equal keys in index order */
        if (a[j].key > a[i].key) tmp[k++] = a[j++];
        else                     tmp[k++] = a[i++];
    }
    while (i < h) tmp[k++] = a[i++];
    while (j < n) tmp[k++] = a[j++];
    memcpy(a, tmp, (size_t)n * sizeof(kv_t));
}

/* order[k] = index of the k-th largest score, ties -> lower index first. */
int oracle_argsort_desc(const float *scores, int64_t n, int64_t *order)
{
    if (n <= 0) return 0;
    kv_t *a = (kv_t *)malloc((size_t)n * sizeof(kv_t));
    kv_t *t = (kv_t *)malloc((size_t)n * sizeof(kv_t));
    if (!a || !t) { free(a); free(t); return -1; }
    for (int64_t i = 0; i < n; ++i) { a[i].key = scores[i]; a[i].idx = i; }
    merge_sort_desc(a, t, n);
    for (int64_t i = 0; i < n; ++i) order[i] = a[i].idx;
    free(a); free(t);
    return 0;
}

/*
 * torchvision.ops.nms(boxes[N,4] xyxy, scores[N], thr) -> int64 kept indices, in
 * descending-score order.  Greedy; suppress j iff IoU(i,j) > thr (strict).
 * IoU = inter / (area_i + area_j - inter): no +1, no epsilon, 0/0 = NaN never suppresses.
 * Returns the number kept (<= n), or -1 on allocation failure.
 */
int64_t oracle_nms(const float *boxes, const float *scores, int64_t n, float thr, int64_t *keep)
{
    if (n <= 0) return 0;
    int64_t *order = (int64_t *)malloc((size_t)n * sizeof(int64_t));
    float   *area  = (float *)malloc((size_t)n * sizeof(float));
    uint8_t *dead  = (uint8_t *)calloc((size_t)n, 1);
    if (!order || !area || !dead) { free(order); free(area); free(dead); return -1; }
    oracle_argsort_desc(scores, n, order);
    for (int64_t i = 0; i < n; ++i) {
        const float *b = boxes + 4 * i;
        area[i] = (b[2] - b[0]) * (b[3] - b[1]);
    }
    int64_t kept = 0;
    for (int64_t oi = 0; oi < n; ++oi) {
        int64_t i = order[oi];
        if (dead[i]) continue;
        keep[kept++] = i;
        const float ix1 = boxes[4*i], iy1 = boxes[4*i+1], ix2 = boxes[4*i+2], iy2 = boxes[4*i+3];
        const float ia = area[i];
        for (int64_t oj = oi + 1; oj < n; ++oj) {
            int64_t j = order[oj];
            if (dead[j]) continue;
            float xx1 = ix1 > boxes[4*j]   ? ix1 : boxes[4*j];
            float yy1 = iy1 > boxes[4*j+1] ? iy1 : boxes[4*j+1];
            float xx2 = ix2 < boxes[4*j+2] ? ix2 : boxes[4*j+2];
            float yy2 = iy2 < boxes[4*j+3] ? iy2 : boxes[4*j+3];
            float w = xx2 - xx1; if (!(w > 0.f)) w = 0.f;
            float h = yy2 - yy1; if (!(h > 0.f)) h = 0.f;
            float inter = w * h;
            float ovr = inter / (ia + area[j] - inter);
            if (ovr > thr) dead[j] = 1;
        }
    }
    free(order); free(area); free(dead);
    return kept;
}

/*
 * Tail of ProposalCreator.__call__ (/root/reference/nets/rpn.py:63-69) for boxes that
 * are ALREADY sorted by descending score (rpn.py:57-61 did that): nms, then if fewer
 * than n_post survive append indices 0,1,2,... (duplicates of the best pre-NMS boxes,
 * SURVEY Q4), then truncate to n_post.  out_idx[n_post] indexes the sorted list.
 * Returns number kept by NMS before padding, or -2 when the pad would index past n
 * (the reference raises IndexError there).
 */
int64_t oracle_nms_pad(const float *boxes_sorted, int64_t n, float thr, int64_t n_post, int64_t *out_idx)
{
    int64_t *keep = (int64_t *)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
    float *desc = (float *)malloc((size_t)(n > 0 ? n : 1) * sizeof(float));
    if (!keep || !desc) { free(keep); free(desc); return -1; }
    for (int64_t i = 0; i < n; ++i) desc[i] = (float)(n - i);   /* strictly descending keys */
    int64_t kept = oracle_nms(boxes_sorted, desc, n, thr, keep);
    int64_t k = 0;
    for (; k < kept && k < n_post; ++k) out_idx[k] = keep[k];
    int64_t extra = 0;
    int64_t rc = kept;
    for (; k < n_post; ++k, ++extra) {
        if (extra >= n) { rc = -2; out_idx[k] = 0; } else out_idx[k] = extra;
    }
    free(keep); free(desc);
    return rc;
}

/*
 * torchvision.ops.RoIPool((PH,PW), spatial_scale)(x[B,C,H,W], rois[K,5]).
 * rois rows are (batch_index, x1, y1, x2, y2).  Output [K,C,PH,PW].
 * round() is C round (half away from zero); +1 widths; empty bin -> 0; max starts at
 * -FLT_MAX and updates on strict '>'.
 */
int oracle_roi_pool(const float *x, int64_t B, int64_t C, int64_t H, int64_t W,
                    const float *rois, int64_t K, float spatial_scale,
                    int64_t PH, int64_t PW, float *out)
{
    for (int64_t k = 0; k < K; ++k) {
        const float *r = rois + 5 * k;
        int64_t b = (int64_t)r[0];
        if (b < 0 || b >= B) return -1;
        int rsw = (int)roundf(r[1] * spatial_scale);
        int rsh = (int)roundf(r[2] * spatial_scale);
        int rew = (int)roundf(r[3] * spatial_scale);
        int reh = (int)roundf(r[4] * spatial_scale);
        int rw = rew - rsw + 1; if (rw < 1) rw = 1;
        int rh = reh - rsh + 1; if (rh < 1) rh = 1;
        float bin_h = (float)rh / (float)PH;
        float bin_w = (float)rw / (float)PW;
        for (int64_t ph = 0; ph < PH; ++ph) {
            int hs = (int)floorf((float)ph * bin_h);
            int he = (int)ceilf((float)(ph + 1) * bin_h);
            hs += rsh; he += rsh;
            hs = hs < 0 ? 0 : (hs > (int)H ? (int)H : hs);
            he = he < 0 ? 0 : (he > (int)H ? (int)H : he);
            for (int64_t pw = 0; pw < PW; ++pw) {
                int ws = (int)floorf((float)pw * bin_w);
                int we = (int)ceilf((float)(pw + 1) * bin_w);
                ws += rsw; we += rsw;
                ws = ws < 0 ? 0 : (ws > (int)W ? (int)W : ws);
                we = we < 0 ? 0 : (we > (int)W ? (int)W : we);
                int empty = (he <= hs) || (we <= ws);
                for (int64_t c = 0; c < C; ++c) {
                    const float *plane = x + ((b * C + c) * H) * W;
                    float m = empty ? 0.f : -FLT_MAX;
                    for (int h = hs; h < he; ++h)
                        for (int w = ws; w < we; ++w) {
                            float v = plane[(int64_t)h * W + w];
                            if (v > m) m = v;
                        }
                    out[((k * C + c) * PH + ph) * PW + pw] = m;
                }
            }
        }
    }
    return 0;
}

/*
 * torchvision.ops.roi_align(x[B,C,H,W], rois[K,5], (PH,PW), spatial_scale, sampling_ratio, aligned) -> [K,C,PH,PW].
 * Restated from the published torchvision CPU kernel (ops/cpu/roi_align_kernel.cpp + roi_align_common.h,
 * pre_calc_for_bilinear_interpolate).  PARITY UNPINNED: torchvision is absent from /root/reference and from the image and
 * the reference itself uses RoIPool (nets/classify.py:17); this is the checker of the ADDED roi_op="align" option only.
 */
int oracle_roi_align(const float *x, int64_t B, int64_t C, int64_t H, int64_t W, const float *rois, int64_t K,
                     float spatial_scale, int64_t PH, int64_t PW, int64_t sampling_ratio, int aligned, float *out)
{
    for (int64_t k = 0; k < K; ++k) {
        const float *r = rois + 5 * k;
        int64_t b = (int64_t)r[0];
        if (b < 0 || b >= B) return -1;
        float offset = aligned ? 0.5f : 0.f;
        float roi_start_w = r[1] * spatial_scale - offset;
        float roi_start_h = r[2] * spatial_scale - offset;
        float roi_end_w = r[3] * spatial_scale - offset;
        float roi_end_h = r[4] * spatial_scale - offset;
        float roi_width = roi_end_w - roi_start_w;
        float roi_height = roi_end_h - roi_start_h;
        if (!aligned) {
            if (roi_width < 1.f) roi_width = 1.f;
            if (roi_height < 1.f) roi_height = 1.f;
        }
        float bin_size_h = roi_height / (float)PH;
        float bin_size_w = roi_width / (float)PW;
        int grid_h = sampling_ratio > 0 ? (int)sampling_ratio : (int)ceilf(roi_height / (float)PH);
        int grid_w = sampling_ratio > 0 ? (int)sampling_ratio : (int)ceilf(roi_width / (float)PW);
        float count = (float)(grid_h * grid_w > 1 ? grid_h * grid_w : 1);
        for (int64_t c = 0; c < C; ++c) {
            const float *plane = x + ((b * C + c) * H) * W;
            for (int64_t ph = 0; ph < PH; ++ph)
                for (int64_t pw = 0; pw < PW; ++pw) {
                    float acc = 0.f;
                    for (int iy = 0; iy < grid_h; ++iy) {
                        float yy = roi_start_h + (float)ph * bin_size_h + ((float)iy + .5f) * bin_size_h / (float)grid_h;
                        for (int ix = 0; ix < grid_w; ++ix) {
                            float xx = roi_start_w + (float)pw * bin_size_w + ((float)ix + .5f) * bin_size_w / (float)grid_w;
                            float yv = yy, xv = xx;
                            if (yv < -1.f || yv > (float)H || xv < -1.f || xv > (float)W) continue;
                            if (yv <= 0.f) yv = 0.f;
                            if (xv <= 0.f) xv = 0.f;
                            int y_low = (int)yv, x_low = (int)xv, y_high, x_high;
                            if (y_low >= (int)H - 1) { y_high = y_low = (int)H - 1; yv = (float)y_low; } else y_high = y_low + 1;
                            if (x_low >= (int)W - 1) { x_high = x_low = (int)W - 1; xv = (float)x_low; } else x_high = x_low + 1;
                            float ly = yv - (float)y_low, lx = xv - (float)x_low, hy = 1.f - ly, hx = 1.f - lx;
                            float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                            acc += w1 * plane[(int64_t)y_low * W + x_low] + w2 * plane[(int64_t)y_low * W + x_high] +
                                   w3 * plane[(int64_t)y_high * W + x_low] + w4 * plane[(int64_t)y_high * W + x_high];
                        }
                    }
                    out[((k * C + c) * PH + ph) * PW + pw] = acc / count;
                }
        }
    }
    return 0;
}

/* bbox_iou (/root/reference/utils/loc_bbox_iou.py:4-27): dense [Na,Nb], +eps in the denominator. */
void oracle_bbox_iou(const float *a, int64_t na, const float *b, int64_t nb, float eps, float *out)
{
    for (int64_t i = 0; i < na; ++i) {
        float aa = (a[4*i+2] - a[4*i]) * (a[4*i+3] - a[4*i+1]);
        for (int64_t j = 0; j < nb; ++j) {
            float tlx = a[4*i]   > b[4*j]   ? a[4*i]   : b[4*j];
            float tly = a[4*i+1] > b[4*j+1] ? a[4*i+1] : b[4*j+1];
            float brx = a[4*i+2] < b[4*j+2] ? a[4*i+2] : b[4*j+2];
            float bry = a[4*i+3] < b[4*j+3] ? a[4*i+3] : b[4*j+3];
            float w = brx - tlx; if (w < 0.f) w = 0.f;
            float h = bry - tly; if (h < 0.f) h = 0.f;
            float ai = w * h;
            float ab = (b[4*j+2] - b[4*j]) * (b[4*j+3] - b[4*j+1]);
            out[i * nb + j] = ai / (aa + ab - ai + eps);
        }
    }
}
