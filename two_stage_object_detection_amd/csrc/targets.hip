// targets.hip -- training-side box ops: IoU arg-max assignment of anchors / proposals to ground-truth boxes.
// HBM / latency-bound integer + f32 box work on wave-level reductions, no MFMA.  -ffp-contract=off: the IoU
// expression rounds exactly like utils/loc_bbox_iou.py:17-26 (and tsod_bbox_iou_f32), so every threshold
// decision and every arg-max is the reference's.
//
// Replaces (reference file:line)
//   nets/frcnn_training.py:19-103   AnchorTargetCreator   (_calc_ious :43-68, _create_label :70-103, __call__ :28-41)
//   nets/frcnn_training.py:105-177  ProposalTargetCreator (__call__ :123-177)
//   utils/loc_bbox_iou.py:63-88     bbox2loc
// Both creators are deterministic in the reference ("first n by index"), and both are restated with their indexing
// quirks (oracle/targets.py T1-T4):
//   T1  anchor negatives are only touched when n_neg < 1, and then ALL of them are disabled (len() of a 1-tuple)
//   T2  the kept proposal labels are zeroed at positions equal to the ORIGINAL indices of the sampled negatives;
//       an original index >= the kept length is the reference's IndexError -> status word
//   T4  first maximum wins in both arg-max directions; the last ground-truth box claiming an anchor wins the override
#include "tsod_internal.h"
#include <limits.h>
#include <math.h>

namespace {

constexpr float kF32Eps = 1.1920928955078125e-07f;   // torch.finfo(torch.float32).eps (utils/loc_bbox_iou.py:77)

__device__ __forceinline__ float iou_eps(const float4 A, const float4 B, float eps) {
    const float tlx = fmaxf(A.x, B.x), tly = fmaxf(A.y, B.y);
    const float brx = fminf(A.z, B.z), bry = fminf(A.w, B.w);
    const float w = fmaxf(brx - tlx, 0.f), h = fmaxf(bry - tly, 0.f);
    const float ai = w * h;
    const float aa = (A.z - A.x) * (A.w - A.y);
    const float ab = (B.z - B.x) * (B.w - B.y);
    return ai / (aa + ab - ai + eps);
}

__device__ __forceinline__ float4 bbox2loc_dev(const float4 s, const float4 d) {
    float w = s.z - s.x, h = s.w - s.y;
    const float cx = s.x + 0.5f * w, cy = s.y + 0.5f * h;
    const float bw = d.z - d.x, bh = d.w - d.y;
    const float bcx = d.x + 0.5f * bw, bcy = d.y + 0.5f * bh;
    w = fmaxf(w, kF32Eps);
    h = fmaxf(h, kF32Eps);
    return make_float4((bcx - cx) / w, (bcy - cy) / h, logf(bw / w), logf(bh / h));
}

// row-wise torch.max(ious, dim=1): first maximum wins.  One thread per candidate box; the G ground-truth boxes are
// walked through LDS in chunks of 256.  cand = concat(a [na], b [nb]) (ProposalTargetCreator appends the gt boxes).
__global__ void __launch_bounds__(256)
rowmax_kernel(const float4 *__restrict__ a, int na, const float4 *__restrict__ b, int nb,
              const float4 *__restrict__ gt, int G, float eps, float *__restrict__ max_iou, int *__restrict__ argmax) {
    __shared__ float4 s_gt[256];
    const int n = blockIdx.x * 256 + threadIdx.x;
    const int N = na + nb;
    const float4 box = n < na ? a[n] : (n < N ? b[n - na] : make_float4(0.f, 0.f, 0.f, 0.f));
    float best = G > 0 ? -INFINITY : 0.f;          // len(bbox) == 0: zeros (nets/frcnn_training.py:47-48, 140-143)
    int bi = 0;
    for (int g0 = 0; g0 < G; g0 += 256) {
        __syncthreads();
        if (g0 + (int)threadIdx.x < G) s_gt[threadIdx.x] = gt[g0 + threadIdx.x];
        __syncthreads();
        const int lim = min(256, G - g0);
        for (int j = 0; j < lim; ++j) {
            const float v = iou_eps(box, s_gt[j], eps);
            if (v > best) { best = v; bi = g0 + j; }
        }
    }
    if (n < N) { max_iou[n] = best; argmax[n] = bi; }
}

// column-wise ious.argmax(dim=0): for every gt box the FIRST anchor of maximal IoU.  One workgroup per gt box.
__global__ void __launch_bounds__(256)
colargmax_kernel(const float4 *__restrict__ anchor, int A, const float4 *__restrict__ gt, float eps, int *__restrict__ gt_argmax) {
    __shared__ float s_v[4];
    __shared__ int s_i[4];
    const float4 g = gt[blockIdx.x];
    float best = -INFINITY;
    int bi = INT_MAX;
    for (int n = threadIdx.x; n < A; n += 256) {
        const float v = iou_eps(anchor[n], g, eps);
        if (v > best) { best = v; bi = n; }                          // n ascending per thread: strict > keeps the first
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off);
        const int oi = __shfl_xor(bi, off);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { s_v[threadIdx.x >> 6] = best; s_i[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (s_v[w] > best || (s_v[w] == best && s_i[w] < bi)) { best = s_v[w]; bi = s_i[w]; }
        gt_argmax[blockIdx.x] = bi == INT_MAX ? 0 : bi;
    }
}

// exclusive prefix of one int per thread over a 1024-thread workgroup (wave shuffles + 16 wave totals in LDS);
// returns the exclusive prefix, *total gets the sum
__device__ __forceinline__ int block_exclusive_scan(int v, int *s_wave /* [17] */, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    __syncthreads();
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int w = 0; w < 16; ++w) { const int t = s_wave[w]; s_wave[w] = run; run += t; }
        s_wave[16] = run;
    }
    __syncthreads();
    *total = s_wave[16];
    return s_wave[wave] + inc - v;
}

// _create_label (:70-103) + the override loop of _calc_ious (:61-63).  ONE workgroup: the positive cap is a prefix
// count over all anchors in index order.
__global__ void __launch_bounds__(1024)
anchor_label_kernel(const float *__restrict__ max_iou, int *__restrict__ argmax, const int *__restrict__ gt_argmax, int A, int G,
                    float pos_thr, float neg_thr, int n_pos, int n_sample, long long *__restrict__ label, int *__restrict__ flag) {
    __shared__ int s_wave[17];
    const int tid = threadIdx.x;
    const int chunk = (A + 1023) / 1024;
    const int lo = min(A, tid * chunk), hi = min(A, lo + chunk);
    for (int n = tid; n < A; n += 1024) {
        const float m = max_iou[n];
        long long l = -1;
        if (m < neg_thr) l = 0;
        if (m >= pos_thr) l = 1;
        label[n] = l;
    }
    __threadfence_block();
    __syncthreads();
    for (int g = tid; g < G; g += 1024) {
        const int a = gt_argmax[g];
        label[a] = 1;                                                  // label[gt_argmax_ious] = 1 (:84-85)
        bool last = true;                                              // T4: argmax_ious[gt_argmax_ious[i]] = i in gt order
        for (int g2 = g + 1; g2 < G; ++g2) last = last && gt_argmax[g2] != a;
        if (last) argmax[a] = g;
    }
    __threadfence_block();
    __syncthreads();
    int cnt = 0;
    for (int n = lo; n < hi; ++n) cnt += label[n] == 1;
    int total_pos;
    int rank = block_exclusive_scan(cnt, s_wave, &total_pos);
    if (total_pos > n_pos) {                                           // keep the first n_pos positives by index (:88-93)
        for (int n = lo; n < hi; ++n)
            if (label[n] == 1) { if (rank >= n_pos) label[n] = -1; ++rank; }
    }
    const int pos_length = min(total_pos, n_pos);
    const int n_neg = n_sample - pos_length;
    if (1 > n_neg) {                                                   // T1: len(neg_index) is the tuple's length, 1 (:96-99)
        for (int n = lo; n < hi; ++n)
            if (label[n] == 0) label[n] = -1;
    }
    if (tid == 0) *flag = pos_length > 0 ? 1 : 0;                      // (label > 0).any() (:33)
}

__global__ void __launch_bounds__(256)
anchor_loc_kernel(const float4 *__restrict__ anchor, int A, const float4 *__restrict__ gt, const int *__restrict__ argmax,
                  const int *__restrict__ flag, float4 *__restrict__ loc) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= A) return;
    loc[n] = *flag ? bbox2loc_dev(anchor[n], gt[argmax[n]]) : make_float4(0.f, 0.f, 0.f, 0.f);
}

// ProposalTargetCreator.__call__ after the row maxima: thresholds, "first n by index" selection, gathers, bbox2loc,
// labels and quirk T2.  ONE workgroup (the selection is a prefix count in index order).
__global__ void __launch_bounds__(1024)
proposal_select_kernel(const float4 *__restrict__ roi, int R, const float4 *__restrict__ gt, int G,
                       const long long *__restrict__ gt_label, const float *__restrict__ max_iou, const int *__restrict__ assign,
                       int n_sample, int pos_per_image, float pos_thr, float neg_hi, float neg_lo,
                       float4 *__restrict__ sample_roi, float4 *__restrict__ gt_roi_loc, long long *__restrict__ gt_roi_label,
                       int *__restrict__ neg_orig /* [n_sample] scratch */, int *__restrict__ counts /* [4] */) {
    __shared__ int s_wave[17];
    const int tid = threadIdx.x;
    const int N = R + G;
    const int chunk = (N + 1023) / 1024;
    const int lo = min(N, tid * chunk), hi = min(N, lo + chunk);
    int cp = 0, cn = 0;
    for (int n = lo; n < hi; ++n) {
        const float m = max_iou[n];
        cp += m >= pos_thr;
        cn += (m < neg_hi) && (m >= neg_lo);
    }
    int total_pos, total_neg;
    int rp = block_exclusive_scan(cp, s_wave, &total_pos);
    int rn = block_exclusive_scan(cn, s_wave, &total_neg);
    const int pos_len = min(total_pos, pos_per_image);
    const int neg_len = max(0, min(total_neg, n_sample - pos_len));
    const int S = pos_len + neg_len;
    for (int n = lo; n < hi; ++n) {
        const float m = max_iou[n];
        const bool is_pos = m >= pos_thr, is_neg = (m < neg_hi) && (m >= neg_lo);
        int slots[2] = {-1, -1};
        if (is_pos) { if (rp < pos_len) slots[0] = rp; ++rp; }
        if (is_neg) { if (rn < neg_len) { slots[1] = pos_len + rn; neg_orig[rn] = n; } ++rn; }
        for (int q = 0; q < 2; ++q) {
            const int slot = slots[q];
            if (slot < 0) continue;
            const float4 box = n < R ? roi[n] : gt[n - R];
            sample_roi[slot] = box;
            if (G > 0) {
                const int g = assign[n];
                gt_roi_loc[slot] = bbox2loc_dev(box, gt[g]);
                gt_roi_label[slot] = gt_label[g] + 1;
            } else {
                gt_roi_loc[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
                gt_roi_label[slot] = 0;
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    int bad = 0;
    if (G > 0) {                                                       // T2: gt_roi_label[neg_index] = 0 on the KEPT list
        for (int j = tid; j < neg_len; j += 1024) {
            const int v = neg_orig[j];
            if (v >= S) bad = 1; else gt_roi_label[v] = 0;
        }
    }
    bad = __syncthreads_or(bad);
    if (tid == 0) { counts[0] = S; counts[1] = pos_len; counts[2] = neg_len; counts[3] = bad ? 1 : 0; }
}

size_t align16(size_t v) { return (v + 15) & ~(size_t)15; }

// utils/loc_bbox_iou.py:63-88 as a stand-alone op: one thread per box pair
__global__ void __launch_bounds__(256)
bbox2loc_kernel(const float4 *__restrict__ src, const float4 *__restrict__ dst, long n, float4 *__restrict__ out) {
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long)gridDim.x * blockDim.x)
        out[t] = bbox2loc_dev(src[t], dst[t]);
}

}  // namespace

extern "C" size_t tsod_anchor_targets_workspace_bytes(int32_t A, int32_t G) {
    if (A <= 0 || G < 0) return 0;
    return align16((size_t)A * 4) + align16((size_t)(G > 0 ? G : 1) * 4) + 16;
}

extern "C" int tsod_anchor_targets_f32(const float *anchor, int32_t A, const float *bbox, int32_t G, float pos_iou_thresh,
                                       float neg_iou_thresh, int32_t n_pos, int32_t n_sample, float *loc, int64_t *label,
                                       int32_t *argmax, void *workspace, size_t workspace_bytes, tsod_stream_t stream) {
    TSOD_REQUIRE(anchor && loc && label && argmax && (bbox || G == 0), TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(A > 0 && G >= 0 && n_pos >= 0 && n_sample >= 0, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(tsod_aligned16(anchor) && tsod_aligned16(loc) && (G == 0 || tsod_aligned16(bbox)), TSOD_ERR_ALIGNMENT);
    TSOD_REQUIRE(workspace && workspace_bytes >= tsod_anchor_targets_workspace_bytes(A, G) && tsod_aligned16(workspace),
                 TSOD_ERR_WORKSPACE);
    char *ws = static_cast<char *>(workspace);
    float *max_iou = reinterpret_cast<float *>(ws);
    int *gt_argmax = reinterpret_cast<int *>(ws + align16((size_t)A * 4));
    int *flag = reinterpret_cast<int *>(ws + align16((size_t)A * 4) + align16((size_t)(G > 0 ? G : 1) * 4));
    hipStream_t s = tsod_stream(stream);
    const float4 *a4 = reinterpret_cast<const float4 *>(anchor), *g4 = reinterpret_cast<const float4 *>(bbox);
    hipLaunchKernelGGL(rowmax_kernel, dim3((A + 255) / 256), dim3(256), 0, s, a4, A, (const float4 *)nullptr, 0, g4, G, 1e-8f,
                       max_iou, argmax);
    if (G > 0) hipLaunchKernelGGL(colargmax_kernel, dim3(G), dim3(256), 0, s, a4, A, g4, 1e-8f, gt_argmax);
    hipLaunchKernelGGL(anchor_label_kernel, dim3(1), dim3(1024), 0, s, max_iou, argmax, gt_argmax, A, G, pos_iou_thresh,
                       neg_iou_thresh, n_pos, n_sample, reinterpret_cast<long long *>(label), flag);
    hipLaunchKernelGGL(anchor_loc_kernel, dim3((A + 255) / 256), dim3(256), 0, s, a4, A, g4, argmax, flag,
                       reinterpret_cast<float4 *>(loc));
    return tsod_launch_status();
}

extern "C" size_t tsod_proposal_targets_workspace_bytes(int32_t R, int32_t G, int32_t n_sample) {
    if (R < 0 || G < 0 || R + G <= 0 || n_sample <= 0) return 0;
    return 2 * align16((size_t)(R + G) * 4) + align16((size_t)n_sample * 4);
}

extern "C" int tsod_proposal_targets_f32(const float *roi, int32_t R, const float *bbox, int32_t G, const int64_t *gt_label,
                                         int32_t n_sample, int32_t pos_per_image, float pos_iou_thresh,
                                         float neg_iou_thresh_high, float neg_iou_thresh_low, float *sample_roi,
                                         float *gt_roi_loc, int64_t *gt_roi_label, int32_t *counts, void *workspace,
                                         size_t workspace_bytes, tsod_stream_t stream) {
    TSOD_REQUIRE(sample_roi && gt_roi_loc && gt_roi_label && counts, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(R >= 0 && G >= 0 && R + G > 0 && n_sample > 0 && pos_per_image >= 0, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE((roi || R == 0) && ((bbox && gt_label) || G == 0), TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE((R == 0 || tsod_aligned16(roi)) && (G == 0 || tsod_aligned16(bbox)) && tsod_aligned16(sample_roi) &&
                     tsod_aligned16(gt_roi_loc), TSOD_ERR_ALIGNMENT);
    TSOD_REQUIRE(workspace && workspace_bytes >= tsod_proposal_targets_workspace_bytes(R, G, n_sample) &&
                     tsod_aligned16(workspace), TSOD_ERR_WORKSPACE);
    const int N = R + G;
    char *ws = static_cast<char *>(workspace);
    float *max_iou = reinterpret_cast<float *>(ws);
    int *assign = reinterpret_cast<int *>(ws + align16((size_t)N * 4));
    int *neg_orig = reinterpret_cast<int *>(ws + 2 * align16((size_t)N * 4));
    hipStream_t s = tsod_stream(stream);
    const float4 *r4 = reinterpret_cast<const float4 *>(roi), *g4 = reinterpret_cast<const float4 *>(bbox);
    // roi = torch.cat((roi, bbox)) (:133) is never materialised: candidate n >= R is gt box n - R
    hipLaunchKernelGGL(rowmax_kernel, dim3((N + 255) / 256), dim3(256), 0, s, r4, R, g4, G, g4, G, 1e-8f, max_iou, assign);
    hipLaunchKernelGGL(proposal_select_kernel, dim3(1), dim3(1024), 0, s, r4, R, g4, G,
                       reinterpret_cast<const long long *>(gt_label), max_iou, assign, n_sample, pos_per_image, pos_iou_thresh,
                       neg_iou_thresh_high, neg_iou_thresh_low, reinterpret_cast<float4 *>(sample_roi),
                       reinterpret_cast<float4 *>(gt_roi_loc), reinterpret_cast<long long *>(gt_roi_label), neg_orig, counts);
    return tsod_launch_status();
}

extern "C" int tsod_bbox2loc_f32(const float *src, const float *dst, int64_t n, float *out, tsod_stream_t stream) {
    TSOD_REQUIRE(src && dst && out && n > 0, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(tsod_aligned16(src) && tsod_aligned16(dst) && tsod_aligned16(out), TSOD_ERR_ALIGNMENT);
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(bbox2loc_kernel, dim3(blocks), dim3(256), 0, tsod_stream(stream), reinterpret_cast<const float4 *>(src),
                       reinterpret_cast<const float4 *>(dst), (long)n, reinterpret_cast<float4 *>(out));
    return tsod_launch_status();
}
