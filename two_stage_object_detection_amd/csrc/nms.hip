// nms.hip -- batched greedy NMS on score-sorted boxes + the reference's pad/truncate tail, and the
// dense pairwise IoU op.  Wavefront-level (wave64 ballots / shuffles / scalar bit scans), no MFMA.
//
//   torchvision.ops.nms      (call site nets/rpn.py:63; algorithm = torchvision ops/cpu/nms_kernel.cpp)
//   nets/rpn.py:65-69        pad with indices 0,1,2,... up to n_post, truncate (quirk Q4)
//   utils/loc_bbox_iou.py:4-27  bbox_iou (eps in the denominator)
//
// Phase 1 (nms_mask_kernel, whole chip): mask[b][i][w] bit j = IoU(box i, box 64w+j) > thr, j > i.
// Phase 2 (nms_scan_kernel, one wave per image): walk 64-box blocks in score order; inside a
// block resolve suppression with a scalar bit scan over the diagonal words (readlane), then OR
// the rows of the survivors into the per-lane `removed` words; stop as soon as n_post boxes are
// kept (only keep[:n_post] is ever used).  Compiled with -ffp-contract=off: the IoU expression
// rounds exactly like the scalar f32 code in oracle/box_ops.c.
#include "tsod_internal.h"

namespace {

__device__ __forceinline__ unsigned long long readlane64(unsigned long long v, int lane) {
    const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, lane);
    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), lane);
    return ((unsigned long long)hi << 32) | lo;
}

__device__ __forceinline__ float iou_nms(const float4 a, const float4 b) {
    const float area_a = (a.z - a.x) * (a.w - a.y);
    const float area_b = (b.z - b.x) * (b.w - b.y);
    const float xx1 = fmaxf(a.x, b.x), yy1 = fmaxf(a.y, b.y);
    const float xx2 = fminf(a.z, b.z), yy2 = fminf(a.w, b.w);
    const float w = fmaxf(0.f, xx2 - xx1), h = fmaxf(0.f, yy2 - yy1);
    const float inter = w * h;
    return inter / (area_a + area_b - inter);
}

// Row r of a [n][stride] f32 table as a box: stride 4 is the aligned RPN layout, 6 the detection records.
__device__ __forceinline__ float4 load_box(const float *__restrict__ base, int stride, int r) {
    if (stride == 4) return reinterpret_cast<const float4 *>(base)[r];
    const float *q = base + (long)r * stride;
    return make_float4(q[0], q[1], q[2], q[3]);
}

// LABEL_COL >= 0: column of the row that holds the class; pairs of different classes never suppress each other.
template <int STRIDE, int LABEL_COL>
__global__ void __launch_bounds__(64)
nms_mask_kernel(const float *__restrict__ boxes, const int *__restrict__ counts, int n_max, int words, float thr,
                unsigned long long *__restrict__ mask) {
    constexpr int stride = STRIDE, label_col = LABEL_COL;
    const int cb = blockIdx.x, rb = blockIdx.y, b = blockIdx.z;
    if (cb < rb) return;  // words left of the diagonal are never read
    const int n = counts[b];
    if (rb * 64 >= n) return;  // rows past n are never read either
    __shared__ float4 col[64];
    __shared__ float col_label[64];
    const int lane = threadIdx.x;
    const float *bx = boxes + (long)b * n_max * stride;
    const int cj = cb * 64 + lane;
    col[lane] = cj < n ? load_box(bx, stride, cj) : make_float4(0.f, 0.f, 0.f, 0.f);
    col_label[lane] = (label_col >= 0 && cj < n) ? bx[(long)cj * stride + label_col] : 0.f;
    __syncthreads();
    const int i = rb * 64 + lane;
    unsigned long long bits = 0;
    if (i < n) {
        const float4 me = load_box(bx, stride, i);
        const float my_label = label_col >= 0 ? bx[(long)i * stride + label_col] : 0.f;
        const int jmax = min(64, n - cb * 64);
        for (int j = 0; j < jmax; ++j) {
            const int gj = cb * 64 + j;
            if (gj > i && (label_col < 0 || col_label[j] == my_label) && iou_nms(me, col[j]) > thr) bits |= 1ull << j;
        }
    }
    if (i < n_max) mask[((long)b * n_max + i) * words + cb] = bits;
}

template <int WPL>  // mask words per lane = ceil(words / 64)
__global__ void __launch_bounds__(64)
nms_scan_kernel(const float *__restrict__ boxes, int stride, const int *__restrict__ counts, int n_max, int words,
                int n_post, const unsigned long long *__restrict__ mask, int *__restrict__ keep_idx,
                float *__restrict__ rois, int *__restrict__ n_kept_out, int *__restrict__ status, int pad) {
    extern __shared__ int s_keep[];  // n_post kept indices
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    const int n = counts[b];
    const unsigned long long *mrow = mask + (long)b * n_max * words;
    unsigned long long removed[WPL];
#pragma unroll
    for (int s = 0; s < WPL; ++s) removed[s] = 0ull;

    int total = 0;
    const int nblk = (n + 63) >> 6;
    for (int blk = 0; blk < nblk && total < n_post; ++blk) {
        // removed word of this block lives in lane (blk & 63), slot (blk >> 6)
        unsigned long long cur = 0;
#pragma unroll
        for (int s = 0; s < WPL; ++s) {
            const unsigned long long v = readlane64(removed[s], blk & 63);
            if (s == (blk >> 6)) cur = v;
        }
        const int row = blk * 64 + lane;
        const unsigned long long diag = row < n ? mrow[(long)row * words + blk] : 0ull;
        const int in_blk = min(64, n - blk * 64);
        unsigned long long alive = ~cur & (in_blk == 64 ? ~0ull : ((1ull << in_blk) - 1ull));
        unsigned long long kept = 0;
        while (alive) {  // wave-uniform scalar loop: at most one iteration per kept box
            const int bit = __ffsll((long long)alive) - 1;
            kept |= 1ull << bit;
            alive &= ~(1ull << bit);
            alive &= ~readlane64(diag, bit);
        }
        if ((kept >> lane) & 1ull) {
            const int pos = total + __popcll(kept & ((1ull << lane) - 1ull));
            if (pos < n_post) s_keep[pos] = row;
        }
        total += __popcll(kept);
        if (total >= n_post || kept == 0ull) continue;
        // OR the survivors' rows into the removed words of the later blocks
#pragma unroll
        for (int s = 0; s < WPL; ++s) {
            const int w = lane + 64 * s;
            if (w < words && w > blk) {
                unsigned long long acc = 0;
                const unsigned long long *col = mrow + (long)blk * 64 * words + w;
                if (__popcll(kept) >= 32) {
                    // most of the block survived: sweep all 64 rows, 16 loads in flight
#pragma unroll 16
                    for (int bit = 0; bit < 64; ++bit) {
                        const int r = blk * 64 + bit;
                        const unsigned long long v = r < n ? col[(long)bit * words] : 0ull;
                        acc |= ((kept >> bit) & 1ull) ? v : 0ull;
                    }
                } else {
                    // few survivors: only their rows matter - walk the set bits of `kept` (wave-uniform), eight rows
                    // in flight per round
                    unsigned long long k = kept;
                    while (k) {
                        int b[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) {
                            b[q] = k ? __ffsll((long long)k) - 1 : -1;
                            k &= k - 1;                   // 0 stays 0
                        }
                        unsigned long long v[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = b[q] >= 0 ? col[(long)b[q] * words] : 0ull;
                        acc |= ((v[0] | v[1]) | (v[2] | v[3])) | ((v[4] | v[5]) | (v[6] | v[7]));
                    }
                }
                removed[s] |= acc;
            }
        }
    }
    __syncthreads();
    const int n_kept = min(total, n_post);
    if (lane == 0) n_kept_out[b] = n_kept;
    const float *bx = boxes + (long)b * n_max * stride;
    for (int pos = lane; pos < n_post; pos += 64) {
        int src;
        bool bad = false;
        if (pos < n_kept) {
            src = s_keep[pos];
        } else if (pad) {
            src = pos - n_kept;  // pad with 0,1,2,...  (nets/rpn.py:66-67)
            if (src >= n) { bad = true; src = 0; }
        } else {
            src = -1;            // no padding: rows past n_kept are marked empty
            bad = true;
        }
        keep_idx[(long)b * n_post + pos] = src;
        if (rois != nullptr) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (!bad) v = load_box(bx, stride, src);
            reinterpret_cast<float4 *>(rois)[(long)b * n_post + pos] = v;
        }
        if (bad && pad) atomicOr(status, 1);
    }
}

__global__ void __launch_bounds__(256)
bbox_iou_kernel(const float *__restrict__ a, int Na, const float *__restrict__ bq, int Nb, float eps,
                float *__restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = blockIdx.y;
    if (j >= Nb || i >= Na) return;
    const float4 A = reinterpret_cast<const float4 *>(a)[i];
    const float4 Bx = reinterpret_cast<const float4 *>(bq)[j];
    const float tlx = fmaxf(A.x, Bx.x), tly = fmaxf(A.y, Bx.y);
    const float brx = fminf(A.z, Bx.z), bry = fminf(A.w, Bx.w);
    const float w = fmaxf(brx - tlx, 0.f), h = fmaxf(bry - tly, 0.f);
    const float ai = w * h;
    const float aa = (A.z - A.x) * (A.w - A.y);
    const float ab = (Bx.z - Bx.x) * (Bx.w - Bx.y);
    out[(long)i * Nb + j] = ai / (aa + ab - ai + eps);
}

}  // namespace

extern "C" size_t tsod_nms_workspace_bytes(int32_t B, int32_t n_max) {
    if (B <= 0 || n_max <= 0) return 0;
    const size_t words = (size_t)(n_max + 63) / 64;
    return (size_t)B * (size_t)n_max * words * sizeof(unsigned long long);
}

namespace {
int launch_nms(const float *boxes, int stride, int label_col, const int32_t *counts, int B, int n_max, float iou_thr,
               int n_post, int32_t *keep_idx, float *rois, int32_t *n_kept, int32_t *status, int pad, void *workspace,
               tsod_stream_t stream) {
    const int words = (n_max + 63) / 64;
    unsigned long long *mask = static_cast<unsigned long long *>(workspace);
    hipStream_t s = tsod_stream(stream);
    if (stride == 4)
        hipLaunchKernelGGL((nms_mask_kernel<4, -1>), dim3(words, words, B), dim3(64), 0, s, boxes, counts, n_max, words,
                           iou_thr, mask);
    else if (label_col >= 0)
        hipLaunchKernelGGL((nms_mask_kernel<6, 5>), dim3(words, words, B), dim3(64), 0, s, boxes, counts, n_max, words,
                           iou_thr, mask);
    else
        hipLaunchKernelGGL((nms_mask_kernel<6, -1>), dim3(words, words, B), dim3(64), 0, s, boxes, counts, n_max, words,
                           iou_thr, mask);
    const size_t lds = (size_t)n_post * sizeof(int);
    const int wpl = (words + 63) / 64;
#define TSOD_NMS_SCAN(W)                                                                                                \
    hipLaunchKernelGGL(nms_scan_kernel<W>, dim3(B), dim3(64), lds, s, boxes, stride, counts, n_max, words, n_post, mask, \
                       keep_idx, rois, n_kept, status, pad)
    switch (wpl) {
        case 1: TSOD_NMS_SCAN(1); break;
        case 2: TSOD_NMS_SCAN(2); break;
        case 3: TSOD_NMS_SCAN(3); break;
        default: TSOD_NMS_SCAN(4); break;
    }
#undef TSOD_NMS_SCAN
    return tsod_launch_status();
}
}  // namespace

extern "C" int tsod_nms_f32(const float *boxes, const int32_t *counts, int32_t B, int32_t n_max, float iou_thr,
                            int32_t n_post, int32_t *keep_idx, float *rois, int32_t *n_kept, int32_t *status,
                            void *workspace, size_t workspace_bytes, tsod_stream_t stream) {
    TSOD_REQUIRE(boxes && counts && keep_idx && rois && n_kept && status, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(B > 0 && n_max > 0 && n_post > 0, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(n_max <= 16384 && n_post <= 8192, TSOD_ERR_UNSUPPORTED);
    TSOD_REQUIRE(tsod_aligned16(boxes) && tsod_aligned16(rois), TSOD_ERR_ALIGNMENT);
    TSOD_REQUIRE(workspace != nullptr && workspace_bytes >= tsod_nms_workspace_bytes(B, n_max), TSOD_ERR_WORKSPACE);
    return launch_nms(boxes, 4, -1, counts, B, n_max, iou_thr, n_post, keep_idx, rois, n_kept, status, 1, workspace, stream);
}

extern "C" int tsod_detection_nms_f32(const float *det_sorted, const int32_t *counts, int32_t B, int32_t R, float iou_thr,
                                      int32_t per_class, int32_t *keep_idx, int32_t *n_kept, void *workspace,
                                      size_t workspace_bytes, tsod_stream_t stream) {
    TSOD_REQUIRE(det_sorted && counts && keep_idx && n_kept, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(B > 0 && R > 0, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(R <= 8192, TSOD_ERR_UNSUPPORTED);
    TSOD_REQUIRE(workspace != nullptr && workspace_bytes >= tsod_nms_workspace_bytes(B, R), TSOD_ERR_WORKSPACE);
    return launch_nms(det_sorted, 6, per_class ? 5 : -1, counts, B, R, iou_thr, R, keep_idx, nullptr, n_kept, nullptr, 0,
                      workspace, stream);
}

extern "C" int tsod_bbox_iou_f32(const float *a, int32_t Na, const float *b, int32_t Nb, float eps, float *out,
                                 tsod_stream_t stream) {
    TSOD_REQUIRE(a && b && out, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(Na > 0 && Nb > 0 && Na <= 65535, TSOD_ERR_INVALID_ARG);
    TSOD_REQUIRE(tsod_aligned16(a) && tsod_aligned16(b), TSOD_ERR_ALIGNMENT);
    hipLaunchKernelGGL(bbox_iou_kernel, dim3((Nb + 255) / 256, Na), dim3(256), 0, tsod_stream(stream), a, Na, b, Nb, eps,
                       out);
    return tsod_launch_status();
}
