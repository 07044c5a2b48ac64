// nms.hip -- batched greedy NMS on score-sorted boxes + the reference's pad/truncate tail, and the
// dense pairwise IoU op.  Wavefront-level (wave64 ballots / shuffles / scalar bit scans), no MFMA.
//
//   torchvision.ops.nms      (call site nets/rpn.py:63; algorithm = torchvision ops/cpu/nms_kernel.cpp)
//   nets/rpn.py:65-69        pad with indices 0,1,2,... up to n_post, truncate (quirk Q4)
//   utils/loc_bbox_iou.py:4-27  bbox_iou (eps in the denominator)
//
// Phase 1 (nms_mask_kernel, whole chip): mask[b][i][w] bit j = IoU(box i, box 64w+j) > thr, j > i.
// Phase 2 (nms_scan_kernel, one wave per image): walk 64-box blocks in score order; a block starts
// from the OR of one mask word per earlier survivor (gathered when the block is reached, requested
// one block ahead), resolves suppression inside the block with a scalar bit scan over the diagonal
// words (readlane) unless the diagonal words show no conflict at all; stop as soon as n_post boxes
// are kept (only keep[:n_post] is ever used).  Compiled with -ffp-contract=off: the IoU expression
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
// One workgroup = one 64 x 64 block of pairs, FOUR waves: wave w tests the rows against columns 16 w .. 16 w + 15, so a lane's chain
// is 16 IoUs instead of 64 (the kernel is bound by that chain - one round of resident waves either way: 13.6 -> 7 us at 3000 boxes),
// and the four 16-bit parts of a row's word meet in LDS.  The same expressions on the same operands: bit-exact.
template <int STRIDE, int LABEL_COL>
__global__ void __launch_bounds__(256)
nms_mask_kernel(const float *__restrict__ boxes, const int *__restrict__ counts, int n_max, int words, float thr,
                unsigned long long *__restrict__ mask) {
    constexpr int stride = STRIDE, label_col = LABEL_COL;
    const int cb = blockIdx.x, rb = blockIdx.y, b = blockIdx.z;
    if (cb < rb) return;  // words left of the diagonal are never read
    const int n = counts[b];
    if (rb * 64 >= n) return;  // rows past n are never read either
    __shared__ float4 col[64];
    __shared__ float col_label[64];
    __shared__ unsigned part[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float *bx = boxes + (long)b * n_max * stride;
    if (wave == 0) {
        const int cj = cb * 64 + lane;
        col[lane] = cj < n ? load_box(bx, stride, cj) : make_float4(0.f, 0.f, 0.f, 0.f);
        col_label[lane] = (label_col >= 0 && cj < n) ? bx[(long)cj * stride + label_col] : 0.f;
    }
    const int i = rb * 64 + lane;
    float4 me = make_float4(0.f, 0.f, 0.f, 0.f);
    float my_label = 0.f;
    if (i < n) {
        me = load_box(bx, stride, i);
        if (label_col >= 0) my_label = bx[(long)i * stride + label_col];
    }
    __syncthreads();
    unsigned bits = 0;
    if (i < n) {
        const int jmax = min(64, n - cb * 64);
#pragma unroll 4
        for (int jj = 0; jj < 16; ++jj) {
            const int j = 16 * wave + jj, gj = cb * 64 + j;
            if (j < jmax && gj > i && (label_col < 0 || col_label[j] == my_label) && iou_nms(me, col[j]) > thr) bits |= 1u << jj;
        }
    }
    part[wave][lane] = bits;
    __syncthreads();
    if (wave == 0 && i < n_max)
        mask[((long)b * n_max + i) * words + cb] = (unsigned long long)part[0][lane] | ((unsigned long long)part[1][lane] << 16) |
                                                   ((unsigned long long)part[2][lane] << 32) | ((unsigned long long)part[3][lane] << 48);
}

__device__ __forceinline__ unsigned long long wave_or64(unsigned long long v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)v, off);
        const unsigned hi = __shfl_xor((unsigned)(v >> 32), off);
        v |= ((unsigned long long)hi << 32) | lo;
    }
    return v;
}

// Phase 2, one wave per image.  Block blk (64 boxes in score order) needs ONE mask word per earlier survivor: the word of
// column block blk in that survivor's row.  They are gathered when the block is reached - lane l takes survivors l, l + 64, ...
// of the keep list (at most n_post of them, so a handful of loads per lane) - instead of OR-ing every survivor's whole row
// into per-lane `removed` words as soon as it is kept (rows x words loads, most of them for blocks the scan never reaches
// once n_post boxes are kept).  Everything block blk + 1 needs is requested while block blk is being resolved:
//   its diagonal word per row, the words of the survivors known so far, and - unconditionally - the words of all 64 rows
//   of block blk (masked by that block's keep bits afterwards), so no load depends on the resolution it overlaps.
// A block whose surviving rows do not suppress each other at all (one wave-wide OR of the diagonal words tells) keeps every
// alive box without the bit-by-bit scan; otherwise the scalar scan walks the kept boxes as before.  Same greedy result.
__global__ void __launch_bounds__(64)
nms_scan_kernel(const float *__restrict__ boxes, int stride, const int *__restrict__ counts, int n_max, int words,
                int n_post, const unsigned long long *__restrict__ mask, int *__restrict__ keep_idx,
                float *__restrict__ rois, int *__restrict__ n_kept_out, int *__restrict__ status, int pad) {
    extern __shared__ int s_keep[];  // n_post kept indices
    const int lane = threadIdx.x;
    const int b = blockIdx.x;
    const int n = counts[b];
    const unsigned long long *mrow = mask + (long)b * n_max * words;
    int total = 0;
    const int nblk = (n + 63) >> 6;
    // state of the block about to be resolved
    unsigned long long diag = lane < n ? mrow[(long)lane * words] : 0ull;     // block 0: its diagonal words; nothing removed yet
    unsigned long long cur = 0ull;
    for (int blk = 0; blk < nblk && total < n_post; ++blk) {
        const int row = blk * 64 + lane;
        // ---- requests for block blk + 1 (none of them depends on this block's outcome)
        const bool more = blk + 1 < nblk;
        unsigned long long nx_diag = 0ull, nx_rows = 0ull, nx_pre = 0ull;
        if (more) {
            const int nrow = row + 64;
            nx_diag = nrow < n ? mrow[(long)nrow * words + blk + 1] : 0ull;
            nx_rows = row < n ? mrow[(long)row * words + blk + 1] : 0ull;
            for (int i = lane; i < min(total, n_post); i += 64) nx_pre |= mrow[(long)s_keep[i] * words + blk + 1];
        }
        // ---- resolve this block
        const int in_blk = min(64, n - blk * 64);
        unsigned long long alive = ~cur & (in_blk == 64 ? ~0ull : ((1ull << in_blk) - 1ull));
        unsigned long long kept;
        const unsigned long long conflicts = wave_or64(((alive >> lane) & 1ull) ? (diag & alive) : 0ull);
        if (conflicts == 0ull) {
            kept = alive;                              // no alive box of the block suppresses another one
        } else {
            kept = 0ull;
            while (alive) {                            // wave-uniform scalar loop: one iteration per kept box
                const int bit = __ffsll((long long)alive) - 1;
                kept |= 1ull << bit;
                alive &= ~(1ull << bit);
                alive &= ~readlane64(diag, bit);
            }
        }
        if ((kept >> lane) & 1ull) {
            const int pos = total + __popcll(kept & ((1ull << lane) - 1ull));
            if (pos < n_post) s_keep[pos] = row;
        }
        total += __popcll(kept);
        // ---- what the next block starts from
        if (more) cur = wave_or64(nx_pre | (((kept >> lane) & 1ull) ? nx_rows : 0ull));
        diag = nx_diag;
        __syncthreads();                               // (one wave: orders the keep list's LDS writes before the next gather)
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
        hipLaunchKernelGGL((nms_mask_kernel<4, -1>), dim3(words, words, B), dim3(256), 0, s, boxes, counts, n_max, words,
                           iou_thr, mask);
    else if (label_col >= 0)
        hipLaunchKernelGGL((nms_mask_kernel<6, 5>), dim3(words, words, B), dim3(256), 0, s, boxes, counts, n_max, words,
                           iou_thr, mask);
    else
        hipLaunchKernelGGL((nms_mask_kernel<6, -1>), dim3(words, words, B), dim3(256), 0, s, boxes, counts, n_max, words,
                           iou_thr, mask);
    const size_t lds = (size_t)n_post * sizeof(int);
    hipLaunchKernelGGL(nms_scan_kernel, dim3(B), dim3(64), lds, s, boxes, stride, counts, n_max, words, n_post, mask, keep_idx,
                       rois, n_kept, status, pad);
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
