/*
 * tsod.h -- C ABI of libtsod.so: the MI355X (gfx950) two-stage-detector forward path.
 *
 * Drop-in boundary.  The reference (3SAILab/two_stage_object_detection) is pure Python and has
 * no FFI of its own: its hot path bottoms out in torch / torchvision operators.  Each entry
 * point below replaces the operator(s) named in its comment (file:line in the reference), and
 * is what the reference's Python modules bind through ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions (all entry points)
 *   - plain C, no exceptions; return 0 (TSOD_OK) or a negative tsod_status.
 *   - pointers are RAW DEVICE pointers (hipMalloc / torch tensor.data_ptr()), f32 unless said;
 *     activation and weight pointers must be 16-byte aligned.
 *   - asynchronous and stream-ordered on `stream` (a hipStream_t passed as void*; NULL = default).
 *   - stateless, re-entrant, no device allocation, no host synchronisation: the caller owns
 *     every buffer including workspaces (sizes from the *_workspace_bytes helpers), so every
 *     call is legal inside hipStreamBeginCapture / a torch.cuda.graph.
 *   - activations are NHWC ("pixel-major") f32: element (n,h,w,c) of a tensor with channel
 *     pitch P and channel offset O lives at ((n*H + h)*W + w)*P + O + c.  Pitch/offset let a
 *     conv read or write a channel slice of a wider buffer (HarDNet's concat is free).
 *   - boxes are xyxy pixel coordinates, f32.
 *   - the one exchange of the data-parallel job is an all-gather of [B,300,6] detection records per step.  bench.py and
 *     dist.py issue it through torch.distributed (backend "nccl" = RCCL over xGMI), which owns its communicator;
 *     tsod_allgather_f32 below is the same RCCL call for hosts without torch, RCCL bound at run time.
 */
#ifndef TSOD_H
#define TSOD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TSOD_VERSION 242 /* 0.2.0: conv descriptor grew (precision, second source), in-launch K-slice combine (zeroed
                            ticket area in the workspace), pitched tsod_detections_f32, new entry points;
                            0.2.1: conv tiles fed by LDS-DMA (bf16x3), balanced K schedule (split_k = -2);
                            0.2.2: tsod_allgather_f32 + communicator helpers (RCCL bound at run time);
                            0.3.0: tsod_bbox2loc_f32; a dual-source conv's c2 must be whole K-steps of the tile;
                            0.4.0: range words (conv descriptor grew: amax_in / amax_in2 / amax_out; tsod_absmax_f32 and the
                            *_amax_f32 entry points): the fp16x2 activation scale follows the tensor per forward */

typedef void *tsod_stream_t; /* hipStream_t */

typedef enum tsod_status {
    TSOD_OK = 0,
    TSOD_ERR_INVALID_ARG = -1, /* NULL pointer, non-positive size, inconsistent geometry */
    TSOD_ERR_UNSUPPORTED = -2, /* valid request this build has no kernel for */
    TSOD_ERR_ALIGNMENT = -3,   /* pointer / pitch / offset not aligned as documented */
    TSOD_ERR_WORKSPACE = -4,   /* workspace NULL or too small */
    TSOD_ERR_LAUNCH = -5       /* hipLaunchKernel reported an error */
} tsod_status;

const char *tsod_status_str(int status);
int tsod_version(void);
/* Number of compute units of the current device (used by the tile heuristics); <0 on error. */
int tsod_device_cu_count(void);

/* ------------------------------------------------------------------------------------------
 * Dense convolution / linear as implicit GEMM on v_mfma_f32_32x32x2_f32 (f32 in, f32 acc).
 * Replaces nn.Conv2d(groups=1) + eval nn.BatchNorm2d + nn.PReLU/ReLU6/ReLU [+ residual add]:
 *   models/resnet.py:62-74 (Bottleneck), :21-31 (BasicBlock), :136-138 (stem), :114-116 (downsample)
 *   models/hardnet.py:38-55 (ConvLayer)
 *   nets/rpn.py:86-88,107,111 (loc / score 1x1 convs)      nets/classify.py:13,15,48,50 (nn.Linear)
 *
 *   out[m, n] = act( (sum_k A[m,k] * Wp[n,k]) * scale[n] + shift[n] + residual[m,n] )
 *   m = (img, oh, ow);  k = (kh, kw, ci) over the input SEGMENTS (see below);  n = output channel.
 * ---------------------------------------------------------------------------------------- */
enum { TSOD_ACT_NONE = 0, TSOD_ACT_PRELU = 1, TSOD_ACT_RELU6 = 2, TSOD_ACT_RELU = 3 };
/* workgroup tile (rows x output channels); _W8 = 8 waves (512 threads) instead of 4; _S1 = single LDS stage
 * (half the LDS per workgroup: more workgroups per CU); _K64 = 64-float K-steps (half the barriers per FLOP);
 * _W1 / _W2 = one / two waves per workgroup, each computing a 64x64 block (no or cheap barriers, half the LDS reads per MFMA) */
enum { TSOD_TILE_AUTO = 0, TSOD_TILE_128x128 = 1, TSOD_TILE_128x64 = 2, TSOD_TILE_64x64 = 3, TSOD_TILE_64x128 = 4,
       TSOD_TILE_128x128_W8 = 5, TSOD_TILE_128x64_W8 = 6, TSOD_TILE_256x128_W8 = 7, TSOD_TILE_64x64_S1 = 8,
       TSOD_TILE_128x64_W8_S1 = 9, TSOD_TILE_64x64_S1_K64 = 10, TSOD_TILE_128x64_W8_S1_K64 = 11,
       TSOD_TILE_64x64_W1_S1 = 12, TSOD_TILE_128x64_W2_S1 = 13, TSOD_TILE_128x64_S1 = 14, TSOD_TILE_64x128_S1 = 15,
       TSOD_TILE_128x128_S1 = 16,
       /* bf16x3 only, fed by LDS-DMA (conv_dma_kernel): one channel segment, Cin a multiple of the K stage (16 / 32 floats),
        * KH * KW <= 31 and K / stage + 8 <= 640 (one table entry per K-step of a workgroup's K range); anything else is
        * TSOD_ERR_UNSUPPORTED for these tiles (TSOD_TILE_AUTO then resolves to another tile).  Filters with more than one tap
        * run their K-steps in (32-channel block, tap) order: another f32 summation order than the other tiles, same weights */
       TSOD_TILE_D128x128 = 17, TSOD_TILE_D64x128 = 18, TSOD_TILE_D256x128 = 19,
       TSOD_TILE_D64x128_S2 = 20 /* two ring stages: two workgroups per CU */,
       TSOD_TILE_D128x256 = 21 /* two columns of waves share the activation stage */,
       TSOD_TILE_D128x128_K32 = 22 /* 8 waves (4 along M x 2 along K), 32-float stages, 3-stage ring: two waves per SIMD at one
                                      workgroup per CU - the small-M (batch-1) tile */,
       TSOD_TILE_D192x128 = 23 /* FP16X2 only: 6 waves along M, 16-float stages, 4-stage ring, one workgroup per CU.  A row-tile size
                                  of its own against tile-count cliffs: M = 4200 rows x 1024 channels (layer3's 1x1 expand conv at
                                  batch 1) is 264 tiles of 128 x 128 - one more chip-wave for 8 tiles - and 176 of these */,
       TSOD_TILE_D64x128_K64 = 24 /* FP16X2 only: 8 waves (2 along M x 4 along K), 64-float stages, 3-stage ring, one workgroup per CU:
                                     half the row granularity of D128x128_K32 at the same MFMAs per wave and phase - small-M layers
                                     fill the chip with whole tiles (four K quarters meet in LDS) instead of K-slices that meet in memory */,
       TSOD_TILE_COUNT = 25 };
/* arithmetic of the contraction.  F32: v_mfma_f32_32x32x2_f32 (a k-ordered f32 fma chain).  BF16X3: every f32 operand cut
 * exactly into three bf16 pieces (hi + mid + lo == x), six piece products per k accumulated in f32 on
 * v_mfma_f32_32x32x16_bf16: f32-level accuracy (error ~1.3e-7 of sum|a*b|) at 0.375x the matrix-pipe time; storage,
 * accumulation and epilogue are f32 either way.  Tiles available in BF16X3: 64x64, 64x64_S1, 128x64_W8_S1, 64x64_S1_K64,
 * 128x64_S1, 64x128_S1, 128x128_S1 (TSOD_ERR_UNSUPPORTED for the others). */
enum { TSOD_PREC_F32 = 0, TSOD_PREC_BF16X3 = 1, TSOD_PREC_FP16X2 = 2 };
/* FP16X2 (every BF16X3 tile but TSOD_TILE_D64x128 / _S2): every f32 operand as TWO fp16 pieces of s * x
 * (hi = rne(s x), lo = rne(s x - hi), s a power of two per tensor), THREE piece products per f32 product on
 * v_mfma_f32_32x32x16_f16, f32 accumulation: the f32 kernel's accuracy with half the MFMAs of BF16X3 - while |s x| stays
 * below fp16's 65504 (the CALLER picks desc.a_scale_exp for its activations' range; beyond it the piece products are inf /
 * NaN, reported through desc.range_flag).
 * `w_packed` is then the image made by tsod_pack_conv_weight_fp16x2 with desc.w_scale_exp: [Cout][ceil(K/8)][hi | lo][8]
 * fp16 of 2^w_scale_exp * w, 32 bytes per 8 k.  The accumulators are scaled back by 2^-(a_scale_exp + w_scale_exp) before
 * the epilogue (exact), so scale / shift / residual / activation mean what they mean for the other arithmetics. */
/* With TSOD_PREC_BF16X3 the `w_packed` argument of tsod_conv2d_f32 is the PRE-SPLIT weight image made once by
 * tsod_pack_conv_weight_bf16x3 from the f32 packed weights [Cout][K]: [Cout][ceil(K/8)][hi | mid | lo][8] bf16, 48 bytes per
 * 8 k (tsod_conv_weight_bf16x3_bytes), every weight cut exactly (hi + mid + lo == w).  Activations are split on the fly. */
#define TSOD_MAX_SEGMENTS 16

typedef struct tsod_conv2d_desc {
    int32_t N, H, W;       /* input images, height, width */
    int32_t in_pitch;      /* floats between consecutive input pixels; multiple of 4 */
    int32_t n_seg;         /* 1..TSOD_MAX_SEGMENTS channel segments gathered from each input pixel */
    int32_t seg_off[TSOD_MAX_SEGMENTS]; /* first channel of segment s inside the pixel; multiple of 4 */
    int32_t seg_len[TSOD_MAX_SEGMENTS]; /* channels in segment s; multiple of 4.  Cin = sum(seg_len) */
    int32_t Cout;          /* output channels (any positive value) */
    int32_t out_pitch;     /* floats between consecutive output pixels (>= out_off + Cout) */
    int32_t out_off;       /* first output channel inside the output pixel */
    int32_t KH, KW;        /* filter size */
    int32_t stride;        /* same in h and w */
    int32_t pad_h, pad_w;  /* zero padding (top/left; bottom/right implied by OH/OW) */
    int32_t OH, OW;        /* output height / width */
    int32_t act;           /* TSOD_ACT_* */
    float slope;           /* PReLU negative slope (single-parameter nn.PReLU) */
    int32_t res_pitch;     /* residual pixel pitch (ignored when residual == NULL) */
    int32_t res_off;       /* residual channel offset */
    int32_t tile;          /* TSOD_TILE_*; AUTO = built-in heuristic */
    int32_t split_k;       /* 1 = whole tiles only; S > 1 = every tile cut into S K-slices; -1 = hybrid (full
                              chip-waves of whole tiles, left-over tiles K-sliced to fill the last wave);
                              -2 = balanced (TSOD_TILE_D* only): one workgroup per CU slot, each the same number of
                              K-steps of the tile-major K-step sequence; 0 = built-in cost model chooses */
    int32_t precision;     /* TSOD_PREC_* (0 = F32) */
    /* optional SECOND source (tsod_conv2d_dual_f32; c2 = 0: none).  k in [KH*KW*Cin, KH*KW*Cin + c2) contracts channel
     * in2_off + (k - KH*KW*Cin) of pixel (oh*stride2, ow*stride2) of in2 [N][H2][W2][in2_pitch]: a strided 1x1 tap, i.e. a
     * bottleneck's last 1x1 conv and its projection shortcut (models/resnet.py:70-74 with :114-116) as ONE GEMM
     *   out = act( [y | x_strided] . [W3*s3 | Wd*sd]^T + (b3 + bd) )
     * (the caller folds both BN scales into the stacked weights [Cout][KH*KW*Cin + c2] and adds the shifts).
     * Requires one channel segment and Cin, KH*KW*Cin AND c2 multiples of the tile's K-step (32; 64 for the _K64 tiles; the
     * stage of the TSOD_TILE_D* tiles): K-steps never straddle the sources nor run past c2.  A named tile that does not
     * divide them returns TSOD_ERR_UNSUPPORTED; TSOD_TILE_AUTO only considers tiles that do. */
    int32_t c2, in2_pitch, in2_off, stride2, H2, W2;
    /* TSOD_PREC_FP16X2 only (ignored otherwise): the activations are split as 2^a_scale_exp * x, the weight image holds
     * 2^w_scale_exp * w (the exponent it was packed with) */
    int32_t a_scale_exp, w_scale_exp;
    /* TSOD_PREC_FP16X2 only, optional (NULL: no report): a device int32 that the launch ORs 1 into when a workgroup ends its K
     * loop with a non-finite accumulator - which is what an activation outside the range produces in every output it feeds
     * (and what genuinely non-finite input produces).  The outputs of such a launch are not to be used; clear the word and
     * run the layer with TSOD_PREC_BF16X3 or a smaller a_scale_exp. */
    int32_t *range_flag;
    /* Range words (optional, NULL = off; see "Range words" below).  amax_out: this launch adds the abs-max of the values it
     * stores to the words.  amax_in (and amax_in2 for the second source): TSOD_PREC_FP16X2 only - the launch takes its
     * activation exponent from the words (2^e * absmax < 2^15) instead of a_scale_exp: the scale follows the tensor per forward,
     * so no input range can leave the arithmetic (range_flag then only reports non-finite input). */
    const uint32_t *amax_in, *amax_in2;
    uint32_t *amax_out;
} tsod_conv2d_desc;

/* Range words: the abs-max of an activation tensor, kept by its PRODUCERS for its consumers.  One tensor = TSOD_AMAX_WORDS
 * uint32 words TSOD_AMAX_STRIDE bytes apart (TSOD_AMAX_BYTES in all, 64-byte aligned), each the bit pattern of a non-negative
 * f32; the abs-max is the largest word.  The caller zero-fills the words once per forward (stream-ordered, e.g.
 * hipMemsetAsync) before the first producer runs; every producing launch adds one atomic max per workgroup (spread over the
 * words: a single word would serialise ~11.5 ns per workgroup); several producers may share the words of one buffer (HarDNet's
 * block buffers), a tensor that is a max-pool or RoI-pool of another may share its producer's.  Everything is stream-ordered and
 * capturable; nothing is read by the host. */
#define TSOD_AMAX_WORDS 64
#define TSOD_AMAX_STRIDE 64
#define TSOD_AMAX_BYTES (TSOD_AMAX_WORDS * TSOD_AMAX_STRIDE)
/* zero the words of `n_tensors` consecutive tensors (a memset node under stream capture): once per forward */
int tsod_amax_reset(uint32_t *words, int32_t n_tensors, tsod_stream_t stream);
/* abs-max of n floats into the words (for tensors no libtsod kernel produced: an image handed over in NHWC(4) layout) */
int tsod_absmax_f32(const float *x, int64_t n, uint32_t *amax_out, tsod_stream_t stream);

/* A device word the HOST can read without a device call (serving: one range-flag read per request, nets/.. serving.py).
 * tsod_host_mapped_pointer: the device-side address of page-locked, mapped host memory (hipHostMalloc, torch's pin_memory()) -
 * a query, made once; TSOD_ERR_UNSUPPORTED when the memory is not mapped into the current device.  tsod_word_publish_i32: one
 * thread stores *src_device to that address, stream-ordered and capturable (a kernel node, no blit); the host reads its own
 * memory once the forward's event has completed. */
int tsod_host_mapped_pointer(void *host, void **device);
int tsod_word_publish_i32(const int32_t *src_device, int32_t *dst_mapped, tsod_stream_t stream);

/* Packed weight layout Wp: [Cout][KH][KW][Cin] f32 (k = (kh*KW + kw)*Cin + ci, ci running over
 * the concatenated segments).  tsod_pack_conv_weight_f32 converts torch's [Cout][Cin_src][KH][KW]:
 * Cin >= Cin_src, extra input channels get zero weights (used to pad 3 -> 4 channels in the
 * stems); KW >= KW_src, extra taps on the right get zero weights (the 7x7 stem runs as 7x8). */
int tsod_pack_conv_weight_f32(const float *w_oihw, int32_t Cout, int32_t Cin_src, int32_t KH, int32_t KW_src,
                              int32_t Cin, int32_t KW, float *w_packed, tsod_stream_t stream);

size_t tsod_conv_weight_bf16x3_bytes(int32_t Cout, int32_t K);
int tsod_pack_conv_weight_bf16x3(const float *w_packed /* [Cout][K] f32 */, int32_t Cout, int32_t K, void *w_bf16x3,
                                 tsod_stream_t stream);
size_t tsod_conv_weight_fp16x2_bytes(int32_t Cout, int32_t K);
int tsod_pack_conv_weight_fp16x2(const float *w_packed /* [Cout][K] f32 */, int32_t Cout, int32_t K, int32_t w_scale_exp,
                                 void *w_fp16x2, tsod_stream_t stream);

/* Bytes of workspace tsod_conv2d_f32 needs for this descriptor (0 unless some tile is K-sliced).
 * Workspace contract: 16-byte aligned, private to one stream at a time, layout [one int32 arrival ticket per K-sliced
 * tile, padded to 256 bytes | partial-sum slabs].  K-slices are combined INSIDE the launch: every slice stores its slab
 * write-through, the slice that arrives last at the tile's ticket sums the slabs in slice order (bit-reproducible) and
 * applies the epilogue.  The tickets must be ZERO when a launch starts; every launch leaves them zero, so zero-fill the
 * buffer once when it is allocated and never lend it to another kernel in between. */
size_t tsod_conv2d_workspace_bytes(const tsod_conv2d_desc *d);
/* Resolve TSOD_TILE_AUTO / split_k == 0 to the concrete choice the heuristic makes. */
int tsod_conv2d_resolve(const tsod_conv2d_desc *d, int32_t *tile, int32_t *split_k);

int tsod_conv2d_f32(const tsod_conv2d_desc *d, const float *in, const float *w_packed,
                    const float *scale /* [Cout] or NULL (=1) */, const float *shift /* [Cout] or NULL (=0) */,
                    const float *residual /* or NULL */, float *out,
                    void *workspace, size_t workspace_bytes, tsod_stream_t stream);

/* The same with a second source tensor (desc.c2 > 0, see tsod_conv2d_desc); in2 == NULL iff desc.c2 == 0. */
int tsod_conv2d_dual_f32(const tsod_conv2d_desc *d, const float *in, const float *in2, const float *w_packed,
                         const float *scale, const float *shift, const float *residual, float *out,
                         void *workspace, size_t workspace_bytes, tsod_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * A whole identity-shortcut ResNet bottleneck in ONE launch (models/resnet.py:57-76 with stride 1 and downsample None: the
 * blocks 1.. of `layer1 = _make_layer(64, 3)`, :99):
 *     out = PReLU(BN3(conv1x1(PReLU(BN2(conv3x3(PReLU(BN1(conv1x1(x)))))))) + x)        Cin -> 64 -> 64 -> Cout, Cout == Cin
 * The two 64-channel intermediates never leave the CU (LDS), so the block moves x (+ a one-pixel halo), x again for the
 * residual (an L2 / Infinity-Cache hit) and out, instead of 273 MB per image at 3x800x1333 in three launches.  FP16X2
 * arithmetic (see TSOD_PREC_FP16X2): x is split with the scale of amax_in (or the static a_scale_exp), the intermediates with
 * the scale of each tile's own abs-max.  This build: Cmid == 64, Cin == Cout, both multiples of 64.
 *
 * `wstream`: the three convs' weights as ONE stream of 8 KB steps in consumption order - Cin/32 steps of conv1 (k = 32 s ..),
 * 18 of conv2 (step = (tap kh*3+kw, channel half)), 2 per 64 output channels of conv3 - each step 64 output channels x 32 k of
 * fp16 pieces of 2^w_exp[i] * w (hi = rne(.), lo = rne(2^e w - hi)), stored as the MFMA fragments the kernel's lanes load
 * straight from L2 into registers: [channel block cb (2)][lane (64) = 32 hh + j][chunk c (2)][hi | lo][8 k], where lane (j, hh) of
 * block cb holds output channel base + 32 cb + pi(j), pi(j) = 16 ((j >> 2) & 1) + 4 (j >> 3) + (j & 3) (so that an accumulator
 * lane owns 16 consecutive channels), and k = 16 c + 8 hh ...  `bn`: f32 [s1(64) | b1(64) | s2(64) | b2(64) | s3(Cout) | b3(Cout)],
 * the folded BatchNorm scale / shift of the three convs.  No workspace. */
typedef struct tsod_bottleneck_desc {
    int32_t N, H, W;              /* images, height, width (input and output) */
    int32_t Cin, in_pitch;        /* x: [N][H][W][in_pitch], channels [0, Cin) */
    int32_t Cmid;                 /* 64 */
    int32_t Cout, out_pitch;      /* out: [N][H][W][out_pitch] */
    float slope;                  /* the block's one PReLU slope */
    int32_t w_exp[3];             /* exponents the three convs' weights were scaled with in wstream */
    int32_t a_scale_exp;          /* static exponent for x when amax_in == NULL */
    int32_t projection;           /* 0: identity shortcut (Cout == Cin).  1 (version 242): the block's 1x1 projection shortcut at stride 1
                                   * (models/resnet.py:114-116, layer1's first block) as part of ONE stacked-K GEMM:
                                   *   out = PReLU([y2 | x] . [W3 s3 | Wd sd]^T + (b3 + bd)),   Cin -> 64 -> 64 -> Cout, Cin % 64 == 0
                                   * wstream then carries 2 + Cin / 32 steps per 64 output channels of "conv3" (k = the 64 channels of
                                   * y2, then the Cin channels of x; both BatchNorm scales folded into the weights, ONE exponent
                                   * w_exp[2] for the stacked matrix: tsod_bottleneck_proj_wstream_bytes), bn's s3 is all ones and its
                                   * b3 the two shifts added up; the x chunks of a tile's own pixels are read a second time from L2
                                   * straight into MFMA fragments, there is no residual pass.  (The field sits in what was padding.) */
    int32_t *range_flag;          /* optional, as in tsod_conv2d_desc */
    const uint32_t *amax_in;      /* optional range words of x */
    uint32_t *amax_out;           /* optional range words of out */
} tsod_bottleneck_desc;
size_t tsod_bottleneck_wstream_bytes(int32_t Cin, int32_t Cout);
size_t tsod_bottleneck_proj_wstream_bytes(int32_t Cin, int32_t Cout);    /* desc.projection == 1 */
int tsod_bottleneck_fp16x2(const tsod_bottleneck_desc *d, const float *x, const void *wstream, const float *bn, float *out,
                           tsod_stream_t stream);

/* The ResNet stem in one launch (models/resnet.py:136-139: conv1 7x7 / 2 pad 3, 3 -> 64, no bias; bn1; relu = nn.PReLU (one
 * slope); maxpool 3x3 / 2 pad 1):  out[N][PH][PW][out_pitch] (channels [0, 64)) from the images x, which are either the reference's
 * NCHW [N][3][H][W] or the input step's NHWC with 4 floats per pixel (channel 3 is ignored).  OH = (H - 1) / 2 + 1, PH = (OH - 1) / 2 + 1
 * (same for W).  fp16x2 arithmetic (see TSOD_PREC_FP16X2) with the pixel scale taken per tile from the tile's own input patch: no
 * range words and no pass over the image are needed.  `wfrag` = tsod_stem_wfrag_bytes() bytes: the weights times 2^w_exp as two
 * fp16 pieces in the MFMA fragments the lanes load, [channel block cb (2)][chunk c (14)][hi | lo][lane (64) = 32 hh + j][8 k], where
 * lane (j, hh) holds output channel 32 cb + pi(j) (pi as in tsod_bottleneck_fp16x2) and k = 16 c + 8 hh .. + 7 with
 * k = 32 kh + 4 kw + ci (kw = 7 and ci = 3: zeros).  `bn`: f32 [scale(64) | shift(64)].  amax_out: the pooled map's range words. */
#define TSOD_STEM_NCHW 0
#define TSOD_STEM_NHWC4 1
typedef struct tsod_stem_desc {
    int32_t N, H, W;              /* images, input height, width */
    int32_t in_layout;            /* TSOD_STEM_NCHW / TSOD_STEM_NHWC4 */
    int32_t out_pitch;            /* floats per pooled pixel (>= 64, multiple of 4) */
    float slope;                  /* PReLU slope */
    int32_t w_exp;                /* the weights in wfrag are scaled by 2^w_exp */
    int32_t *range_flag;          /* optional, as in tsod_conv2d_desc (raised by non-finite input) */
    uint32_t *amax_out;           /* optional range words of out */
} tsod_stem_desc;
size_t tsod_stem_wfrag_bytes(void);
int tsod_stem_fp16x2(const tsod_stem_desc *d, const float *x, const void *wfrag, const float *bn, float *out, tsod_stream_t stream);

/* nn.Linear (nets/classify.py:13,15): out[M,N] = in[M,K] @ w[N,K]^T + bias.  K % 4 == 0. */
int tsod_linear_f32(const float *in, int32_t M, int32_t K, int32_t in_pitch, const float *w /* [N][K] */,
                    const float *bias /* [N] or NULL */, int32_t N, float *out, int32_t out_pitch,
                    void *workspace, size_t workspace_bytes, tsod_stream_t stream);
size_t tsod_linear_workspace_bytes(int32_t M, int32_t K, int32_t N);

/* ------------------------------------------------------------------------------------------
 * HBM-bound layer kernels (NHWC).
 * ---------------------------------------------------------------------------------------- */
/* nn.MaxPool2d(3, 2, 1): models/resnet.py:98,139.  C % 4 == 0; OH = (H-1)/2+1, OW = (W-1)/2+1. */
int tsod_maxpool3x3s2_f32(const float *in, int32_t N, int32_t H, int32_t W, int32_t C, int32_t in_pitch,
                          float *out, int32_t out_pitch, tsod_stream_t stream);

/* Depthwise 3x3, pad 1, stride 1|2, + per-channel scale/shift (folded BN or bias) + optional ReLU:
 * models/hardnet.py:21-36 (DWConvLayer) and :193-195 (tail).  w is [3][3][C]; C % 4 == 0. */
int tsod_dwconv3x3_f32(const float *in, int32_t N, int32_t H, int32_t W, int32_t C, int32_t in_pitch, int32_t in_off,
                       const float *w, const float *scale, const float *shift, int32_t stride, int32_t relu,
                       float *out, int32_t out_pitch, int32_t out_off, tsod_stream_t stream);

/* ... the same, adding the abs-max of what it stores to the range words `amax_out` (NULL: exactly tsod_dwconv3x3_f32) */
int tsod_dwconv3x3_amax_f32(const float *in, int32_t N, int32_t H, int32_t W, int32_t C, int32_t in_pitch, int32_t in_off,
                            const float *w, const float *scale, const float *shift, int32_t stride, int32_t relu,
                            float *out, int32_t out_pitch, int32_t out_off, uint32_t *amax_out, tsod_stream_t stream);

/* Grouped 3x3 conv, pad 1, stride 1|2, C -> C channels in `groups` groups + per-channel scale/shift (folded BN) + activation:
 * the conv2 of the ResNeXt bottleneck (models/resnet.py:46-47 with groups = 32, width_per_group = 4; factory :167-172).
 * w is [C][3][3][C/groups]; C/groups must be a multiple of 4.  act / slope as in tsod_conv2d_desc. */
int tsod_gconv3x3_f32(const float *in, int32_t N, int32_t H, int32_t W, int32_t C, int32_t in_pitch, int32_t groups,
                      const float *w, const float *scale, const float *shift, int32_t stride, int32_t act, float slope,
                      float *out, int32_t out_pitch, tsod_stream_t stream);

int tsod_gconv3x3_amax_f32(const float *in, int32_t N, int32_t H, int32_t W, int32_t C, int32_t in_pitch, int32_t groups,
                           const float *w, const float *scale, const float *shift, int32_t stride, int32_t act, float slope,
                           float *out, int32_t out_pitch, uint32_t *amax_out, tsod_stream_t stream);

/* nn.Conv2d(2G, G, 1, groups=G) + bias: models/hardnet.py:196.
 * out[.., g] = w[g][0]*in[.., 2g] + w[g][1]*in[.., 2g+1] + bias[g].  w is [G][2]. */
int tsod_gconv1x1_pair_f32(const float *in, int64_t pixels, int32_t G, int32_t in_pitch, const float *w,
                           const float *bias, float *out, int32_t out_pitch, tsod_stream_t stream);

int tsod_gconv1x1_pair_amax_f32(const float *in, int64_t pixels, int32_t G, int32_t in_pitch, const float *w,
                                const float *bias, float *out, int32_t out_pitch, uint32_t *amax_out, tsod_stream_t stream);

/* Layout changes at the module boundary (the reference's tensors are NCHW).
 * nchw_to_nhwc writes channels [0,C) of each pixel and zero-fills [C, C_pad) (C_pad <= out_pitch). */
int tsod_nchw_to_nhwc_f32(const float *in, int32_t N, int32_t C, int32_t H, int32_t W,
                          float *out, int32_t out_pitch, int32_t C_pad, tsod_stream_t stream);
/* ... adding the image batch's abs-max to the range words `amax_out` (NULL: exactly tsod_nchw_to_nhwc_f32) */
int tsod_nchw_to_nhwc_amax_f32(const float *in, int32_t N, int32_t C, int32_t H, int32_t W,
                               float *out, int32_t out_pitch, int32_t C_pad, uint32_t *amax_out, tsod_stream_t stream);
int tsod_nhwc_to_nchw_f32(const float *in, int32_t N, int32_t C, int32_t H, int32_t W, int32_t in_pitch,
                          int32_t in_off, float *out, tsod_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * RPN proposal path.
 * ---------------------------------------------------------------------------------------- */
/* Fused anchor shift + fg softmax + box decode + clamp + min-size test.  Replaces
 *   utils/basic_anchors.py:27-57 (enumerate_shifted_anchor), nets/rpn.py:115-118 (softmax, fg),
 *   utils/loc_bbox_iou.py:29-61 (loc2bbox), nets/rpn.py:45-54 (clamp, min-size keep).
 * Anchor index a' = (y*Wf + x)*A + a.  locs holds 4 floats per anchor at
 * locs[(img*Hf*Wf + y*Wf + x)*loc_pitch + 4a ..], scores 2 logits (bg, fg) at
 * scores[(...)*score_pitch + 2a ..].  x is clamped to [0, clamp_x], y to [0, clamp_y]
 * (the caller passes img_size[1], img_size[2]: reference quirk Q1 lives in the caller).
 * Outputs: boxes [B][Hf*Wf*A][4]; fg [B][Hf*Wf*A] (softmax probability);
 *          keys [B][Hf*Wf*A] = fg where both sides >= min_size, else -inf;
 *          anchors_out (optional, may be NULL) [Hf*Wf*A][4] the shifted anchors. */
int tsod_rpn_decode_f32(const float *locs, int32_t loc_pitch, const float *scores, int32_t score_pitch,
                        const float *anchor_base /* [A][4] */, int32_t A, int32_t B, int32_t Hf, int32_t Wf,
                        int32_t feat_stride, float clamp_x, float clamp_y, float min_size,
                        float *boxes, float *fg, float *keys, float *anchors_out, tsod_stream_t stream);

/* The same decode for an explicit anchor tensor and ready-made fg scores: the head of
 * ProposalCreator.__call__ (nets/rpn.py:44-54) for one image.  anchor [n][4], loc [n][4], score [n] ->
 * boxes [n][4] (decoded, clamped), keys [n] = score where both sides >= min_size else -inf. */
int tsod_proposal_decode_f32(const float *anchor, const float *loc, const float *score, int64_t n, float clamp_x,
                             float clamp_y, float min_size, float *boxes, float *keys, tsod_stream_t stream);

/* utils/basic_anchors.py:27-57 as a stand-alone op: out [Hf*Wf*A][4] = base[a] + (x*s, y*s, x*s, y*s). */
int tsod_enumerate_anchors_f32(const float *anchor_base, int32_t A, int32_t Hf, int32_t Wf, int32_t feat_stride,
                               float *out, tsod_stream_t stream);

/* utils/loc_bbox_iou.py:29-61 as a stand-alone op: src [n][4] xyxy, loc [n][4] (dx,dy,dw,dh) -> out [n][4]. */
int tsod_loc2bbox_f32(const float *src, const float *loc, int64_t n, float *out, tsod_stream_t stream);

/* utils/loc_bbox_iou.py:63-88 (bbox2loc) as a stand-alone op, the inverse of loc2bbox: src [n][4], dst [n][4] xyxy ->
 * out [n][4] = ((cx_d - cx_s) / w_s, (cy_d - cy_s) / h_s, log(w_d / w_s), log(h_d / h_s)), w_s / h_s floored at f32 eps.
 * The same device function the two target creators below call. */
int tsod_bbox2loc_f32(const float *src, const float *dst, int64_t n, float *out, tsod_stream_t stream);

/* Per-image stable descending top-k.  Replaces torch.argsort(score, descending=True)[:n_pre] and
 * the gathers at nets/rpn.py:56-61.  keys [B][n]; entries equal to -inf are "filtered out" and
 * never selected; ties keep lower index first.  Outputs, per image b:
 *   counts[b]      = n_sel = min(n_pre, #keys > -inf)                       (int32)
 *   idx[b][k]      = source index of the k-th best, k < n_sel; -1 beyond    (int32, [B][n_pre])
 *   boxes_out[b][k]= boxes[b][idx] (zeros beyond n_sel)                     ([B][n_pre][4]) (may be NULL)
 *   keys_out[b][k] = keys[b][idx]  (-inf beyond n_sel)                      ([B][n_pre])    (may be NULL)
 * n_pre <= 16384.  Small problems (B * n^2 <= 3e8): one launch that ranks every key against every key of its image on the
 * whole chip.  Larger: one workgroup per image, LDS radix-select (rows of up to 81920 keys are held in registers after one
 * pass over memory, longer rows are re-read per pass), then a bitonic sort there - or, with scratch (the _ws_ form below),
 * the rank kernel over the selection. */
int tsod_sort_topk_desc_f32(const float *keys, const float *boxes, int32_t B, int32_t n, int32_t n_pre,
                            int32_t *counts, int32_t *idx, float *boxes_out, float *keys_out,
                            tsod_stream_t stream);
/* The same with caller-owned scratch (tsod_sort_topk_workspace_bytes; may be 0): the order of the selection is then
 * computed by RANK on the whole chip (every key counts the keys before it: the composite keys (score, index) are
 * distinct) instead of by a sorting network inside one workgroup per image.  Results are identical. */
size_t tsod_sort_topk_workspace_bytes(int32_t B, int32_t n, int32_t n_pre);
int tsod_sort_topk_desc_ws_f32(const float *keys, const float *boxes, int32_t B, int32_t n, int32_t n_pre,
                               int32_t *counts, int32_t *idx, float *boxes_out, float *keys_out, void *workspace,
                               size_t workspace_bytes, tsod_stream_t stream);

/* Batched greedy NMS on boxes already sorted by descending score + the pad/truncate tail.
 * Replaces torchvision.ops.nms (nets/rpn.py:63) and nets/rpn.py:65-69.
 *   boxes [B][n_max][4], counts[b] <= n_max valid rows per image.
 *   suppress j when IoU(i,j) > thr (strict), IoU = inter / (area_i + area_j - inter).
 *   keep_idx [B][n_post] int32: kept indices in order, then 0,1,2,... padding (quirk Q4).
 *   rois     [B][n_post][4]   : boxes[b][keep_idx]
 *   n_kept   [B] int32        : min(number surviving NMS, n_post) before padding
 *   status   [1] int32 (zeroed by the caller): bit 0 set when the pad ran past counts[b]
 *            (the reference raises IndexError there); the offending rows are zero-filled.
 * workspace: tsod_nms_workspace_bytes(B, n_max). */
size_t tsod_nms_workspace_bytes(int32_t B, int32_t n_max);
int tsod_nms_f32(const float *boxes, const int32_t *counts, int32_t B, int32_t n_max, float iou_thr,
                 int32_t n_post, int32_t *keep_idx, float *rois, int32_t *n_kept, int32_t *status,
                 void *workspace, size_t workspace_bytes, tsod_stream_t stream);

/* Dense pairwise IoU with eps in the denominator: utils/loc_bbox_iou.py:4-27.  out [Na][Nb]. */
int tsod_bbox_iou_f32(const float *a, int32_t Na, const float *b, int32_t Nb, float eps, float *out,
                      tsod_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * RoI head.
 * ---------------------------------------------------------------------------------------- */
/* torchvision.ops.RoIPool((PH,PW), spatial_scale) (nets/classify.py:17,43) on an NHWC feature map.
 *   feat [B][Hf][Wf] pixels with pitch feat_pitch, C channels (C % 4 == 0);
 *   rois5 [K][5] = (batch_index, x1, y1, x2, y2) in feature coordinates before spatial_scale.
 *   out  [K][C][PH][PW]  (torchvision's layout). */
int tsod_roi_pool_f32(const float *feat, int32_t B, int32_t Hf, int32_t Wf, int32_t C, int32_t feat_pitch,
                      const float *rois5, int32_t K, float spatial_scale, int32_t PH, int32_t PW,
                      float *out, tsod_stream_t stream);

/* Fused RoI rescale + index + RoIPool + mean over the PHxPW bins.  Replaces
 *   nets/classify.py:29-38 (rescale: x / img_w * Wf, y / img_h * Hf; row index = roi_indices[b]),
 *   nets/classify.py:43 (RoIPool), models/hardnet.py:203-212 (AdaptiveAvgPool2d(1) + Flatten).
 *   rois [B][R][4] image coordinates; roi_indices [B] int32; out [B*R][C] (pitch out_pitch). */
int tsod_roi_pool_avg_f32(const float *feat, int32_t B, int32_t Hf, int32_t Wf, int32_t C, int32_t feat_pitch,
                          const float *rois, const int32_t *roi_indices, int32_t R, float img_h, float img_w,
                          float spatial_scale, int32_t PH, int32_t PW, float *out, int32_t out_pitch,
                          tsod_stream_t stream);

/* RoIAlign: the added `roi_op="align"` option of the head (the reference builds RoIPool).  Semantics of
 * torchvision.ops.roi_align(input, rois5, (PH,PW), spatial_scale, sampling_ratio, aligned): bilinear samples on a
 * sampling_ratio x sampling_ratio grid per bin (0 = adaptive: ceil(roi extent / P)), averaged; layouts as tsod_roi_pool_f32.
 * tsod_roi_align_avg_f32 is the fused form of tsod_roi_pool_avg_f32 (RoI rescale + index + align + mean over the bins). */
int tsod_roi_align_f32(const float *feat, int32_t B, int32_t Hf, int32_t Wf, int32_t C, int32_t feat_pitch,
                       const float *rois5, int32_t K, float spatial_scale, int32_t PH, int32_t PW, int32_t sampling_ratio,
                       int32_t aligned, float *out, tsod_stream_t stream);
int tsod_roi_align_avg_f32(const float *feat, int32_t B, int32_t Hf, int32_t Wf, int32_t C, int32_t feat_pitch,
                           const float *rois, const int32_t *roi_indices, int32_t R, float img_h, float img_w,
                           float spatial_scale, int32_t PH, int32_t PW, int32_t sampling_ratio, int32_t aligned,
                           float *out, int32_t out_pitch, tsod_stream_t stream);

/* Final detection records (nets/frcnn_training.py:311-319): per RoI the arg-max class
 * over all n_class logits (first max wins), its raw logit, and loc2bbox(roi, loc of that class).
 *   cls_locs [K] rows of 4*n_class floats (row pitch loc_pitch), scores [K] rows of n_class (pitch score_pitch) - both may
 *   be column slices of one wider matrix, as the fused head GEMM writes them - rois [K][4]
 *   -> det [K][6] = (x1,y1,x2,y2,score,class). */
int tsod_detections_f32(const float *cls_locs, int32_t loc_pitch, const float *scores, int32_t score_pitch,
                        const float *rois, int32_t K, int32_t n_class, float *det, tsod_stream_t stream);

/* ---- inference-time filtering of the records (the step after the path) ------------------
 * The reference's demo keeps `nms(boxes_pred, labels_score_pred, iou_threshold=0.1)` over the records of an
 * image, class-agnostic, no score threshold (multi_inference.py:84).  These three calls are that step with the
 * two switches a deployment adds (score threshold / background class, per-class suppression):
 *   1. tsod_detection_keys_f32   keys[t] = score if (score >= score_thresh and class != background_class) else -inf
 *                                (background_class < 0: no class is dropped; NaN scores are dropped)
 *   2. tsod_sort_topk_desc_f32(keys, NULL, B, R, R, counts, idx, NULL, NULL)   stable descending order, -inf rows dropped
 *      tsod_gather_rows_f32      det_sorted[b][r][:] = det[b][idx[b][r]][:]  (zero rows where idx < 0)
 *   3. tsod_detection_nms_f32    greedy NMS over det_sorted [B][R][6] (columns 0-3 box, 5 class), suppress j when
 *                                IoU(i,j) > thr (strict) and - with per_class != 0 - class_i == class_j.
 *                                keep_idx [B][R] int32: surviving rows of det_sorted in score order, -1 after n_kept[b].
 *                                workspace: tsod_nms_workspace_bytes(B, R).  R <= 8192. */
int tsod_detection_keys_f32(const float *det, int64_t n, float score_thresh, int32_t background_class, float *keys,
                            tsod_stream_t stream);
int tsod_gather_rows_f32(const float *src, const int32_t *idx, int32_t B, int32_t n, int32_t m, int32_t C, float *out,
                         tsod_stream_t stream);
int tsod_detection_nms_f32(const float *det_sorted, const int32_t *counts, int32_t B, int32_t R, float iou_thr,
                           int32_t per_class, int32_t *keep_idx, int32_t *n_kept, void *workspace,
                           size_t workspace_bytes, tsod_stream_t stream);

/* ---- training-side box ops ---------------------------------------------------------------
 * The reference's two target creators are deterministic IoU arg-max assignments ("first n by index" sampling); both are
 * restated with their indexing quirks (oracle/targets.py T1-T4).  IoU is utils/loc_bbox_iou.py:4-27 (eps 1e-8), offsets
 * are bbox2loc (utils/loc_bbox_iou.py:63-88).
 *
 * tsod_anchor_targets_f32 = AnchorTargetCreator(n_sample, pos_iou_thresh, neg_iou_thresh, pos_ratio)(bbox, anchor),
 * nets/frcnn_training.py:19-103, with n_pos = int(pos_ratio * n_sample) computed by the caller:
 *   anchor [A][4], bbox [G][4] (G may be 0) ->
 *   label  [A] int64: -1 ignore / 0 negative / 1 positive     loc [A][4]: bbox2loc(anchor, bbox[argmax]) or zeros when
 *   argmax [A] int32: the gt index each anchor is assigned to  there is no positive
 * tsod_proposal_targets_f32 = ProposalTargetCreator(n_sample, pos_ratio, pos_iou_thresh, neg_iou_thresh_high,
 * neg_iou_thresh_low)(roi, bbox, label), nets/frcnn_training.py:105-177, pos_per_image = int(n_sample * pos_ratio):
 *   roi [R][4], bbox [G][4], gt_label [G] int64 -> the first counts[0] = S <= n_sample rows of
 *   sample_roi [n_sample][4], gt_roi_loc [n_sample][4], gt_roi_label [n_sample] int64;
 *   counts [4] int32 = (S, kept positives, kept negatives, status): status 1 = the reference raises IndexError here
 *   (quirk T2: a sampled negative's original index lies beyond the kept list). */
size_t tsod_anchor_targets_workspace_bytes(int32_t A, int32_t G);
int tsod_anchor_targets_f32(const float *anchor, int32_t A, const float *bbox, int32_t G, float pos_iou_thresh,
                            float neg_iou_thresh, int32_t n_pos, int32_t n_sample, float *loc, int64_t *label,
                            int32_t *argmax, void *workspace, size_t workspace_bytes, tsod_stream_t stream);
size_t tsod_proposal_targets_workspace_bytes(int32_t R, int32_t G, int32_t n_sample);
int tsod_proposal_targets_f32(const float *roi, int32_t R, const float *bbox, int32_t G, const int64_t *gt_label,
                              int32_t n_sample, int32_t pos_per_image, float pos_iou_thresh, float neg_iou_thresh_high,
                              float neg_iou_thresh_low, float *sample_roi, float *gt_roi_loc, int64_t *gt_roi_label,
                              int32_t *counts, void *workspace, size_t workspace_bytes, tsod_stream_t stream);

/* ---- input step (the step before the path) ----------------------------------------------
 * dataset/dataloader.py:35-44 + dataset/transform.py:14-17: a decoded RGB image becomes an f32 CHW tensor with
 * values 0..255 and is resized to the detector's fixed size by torchvision v2 Resize, i.e. ATen's antialiased
 * bilinear interpolation (align_corners = false).  Here: u8 HWC in device memory -> f32, resized, already in the
 * layout the backbone reads (NHWC with a zero 4th channel, or NCHW).
 *   tsod_resize_aa_taps(in, out)      taps per output index of one axis (row length of the weight table)
 *   tsod_resize_aa_tables_f32         HOST function: first[out], count[out], weights[out][taps] of one axis in
 *                                     ATen's f32 arithmetic (triangle filter widened by the down-scale factor,
 *                                     normalised per output index); copy the tables to the device once per size pair
 *   tsod_resize_bilinear_aa_u8_f32    out[oy*stride_y + ox*stride_x + c*stride_c] =
 *                                       mul * sum_j wy[oy][j] * (sum_i wx[ox][i] * src[y0+j][x0+i][c]),  c < C;
 *                                     channels C..C_out-1 are written as 0.  Strides in floats:
 *                                     NHWC4 = (4*OW, 4, 1) with C_out = 4;  NCHW plane = (OW, 1, OH*OW) with C_out = C.
 *                                     mul = 1 reproduces the reference (values stay 0..255), 1/255 gives [0,1]. */
int32_t tsod_resize_aa_taps(int32_t in_size, int32_t out_size);
int tsod_resize_aa_tables_f32(int32_t in_size, int32_t out_size, int32_t *first, int32_t *count, float *weights);
int tsod_resize_bilinear_aa_u8_f32(const uint8_t *src, int32_t H, int32_t W, int32_t C, int64_t src_row_bytes,
                                   const int32_t *yfirst, const int32_t *ycount, const float *ywt,
                                   const int32_t *xfirst, const int32_t *xcount, const float *xwt, int32_t OH,
                                   int32_t OW, float mul, float *out, int64_t stride_y, int64_t stride_x,
                                   int64_t stride_c, int32_t C_out, tsod_stream_t stream);

/* ---- collective: thin wrapper over ncclAllGather (RCCL over xGMI) on the caller's stream.
 * `comm` is an ncclComm_t (from tsod_comm_init_rank below, or any communicator the host already owns); every rank sends
 * `count_per_rank` floats and receives n_ranks * count_per_rank in rank order.  Stream-ordered, no host synchronisation.
 * RCCL is bound at run time: TSOD_ERR_UNSUPPORTED when no librccl can be loaded.  The reference has no distributed code
 * (nothing to match); the payload is tsod_detections_f32's fixed-size records. */
int tsod_allgather_f32(void *comm, const float *send, float *recv, size_t count_per_rank, tsod_stream_t stream);
/* communicator helpers for a host without torch.distributed: rank 0 makes the 128-byte id and ships it to the others by any
 * means, then every rank calls init_rank (collective: blocks until all n_ranks have called it) */
int tsod_comm_unique_id(void *id128);
int tsod_comm_init_rank(void **comm, int32_t n_ranks, const void *id128, int32_t rank);
int tsod_comm_destroy(void *comm);

#ifdef __cplusplus
}
#endif
#endif /* TSOD_H */
