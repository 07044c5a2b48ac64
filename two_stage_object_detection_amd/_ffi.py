"""ctypes binding of include/tsod.h (libtsod.so, HIP, gfx950).

This is the only way the package reaches compute: there is no CPU or PyTorch-eager fallback.
If the shared library is missing the import of any compute entry point raises immediately.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, byref, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

import torch

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TSOD_LIB") or os.path.join(_PKG_DIR, "libtsod.so")   # TSOD_LIB: alternate build (experiments)

TSOD_MAX_SEGMENTS = 16
AMAX_WORDS, AMAX_STRIDE = 64, 64                 # range words of one tensor (include/tsod.h): 64 uint32, 64 bytes apart
AMAX_BYTES = AMAX_WORDS * AMAX_STRIDE
ACT_NONE, ACT_PRELU, ACT_RELU6, ACT_RELU = 0, 1, 2, 3
TILE_AUTO, TILE_128x128, TILE_128x64, TILE_64x64, TILE_64x128 = 0, 1, 2, 3, 4
TILE_NAMES = {0: "auto", 1: "128x128", 2: "128x64", 3: "64x64", 4: "64x128", 5: "128x128w8", 6: "128x64w8", 7: "256x128w8",
              8: "64x64s1", 9: "128x64w8s1", 10: "64x64s1k64", 11: "128x64w8s1k64", 12: "64x64w1s1", 13: "128x64w2s1", 14: "128x64s1", 15: "64x128s1", 16: "128x128s1",
              17: "d128x128", 18: "d64x128", 19: "d256x128", 20: "d64x128s2", 21: "d128x256", 22: "d128x128k32", 23: "d192x128", 24: "d64x128k64"}
TILE_IDS = tuple(range(1, 17))
PREC_F32, PREC_BF16X3, PREC_FP16X2 = 0, 1, 2
PREC_NAMES = {0: "f32", 1: "bf16x3", 2: "fp16x2"}
DMA_TILE_IDS = (17, 18, 19, 20, 21, 22, 23, 24)                         # through LDS-DMA (23: fp16x2 only): one channel segment, Cin % 16 / % 32 == 0, bf16x3 ONLY
BF16X3_TILE_IDS = (3, 8, 9, 10, 14, 15, 16) + tuple(t for t in DMA_TILE_IDS if t not in (23, 24))   # tiles that exist as bf16x3 variants (include/tsod.h)
FP16X2_TILE_IDS = tuple(t for t in BF16X3_TILE_IDS if t not in (18, 20)) + (23, 24)   # fp16x2: every bf16x3 tile but the 64-row LDS-DMA ones, + d192x128


class TsodError(RuntimeError):
    pass


class ConvDesc(Structure):
    """Mirror of ``tsod_conv2d_desc`` (include/tsod.h)."""
    _fields_ = [
        ("N", c_int32), ("H", c_int32), ("W", c_int32),
        ("in_pitch", c_int32), ("n_seg", c_int32),
        ("seg_off", c_int32 * TSOD_MAX_SEGMENTS), ("seg_len", c_int32 * TSOD_MAX_SEGMENTS),
        ("Cout", c_int32), ("out_pitch", c_int32), ("out_off", c_int32),
        ("KH", c_int32), ("KW", c_int32), ("stride", c_int32), ("pad_h", c_int32), ("pad_w", c_int32),
        ("OH", c_int32), ("OW", c_int32), ("act", c_int32), ("slope", c_float),
        ("res_pitch", c_int32), ("res_off", c_int32), ("tile", c_int32), ("split_k", c_int32), ("precision", c_int32),
        ("c2", c_int32), ("in2_pitch", c_int32), ("in2_off", c_int32), ("stride2", c_int32), ("H2", c_int32), ("W2", c_int32),
        ("a_scale_exp", c_int32), ("w_scale_exp", c_int32), ("range_flag", c_void_p),
        ("amax_in", c_void_p), ("amax_in2", c_void_p), ("amax_out", c_void_p),
    ]


class BottleneckDesc(Structure):
    """Mirror of ``tsod_bottleneck_desc`` (include/tsod.h)."""
    _fields_ = [
        ("N", c_int32), ("H", c_int32), ("W", c_int32), ("Cin", c_int32), ("in_pitch", c_int32), ("Cmid", c_int32),
        ("Cout", c_int32), ("out_pitch", c_int32), ("slope", c_float), ("w_exp", c_int32 * 3), ("a_scale_exp", c_int32),
        ("projection", c_int32), ("range_flag", c_void_p), ("amax_in", c_void_p), ("amax_out", c_void_p),
    ]


class StemDesc(Structure):
    """Mirror of ``tsod_stem_desc`` (include/tsod.h)."""
    _fields_ = [
        ("N", c_int32), ("H", c_int32), ("W", c_int32), ("in_layout", c_int32), ("out_pitch", c_int32), ("slope", c_float),
        ("w_exp", c_int32), ("range_flag", c_void_p), ("amax_out", c_void_p),
    ]


STEM_NCHW, STEM_NHWC4 = 0, 1

# name -> (restype, argtypes); every symbol include/tsod.h declares
_SIGNATURES = {
    "tsod_status_str": (c_char_p, [c_int]),
    "tsod_version": (c_int, []),
    "tsod_device_cu_count": (c_int, []),
    "tsod_pack_conv_weight_f32": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "tsod_conv_weight_bf16x3_bytes": (c_size_t, [c_int32, c_int32]),
    "tsod_pack_conv_weight_bf16x3": (c_int, [c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "tsod_conv_weight_fp16x2_bytes": (c_size_t, [c_int32, c_int32]),
    "tsod_pack_conv_weight_fp16x2": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "tsod_conv2d_workspace_bytes": (c_size_t, [POINTER(ConvDesc)]),
    "tsod_conv2d_resolve": (c_int, [POINTER(ConvDesc), POINTER(c_int32), POINTER(c_int32)]),
    "tsod_conv2d_f32": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                c_void_p, c_size_t, c_void_p]),
    "tsod_conv2d_dual_f32": (c_int, [POINTER(ConvDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                     c_void_p, c_size_t, c_void_p]),
    "tsod_bottleneck_wstream_bytes": (c_size_t, [c_int32, c_int32]),
    "tsod_bottleneck_proj_wstream_bytes": (c_size_t, [c_int32, c_int32]),
    "tsod_bottleneck_fp16x2": (c_int, [POINTER(BottleneckDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "tsod_stem_wfrag_bytes": (c_size_t, []),
    "tsod_stem_fp16x2": (c_int, [POINTER(StemDesc), c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "tsod_linear_f32": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_int32,
                                c_void_p, c_size_t, c_void_p]),
    "tsod_linear_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "tsod_maxpool3x3s2_f32": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p]),
    "tsod_dwconv3x3_f32": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                   c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int32, c_void_p]),
    "tsod_dwconv3x3_amax_f32": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                        c_void_p, c_int32, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "tsod_gconv3x3_amax_f32": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                       c_int32, c_int32, c_float, c_void_p, c_int32, c_void_p, c_void_p]),
    "tsod_gconv1x1_pair_amax_f32": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "tsod_nchw_to_nhwc_amax_f32": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "tsod_absmax_f32": (c_int, [c_void_p, c_int64, c_void_p, c_void_p]),
    "tsod_amax_reset": (c_int, [c_void_p, c_int32, c_void_p]),
    "tsod_host_mapped_pointer": (c_int, [c_void_p, c_void_p]),
    "tsod_word_publish_i32": (c_int, [c_void_p, c_void_p, c_void_p]),
    "tsod_gconv3x3_f32": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                  c_int32, c_int32, c_float, c_void_p, c_int32, c_void_p]),
    "tsod_gconv1x1_pair_f32": (c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p]),
    "tsod_nchw_to_nhwc_f32": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_int32, c_void_p]),
    "tsod_nhwc_to_nchw_f32": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "tsod_rpn_decode_f32": (c_int, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32,
                                    c_int32, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "tsod_proposal_decode_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_float, c_float, c_float, c_void_p,
                                         c_void_p, c_void_p]),
    "tsod_enumerate_anchors_f32": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "tsod_loc2bbox_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "tsod_bbox2loc_f32": (c_int, [c_void_p, c_void_p, c_int64, c_void_p, c_void_p]),
    "tsod_sort_topk_desc_f32": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                        c_void_p, c_void_p]),
    "tsod_sort_topk_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "tsod_sort_topk_desc_ws_f32": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                           c_void_p, c_void_p, c_size_t, c_void_p]),
    "tsod_nms_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "tsod_nms_f32": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_float, c_int32, c_void_p, c_void_p, c_void_p,
                             c_void_p, c_void_p, c_size_t, c_void_p]),
    "tsod_bbox_iou_f32": (c_int, [c_void_p, c_int32, c_void_p, c_int32, c_float, c_void_p, c_void_p]),
    "tsod_roi_pool_f32": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_float,
                                  c_int32, c_int32, c_void_p, c_void_p]),
    "tsod_roi_pool_avg_f32": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32,
                                      c_float, c_float, c_float, c_int32, c_int32, c_void_p, c_int32, c_void_p]),
    "tsod_roi_align_f32": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_float,
                                   c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "tsod_roi_align_avg_f32": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32,
                                       c_float, c_float, c_float, c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32,
                                       c_void_p]),
    "tsod_detections_f32": (c_int, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_void_p]),
    "tsod_resize_aa_taps": (c_int32, [c_int32, c_int32]),
    "tsod_resize_aa_tables_f32": (c_int, [c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "tsod_resize_bilinear_aa_u8_f32": (c_int, [c_void_p, c_int32, c_int32, c_int32, c_int64, c_void_p, c_void_p, c_void_p,
                                               c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_float, c_void_p, c_int64,
                                               c_int64, c_int64, c_int32, c_void_p]),
    "tsod_detection_keys_f32": (c_int, [c_void_p, c_int64, c_float, c_int32, c_void_p, c_void_p]),
    "tsod_gather_rows_f32": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    "tsod_detection_nms_f32": (c_int, [c_void_p, c_void_p, c_int32, c_int32, c_float, c_int32, c_void_p, c_void_p, c_void_p,
                                       c_size_t, c_void_p]),
    "tsod_anchor_targets_workspace_bytes": (c_size_t, [c_int32, c_int32]),
    "tsod_anchor_targets_f32": (c_int, [c_void_p, c_int32, c_void_p, c_int32, c_float, c_float, c_int32, c_int32, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "tsod_proposal_targets_workspace_bytes": (c_size_t, [c_int32, c_int32, c_int32]),
    "tsod_proposal_targets_f32": (c_int, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_float, c_float,
                                          c_float, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "tsod_allgather_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]),
    "tsod_comm_unique_id": (c_int, [c_void_p]),
    "tsod_comm_init_rank": (c_int, [POINTER(c_void_p), c_int32, c_void_p, c_int32]),
    "tsod_comm_destroy": (c_int, [c_void_p]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


def lib() -> ctypes.CDLL:
    """The loaded libtsod.so.  Raises (no fallback) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TsodError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
                "g.build()'` (or `make -C two_stage_object_detection_amd/csrc`). There is no CPU fallback.")
        l = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(l, name)  # AttributeError if the library lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        if l.tsod_version() != 242:
            raise TsodError(f"{LIB_PATH} is version {l.tsod_version()}, this package binds version 242 of include/tsod.h: rebuild it "
                            "(`make -C two_stage_object_detection_amd/csrc`)")
        _lib = l
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise TsodError(f"{what or 'tsod call'} failed: {lib().tsod_status_str(rc).decode()} ({rc})")


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t) -> int:
    """Raw device pointer of a tensor (or 0 for None)."""
    return 0 if t is None else t.data_ptr()


class NHWC4Images:
    """A batch of images already in the backbone's input layout: ``data`` is [N,H,W,4] f32 on the GPU (RGB + a zero
    4th channel), as written by ``dataset.transform.EvalTransform.batch``.  ``shape`` reports the NCHW shape the
    reference's code reads (``x.shape[1:]`` = (3,H,W), nets/frcnn.py:33), so the detector accepts it wherever it accepts
    an NCHW tensor and skips its own layout kernel."""

    def __init__(self, data: torch.Tensor):
        if data.dim() != 4 or data.shape[3] != 4 or data.dtype != torch.float32 or not data.is_contiguous():
            raise TsodError(f"NHWC4Images needs a contiguous f32 [N,H,W,4] tensor, got {tuple(data.shape)} {data.dtype}")
        self.data = data

    @property
    def shape(self):
        n, h, w, _ = self.data.shape
        return torch.Size((n, 3, h, w))

    device = property(lambda self: self.data.device)
    is_cuda = property(lambda self: self.data.is_cuda)
    dtype = property(lambda self: self.data.dtype)

    def dim(self):
        return 4


def require_cuda(t, what: str) -> None:
    if isinstance(t, NHWC4Images):
        t = t.data
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TsodError(f"{what}: this package is a HIP-only path and needs a CUDA/ROCm tensor "
                        "(the CPU restatement lives under oracle/ and is test infrastructure only)")
    if t.dtype != torch.float32:
        raise TsodError(f"{what}: float32 required, got {t.dtype}")


def make_conv_desc(*, N, H, W, in_pitch, segs, Cout, out_pitch, out_off=0, KH=1, KW=1, stride=1, pad_h=0, pad_w=0,
                   OH=None, OW=None, act=ACT_NONE, slope=0.0, res_pitch=0, res_off=0, tile=TILE_AUTO, split_k=0,
                   precision=0, src2=None) -> ConvDesc:
    """``src2`` = (channels, pitch, channel offset, stride, H2, W2) of the optional second source (tsod_conv2d_dual_f32)."""
    d = ConvDesc()
    d.N, d.H, d.W, d.in_pitch = N, H, W, in_pitch
    d.n_seg = len(segs)
    for i, (off, ln) in enumerate(segs):
        d.seg_off[i], d.seg_len[i] = off, ln
    d.Cout, d.out_pitch, d.out_off = Cout, out_pitch, out_off
    d.KH, d.KW, d.stride, d.pad_h, d.pad_w = KH, KW, stride, pad_h, pad_w
    d.OH = OH if OH is not None else (H + 2 * pad_h - KH) // stride + 1
    d.OW = OW if OW is not None else (W + 2 * pad_w - KW) // stride + 1
    d.act, d.slope = act, float(slope)
    d.res_pitch, d.res_off, d.tile, d.split_k = res_pitch, res_off, tile, split_k
    d.precision = int(precision)
    if src2 is not None:
        d.c2, d.in2_pitch, d.in2_off, d.stride2, d.H2, d.W2 = (int(v) for v in src2)
    return d
