"""Error behaviour of the C ABI: every entry point validates its arguments on the host BEFORE it launches anything and
reports a negative tsod_status (include/tsod.h) - nothing here reaches a GPU, so it runs on the CPU box."""
from ctypes import byref, c_int32

from two_stage_object_detection_amd import _ffi
from two_stage_object_detection_amd._ffi import lib, make_conv_desc

OK, INVALID, UNSUPPORTED, ALIGNMENT, WORKSPACE = 0, -1, -2, -3, -4
P = 0x10000          # a fake, 16-byte aligned "device pointer": validation must fail before it is ever dereferenced
ODD = P + 4          # not 16-byte aligned


def _desc(**kw):
    base = dict(N=1, H=16, W=16, in_pitch=64, segs=[(0, 64)], Cout=64, out_pitch=64, KH=3, KW=3, pad_h=1, pad_w=1)
    base.update(kw)
    return make_conv_desc(**base)


def test_conv_validation():
    L = lib()
    good = _desc(tile=8, split_k=1)
    assert L.tsod_conv2d_f32(None, P, P, None, None, None, P, None, 0, None) == INVALID
    assert L.tsod_conv2d_f32(byref(good), None, P, None, None, None, P, None, 0, None) == INVALID
    assert L.tsod_conv2d_f32(byref(good), ODD, P, None, None, None, P, None, 0, None) == ALIGNMENT
    assert L.tsod_conv2d_f32(byref(_desc(N=0)), P, P, None, None, None, P, None, 0, None) == INVALID
    assert L.tsod_conv2d_f32(byref(_desc(in_pitch=62, segs=[(0, 60)])), P, P, None, None, None, P, None, 0, None) == ALIGNMENT
    assert L.tsod_conv2d_f32(byref(_desc(segs=[(0, 60), (62, 4)])), P, P, None, None, None, P, None, 0, None) == ALIGNMENT
    assert L.tsod_conv2d_f32(byref(_desc(segs=[(32, 64)])), P, P, None, None, None, P, None, 0, None) == INVALID   # past the pitch
    assert L.tsod_conv2d_f32(byref(_desc(out_pitch=32)), P, P, None, None, None, P, None, 0, None) == INVALID
    assert L.tsod_conv2d_f32(byref(_desc(tile=99)), P, P, None, None, None, P, None, 0, None) == INVALID
    assert L.tsod_conv2d_f32(byref(_desc(split_k=65)), P, P, None, None, None, P, None, 0, None) == INVALID
    assert L.tsod_conv2d_f32(byref(_desc(act=7)), P, P, None, None, None, P, None, 0, None) == INVALID
    bad = _desc()
    bad.n_seg = 17
    assert L.tsod_conv2d_f32(byref(bad), P, P, None, None, None, P, None, 0, None) == INVALID
    # K-sliced schedules need their slab workspace
    sliced = _desc(tile=8, split_k=4)
    need = L.tsod_conv2d_workspace_bytes(byref(sliced))
    assert need > 0 and L.tsod_conv2d_workspace_bytes(byref(good)) == 0
    assert L.tsod_conv2d_f32(byref(sliced), P, P, None, None, None, P, None, 0, None) == WORKSPACE
    assert L.tsod_conv2d_f32(byref(sliced), P, P, None, None, None, P, P, need - 1, None) == WORKSPACE
    tile, split = c_int32(-7), c_int32(-7)
    assert L.tsod_conv2d_resolve(byref(_desc()), byref(tile), byref(split)) == OK
    assert tile.value in _ffi.TILE_IDS and -2 <= split.value <= 64 and split.value != 0
    assert L.tsod_conv2d_resolve(byref(_desc(N=-1)), byref(tile), byref(split)) == INVALID
    assert L.tsod_conv2d_workspace_bytes(byref(_desc(N=-1))) == 0


def test_layer_kernel_validation():
    L = lib()
    assert L.tsod_linear_f32(None, 4, 8, 8, P, None, 4, P, 4, None, 0, None) == INVALID
    assert L.tsod_maxpool3x3s2_f32(P, 1, 8, 8, 6, 8, P, 8, None) == ALIGNMENT          # C % 4
    assert L.tsod_maxpool3x3s2_f32(P, 0, 8, 8, 8, 8, P, 8, None) == INVALID
    assert L.tsod_dwconv3x3_f32(P, 1, 8, 8, 8, 8, 0, P, None, None, 3, 0, P, 8, 0, None) == INVALID     # stride 3
    assert L.tsod_dwconv3x3_f32(P, 1, 8, 8, 8, 8, 4, P, None, None, 1, 0, P, 8, 0, None) == INVALID     # slice past the pitch
    assert L.tsod_dwconv3x3_f32(ODD, 1, 8, 8, 8, 8, 0, P, None, None, 1, 0, P, 8, 0, None) == ALIGNMENT
    assert L.tsod_gconv1x1_pair_f32(P, 16, 8, 12, P, None, P, 8, None) == INVALID       # in_pitch < 2G
    assert L.tsod_nchw_to_nhwc_f32(P, 1, 3, 8, 8, P, 2, 4, None) == INVALID             # out_pitch < C_pad
    assert L.tsod_nhwc_to_nchw_f32(P, 1, 8, 4, 4, 4, 0, P, None) == INVALID             # in_pitch < C
    assert L.tsod_pack_conv_weight_f32(None, 8, 8, 3, 3, 8, 3, P, None) != OK


def test_proposal_path_validation():
    L = lib()
    assert L.tsod_rpn_decode_f32(P, 35, P, 18, P, 9, 1, 4, 4, 16, 64., 64., 16., P, P, P, None, None) == INVALID   # loc_pitch < 4A
    assert L.tsod_rpn_decode_f32(P, 36, P, 18, P, 9, 1, 4, 4, 16, 64., 64., 16., ODD, P, P, None, None) == ALIGNMENT
    assert L.tsod_proposal_decode_f32(P, P, P, 0, 1., 1., 1., P, P, None) == INVALID
    assert L.tsod_enumerate_anchors_f32(P, 9, 0, 4, 16, P, None) == INVALID
    assert L.tsod_loc2bbox_f32(P, ODD, 5, P, None) == ALIGNMENT
    assert L.tsod_sort_topk_desc_f32(P, None, 1, 100, 20000, P, P, None, None, None) == UNSUPPORTED     # n_pre > 16384
    assert L.tsod_sort_topk_desc_f32(P, None, 1, 100, 16, P, P, P, None, None) == INVALID              # boxes_out without boxes
    assert L.tsod_sort_topk_desc_f32(None, None, 1, 100, 16, P, P, None, None, None) == INVALID
    ws = L.tsod_nms_workspace_bytes(2, 3000)
    assert ws == 2 * 3000 * 47 * 8 and L.tsod_nms_workspace_bytes(0, 5) == 0
    assert L.tsod_nms_f32(P, P, 2, 3000, 0.7, 300, P, P, P, P, P, ws - 1, None) == WORKSPACE
    assert L.tsod_nms_f32(P, P, 2, 3000, 0.7, 300, P, P, P, P, None, ws, None) == WORKSPACE
    assert L.tsod_nms_f32(P, P, 1, 20000, 0.7, 300, P, P, P, P, P, 1 << 40, None) == UNSUPPORTED
    assert L.tsod_nms_f32(ODD, P, 2, 3000, 0.7, 300, P, P, P, P, P, ws, None) == ALIGNMENT
    assert L.tsod_bbox_iou_f32(P, 0, P, 4, 1e-8, P, None) == INVALID
    assert L.tsod_roi_pool_f32(P, 1, 8, 8, 6, 8, P, 4, 1.0, 7, 7, P, None) == ALIGNMENT               # C % 4
    assert L.tsod_roi_pool_avg_f32(P, 1, 8, 8, 8, 8, P, P, 4, 0., 64., 1.0, 7, 7, P, 8, None) == INVALID   # img_h = 0
    assert L.tsod_detections_f32(P, 324, P, 81, P, 0, 81, P, None) == INVALID
    assert L.tsod_detections_f32(P, 320, P, 81, P, 300, 81, P, None) == INVALID       # row pitch shorter than a row


def test_filter_and_input_step_validation():
    L = lib()
    assert L.tsod_detection_keys_f32(P, 0, 0.5, -1, P, None) == INVALID
    assert L.tsod_gather_rows_f32(P, P, 0, 4, 4, 6, P, None) == INVALID
    ws = L.tsod_nms_workspace_bytes(1, 300)
    assert L.tsod_detection_nms_f32(P, P, 1, 9000, 0.1, 1, P, P, P, 1 << 40, None) == UNSUPPORTED      # R > 8192
    assert L.tsod_detection_nms_f32(P, P, 1, 300, 0.1, 1, P, P, P, ws - 8, None) == WORKSPACE
    assert L.tsod_resize_aa_taps(0, 5) == 0 and L.tsod_resize_aa_taps(1333, 600) == 7 and L.tsod_resize_aa_taps(600, 1333) == 3
    assert L.tsod_resize_aa_tables_f32(10, 5, None, None, None) == INVALID
    assert L.tsod_resize_bilinear_aa_u8_f32(P, 8, 8, 5, 40, P, P, P, P, P, P, 4, 4, 1.0, P, 16, 4, 1, 4, None) == INVALID   # C = 5
    assert L.tsod_resize_bilinear_aa_u8_f32(P, 8, 8, 3, 20, P, P, P, P, P, P, 4, 4, 1.0, P, 16, 4, 1, 4, None) == INVALID   # row < W*C
    assert L.tsod_resize_bilinear_aa_u8_f32(P, 8, 8, 3, 24, P, P, P, P, P, P, 4, 4, 1.0, P, 16, 4, 1, 2, None) == INVALID   # C_out < C


def test_status_strings():
    L = lib()
    seen = {L.tsod_status_str(c) for c in range(0, -6, -1)}
    assert len(seen) == 6 and all(isinstance(s, bytes) and s for s in seen)
    assert L.tsod_status_str(-99) == b"unknown status"


def test_target_creator_validation():
    L = lib()
    assert L.tsod_anchor_targets_f32(None, 10, P, 2, 0.7, 0.3, 128, 256, P, P, P, P, 1 << 20, None) == INVALID
    assert L.tsod_anchor_targets_f32(P, 0, P, 2, 0.7, 0.3, 128, 256, P, P, P, P, 1 << 20, None) == INVALID
    assert L.tsod_anchor_targets_f32(P, 10, None, 2, 0.7, 0.3, 128, 256, P, P, P, P, 1 << 20, None) == INVALID   # G > 0 needs boxes
    assert L.tsod_anchor_targets_f32(ODD, 10, P, 2, 0.7, 0.3, 128, 256, P, P, P, P, 1 << 20, None) == ALIGNMENT
    assert L.tsod_anchor_targets_f32(P, 10, P, 2, 0.7, 0.3, 128, 256, P, P, P, P, 8, None) == WORKSPACE
    assert L.tsod_anchor_targets_f32(P, 10, P, 2, 0.7, 0.3, 128, 256, P, P, P, None, 0, None) == WORKSPACE
    assert L.tsod_proposal_targets_f32(P, 0, P, 0, P, 128, 64, 0.5, 0.5, 0.0, P, P, P, P, P, 1 << 20, None) == INVALID
    assert L.tsod_proposal_targets_f32(P, 10, P, 2, None, 128, 64, 0.5, 0.5, 0.0, P, P, P, P, P, 1 << 20, None) == INVALID
    assert L.tsod_proposal_targets_f32(P, 10, P, 2, P, 0, 64, 0.5, 0.5, 0.0, P, P, P, P, P, 1 << 20, None) == INVALID
    assert L.tsod_proposal_targets_f32(P, 10, P, 2, P, 128, 64, 0.5, 0.5, 0.0, P, P, P, P, P, 16, None) == WORKSPACE
    assert L.tsod_anchor_targets_workspace_bytes(37800, 20) >= 37800 * 4 + 80
    assert L.tsod_proposal_targets_workspace_bytes(600, 20, 128) >= 2 * 620 * 4 + 512
    assert L.tsod_anchor_targets_workspace_bytes(0, 3) == 0 and L.tsod_proposal_targets_workspace_bytes(0, 0, 128) == 0


def test_collective_validation():
    """tsod_allgather_f32 / tsod_comm_* (SURVEY 8(b)): arguments are checked before RCCL is even looked for, so these run
    on the CPU box; with valid arguments and no loadable librccl the answer would be UNSUPPORTED, never a crash."""
    import ctypes
    L = lib()
    comm = ctypes.c_void_p()
    assert L.tsod_allgather_f32(None, P, P, 16, None) == INVALID
    assert L.tsod_allgather_f32(P, None, P, 16, None) == INVALID
    assert L.tsod_allgather_f32(P, P, P, 0, None) == INVALID
    assert L.tsod_allgather_f32(P, ODD, P, 16, None) == ALIGNMENT
    assert L.tsod_comm_unique_id(None) == INVALID
    ident = ctypes.create_string_buffer(128)
    assert L.tsod_comm_init_rank(None, 1, ident, 0) == INVALID
    assert L.tsod_comm_init_rank(byref(comm), 0, ident, 0) == INVALID
    assert L.tsod_comm_init_rank(byref(comm), 2, ident, 2) == INVALID
    assert L.tsod_comm_init_rank(byref(comm), 1, None, 0) == INVALID
    assert L.tsod_comm_destroy(None) == INVALID


def test_balanced_schedule_and_dma_tile_gates_are_host_side():
    """split_k = -2 (balanced K ranges) and the TSOD_TILE_D* tiles (conv_dma_kernel) are refused on the host, before any launch,
    whenever they cannot run the problem: f32 arithmetic, a tile that does not know the schedule, a channel count that is not
    whole K stages, concatenated inputs.  And where they can, the resolver reports them with a workspace that holds two slabs
    per workgroup behind the fixed-size ticket area."""
    L = lib()
    PREC_BF16X3 = 1
    d = _desc(tile=17, split_k=-2, precision=PREC_BF16X3)                       # d128x128, balanced: 64 channels = 4 stages of 16
    t, s = c_int32(), c_int32()
    assert L.tsod_conv2d_resolve(byref(d), byref(t), byref(s)) == OK and (t.value, s.value) == (17, -2)
    ws = L.tsod_conv2d_workspace_bytes(byref(d))
    assert ws > 256 * 1024 and (ws - 256 * 1024) % (2 * 128 * 128 * 4) == 0
    assert L.tsod_conv2d_f32(byref(d), P, P, None, None, None, P, None, 0, None) == WORKSPACE
    assert L.tsod_conv2d_resolve(byref(_desc(tile=17, split_k=-2)), byref(t), byref(s)) == UNSUPPORTED            # f32
    assert L.tsod_conv2d_resolve(byref(_desc(tile=8, split_k=-2, precision=PREC_BF16X3)), byref(t), byref(s)) == UNSUPPORTED
    assert L.tsod_conv2d_resolve(byref(_desc(tile=17, split_k=1, precision=PREC_BF16X3, in_pitch=72, segs=[(0, 72)])),
                                 byref(t), byref(s)) == UNSUPPORTED                                                # 72 % 16 != 0
    assert L.tsod_conv2d_resolve(byref(_desc(tile=18, split_k=1, precision=PREC_BF16X3, in_pitch=128, segs=[(0, 32), (64, 32)])),
                                 byref(t), byref(s)) == UNSUPPORTED                                                # two segments
    assert L.tsod_conv2d_resolve(byref(_desc(split_k=-3)), byref(t), byref(s)) == INVALID
    auto = _desc(precision=PREC_BF16X3)                                          # the cost model's pick is one of the bf16x3 tiles
    assert L.tsod_conv2d_resolve(byref(auto), byref(t), byref(s)) == OK and t.value in _ffi.BF16X3_TILE_IDS
