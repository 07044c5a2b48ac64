"""Host-side logic that needs no GPU: module surface, state_dict contract, checkpoint remapping, plan heuristics."""
import os

import pytest
import torch

from two_stage_object_detection_amd.nets.frcnn import FasterRCNN
from two_stage_object_detection_amd.nets.rpn import ProposalCreator, RegionProposalNetwork
from two_stage_object_detection_amd.models.hardnet import HarDBlock, HarDNetFeatureExtraction, hard_block_links
from two_stage_object_detection_amd.models.resnet import resnet50

REF = "/root/reference"


def test_default_detector_is_the_references_hardnet39():
    torch.manual_seed(0)
    m = FasterRCNN(num_classes=80)
    keys = list(m.state_dict().keys())
    assert keys[0] == "extractor.base.0.conv.weight"
    assert m.feat_stride == 16 and m.rpn.loc.in_channels == 512 and m.head.cls_loc.in_features == 512
    assert m.head.score.out_features == 81 and m.head.cls_loc.out_features == 324
    assert {k for k in keys if not k.startswith("extractor.")} == {
        "rpn.score.weight", "rpn.score.bias", "rpn.loc.weight", "rpn.loc.bias",
        "head.cls_loc.weight", "head.cls_loc.bias", "head.score.weight", "head.score.bias"}
    assert sum(p.numel() for p in m.extractor.parameters()) == 2485244


def test_resnet50_composition_d1():
    m = FasterRCNN(num_classes=80, backbone="resnet50")
    assert m.feat_stride == 32 and m.rpn.loc.in_channels == 2048 and m.head.score.in_features == 2048
    assert sum(p.numel() for p in m.extractor.parameters()) == 23508049


def test_trainer_checkpoint_remap(tmp_path):
    torch.manual_seed(3)
    src = FasterRCNN(num_classes=20)
    trainer_sd = {("feat_extra." + k[len("extractor."):] if k.startswith("extractor.") else k): v
                  for k, v in src.state_dict().items()}
    path = tmp_path / "FasterRCNNTrainer_best.pth"
    torch.save({"model_state_dict": trainer_sd, "optimizer_state_dict": {}, "scheduler_state_dict": {}}, path)
    dst = FasterRCNN(num_classes=20)
    res = dst.load_trainer_checkpoint(str(path))
    assert not res.missing_keys and not res.unexpected_keys
    assert all(torch.equal(a, b) for a, b in zip(src.state_dict().values(), dst.state_dict().values()))


def test_mode_strings_and_proposal_counts():
    assert ProposalCreator("training").counts() == (3000, 300)       # default module mode -> TEST numbers (quirk Q3)
    assert ProposalCreator("train").counts() == (12000, 600)
    with pytest.raises(TypeError):
        RegionProposalNetwork(512, 512)                               # the dead call shape of nets/frcnn.py:16 (Q7)


def test_hardblock_links_and_channels():
    assert [hard_block_links(i) for i in (1, 2, 3, 4, 8, 12, 16)] == [[0], [1, 0], [2], [3, 2, 0], [7, 6, 4, 0], [11, 10, 8],
                                                                      [15, 14, 12, 8, 0]]
    assert [HarDNetFeatureExtraction(True, a).base[i].get_out_ch() for a, i in ((39, 3), (68, 3), (68, 6))] == [72, 124, 262]
    blk = HarDBlock(64, 14, 1.7, 8, dwconv=True)
    real, offs, pitch = blk.slice_table()
    assert real[0] == 64 and all(o % 4 == 0 for o in offs) and pitch % 4 == 0 and blk.output_slices() == [1, 3, 5, 7, 8]


def test_modules_refuse_cpu_and_training_mode():
    from two_stage_object_detection_amd._ffi import TsodError
    m = resnet50(include_top=False).eval()
    with pytest.raises(TsodError, match="HIP-only"):
        m(torch.zeros(1, 3, 64, 64))


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout only exists in the build container")
def test_state_dicts_equal_the_references_under_the_same_seed():
    """Same keys, same order, same seeded values as the reference's own modules (checkpoint + RNG contract)."""
    import importlib.util

    def load(name):
        spec = importlib.util.spec_from_file_location("ref_" + name, f"{REF}/models/{name}.py")
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    ref_resnet, ref_hardnet = load("resnet"), load("hardnet")
    from two_stage_object_detection_amd.models import resnet as my_resnet, hardnet as my_hardnet
    pairs = [(lambda: my_resnet.resnet50(include_top=False), lambda: ref_resnet.resnet50(include_top=False)),
             (lambda: my_resnet.resnet34(), lambda: ref_resnet.resnet34()),
             (lambda: my_hardnet.HarDNetFeatureExtraction(True, 39), lambda: ref_hardnet.HarDNetFeatureExtraction(True, 39)),
             (lambda: my_hardnet.HarDNetFeatureExtraction(True, 68), lambda: ref_hardnet.HarDNetFeatureExtraction(True, 68))]
    for mine, ref in pairs:
        torch.manual_seed(0); a = mine().state_dict()
        torch.manual_seed(0); b = ref().state_dict()
        assert list(a.keys()) == list(b.keys())
        assert all(torch.equal(a[k], b[k]) for k in a)


def test_install_dropin_resolves_the_references_import_paths():
    """code written against the reference (``from nets.frcnn import FasterRCNN`` ...) imports unchanged after
    ``install_dropin()``.  Run in a child interpreter so that the aliases do not leak into this test session."""
    import subprocess
    import sys
    code = """
import two_stage_object_detection_amd as pkg
pkg.install_dropin()
from nets.frcnn import FasterRCNN
from nets.rpn import RegionProposalNetwork, ProposalCreator
from nets.classify import HarNetRoIHead
from models.resnet import resnet34, resnet50, resnet101, resnext50_32x4d, ResNet, Bottleneck, BasicBlock
from models.hardnet import HarDNetFeatureExtraction, HarNetClassifier, HarDBlock, ConvLayer, DWConvLayer, CombConvLayer
from utils.basic_anchors import generate_basic_anchor, enumerate_shifted_anchor
from utils.loc_bbox_iou import bbox_iou, loc2bbox
from dataset.transform import eval_transform
import nets.frcnn, two_stage_object_detection_amd.nets.frcnn as mine
assert nets.frcnn is mine and FasterRCNN is mine.FasterRCNN
pkg.install_dropin()                      # idempotent
m = FasterRCNN(num_classes=3)
assert type(m.rpn) is RegionProposalNetwork and type(m.head) is HarNetRoIHead
import sys, types
sys.modules['utils'] = types.ModuleType('utils')        # a foreign top-level package of the same name
try:
    pkg.install_dropin()
except ImportError:
    pass
else:
    raise SystemExit('install_dropin() silently replaced a foreign package')
pkg.install_dropin(force=True)
print('dropin ok')
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "dropin ok" in r.stdout, r.stdout + r.stderr


def test_loading_weights_through_the_detector_invalidates_every_packed_copy(tmp_path):
    """nn.Module.load_state_dict on a parent recurses through _load_from_state_dict and never calls a child's
    load_state_dict: the backbone / RPN / head register a post-hook instead, which fires on every path."""
    torch.manual_seed(5)
    for backbone in ("resnet50", "hardnet39"):
        m = FasterRCNN(num_classes=4, backbone=backbone).eval()
        for sub in (m.extractor, m.rpn, m.head):
            sub._packed_cache[("sentinel", "cpu")] = object()
            sub._plans[("sentinel",)] = object()
        v0 = m.weights_version()
        m.load_state_dict(m.state_dict())
        assert all(not sub._packed_cache and not sub._plans for sub in (m.extractor, m.rpn, m.head))
        v1 = m.weights_version()
        assert all(b > a for a, b in zip(v0, v1))
        trainer_sd = {("feat_extra." + k[len("extractor."):] if k.startswith("extractor.") else k): v
                      for k, v in m.state_dict().items()}
        m.extractor._packed_cache[("sentinel", "cpu")] = object()
        m.load_trainer_checkpoint({"model_state_dict": trainer_sd})
        assert not m.extractor._packed_cache and all(b > a for a, b in zip(v1, m.weights_version()))
        v2 = m.weights_version()
        m.extractor.load_state_dict(m.extractor.state_dict())           # directly on the child as well
        assert m.weights_version()[0] > v2[0] and m.weights_version()[1:] == v2[1:]
        m.invalidate_packed()
        assert all(b > a for a, b in zip(v2, m.weights_version()))
    import copy
    m2 = copy.deepcopy(m)                                                # plans / packed weights are not copied
    assert not m2.extractor._plans and m2.state_dict().keys() == m.state_dict().keys()
    assert m2._uid != m._uid                                             # its own scratch ownership (arena slabs / tickets)
    import pickle
    m3 = pickle.loads(pickle.dumps(m))
    assert len({m._uid, m2._uid, m3._uid}) == 3


def test_plan_cache_is_lru_bounded():
    from collections import OrderedDict
    m = resnet50(include_top=False).eval()
    m.max_plans = 3
    built = []
    for i in range(5):
        m._cached_plan(((1, 3, 32 * (i + 1), 64), "cuda:0", 0), lambda i=i: built.append(i) or f"plan{i}")
    assert list(m._plans.values()) == ["plan2", "plan3", "plan4"]
    m._cached_plan(((1, 3, 96, 64), "cuda:0", 0), lambda: built.append("again") or "never")     # hit: moves to the end
    assert built == [0, 1, 2, 3, 4] and list(m._plans.values())[-1] == "plan2"
    m.drop_plan(shape=(1, 3, 128, 64))
    assert list(m._plans.values()) == ["plan4", "plan2"]
    assert isinstance(m._plans, OrderedDict)


def test_weight_cache_hash_and_plain_data_roundtrip(tmp_path):
    """weight_cache.py on the CPU: the key is a content hash of the state_dict (any changed value or name changes it),
    and packed entries serialise to plain data that torch.load(weights_only=True) accepts."""
    from two_stage_object_detection_amd import weight_cache as wc
    from two_stage_object_detection_amd.engine import PackedConv
    torch.manual_seed(1)
    m = FasterRCNN(num_classes=3).eval()
    h0 = wc.state_dict_hash(m.state_dict())
    assert h0 == wc.state_dict_hash({k: v.clone() for k, v in m.state_dict().items()}) and len(h0) == 32
    with torch.no_grad():
        m.head.score.bias[0] += 1e-3
    assert wc.state_dict_hash(m.state_dict()) != h0
    pc = PackedConv.__new__(PackedConv)
    pc.w, pc.scale, pc.shift = torch.randn(8, 1, 1, 4), None, torch.randn(8)
    pc.cout, pc.cin_src, pc.kh, pc.kw_logical, pc.cin, pc.kw, pc.stride, pc.pad, pc.act, pc.slope = 8, 3, 1, 1, 4, 1, 1, 0, 1, 0.25
    entry = (pc, torch.arange(4.0), 36, 18)
    path = tmp_path / "x.pt"
    torch.save({"e": wc._to_state(entry)}, path)
    back = wc._from_state(torch.load(path, weights_only=True)["e"], "cpu")
    assert isinstance(back, tuple) and isinstance(back[0], PackedConv) and back[2:] == (36, 18)
    assert torch.equal(back[0].w, pc.w) and back[0].scale is None and back[0].slope == 0.25 and back[0].out_hw(5, 7) == (5, 7)
    # the file name hashes the state_dict AND what the packed entries depend on besides it (anchors, stride, RoI op, BN eps)
    p0 = wc.cache_path(str(tmp_path), m)
    assert os.path.basename(p0).startswith("hardnet39-") and p0.endswith(".tsodpack") and p0 == wc.cache_path(str(tmp_path), m)
    torch.manual_seed(1)
    m_anchor = FasterRCNN(num_classes=3, anchor_scales=[4, 8, 16]).eval()
    m_anchor.load_state_dict(m.state_dict())
    assert wc.state_dict_hash(m_anchor.state_dict()) == wc.state_dict_hash(m.state_dict())
    assert wc.cache_path(str(tmp_path), m_anchor) != p0                # same weights, other base anchors: another file
    eps0 = m.extractor.base[0].norm.eps
    m.extractor.base[0].norm.eps = eps0 * 2
    assert wc.cache_path(str(tmp_path), m) != p0                       # BN eps is folded into the packed weights
    m.extractor.base[0].norm.eps = eps0
    assert wc.load_packed(m, str(tmp_path), "cpu") is False          # nothing cached yet: the model is left alone


def test_bench_compare_records_is_position_wise():
    """bench.py --check (default for N > 1): all ranks run rank 0's tile tables, so records are compared position by position."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    g = torch.Generator().manual_seed(4)
    ref = torch.rand(3, 300, 6, generator=g) * 100
    ref[..., 5] = torch.randint(0, 81, (3, 300), generator=g).float()
    rep = bench.compare_records(ref.clone(), ref)
    assert rep["ok"] and rep["bit_exact"] and rep["records_unmatched"] == 0 and rep["images"] == 3
    near = ref.clone()
    near[1, 7, 2] += 5e-4
    rep = bench.compare_records(near, ref)
    assert rep["ok"] and not rep["bit_exact"] and 4e-4 < rep["max_abs_box_on_matched"] < 6e-4
    swapped = ref.clone()
    swapped[2, [10, 11]] = swapped[2, [11, 10]]
    rep = bench.compare_records(swapped, ref)
    assert not rep["ok"] and rep["images_off"] == [2] and rep["records_unmatched"] == 2    # a set-based check would pass this
    cls = ref.clone()
    cls[0, 0, 5] += 1
    rep = bench.compare_records(cls, ref)
    assert not rep["ok"] and rep["class_mismatch_on_matched"] == 1 and rep["images_off"] == [0]
    nan = ref.clone()
    nan[0, 3, 4] = float("nan")
    assert not bench.compare_records(nan, ref)["ok"]


def test_head_choices_roundtrip_as_json():
    """FasterRCNN.head_choices / set_head_choices: the pinned (tile, K schedule, arithmetic) of the two GEMMs outside the
    backbone plan travel as plain JSON (bench.py persists them beside the tile table and broadcasts them to every rank)."""
    import json
    torch.manual_seed(0)
    m = FasterRCNN(num_classes=3, backbone="resnet50").eval()
    assert m.head_choices() == {"rpn": {}, "head": {}}
    m.rpn.__dict__.setdefault("_gemm_choice", {})[(8, 25, 42)] = (3, 12, 1)
    m.head.__dict__.setdefault("_gemm_choice", {})[2400] = (10, 2, 1)
    blob = json.loads(json.dumps(m.head_choices()))
    assert blob == {"rpn": {"8x25x42": [3, 12, 1]}, "head": {"2400": [10, 2, 1]}}
    m2 = FasterRCNN(num_classes=3, backbone="resnet50").eval()
    m2.set_head_choices(blob)
    assert m2.rpn._gemm_choice == {(8, 25, 42): (3, 12, 1)} and m2.head._gemm_choice == {2400: (10, 2, 1)}
    m2.set_head_choices(None)                                   # nothing persisted: a no-op
    assert m2.head_choices() == blob


def test_fp16x2_host_side_choices():
    """The host side of the fp16x2 arithmetic: which tiles have it (every bf16x3 tile but the two 64-row LDS-DMA shapes, plus d192x128 and d64x128k64), the
    activation exponent picked for a measured range (16x headroom under fp16's 65504, clamped), and the weight exponent
    (max |w| just below 2^14)."""
    import torch
    from two_stage_object_detection_amd import _ffi
    from two_stage_object_detection_amd.engine import FP16X2_A_SCALE_EXP, fp16x2_activation_exp
    from two_stage_object_detection_amd.hip_ops import fp16x2_weight_scale_exp
    assert set(_ffi.FP16X2_TILE_IDS) == (set(_ffi.BF16X3_TILE_IDS) - {18, 20}) | {23, 24} and 22 in _ffi.FP16X2_TILE_IDS   # (d192x128, d64x128k64: fp16x2 only)
    assert _ffi.PREC_NAMES[_ffi.PREC_FP16X2] == "fp16x2" and FP16X2_A_SCALE_EXP == 4
    for m in (1e-3, 0.7, 100.0, 4093.0, 3e4, 1e9):
        e = fp16x2_activation_exp(m)
        assert -24 <= e <= 8
        if -24 < e < 8:
            assert (2.0 ** e) * m * 16 <= 65504.0 < (2.0 ** (e + 1)) * m * 16
    assert fp16x2_activation_exp(100.0) == 5 and fp16x2_activation_exp(0.0) == 8 and fp16x2_activation_exp(float("inf")) == 8
    assert fp16x2_activation_exp(1e12) == -24                                  # clamped: the guard reports what does not fit
    for wmax in (1e-4, 0.02, 0.9, 37.0):
        e = fp16x2_weight_scale_exp(torch.tensor([wmax, -wmax / 3, 0.0]))
        assert 8192.0 < (2.0 ** e) * wmax <= 16384.0
    assert fp16x2_weight_scale_exp(torch.zeros(4)) == 0


def test_compare_detector_outputs_is_one_to_one():
    """The matcher behind every end-to-end parity claim pairs rows ONE TO ONE: a reference RoI that the GPU side replaced by a
    duplicate of another row is a missing row (it used to pass: both copies claimed the same reference row), a pure
    permutation is not, and nothing unmatched is tolerated by ``ok``."""
    from two_stage_object_detection_amd.testing import compare_detector_outputs
    g = torch.Generator().manual_seed(3)
    R, n = 12, 5
    rois = torch.rand(1, R, 4, generator=g) * 500
    scores = torch.randn(1, R, n, generator=g)
    locs = torch.randn(1, R, 4 * n, generator=g)
    idx = torch.zeros(1, dtype=torch.int32)
    ref = (locs, scores, rois, idx)
    rep = compare_detector_outputs(ref, ref)
    assert rep["ok"] and rep["rows_unmatched"] == 0 and rep["rows_positional_mismatch"] == 0 and rep["matching"] == "one-to-one"
    # two rows trade places: every row still has its own partner
    perm = torch.arange(R)
    perm[3], perm[4] = 4, 3
    swapped = (locs[:, perm], scores[:, perm], rois[:, perm], idx)
    rep = compare_detector_outputs(swapped, ref)
    assert rep["ok"] and rep["rows_unmatched"] == 0 and rep["rows_positional_mismatch"] == 2
    # row 7 replaced by a copy of row 2: reference row 7 has no partner any more
    dup = [t.clone() for t in (locs, scores, rois)]
    for t in dup:
        t[0, 7] = t[0, 2]
    rep = compare_detector_outputs((dup[0], dup[1], dup[2], idx), ref)
    assert not rep["ok"] and rep["rows_unmatched"] == 1 and rep["rows_positional_mismatch"] == 1
    # genuine duplicates on BOTH sides (the reference's padding rule repeats rows) pair up among themselves
    both = [t.clone() for t in (locs, scores, rois)]
    for t in both:
        t[0, 9:] = t[0, 0:3]
    both_sw = [t[:, [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 9, 11]] for t in both]
    rep = compare_detector_outputs((both_sw[0], both_sw[1], both_sw[2], idx), (both[0], both[1], both[2], idx))
    assert rep["ok"] and rep["rows_unmatched"] == 0
    # a class flip on a matched row is a failure of its own
    flip = scores.clone()
    flip[0, 1] = flip[0, 1].flip(0)
    rep = compare_detector_outputs((locs, flip, rois, idx), ref)
    assert not rep["ok"] and rep["rows_unmatched"] == 0
    # a NaN box matches nothing
    bad = rois.clone()
    bad[0, 5, 0] = float("nan")
    rep = compare_detector_outputs((locs, scores, bad, idx), ref)
    assert not rep["ok"] and rep["rows_unmatched"] == 1


def test_a_score_tie_at_the_cut_of_the_proposal_list_is_a_tie_not_a_missing_row():
    """compare_detector_outputs(ref_cutoff=): the proposal layer keeps the R best NMS survivors; when the R-th and the (R+1)-th
    differ by a couple of ulps of their fg probability either pipeline may put either in the last place (config 4 on bench.py's
    input: 0.99836999 against 0.99836987).  With the reference's own record of what it cut off (testing.cutoff_candidates of the
    oracle's RPN debug) such a row is a tie - counted, reported with its score gap - and nothing else is: the same displaced row
    with a REAL score gap, or a row that is no candidate of the reference at all, stays unmatched."""
    from oracle.box import proposal_layer
    from two_stage_object_detection_amd.testing import compare_detector_outputs, cutoff_candidates
    g = torch.Generator().manual_seed(5)
    R, n = 12, 5
    rois = torch.rand(1, R, 4, generator=g) * 500
    scores = torch.randn(1, R, n, generator=g)
    locs = torch.randn(1, R, 4 * n, generator=g)
    idx = torch.zeros(1, dtype=torch.int32)
    ref = (locs, scores, rois, idx)
    nxt = torch.tensor([[10., 20., 90., 140.], [300., 310., 420., 470.]])          # the two candidates the reference cut off
    kept_sc = torch.linspace(0.9999, 0.9984, R)
    cut = [(nxt, torch.tensor([float(kept_sc[-1]) - 1.2e-7, 0.9950]), kept_sc)]
    got_rois = rois.clone()
    got_rois[0, R - 1] = nxt[0] + 3e-4                                              # the other pipeline kept candidate R + 1 instead
    got = (locs, scores, got_rois, idx)
    plain = compare_detector_outputs(got, ref)
    assert not plain["ok"] and plain["rows_unmatched"] == 1
    rep = compare_detector_outputs(got, ref, ref_cutoff=cut)
    assert rep["ok"] and rep["rows_unmatched"] == 0 and rep["rows_tied_at_cutoff"] == 1 and 1.0e-7 < rep["max_tie_score_gap"] < 1.5e-7
    # the same swap with a score gap that is no tie
    far = [(nxt, torch.tensor([float(kept_sc[-1]) - 1e-4, 0.9950]), kept_sc)]
    rep = compare_detector_outputs(got, ref, ref_cutoff=far)
    assert not rep["ok"] and rep["rows_unmatched"] == 1 and rep["rows_tied_at_cutoff"] == 0
    # a row that is none of the reference's candidates
    other = rois.clone()
    other[0, R - 1] = torch.tensor([1., 2., 3., 4.])
    rep = compare_detector_outputs((locs, scores, other, idx), ref, ref_cutoff=cut)
    assert not rep["ok"] and rep["rows_unmatched"] == 1 and rep["rows_tied_at_cutoff"] == 0
    # a displaced row in the MIDDLE of the list whose score is nowhere near the cut: not a tie either
    mid = rois.clone()
    mid[0, 2] = nxt[0]
    mid_cut = [(nxt, torch.tensor([float(kept_sc[-1]) - 1.2e-7, 0.9950]), kept_sc)]
    rep = compare_detector_outputs((locs, scores, mid, idx), ref, ref_cutoff=mid_cut)
    assert not rep["ok"] and rep["rows_unmatched"] == 1
    # cutoff_candidates reads the oracle's debug record: the survivors past n_post in score order, None for a padded list
    A = 400
    gg = torch.Generator().manual_seed(9)
    anchor = torch.rand(A, 2, generator=gg) * 600
    anchor = torch.cat([anchor, anchor + 40 + torch.rand(A, 2, generator=gg) * 100], dim=1)
    loc = torch.zeros(A, 4)
    sc = torch.rand(A, generator=gg)
    out, dbg = proposal_layer(loc, sc, anchor, (3, 800, 800), return_debug=True)
    n_post = out.shape[0]
    c = cutoff_candidates({"per_image": [dbg]}, n_post)[0]
    if dbg["n_kept"] > n_post:
        ext, ext_sc, k_sc = c
        assert torch.equal(k_sc, dbg["score_sorted"][dbg["keep_all"][:n_post]]) and float(ext_sc[0]) <= float(k_sc[-1])
        assert torch.equal(ext[0], dbg["roi_sorted"][dbg["keep_all"][n_post]])
    else:
        assert c is None
    assert cutoff_candidates({"per_image": [dbg]}, 10)[0] is not None and cutoff_candidates({"per_image": [dbg]}, 10 ** 6)[0] is None


def test_bottleneck_weight_stream_layout():
    """hip_ops.pack_bottleneck_wstream against the layout include/tsod.h documents (what bottleneck_kernel's lanes load): 8 KB
    steps in consumption order (conv1's K-steps, conv2's (tap, channel half) steps, conv3's 64-channel slices x 2 K-steps); a
    step = [channel block cb][lane = 32 hh + j][chunk c][hi | lo][8 k]: lane (j, hh) of block cb holds output channel
    32 cb + pi(j), k = 16 c + 8 hh ..; hi + lo reproduce 2^e * w to fp16x2 accuracy."""
    from two_stage_object_detection_amd import hip_ops
    g = torch.Generator().manual_seed(5)
    cin = cout = 128
    w1 = torch.randn(64, cin, generator=g)
    w2 = torch.randn(64, 3, 3, 64, generator=g)                   # packed conv layout [Cout][KH][KW][Cin]
    w3 = torch.randn(cout, 64, generator=g) * 0.01
    stream, (e1, e2, e3) = hip_ops.pack_bottleneck_wstream(w1, w2, w3)
    n1, n3 = cin // 32, (cout // 64) * 2
    assert stream.dtype == torch.uint8 and stream.numel() == (n1 + 18 + n3) * 8192
    steps = stream.view(torch.float16).view(-1, 2, 64, 2, 2, 8)   # [step][cb][lane][chunk][plane][8 k]

    def pi(j):
        return 16 * ((j >> 2) & 1) + 4 * (j >> 3) + (j & 3)

    def check(step, w_rows_k, e):
        """w_rows_k [64 channels of the step, 32 k]"""
        for cb in (0, 1):
            for lane in (0, 1, 5, 17, 31, 32, 44, 63):
                j, hh = lane & 31, lane >> 5
                for c in (0, 1):
                    hi, lo = steps[step, cb, lane, c, 0].float(), steps[step, cb, lane, c, 1].float()
                    want = w_rows_k[32 * cb + pi(j), 16 * c + 8 * hh:16 * c + 8 * hh + 8] * (2.0 ** e)
                    assert torch.equal(hi, want.half().float())
                    assert torch.equal(lo, (want - want.half().float()).half().float())
                    assert float((hi + lo - want).abs().max()) <= float(want.abs().max()) * 2.0 ** -21
    for ks in range(n1):
        check(ks, w1[:, 32 * ks:32 * ks + 32], e1)
    w2f = w2.reshape(64, 576)
    for t in (0, 1, 8, 17):                                        # step t: tap t >> 1, channel half t & 1 = k 32 t .. 32 t + 31
        check(n1 + t, w2f[:, 32 * t:32 * t + 32], e2)
        tap, half = t >> 1, t & 1
        assert torch.equal(w2f[:, 32 * t:32 * t + 32], w2[:, tap // 3, tap % 3, 32 * half:32 * half + 32])
    for q in range(cout // 64):
        for ks in range(2):
            check(n1 + 18 + 2 * q + ks, w3[64 * q:64 * q + 64, 32 * ks:32 * ks + 32], e3)
    # a lane of the accumulator owns 16 CONSECUTIVE channels: rows 4 h + (e & 3) + 8 (e >> 2) of a block are channels 16 h + e
    for h in (0, 1):
        assert [pi(4 * h + (e & 3) + 8 * (e >> 2)) for e in range(16)] == list(range(16 * h, 16 * h + 16))
    for m in (max(abs(float(w.abs().max()) * 2.0 ** e) for w, e in ((w1, e1), (w2, e2), (w3, e3))),):
        assert 8192.0 <= m < 16384.0                                # every weight matrix scaled to just below 2^14


def test_stem_weight_fragment_layout():
    """hip_ops.pack_stem_wfrag against the layout include/tsod.h documents (what stem_kernel's lanes load): [channel block cb][chunk c]
    [hi | lo][lane = 32 hh + j][8 k], lane (j, hh) of block cb holding output channel 32 cb + pi(j) and k = 16 c + 8 hh .. + 7 with
    k = 32 kh + 4 kw + ci; kw = 7 and ci = 3 are zeros; hi + lo reproduce 2^e * w to fp16x2 accuracy."""
    from two_stage_object_detection_amd import hip_ops
    g = torch.Generator().manual_seed(11)
    w = torch.randn(64, 3, 7, 7, generator=g) * 0.05
    frag, e = hip_ops.pack_stem_wfrag(w)
    assert frag.dtype == torch.uint8 and frag.numel() == 2 * 14 * 2 * 1024
    f = frag.view(torch.float16).view(2, 14, 2, 64, 8).float()

    def pi(j):
        return 16 * ((j >> 2) & 1) + 4 * (j >> 3) + (j & 3)
    wk = torch.zeros(64, 7, 8, 4)
    wk[:, :, :7, :3] = w.permute(0, 2, 3, 1)
    wk = wk.reshape(64, 224) * (2.0 ** e)
    assert float(wk.abs().max()) < 16384.0
    for cb in (0, 1):
        for lane in (0, 3, 12, 31, 32, 45, 63):
            j, hh = lane & 31, lane >> 5
            for c in range(14):
                want = wk[32 * cb + pi(j), 16 * c + 8 * hh:16 * c + 8 * hh + 8]
                hi, lo = f[cb, c, 0, lane], f[cb, c, 1, lane]
                assert torch.equal(hi, want.half().float())
                assert torch.equal(lo, (want - want.half().float()).half().float())
                # k = 16 c + 8 hh + i: filter row c >> 1, filter columns 4 (c & 1) + 2 hh and + 1, channels 0..3 of each
                kh, kw0 = c >> 1, 4 * (c & 1) + 2 * hh
                for i in range(8):
                    kw, ci = kw0 + (i >> 2), i & 3
                    ref = 0.0 if kw == 7 or ci == 3 else float(w[32 * cb + pi(j), ci, kh, kw]) * 2.0 ** e
                    assert abs(float(hi[i] + lo[i]) - ref) <= abs(ref) * 2.0 ** -21


def test_precomputed_bn_statistics_of_the_synthetic_hardnets_are_what_the_oracle_computes():
    """bench.py builds its HarDNet model with BatchNorm statistics loaded from a committed file (no oracle on the path that
    builds the timed model); the file must hold exactly what the tests' own conditioning (oracle.calibrate_bn on two seeded
    images) computes for these weights, and a file made for other weights must be refused."""
    import numpy as np
    import oracle
    import pytest
    import torch
    from two_stage_object_detection_amd import testing
    model, sd = testing.synthetic_detector("hardnet39", num_classes=20, seed=0, conditioned=True)
    _, ref = testing.synthetic_detector("hardnet39", num_classes=20, seed=0)
    x = torch.rand((2, 3, 256, 320), generator=torch.Generator().manual_seed(99))
    oracle.calibrate_bn(ref, x, oracle.hardnet_trunk, arch=39, prefix="extractor.")
    assert set(sd) == set(ref)
    for k in sd:
        assert torch.allclose(sd[k], ref[k], rtol=1e-5, atol=1e-7), k
    live = model.state_dict()
    for k in sd:
        assert torch.equal(live[k], sd[k]), k
    for bb in ("hardnet68", "hardnet85"):
        data = np.load(testing.synthetic_bn_path(bb, 0))
        assert "__weights_checksum__" in data.files and len(data.files) > 100
    with pytest.raises(RuntimeError):
        testing.synthetic_detector("hardnet39", num_classes=20, seed=1, conditioned=True)      # no file for that seed
    # resnet: nothing to condition (its BN is identity by construction of the reference's init)
    _, a = testing.synthetic_detector("resnet50", num_classes=20, seed=0, conditioned=True)
    _, b = testing.synthetic_detector("resnet50", num_classes=20, seed=0)
    assert all(torch.equal(a[k], b[k]) for k in a)


def test_tuned_table_cache_key_and_round_trip(tmp_path, monkeypatch):
    """weight_cache.save_tuning / load_tuning on the CPU: the key moves with the arguments, the geometry, the weights and the
    library's hash; a damaged file is a miss, not an error."""
    import torch
    from two_stage_object_detection_amd import testing, weight_cache
    model, _ = testing.synthetic_detector("resnet50", num_classes=20, seed=0)
    table = {"serial": [["conv1", 3, 1, 2]], "heads": {"rpn": {}, "head": {}}, "fuse_stem": False, "fuse_bottleneck": False}
    args = {"precisions": [0, 2], "in_flight": 2}
    shape = (1, 3, 224, 288)
    p = weight_cache.save_tuning(model, str(tmp_path), shape, "cpu", args, table)
    assert weight_cache.load_tuning(model, str(tmp_path), shape, "cpu", args) == table
    assert weight_cache.load_tuning(model, str(tmp_path), shape, "cpu", dict(args, in_flight=4)) is None
    assert weight_cache.load_tuning(model, str(tmp_path), (1, 3, 256, 288), "cpu", args) is None
    keep = model.rpn.loc.bias.detach().clone()
    with torch.no_grad():
        model.rpn.loc.bias.add_(1.0)
    assert weight_cache.load_tuning(model, str(tmp_path), shape, "cpu", args) is None            # other weights
    with torch.no_grad():
        model.rpn.loc.bias.copy_(keep)
    assert weight_cache.load_tuning(model, str(tmp_path), shape, "cpu", args) == table
    monkeypatch.setattr(weight_cache, "library_hash", lambda: "another-build")
    assert weight_cache.load_tuning(model, str(tmp_path), shape, "cpu", args) is None            # another libtsod.so
    monkeypatch.undo()
    open(p, "w").write("{not json")
    assert weight_cache.load_tuning(model, str(tmp_path), shape, "cpu", args) is None
    assert len(weight_cache.library_hash()) == 16


def test_bench_parity_rule_with_two_references():
    """bench.parity_of: ok = within 1e-3 of the oracle's float32 run, or within 1e-3 of its float64 evaluation AND within 1.25e-3 of
    the float32 run with equal classes; beyond both is not ok."""
    import torch
    import bench
    R, C = 4, 3
    rois = torch.tensor([[[10., 20., 110., 220.], [300., 40., 900., 700.], [5., 5., 50., 60.], [400., 300., 800., 600.]]])
    scores = torch.tensor([[[0.1, 2.0, 0.3], [1.5, 0.2, 0.1], [0.0, 0.1, 3.0], [0.3, 0.2, 0.9]]])
    locs = torch.zeros(1, R, 4 * C)
    idx = torch.zeros(1, dtype=torch.int32)
    ref = (locs, scores, rois, idx)
    exact = (locs, scores, rois + 7e-4, idx)                  # the f32 run itself is 7e-4 from the exact value
    same = {"serial": [locs, scores, rois.clone(), idx]}
    p = bench.parity_of(same, ref, exact)
    assert p["ok"] and p["within_atol"] and p["exact"]["within_atol"] and p["max_abs_roi"] == 0.0
    far_from_f32 = {"serial": [locs, scores, rois + 1.15e-3, idx]}   # 1.15e-3 from the f32 run, 4.5e-4 from the exact value
    p = bench.parity_of(far_from_f32, ref, exact)
    assert not p["within_atol"] and p["exact"]["within_atol"] and p["ok"] and p["float32_run_at_1.25e-3"]["rows_unmatched"] == 0
    assert bench.parity_of(far_from_f32, ref)["ok"] is False   # without the exact evaluation: the float32 rule alone
    wrong = {"serial": [locs, scores, rois + 3e-3, idx]}
    assert bench.parity_of(wrong, ref, exact)["ok"] is False
    flipped = scores.clone()
    flipped[0, 3] = torch.tensor([0.95, 0.2, 0.9])            # another arg-max class: never ok
    assert bench.parity_of({"serial": [locs, flipped, rois + 1.15e-3, idx]}, ref, exact)["ok"] is False
