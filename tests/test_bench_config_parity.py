"""GPU parity of exactly the configurations bench.py times (VERDICT r01 "what's weak" 1, 4):

* the detector forward with the AUTOTUNED tile / K-slice table pinned (the committed tables under profiles/ and a
  fresh ``Plan.autotune()``), not only the cost model's default choices that the other end-to-end tests run with;
* the serving path bench.py's default line goes through (``InFlightDetector`` with those tiles: HIP graphs on
  several streams);
* BASELINE config 4: HarDNet-68 at batch 8, 3x800x1333.

Bars as everywhere: boxes / scores <= 1e-3 absolute, arg-max classes bit-exact, RoIs position-wise equal where the
scores are well separated (ResNet-50 seed 0: zero positional mismatches).
"""
import glob
import json
import os

import pytest
import torch

import oracle
import oracle.detector

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _img(shape, seed=1234):
    return torch.rand(shape, generator=torch.Generator().manual_seed(seed))


@pytest.fixture(scope="module")
def r50():
    from two_stage_object_detection_amd.testing import synthetic_detector
    model, sd = synthetic_detector("resnet50", num_classes=80, seed=0)
    x = _img((1, 3, 800, 1333))
    with torch.inference_mode():
        ref = oracle.detector_forward(sd, x, backbone="resnet50")
    return model.to("cuda:0").eval(), sd, x, ref


def _tile_tables():
    """(file, key) of every committed table: round-1 files hold one list, later ones bench.py's
    {"serial": [...], "in_flight": [...]} pair."""
    out = []
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_autotuned_tiles_b1*.json")) +
                   glob.glob(os.path.join(ROOT, "profiles", "r*_b1_autotuned_tiles.json")))       # round 3 naming
    for f in files:
        d = json.load(open(f))
        out += [(f, None)] if isinstance(d, list) else [(f, k) for k in sorted(d) if isinstance(d[k], list)]   # (heads, structure flags: not tables)
    return out


@pytest.mark.parametrize("table", _tile_tables(), ids=lambda t: os.path.basename(t[0]) + (":" + t[1] if t[1] else ""))
def test_detector_with_the_committed_autotuned_tile_tables(dev, r50, table):
    """bench.py pins (tile, split_k) per layer from Plan.autotune(); the committed tables hold tiles 3..15 and
    K-slice counts -1, 1, 3, 4, 6, 8, 12.  Same forward, those choices imported: same RoIs position for position."""
    from two_stage_object_detection_amd.testing import compare_detector_outputs
    model, sd, x, ref = r50
    table, key = table
    blob = json.load(open(table))
    tiles = blob if key is None else blob[key]
    # (the comparison legs' tables "f32" / "bf16x3" are always recorded on the plain structure)
    structure = {k: bool(blob.get(k, False)) for k in ("fuse_bottleneck", "fuse_projection", "fuse_stem")} if key in ("serial", "in_flight") else {}
    if structure.get("fuse_bottleneck") and "fuse_projection" not in blob:        # (a table without the key: what its rows say)
        structure["fuse_projection"] = not any(r[0] == "layer1.0.conv1" for r in tiles)
    xg = x.to(dev)
    legacy = any(r[0].endswith(".downsample") for r in tiles)       # tables recorded before the shortcut fusion: 53 convs
    with torch.inference_mode():
        try:
            if legacy:
                model.extractor.fuse_shortcut = False
                model.extractor.invalidate_packed()
            model.extractor.set_structure(structure)                    # (round 4 on: a table belongs to a launch structure)
            model(xg)
            plan = model.extractor._plan_for(xg)
            # (round 5: fuse_projection adds layer1's first block - 9 convs in 3 launches instead of 6 in 2; tables of round 4 lack the key)
            assert len(plan.conv_steps) == ((53 if legacy else 49) - (6 if structure.get("fuse_bottleneck") else 0)
                                            - (3 if structure.get("fuse_projection") else 0) - (1 if structure.get("fuse_stem") else 0))
            if len(tiles) == len(plan.conv_steps):
                plan.import_tiles(tiles)
                assert plan.export_tiles() == [tuple(t) + ((0,) if len(t) == 3 else ()) for t in tiles]
            else:                                                    # a table of round 4's structure: its rows for the layers that are still launches
                plan.import_tiles_by_name(tiles)
                by = {t[0]: tuple(t) for t in tiles}
                now = plan.export_tiles()
                assert len(tiles) - len(now) == 3 and all(r == by[r[0]] for r in now)
            got = [o.cpu() for o in model(xg)]
            model.raise_if_error()
        finally:                                                     # leave the shared model as the other tests expect it, whatever happened
            model.extractor.set_structure(None)
            if legacy:
                model.extractor.fuse_shortcut = True
                model.extractor.invalidate_packed()
    rep = compare_detector_outputs(got, ref)
    print(os.path.basename(table), sorted({tuple(r[1:]) for r in tiles}), rep)
    assert rep["ok"], rep
    # every RoI has its partner at the bar and with the same class; a pair of near-tied scores may swap two positions when the
    # summation order changes with the tile table (K-slices, bf16x3 piece products, the LDS-DMA kernel's K chunking)
    assert rep["rows_positional_mismatch"] <= 4 and rep["rows_unmatched"] == 0 and rep["class_mismatch"] == 0, rep


def test_detector_after_plan_autotune_and_through_the_serving_path(dev, r50):
    """What `python bench.py` does: Plan.autotune() on the box it runs on (candidates timed as two copies in flight),
    then InFlightDetector(depth=4) replaying one HIP graph per slot on four streams.  Every slot's outputs are
    checked against the oracle, and the eager forward with the tuned table as well."""
    from two_stage_object_detection_amd.serving import InFlightDetector
    from two_stage_object_detection_amd.testing import compare_detector_outputs
    model, sd, x, ref = r50
    xg = x.to(dev)
    with torch.inference_mode():
        model(xg)
        plan = model.extractor._plan_for(xg)
        before = plan.export_tiles()
        res = plan.autotune(reps=2, concurrent=2, precisions=(0, 1, 2))  # bench.py --precision auto: f32, bf16x3 and fp16x2 compete
        heads = model.autotune_heads(xg)                                 # ... and the fused RPN conv / head GEMM are tuned too
        assert all(len(c) == 3 and c[2] in (0, 1, 2) for c in heads)          # (fp16x2 too: scaled by the feature map's range words)
        tuned = plan.export_tiles()
        assert len(res) == len(plan.conv_steps) == 49          # 53 convs, the four projection shortcuts ride in their conv3's GEMM
        from two_stage_object_detection_amd._ffi import BF16X3_TILE_IDS, FP16X2_TILE_IDS, TILE_IDS
        assert all(t in (FP16X2_TILE_IDS if p == 2 else BF16X3_TILE_IDS if p else TILE_IDS) and p in (0, 1, 2) for _, t, _, p in tuned)
        got = [o.cpu() for o in model(xg)]
        model.raise_if_error()
        rep = compare_detector_outputs(got, ref)
        print("autotuned eager", sorted({tuple(r[1:]) for r in tuned}), rep)
        assert rep["ok"] and rep["rows_positional_mismatch"] <= 4 and rep["rows_unmatched"] == 0, rep   # (near-tie swaps: see above)
        server = InFlightDetector(model, xg, depth=4, tiles=tuned)
        tickets = [server.submit(xg) for _ in range(8)]
        for t in tickets[4:]:
            outs = [o.cpu() for o in server.result(t)]
            r = compare_detector_outputs(outs[:4], ref)
            assert r["ok"] and r["rows_positional_mismatch"] <= 4 and r["rows_unmatched"] == 0, (t, r)
            det_ref = oracle.detections_from_outputs(ref[0], ref[1], ref[2])
            assert torch.equal(outs[4][..., 5], det_ref[..., 5])                     # class indices bit-exact
            assert (outs[4][..., :5] - det_ref[..., :5]).abs().max().item() <= 1e-3
        server.drain()
        plan.import_tiles(before)
        model.rpn._gemm_choice.clear()
        model.head._gemm_choice.clear()


def test_detector_after_the_in_sequence_autotune(dev, r50):
    """bench.py's SERIAL table (round 3): Plan.autotune(in_sequence=5) times the five fastest candidates of every layer again
    as launches of the whole conv sequence and pins the winner there.  A speed choice like the first look: same RoIs."""
    from two_stage_object_detection_amd.testing import compare_detector_outputs
    model, sd, x, ref = r50
    xg = x.to(dev)
    with torch.inference_mode():
        model(xg)
        plan = model.extractor._plan_for(xg)
        before = plan.export_tiles()
        res = plan.autotune(reps=2, concurrent=1, precisions=(0, 1, 2), in_sequence=5)
        assert len(res) == len(plan.conv_steps) == 49 and all(r[1] > 0 for r in res)
        tuned = plan.export_tiles()
        assert [(n, t, s_, p) for n, _, t, s_, _, p in res] == tuned        # what it reports is what it pinned
        got = [o.cpu() for o in model(xg)]
        model.raise_if_error()
        rep = compare_detector_outputs(got, ref)
        print("in-sequence autotune", sorted({tuple(r[1:]) for r in tuned}), rep)
        assert rep["ok"] and rep["rows_positional_mismatch"] <= 4 and rep["rows_unmatched"] == 0 and rep["class_mismatch"] == 0, rep
        plan.import_tiles(before)


def test_detector_with_the_fp16x2_arithmetic_among_the_candidates(dev, r50):
    """`bench.py --precision auto` since the end of round 3: the third arithmetic - two fp16 pieces of 16 x per operand, three
    piece products per f32 product on v_mfma_f32_32x32x16_f16 - competes per layer; then the same with every layer that can
    take it FORCED onto it (so the gate does not depend on what the clock picked).  Same bars as every other tile table: the
    oracle's RoIs, no unmatched row, no class mismatch.  (The synthetic trunk's activations reach abs-max ~100: inside the
    arithmetic's range of 4094.)"""
    import ctypes
    from two_stage_object_detection_amd import _ffi
    from two_stage_object_detection_amd.testing import compare_detector_outputs
    model, sd, x, ref = r50
    xg = x.to(dev)
    with torch.inference_mode():
        model(xg)
        plan = model.extractor._plan_for(xg)
        before = plan.export_tiles()
        plan.autotune(reps=2, concurrent=1, precisions=(0, 1, 2))
        tuned = plan.export_tiles()
        n_h2 = sum(1 for r in tuned if r[3] == _ffi.PREC_FP16X2)
        got = [o.cpu() for o in model(xg)]
        model.raise_if_error()
        rep = compare_detector_outputs(got, ref)
        print("fp16x2 among the candidates: picked for", n_h2, "of", len(tuned), "layers", rep)
        assert n_h2 >= 8, tuned                                           # it wins wherever the LDS-DMA tile does
        assert rep["ok"] and rep["rows_positional_mismatch"] <= 4 and rep["rows_unmatched"] == 0 and rep["class_mismatch"] == 0, rep
        forced = []
        for st, (name, tile, split, prec) in zip(plan.conv_steps, tuned):
            try:
                st.choose(22, split if tile == 22 else -1, _ffi.PREC_FP16X2)      # (the LDS-DMA tile where the layer can take it)
                t_, s_ = ctypes.c_int32(0), ctypes.c_int32(0)
                if _ffi.lib().tsod_conv2d_resolve(ctypes.byref(st.desc), ctypes.byref(t_), ctypes.byref(s_)) != 0:
                    raise _ffi.TsodError("unsupported")
                forced.append((name, 22, split if tile == 22 else -1, _ffi.PREC_FP16X2))
            except _ffi.TsodError:
                forced.append((name, tile, split, prec))
        plan.import_tiles(forced)
        n_forced = sum(1 for r in forced if r[3] == _ffi.PREC_FP16X2)
        got = [o.cpu() for o in model(xg)]
        model.raise_if_error()
        rep = compare_detector_outputs(got, ref)
        print("fp16x2 forced on", n_forced, "layers", rep)
        assert n_forced >= 30, forced
        assert rep["ok"] and rep["rows_positional_mismatch"] <= 4 and rep["rows_unmatched"] == 0 and rep["class_mismatch"] == 0, rep
        # the guard end to end.  The scale follows every tensor (range words), so an image 10^4 times brighter is served, not
        # refused; a NON-FINITE pixel is what must raise instead of returning boxes
        model(xg * 1e4)
        model.raise_if_error()
        xnan = xg.clone()
        xnan[0, 1, 400, 600] = float("nan")
        model(xnan)
        with pytest.raises(_ffi.TsodError, match="fp16x2"):
            model.raise_if_error()
        model(xg)
        model.raise_if_error()                                             # ... and the flag was cleared by the raise
        plan.import_tiles(before)


@pytest.mark.parametrize("gain", [300.0, 1e-3])
def test_fp16x2_scale_follows_the_input_without_calibration(dev, r50, gain):
    """every layer on fp16x2, NO calibration pass, an image 300 times brighter (activations far beyond the
    old static range of 4094) or 1000 times darker: the trunk must give the oracle's features to the f32 path's bar and raise
    nothing - the activation exponent of every launch comes from the range words its input's producers left (layout kernel,
    conv epilogues; the pooled map shares the stem's).  A second slot's plan behaves the same (own words)."""
    from two_stage_object_detection_amd import _ffi
    model, sd, x, ref = r50
    xb = (x * gain).to(dev)
    model.extractor.set_conv_precision("fp16x2")
    try:
        with torch.inference_mode():
            assert not model.extractor.__dict__.get("_a_exps")
            feat = model(xb, mode="extractor").cpu()
            model.raise_if_error()
            plan = model.extractor._plan_for(xb)
            assert plan.dynamic_scale and all(p == _ffi.PREC_FP16X2 for *_, p in plan.export_tiles())
            assert all(st.desc.amax_in and st.desc.amax_out for st in plan.conv_steps)
            assert sum(1 for st in plan.conv_steps if st.desc.amax_in2) == 4              # the four stacked-K shortcut GEMMs
            feat_ref = oracle.detector.extractor_forward(sd, x * gain, "resnet50")
            scale = float(feat_ref.abs().max())
            assert float((feat - feat_ref).abs().max()) <= 2e-5 * scale + 1e-6 * gain
            feat1 = model.extractor.forward_nhwc(xb, slot=1)
            model.raise_if_error()
            assert torch.equal(model.extractor.forward_nhwc(xb, slot=0), feat1)           # same kernels, same words: same bits
            # the words hold what the forward saw: the input's abs-max and the feature map's
            from two_stage_object_detection_amd import hip_ops
            in_slot = (plan.amax_ptr(plan.input_nhwc) - plan.amax.data_ptr()) // _ffi.AMAX_BYTES
            assert hip_ops.amax_value(plan.amax[in_slot * 1024:(in_slot + 1) * 1024]) == float(xb.abs().max())
            out_slot = (plan.output_amax - plan.amax.data_ptr()) // _ffi.AMAX_BYTES
            assert hip_ops.amax_value(plan.amax[out_slot * 1024:(out_slot + 1) * 1024]) == float(plan.output_nhwc.abs().max())
    finally:
        model.extractor.set_conv_precision("f32")
        model.extractor.drop_plan(slot=1)


def test_fp16x2_static_exponents_and_their_calibration(dev, r50, monkeypatch):
    """The path WITHOUT range words (Plan.dynamic_scale off: what a caller of the C ABI gets who passes no amax_in): every layer
    on fp16x2 at the default exponent (2^4: |x| < 4094), an image scaled by 300 drives the trunk out of range and the forward
    must RAISE; after Plan.calibrate_fp16x2 on that input the same forward gives the oracle's features, raises nothing, and a new
    plan of the same extractor starts from the calibrated exponents.  New weights drop the calibration."""
    from two_stage_object_detection_amd import _ffi
    from two_stage_object_detection_amd.engine import Plan
    model, sd, x, ref = r50
    xb = (x * 300.0).to(dev)
    monkeypatch.setattr(Plan, "DEFAULT_DYNAMIC_SCALE", False)
    model.extractor.set_conv_precision("fp16x2")
    try:
        with torch.inference_mode():
            model.extractor.__dict__.get("_a_exps", {}).clear()
            feat = model(xb, mode="extractor")
            with pytest.raises(_ffi.TsodError, match="fp16x2"):
                model.raise_if_error()
            plan = model.extractor._plan_for(xb)
            assert not plan.dynamic_scale and not any(st.desc.amax_in or st.desc.amax_out for st in plan.conv_steps)
            assert all(p == _ffi.PREC_FP16X2 for *_, p in plan.export_tiles())
            v0 = model.weights_version()
            seen = plan.calibrate_fp16x2(xb)
            assert model.weights_version() != v0                                       # detector-level graphs are stale now
            assert len(seen) == 49 and max(m for m, _ in seen.values()) > 4094 and min(e for _, e in seen.values()) < 4
            assert all(int(st.desc.a_scale_exp) == seen[st.name][1] for st in plan.conv_steps)
            feat = model(xb, mode="extractor").cpu()
            model.raise_if_error()
            feat_ref = oracle.detector.extractor_forward(sd, x * 300.0, "resnet50")
            scale = float(feat_ref.abs().max())
            assert float((feat - feat_ref).abs().max()) <= 2e-5 * scale + 1e-6
            model(xb, slot=1, mode="extractor")
            plan1 = model.extractor._plan_for(xb, 1)
            assert all(int(st.desc.a_scale_exp) == seen[st.name][1] for st in plan1.conv_steps)
            model.raise_if_error()
        exps = model.extractor.__dict__["_a_exps"]
        assert len(exps) == 49
        model.load_state_dict(sd)                                                      # new weights: the old ranges mean nothing
        assert len(exps) == 0 and not model.extractor.__dict__["_a_exps"]
        model.to(dev)
    finally:
        model.extractor.__dict__.get("_a_exps", {}).clear()
        model.extractor.set_conv_precision("f32")
        model.extractor.invalidate_packed()


def test_detector_after_the_in_flight_refinement(dev, r50):
    """bench.py's IN-FLIGHT table (round 3): after the first look (copies of one layer side by side) the three fastest
    candidates of every layer are tried again while all slots' streams run the whole conv sequence staggered around it
    (engine.refine_in_flight).  The table it returns is pinned in every slot's plan; served through InFlightDetector it must
    give the oracle's RoIs like any other tile table."""
    from two_stage_object_detection_amd.engine import refine_in_flight
    from two_stage_object_detection_amd.serving import InFlightDetector
    from two_stage_object_detection_amd.testing import compare_detector_outputs
    model, sd, x, ref = r50
    xg = x.to(dev)
    with torch.inference_mode():
        model(xg)
        plan = model.extractor._plan_for(xg)
        before = plan.export_tiles()
        plan.autotune(reps=2, concurrent=3, precisions=(0, 1, 2), keep_shortlist=2)
        assert plan.last_shortlist is not None and len(plan.last_shortlist) == 49 and all(1 <= len(c) <= 5 for c in plan.last_shortlist)
        first = plan.export_tiles()
        plans = [plan]
        for sl in (1, 2):
            model(xg, slot=sl)
            plans.append(model.extractor._plan_for(xg, sl))
            plans[-1].import_tiles(first)
        table = refine_in_flight(plans, plan.last_shortlist, rounds=2)
        assert [t[0] for t in table] == [t[0] for t in first]
        assert all(pl.export_tiles() == table for pl in plans)                      # pinned in every slot
        assert all(tuple(t[1:]) in [tuple(c) for c in plan.last_shortlist[i]] or t == first[i] for i, t in enumerate(table))
        server = InFlightDetector(model, xg, depth=3, tiles=table)
        for t in [server.submit(xg) for _ in range(6)][3:]:
            r = compare_detector_outputs([o.cpu() for o in server.result(t)][:4], ref)
            assert r["ok"] and r["rows_positional_mismatch"] <= 4 and r["rows_unmatched"] == 0 and r["class_mismatch"] == 0, (t, r)
        server.drain()
        for pl in plans:
            pl.import_tiles(before)
        model.extractor.drop_plan(slot=1)
        model.extractor.drop_plan(slot=2)


@pytest.mark.parametrize("backbone,shape", [("resnet50", (1, 3, 800, 1333)), ("resnet50", (2, 3, 320, 448)), ("hardnet39", (2, 3, 320, 448))])
@pytest.mark.parametrize("arith", ["bf16x3", "fp16x2"])
def test_detector_with_every_dense_conv_in_bf16x3(dev, backbone, shape, arith):
    """SURVEY 8(f) rank 4, second half: the split-operand conv paths (bf16x3: three bf16 pieces, six products; fp16x2: two fp16
    pieces, three products), gated by the SAME parity suite: the whole trunk - HarDNet's concatenated-input layers included -
    on them must give the oracle's RoIs / scores / classes to the bars of the f32 path."""
    from two_stage_object_detection_amd.testing import compare_detector_outputs, synthetic_detector
    model, sd = synthetic_detector(backbone, num_classes=20, seed=0)
    if backbone.startswith("hardnet"):
        oracle.calibrate_bn(sd, _img((2, 3, 256, 320), seed=99), oracle.hardnet_trunk, arch=int(backbone[-2:]), prefix="extractor.")
        model.load_state_dict(sd)
    model = model.to(dev).eval()
    model.extractor.set_conv_precision(arith)
    x = _img(shape)
    with torch.inference_mode():
        ref = oracle.detector_forward(sd, x, backbone=backbone)
        got = [o.cpu() for o in model(x.to(dev))]
        model.raise_if_error()
        plan = model.extractor._plan_for(x.to(dev))
        assert all(p == (1 if arith == "bf16x3" else 2) for _, _, _, p in plan.export_tiles())
        feat_ref = oracle.detector.extractor_forward(sd, x, backbone)
        feat = model(x.to(dev), mode="extractor").cpu()
    scale = float(feat_ref.abs().max())
    assert float((feat - feat_ref).abs().max()) <= (2e-5 if backbone == "resnet50" else 5e-5) * scale + 1e-6     # the f32 path's feature bar
    rep = compare_detector_outputs(got, ref)
    print(arith, backbone, shape, rep)
    assert rep["ok"] and rep["rows_unmatched"] == 0 and rep["class_mismatch"] == 0, rep
    assert rep["rows_positional_mismatch"] <= 4, rep      # another arithmetic rounds differently: a near-tie may swap two rows


def test_config4_hardnet68_batch8_full_size(dev):
    """BASELINE config 4: HarDNet-68 (models/hardnet.py, depth_wise=True), batch 8, 3x800x1333.  Images 0 and 5 of the
    batch are checked against the oracle; batched and single-image GPU forwards must agree to the parity bars (images
    are independent units; the K-slice schedule - hence the f32 summation order - depends on the launch size)."""
    from two_stage_object_detection_amd.testing import compare_detector_outputs, synthetic_detector
    model, sd = synthetic_detector("hardnet68", num_classes=80, seed=0)
    # random-init HarDNet with identity BN collapses to a spatially constant feature map (thousands of exactly tied RPN
    # scores, for which the reference's argsort has no defined order): give BN the statistics a trained net would hold
    oracle.calibrate_bn(sd, _img((2, 3, 256, 320), seed=99), oracle.hardnet_trunk, arch=68, prefix="extractor.")
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    x = _img((8, 3, 800, 1333), seed=21)
    with torch.inference_mode():
        got = [o.cpu() for o in model(x.to(dev))]
        model.raise_if_error()
        singles = {i: [o.cpu() for o in model(x[i:i + 1].to(dev))] for i in (0, 5)}
        refs = {i: oracle.detector_forward(sd, x[i:i + 1], backbone="hardnet68") for i in (0, 5)}
    assert got[2].shape == (8, 300, 4) and got[1].shape == (8, 300, 81)
    report = {}
    for i in (0, 5):
        row = [got[0][i:i + 1], got[1][i:i + 1], got[2][i:i + 1], got[3][:1]]
        r_single = compare_detector_outputs(row, singles[i])
        r_oracle = compare_detector_outputs(row, refs[i])
        report[i] = {"vs_single": {k: r_single[k] for k in ("rows_positional_mismatch", "rows_unmatched", "max_abs_roi",
                                                              "max_abs_score", "class_mismatch")},
                     "vs_oracle": {k: r_oracle[k] for k in ("rows_positional_mismatch", "rows_unmatched", "max_abs_roi",
                                                              "max_abs_score", "class_mismatch")}}
        assert r_single["ok"] and r_oracle["ok"], (i, r_single, r_oracle)
        # every RoI of the batched run exists in the oracle's list (isolated sort swaps move positions, not members)
        assert r_oracle["rows_unmatched"] == 0 and r_oracle["class_mismatch"] == 0, (i, r_oracle)
        assert r_oracle["rows_positional_mismatch"] <= 12 and r_single["rows_unmatched"] == 0, (i, r_oracle, r_single)
    print("config4", json.dumps(report))
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        json.dump(report, open(os.path.join(out_dir, "config4_parity.json"), "w"))



def test_detector_with_one_launch_bottlenecks(dev, r50):
    """layer1's identity bottlenecks as ONE launch each (tsod_bottleneck_fp16x2: conv1 -> conv2 -> conv3 + x with
    both 64-channel intermediates in LDS, tile-local fp16x2 scales).  The detector with that launch structure - under the cost
    model's f32 plan for everything else, and chosen by FasterRCNN.tune among the candidates - against the oracle: same RoIs."""
    from two_stage_object_detection_amd.serving import InFlightDetector
    from two_stage_object_detection_amd.testing import compare_detector_outputs
    model, sd, x, ref = r50
    xg = x.to(dev)
    try:
        with torch.inference_mode():
            model.extractor.set_fuse_bottleneck(True, projection=True)
            got = [o.cpu() for o in model(xg)]
            model.raise_if_error()
            plan = model.extractor._plan_for(xg)
            assert len(plan.fused_steps) == 3 and len(plan.conv_steps) == 40 and len(plan.gemm_steps) == 43      # (projection block included)
            assert [st.name for st in plan.fused_steps] == ["layer1.0.fused", "layer1.1.fused", "layer1.2.fused"]
            assert all(st.desc.amax_in and st.desc.amax_out for st in plan.fused_steps)
            rep = compare_detector_outputs(got, ref)
            print("one-launch bottlenecks, cost-model plan:", rep)
            assert rep["ok"] and rep["rows_positional_mismatch"] <= 4 and rep["rows_unmatched"] == 0 and rep["class_mismatch"] == 0, rep
            # what bench.py does: tune() decides the structure by timing one pass with and without; force it on to gate the form
            table = model.tune(xg, precisions=(0, 1, 2), in_flight=2, reps=2, fuse_bottleneck=True, fuse_stem=False)
            assert table["fuse_bottleneck"] is True and len(table["serial"]) == len(table["in_flight"]) == (40 if table["fuse_projection"] else 43)
            for depth, sched in ((1, "serial"), (2, "in_flight")):
                server = InFlightDetector(model, xg, depth=depth, tiles=table)
                outs = [o.cpu() for o in server.result(server.submit(xg))]
                server.drain()
                r = compare_detector_outputs(outs[:4], ref)
                print("one-launch bottlenecks, tuned,", sched, r)
                assert r["ok"] and r["rows_positional_mismatch"] <= 4 and r["rows_unmatched"] == 0 and r["class_mismatch"] == 0, r
            auto = model.tune(xg, precisions=(0, 1, 2), schedules=("serial",), reps=2, heads=False)
            print("tune(fuse_bottleneck='auto', fuse_stem='auto') chose", auto["fuse_bottleneck"], auto["fuse_stem"])
            assert len(auto["serial"]) == 49 - (6 if auto["fuse_bottleneck"] else 0) - (3 if auto["fuse_projection"] else 0) - (1 if auto["fuse_stem"] else 0)
            assert model.extractor.fuse_bottleneck == auto["fuse_bottleneck"] and model.extractor.fuse_projection == auto["fuse_projection"] and model.extractor.fuse_stem == auto["fuse_stem"]
    finally:
        model.extractor.set_structure(None)
        model.rpn.__dict__.get("_gemm_choice", {}).clear()
        model.head.__dict__.get("_gemm_choice", {}).clear()
        model.extractor.drop_plan(slot=1)


def test_detector_with_the_one_launch_stem(dev, r50):
    """conv1 + bn1 + PReLU + max pool as ONE launch (tsod_stem_fp16x2) that reads the images where the caller holds them.  The
    detector with that stem - under the cost model's f32 plan for everything else, from NCHW images and from the input step's
    NHWC4 images, and in the form FasterRCNN.tune serves - against the oracle: same RoIs; inputs 300x larger / 1000x smaller go
    through (the pixel scale is per tile); a NaN pixel raises."""
    from two_stage_object_detection_amd._ffi import NHWC4Images, TsodError
    from two_stage_object_detection_amd.serving import InFlightDetector
    from two_stage_object_detection_amd.testing import compare_detector_outputs
    model, sd, x, ref = r50
    xg = x.to(dev)
    try:
        with torch.inference_mode():
            model.extractor.set_fuse_stem(True)
            got = [o.cpu() for o in model(xg)]
            model.raise_if_error()
            plan = model.extractor._plan_for(xg)
            assert plan.stem_step is not None and plan.gemm_steps[0] is plan.stem_step and len(plan.conv_steps) == 48
            assert plan.stem_step.desc.amax_out and plan.stem_step.desc.in_layout == 0
            rep = compare_detector_outputs(got, ref)
            print("one-launch stem, cost-model plan, NCHW:", rep)
            assert rep["ok"] and rep["rows_positional_mismatch"] <= 4 and rep["rows_unmatched"] == 0 and rep["class_mismatch"] == 0, rep
            # the same images as the input step hands them over: straight from the caller's buffer, bit-equal features
            feat = model.extractor.forward_nhwc(xg).clone()
            x4 = torch.zeros((xg.shape[0], xg.shape[2], xg.shape[3], 4), device=dev)
            x4[..., :3] = xg.permute(0, 2, 3, 1)
            feat4 = model.extractor.forward_nhwc(NHWC4Images(x4))
            assert plan.stem_step.desc.in_layout == 1 and torch.equal(feat4, feat)
            # other ranges: no calibration, no range words for the image
            for gain in (300.0, 1e-3):
                f_f32 = None
                for on in (False, True):
                    model.extractor.set_fuse_stem(on)
                    f = model.extractor.forward_nhwc(xg * gain).clone()
                    model.raise_if_error()
                    if on:
                        err = (f - f_f32).abs().max().item() / f_f32.abs().max().item()
                        print("gain", gain, "feature map, one-launch stem vs three launches:", err)
                        assert err <= 2e-5, (gain, err)
                    f_f32 = f
            bad = xg.clone()
            bad[0, 1, 100, 200] = float("nan")
            model.extractor.forward_nhwc(bad)
            with pytest.raises(TsodError, match="non-finite"):
                model.raise_if_error()
            # what bench.py serves: tune() decides by timing; force it on to gate the form
            table = model.tune(xg, precisions=(0, 1, 2), in_flight=2, reps=2, fuse_stem=True)
            chk = table.get("in_flight_check")               # (batch 1: the in-flight table is refined, then held against the serial one)
            if table["serial"] != table["in_flight"] or chk is not None:
                assert chk is not None and chk["kept"] in ("in_flight", "serial") and set(chk["conv_us_per_forward"]) == {"in_flight", "serial"}
                assert all(v > 0 for v in chk["conv_us_per_forward"].values())
                if chk["kept"] == "serial":
                    assert table["in_flight"] == table["serial"]
            assert table["fuse_stem"] is True and len(table["serial"]) == 48 - (6 if table["fuse_bottleneck"] else 0) - (3 if table["fuse_projection"] else 0)
            for depth, sched in ((1, "serial"), (2, "in_flight")):
                server = InFlightDetector(model, xg, depth=depth, tiles=table)
                outs = [o.cpu() for o in server.result(server.submit(xg))]
                server.drain()
                r = compare_detector_outputs(outs[:4], ref)
                print("one-launch stem, tuned,", sched, r)
                assert r["ok"] and r["rows_positional_mismatch"] <= 4 and r["rows_unmatched"] == 0 and r["class_mismatch"] == 0, r
    finally:
        model.extractor.set_structure(None)
        model.rpn.__dict__.get("_gemm_choice", {}).clear()
        model.head.__dict__.get("_gemm_choice", {}).clear()
        model.extractor.drop_plan(slot=1)


def _check_images(got, sd, x, backbone, images, max_pos, roi_atol=1e-3, exact=False):
    """Rows of the batched outputs ``got`` for ``images`` against single-image oracle forwards; returns the worst figures.
    ``exact``: the reference is the oracle's float64 evaluation of the network (oracle.detector_forward(exact=True): the exact value
    of the reference's math, rounded once) instead of its float32 run.  ``roi_atol`` > 1e-3 (config 4 against the FLOAT32 oracle
    only, see there): rows pair up at that distance, scores / offsets / classes keep the 1e-3 bar, and ``rows_beyond_1e-3`` counts
    the rows that have no partner at the bar itself."""
    from two_stage_object_detection_amd.testing import compare_detector_outputs
    worst = {"rows_positional_mismatch": 0, "rows_unmatched": 0, "class_mismatch": 0, "max_abs_roi": 0.0, "max_abs_score": 0.0,
             "max_abs_cls_loc": 0.0, "rows_beyond_1e-3": 0}
    for i in images:
        with torch.inference_mode():
            ref = oracle.detector_forward(sd, x[i:i + 1], backbone=backbone, exact=exact)
        row = [got[0][i:i + 1], got[1][i:i + 1], got[2][i:i + 1], got[3][:1]]
        r = compare_detector_outputs(row, ref, atol=roi_atol)
        r["rows_beyond_1e-3"] = r["rows_unmatched"] if roi_atol == 1e-3 else compare_detector_outputs(row, ref)["rows_unmatched"]
        print("image", i, "against the float64 evaluation" if exact else "against the float32 oracle", json.dumps(r))
        assert r["roi_indices_equal"] and r["rows_unmatched"] == 0 and r["class_mismatch"] == 0 and r["max_abs_roi"] <= roi_atol, (i, r)
        assert r["max_abs_score"] <= 1e-3 and r["max_abs_cls_loc"] <= 1e-3 and r["rows_positional_mismatch"] <= max_pos, (i, r)
        for k in worst:
            worst[k] = max(worst[k], r[k])
    return worst


def test_config3_batch16_in_the_form_bench_times(dev):
    """what `bench.py --batch 16` TIMES is not the cost model's f32 plan but the table FasterRCNN.tune
    returns (per layer the fastest of f32 / bf16x3 / fp16x2, K schedules 1, -1, -2, 2, 4; the two head GEMMs tuned too; no
    calibration - range words), served as a HIP graph.  That form, on BASELINE config 3 (ResNet-50, batch 16, 3x800x1333),
    against the oracle for three images of the batch: no unmatched row, no class mismatch."""
    from two_stage_object_detection_amd.serving import InFlightDetector
    from two_stage_object_detection_amd.testing import synthetic_detector
    model, sd = synthetic_detector("resnet50", num_classes=80, seed=0)
    model = model.to(dev).eval()
    x = _img((16, 3, 800, 1333), seed=1234)
    xg = x.to(dev)
    with torch.inference_mode():
        table = model.tune(xg, precisions=(0, 1, 2), in_flight=2, reps=2)
        assert set(table) - {"in_flight_check"} == {"serial", "in_flight", "heads", "fuse_bottleneck", "fuse_projection", "fuse_stem"}
        assert len(table["serial"]) == len(table["in_flight"]) == 49 - (6 if table["fuse_bottleneck"] else 0) - (3 if table["fuse_projection"] else 0) - (1 if table["fuse_stem"] else 0)
        n_h2 = sum(1 for r in table["serial"] if r[3] == 2)
        for sched, depth in (("serial", 1), ("in_flight", 2)):
            server = InFlightDetector(model, xg, depth=depth, tiles=table)
            assert server.tiles == [tuple(r) for r in table[sched]]
            outs = None
            for t in [server.submit(xg) for _ in range(depth + 1)][-depth:]:     # (a slot's outputs live until it is reused)
                outs = [o.cpu() for o in server.result(t)]
            server.drain()
            worst = _check_images(outs[:4], sd, x, "resnet50", (0, 7, 15), max_pos=6)
            print("config 3 as benched:", sched, "fp16x2 layers", n_h2, "one-launch bottlenecks:", table["fuse_bottleneck"],
                  "one-launch stem:", table["fuse_stem"], worst)
    assert n_h2 >= 20, table["serial"]


def test_config4_hardnet68_batch8_in_the_form_bench_times(dev):
    """... and BASELINE config 4 (HarDNet-68, batch 8): `bench.py --backbone hardnet68 --batch 8` runs ~60 of the 67 dense layers
    in fp16x2 after tuning, every one taking its scale from range words that SEVERAL producers share (a HarDBlock's buffer:
    block input + every layer's depthwise output).  Two images of the batch.

    The bar, and against what.  This detector's proposals reach ~900 px; RoI coordinates there are f32 values 6.1e-5 apart, so
    1e-3 is 16.4 of them.  scripts/config4_truth.py (round 5) evaluated the reference's network in float64 and measured every
    float32 pipeline against it: the reference's own CPU f32 run sits 11.5-12 spacings from the exact RoIs, the HIP f32 plan
    11-14, bf16x3 12, fp16x2 11-12.5, the tuned mix 11-12 (feature maps: 1.35-1.76e-6 rms of the abs-max for ALL of them, the
    fp16x2 / bf16x3 / tuned plans a little below the two f32 chains) - and any two of them 13-17 spacings from EACH OTHER, whatever
    the arithmetic: the sum of two independent distances to the truth.  So "within 1e-3 of the reference's f32 run" is a coin
    flip for ANY correct f32-accurate pipeline over these 134 layers (it is what round 4's 1.04-1.08e-3 on 1-3 of 600 rows was),
    and no choice of arithmetic by the tuner can change that (FasterRCNN.tune(parity_budget=...) exists and ends at the f32
    plan for any budget below that distance).  What CAN be held, and is held here STRICTLY - 1e-3, no row beyond it - is the
    distance of the benched form to the exact result: the float64 evaluation of the reference's network
    (oracle.detector_forward(exact=True)).  Against the reference's f32 run the test keeps scores / offsets at 1e-3, classes
    exact, pairs rows at 1.25e-3 (the two pipelines' distances to the truth added up) and reports the rows beyond 1e-3.
    ResNet-50 - the headline - keeps 1e-3 against the f32 run everywhere (2.4e-4)."""
    from two_stage_object_detection_amd.testing import synthetic_detector
    model, sd = synthetic_detector("hardnet68", num_classes=80, seed=0, conditioned=True)    # (the weights bench.py times)
    model = model.to(dev).eval()
    x = _img((8, 3, 800, 1333), seed=21)
    xg = x.to(dev)
    with torch.inference_mode():
        table = model.tune(xg, precisions=(0, 1, 2), schedules=("serial",), reps=2)
        n_h2 = sum(1 for r in table["serial"] if r[3] == 2)
        run, _, outs = model.make_graphed(xg)
        run(xg)
        torch.cuda.synchronize()
        model.raise_if_error()
        got = [o.cpu() for o in outs[:4]]
        exact = _check_images(got, sd, x, "hardnet68", (0, 5), max_pos=12, roi_atol=1e-3, exact=True)
        worst = _check_images(got, sd, x, "hardnet68", (0, 5), max_pos=12, roi_atol=1.25e-3)
    print("config 4 as benched: fp16x2 layers", n_h2, "of", len(table["serial"]), "against the float64 evaluation:", exact,
          "margin to 1e-3:", 1e-3 - exact["max_abs_roi"], "| against the float32 oracle:", worst)
    assert exact["rows_beyond_1e-3"] == 0 and exact["rows_unmatched"] == 0 and exact["max_abs_roi"] <= 1e-3, exact
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):
        json.dump({"fp16x2_layers": n_h2, "against_float64_evaluation": exact, "against_float32_oracle": worst},
                  open(os.path.join(out_dir, "config4_benched_form_parity.json"), "w"))
    assert n_h2 >= 30, table["serial"]


def test_tune_with_a_parity_budget_demotes_layers_until_it_holds(dev):
    """FasterRCNN.tune(parity_budget=...): GPU against GPU (the all-f32 plan on the tuning input), layers leave their tuned
    arithmetic for the f32 MFMA kernel in the order of what they buy until the decoded RPN boxes are within the budget; the
    demotions are in the table and travel with it.  A budget of zero ends at the f32 plan itself (bit-identical outputs); a
    budget above what the tuned plan measures demotes nothing."""
    from two_stage_object_detection_amd.testing import compare_detector_outputs, synthetic_detector
    model, sd = synthetic_detector("resnet50", num_classes=20, seed=0)
    model = model.to(dev).eval()
    x = _img((1, 3, 320, 448), seed=3)
    xg = x.to(dev)
    kw = dict(precisions=(0, 2), schedules=("serial",), in_sequence=0, reps=1, fuse_bottleneck=False, fuse_stem=False, splits=[1, -1])
    with torch.inference_mode():
        ref = oracle.detector_forward(sd, x, backbone="resnet50")
        loose = model.tune(xg, parity_budget=1.0, **kw)
        pb = loose["parity_budget"]
        assert pb["held"] and pb["demoted"] == [] and 0.0 < pb["before_px"] == pb["after_px"] <= 1.0, pb
        n_h2 = sum(1 for r in loose["serial"] if r[3] == 2)
        assert n_h2 >= 20
        # the same table under half the distance it measured (the budget step alone: re-tuning would draw another table)
        import copy
        tight = copy.deepcopy(loose)
        model.import_tuning(tight, xg)
        pt = tight["parity_budget"] = model._hold_parity_budget(xg, tight, ["serial"], 0.5 * pb["before_px"])
        print("parity budget", pt)
        assert pt["held"] and 1 <= len(pt["demoted"]) and pt["after_px"] <= pt["budget_px"] < pt["before_px"], pt
        assert abs(pt["before_px"] - pb["before_px"]) <= 1e-6
        for name in pt["demoted"]:
            if not name.startswith("rpn"):
                assert [r for r in tight["serial"] if r[0] == name][0][3] == 0, name
        out = [o.cpu() for o in model(xg)]
        model.raise_if_error()
        assert compare_detector_outputs(out, ref)["ok"]
        zero = model.tune(xg, parity_budget=0.0, **kw)
        pz = zero["parity_budget"]
        assert pz["held"] and pz["after_px"] == 0.0 and all(r[3] == 0 for r in zero["serial"]), pz
        out0 = [o.clone() for o in model(xg)]
        # the same model under the cost model's all-f32 plan: the very same launches
        model.extractor.set_structure(None)
        model.import_tuning({"serial": [[r[0], 0, 0, 0] for r in zero["serial"]], "heads": zero["heads"]}, xg)
        outf = model(xg)
        for a, b in zip(out0, outf):
            assert torch.equal(a, b)
        # a second detector pins the tight table through import_tuning: the demotions travel
        m2, _ = synthetic_detector("resnet50", num_classes=20, seed=0)
        m2 = m2.to(dev).eval()
        m2.import_tuning(tight, xg)
        assert m2.extractor._plan_for(xg).export_tiles() == [tuple(r) for r in tight["serial"]]
