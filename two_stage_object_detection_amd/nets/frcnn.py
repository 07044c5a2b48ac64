"""Faster R-CNN top module with the reference's surface (nets/frcnn.py) on the HIP path.

The reference file is dead code as written (its ctor raises TypeError, its forward unpacks 5 values
from an RPN that returns 4); the semantics here are those of the one working wiring,
FasterRCNNTrainer (nets/frcnn_training.py:203-217, 251-260, 289-298):

  * the RPN receives img_size = x.shape[1:] = (C,H,W)   (frcnn_training.py:252, quirk Q1)
  * the head receives img_size = x.shape[2:] = (H,W)     (nets/frcnn.py:33,39, quirk Q2)
  * roi_indices = arange(B), int32                        (frcnn_training.py:291, quirk Q6)

Added, non-breaking: ``roi_op`` = "pool" (the reference's RoIPool, default) | "align" (RoIAlign); ``backbone`` = "hardnet39" (the reference's default extractor) | "hardnet68" |
"resnet50" (resnet50(include_top=False): stride 32, 2048 channels - the composition BASELINE names).
"""
from __future__ import annotations

import itertools

import torch
import torch.nn as nn

from .. import hip_ops
from .._ffi import TsodError, require_cuda, stream_ptr
from .classify import HarNetRoIHead
from .rpn import RegionProposalNetwork
from ..models.hardnet import HarDNetFeatureExtraction, HarNetClassifier


_UID = itertools.count(1)


def _make_extractor(backbone):
    if backbone == "resnet50":
        from ..models.resnet import resnet50
        return resnet50(include_top=False), 2048, 32
    if backbone in ("hardnet39", "hardnet68", "hardnet85"):
        return HarDNetFeatureExtraction(depth_wise=True, arch=int(backbone[-2:])), 512, 16
    raise ValueError(f"unknown backbone {backbone!r}")


class FasterRCNN(nn.Module):
    def __init__(self, num_classes, mode="training", feat_stride=16, anchor_scales=[8, 16, 32], ratios=[0.5, 1, 2],
                 backbone="hardnet39", roi_op="pool"):
        super().__init__()
        self.backbone = backbone
        self.extractor, feat_ch, native_stride = _make_extractor(backbone)
        self.classifier = HarNetClassifier()
        # the reference's default 16 is the HarDNet stride; a stride-32 trunk overrides it
        self.feat_stride = native_stride if (feat_stride == 16 and native_stride != 16) else feat_stride
        self.rpn = RegionProposalNetwork(feat_ch, ratios=ratios, anchor_scales=anchor_scales,
                                         feat_stride=self.feat_stride, mode=mode)
        self.head = HarNetRoIHead(n_class=num_classes + 1, roi_size=7, spatial_scale=1, classifier=self.classifier,
                                  in_channels=feat_ch, roi_op=roi_op)
        self.__dict__["_uid"] = next(_UID)          # scratch ownership: (this detector, slot), see hip_ops._Arena

    def __setstate__(self, state):
        """copy.deepcopy / unpickling: the copy is ANOTHER detector - it gets its own scratch-ownership id (sharing one would
        make two detectors share K-slice slabs, arrival tickets and NMS scratch across streams) and no cached constants."""
        super().__setstate__(state)
        self.__dict__["_uid"] = next(_UID)
        self.__dict__.pop("_roi_idx_cache", None)

    def weights_version(self):
        """Changes whenever packed weights were invalidated (load_state_dict / .to() / invalidate_packed)."""
        return (self.extractor.weights_version, self.rpn.weights_version, self.head.weights_version)

    def invalidate_packed(self):
        """Call after editing parameters in place: drops every packed weight / plan of the detector."""
        for m in (self.extractor, self.rpn, self.head):
            m.invalidate_packed()

    def _roi_indices(self, n, device):
        """arange(B) int32 (frcnn_training.py:291), cached per (B, device): a constant, not a per-forward launch."""
        cache = self.__dict__.setdefault("_roi_idx_cache", {})
        t = cache.get((n, device))
        if t is None:
            t = cache[(n, device)] = torch.arange(n, dtype=torch.int32, device=device)
        return t

    def _feature_words(self, x, slot):
        """(range words of the backbone's output for this input geometry / slot or None, the range word of that plan's slot)."""
        plan = self.extractor._plan_for(x, slot)
        return (getattr(plan, "output_amax", 0) or None), getattr(plan, "range_flag", None)

    def forward(self, x, scale=1., mode="forward", slot=0):
        """``slot`` (added, non-breaking) selects an independent set of backbone buffers, so that forwards issued on
        different HIP streams can be in flight together."""
        if mode == "forward":
            require_cuda(x, "FasterRCNN.forward")
            with hip_ops.ARENA.scope((self._uid, slot)):          # scratch owned by (detector, slot), not by the stream
                feat = self.extractor.forward_nhwc(x, slot)
                feat_amax, flag = self._feature_words(x, slot)    # (what an fp16x2 choice of the two GEMMs below scales with)
                _, rois, _ = self.rpn.propose(feat, tuple(x.shape[1:]), scale, feat_amax=feat_amax, range_flag=flag)
                roi_indices = self._roi_indices(x.shape[0], x.device)
                roi_cls_locs, roi_scores = self.head.forward_nhwc(feat, rois, roi_indices, tuple(x.shape[2:]), feat_amax=feat_amax,
                                                                  range_flag=flag)
                # the slot's range word -> the host's copy (one thread; serving.result() then needs no device call to see whether
                # THIS forward ended with non-finite accumulators)
                self.extractor.publish_range_word(self.extractor._plan_for(x, slot))
            return roi_cls_locs, roi_scores, rois, roi_indices
        elif mode == "extractor":
            return self.extractor.forward(x)
        elif mode == "rpn":
            base_feature, img_size = x
            rpn_locs, rpn_scores, rois, anchor = self.rpn.forward(base_feature, img_size, scale)
            roi_indices = torch.arange(base_feature.shape[0], dtype=torch.int32, device=base_feature.device)
            return rpn_locs, rpn_scores, rois, roi_indices, anchor
        elif mode == "head":
            base_feature, rois, roi_indices, img_size = x
            return self.head.forward(base_feature, rois, roi_indices, img_size)
        raise ValueError(f"unknown mode {mode!r}")

    def autotune_heads(self, x, slot=0):
        """Time the candidates of the two GEMMs outside the backbone plan (fused RPN conv, fused head GEMM) for this input
        geometry and pin the fastest (``extractor._plan_for(x).autotune()`` does the same for the backbone)."""
        require_cuda(x, "FasterRCNN.autotune_heads")
        with torch.inference_mode(), hip_ops.ARENA.scope((self._uid, slot)):
            feat = self.extractor.forward_nhwc(x, slot)
            feat_amax, flag = self._feature_words(x, slot)
            rpn_choice = self.rpn.autotune(feat, feat_amax, flag)
            # the head's GEMM on the pooled features of THIS forward (real values: the fp16x2 candidates then run in range)
            _, rois, _ = self.rpn.propose(feat, tuple(x.shape[1:]), 1., feat_amax=feat_amax, range_flag=flag)
            fc7 = self.head.pooled(feat, rois, self._roi_indices(x.shape[0], x.device), tuple(x.shape[2:]))
            return rpn_choice, self.head.autotune(fc7, feat_amax, flag)

    def tune(self, example, precisions=(0, 1, 2), in_flight=1, schedules=("serial", "in_flight"), splits=None, in_sequence=None,
             in_flight_refine=None, reps=3, heads=True, fuse_bottleneck="auto", fuse_stem="auto", verbose=False, cache_dir=None,
             parity_budget=None):
        """Autotune every GEMM of the forward for ``example``'s geometry ([B,3,H,W] on the GPU) and pin the result: per conv
        layer the fastest (tile, K-slice schedule, arithmetic) among ``precisions`` (0 f32 MFMA, 1 bf16x3, 2 fp16x2 - all three
        f32-accurate; the fp16x2 scale follows every tensor per forward through its range words, so no calibration pass and no
        input range is involved), plus the fused RPN conv and the fused head GEMM.  A speed choice only.

        ``schedules``: "serial" = one forward at a time (candidates timed alone; below batch 4 the five fastest of a layer are
        timed again inside the conv sequence, ``in_sequence``); "in_flight" (only with ``in_flight`` > 1) = the objective of
        ``serving.InFlightDetector(depth=in_flight)``: candidates timed as that many copies side by side, then
        (``in_flight_refine``, default 3 below batch 4) re-tried with every slot's stream running the whole conv sequence.
        ``fuse_bottleneck`` ("auto" | True | False; needs fp16x2 among ``precisions``): ResNet's identity bottlenecks with 64 mid
        channels as ONE launch each (tsod_bottleneck_fp16x2) - "auto" times one pass over the matrix launches with and without
        and keeps the faster structure (ties within 3 % go to the one-launch form).  ``fuse_stem`` (same values, same condition): ResNet's conv1 + bn1 + PReLU + max pool as
        ONE launch that reads the images where they are, NCHW or NHWC4 (tsod_stem_fp16x2: no layout pass, no 64-channel conv output
        in memory) - "auto" times the input step + the backbone's launches with and without.
        Returns the tables as plain JSON-able data {"serial": [...], "in_flight": [...], "heads": {...}, "fuse_bottleneck": bool,
        "fuse_projection": bool (round 5: layer1's first block, whose shortcut is a 1x1 projection, as one launch too), "fuse_stem": bool}
        - feed it back through ``import_tuning`` (another process, another rank) or ``InFlightDetector(tiles=...)``.  Afterwards
        the plan of slot 0 runs the serial table when that was tuned, else the in-flight one.
        ``cache_dir``: keep the table on disk (weight_cache.save_tuning), keyed by the weights + config hash, the device name, the
        input geometry, these arguments and the sha256 of libtsod.so: a later call with the same key - another process start of the
        same server - pins the stored table in about a second instead of tuning for 25-60 s (the returned table then carries
        "cached": True).  A table is a speed choice for one device and one library build; anything else misses.
        ``parity_budget`` (pixels; default None = off): after the timing passes, hold the tuned plan to the ALL-f32 plan of the
        same model on ``example`` (GPU against GPU - no oracle on the product path): the figure is the largest distance between
        the two plans' decoded RPN boxes (every anchor that passes the min-size filter in both; the continuous quantity in front
        of the sort / NMS decisions).  While it exceeds the budget, layers leave their tuned arithmetic for the f32 MFMA kernel,
        the layer whose demotion moves the figure most first (``_hold_parity_budget``); the demotions, the figure before and
        after and the budget are recorded under "parity_budget" and travel with the table (import_tuning, the N > 1 broadcast).
        What to expect of it (scripts/config4_truth.py): two f32-accurate pipelines over a deep net differ
        by the SUM of their distances to the exact result, whatever their arithmetic - on HarDNet-68 at 3x800x1333 every
        pipeline (f32, bf16x3, fp16x2, the tuned mix AND the reference's own CPU f32 path) sits 11-14 RoI ulps from a float64
        evaluation and 13-17 from each other - so a budget below that distance demotes every layer and ends at the f32 plan."""
        from ..engine import best_table_in_flight, refine_in_flight
        require_cuda(example, "FasterRCNN.tune")
        B = example.shape[0]
        in_sequence = (5 if B < 4 else 0) if in_sequence is None else in_sequence
        in_flight_refine = (3 if B < 4 else 0) if in_flight_refine is None else in_flight_refine
        if splits is None and B >= 4:
            splits = [1, -1, -2, 2, 4]       # large M: tiles outnumber the chip's slots many times; deep K-slicing never wins there
        want = [k for k in ("serial", "in_flight") if k in schedules and (k == "serial" or in_flight > 1)]
        if not want:
            raise TsodError("FasterRCNN.tune: nothing to tune (schedules / in_flight)")
        cache_args = None
        if cache_dir is not None:
            from .. import weight_cache
            cache_args = {"precisions": [int(v) for v in precisions], "in_flight": int(in_flight), "schedules": want,
                          "parity_budget": None if parity_budget is None else float(parity_budget),
                          "splits": None if splits is None else [int(v) for v in splits], "in_sequence": int(in_sequence),
                          "in_flight_refine": int(in_flight_refine), "reps": int(reps), "heads": bool(heads),
                          "fuse_bottleneck": str(fuse_bottleneck), "fuse_stem": str(fuse_stem)}
            hit = weight_cache.load_tuning(self, cache_dir, example.shape, example.device, cache_args)
            if hit is not None:
                try:
                    self.import_tuning(hit, example, schedule=want[0] if "serial" not in want else "serial")
                    hit["cached"] = True
                    return hit
                except TsodError:
                    pass                     # (a table this build refuses: tune again and overwrite it)
        ext = self.extractor
        table = {}
        flight_plans = []                    # the slots' plans of the in-flight refinement (kept for the last check)

        def tune_schedule(plan, sched):
            if sched == "serial":
                plan.autotune(reps=reps, verbose=verbose, splits=splits, concurrent=1, precisions=precisions, in_sequence=in_sequence)
                return plan.export_tiles()
            plan.autotune(reps=reps, verbose=False, splits=splits, concurrent=max(2, in_flight), precisions=precisions,
                          keep_shortlist=in_flight_refine)
            tiles = plan.export_tiles()
            if in_flight_refine > 0 and plan.last_shortlist:
                slot_plans = [plan]
                for sl in range(1, in_flight):
                    self(example, slot=sl)
                    slot_plans.append(ext._plan_for(example, sl))
                    slot_plans[-1].import_tiles(tiles)
                tiles = refine_in_flight(slot_plans, plan.last_shortlist, verbose=verbose)
                flight_plans[:] = slot_plans
            return tiles

        with torch.inference_mode():
            can_fuse = hasattr(ext, "set_fuse_bottleneck")
            can_stem = hasattr(ext, "conv1") and hasattr(ext, "set_fuse_stem") and 2 in tuple(precisions) and bool(fuse_stem)
            stem = False
            if hasattr(ext, "set_fuse_stem"):
                ext.set_fuse_stem(False)                            # (the three-launch stem's conv1 gets its best kernel first)
            if can_fuse:
                ext.set_fuse_bottleneck(False)                      # every layer first gets its own best kernel
            self(example)                                           # builds the plan; leaves real activations (and range words) behind
            plan = ext._plan_for(example)
            table[want[0]] = tune_schedule(plan, want[0])
            fused, fused_proj = False, False
            if can_fuse and fuse_bottleneck and 2 in tuple(precisions):
                # three structures of layer1: three launches per block; the identity blocks as ONE launch each; the block with the
                # projection shortcut too.  Each is timed as whole passes of its plan on the example (every launch, real activations
                # at every layer: a pass over the matrix launches alone runs on what the pooled buffers happen to hold, and at batch 8
                # that put the three-launch structure 10 % below what a forward then took), the structures in turn, the best of three
                # turns each.  A tie within 3 % goes to fewer launches: what they save - the intermediates' bytes, the launches -
                # counts for more with several forwards in flight than the serial pass shows (one box, batch 1, serial pass 1340
                # against 1346 us: 1187 against 1127 images/s with four in flight)
                plans_s = {None: (plan, 0)}
                for proj in (False, True):
                    ext.set_fuse_bottleneck(True, projection=proj)
                    self(example)
                    plan_f = ext._plan_for(example)
                    n_f = len([st for st in plan_f.fused_steps if st is not plan_f.stem_step])
                    if n_f and n_f not in [v[1] for v in plans_s.values()]:
                        plan_f.import_tiles_by_name(table[want[0]])
                        plans_s[proj] = (plan_f, n_f)
                best = {k: float("inf") for k in plans_s}
                for _turn in range(3):
                    for k, (pl, _) in plans_s.items():
                        best[k] = min(best[k], pl.forward_time())
                t_plain = best[None]
                tried = {k: (best[k], plans_s[k][1]) for k in plans_s if k is not None}
                if verbose:
                    for k, (t_f, n_f) in tried.items():
                        print(f"  one-launch bottlenecks ({n_f}): {t_f * 1e3:.1f} us per pass against {t_plain * 1e3:.1f} us for three launches each")
                del plans_s
                cands = [] if fuse_bottleneck is True else [(t_plain, 0, False, False)]
                cands += [(t_f, n_f, True, proj) for proj, (t_f, n_f) in tried.items()]
                if cands:                                           # within 3 % of the fastest structure: the one with the most one-launch blocks
                    t_min = min(c[0] for c in cands)
                    _, _, fused, fused_proj = max((c for c in cands if c[0] <= 1.03 * t_min), key=lambda c: (c[1], -c[0]))
                ext.set_fuse_bottleneck(fused, projection=fused_proj)
                self(example)
                plan = ext._plan_for(example)
                plan.import_tiles_by_name(table[want[0]])
                table[want[0]] = plan.export_tiles()
            if can_stem:
                # the stem's structure: the tuned three launches (layout pass, conv1, max pool) against the one launch
                stem = self._stem_pays(example, plan, verbose) or fuse_stem is True
                ext.set_fuse_stem(stem)                             # (the timing above dropped the owner's plans: rebuild either way)
                self(example)
                plan = ext._plan_for(example)
                plan.import_tiles_by_name(table[want[0]])
                table[want[0]] = plan.export_tiles()
            for sched in want[1:]:
                table[sched] = tune_schedule(plan, sched)
            if flight_plans and "serial" in table and "in_flight" in table and table["serial"] != table["in_flight"]:
                # the last word on the in-flight table: it and the serial table, whole, under the in-flight measure, in turn - the
                # refinement moves one layer at a time on a 0.3 % margin and its sum of small wins need not beat the table it
                # started from, let alone the one tuned for latency (boxes of round 5: the same code served 1168 and 1206 images/s
                # with tables that differed in eight near-equivalent rows)
                kept, us = best_table_in_flight(flight_plans, {"in_flight": table["in_flight"], "serial": table["serial"]})
                table["in_flight_check"] = {"conv_us_per_forward": {k: round(v, 1) for k, v in us.items()}, "kept": kept}
                if verbose:
                    print(f"  in flight, whole tables: {us['in_flight']:.1f} us per forward with the in-flight table, {us['serial']:.1f} with the serial one -> {kept}")
                if kept == "serial":
                    table["in_flight"] = [list(r) for r in table["serial"]]
            del flight_plans[:]
            plan.import_tiles(table.get("serial") or table["in_flight"])
            table["fuse_bottleneck"] = bool(fused)
            table["fuse_projection"] = bool(fused_proj)
            table["fuse_stem"] = bool(stem)
            if heads:
                self.autotune_heads(example)
                table["heads"] = self.head_choices()
        if parity_budget is not None:
            with torch.inference_mode():
                table["parity_budget"] = self._hold_parity_budget(example, table, want, float(parity_budget), verbose)
        if cache_args is not None:
            weight_cache.save_tuning(self, cache_dir, example.shape, example.device, cache_args, table)
        return table

    def _rpn_boxes(self, x, slot=0):
        """(decoded + clamped RPN boxes [B,A,4], keys [B,A] (-inf: filtered)) of one forward of the backbone + fused RPN conv: the
        continuous quantity the proposal layer's discrete decisions are taken on."""
        with hip_ops.ARENA.scope((self._uid, slot)):
            feat = self.extractor.forward_nhwc(x, slot)
            feat_amax, flag = self._feature_words(x, slot)
            fused, _, _ = self.rpn.propose(feat, tuple(x.shape[1:]), 1., feat_amax=feat_amax, range_flag=flag)
            n, h, w, _ = feat.shape
            _, base, n_loc, n_sc = self.rpn._pack(feat.device)
            boxes, _, keys, _ = hip_ops.rpn_decode(fused[:, :n_loc], fused[:, n_loc:n_loc + n_sc], base, n, h, w, self.feat_stride,
                                                   x.shape[2], x.shape[3], self.rpn.proposal_layer.min_size * 1.)
        return boxes.clone(), keys.clone()

    def _hold_parity_budget(self, example, table, schedules, budget, verbose=False):
        """``tune(parity_budget=...)``: demote layers of the tuned table(s) to the f32 MFMA kernel until the tuned plan's decoded
        RPN boxes are within ``budget`` pixels of the all-f32 plan's on ``example``.  One sweep prices every non-f32 layer (the
        figure with that layer alone demoted), then layers are demoted in the order of what they buy, re-measuring after each,
        until the budget holds.  The fused RPN conv is the last "layer" of the list.  One-launch stem / bottlenecks exist in
        fp16x2 only and stay (both plans run them)."""
        ext = self.extractor
        plan = ext._plan_for(example)
        key = (example.shape[0],) + tuple(ext.forward_nhwc(example).shape[1:3])
        rpn_choice = self.rpn.__dict__.setdefault("_gemm_choice", {})
        tuned_rpn = rpn_choice.get(key, (0, 0, 0))
        sched = "serial" if "serial" in schedules else schedules[0]
        tuned = [tuple(r) for r in table[sched]]
        f32_rows = [(r[0], 0, 0, 0) for r in tuned]

        def figure():
            b, k = self._rpn_boxes(example)
            ok = torch.isfinite(k) & torch.isfinite(ref_k)
            d = (b - ref_b).abs().amax(-1)
            return float(torch.where(ok, d, torch.zeros_like(d)).max())
        plan.import_tiles(f32_rows)
        rpn_choice[key] = (0, 0, 0)
        ref_b, ref_k = self._rpn_boxes(example)
        rows, rpn_now = list(tuned), tuned_rpn

        def apply():
            plan.import_tiles(rows)
            rpn_choice[key] = rpn_now
        apply()
        d0 = d = figure()
        demoted = []
        if d > budget:
            # (a layer tuned to the f32 kernel under another tile or K split sums in another order than the reference plan: it is a
            # candidate too, so that a budget of zero ends at the reference plan itself)
            cands = [i for i, r in enumerate(rows) if tuple(int(v) for v in r[1:4]) != (0, 0, 0)] + ([-1] if tuple(tuned_rpn) != (0, 0, 0) else [])
            price = []
            for i in cands:                                         # the figure with layer i alone on the f32 kernel
                keep = rpn_now if i < 0 else rows[i]
                if i < 0:
                    rpn_now = (0, 0, 0)
                else:
                    rows[i] = f32_rows[i]
                apply()
                price.append((figure(), i))
                if i < 0:
                    rpn_now = keep
                else:
                    rows[i] = keep
            price.sort()
            for _, i in price:
                if d <= budget:
                    break
                if i < 0:
                    rpn_now = (0, 0, 0)
                else:
                    rows[i] = f32_rows[i]
                apply()
                d = figure()
                demoted.append("rpn (fused loc + score conv)" if i < 0 else rows[i][0])
                if verbose:
                    print(f"  parity budget {budget:g}: {demoted[-1]} -> f32, figure {d:.3g}")
        apply()
        gone = set(demoted)
        for s_ in schedules:                                        # the same layers leave their arithmetic in every schedule's table
            table[s_] = [[r[0], 0, 0, 0] if r[0] in gone else list(r) for r in table[s_]]
        if int(rpn_now[2]) == 0 and "heads" in table:
            table["heads"] = self.head_choices()
        plan.import_tiles(table[sched])
        self.extractor.raise_if_error()
        return {"budget_px": budget, "figure": "max |decoded RPN box - the all-f32 plan's| over the anchors both plans keep, on the tuning input",
                "before_px": d0, "after_px": d, "demoted": demoted, "layers": len(rows), "held": bool(d <= budget)}

    def _stem_pays(self, example, plan, verbose=False, reps=20) -> bool:
        """HIP-event time of the stem alone, back to back: the three launches of ``plan`` (layout pass + its tuned conv1 + max
        pool) against the one launch of a plan built with ``fuse_stem`` (the rest of the backbone is the same either way)."""
        from ..engine import stage_input
        ext = self.extractor

        def head_time(pl):
            n = 1 if pl.stem_step is not None else 2                # leading launches of the plan that belong to the stem
            s = stream_ptr()
            for _ in range(2):
                stage_input(pl, example)
                for fn, args in pl.steps[:n]:
                    fn(*args, s)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                stage_input(pl, example)
                for fn, args in pl.steps[:n]:
                    fn(*args, s)
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / reps
        t3 = head_time(plan)
        ext.set_fuse_stem(True)
        ext.forward_nhwc(example)
        pf = ext._plan_for(example)
        t1 = head_time(pf) if pf.stem_step is not None else float("inf")   # (inf: a backbone whose stem the kernel does not cover)
        ext.set_fuse_stem(False)                                    # (the caller switches; every switch drops the owner's plans, packed weights stay)
        ext.forward_nhwc(example)
        if verbose:
            print(f"  one-launch stem: {t1 * 1e3:.1f} us against {t3 * 1e3:.1f} us for layout pass + conv1 + max pool -> "
                  f"{'one launch' if t1 < t3 else 'three launches'}")
        return t1 < t3

    def import_tuning(self, table, example, schedule="serial", slot=0):
        """Pin a table made by ``tune`` (same model, same input geometry; e.g. rank 0's on every rank) in the plan of ``slot``."""
        require_cuda(example, "FasterRCNN.import_tuning")
        self.set_head_choices(table.get("heads"))
        if hasattr(self.extractor, "set_structure"):
            self.extractor.set_structure(table)
        with torch.inference_mode():
            self(example, slot=slot)
            self.extractor._plan_for(example, slot).import_tiles(table.get(schedule) or table["serial"])

    def head_choices(self):
        """The pinned (tile, K-slice schedule, arithmetic) choices of the two GEMMs outside the backbone plan as plain data
        (JSON-able): {"rpn": {"NxHxW": [tile, split_k, precision]}, "head": {"M": [...]}} - what ``autotune_heads`` found,
        to be persisted beside the backbone's tile table or shipped to other ranks."""
        rpn = {"x".join(str(v) for v in k): [int(c) for c in v] for k, v in self.rpn.__dict__.get("_gemm_choice", {}).items()}
        head = {str(k): [int(c) for c in v] for k, v in self.head.__dict__.get("_gemm_choice", {}).items()}
        return {"rpn": rpn, "head": head}

    def set_head_choices(self, choices):
        """Pin choices exported by ``head_choices`` (no tuning launches; every rank of a job then sums in the same order)."""
        for k, v in (choices or {}).get("rpn", {}).items():
            self.rpn.__dict__.setdefault("_gemm_choice", {})[tuple(int(t) for t in k.split("x"))] = tuple(int(c) for c in v)
        for k, v in (choices or {}).get("head", {}).items():
            self.head.__dict__.setdefault("_gemm_choice", {})[int(k)] = tuple(int(c) for c in v)

    def detections(self, x, scale=1.):
        """[B,R,6] rows (x1,y1,x2,y2, max logit, arg-max class): SURVEY D5 / frcnn_training.py:311-319."""
        cls_locs, scores, rois, _ = self.forward(x, scale)
        return hip_ops.detections(cls_locs, scores, rois)

    def load_trainer_checkpoint(self, ckpt, strict=True, packed_cache=None):
        """Load weights saved by the reference's training script (train/train.py:120-128 writes
        ``{'model_state_dict': FasterRCNNTrainer.state_dict(), ...}``; the trainer names the backbone
        ``feat_extra`` where this module - like nets/frcnn.py:15 - says ``extractor``).  ``ckpt`` is a path
        (loaded with weights_only=True, as train/train.py:60-71 does) or an already loaded dict.
        ``packed_cache``: a directory for the on-disk cache of the packed weights (``use_packed_cache``); the module must
        already live on its GPU."""
        if isinstance(ckpt, (str, bytes)) or hasattr(ckpt, "__fspath__"):
            ckpt = torch.load(ckpt, map_location="cpu", weights_only=True)
        sd = ckpt.get("model_state_dict", ckpt)
        remapped = {("extractor." + k[len("feat_extra."):] if k.startswith("feat_extra.") else k): v for k, v in sd.items()}
        res = self.load_state_dict(remapped, strict=strict)
        if packed_cache is not None:
            self.use_packed_cache(packed_cache)
        return res

    def use_packed_cache(self, directory, device=None):
        """Fold / pack the weights once and keep the result on disk, keyed by the hash of the state_dict
        (weight_cache.py): returns "hit" when ``directory`` held the packed form of exactly these weights (no folding,
        gathering or packing work is done), "miss" after packing and writing it.  Call with the module on its GPU and in
        eval mode, after the weights are final (``.to()`` and ``load_state_dict`` drop packed weights)."""
        from .. import weight_cache
        return weight_cache.ensure_packed(self, directory, device)

    def postprocess(self, det, iou_threshold=0.1, score_thresh=None, per_class=False, background_class=-1):
        """The inference-time filtering of the reference's demo script (multi_inference.py:80-87): per image,
        class-agnostic ``nms(boxes_pred, labels_score_pred, iou_threshold)`` over the decoded per-RoI boxes with the
        arg-max logit as score.  det [B,R,6] from ``detections()`` -> (det_sorted [B,R,6] in descending-score order,
        keep [B,R] int32 indices into det_sorted (-1 after n_kept), n_kept [B] int32): the survivors of image b are
        ``det_sorted[b][keep[b, :n_kept[b]].long()]``.  The defaults are the reference's behaviour; ``score_thresh``,
        ``background_class`` and ``per_class`` are the switches of a deployed detector.  Four
        libtsod launches (keys, top-k sort, row gather, bitmask NMS), no torch kernels."""
        require_cuda(det, "FasterRCNN.postprocess")
        return hip_ops.filter_detections(det, iou_threshold, score_thresh, per_class, background_class)

    def predict(self, x, scale=1., iou_threshold=0.1, score_thresh=None, per_class=False, background_class=-1):
        """forward -> detection records -> ``postprocess``: list (one entry per image) of [n_i,6] tensors
        (x1,y1,x2,y2,score,class) in descending-score order.  The only host synchronisation is reading n_kept."""
        outs = self.forward(x, scale)
        det = hip_ops.detections(outs[0], outs[1], outs[2])
        det_sorted, keep, n_kept = self.postprocess(det, iou_threshold, score_thresh, per_class, background_class)
        self.raise_if_error()
        ns = n_kept.tolist()
        return [det_sorted[b][keep[b, :ns[b]].long()] for b in range(det.shape[0])]

    def make_graphed(self, x_example, slot=0):
        """Capture forward + detection records for this input geometry into ONE HIP graph (rebuild it after a weight
        change: ``run`` raises once the weights it was captured with have been invalidated).
        Returns (run, static_input, static_outputs): copy images into ``static_input`` (or pass them
        to ``run(x)``), call ``run()``, read ``static_outputs`` = (roi_cls_locs, roi_scores, rois,
        roi_indices, detections).  torch.cuda.CUDAGraph is only the stream-capture plumbing: every
        node of the graph is a libtsod kernel."""
        require_cuda(x_example, "FasterRCNN.make_graphed")
        static_in = x_example.clone()
        side = torch.cuda.Stream(x_example.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.inference_mode():
            for _ in range(2):                                   # builds plans, sizes workspaces
                outs = self.forward(static_in, slot=slot)
                hip_ops.detections(outs[0], outs[1], outs[2])
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        # thread_local: another thread of the process (e.g. RCCL's watchdog polling its events) must not abort the capture
        with torch.inference_mode(), torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
            outs = self.forward(static_in, slot=slot)
            det = hip_ops.detections(outs[0], outs[1], outs[2])
        static_out = tuple(outs) + (det,)
        version = self.weights_version()
        plan = self.extractor._plan_for(static_in, slot)       # the graph's buffers live as long as the closure does

        def run(x=None):
            if self.weights_version() != version:
                raise TsodError("the detector's weights changed after this graph was captured (load_state_dict / .to() / "
                                "invalidate_packed): it would replay the old folded weights - call make_graphed() again")
            if x is not None:
                with torch.inference_mode():                    # static_in may have been created under inference mode
                    static_in.copy_(x, non_blocking=True)
            graph.replay()
            return static_out
        run.plan = plan
        return run, static_in, static_out

    def raise_if_error(self):
        """Surface the deferred IndexError of the proposal padding and a range violation of the fp16x2 conv
        arithmetic (one device sync each)."""
        self.rpn.raise_if_error()
        self.extractor.raise_if_error()

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()
