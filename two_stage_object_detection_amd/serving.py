"""serving.py -- request-level pipelining of the detector forward: several steps in flight on one GPU.

A batch-1 forward of a two-stage detector is a chain of ~60 small launches; many of them (2-5 GFLOP layers, single
workgroup top-k / NMS scans) cannot fill 256 CUs on their own.  ``InFlightDetector`` keeps ``depth`` independent copies
of the step (own HIP graph, own backbone buffers and scratch: ``FasterRCNN.make_graphed(x, slot)``) and issues
consecutive requests round-robin on ``depth`` HIP streams, so the tail of one forward overlaps the next one's kernels.
Every request still executes the whole path; nothing is shared between slots but the (read-only) packed weights
(one copy per layer and device, engine.PlanOwner); scratch is owned per (detector, slot), not per stream handle.

This is what ``bench.py`` measures by default (``--in-flight 4``): ~585 images/s against ~420 strictly serial at batch 1
on MI355X.  Depth 2 already gives ~540; depths 3-8 are equivalent.
"""
from __future__ import annotations

from typing import Callable

import torch

from ._ffi import TsodError, require_cuda


class InFlightDetector:
    """``det = InFlightDetector(model, example, depth=4)``; ``t = det.submit(images)``; ``outs = det.result(t)``.

    ``example`` fixes the input geometry ([B,3,H,W] f32 on the GPU).  ``submit`` copies the images into the slot's
    resident input (``None`` = re-run on what is already there), replays the slot's graph on the slot's stream and
    returns a ticket; ``result(ticket)`` waits for that step only and returns the slot's output tensors
    ``(roi_cls_locs, roi_scores, rois, roi_indices, detections)`` - views that stay valid until the slot is used again,
    i.e. for the next ``depth - 1`` submissions.  ``after(outputs)``, if given, runs inside the slot's stream right
    behind the graph (e.g. the all-gather of the records in a data-parallel job)."""

    def __init__(self, model, example: torch.Tensor, depth: int = 4, autotune: bool = False, tiles=None):
        """``tiles``: a table from ``FasterRCNN.tune`` (the dict, or one of its per-schedule lists); ``autotune`` = tune here, with
        this server's overlap as the objective."""
        require_cuda(example, "InFlightDetector")
        if depth < 1:
            raise TsodError("InFlightDetector: depth must be >= 1")
        self.model, self.depth, self.device = model, depth, example.device
        with torch.inference_mode():
            if tiles is None and autotune:
                tiles = model.tune(example, in_flight=depth, schedules=("in_flight",) if depth > 1 else ("serial",))
            if isinstance(tiles, dict):
                model.set_head_choices(tiles.get("heads"))
                if hasattr(model.extractor, "set_structure"):
                    model.extractor.set_structure(tiles)                                             # the table's launch structure
                tiles = tiles.get("in_flight" if depth > 1 else "serial") or tiles.get("serial")
            model(example)                                               # builds slot 0's plan
            plan0 = model.extractor._plan_for(example, 0)
            if tiles is not None:
                plan0.import_tiles(tiles)
            self.tiles = plan0.export_tiles()
            for s in range(1, depth):                                    # the same tile choices in every slot's plan
                model(example, slot=s)
                model.extractor._plan_for(example, s).import_tiles(self.tiles)
            made = [model.make_graphed(example, slot=s) for s in range(depth)]
        self._run = [m[0] for m in made]
        self._inputs = [m[1] for m in made]
        self._outputs = [m[2] for m in made]
        self._streams = [torch.cuda.Stream(self.device) for _ in range(depth)] if depth > 1 else [None]
        self._done = [torch.cuda.Event() for _ in range(depth)]
        self._ticket_of = [None] * depth
        self._next = 0

    def submit(self, images: torch.Tensor | None = None, after: Callable | None = None) -> int:
        ticket = self._next
        slot = ticket % self.depth
        self._next += 1
        stream = self._streams[slot]
        if stream is None:
            outs = self._run[slot](images)
            if after is not None:
                after(outs)
            self._done[slot].record(torch.cuda.current_stream(self.device))
        else:
            if images is not None:                                       # the caller's writes to ``images`` are ordered first
                stream.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(stream):
                outs = self._run[slot](images)
                if after is not None:
                    after(outs)
            self._done[slot].record(stream)
        self._ticket_of[slot] = ticket
        return ticket

    def result(self, ticket: int):
        slot = ticket % self.depth
        if self._ticket_of[slot] != ticket:
            raise TsodError(f"InFlightDetector: ticket {ticket} is no longer resident (its slot was reused)")
        self._done[slot].synchronize()
        # the range word of THIS slot's conv launches (one small device read: ~10 us of host time per request, nothing on the
        # GPU's critical path).  Every slot has its own plan and its own word, so a set word means THIS step ran an fp16x2 layer
        # into non-finite accumulators (non-finite input): its outputs are garbage - never hand them out as valid - while the
        # other requests in flight are unaffected and keep their results.
        # (host=True: the word as the slot's forward published it into page-locked host memory at its end - a read of host memory, no
        #  device-to-host copy per request; the raise itself reads and clears the device word)
        if self.model.extractor.range_flag_raised(slot, host=True):
            self.model.extractor.raise_if_error(slot)
        return self._outputs[slot]

    def drain(self) -> None:
        """Wait for every step in flight (and surface a deferred proposal-layer error)."""
        for slot, t in enumerate(self._ticket_of):
            if t is not None:
                self._done[slot].synchronize()
        self.model.raise_if_error()
