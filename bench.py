#!/usr/bin/env python3
"""bench.py -- images/sec of the full Faster R-CNN ResNet-50 inference forward on MI355X.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no WORLD_SIZE in the environment this process only LAUNCHES: it starts N rank processes
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` as a child, before any GPU
call of its own) and exits with their code; under torchrun (WORLD_SIZE set) it is a rank.  One rank per GPU, backend
"nccl" (= RCCL over xGMI).

A step = one detector forward (NCHW->NHWC, 53 conv GEMMs, max pool, fused RPN conv + decode + top-k + NMS + pad, fused
RoI pool + mean, fused head GEMM, detection records) over one batch of synthetic 3x800x1333 images already resident in
HBM, plus - for N > 1 - the RCCL all-gather of the [B,300,6] detection records.  Each step is one HIP-graph launch.

Workloads: N = 1 -> BASELINE.json configs[1] (batch 1); N > 1 -> configs[4] (data-parallel, batch 8 per rank: 64 images
over 8 GPUs), weak scaling: every rank processes its own images, value = images of all ranks / max-over-ranks time.
`--batch` overrides the per-rank batch.

Two schedules are measured and BOTH are reported in the one JSON line:
  * serial     : one forward at a time on one stream (`serial`: images/s, ms per step = latency of a forward).  The
                 `roofline` object belongs to this schedule: algorithmic conv FLOPs / HIP-event time of one pass over
                 the conv launches in forward order (on their launch stream, tile table tuned for this schedule, cache
                 state and launch boundaries of a forward: the figure agrees with the rocprofv3 kernel trace committed
                 under profiles/); its kernel_ms_per_forward is <= serial.ms_per_step, the step it is part of.
  * in flight  : consecutive steps issued round-robin on --in-flight HIP streams (default 4, each with its own graph,
                 buffers and scratch): a batch-1 server with several requests in flight, the tail of one forward
                 overlapping the next one's kernels.  `value` / `ms_per_step` are this schedule's (with --in-flight 1
                 they are the serial numbers); `throughput_mode` gives its whole-step bound
                 (conv FLOPs per step / ms_per_step vs the f32-MFMA peak) - per-kernel times do not exist for
                 overlapped kernels.
Every timed region (exactly K steps between barrier + synchronize on both sides, max over ranks) is repeated
--repeats times (default 5); the median repeat is reported, min / max alongside.

  roofline         : per layer the autotuner picks the arithmetic (--precision auto): f32 MFMA (peak 157.3 TFLOP/s) or bf16x3
                     (three exact bf16 pieces per operand, six bf16 MFMAs per f32 product: peak 2516.8 / 6 = 419.5 TFLOP/s
                     f32-equivalent).  achieved = algorithmic f32 conv FLOPs / kernel time; peak = the FLOP-weighted
                     (harmonic) peak of the layers' arithmetics, so frac = sum of the layers' ideal matrix-pipe times /
                     measured time.  roofline_bf16x3 is the same measurement with a tile table tuned among f32 and bf16x3 only
                     (no fp16x2: the arithmetic of rounds 2-3); roofline_f32_mfma with every layer pinned to the f32-MFMA
                     kernels (the number comparable with round 1).
  roofline.traffic : HBM bytes per conv launch from FETCH_SIZE / WRITE_SIZE, collected by two short `rocprofv3 --pmc`
                     child runs of this script (rank 0, N = 1; --no-pmc skips them).
  cpu_baseline     : the CPU oracle (torch CPU ops + C nms / roi_pool restatement of the reference's path) timed on this
                     box's host cores on the same workload (rank 0, N = 1 only).
  --check          : (N > 1, on by default) the gathered [N*B,300,6] records of the last step are compared on rank 0 with
                     single-GPU forwards of every rank's images, POSITION-WISE (record i of image g against record i: boxes /
                     scores <= 1e-3, classes equal, no unmatched record; `bit_exact` says whether they are even identical):
                     rank 0 tunes the tile tables once and broadcasts them, so every rank sums in the same order.  A
                     mismatch fails the run.
  N > 1 start-up   : ONE autotune pass on rank 0 (objective = the headline schedule), broadcast as JSON together with the two
                     head-GEMM choices; no f32-MFMA leg, no second (serial-objective) pass; `tuning_seconds` is in the line.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
BF16_MFMA_PEAK_TFLOPS = 2516.8   # same guide: ~2.5 PF dense bf16 MFMA = 16x the f32 MFMA rate
# bf16x3 executes SIX bf16 MFMA products per f32 product: its roofline in f32-equivalent (algorithmic) FLOPs
BF16X3_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 6.0
FP16X2_PEAK_TFLOPS = BF16_MFMA_PEAK_TFLOPS / 3.0       # fp16 dense MFMA = the bf16 rate; three piece products per f32 product
R_POST = 300


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--repeats", type=int, default=5, help="how many times the K-step timed region is repeated (median reported)")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step (default: 1 at --gpus 1 = configs[1]; "
                    "8 at --gpus > 1 = configs[4], 64 images over 8 GPUs; 16 = configs[2])")
    ap.add_argument("--backbone", default="resnet50")
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=1333)
    ap.add_argument("--num-classes", type=int, default=80)
    ap.add_argument("--no-autotune", action="store_true")
    ap.add_argument("--precision", default="auto", choices=("f32", "bf16x3", "auto", "auto-bf16x3", "fp16x2"), help="arithmetic of the "
                    "conv GEMMs: f32 = v_mfma_f32_32x32x2_f32; bf16x3 = three exact bf16 pieces per operand, six products on "
                    "v_mfma_f32_32x32x16_bf16; fp16x2 (for the layers whose tile has it, candidates like auto) = two fp16 pieces of "
                    "2^e x per operand, three products on v_mfma_f32_32x32x16_f16, e from the abs-max the tensor's producers left "
                    "in its range words (per forward, inside the launches: any finite input range) - all three f32-accurate and "
                    "gated by the same parity suite; auto (default) = the autotuner picks per layer among all three; "
                    "auto-bf16x3 = among f32 and bf16x3 only (rounds 2-3)")
    ap.add_argument("--fuse-stem", default="auto", choices=("auto", "on", "off"), help="ResNet's conv1 + bn1 + PReLU + max pool (and the "
                    "NCHW -> NHWC pass) as one launch (tsod_stem_fp16x2): auto = FasterRCNN.tune times the backbone with and without")
    ap.add_argument("--fuse-bottleneck", default="auto", choices=("auto", "on", "off"), help="ResNet layer1's identity bottlenecks as one "
                    "launch each (tsod_bottleneck_fp16x2): auto = FasterRCNN.tune times one pass over the matrix launches with and "
                    "without and keeps the faster structure")
    ap.add_argument("--autotune-splits", default=None, help="comma list restricting the K-slice candidates of the autotuner")
    ap.add_argument("--autotune-in-sequence", type=int, default=None, help="serial tile table: the n fastest candidates of every layer "
                    "(timed in isolation) are timed again as launches of the whole conv sequence and the winner THERE is pinned "
                    "(cache state of a forward instead of self-warmed operands); 0 = isolated timing only.  Default 5 below batch 4 "
                    "(measured on one box: serial 556 -> 567 images/s, conv time 1.636 -> 1.599 ms), 0 from batch 4 on (no effect "
                    "there: 809 vs 810 images/s at batch 8 for 13 s more tuning)")
    ap.add_argument("--autotune-in-flight-refine", type=int, default=None, help="in-flight tile table: the n fastest candidates of every "
                    "layer are tried again while all --in-flight streams run the whole conv sequence, staggered around it (the load "
                    "of a pipelined server), and the one that makes those passes fastest is pinned; 0 = first look only (copies of "
                    "one layer side by side).  Default 3 below batch 4, 0 from batch 4 on")
    ap.add_argument("--in-flight", type=int, default=4, help="steps in flight: consecutive steps are issued round-robin on this "
                    "many HIP streams, each with its own graph and buffers (request-level pipelining of a batch-1 server)")
    ap.add_argument("--tiles-file", default=None, help="JSON cache of the autotuned (tile, split) tables {'serial': [...], "
                    "'in_flight': [...]}: loaded if present, else written after autotuning (keeps profiler runs free of "
                    "tuning launches)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of HIP-graph replays, serial schedule only "
                    "(for rocprofv3 --pmc passes: every dispatch then carries its own counter sample)")
    ap.add_argument("--no-fuse-shortcut", action="store_true", help="ResNet: one launch per conv (the projection shortcuts as their "
                    "own GEMMs + residual reads) instead of stacking them into their block's last 1x1 conv (A/B switch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc child runs behind roofline.traffic")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--dump-layers", default=None, help="write the serial plan's conv launches (name, FLOPs, algorithmic bytes, tile, "
                    "K-slice slab bytes, HIP-event time) as JSON: the per-layer columns of scripts/summarize_pmc.py")
    ap.add_argument("--cpu-reps", type=int, default=8)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--check", action="store_true", default=None, help="N > 1: compare the gathered records with single-GPU forwards "
                    "on rank 0 (the DEFAULT for N > 1; --no-check skips it)")
    ap.add_argument("--no-check", dest="check", action="store_false")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI; the real path) | gloo (rehearsal of "
                    "the N>1 control flow on a box with fewer GPUs than ranks: ranks share devices, gather goes through host)")
    ap.add_argument("--collective", default="torch", choices=("torch", "tsod"), help="N > 1 with the nccl backend: who issues the "
                    "all-gather of the detection records: torch.distributed's all_gather_into_tensor (default) or the C-ABI's "
                    "tsod_allgather_f32 on a communicator of its own (include/tsod.h; the same ncclAllGather either way)")
    ap.add_argument("--rehearse-cpu", action="store_true", help="launcher / rendezvous / gather / timing / --check control flow "
                    "with fabricated records and NO GPU work (gloo; for the CPU test of the N>1 path - never a measurement)")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------- launcher
def launch_ranks(args, argv):
    """--gpus N > 1 outside torchrun: start the N ranks as a child job and exit with its code.  Nothing in this process has
    touched the GPU (importing torch does not); the ranks are fresh interpreters."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this pool (RCCL needs it)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


# ----------------------------------------------------------------------------------------------- measurements
def conv_event_times(plan, reps=5):
    """HIP-event duration of every conv launch of the plan (ms), on the launch stream."""
    from two_stage_object_detection_amd._ffi import stream_ptr
    out = []
    s = stream_ptr()
    for st in plan.gemm_steps:
        if hasattr(st, "x"):
            plan._refresh_amax_for(st)       # (the pooled input buffer may hold a later tensor by now: words to match its bytes)
        st.fn(*st.args, s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            st.fn(*st.args, s)
        e1.record()
        e1.synchronize()
        out.append(e0.elapsed_time(e1) / reps)
    plan.clear_range_flag()      # (these launches ran on whatever the pooled buffers held, not on a forward's activations)
    return out


def conv_sequence_time(plan, reps=10):
    """HIP-event time (ms) of ONE pass over the plan's matrix launches (convs and one-launch bottlenecks) in forward order, back
    to back on the launch stream, averaged over `reps` passes: every layer finds its input where the previous launch left it and
    its weights as cold as a forward leaves them - the state the kernels run in inside the serial graph (the per-layer figure of
    conv_event_times, five repeats of one launch on hot operands, reads ~10 % lower than the rocprofv3 kernel trace of the
    forward; this one agrees with it).  Launch boundaries between the kernels are inside the interval, as they are inside a
    forward.  (engine.Plan.sequence_time: FasterRCNN.tune uses the same figure to choose the launch structure.)"""
    return plan.sequence_time(reps)


def conv_in_sequence_times(plan, reps=7):
    """Per conv launch: HIP-event time (ms, median over `reps` passes) INSIDE one pass over the conv sequence - every launch
    bracketed by its own pair of events while the sequence runs in forward order.  The per-layer companion of
    conv_sequence_time (the isolated figure of conv_event_times re-runs one launch on hot operands and reads up to 3x high on
    the HBM-bound layer1 launches, whose input then comes from HBM instead of being written just before)."""
    from two_stage_object_detection_amd._ffi import stream_ptr
    s = stream_ptr()
    n = len(plan.gemm_steps)
    samples = [[] for _ in range(n)]
    for r in range(reps + 1):
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for (e0, e1), st in zip(ev, plan.gemm_steps):
            e0.record()
            st.fn(*st.args, s)
            e1.record()
        ev[-1][1].synchronize()
        if r:                                                   # (the first pass warms up)
            for i, (e0, e1) in enumerate(ev):
                samples[i].append(e0.elapsed_time(e1))
    plan.clear_range_flag()
    return [statistics.median(v) for v in samples]


def cpu_baseline(sd, backbone, x_cpu, reps):
    """(the oracle's outputs on x_cpu, the cpu_baseline object).  The outputs are what `parity` checks the timed plans against:
    the line validates what it times."""
    import oracle
    # the GPU box exposes all host cores but grants this job a 16-core share (worker pools must be sized to it)
    torch.set_num_threads(min(os.cpu_count() or 1, int(os.environ.get("TSOD_CPU_THREADS", "16"))))
    with torch.inference_mode():
        # warm-up (and, with reps = 0, the reference of `parity` only); the RPN's debug record says what lies just beyond the
        # proposal list's cut (`parity`'s tie rule: testing.cutoff_candidates)
        from two_stage_object_detection_amd.testing import cutoff_candidates
        from oracle.box import proposal_counts
        n_post = proposal_counts("training")[1]
        outs, dbg = oracle.detector_forward(sd, x_cpu, backbone=backbone, return_debug=True)
        cpu_baseline.cutoff = cutoff_candidates(dbg, n_post)
        cpu_baseline.exact_outs, dbg = oracle.detector_forward(sd, x_cpu, backbone=backbone, exact=True, return_debug=True)   # (untimed: `parity.exact`)
        cpu_baseline.exact_cutoff = cutoff_candidates(dbg, n_post)
        del dbg
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            outs = oracle.detector_forward(sd, x_cpu, backbone=backbone)
            ts.append(time.perf_counter() - t0)
    if not ts:
        return outs, None
    med = statistics.median(ts)
    return outs, {"value": x_cpu.shape[0] / med, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} timed forwards (median) of the same workload: batch {x_cpu.shape[0]} x 3x{x_cpu.shape[2]}x"
                      f"{x_cpu.shape[3]}, {backbone} detector, torch {torch.__version__} CPU f32 + oracle/box_ops.c",
            "seconds_per_forward": med}


def step_algorithmic_bytes(st):
    """Input + output + weights (+ residual, + second source) of one conv launch, each counted once (f32); a one-launch
    bottleneck: its input, its output and its three weight matrices (the intermediates never exist in memory)."""
    if hasattr(st, "algorithmic_bytes"):
        return st.algorithmic_bytes
    d = st.desc
    cin = sum(d.seg_len[i] for i in range(d.n_seg))
    tot = 4 * (d.N * d.H * d.W * cin + d.N * d.OH * d.OW * d.Cout + d.Cout * d.KH * d.KW * cin)
    if d.c2 > 0:                                      # second source of a fused shortcut: the pixels it taps + its weights
        tot += 4 * (d.N * d.OH * d.OW * d.c2 + d.Cout * d.c2)
    if d.res_pitch > 0:
        tot += 4 * d.N * d.OH * d.OW * d.Cout
    return tot


def conv_algorithmic_bytes(plan):
    return sum(step_algorithmic_bytes(st) for st in plan.gemm_steps)


def dump_layers(plan, path, conv_ms=None, seq_ms=None):
    """Per matrix launch of the plan, in launch order: what scripts/summarize_pmc.py needs to put names, algorithmic bytes and
    K-slice slab bytes beside the per-dispatch counters."""
    from ctypes import byref
    from two_stage_object_detection_amd._ffi import TILE_NAMES, lib
    from two_stage_object_detection_amd.engine import FusedStep
    rows = []
    for i, st in enumerate(plan.gemm_steps):
        d = st.desc
        times = {"seq_us": None if seq_ms is None else round(seq_ms[i] * 1e3, 2),          # inside the sequence (the one to quote)
                 "event_us": None if conv_ms is None else round(conv_ms[i] * 1e3, 2)}      # isolated repeats on hot operands
        if isinstance(st, FusedStep) and st is getattr(plan, "stem_step", None):
            oh, ow = (d.H - 1) // 2 + 1, (d.W - 1) // 2 + 1
            rows.append({"name": st.name, "flops": int(st.flops), "algorithmic_bytes": int(st.algorithmic_bytes), "tile": "stem4x16",
                         "split_k": 1, "precision": 2, "slab_bytes": 0, "M": int(d.N * oh * ow), "Cout": 64, "K": 147, **times})
            continue
        if isinstance(st, FusedStep):
            rows.append({"name": st.name, "flops": int(st.flops), "algorithmic_bytes": int(st.algorithmic_bytes), "tile": "bottleneck10x16",
                         "split_k": 1, "precision": 2, "slab_bytes": 0, "M": int(d.N * d.H * d.W), "Cout": int(d.Cout),
                         "K": int(d.Cin + 9 * d.Cmid + d.Cmid), **times})
            continue
        ws = int(lib().tsod_conv2d_workspace_bytes(byref(d)))
        rows.append({"name": st.name, "flops": int(st.flops), "algorithmic_bytes": int(step_algorithmic_bytes(st)),
                     "tile": TILE_NAMES[int(d.tile)], "split_k": int(d.split_k), "precision": int(d.precision),
                     "slab_bytes": max(0, ws - 256 * 1024) if ws else 0, "M": int(d.N * d.OH * d.OW), "Cout": int(d.Cout),
                     "K": int(d.KH * d.KW * sum(d.seg_len[j] for j in range(d.n_seg)) + max(0, int(d.c2))), **times})
    json.dump(rows, open(path, "w"), indent=0)


def pmc_child(args, dev):
    """Body of the `rocprofv3 --pmc` child: build the plan with the parent's tile choices, then issue the conv
    launches of one forward twice, eagerly, so that every dispatch carries its own counter sample."""
    from two_stage_object_detection_amd._ffi import lib, stream_ptr
    from two_stage_object_detection_amd.testing import synthetic_detector
    model, _ = synthetic_detector(args.backbone, num_classes=args.num_classes, seed=0, conditioned=True)
    model = model.to(dev).eval()
    x = torch.rand(args.batch, 3, args.height, args.width, generator=torch.Generator().manual_seed(1234)).to(dev)
    with torch.inference_mode():
        table = json.load(open(args.tiles_file)) if args.tiles_file and os.path.exists(args.tiles_file) else None
        if table is not None:
            model.extractor.set_structure(table)
        plan = model.extractor._plan_for(x)                 # packs weights, launches no conv
        if table is not None:
            plan.import_tiles(table["serial"])
        torch.cuda.synchronize()
        L, s = lib(), stream_ptr()
        model(x)                                            # a real forward: warms caches / code objects and leaves this input's
        torch.cuda.synchronize()                            # activations and range words behind (the fp16x2 scales of the sample)
        for st in plan.gemm_steps:                          # the sample: one pass over the matrix launches
            st.fn(*st.args, s)
        torch.cuda.synchronize()


def pmc_traffic(args, tiles, n_launches):
    """HBM bytes per conv launch from rocprofv3's FETCH_SIZE / WRITE_SIZE (KiB; separate passes: both do not fit
    the TCC counter budget at once).  gfx950 tallies a wide coalesced read at half its size, so reads are doubled
    (MI355X_MICROARCH.md, HBM section).  Returns (bytes_per_launch or None, note)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    prof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(prof):
        return None, "rocprofv3 not found"
    work = tempfile.mkdtemp(prefix="tsod_pmc_", dir="/tmp")
    try:
        tiles_path = os.path.join(work, "tiles.json")
        json.dump({"serial": tiles["serial"], "fuse_bottleneck": bool(tiles.get("fuse_bottleneck", False)),
                   "fuse_projection": bool(tiles.get("fuse_projection", False)),
                   "fuse_stem": bool(tiles.get("fuse_stem", False))}, open(tiles_path, "w"))
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        sums = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(work, counter)
            cmd = [prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "--tiles-file", tiles_path,
                   "--backbone", args.backbone, "--batch", str(args.batch), "--height", str(args.height),
                   "--width", str(args.width), "--num-classes", str(args.num_classes)]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
            files = glob.glob(os.path.join(out, "*", "*counter_collection.csv"))
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode})"
            rows = [(int(q["Dispatch_Id"]), float(q["Counter_Value"])) for q in csv.DictReader(open(files[0]))
                    if q["Counter_Name"] == counter and ("conv_igemm_kernel" in q["Kernel_Name"]
                                                         or "conv_dma_kernel" in q["Kernel_Name"]
                                                         or "bottleneck_kernel" in q["Kernel_Name"] or "stem_kernel" in q["Kernel_Name"])]
            rows.sort()
            if len(rows) < 2 * n_launches:
                return None, f"unexpected dispatch count {len(rows)} in the {counter} pass"
            sums[counter] = sum(v for _, v in rows[-n_launches:])             # the conv-only pass behind the warm-up forward
        total = (2.0 * sums["FETCH_SIZE"] + sums["WRITE_SIZE"]) * 1024.0
        return total / n_launches, None
    except Exception as e:                                                     # noqa: BLE001 - never fail the bench on this leg
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(work, ignore_errors=True)


def parity_of(gpu_outs, ref_outs, exact_outs=None, ref_cutoff=None, exact_cutoff=None):
    """`parity` of the JSON line: the outputs the TIMED plans produced on this rank's images (the serial graph and one slot of
    the in-flight server, tile tables and arithmetic exactly as timed) against the CPU oracle on the same images - boxes /
    scores <= 1e-3, classes equal, every row paired one to one.  Two references: the oracle's float32 run (the reference's own
    arithmetic: the top-level figures) and, `exact`, the same oracle evaluated in float64 (oracle.detector_forward(exact=True):
    the exact value of the reference's math, every tensor entering the f32 box code rounded once).  `ok` = within 1e-3 of the
    float32 run; or - a deep net whose two f32 pipelines are each most of 1e-3 from the truth (HarDNet-68: the reference's own
    CPU run is 7e-4 from its float64 evaluation, scripts/config4_truth.py) - within 1e-3 of the exact evaluation AND within
    1.25e-3 of the float32 run with every class equal and every score / offset within 1e-3.  ``*_cutoff``
    (testing.cutoff_candidates of each reference's RPN debug record): a row whose only difference is WHICH of two candidates with
    fg scores within 1e-6 of each other took the last place of an image's proposal list is reported as `rows_tied_at_cutoff`,
    not as unmatched (testing.compare_detector_outputs).  A line that is not ok exits non-zero."""
    from two_stage_object_detection_amd.testing import compare_detector_outputs
    keys = ("rows", "rows_positional_mismatch", "rows_unmatched", "rows_tied_at_cutoff", "max_tie_score_gap", "class_mismatch", "max_abs_roi",
            "max_abs_score", "max_abs_cls_loc", "ok")

    def legs_of(ref, cutoff, atol=1e-3):
        legs = {}
        for name, outs in gpu_outs.items():
            rep = compare_detector_outputs(outs, ref, atol=atol, ref_cutoff=cutoff)
            legs[name] = {k: (round(rep[k], 10) if isinstance(rep.get(k), float) else rep.get(k)) for k in keys}
        return legs

    def summary(legs):
        worst = lambda k: max(v[k] for v in legs.values())      # noqa: E731
        return {"rows_unmatched": worst("rows_unmatched"), "rows_tied_at_cutoff": worst("rows_tied_at_cutoff"),
                "max_tie_score_gap": worst("max_tie_score_gap"), "class_mismatch": worst("class_mismatch"),
                "max_abs_roi": worst("max_abs_roi"), "max_abs_score": worst("max_abs_score"), "within_atol": all(v["ok"] for v in legs.values())}
    legs = legs_of(ref_outs, ref_cutoff)
    out = dict(summary(legs), atol=1e-3, matching="one-to-one",
               against="CPU oracle (oracle.detector_forward, float32: the reference's arithmetic) on the timed input, rank 0's images")
    ok = out["within_atol"]
    if exact_outs is not None:
        ex = summary(legs_of(exact_outs, exact_cutoff))
        ex["against"] = "the same oracle evaluated in float64 (exact=True): the exact value of the reference's math, rounded once"
        out["exact"] = ex
        if not ok and ex["within_atol"]:
            wide = summary(legs_of(ref_outs, ref_cutoff, atol=1.25e-3))
            ok = wide["rows_unmatched"] == 0 and wide["class_mismatch"] == 0 and wide["max_abs_score"] <= 1e-3 and wide["max_abs_roi"] <= 1.25e-3
            out["float32_run_at_1.25e-3"] = wide
    out["ok"] = bool(ok)
    out["ok_rule"] = ("within atol of the float32 run, or within atol of the float64 evaluation and within 1.25e-3 of the float32 run (classes equal, "
                      "scores <= atol); rows_tied_at_cutoff: rows that differ only in which of two candidates with fg scores within 1e-6 took "
                      "the last place of a proposal list (not counted as unmatched)")
    out["legs"] = legs
    return out


class Timer:
    """The contract's timed region: exactly K steps between barrier + synchronize on both sides, max over ranks;
    repeated R times."""

    def __init__(self, world, sync, reduce_device):
        self.world, self.sync, self.reduce_device = world, sync, reduce_device

    def region(self, step, steps):
        self.sync()
        if self.world > 1:
            dist.barrier()
        self.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        self.sync()
        if self.world > 1:
            dist.barrier()
        self.sync()
        elapsed = time.perf_counter() - t0
        if self.world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=self.reduce_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed

    def measure(self, step, steps, warmup, repeats):
        for _ in range(max(warmup, 1)):
            step()
        ts = sorted(self.region(step, steps) for _ in range(max(1, repeats)))
        per = [t / steps * 1e3 for t in ts]
        return {"ms_per_step": statistics.median(per), "min": per[0], "max": per[-1], "n": len(per), "steps": steps}


def compare_records(got, ref, atol=1e-3):
    """[n,300,6] detection records (x1,y1,x2,y2,score,class) of the gathered result against single-GPU forwards of the same
    images on rank 0.  Every rank runs rank 0's tile table and head-GEMM choices (broadcast at start-up), i.e. the same
    kernels in the same summation order, so the comparison is POSITION-WISE: record i of image g against record i, boxes and
    scores within atol, classes equal, nothing unmatched.  `bit_exact` reports whether the two are even identical."""
    n, R, _ = got.shape
    d_box = (got[..., :4] - ref[..., :4]).abs().amax(-1)                            # [n,R]
    d_score = (got[..., 4] - ref[..., 4]).abs()
    bad_row = (d_box > atol) | (d_score > atol) | ~torch.isfinite(d_box) | ~torch.isfinite(d_score)
    cls_bad = got[..., 5] != ref[..., 5]
    off = sorted(set(torch.nonzero((bad_row | cls_bad).any(dim=1)).flatten().tolist()))
    ok_rows = ~bad_row
    return {"ok": not off, "images": int(n), "records_unmatched": int(bad_row.sum()),
            "class_mismatch_on_matched": int((cls_bad & ok_rows).sum()),
            "max_abs_box_on_matched": float(d_box[ok_rows].max()) if ok_rows.any() else 0.0,
            "max_abs_score_on_matched": float(d_score[ok_rows].max()) if ok_rows.any() else 0.0,
            "bit_exact": bool(torch.equal(got, ref)), "matching": "position-wise", "images_off": off}


def broadcast_json(obj, rank, world):
    """rank 0's JSON-able object on every rank (torch.distributed object broadcast; identity for one process)."""
    if world <= 1:
        return obj
    box = [json.dumps(obj) if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return json.loads(box[0])


# ----------------------------------------------------------------------------------------------- rehearsal (no GPU)
def rehearse_cpu(args, rank, world):
    """The N>1 control flow on CPU ranks (gloo): same barrier / timed loop / gather / max-over-ranks / --check code as
    the real run, with fabricated detection records instead of forwards.  Exists for tests/test_dist_gloo.py."""
    from two_stage_object_detection_amd.dist import all_gather_detections, shard_range
    B = args.batch
    lo, hi = shard_range(world * B, rank, world)
    # the start-up protocol of the real run: rank 0 "tunes" (here: fabricates a table only it knows), every rank gets it
    t0 = time.perf_counter()
    mine = {"serial": [["conv1", 14, -1, 1], ["layer1.0.conv1", 8, 1, 1]], "heads": {"rpn": {"8x25x42": [3, 1, 1]}, "head": {"2400": [8, 2, 1]}},
            "fuse_stem": True, "fuse_bottleneck": False, "fuse_projection": False,
            "parity_budget": {"budget_px": 5e-4, "before_px": 9e-4, "after_px": 4e-4, "demoted": ["layer3.0.conv1"], "held": True},
            "tuned_by_rank": 0} if rank == 0 else {"tuned_by_rank": rank}
    tiles = broadcast_json(mine, rank, world)
    tiles_ok = (tiles.get("tuned_by_rank") == 0 and tiles["serial"][0] == ["conv1", 14, -1, 1] and tiles["heads"]["head"]["2400"] == [8, 2, 1]
                and tiles.get("fuse_stem") is True and tiles.get("fuse_bottleneck") is False        # the launch structure ...
                and tiles.get("parity_budget", {}).get("demoted") == ["layer3.0.conv1"])            # ... and the demotions travel with the table
    flags = [None] * world
    if world > 1:
        dist.all_gather_object(flags, bool(tiles_ok))
    else:
        flags = [bool(tiles_ok)]
    tuning_s = time.perf_counter() - t0
    det = torch.stack([torch.full((R_POST, 6), float(g)) for g in range(lo, hi)])
    gathered = torch.empty((world * B, R_POST, 6))
    timer = Timer(world, lambda: None, "cpu")
    res = timer.measure(lambda: all_gather_detections(det, out=gathered), args.steps, args.warmup, args.repeats)
    check = None
    if args.check:
        ref = torch.stack([torch.full((R_POST, 6), float(g)) for g in range(world * B)])
        check = compare_records(gathered, ref)
    if rank == 0:
        line = {"metric": "rehearsal of the N>1 control flow (no GPU work, not a measurement)", "value": None,
                "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": round(res["ms_per_step"], 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32", "data": "rehearsal", "rehearsal": True, "tiles_broadcast_ok_per_rank": flags,
                "tuning_seconds": {"rank0": round(tuning_s, 3)},
                "config": {"workload": "fabricated [B,300,6] records", "global_batch": world * B, "parallelism": f"dp{world}",
                           "collective": f"all_gather_into_tensor [{world * B},300,6] f32 (gloo)"}, "check": check}
        print(json.dumps(line), flush=True)
    return 0 if ((check is None or check["ok"]) and all(flags)) else 3


# ----------------------------------------------------------------------------------------------- rank body
def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args, argv))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world != args.gpus and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE", file=sys.stderr)
    n_gpus = world
    if args.batch is None:
        args.batch = 8 if world > 1 else 1
    B = args.batch
    if args.check is None:
        args.check = world > 1                              # a driver-run --gpus N validates what it gathers
    if args.autotune_in_sequence is None:
        args.autotune_in_sequence = 5 if B < 4 else 0
    if args.autotune_in_flight_refine is None:
        args.autotune_in_flight_refine = 3 if B < 4 else 0

    if args.rehearse_cpu:
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        rc = rehearse_cpu(args, rank, world)
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        sys.exit(rc)

    n_dev = torch.cuda.device_count()                     # does not initialise the GPU
    if n_dev == 0:
        sys.exit("bench.py needs a GPU (the package has no CPU path)")
    if world > 1 and args.dist_backend == "nccl" and n_dev < world:
        sys.exit(f"bench.py: {world} ranks but only {n_dev} visible GPU(s): RCCL needs one GPU per rank "
                 f"(use --dist-backend gloo to rehearse the control flow on fewer GPUs)")
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    nccl = args.dist_backend == "nccl"
    if world > 1:
        if nccl:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    if args.pmc_child:
        pmc_child(args, dev)
        return

    from two_stage_object_detection_amd import hip_ops
    from two_stage_object_detection_amd.dist import all_gather_detections
    from two_stage_object_detection_amd.serving import InFlightDetector
    from two_stage_object_detection_amd.testing import synthetic_detector

    # (HarDNet: BatchNorm statistics pre-computed for these weights - configs/synthetic_bn_*.npz, scripts/make_synthetic_bn_stats.py;
    #  with identity BN a random-init HarDNet maps every image to a constant feature map: a degenerate workload)
    model, sd = synthetic_detector(args.backbone, num_classes=args.num_classes, seed=0, conditioned=True)
    model = model.to(dev).eval()
    if args.precision == "bf16x3":
        model.extractor.set_conv_precision("bf16x3")
    if args.no_fuse_shortcut:
        model.extractor.fuse_shortcut = False

    def images(r):
        return torch.rand(B, 3, args.height, args.width, generator=torch.Generator().manual_seed(1234 + r))
    x_cpu = images(rank)
    x = x_cpu.to(dev)
    n_fly = max(1, args.in_flight)
    timer = Timer(world, torch.cuda.synchronize, dev if nccl else "cpu")

    with torch.inference_mode():
        model(x)                                                       # builds the plan
        torch.cuda.synchronize()
        plan = model.extractor._plan_for(x)                           # (the measurement legs below time its launches)
        from two_stage_object_detection_amd.engine import step_precision

        def use_table(key):
            """The extractor's plan for x with table `key` pinned.  The serial / in-flight tables may belong to the launch structure
            with one-launch bottlenecks and stem (tiles["fuse_bottleneck"], tiles["fuse_stem"]); the f32 / bf16x3 comparison legs
            never do."""
            model.extractor.set_structure(tiles if key in ("serial", "in_flight") else None)
            model(x)
            pl = model.extractor._plan_for(x)
            pl.import_tiles(tiles[key])
            return pl
        default_tiles = plan.export_tiles()
        tiles = {"serial": default_tiles, "in_flight": default_tiles}
        splits = [int(v) for v in args.autotune_splits.split(",")] if args.autotune_splits else None
        tiles_loaded = bool(args.tiles_file and os.path.exists(args.tiles_file))
        t_tune = time.perf_counter()
        if not tiles_loaded and not args.no_autotune:
            # bring the chip into the state the tuned kernels will run in before anything is timed: on a box that has just started,
            # the first candidates of every layer see boost clocks the later ones do not (the arithmetics are timed one after the
            # other), and the serial table came out 0.2 ms slower on the first run of a box than on the next ones
            t_w = time.perf_counter()
            while time.perf_counter() - t_w < 1.5:
                conv_sequence_time(plan, reps=20)
        precs = {"f32": (0,), "bf16x3": (1,), "auto-bf16x3": (0, 1), "auto": (0, 1, 2), "fp16x2": (0, 1, 2)}[args.precision]
        fuse_arg = {"auto": "auto", "on": True, "off": False}[args.fuse_bottleneck]
        stem_arg = {"auto": "auto", "on": True, "off": False}[args.fuse_stem]
        # The tuning goes through the public call (FasterRCNN.tune): what a user of the module surface gets is what is timed.
        # fp16x2 needs no calibration pass: its activation scale follows every tensor per forward (range words).
        if tiles_loaded:
            tiles = json.load(open(args.tiles_file))
        elif not args.no_autotune and world > 1:
            # N > 1: ONE pass on rank 0 with the headline schedule's objective, shipped to every rank below (all ranks then
            # run the same kernels in the same summation order; start-up stays far inside the driver's limit)
            if rank == 0:
                tiles = model.tune(x, precisions=precs, in_flight=n_fly, schedules=("in_flight",) if n_fly > 1 else ("serial",),
                                   splits=splits, in_sequence=0, in_flight_refine=0, verbose=args.verbose, fuse_bottleneck=fuse_arg, fuse_stem=stem_arg)
                tiles["serial"] = tiles["in_flight"] = tiles.get("in_flight") or tiles["serial"]
        elif not args.no_autotune:
            tiles = model.tune(x, precisions=precs, in_flight=n_fly, splits=splits, in_sequence=args.autotune_in_sequence,
                               in_flight_refine=args.autotune_in_flight_refine, verbose=args.verbose and rank == 0, fuse_bottleneck=fuse_arg,
                               fuse_stem=stem_arg)
            tiles.setdefault("in_flight", tiles["serial"])
        tuning_s = time.perf_counter() - t_tune
        if world > 1:
            tiles = broadcast_json(tiles, rank, world)                 # rank 0's tables on every rank
        model.set_head_choices(tiles.get("heads"))                     # (persisted / broadcast choices: no tuning launches here)
        if world > 1:
            all_tuning = [None] * world
            dist.all_gather_object(all_tuning, round(tuning_s, 2))
        else:
            all_tuning = [round(tuning_s, 2)]
        if args.tiles_file and rank == 0 and not tiles_loaded and not args.no_autotune:
            json.dump(tiles, open(args.tiles_file, "w"))

        gathered = [torch.empty((world * B, R_POST, 6), dtype=torch.float32, device=dev if nccl else "cpu")
                    for _ in range(n_fly)] if world > 1 else None

        tsod_comm = None
        if world > 1 and nccl and args.collective == "tsod":
            from two_stage_object_detection_amd.dist import TsodCommunicator
            tsod_comm = TsodCommunicator(rank=rank, world=world)      # id from rank 0 over the torch.distributed group

        def make_step(server, depth):
            """One step of the timed loop.  With several steps in flight (the headline) a step is what a server does per request:
            COLLECT the result of the request that last used the slot (InFlightDetector.result: wait for that step's event, read
            the slot's range word - host work that overlaps the other slots' kernels), then submit the next one into it.  The
            serial schedule queues its forwards back to back on one stream (the latency chain of the kernels, no host round trip
            between two forwards); its results are collected behind the timed region."""
            collect = depth > 1 and hasattr(server, "_ticket_of")

            def step():
                slot = server._next % depth
                if collect and server._ticket_of[slot] is not None:
                    server.result(server._ticket_of[slot])
                if world > 1:                           # the gather of step i is ordered behind step i on ITS stream
                    if tsod_comm is not None:
                        return server.submit(after=lambda outs: tsod_comm.all_gather(outs[4], out=gathered[slot]))
                    return server.submit(after=lambda outs: all_gather_detections(
                        outs[4] if nccl else outs[4].cpu(), out=gathered[slot]))
                return server.submit()
            return step

        # ---- the f32-MFMA kernels alone (every layer pinned to precision f32): the roofline comparable with round 1
        f32_leg = None
        if world == 1 and not args.no_autotune and not args.no_graph and args.precision != "f32" and not (tiles_loaded and "f32" not in tiles):
            if "f32" not in tiles:
                tiles["f32"] = model.tune(x, precisions=(0,), schedules=("serial",), splits=splits, in_sequence=0, heads=False,
                                          fuse_bottleneck=False)["serial"]
                if args.tiles_file and rank == 0:
                    json.dump(tiles, open(args.tiles_file, "w"))
            plan = use_table("f32")
            ms32 = conv_sequence_time(plan)
            fl32 = sum(st.flops for st in plan.gemm_steps)
            f32_leg = {"bound": "mfma", "achieved": round(fl32 / (ms32 * 1e-3) / 1e12, 3), "peak": F32_MFMA_PEAK_TFLOPS,
                       "unit": "TFLOP/s", "frac": round(fl32 / (ms32 * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                       "kernel_ms_per_forward": round(ms32, 4), "dtype": "f32 (v_mfma_f32_32x32x2_f32)",
                       "note": "the same conv launches with every layer pinned to the f32-MFMA kernels (one pass in forward order, "
                               "HIP events); not the timed path when --precision auto picks bf16x3"}
        # ---- the same with the candidates of rounds 2-3 (f32 / bf16x3 per layer, no fp16x2): what the third arithmetic buys
        bf_leg = None
        if world == 1 and not args.no_autotune and not args.no_graph and 2 in precs and not (tiles_loaded and "bf16x3" not in tiles):
            if "bf16x3" not in tiles:
                tiles["bf16x3"] = model.tune(x, precisions=(0, 1), schedules=("serial",), splits=splits,
                                             in_sequence=args.autotune_in_sequence, heads=False, fuse_bottleneck=False)["serial"]
                if args.tiles_file and rank == 0:
                    json.dump(tiles, open(args.tiles_file, "w"))
            plan = use_table("bf16x3")
            msb = conv_sequence_time(plan)
            flb = sum(st.flops for st in plan.gemm_steps)
            prb = [step_precision(st) for st in plan.gemm_steps]
            idb = sum(st.flops / ((BF16X3_PEAK_TFLOPS if pr == 1 else F32_MFMA_PEAK_TFLOPS) * 1e12) * 1e3 for st, pr in zip(plan.gemm_steps, prb))
            bf_leg = {"bound": "mfma", "achieved": round(flb / (msb * 1e-3) / 1e12, 3), "peak": round(flb / (idb * 1e-3) / 1e12, 1),
                      "unit": "TFLOP/s", "frac": round(idb / msb, 4), "kernel_ms_per_forward": round(msb, 4),
                      "dtype": "bf16x3 / f32 MFMA per layer (no fp16x2)",
                      "note": "the same conv launches with a tile table tuned among f32 and bf16x3 only (one pass in forward order, HIP "
                              "events): the arithmetic of rounds 2-3, not the timed path when fp16x2 is among the candidates"}
        # ---- schedule 1: strictly serial (one graph, one stream) + the per-kernel roofline that belongs to it
        plan = use_table("serial")
        conv_ms = conv_event_times(plan)
        conv_seq_ms = conv_sequence_time(plan)
        if args.dump_layers and rank == 0:
            dump_layers(plan, args.dump_layers, conv_ms, conv_in_sequence_times(plan))
        conv_flops = sum(st.flops for st in plan.gemm_steps)
        algo_bytes = conv_algorithmic_bytes(plan)
        precs = [step_precision(st) for st in plan.gemm_steps]
        has_stem = getattr(plan, "stem_step", None) is not None
        n_fused = len(plan.fused_steps) - (1 if has_stem else 0)
        flops_bf = sum(st.flops * (6 if pr == 1 else 3) for st, pr in zip(plan.gemm_steps, precs) if pr >= 1) / 6.0
        peak_of = {0: F32_MFMA_PEAK_TFLOPS, 1: BF16X3_PEAK_TFLOPS, 2: FP16X2_PEAK_TFLOPS}
        ideal_ms = sum(st.flops / (peak_of[pr] * 1e12) * 1e3 for st, pr in zip(plan.gemm_steps, precs))
        if args.no_graph:
            n_fly = 1

            class _Eager:                                   # the serial schedule as plain launches (profiling aid)
                _next = 0

                def submit(self, after=None):
                    outs = model(x)
                    det = hip_ops.detections(outs[0], outs[1], outs[2])
                    if after is not None:
                        after(tuple(outs) + (det,))
                    self._next += 1

                def drain(self):
                    torch.cuda.synchronize()
            serial_server = _Eager()
        else:
            serial_server = InFlightDetector(model, x, depth=1, tiles=tiles["serial"])
        serial = timer.measure(make_step(serial_server, 1), args.steps, args.warmup, args.repeats)
        serial_server.drain()
        timed_outs = {}                                                   # what the timed plans computed on x (checked in `parity`)
        if rank == 0 and not args.no_graph:
            timed_outs["serial"] = [o.cpu() for o in serial_server.result(serial_server.submit())[:4]]
        # ---- schedule 2: --in-flight steps overlapped on as many streams (the default headline)
        if n_fly > 1:
            server = InFlightDetector(model, x, depth=n_fly, tiles=tiles["in_flight"])
            fly = timer.measure(make_step(server, n_fly), args.steps, args.warmup, args.repeats)
            server.drain()
            if rank == 0:                                                  # the slot the next request would use
                timed_outs[f"in_flight_slot{server._next % n_fly}"] = [o.cpu() for o in server.result(server.submit())[:4]]
                server.drain()
        else:
            server, fly = serial_server, serial
        model.raise_if_error()
        if rank == 0 and args.no_graph:
            timed_outs["serial_eager"] = [o.cpu() for o in model(x)]
            model.raise_if_error()

        check = None
        if args.check and world > 1:
            last = gathered[(server._next - 1) % n_fly].to(dev)
            if rank == 0:
                ref = []
                for r in range(world):                                 # single-GPU forwards of every rank's images
                    outs = model(images(r).to(dev))
                    ref.append(hip_ops.detections(outs[0], outs[1], outs[2]).clone())
                check = compare_records(last.cpu(), torch.cat(ref).cpu())
            flag = torch.tensor([1 if (check is None or check["ok"]) else 0], device=dev if nccl else "cpu")
            dist.broadcast(flag, src=0)
            if int(flag.item()) == 0:
                if rank == 0:
                    print(json.dumps({"check": check}), file=sys.stderr, flush=True)
                dist.barrier()
                dist.destroy_process_group()
                sys.exit(3)

    exit_code = 0
    if rank == 0:
        head = fly
        conv_total_ms = conv_seq_ms                                       # one pass in forward order (agrees with the kernel trace)
        achieved = conv_flops / (conv_total_ms * 1e-3) / 1e12
        eff_peak = conv_flops / (ideal_ms * 1e-3) / 1e12               # FLOP-weighted harmonic peak of the layers' arithmetics
        n_bf = sum(1 for pr in precs if pr == 1)
        n_h2 = sum(1 for pr in precs if pr == 2)
        traffic, traffic_note = (None, "skipped") if (n_gpus > 1 or args.no_pmc) else pmc_traffic(args, tiles, len(conv_ms))
        step_flops = conv_flops                                           # conv GEMM FLOPs of one step (B images)
        line = {
            "metric": "images/sec Faster R-CNN ResNet-50 @800x1333" if args.backbone == "resnet50"
                      else f"images/sec Faster R-CNN {args.backbone} @{args.height}x{args.width}",
            "value": round(n_gpus * B / (head["ms_per_step"] * 1e-3), 3), "unit": "images/s", "n_gpus": n_gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(head["ms_per_step"], 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if n_bf + n_h2 == 0 else "f32 storage/accumulate; conv products as bf16x3 (3 exact bf16 pieces per operand, "
                                              f"6 bf16 MFMAs per product) in {n_bf} of {len(precs)} conv layers, "
                                              + (f"as fp16x2 (2 fp16 pieces of 2^e x per operand, 3 fp16 MFMAs per product; e follows the tensor's abs-max per "
                                                 f"forward: range words) in {n_h2}, " if n_h2 else "")
                                              + "f32 MFMA in the rest",
            "data": "synthetic" + (" (seeded random-init weights; BatchNorm running statistics pre-computed from two seeded images, "
                                   "configs/synthetic_bn_*.npz)" if args.backbone.startswith("hardnet") else ""),
            "config": {"workload": f"Full Faster R-CNN {args.backbone} inference forward, batch={B} per GPU, "
                                   f"3x{args.height}x{args.width}, {args.num_classes}+1 classes, 3000->300 proposals"
                                   + (" (BASELINE configs[4]: data-parallel, 8 images per rank)" if world > 1 and B == 8 else ""),
                       "global_batch": n_gpus * B, "parallelism": f"dp{n_gpus}",
                       "hip_graph": not args.no_graph, "autotuned_tiles": not args.no_autotune, "steps_in_flight": n_fly,
                       "timed_step": ("result() of the request that last used the slot (event wait + the slot's range-word read), then submit()"
                                      if n_fly > 1 else "submit() only: forwards queued back to back on one stream, results collected behind the region"),
                       # the two schedules side by side (value / ms_per_step above = the in-flight one)
                       "images_per_s_in_flight": round(n_gpus * B / (head["ms_per_step"] * 1e-3), 3),
                       "images_per_s_serial": round(n_gpus * B / (serial["ms_per_step"] * 1e-3), 3),
                       "ms_per_step_serial": round(serial["ms_per_step"], 4),
                       "tiles_tuned_on": "rank 0, broadcast" if world > 1 else "this rank",
                       "collective": None if n_gpus == 1 else (f"tsod_allgather_f32 [{n_gpus * B},300,6] f32 (RCCL)" if args.collective == "tsod" and args.dist_backend == "nccl"
                                                              else f"all_gather_into_tensor [{n_gpus * B},300,6] f32 ({args.dist_backend})")},
            "repeats": {"n": head["n"], "steps_each": head["steps"], "ms_per_step_median": round(head["ms_per_step"], 4),
                        "ms_per_step_min": round(head["min"], 4), "ms_per_step_max": round(head["max"], 4)},
            "serial": {"images_per_s": round(n_gpus * B / (serial["ms_per_step"] * 1e-3), 3),
                       "ms_per_step": round(serial["ms_per_step"], 4), "ms_per_step_min": round(serial["min"], 4),
                       "ms_per_step_max": round(serial["max"], 4), "repeats": serial["n"],
                       "note": "one forward at a time on one stream (--in-flight 1 semantics) = latency of a step"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 3), "peak": round(eff_peak, 1), "unit": "TFLOP/s",
                         "frac": round(achieved / eff_peak, 4),
                         "peak_note": f"f32-equivalent: {n_bf} layers bf16x3 (bf16 dense MFMA {BF16_MFMA_PEAK_TFLOPS} / 6 products = "
                                      f"{BF16X3_PEAK_TFLOPS:.1f}), "
                                      + (f"{n_h2} layers fp16x2 (/ 3 products = {FP16X2_PEAK_TFLOPS:.1f}), " if n_h2 else "")
                                      + f"{len(precs) - n_bf - n_h2} layers f32 MFMA ({F32_MFMA_PEAK_TFLOPS}); FLOP-weighted "
                                      "harmonic mean, i.e. frac = sum of ideal matrix-pipe times / measured time",
                         "executed_bf16_mfma_tflops": round(6.0 * flops_bf / (conv_total_ms * 1e-3) / 1e12, 1),
                         "traffic": None if traffic is None else round(traffic),
                         "traffic_unit": "HBM bytes per conv launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 / launches"
                                         + ("" if traffic_note is None else f" [{traffic_note}]"),
                         "algorithmic_bytes_per_launch": round(algo_bytes / len(conv_ms)),
                         "traffic_over_algorithmic": None if traffic is None else round(traffic * len(conv_ms) / algo_bytes, 3),
                         "kernel": f"conv_igemm_kernel / conv_dma_kernel (implicit GEMM; per layer f32 MFMA, bf16x3 or fp16x2 MFMA, register-staged or "
                                   f"fed by LDS-DMA)" + (f" + bottleneck_kernel ({n_fused} bottlenecks of layer1 as one launch each)" if n_fused else "")
                                   + (" + stem_kernel (conv1 + bn1 + PReLU + max pool from the NCHW images as one launch)" if has_stem else "")
                                   + f", {len(conv_ms)} launches per forward",
                         "schedule": "serial", "flops_per_forward": conv_flops,
                         "kernel_ms_per_forward": round(conv_total_ms, 4),
                         "kernel_ms_measured_as": "HIP events around one pass over the conv launches in forward order (cache state of a "
                                                  "forward, launch boundaries included); `achieved` / `frac` use it",
                         "kernel_ms_per_forward_isolated": round(sum(conv_ms), 4),
                         "isolated_note": "sum over layers of 5 back-to-back repeats of each launch on hot operands (per-layer figures of "
                                          "--dump-layers / --verbose): the optimistic bound",
                         "serial_ms_per_step": round(serial["ms_per_step"], 4),
                         "share_of_serial_step": round(conv_total_ms / serial["ms_per_step"], 4)},
            "throughput_mode": {"steps_in_flight": n_fly, "conv_tflops_per_step_time": round(step_flops / (head["ms_per_step"] * 1e-3) / 1e12, 3),
                                "frac_of_peak": round(step_flops / (head["ms_per_step"] * 1e-3) / 1e12 / eff_peak, 4),
                                "note": "conv FLOPs of a step / ms_per_step of the headline schedule: a whole-step bound "
                                        "(non-GEMM kernels included in the time), not a per-kernel measurement"},
        }
        line["tuning_seconds"] = {"per_rank": all_tuning, "note": "autotune + head tuning wall time before the first timed step"
                                  + (" (rank 0 tunes, the others wait for its broadcast)" if world > 1 else "")}
        if f32_leg is not None:
            line["roofline_f32_mfma"] = f32_leg
        if bf_leg is not None:
            line["roofline_bf16x3"] = bf_leg
        if check is not None:
            line["check"] = check
        parity = None
        if not args.no_cpu_baseline:
            # N = 1: the timed oracle forwards double as the reference of `parity`; N > 1: one untimed oracle forward of rank 0's shard
            ref_outs, cpu = cpu_baseline(sd, args.backbone, x_cpu, args.cpu_reps if n_gpus == 1 else 0)
            if n_gpus == 1:
                line["cpu_baseline"] = cpu
            parity = parity_of(timed_outs, ref_outs, getattr(cpu_baseline, "exact_outs", None), getattr(cpu_baseline, "cutoff", None),
                               getattr(cpu_baseline, "exact_cutoff", None))
        line["parity"] = parity
        print(json.dumps(line), flush=True)
        if parity is not None and not parity["ok"]:
            exit_code = 4
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if exit_code:
        sys.exit(exit_code)


if __name__ == "__main__":
    main()
