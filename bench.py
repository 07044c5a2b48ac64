#!/usr/bin/env python3
"""bench.py -- images/sec of the full Faster R-CNN ResNet-50 inference forward on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one detector forward (NCHW->NHWC, 53 conv GEMMs, max pool, RPN convs + decode + top-k +
NMS + pad, fused RoI pool + mean, 2 linears, detection records) over one batch of synthetic
3x800x1333 images that is already resident in HBM, plus - for N > 1 - the RCCL all-gather of the
[B,300,6] detection records.  Each step is one HIP-graph launch; consecutive steps are issued round-robin
on --in-flight HIP streams (default 4, each with its own graph + buffers), i.e. a batch-1 server with
several requests in flight: the tail of one forward overlaps the next one's kernels.  --in-flight 1 gives the
strictly serial number; the single-stream latency of one forward is reported as latency_ms_single_stream.  Workload at N=1 = BASELINE.json configs[1] (batch 1).  Weak scaling:
every rank processes its own batch; value = images of all ranks / max-over-ranks time.

The JSON line also carries
  roofline     : f32-MFMA roofline of the dominant kernel family (conv_igemm_kernel, all launches of one
                 forward): algorithmic FLOPs / HIP-event time of those launches, vs 157.3 TFLOP/s.
                 roofline.traffic = HBM bytes per conv launch from the FETCH_SIZE / WRITE_SIZE counters, collected
                 by two short `rocprofv3 --pmc` child runs of this script (rank 0, N=1; --no-pmc skips them).
  cpu_baseline : the CPU oracle (torch CPU ops + C nms/roi_pool restatement of the reference's path)
                 timed on this box's host cores on the same workload (rank 0, N=1 only).
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1, help="images per GPU per step (configs[1] = 1, configs[2] = 16)")
    ap.add_argument("--backbone", default="resnet50")
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=1333)
    ap.add_argument("--num-classes", type=int, default=80)
    ap.add_argument("--no-autotune", action="store_true")
    ap.add_argument("--autotune-concurrent", type=int, default=None, help="time autotune candidates as this many copies in "
                    "flight on separate streams (default: 2 when --in-flight > 1, else 1)")
    ap.add_argument("--autotune-splits", default=None, help="comma list restricting the K-slice candidates of the autotuner")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--in-flight", type=int, default=4, help="steps in flight: consecutive steps are issued round-robin on this "
                    "many HIP streams, each with its own graph and buffers (request-level pipelining of a batch-1 server)")
    ap.add_argument("--tiles-file", default=None, help="JSON cache of autotuned (tile, split) choices: loaded if present, "
                                                       "else written after autotuning (keeps profiler runs free of tuning launches)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc child runs behind roofline.traffic")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-reps", type=int, default=8)
    ap.add_argument("--verbose", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI; the real path) | gloo (rehearsal of "
                    "the N>1 control flow on a box with fewer GPUs than ranks: ranks share devices, gather goes through host)")
    return ap.parse_args()


def conv_event_times(plan, reps=5):
    """HIP-event duration of every conv_igemm launch of the plan (ms), on the launch stream."""
    from two_stage_object_detection_amd._ffi import lib, stream_ptr
    L = lib()
    out = []
    s = stream_ptr()
    for st in plan.conv_steps:
        L.tsod_conv2d_f32(*st.args, s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            L.tsod_conv2d_f32(*st.args, s)
        e1.record()
        e1.synchronize()
        out.append(e0.elapsed_time(e1) / reps)
    return out


def cpu_baseline(sd, backbone, x_cpu, reps):
    import oracle
    # the GPU box exposes all host cores but grants this job a 16-core share (worker pools must be sized to it)
    torch.set_num_threads(min(os.cpu_count() or 1, int(os.environ.get("TSOD_CPU_THREADS", "16"))))
    with torch.inference_mode():
        oracle.detector_forward(sd, x_cpu, backbone=backbone)           # warm-up
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            oracle.detector_forward(sd, x_cpu, backbone=backbone)
            ts.append(time.perf_counter() - t0)
    med = statistics.median(ts)
    return {"value": x_cpu.shape[0] / med, "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} timed forwards (median) of the same workload: batch {x_cpu.shape[0]} x 3x{x_cpu.shape[2]}x"
                      f"{x_cpu.shape[3]}, {backbone} detector, torch {torch.__version__} CPU f32 + oracle/box_ops.c",
            "seconds_per_forward": med}


def conv_algorithmic_bytes(plan):
    """Input + output + weights (+ residual) of every conv launch, each counted once (f32)."""
    tot = 0
    for st in plan.conv_steps:
        d = st.desc
        cin = sum(d.seg_len[i] for i in range(d.n_seg))
        tot += 4 * (d.N * d.H * d.W * cin + d.N * d.OH * d.OW * d.Cout + d.Cout * d.KH * d.KW * cin)
        if d.res_pitch > 0:
            tot += 4 * d.N * d.OH * d.OW * d.Cout
    return tot


def pmc_child(args, dev):
    """Body of the `rocprofv3 --pmc` child: build the plan with the parent's tile choices, then issue the conv
    launches of one forward twice, eagerly, so that every dispatch carries its own counter sample."""
    from two_stage_object_detection_amd._ffi import lib, stream_ptr
    from two_stage_object_detection_amd.testing import synthetic_detector
    model, _ = synthetic_detector(args.backbone, num_classes=args.num_classes, seed=0)
    model = model.to(dev).eval()
    x = torch.rand(args.batch, 3, args.height, args.width, generator=torch.Generator().manual_seed(1234)).to(dev)
    with torch.inference_mode():
        plan = model.extractor._plan_for(x)                 # packs weights, launches no conv
        if args.tiles_file and os.path.exists(args.tiles_file):
            plan.import_tiles(json.load(open(args.tiles_file)))
        torch.cuda.synchronize()
        L, s = lib(), stream_ptr()
        for _ in range(2):                                  # pass 1 warms caches / code objects, pass 2 is the sample
            for st in plan.conv_steps:
                L.tsod_conv2d_f32(*st.args, s)
        torch.cuda.synchronize()


def pmc_traffic(args, plan, n_launches):
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
        tiles = os.path.join(work, "tiles.json")
        json.dump(plan.export_tiles(), open(tiles, "w"))
        env = dict(os.environ, TMPDIR="/tmp")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
            env.pop(k, None)
        sums = {}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(work, counter)
            cmd = [prof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", out, "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "--tiles-file", tiles,
                   "--backbone", args.backbone, "--batch", str(args.batch), "--height", str(args.height),
                   "--width", str(args.width), "--num-classes", str(args.num_classes)]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
            files = glob.glob(os.path.join(out, "*", "*counter_collection.csv"))
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode})"
            rows = [(int(q["Dispatch_Id"]), float(q["Counter_Value"])) for q in csv.DictReader(open(files[0]))
                    if q["Counter_Name"] == counter and ("conv_igemm_kernel" in q["Kernel_Name"]
                                                         or "conv_reduce_kernel" in q["Kernel_Name"])]
            rows.sort()
            if not rows or len(rows) % 2:
                return None, f"unexpected dispatch count {len(rows)} in the {counter} pass"
            sums[counter] = sum(v for _, v in rows[len(rows) // 2:])          # the second of the two conv passes
        total = (2.0 * sums["FETCH_SIZE"] + sums["WRITE_SIZE"]) * 1024.0
        return total / n_launches, None
    except Exception as e:                                                     # noqa: BLE001 - never fail the bench on this leg
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(work, ignore_errors=True)


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world != args.gpus and rank == 0 and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE", file=sys.stderr)
    n_gpus = world if world > 1 else 1
    assert torch.cuda.is_available(), "bench.py needs a GPU (the package has no CPU path)"
    dev_index = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.dist_backend, rank=rank, world_size=world)

    if args.pmc_child:
        pmc_child(args, dev)
        return

    from two_stage_object_detection_amd import hip_ops
    from two_stage_object_detection_amd.dist import all_gather_detections
    from two_stage_object_detection_amd.testing import synthetic_detector

    model, sd = synthetic_detector(args.backbone, num_classes=args.num_classes, seed=0)
    model = model.to(dev).eval()
    B = args.batch
    x_cpu = torch.rand(B, 3, args.height, args.width, generator=torch.Generator().manual_seed(1234 + rank))
    x = x_cpu.to(dev)

    with torch.inference_mode():
        model(x)                                                       # builds the plan
        torch.cuda.synchronize()
        plan = model.extractor._plan_for(x)
        if args.tiles_file and os.path.exists(args.tiles_file):
            plan.import_tiles(json.load(open(args.tiles_file)))
        elif not args.no_autotune:
            plan.autotune(verbose=args.verbose and rank == 0,
                          splits=[int(v) for v in args.autotune_splits.split(",")] if args.autotune_splits else None,
                          concurrent=args.autotune_concurrent if args.autotune_concurrent
                          else (2 if (args.in_flight > 1 and not args.no_graph) else 1))
            if args.tiles_file and rank == 0:
                json.dump(plan.export_tiles(), open(args.tiles_file, "w"))
        conv_ms = conv_event_times(plan)
        conv_flops = sum(st.flops for st in plan.conv_steps)
        R_post = model.rpn.proposal_layer.counts()[1]
        gathered = [torch.empty((world * B, R_post, 6), dtype=torch.float32, device=dev if args.dist_backend == "nccl" else "cpu")
                    for _ in range(max(1, args.in_flight))] if world > 1 else None
        if args.no_graph:
            def step():
                outs = model(x)
                det = hip_ops.detections(outs[0], outs[1], outs[2])
                if world > 1:
                    all_gather_detections(det if args.dist_backend == "nccl" else det.cpu(), out=gathered[0])
                return det
        else:
            from two_stage_object_detection_amd.serving import InFlightDetector
            n_fly = max(1, args.in_flight)
            server = InFlightDetector(model, x, depth=n_fly, tiles=plan.export_tiles())   # one graph + buffers per slot
            runners = server._run

            def step():
                if world > 1:                           # the gather of step i is ordered behind step i on ITS stream
                    slot = server._next % n_fly
                    return server.submit(after=lambda outs: all_gather_detections(
                        outs[4] if args.dist_backend == "nccl" else outs[4].cpu(), out=gathered[slot]))
                return server.submit()
        def full_step():
            return step()

        # single-stream latency of one forward (informational; the timed region below is the K-step throughput run)
        latency_ms = None
        if not args.no_graph:
            for _ in range(3):
                runners[0]()
            torch.cuda.synchronize()
            t_l = time.perf_counter()
            for _ in range(20):
                runners[0]()
            torch.cuda.synchronize()
            latency_ms = (time.perf_counter() - t_l) / 20 * 1e3
        for _ in range(max(args.warmup, 1)):
            full_step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            full_step()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        model.raise_if_error()

    t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = n_gpus * B * args.steps / elapsed
        conv_total_ms = sum(conv_ms)
        achieved = conv_flops / (conv_total_ms * 1e-3) / 1e12
        traffic, traffic_note = (None, "skipped") if (n_gpus > 1 or args.no_pmc) else pmc_traffic(args, plan, len(conv_ms))
        line = {
            "metric": "images/sec Faster R-CNN ResNet-50 @800x1333" if args.backbone == "resnet50"
                      else f"images/sec Faster R-CNN {args.backbone} @{args.height}x{args.width}",
            "value": round(value, 3), "unit": "images/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"Full Faster R-CNN {args.backbone} inference forward, batch={B} per GPU, "
                                   f"3x{args.height}x{args.width}, {args.num_classes}+1 classes, 3000->300 proposals",
                       "global_batch": n_gpus * B, "parallelism": f"dp{n_gpus}",
                       "hip_graph": not args.no_graph, "autotuned_tiles": not args.no_autotune,
                       "steps_in_flight": 1 if args.no_graph else max(1, args.in_flight),
                       "collective": None if n_gpus == 1 else f"all_gather_into_tensor [{n_gpus * B},300,6] f32 ({args.dist_backend})"},
            "roofline": {"bound": "mfma", "achieved": round(achieved, 3), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(achieved / F32_MFMA_PEAK_TFLOPS, 4),
                         "traffic": None if traffic is None else round(traffic),
                         "traffic_unit": "HBM bytes per conv launch = (2*FETCH_SIZE + WRITE_SIZE) * 1024 / launches, incl. the "
                                         "K-slice reduce kernels" + ("" if traffic_note is None else f" [{traffic_note}]"),
                         "algorithmic_bytes_per_launch": round(conv_algorithmic_bytes(plan) / len(conv_ms)),
                         "kernel": f"conv_igemm_kernel (f32 MFMA implicit GEMM), {len(conv_ms)} launches per forward",
                         "flops_per_forward": conv_flops, "kernel_ms_per_forward": round(conv_total_ms, 4),
                         "share_of_single_stream_forward": None if latency_ms is None else round(conv_total_ms / latency_ms, 4)},
            "latency_ms_single_stream": None if latency_ms is None else round(latency_ms, 4),
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sd, args.backbone, x_cpu, args.cpu_reps)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
