"""The N>1 path on CPU: world_size-2 gloo run of the sharding + detection all-gather used by bench.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from two_stage_object_detection_amd.dist import all_gather_detections, global_roi_indices, shard_range


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        global_batch, R = 6, 300
        lo, hi = shard_range(global_batch, rank, world)
        # the "detections" of image g are filled with g so the gathered order is checkable
        det_local = torch.stack([torch.full((R, 6), float(g)) for g in range(lo, hi)])
        out = all_gather_detections(det_local)
        ok = out.shape == (global_batch, R, 6) and all(bool((out[g] == g).all()) for g in range(global_batch))
        idx = global_roi_indices(torch.arange(hi - lo, dtype=torch.int32), rank, hi - lo)
        ok = ok and idx.tolist() == list(range(lo, hi))
        # pre-allocated output buffer variant (what bench.py uses)
        buf = torch.empty(global_batch, R, 6)
        ok = ok and torch.equal(all_gather_detections(det_local, out=buf), out)
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_allgather_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, True), (1, True)]


def test_shard_range():
    assert [shard_range(64, r, 8) for r in (0, 7)] == [(0, 8), (56, 64)]
    with pytest.raises(ValueError):
        shard_range(10, 0, 4)


def test_single_process_is_identity():
    d = torch.randn(2, 300, 6)
    assert all_gather_detections(d) is d


def test_bench_launcher_starts_the_ranks_itself():
    """`python bench.py --gpus 2` (the shape of the driver's command, no torchrun, no WORLD_SIZE): bench.py must start the
    two ranks itself.  --rehearse-cpu swaps the forwards for fabricated records so that the launcher, the rendezvous
    on 127.0.0.1, the timed-region protocol (barrier / max over ranks / repeats), the all-gather into the pre-allocated
    buffer and the --check comparison all run here on CPU with gloo."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-cpu", "--check",
                        "--steps", "3", "--warmup", "1", "--repeats", "2", "--batch", "3"],
                       cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout                                   # rank 0 prints ONE line
    line = lines[0]
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 6 and line["config"]["parallelism"] == "dp2"
    assert line["rehearsal"] is True and line["value"] is None          # never mistaken for a measurement
    assert line["check"]["ok"] is True and line["check"]["images"] == 6 and line["check"]["records_unmatched"] == 0
    assert line["check"]["max_abs_box_on_matched"] == 0.0 and line["check"]["images_off"] == []
    assert line["steps"] == 3 and line["scaling"] == "weak"
    # start-up protocol of the N>1 run: rank 0's tile table + head choices reached every rank as JSON
    assert line["tiles_broadcast_ok_per_rank"] == [True, True] and line["check"]["matching"] == "position-wise"
    assert line["check"]["bit_exact"] is True


def test_bench_refuses_more_nccl_ranks_than_gpus(monkeypatch):
    """Without a GPU per rank RCCL cannot run: bench.py must say so and exit non-zero instead of printing n_gpus: 1."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "needs a GPU" in (r.stdout + r.stderr) or "visible GPU" in (r.stdout + r.stderr)


def test_tsod_communicator_every_rank_raises_when_rank0_could_not_make_the_id():
    """dist.TsodCommunicator ships rank 0's status WITH the unique id: a rank that receives a failed status raises (instead of
    entering ncclCommInitRank and blocking for a rank 0 that already raised)."""
    from two_stage_object_detection_amd._ffi import TsodError
    from two_stage_object_detection_amd.dist import TsodCommunicator
    seen = {}

    def exchange(raw):
        seen["len"] = len(raw)
        return bytes(128) + bytes([2])                 # what rank 0 would broadcast after TSOD_ERR_UNSUPPORTED (-2)
    with pytest.raises(TsodError, match="rank 0"):
        TsodCommunicator(rank=1, world=2, exchange=exchange)
    assert seen["len"] == 129
