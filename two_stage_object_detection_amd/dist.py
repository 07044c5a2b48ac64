"""Data-parallel sharding of images over the GPUs of one node and the detection all-gather.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on ROCm).  Every image's
forward is independent (eval-mode BN, per-image proposals), so there is NO collective on the data
path; the only exchange is one all-gather of fixed-size detection records per step:
[B_local, R, 6] f32 = (x1,y1,x2,y2, max logit, class) -> 7.2 KB per image.  R is fixed by the
reference's padding rule (quirk Q4), so counts are equal on all ranks: a plain all-gather, no
all-gatherv.  At these sizes the exchange is latency-bound on every xGMI link; it is issued on the
compute stream right behind the detections kernel.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def shard_range(global_batch: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous split: rank r owns images [r*B/W, (r+1)*B/W).  global_batch must divide evenly."""
    if global_batch % world != 0:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    per = global_batch // world
    return rank * per, (rank + 1) * per


def all_gather_detections(det_local: torch.Tensor, group=None, out: torch.Tensor | None = None) -> torch.Tensor:
    """[B_local, R, 6] on every rank -> [world*B_local, R, 6] in rank order (= global image order
    under shard_range).  Single-process runs return the input unchanged."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return det_local
    world = dist.get_world_size(group)
    det_local = det_local.contiguous()
    if out is None:
        out = torch.empty((world * det_local.shape[0],) + tuple(det_local.shape[1:]), dtype=det_local.dtype,
                          device=det_local.device)
    dist.all_gather_into_tensor(out, det_local, group=group)
    return out


class TsodCommunicator:
    """An RCCL communicator owned through the C-ABI (tsod_comm_* / tsod_allgather_f32, include/tsod.h): what a host without
    torch.distributed would use, and a second, independent way to run the job's one collective on the compute stream.
    The 128-byte unique id (+ one status byte) is made on rank 0 and handed to the others through `exchange` (a callable
    bytes -> bytes that returns rank 0's bytes on every rank; default: a torch.distributed broadcast on the already initialised group, else
    identity for a single process)."""

    def __init__(self, rank: int = 0, world: int = 1, exchange=None):
        import ctypes
        from . import _ffi
        self._ffi, self.rank, self.world = _ffi, int(rank), int(world)
        ident = ctypes.create_string_buffer(128)
        # rank 0's status travels WITH the id (one extra byte): when it could not make one (no loadable librccl ->
        # TSOD_ERR_UNSUPPORTED) every rank raises, instead of rank 0 raising alone and the others blocking in the broadcast
        rc0 = _ffi.lib().tsod_comm_unique_id(ident) if self.rank == 0 else 0
        raw = bytes(ident.raw) + bytes([min(255, -int(rc0))])
        if exchange is not None:
            raw = exchange(raw)
        elif self.world > 1:
            t = torch.tensor(list(raw), dtype=torch.uint8)
            if dist.get_backend() == "nccl":
                t = t.cuda()
            dist.broadcast(t, src=0)
            raw = bytes(t.cpu().tolist())
        if len(raw) > 128 and raw[128] != 0:
            _ffi.check(-int(raw[128]), "tsod_comm_unique_id on rank 0")
        raw = raw[:128]
        self._comm = ctypes.c_void_p()
        _ffi.check(_ffi.lib().tsod_comm_init_rank(ctypes.byref(self._comm), self.world, raw, self.rank))

    def all_gather(self, det_local: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """[B_local, ...] f32 on every rank -> [world * B_local, ...] in rank order, on the current stream."""
        det_local = det_local.contiguous()
        if det_local.dtype != torch.float32 or not det_local.is_cuda:
            raise ValueError("tsod_allgather_f32 moves f32 device tensors")
        if out is None:
            out = torch.empty((self.world * det_local.shape[0],) + tuple(det_local.shape[1:]), dtype=torch.float32,
                              device=det_local.device)
        f = self._ffi
        f.check(f.lib().tsod_allgather_f32(self._comm, f.ptr(det_local), f.ptr(out), det_local.numel(), f.stream_ptr()))
        return out

    def close(self):
        if self._comm:
            self._ffi.check(self._ffi.lib().tsod_comm_destroy(self._comm))
            self._comm = None

    def __del__(self):
        try:
            self.close()
        except Exception:       # noqa: BLE001 - interpreter shutdown
            pass


def global_roi_indices(roi_indices_local: torch.Tensor, rank: int, b_local: int) -> torch.Tensor:
    """Local image indices -> indices into the global batch."""
    return roi_indices_local + rank * b_local
