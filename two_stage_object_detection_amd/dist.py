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


def global_roi_indices(roi_indices_local: torch.Tensor, rank: int, b_local: int) -> torch.Tensor:
    """Local image indices -> indices into the global batch."""
    return roi_indices_local + rank * b_local
