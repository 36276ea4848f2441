"""Frame-parallel inference across the GPUs of one node (one process per GPU, RCCL over xGMI).

The reference evaluates on a single device (paddle3d/apis/trainer.py:47-51,402) and has no collective
on the inference path; frames are independent, so the path shards with NO data-path collective: rank r
takes frames r, r + world, ... (or its own slice of a batch).  The only exchange is the result hand-off
the north star asks for: ONE all-gather per batch of frames of a fixed-shape record
    float32 [frames, max_per_img, 11] = 9 box values (7 without velocity, zero padded), score, label
plus int32 [frames] row counts.  At ~22 KB per frame the collective is latency bound, so it is issued
once per batch, never per frame or per task.
`backend="nccl"` is RCCL on ROCm; tests run the same code over gloo on CPU with world_size 2.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

RECORD_WIDTH = 11


def init_from_env(backend: str | None = None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns
    (rank, world, local_rank); a plain single-process run returns (0, 1, 0) without a process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_frames(num_frames: int, rank: int, world: int):
    """Round-robin frame ownership: frame i -> rank i % world."""
    return list(range(rank, num_frames, world))


def pack_records(boxes: torch.Tensor, scores: torch.Tensor, labels: torch.Tensor, counts: torch.Tensor,
                 max_per_img: int = 500) -> torch.Tensor:
    """[F, R, dims] / [F, R] / [F, R] (+ counts [F]) -> float32 [F, max_per_img, 11], rows >= count zeroed."""
    f, r, dims = boxes.shape
    rec = torch.zeros((f, max_per_img, RECORD_WIDTH), dtype=torch.float32, device=boxes.device)
    k = min(r, max_per_img)
    rec[:, :k, :dims] = boxes[:, :k]
    rec[:, :k, 9] = scores[:, :k]
    rec[:, :k, 10] = labels[:, :k].to(torch.float32)
    valid = torch.arange(max_per_img, device=boxes.device).unsqueeze(0) < counts.clamp(max=max_per_img).unsqueeze(1)
    # where(), not a 0/1 multiply: rows >= count may hold anything (NaN * 0 = NaN)
    return torch.where(valid.unsqueeze(-1), rec, torch.zeros((), dtype=rec.dtype, device=rec.device))


def gather_detections(records: torch.Tensor, counts: torch.Tensor):
    """All-gather the per-rank records [F, M, 11] and counts [F] -> ([world*F, M, 11], [world*F]) in rank order.
    Every rank ends up with every frame's boxes (what a metric aggregation needs).  Two collectives of a
    fixed, equal size per rank: RCCL all_gather_into_tensor."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return records, counts
    world = dist.get_world_size()
    out_r = torch.empty((world * records.shape[0],) + tuple(records.shape[1:]), dtype=records.dtype,
                        device=records.device)
    out_c = torch.empty((world * counts.shape[0],), dtype=counts.dtype, device=counts.device)
    dist.all_gather_into_tensor(out_r, records.contiguous())
    dist.all_gather_into_tensor(out_c, counts.contiguous())
    return out_r, out_c


def unpack_records(records: torch.Tensor, counts: torch.Tensor, with_velocity: bool = True):
    """Inverse of pack_records for host-side consumers: list of dict(box3d_lidar, scores, label_preds)."""
    dims = 9 if with_velocity else 7
    out = []
    for rec, k in zip(records.cpu(), counts.cpu().tolist()):
        out.append(dict(box3d_lidar=rec[:k, :dims], scores=rec[:k, 9], label_preds=rec[:k, 10].to(torch.int64)))
    return out
