"""Multi-GPU layer: chromosomes are independent shards; one process per GPU.

Only three things cross ranks, all tiny and latency-bound (single-shot collectives, no ring tuning):
  1. the AS histogram of each BAM (phaser.py:545-553 takes the quantile over ALL chromosomes)  -> all_reduce(SUM)
  2. the two noise counters (phaser.py:610-632 is global over variants)                        -> all_reduce(SUM)
  3. per-chromosome output fragments, gathered to rank 0 which assembles the files in the reference's
     global order (engine.merge_fragments)                                                     -> gather_object
Backend "nccl" (= RCCL over xGMI) on GPUs; the same code runs on "gloo" for the CPU tests.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.distributed as dist


def world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)


def assign_chromosomes(weights: Dict[str, float], world_size: int) -> Dict[str, int]:
    """Longest-processing-time assignment of chromosomes to ranks by record count (deterministic)."""
    load = [0.0] * world_size
    owner: Dict[str, int] = {}
    for chrom, w in sorted(weights.items(), key=lambda kv: (-kv[1], kv[0])):
        r = min(range(world_size), key=lambda i: (load[i], i))
        owner[chrom] = r
        load[r] += w
    return owner


def allreduce_sum_(t: torch.Tensor) -> torch.Tensor:
    """In-place SUM over ranks; tensors must live where the backend wants them (cuda for nccl, cpu for gloo)."""
    r, w = world()
    if w > 1:
        backend = dist.get_backend()
        if backend == "nccl" and t.device.type != "cuda":
            tmp = t.cuda()
            dist.all_reduce(tmp)
            t.copy_(tmp.cpu())
        elif backend == "gloo" and t.device.type != "cpu":
            tmp = t.cpu()
            dist.all_reduce(tmp)
            t.copy_(tmp.to(t.device))
        else:
            dist.all_reduce(t)
    return t


def allreduce_counts(match: int, mism: int):
    r, w = world()
    if w == 1:
        return match, mism
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([match, mism], dtype=torch.int64, device=dev)
    dist.all_reduce(t)
    return int(t[0]), int(t[1])


def gather_fragments(local: Dict[str, dict]) -> Optional[Dict[str, dict]]:
    """-> on rank 0 the union of all ranks' {chrom: fragment}; None elsewhere."""
    r, w = world()
    if w == 1:
        return dict(local)
    bucket: List[Optional[dict]] = [None] * w if r == 0 else None
    # row text lives in native buffers (memoryviews): materialise it for pickling
    local = {c: {k: (bytes(v) if isinstance(v, memoryview) else v) for k, v in f.items()} for c, f in local.items()}
    dist.gather_object(local, bucket, dst=0)
    if r != 0:
        return None
    merged: Dict[str, dict] = {}
    for part in bucket:
        merged.update(part)
    return merged
