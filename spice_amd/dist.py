"""Multi-GPU plumbing: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI on a GPU
node, "gloo" in CPU tests).  The hot path shards without data-path collectives: ranks take
disjoint candidates / calibration batches / evaluation batches and exchange only small flat
buffers at the end (Taylor scores: sum of out-channels floats; confusion tables: C*C int64 each).
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str = None) -> Tuple[int, int, int]:
    """(rank, world, local_rank) from torchrun's environment; no-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    return rank, world, local


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def partition_by_cost(costs: Sequence[float], n_parts: int) -> List[List[int]]:
    """Longest-processing-time-first assignment of items to `n_parts` bins; deterministic
    (ties by index), every item in exactly one bin."""
    order = sorted(range(len(costs)), key=lambda i: (-costs[i], i))
    loads = [0.0] * n_parts
    parts: List[List[int]] = [[] for _ in range(n_parts)]
    for i in order:
        k = min(range(n_parts), key=lambda b: (loads[b], b))
        parts[k].append(i)
        loads[k] += costs[i]
    for p in parts:
        p.sort()
    return parts


def all_reduce_sum_(t: torch.Tensor) -> torch.Tensor:
    """In-place sum over ranks (identity for one process)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def barrier() -> None:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def sharded_tables(n_items: int, indices: Sequence[int], costs: Sequence[float], evaluate, tables: Sequence[torch.Tensor]) -> None:
    """Every rank runs `evaluate(i)` for its cost-balanced share of `indices` (each call fills row
    i of the zero-initialised `tables`), then the tables are summed over ranks: afterwards every
    rank holds every row.  No communication happens before the final all-reduce."""
    rank, n_ranks = world()
    parts = partition_by_cost(list(costs), n_ranks)
    for k in parts[rank]:
        evaluate(indices[k])
    for t in tables:
        all_reduce_sum_(t)
