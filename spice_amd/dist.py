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
    """(rank, world, local_rank) from torchrun's environment; no-op for a single process.
    SPICE_DIST_BACKEND overrides the backend (e.g. "gloo" to rehearse several ranks on one GPU:
    device tensors are then staged through the host in the helpers below); local_rank is folded
    onto the visible devices so that such a rehearsal maps every rank to an existing GPU."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("SPICE_DIST_BACKEND", backend)
    n_dev = torch.cuda.device_count()
    if n_dev > 0 and local >= n_dev:
        if backend == "nccl" or backend is None:
            raise RuntimeError(f"LOCAL_RANK={local} but only {n_dev} GPU(s) visible (RCCL needs one GPU per rank)")
        local %= n_dev
    if world > 1:
        # one MIOpen user database per rank: ranks of one node would otherwise serialise on its file lock
        os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(
            os.environ.get("TMPDIR", "/tmp"), f"spice_miopen_{os.getuid()}_r{local}"))
        os.makedirs(os.environ["MIOPEN_USER_DB_PATH"], exist_ok=True)
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


def _all_reduce_(t: torch.Tensor, op) -> torch.Tensor:
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if t.is_cuda and dist.get_backend() == "gloo":
            host = t.cpu()
            dist.all_reduce(host, op=op)
            t.copy_(host)
        else:
            dist.all_reduce(t, op=op)
    return t


def all_reduce_sum_(t: torch.Tensor) -> torch.Tensor:
    """In-place sum over ranks (identity for one process)."""
    return _all_reduce_(t, dist.ReduceOp.SUM)


def all_reduce_max_(t: torch.Tensor) -> torch.Tensor:
    """In-place maximum over ranks (identity for one process)."""
    return _all_reduce_(t, dist.ReduceOp.MAX)


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
