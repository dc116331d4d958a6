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
    """(rank, world, local_rank) from torchrun's environment; no-op for a single process
    (unless SPICE_DIST_FORCE_INIT=1).
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
    if n_dev > 0:
        # whatever the backend: launches address the current device's handles and per-device state
        torch.cuda.set_device(local)
    if world > 1:
        # one MIOpen user database per rank: ranks of one node would otherwise serialise on its file lock
        os.environ.setdefault("MIOPEN_USER_DB_PATH", os.path.join(
            os.environ.get("TMPDIR", "/tmp"), f"spice_miopen_{os.getuid()}_r{local}"))
        os.makedirs(os.environ["MIOPEN_USER_DB_PATH"], exist_ok=True)
    # SPICE_DIST_FORCE_INIT=1 creates the process group for a single rank too (RCCL on one GPU: how the
    # collective path is exercised where only one GPU is at hand)
    force = os.environ.get("SPICE_DIST_FORCE_INIT", "0") == "1"
    if (world > 1 or force) and not dist.is_initialized():
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


class FlatGradBuckets:
    """Data-parallel gradient averaging for the fine-tuning step, one process per GPU.

    The gradients of the trainable parameters live as views inside a few flat buffers (filled in
    reverse registration order, the order backward produces them).  A post-accumulate hook counts
    the gradients of each buffer; the moment a buffer is complete its all-reduce is launched
    asynchronously (RCCL runs it on its own stream), so the exchange of the late layers overlaps
    with the backward pass of the early ones.  xGMI is point-to-point and a ring all-reduce is
    bound by one link, so a few large messages beat many small ones: the default 32 MiB gives four
    messages for a 107 MB ResNet-50 segmentation model.  `finish()` launches whatever is left
    (parameters that received no gradient), waits, and turns sums into means.

    Zeroing is one memset per buffer (`zero_grad`) instead of one per parameter; the optimizer
    must not replace `.grad` (do not call `optimizer.zero_grad(set_to_none=True)`).
    """

    def __init__(self, params, bucket_bytes: int = 32 << 20, align: int = 64):
        self.rank, self.world = world()
        # set True around every backward of an optimisation step except the last one (gradient
        # accumulation over chunks): buffers are exchanged only once the last backward fills them
        self.hold = False
        ps = [p for p in params if p.requires_grad]
        ps.reverse()
        self.buckets: List[dict] = []
        cur: List[torch.nn.Parameter] = []
        cur_bytes = 0

        def close():
            nonlocal cur, cur_bytes
            if cur:
                self.buckets.append(self._make_bucket(cur, align))
            cur, cur_bytes = [], 0

        for p in ps:
            if cur and (p.dtype != cur[0].dtype or p.device != cur[0].device or
                        cur_bytes + p.numel() * p.element_size() > bucket_bytes):
                close()
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
        close()
        self._hooks = []
        for bi, b in enumerate(self.buckets):
            for p in b["params"]:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(bi)))

    @staticmethod
    def _make_bucket(params, align):
        offs, pos = [], 0
        for p in params:
            offs.append(pos)
            pos += (p.numel() + align - 1) // align * align
        flat = torch.zeros(pos, dtype=params[0].dtype, device=params[0].device)
        for p, o in zip(params, offs):
            p.grad = flat[o:o + p.numel()].view_as(p)
        return {"params": list(params), "offsets": offs, "flat": flat, "pending": len(params), "work": None,
                "launched": False}

    def _make_hook(self, bi):
        def hook(_param):
            if self.hold:  # an earlier backward of an accumulation step: more gradient is to come
                return
            b = self.buckets[bi]
            b["pending"] -= 1
            if b["pending"] == 0:
                self._launch(b)
        return hook

    def _launch(self, b) -> None:
        if b["launched"] or self.world <= 1:
            b["launched"] = True
            return
        b["launched"] = True
        flat = b["flat"]
        if flat.is_cuda and dist.get_backend() == "gloo":  # single-GPU rehearsal: staged, synchronous
            host = flat.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            flat.copy_(host)
        else:
            b["work"] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True)

    def zero_grad(self) -> None:
        self.hold = False
        for b in self.buckets:
            b["flat"].zero_()
            b["pending"], b["work"], b["launched"] = len(b["params"]), None, False
            for p, o in zip(b["params"], b["offsets"]):
                if p.grad is None or p.grad.data_ptr() != b["flat"].data_ptr() + o * b["flat"].element_size():
                    p.grad = b["flat"][o:o + p.numel()].view_as(p)

    def finish(self) -> None:
        for b in self.buckets:
            self._launch(b)
        for b in self.buckets:
            if b["work"] is not None:
                b["work"].wait()
                b["work"] = None
            if self.world > 1:
                b["flat"].div_(self.world)

    def close(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []


def average_buffers_(module: torch.nn.Module) -> None:
    """Mean over ranks of every floating-point buffer (batch-norm running statistics), through
    one flat message; integer buffers (step counters) are equal on all ranks already."""
    _, n = world()
    bufs = [b for b in module.buffers() if b.is_floating_point()]
    if n <= 1 or not bufs:
        return
    flat = torch.cat([b.detach().reshape(-1).float() for b in bufs])
    all_reduce_sum_(flat)
    flat.div_(n)
    pos = 0
    for b in bufs:
        b.copy_(flat[pos:pos + b.numel()].view_as(b))
        pos += b.numel()
