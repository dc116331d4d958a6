"""CPU oracle for channel selection and physical slicing (test infrastructure; never imported
by spice_amd/).  Restates torch-pruning 1.6.0 MetaPruner.step() / prune_local / prune_global and
tp.pruner.function.prune_* as invoked at pruning/pruner.py:495.  torch-pruning is absent from
/root/reference and from this image: parity unpinned.  Ties, which upstream's unstable argsort
leaves unspecified, are broken towards the lower channel index.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import numpy as np
import torch


def channels_to_prune(current: int, initial: int, ratio: float, step: int, steps: int) -> int:
    """Linear schedule: n = current - int(initial * (1 - ratio * step / steps)), floored at 0."""
    return max(current - int(initial * (1 - ratio * step / steps)), 0)


def select_local(importance: np.ndarray, n_pruned: int) -> np.ndarray:
    """argsort(imp)[:n], ascending score then ascending index; returned sorted."""
    order = np.argsort(importance, kind="stable")
    return np.sort(order[:n_pruned]).astype(np.int64)


def select_global(importances: Sequence[np.ndarray], n_pruned: int) -> List[np.ndarray]:
    """k-th smallest over the concatenation of all groups; per group every imp <= threshold."""
    if n_pruned <= 0:
        return [np.zeros(0, dtype=np.int64) for _ in importances]
    flat = np.concatenate([np.asarray(x, dtype=np.float32) for x in importances])
    thr = np.partition(flat, n_pruned - 1)[n_pruned - 1]
    return [np.nonzero(np.asarray(x, dtype=np.float32) <= thr)[0].astype(np.int64) for x in importances]


def keep_indices(n_channels: int, pruned: np.ndarray) -> np.ndarray:
    """keep = sorted(set(range(C)) - set(pruned))."""
    mask = np.ones(n_channels, dtype=bool)
    mask[np.asarray(pruned, dtype=np.int64)] = False
    return np.nonzero(mask)[0].astype(np.int64)


def slice_tensor(x: torch.Tensor, keep_out: Optional[np.ndarray], keep_in: Optional[np.ndarray]) -> torch.Tensor:
    """index_select on dim 0 (out channels) and/or dim 1 (in channels), contiguous copy."""
    y = x
    if keep_out is not None:
        y = torch.index_select(y, 0, torch.as_tensor(keep_out, dtype=torch.long))
    if keep_in is not None:
        y = torch.index_select(y, 1, torch.as_tensor(keep_in, dtype=torch.long))
    return y.contiguous()
