"""CPU oracle for a whole pruning step (test infrastructure; never imported by spice_amd/).

Restates torch-pruning 1.6.0's MetaPruner.step() (prune_local / prune_global) plus
tp.pruner.function's index_select slicing, as invoked at pruning/pruner.py:495, with stock torch
CPU ops.  Groups are passed in as plain objects (`.members` with `.module/.kind/.offset()`,
`.size`, `.root`); discovering them is host logic outside the arithmetic checked here.
torch-pruning is absent from /root/reference and from this image: parity unpinned.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import importance as oi
from . import selection as osel


def group_importance(group, p: int, normalizer: str = "mean") -> Optional[np.ndarray]:
    terms = []
    for m in group.members:
        mod = m.module
        if m.kind == "out":
            terms.append(oi.magnitude_rows(mod.weight, p)[: group.size])
        elif m.kind == "dw":
            o = m.offset()
            terms.append(oi.magnitude_rows(mod.weight, p)[o: o + group.size])
        elif m.kind == "in":
            o = m.offset()
            terms.append(oi.magnitude_cols(mod.weight, p)[o: o + group.size])
        elif m.kind == "bn" and mod.affine:
            o = m.offset()
            terms.append(mod.weight.detach().abs().pow(p)[o: o + group.size])
    if not terms:
        return None
    return oi.group_magnitude(terms, normalizer).numpy()


def group_taylor_importance(group) -> Optional[np.ndarray]:
    """Per-channel first-order Taylor criterion of a group from the gradients on its producers."""
    terms = []
    for m in group.members:
        mod = m.module
        if m.kind in ("out", "dw"):
            o = 0 if m.kind == "out" else m.offset()
            terms.append(oi.taylor_rows(mod.weight, mod.weight.grad)[o: o + group.size])
    if not terms:
        return None
    return oi.group_taylor(terms).numpy()


def _slice_param(mod: nn.Module, attr: str, dim: int, keep: np.ndarray) -> None:
    t = getattr(mod, attr, None)
    if t is None:
        return
    new = torch.index_select(t.detach(), dim, torch.as_tensor(keep, dtype=torch.long)).contiguous()
    if isinstance(t, nn.Parameter):
        setattr(mod, attr, nn.Parameter(new, requires_grad=t.requires_grad))
    else:
        mod._buffers[attr] = new


def prune_group(group, pruned: np.ndarray) -> None:
    """Remove `pruned` (group-relative channel ids) from every coupled tensor."""
    for m in group.members:
        mod = m.module
        if m.kind == "out":
            keep = osel.keep_indices(mod.out_channels, pruned)
            _slice_param(mod, "weight", 0, keep)
            _slice_param(mod, "bias", 0, keep)
            mod.out_channels = len(keep)
        elif m.kind == "in":
            keep = osel.keep_indices(mod.in_channels, pruned + m.offset())
            _slice_param(mod, "weight", 1, keep)
            mod.in_channels = len(keep)
        elif m.kind == "dw":
            keep = osel.keep_indices(mod.out_channels, pruned + m.offset())
            _slice_param(mod, "weight", 0, keep)
            _slice_param(mod, "bias", 0, keep)
            mod.out_channels = mod.in_channels = mod.groups = len(keep)
        elif m.kind == "bn":
            keep = osel.keep_indices(mod.num_features, pruned + m.offset())
            for attr in ("weight", "bias", "running_mean", "running_var"):
                _slice_param(mod, attr, 0, keep)
            mod.num_features = len(keep)
    group.size -= len(pruned)


def step_local(groups: Sequence, init_sizes: Dict, ratio_of, step: int, steps: int, p: int = 2) -> List[np.ndarray]:
    """Group after group: importance on the model as already sliced, stable selection, slicing."""
    chosen = []
    for g in groups:
        imp = group_importance(g, p)
        n = osel.channels_to_prune(g.size, init_sizes[g.root], ratio_of(g), step, steps)
        if imp is None or n <= 0 or n >= g.size:
            chosen.append(np.zeros(0, dtype=np.int64))
            continue
        pruned = osel.select_local(imp, n)
        prune_group(g, pruned)
        chosen.append(pruned)
    return chosen


def step_global(groups: Sequence, initial_total: int, ratio: float, step: int, steps: int, p: int = 2) -> List[np.ndarray]:
    """All importances first, one threshold over their concatenation, then slicing."""
    imps = [group_importance(g, p) for g in groups]
    live = [(g, i) for g, i in zip(groups, imps) if i is not None]
    flat_n = sum(len(i) for _, i in live)
    n = flat_n - int(initial_total * (1 - ratio * step / steps))
    picks = osel.select_global([i for _, i in live], min(n, flat_n))
    chosen = []
    for (g, _), pruned in zip(live, picks):
        if 0 < len(pruned) < g.size:
            prune_group(g, pruned)
            chosen.append(pruned)
        else:
            chosen.append(np.zeros(0, dtype=np.int64))
    return chosen
