"""Importance criteria evaluated on channel groups.

`MagnitudeImportance` restates torch-pruning 1.6.0's criterion of the same name with its
defaults (group_reduction="mean", normalizer="mean", bias=False; members: conv out/in channels
and BatchNorm scales), which the reference constructs at pruning/pruner.py:268-270,
pruning/pruning_core.py:143-145 and pruning/strategies.py:53.  The per-channel sums come from the
HIP reductions (one launch for all out-channel terms, one two-stage launch for all in-channel
terms); the handful of per-group scalars is combined on the host in fp32.
Protocol (pruning/sensitivity.py:70-112): `importance(group) -> 1-D float tensor`, low = pruned first.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from .. import _native as nat
from .. import ops
from .depgraph import BN, DW, IN, OUT, ChannelGroup


class Importance:
    """Base class of every criterion (the role of tp.importance.Importance)."""

    def __call__(self, group: ChannelGroup) -> torch.Tensor:  # pragma: no cover - interface
        raise NotImplementedError


def _mode_for(p: int) -> int:
    if p == 1:
        return nat.IMP_L1
    if p == 2:
        return nat.IMP_L2
    raise ValueError(f"importance norm p={p} not supported (the CLI offers 1 and 2)")


def _normalise(imp: np.ndarray, normalizer: Optional[str]) -> np.ndarray:
    if normalizer is None:
        return imp
    if normalizer == "mean":
        return imp / imp.mean(dtype=np.float32)
    if normalizer == "lamp":  # layer-adaptive: score / sum of the scores not smaller than it
        order = np.argsort(-imp, kind="stable")
        srt = imp[order]
        srt = srt / np.cumsum(srt, dtype=np.float32)
        out = np.empty_like(srt)
        out[order] = srt
        return out
    raise ValueError(f"unknown normalizer {normalizer}")


class MagnitudeImportance(Importance):
    def __init__(self, p: int = 2, group_reduction: str = "mean", normalizer: Optional[str] = "mean"):
        self.p = p
        self.mode = _mode_for(p)
        if group_reduction != "mean":
            raise ValueError("only group_reduction='mean' is implemented")
        self.normalizer = normalizer

    # -- device part: per-channel sums of every member tensor of `groups`, two launches in all
    def _member_terms(self, groups: Sequence[ChannelGroup]) -> Dict[int, np.ndarray]:
        row_tensors: List[torch.Tensor] = []
        row_key: Dict[int, int] = {}
        col_mods: List[nn.Conv2d] = []
        col_key: Dict[int, int] = {}
        for g in groups:
            for m in g.members:
                if m.kind in (OUT, DW):
                    if id(m.module) not in row_key:
                        row_key[id(m.module)] = len(row_tensors)
                        row_tensors.append(m.module.weight.detach())
                elif m.kind == BN:
                    if m.module.affine and id(m.module) not in row_key:
                        row_key[id(m.module)] = len(row_tensors)
                        row_tensors.append(m.module.weight.detach().view(-1, 1))
                elif m.kind == IN:
                    if id(m.module) not in col_key:
                        col_key[id(m.module)] = len(col_mods)
                        col_mods.append(m.module)
        if not row_tensors:
            return {}
        device = row_tensors[0].device
        flat, offs = ops.channel_importance(row_tensors, self.mode)
        pieces = [flat]
        col_out_offs: List[int] = []
        if col_mods:
            weights = [c.weight.detach() for c in col_mods]
            for w in weights:
                nat.require_device_tensor(w, "weight", torch.float32)
            _, w_offs = ops._relative_offsets(weights, "weights")
            layers, pos = [], 0
            for c, wo in zip(col_mods, w_offs):
                cin = c.weight.shape[1]
                layers.append(ops.ColsLayout(wo, pos, c.weight.shape[0], cin,
                                             c.weight.shape[2] * c.weight.shape[3]))
                col_out_offs.append(pos)
                pos += cin
            plan = ops.ColsPlan(layers, device)
            cols = torch.empty(max(pos, 1), dtype=torch.float32, device=device)
            plan.run(weights, cols, self.mode)
            pieces.append(cols)
        host = torch.cat(pieces).cpu().numpy()  # the step's single device->host copy
        n_rows = flat.numel()
        terms: Dict[int, np.ndarray] = {}
        for key, i in row_key.items():
            t = row_tensors[i]
            terms[key] = host[offs[i]: offs[i] + t.shape[0]]
        for key, i in col_key.items():
            c = col_mods[i]
            terms[("in", key)] = host[n_rows + col_out_offs[i]: n_rows + col_out_offs[i] + c.weight.shape[1]]
        return terms

    def _combine(self, g: ChannelGroup, terms: Dict) -> Optional[np.ndarray]:
        acc = np.zeros(g.size, dtype=np.float32)
        n = 0
        for m in g.members:
            if m.kind == OUT:
                acc += terms[id(m.module)][: g.size]
            elif m.kind == DW:
                o = m.offset()
                acc += terms[id(m.module)][o: o + g.size]
            elif m.kind == BN:
                if not m.module.affine:
                    continue
                o = m.offset()
                acc += terms[id(m.module)][o: o + g.size]
            elif m.kind == IN:
                o = m.offset()
                acc += terms[("in", id(m.module))][o: o + g.size]
            else:
                continue
            n += 1
        if n == 0:
            return None
        return _normalise(acc / np.float32(n), self.normalizer)

    def batch(self, groups: Sequence[ChannelGroup]) -> List[Optional[np.ndarray]]:
        terms = self._member_terms(groups)
        return [self._combine(g, terms) for g in groups]

    def __call__(self, group: ChannelGroup) -> torch.Tensor:
        imp = self.batch([group])[0]
        return torch.ones(group.size) if imp is None else torch.from_numpy(imp.copy())


class LAMPImportance(MagnitudeImportance):
    """Magnitude (p=2) with the layer-adaptive LAMP normaliser."""

    def __init__(self, p: int = 2):
        super().__init__(p=p, normalizer="lamp")


class RandomImportance(Importance):
    def __call__(self, group: ChannelGroup) -> torch.Tensor:
        return torch.rand(group.size)


class TaylorImportance(Importance):
    """First-order Taylor |w * dL/dw| summed per output channel of the group's producers; needs
    gradients to be present on the model when the pruner steps."""

    def __call__(self, group: ChannelGroup) -> torch.Tensor:
        weights, grads, members = [], [], []
        for m in group.members:
            if m.kind in (OUT, DW):
                if m.module.weight.grad is None:
                    raise RuntimeError("TaylorImportance needs gradients: run a backward pass first")
                weights.append(m.module.weight.detach())
                grads.append(m.module.weight.grad.detach())
                members.append(m)
        if not weights:
            return torch.ones(group.size)
        flat, offs = ops.channel_importance(weights, nat.IMP_TAYLOR, grads=grads)
        host = flat.cpu().numpy()
        acc = np.zeros(group.size, dtype=np.float32)
        for m, off in zip(members, offs):
            o = 0 if m.kind == OUT else m.offset()
            acc += host[off + o: off + o + group.size]
        acc /= np.float32(len(members))
        return torch.from_numpy(acc / acc.mean(dtype=np.float32))
