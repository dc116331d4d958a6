"""The pruner object: channel selection and physical slicing.

Restates what the reference gets from `tp.pruner.MagnitudePruner(...)`
(pruning/pruner.py:272-281, pruning/pruning_core.py:87-95,154-162, pruning/strategies.py:36-44):
an object with `.model` and `.step()`; each step advances a linear ratio schedule, picks the
least important channels of every group (per group, or against one global threshold) and
physically slices all coupled tensors.  Selection arithmetic follows torch-pruning 1.6.0's
MetaPruner; exact ties, which upstream's unstable argsort leaves open, go to the lower channel
index.  Slicing is one launch of the HIP gather kernel per group (local mode, where a group's
importance must see the slicing of the groups before it) or per step (global mode).
"""
from __future__ import annotations

import logging
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from .depgraph import BN, DW, IN, OUT, ChannelGroup, build_groups
from .importance import Importance, MagnitudeImportance

logger = logging.getLogger(__name__)

_ALIGN = 64  # elements: every sliced tensor starts on a 256-byte boundary of the step's buffer


def _expand(mods: Optional[Iterable[nn.Module]]) -> set:
    out = set()
    for m in mods or []:
        for sub in m.modules():
            out.add(sub)
    return out


def keep_after_removing(n: int, removed: np.ndarray) -> np.ndarray:
    mask = np.ones(n, dtype=bool)
    mask[removed] = False
    return np.nonzero(mask)[0].astype(np.int64)


class SliceJob:
    """Physical removal of channels from every tensor coupled to a set of groups, as ONE launch of
    the gather kernel.  `build` does all host work (kept-index lists, launch plan, uploads) without
    touching the model; `run` allocates the destination buffer, launches and installs the sliced
    tensors; `undo` puts the original tensors back (used to evaluate a candidate and move on)."""

    def __init__(self, decisions: Sequence[Tuple[ChannelGroup, np.ndarray]]):
        self.decisions = [(g, np.asarray(p, dtype=np.int64)) for g, p in decisions if len(p)]
        rm = {OUT: {}, IN: {}, BN: {}, DW: {}}
        for g, pruned in self.decisions:
            for m in g.members:
                off = 0 if m.kind == OUT else m.offset()
                rm[m.kind].setdefault(id(m.module), (m.module, []))[1].append(pruned + off)
        # (owner, attr, keep_out, keep_in) per tensor; attribute updates to apply at run time
        self.entries: List[Tuple[nn.Module, str, Optional[np.ndarray], Optional[np.ndarray]]] = []
        self.attr_updates: List[Tuple[nn.Module, Dict[str, int]]] = []

        def add(owner, attr, keep_out, keep_in):
            if getattr(owner, attr, None) is not None:
                self.entries.append((owner, attr, keep_out, keep_in))

        for cid in set(rm[OUT]) | set(rm[IN]):
            conv = (rm[OUT].get(cid) or rm[IN].get(cid))[0]
            ko = ki = None
            upd = {}
            if cid in rm[OUT]:
                ko = keep_after_removing(conv.out_channels, np.concatenate(rm[OUT][cid][1]))
                upd["out_channels"] = len(ko)
            if cid in rm[IN]:
                ki = keep_after_removing(conv.in_channels, np.concatenate(rm[IN][cid][1]))
                upd["in_channels"] = len(ki)
            add(conv, "weight", ko, ki)
            if ko is not None:
                add(conv, "bias", ko, None)
            self.attr_updates.append((conv, upd))
        for conv, lst in rm[DW].values():
            k = keep_after_removing(conv.out_channels, np.concatenate(lst))
            add(conv, "weight", k, None)
            add(conv, "bias", k, None)
            self.attr_updates.append((conv, {"out_channels": len(k), "in_channels": len(k), "groups": len(k)}))
        for bn, lst in rm[BN].values():
            k = keep_after_removing(bn.num_features, np.concatenate(lst))
            for attr in ("weight", "bias", "running_mean", "running_var"):
                add(bn, attr, k, None)
            self.attr_updates.append((bn, {"num_features": len(k)}))
        self.plan: Optional[ops.GatherPlan] = None
        self._undo = None

    def index_arrays(self) -> List[np.ndarray]:
        out = []
        for _, _, ko, ki in self.entries:
            if ko is not None:
                out.append(ko)
            if ki is not None:
                out.append(ki)
        return out

    def build(self, idx_device: Optional[torch.Tensor] = None, idx_start: int = 0,
              with_grads: bool = True, defer_upload: bool = False) -> "SliceJob":
        """Launch plan for the model as it stands now.  `idx_device` (int64) may be a shared buffer
        that already holds this job's index arrays, concatenated, from element `idx_start`."""
        if not self.entries:
            return self
        self.sources: List[Tuple[nn.Module, str, bool]] = []  # (owner, attr, is_grad)
        srcs, keeps = [], []
        for owner, attr, ko, ki in self.entries:
            t = getattr(owner, attr)
            srcs.append(t.detach())
            keeps.append((ko, ki))
            self.sources.append((owner, attr, False))
        if with_grads:
            for owner, attr, ko, ki in self.entries:
                t = getattr(owner, attr)
                if isinstance(t, nn.Parameter) and t.grad is not None:
                    srcs.append(t.grad)
                    keeps.append((ko, ki))
                    self.sources.append((owner, attr, True))
        for t in srcs:
            if t.dtype != torch.float32 or not t.is_contiguous() or t.device.type != "cuda":
                raise RuntimeError("channel slicing needs contiguous fp32 tensors on a HIP device")
        self.device = srcs[0].device
        self.srcs = srcs
        _, src_offs = ops._relative_offsets(srcs, "tensors to slice")
        specs, self.shapes, self.dst_offs = [], [], []
        ipos, dpos = idx_start, 0
        first = {}  # the gradient entries reuse the index arrays of their parameter
        for k, (t, so, (ko, ki)) in enumerate(zip(srcs, src_offs, keeps)):
            rows = t.shape[0]
            cols = t.shape[1] if t.dim() > 1 else 1
            inner = t.numel() // max(rows * cols, 1) if t.dim() > 2 else 1
            owner, attr, is_grad = self.sources[k]
            if is_grad:
                o_off, i_off = first[(id(owner), attr)]
            else:
                o_off = i_off = -1
                if ko is not None:
                    o_off, ipos = ipos, ipos + len(ko)
                if ki is not None:
                    i_off, ipos = ipos, ipos + len(ki)
                first[(id(owner), attr)] = (o_off, i_off)
            d_rows = len(ko) if ko is not None else rows
            d_cols = len(ki) if ki is not None else cols
            specs.append(ops.GatherSpec(so, dpos, o_off, i_off, d_rows, d_cols, cols, max(inner, 1)))
            shape = list(t.shape)
            shape[0] = d_rows
            if t.dim() > 1:
                shape[1] = d_cols
            self.shapes.append(shape)
            self.dst_offs.append(dpos)
            dpos += (d_rows * d_cols * max(inner, 1) + _ALIGN - 1) // _ALIGN * _ALIGN
        self.dst_elems = dpos
        self.plan = ops.GatherPlan(specs, self.device, defer_upload=defer_upload)   # (deferred: the caller sets plan.items)
        if idx_device is None:
            arrays = self.index_arrays()
            idx_device = (ops.upload(torch.from_numpy(np.concatenate(arrays)), self.device) if arrays else
                          torch.zeros(1, dtype=torch.int64, device=self.device))
        self.idx = idx_device
        return self

    def run(self, keep_undo: bool = False) -> int:
        """Launch and install.  Returns the algorithmic bytes of the launch."""
        if not self.entries:
            return 0
        if self.plan is None:
            self.build()
        dst = torch.empty(max(self.dst_elems, 1), dtype=torch.float32, device=self.device)
        self.plan.run(self.srcs, dst, self.idx)
        if keep_undo:
            self._undo = ([(o, a, o._parameters[a] if a in o._parameters else o._buffers[a]) for o, a, _, _ in self.entries],
                          [(m, {k: m.__dict__[k] for k in upd}) for m, upd in self.attr_updates],
                          [(g, g.size) for g, _ in self.decisions])

        def view(k):
            n = int(np.prod(self.shapes[k])) if self.shapes[k] else 1
            return dst[self.dst_offs[k]: self.dst_offs[k] + n].view(self.shapes[k])

        new_grads = {(id(o), a): view(k) for k, (o, a, is_grad) in enumerate(self.sources) if is_grad}
        for k, (owner, attr, is_grad) in enumerate(self.sources):
            if is_grad:
                continue
            # parameters, buffers and the modules' size attributes go straight into their tables: nn.Module's
            # __setattr__ walks several of them per call, and a sweep makes ~20 such writes per candidate and undoes them
            old = owner._parameters.get(attr)
            if old is not None:
                p = nn.Parameter(view(k), requires_grad=old.requires_grad)
                if (id(owner), attr) in new_grads:
                    p.grad = new_grads[(id(owner), attr)]
                owner._parameters[attr] = p
            else:
                owner._buffers[attr] = view(k)
        for m, upd in self.attr_updates:
            m.__dict__.update(upd)
        for g, pruned in self.decisions:
            g.size -= len(pruned)
        return self.plan.algorithmic_bytes()

    def undo(self) -> None:
        if self._undo is None:
            return
        tensors, attrs, sizes = self._undo
        for owner, attr, old in tensors:
            if isinstance(old, nn.Parameter):
                owner._parameters[attr] = old
            else:
                owner._buffers[attr] = old
        for m, upd in attrs:
            m.__dict__.update(upd)
        for g, size in sizes:
            g.size = size
        self._undo = None


def apply_pruning(decisions: Sequence[Tuple[ChannelGroup, np.ndarray]]) -> int:
    """Physically remove `pruned` channels of every (group, pruned) pair with ONE gather launch.
    Returns the number of bytes the launch had to move (algorithmic)."""
    return SliceJob(decisions).build().run()


class MagnitudePruner:
    def __init__(self, model: nn.Module, example_inputs: Optional[torch.Tensor] = None,
                 importance: Optional[Importance] = None, pruning_ratio: float = 0.5,
                 pruning_ratio_dict: Optional[Dict[nn.Module, float]] = None,
                 iterative_steps: int = 1, ignored_layers: Optional[Sequence[nn.Module]] = None,
                 global_pruning: bool = False, round_to: Optional[int] = None,
                 max_pruning_ratio: float = 1.0):
        self.model = model
        self.example_inputs = example_inputs
        self.importance = importance if importance is not None else MagnitudeImportance(p=2)
        self.pruning_ratio = pruning_ratio
        self.iterative_steps = iterative_steps
        self.global_pruning = global_pruning
        self.round_to = round_to
        self.max_pruning_ratio = max_pruning_ratio
        self.ignored = _expand(ignored_layers)
        self.current_step = 0
        self.groups = build_groups(model)
        self.per_step_ratio = [pruning_ratio * i / iterative_steps for i in range(iterative_steps + 1)]
        self.ratio_dict: Dict[nn.Module, List[float]] = {}
        for mod, r in (pruning_ratio_dict or {}).items():
            for sub in mod.modules():
                if isinstance(sub, (nn.Conv2d, nn.Linear)):
                    self.ratio_dict[sub] = [r * i / iterative_steps for i in range(iterative_steps + 1)]
        self.init_channels = {g.root: g.size for g in self.groups}
        self.initial_total_channels = sum(g.size for g in self.prunable_groups())
        self.last_step_bytes = 0

    # a group is left alone if it is frozen or one of its producers is an ignored layer
    def prunable_groups(self) -> List[ChannelGroup]:
        out = []
        for g in self.groups:
            if g.frozen or g.root in self.ignored:
                continue
            if any(m.module in self.ignored for m in g.out_members()):
                continue
            out.append(g)
        return out

    def target_ratio(self, g: ChannelGroup) -> float:
        sched = self.ratio_dict.get(g.root, self.per_step_ratio)
        return sched[self.current_step]

    def _importance_of(self, g: ChannelGroup) -> Optional[np.ndarray]:
        imp = self.importance(g)
        if imp is None:
            return None
        if isinstance(imp, torch.Tensor):
            imp = imp.detach().float().cpu().numpy()
        imp = np.asarray(imp, dtype=np.float32)
        if imp.size != g.size:
            raise RuntimeError(f"importance returned {imp.size} scores for a group of {g.size} channels")
        return imp

    def step(self) -> None:
        self.current_step += 1
        if self.current_step > self.iterative_steps:
            raise IndexError("pruner stepped more often than iterative_steps")
        self.last_step_bytes = 0
        if self.global_pruning:
            self._step_global()
        else:
            self._step_local()

    def _n_to_prune_local(self, g: ChannelGroup) -> int:
        ratio = min(self.target_ratio(g), self.max_pruning_ratio)
        n_pruned = g.size - int(self.init_channels[g.root] * (1 - ratio))
        if self.round_to:
            n_pruned -= n_pruned % self.round_to
        return n_pruned

    @staticmethod
    def _independent_levels(groups: Sequence[ChannelGroup]) -> List[List[ChannelGroup]]:
        """Groups in module order, split into levels: a group's level is one more than the highest level of
        an EARLIER group that shares a layer with it.  A group's importance reads only its own members'
        tensors, so slicing the groups of one level together (after all lower levels) gives exactly what
        slicing group after group in module order gives."""
        last: Dict[int, int] = {}
        levels: List[List[ChannelGroup]] = []
        for g in groups:
            mods = {id(m.module) for m in g.members}
            lvl = 1 + max((last.get(k, -1) for k in mods), default=-1)
            for k in mods:
                last[k] = lvl
            while len(levels) <= lvl:
                levels.append([])
            levels[lvl].append(g)
        return levels

    def _step_local(self) -> None:
        groups = self.prunable_groups()
        if hasattr(self.importance, "batch"):
            # one importance pass (two launches, one device-to-host copy) and one gather launch per LEVEL of
            # mutually independent groups instead of per group: the same channels, fewer host round trips
            for level in self._independent_levels(groups):
                decisions = []
                for g, imp in zip(level, self.importance.batch(level)):
                    if imp is None:
                        continue
                    imp = np.asarray(imp, dtype=np.float32)
                    if imp.size != g.size:
                        raise RuntimeError(f"importance returned {imp.size} scores for a group of {g.size} channels")
                    n_pruned = self._n_to_prune_local(g)
                    if n_pruned <= 0 or n_pruned >= g.size:
                        continue
                    order = np.argsort(imp, kind="stable")  # ascending score, then ascending index
                    decisions.append((g, np.sort(order[:n_pruned])))
                if decisions:
                    self.last_step_bytes += apply_pruning(decisions)
            return
        for g in groups:   # importance plug-ins without a batch form: group after group
            imp = self._importance_of(g)
            if imp is None:
                continue
            n_pruned = self._n_to_prune_local(g)
            if n_pruned <= 0 or n_pruned >= g.size:
                continue
            order = np.argsort(imp, kind="stable")  # ascending score, then ascending index
            pruned = np.sort(order[:n_pruned])
            self.last_step_bytes += apply_pruning([(g, pruned)])

    def _step_global(self) -> None:
        groups = self.prunable_groups()
        if hasattr(self.importance, "batch"):
            imps = self.importance.batch(groups)
        else:
            imps = [self._importance_of(g) for g in groups]
        pairs = [(g, i) for g, i in zip(groups, imps) if i is not None]
        if not pairs:
            return
        flat = np.concatenate([i for _, i in pairs]).astype(np.float32)
        ratio = self.per_step_ratio[self.current_step]
        n_pruned = flat.size - int(self.initial_total_channels * (1 - ratio))
        if n_pruned <= 0:
            return
        n_pruned = min(n_pruned, flat.size)
        thr = np.partition(flat, n_pruned - 1)[n_pruned - 1]
        decisions = []
        for g, imp in pairs:
            pruned = np.nonzero(imp <= thr)[0].astype(np.int64)
            if self.round_to:
                pruned = pruned[: len(pruned) - len(pruned) % self.round_to]
            if 0 < len(pruned) < g.size:
                decisions.append((g, pruned))
        self.last_step_bytes += apply_pruning(decisions)


def count_ops_and_params(model: nn.Module, example_inputs: torch.Tensor) -> Tuple[int, int]:
    """(multiply-accumulates of the conv/linear layers for one forward, parameter count)."""
    macs = [0]
    hooks = []

    def conv_hook(m, _inp, out):
        k = m.kernel_size[0] * m.kernel_size[1] * (m.in_channels // m.groups)
        macs[0] += out.numel() * k

    def lin_hook(m, _inp, out):
        macs[0] += out.numel() * m.in_features

    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            hooks.append(m.register_forward_hook(conv_hook))
        elif isinstance(m, nn.Linear):
            hooks.append(m.register_forward_hook(lin_hook))
    was = model.training
    model.eval()
    try:
        with torch.no_grad():
            model(example_inputs)
    finally:
        for h in hooks:
            h.remove()
        model.train(was)
    return macs[0], sum(p.numel() for p in model.parameters())
