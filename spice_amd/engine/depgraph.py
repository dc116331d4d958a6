"""Coupled-channel analysis by torch.fx.

A *channel space* is the set of feature-map channels one producer convolution creates.  Spaces
that meet in an element-wise add or multiply are merged (residual streams, squeeze-excite
gates); every module that reads or normalises a space becomes a member of it.  The result is
the list of channel groups torch-pruning 1.6.0's DependencyGraph.get_all_groups would yield for
these architectures (root = the first producer convolution in module order), which is what the
reference's pruner iterates inside `.step()` (pruning/pruner.py:495).
"""
from __future__ import annotations

import operator
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import torch
import torch.fx as fx
import torch.nn as nn
import torch.nn.functional as F

OUT, IN, BN, DW = "out", "in", "bn", "dw"


class _Space:
    def __init__(self, size: int, order: int):
        self.size = size
        self.order = order
        self.parent: Optional["_Space"] = None
        self.frozen = False

    def find(self) -> "_Space":
        s = self
        while s.parent is not None:
            s = s.parent
        return s


@dataclass
class Member:
    """One module coupled to a group.  For IN/BN/DW members the group's channels sit at
    `offset()` .. `offset()+size` of the module's channel axis (consumers of a concatenation)."""

    module: nn.Module
    name: str
    kind: str
    segments: List["ChannelGroup"] = field(default_factory=list)  # channel axis, in order
    seg_index: int = 0

    def offset(self) -> int:
        return sum(g.size for g in self.segments[: self.seg_index])


class ChannelGroup:
    def __init__(self, size: int):
        self.size = size
        self.initial_size = size
        self.members: List[Member] = []
        self.frozen = False
        self.root: Optional[nn.Module] = None
        self.root_name = ""

    def out_members(self) -> List[Member]:
        return [m for m in self.members if m.kind in (OUT, DW)]

    def __repr__(self):
        return f"ChannelGroup(root={self.root_name}, size={self.size}, members={len(self.members)}, frozen={self.frozen})"


_PASS_MODULES = (nn.ReLU, nn.ReLU6, nn.SiLU, nn.Sigmoid, nn.Identity, nn.Dropout, nn.Dropout2d,
                 nn.MaxPool2d, nn.AvgPool2d, nn.AdaptiveAvgPool2d, nn.UpsamplingBilinear2d,
                 nn.UpsamplingNearest2d, nn.Upsample)
_PASS_FUNCTIONS = {F.relu, F.silu, F.interpolate, torch.sigmoid, F.sigmoid, F.dropout, F.max_pool2d,
                   F.adaptive_avg_pool2d, torch.relu, F.hardtanh, F.relu6}
_PASS_METHODS = {"mean", "contiguous", "relu", "sigmoid", "float", "half", "to"}


class _Tracer(fx.Tracer):
    def is_leaf_module(self, m: nn.Module, qualname: str) -> bool:
        if isinstance(m, _PASS_MODULES + (nn.BatchNorm2d, nn.Conv2d)):  # also their subclasses defined outside torch.nn
            return True
        return m.__module__.startswith("torch.nn") and not isinstance(m, nn.Sequential)


def build_groups(model: nn.Module) -> List[ChannelGroup]:
    """Channel groups of `model` in module order of their root convolution."""
    graph = _Tracer().trace(model)
    modules = dict(model.named_modules())
    order = {m: i for i, (_, m) in enumerate(model.named_modules())}
    layouts: Dict[fx.Node, Optional[List[_Space]]] = {}
    # records: (space, module, name, kind, layout, seg_index)
    records = []
    spaces: List[_Space] = []

    def new_space(size: int, pos: int) -> _Space:
        s = _Space(size, pos)
        spaces.append(s)
        return s

    def freeze(layout):
        for s in layout or []:
            s.find().frozen = True

    def union(a: Optional[List[_Space]], b: Optional[List[_Space]]) -> Optional[List[_Space]]:
        if a is None:
            return b
        if b is None:
            return a
        if len(a) != len(b) or any(x.size != y.size for x, y in zip(a, b)):
            freeze(a)
            freeze(b)
            return a
        for x, y in zip(a, b):
            rx, ry = x.find(), y.find()
            if rx is not ry:
                keep, drop = (rx, ry) if rx.order <= ry.order else (ry, rx)
                drop.parent = keep
                keep.frozen = keep.frozen or drop.frozen
        return a

    def tensor_args(node: fx.Node) -> List[fx.Node]:
        found = []

        def visit(a):
            if isinstance(a, fx.Node):
                if layouts.get(a) is not None:
                    found.append(a)
            elif isinstance(a, (list, tuple)):
                for x in a:
                    visit(x)

        for a in node.args:
            visit(a)
        for a in node.kwargs.values():
            visit(a)
        return found

    for node in graph.nodes:
        lay = None
        if node.op == "placeholder":
            s = new_space(-1, -1)  # the network input: never prunable
            s.frozen = True
            lay = [s]
        elif node.op == "call_module":
            m = modules[node.target]
            src = layouts.get(node.args[0]) if node.args and isinstance(node.args[0], fx.Node) else None
            if isinstance(m, nn.Conv2d):
                depthwise = m.groups > 1 and m.groups == m.in_channels == m.out_channels
                if src is not None and src[0].size == -1:
                    src[0].size = m.in_channels
                if depthwise and src is not None:
                    for i, s in enumerate(src):
                        records.append((s, m, node.target, DW, src, i))
                    lay = src
                else:
                    if src is not None:
                        for i, s in enumerate(src):
                            records.append((s, m, node.target, IN, src, i))
                        if m.groups != 1:
                            freeze(src)
                    out = new_space(m.out_channels, order[m])
                    if m.groups != 1:
                        out.frozen = True
                    records.append((out, m, node.target, OUT, [out], 0))
                    lay = [out]
            elif isinstance(m, (nn.BatchNorm2d,)):
                if src is not None:
                    for i, s in enumerate(src):
                        records.append((s, m, node.target, BN, src, i))
                lay = src
                for extra in node.args[1:]:  # a residual added inside the module couples its channels
                    if isinstance(extra, fx.Node) and layouts.get(extra) is not None:
                        lay = union(lay, layouts[extra])
            elif isinstance(m, (nn.GroupNorm, nn.LayerNorm, nn.InstanceNorm2d)):
                freeze(src)  # channels normalised in blocks: slicing would change the statistics
                lay = src
            elif isinstance(m, _PASS_MODULES):
                lay = src
            else:
                freeze(src)
                lay = None
        elif node.op == "call_function":
            ins = tensor_args(node)
            if node.target in (operator.add, torch.add, operator.iadd, operator.mul, torch.mul,
                               operator.imul, operator.sub, torch.sub):
                lay = None
                for a in ins:
                    lay = union(lay, layouts[a])
            elif node.target in (torch.cat, torch.concat):
                dim = node.kwargs.get("dim", node.args[1] if len(node.args) > 1 else 0)
                if dim in (1, -3):
                    lay = []
                    for a in ins:
                        lay.extend(layouts[a])
                else:
                    lay = layouts[ins[0]] if ins else None
            elif node.target in _PASS_FUNCTIONS:
                lay = layouts[ins[0]] if ins else None
            elif node.target in (getattr, operator.getitem):
                lay = None
            else:
                for a in ins:
                    freeze(layouts[a])
                lay = None
        elif node.op == "call_method":
            ins = tensor_args(node)
            if node.target in _PASS_METHODS:
                lay = layouts[ins[0]] if ins else None
            else:
                for a in ins:
                    freeze(layouts[a])
        elif node.op == "output":
            for a in tensor_args(node):
                freeze(layouts[a])
        layouts[node] = lay

    # one ChannelGroup per union-find root
    by_root: Dict[_Space, ChannelGroup] = {}
    for s in spaces:
        r = s.find()
        if r not in by_root:
            by_root[r] = ChannelGroup(r.size)
        by_root[r].frozen = by_root[r].frozen or r.frozen
    seen = set()
    for s, m, name, kind, layout, idx in records:
        g = by_root[s.find()]
        key = (id(g), id(m), kind, idx)
        if key in seen:
            continue
        seen.add(key)
        g.members.append(Member(m, name, kind, [by_root[x.find()] for x in layout], idx))
    groups = []
    for g in by_root.values():
        roots = [mm for mm in g.members if mm.kind == OUT]
        if not roots or g.size <= 0:
            continue
        first = min(roots, key=lambda mm: order[mm.module])
        g.root, g.root_name = first.module, first.name
        groups.append(g)
    groups.sort(key=lambda g: order[g.root])
    return groups
