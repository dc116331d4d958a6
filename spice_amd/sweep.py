"""Per-layer x pruning-ratio sensitivity sweep.

A candidate = (channel group g, ratio r): remove the floor(r * C_g) least important channels of g
(importance = the group magnitude criterion, ties to the lower index), evaluate the sliced model
on one batch, and record (a) the mIoU confusion matrix against the labels and (b) the KD-Lite loss
against the un-pruned model's logits.  Composed from the reference's primitives; the reference
itself only runs the one-shot Taylor analysis (pruning/pruning_core.py:99-133).

Per candidate the device sees: one gather launch (slice the group's tensors), the forward of the
sliced network from the first block the group touches up to the head's convolution (the library's
kernels under fp16 autocast; what the blocks in front of it produce is what they produced in the
un-pruned network's forward of the same sweep, bit for bit, and is taken from there) and ONE
launch for the rest: the head's bilinear up-sampling, the KD-Lite loss against the un-pruned
network's logits (whose softmax is taken once per sweep) and the arg-max + confusion histogram,
without the full-size logits ever being written.  Networks whose head does not fit that launch run
the forward to its end, the fused arg-max + histogram launch and the KD-Lite launch.
Everything the host has to prepare (kept-index lists, launch plans, MIOpen's solver lookups for
the convolution shapes the candidates will bring) is done for a rank's whole share before its
first candidate runs, and the fp16 copies of the unchanged weights are made once per sweep, so
the candidate loop only enqueues work and the device queue never drains.
Candidates are independent: ranks evaluate disjoint subsets and the [n_candidates, C, C] table
is combined with one all-reduce.
"""
from __future__ import annotations

import contextlib
import copy
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import dist as sdist
from . import ops
from .engine import MagnitudeImportance, MagnitudePruner
from .engine.pruner import SliceJob
from .nets.layers import Conv1x1, Conv1x1Bias, Conv3x3Bias, StemConv7x7, conv3x3_served_by_library
from .nets.prefix import PrefixTape, StopForward

_LATER = object()   # SliceJob.build: the index buffer is handed over after the build
DEFAULT_RATIOS = (0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9)


@dataclass
class Candidate:
    index: int
    group: int          # index into SensitivitySweep.groups
    ratio: float
    pruned: np.ndarray  # channel ids removed from the group, ascending
    cost: float         # relative cost estimate (for balancing ranks): see SensitivitySweep.candidate_costs


class SensitivitySweep:
    def __init__(self, model: nn.Module, num_classes: int, ratios: Sequence[float] = DEFAULT_RATIOS,
                 importance=None, ignored_layers: Optional[Sequence[nn.Module]] = None,
                 autocast_dtype: Optional[torch.dtype] = torch.float16,
                 kd_temperature: float = 4.0, kd_alpha: float = 0.7):
        self.model = model.eval()
        self._ctor = dict(num_classes=num_classes, ratios=tuple(ratios), importance=importance,
                          ignored_layers=list(ignored_layers) if ignored_layers is not None else None,
                          autocast_dtype=autocast_dtype, kd_temperature=kd_temperature, kd_alpha=kd_alpha)
        self._lanes: List["SensitivitySweep"] = [self]
        self._lane_assign: Dict[int, int] = {}   # candidate -> lane, dealt in the order candidates are first named
        self._stream: Optional[torch.cuda.Stream] = None   # lanes beyond the first: their own stream
        self.num_classes = num_classes
        self.autocast_dtype = autocast_dtype
        self.kd = dict(temperature=kd_temperature, alpha=kd_alpha)
        self.device = next(model.parameters()).device
        self.importance = importance if importance is not None else MagnitudeImportance(p=2)
        helper = MagnitudePruner(model, importance=self.importance, ignored_layers=ignored_layers)
        self.groups = helper.prunable_groups()
        # kernel (1): importance of every group from two launches over the whole model
        self.group_importance = self.importance.batch(self.groups)
        self.candidates: List[Candidate] = []
        for gi, (g, imp) in enumerate(zip(self.groups, self.group_importance)):
            if imp is None:
                continue
            order = np.argsort(imp, kind="stable")
            seen = set()
            for r in ratios:
                n = int(r * g.size)
                if n <= 0 or n >= g.size or n in seen:
                    continue
                seen.add(n)
                self.candidates.append(Candidate(len(self.candidates), gi, float(r), np.sort(order[:n]),
                                                 self._weight_share_cost(g, n)))
        self.jobs: Dict[int, SliceJob] = {}
        # when set to a list, evaluate() appends a (start, end) HIP event pair around each KD-Lite launch
        self.plans_up_front = os.environ.get("SPICE_SWEEP_PLANS_UP_FRONT", "0") == "1"   # run(): every plan before the first candidate
        self.kd_events: Optional[list] = None
        #: the candidate's up-sampling, loss and arg-max histogram in one launch where the network's head allows
        #: it (SPICE_SWEEP_FUSED_HEAD=0: always the three separate launches)
        self.fused_head = os.environ.get("SPICE_SWEEP_FUSED_HEAD", "1") != "0"
        self.head_launches = 0
        #: with the one-launch tail: the teacher's softmax taken once per teacher tensor instead of once per candidate
        #: (SPICE_SWEEP_PREPARED_TEACHER=0: the head kernel reads the teacher's logits)
        self.prepared_teacher = os.environ.get("SPICE_SWEEP_PREPARED_TEACHER", "1") != "0"
        self._teacher_src, self._teacher_key, self._teacher_prepared = None, None, None
        self._geometry, self._geometry_for, self._warmed = {}, None, set()
        #: a candidate's forward starts at the first block its group touches, on the activations the un-pruned
        #: network produced in front of it (recorded by teacher_forward); SPICE_SWEEP_PREFIX_REUSE=0: every candidate
        #: forwards the whole network
        self.prefix_reuse = os.environ.get("SPICE_SWEEP_PREFIX_REUSE", "1") != "0"
        self._tape: Optional[PrefixTape] = None
        self._tape_src, self._tape_key = None, None
        self._first_unit: Dict[int, int] = {}
        self._last_unit: Dict[int, int] = {}
        self.units_reused = 0   # units whose recorded activations stood in for a launch, over all candidates so far
        #: the unchanged units BEHIND a candidate's group as one launch of a captured HIP graph (capture_suffix_graphs;
        #: SPICE_SWEEP_GRAPHS=0: never)
        self.suffix_graphs = os.environ.get("SPICE_SWEEP_GRAPHS", "1") != "0"
        #: SPICE_SWEEP_RUN_GRAPHS=1: run() captures them itself for a share of GRAPHS_FROM_CANDIDATES candidates and
        #: more.  Off by default: 2 ms per graph and lane against 0.4 ms of host time per candidate that goes through
        #: one is about even for ONE sweep of 414 candidates (0.09 s of capture, 0.41 -> 0.34 s of candidates); graphs
        #: the caller captured in an enclosing session() are used either way
        self.graphs_in_run = os.environ.get("SPICE_SWEEP_RUN_GRAPHS", "0") == "1"
        self._graphs: Dict[int, object] = {}   # first unit inside the graph -> _SuffixGraph
        self._runs_on: Dict[tuple, int] = {}   # batch geometry -> run() calls seen
        self._graphs_for = None                # what they were captured on
        self._graph_pool = None
        self._weights_version = 0
        self._session_depth = 0
        self.graph_launches = 0

    # -- cost of a candidate, for balancing ranks ---------------------------------------------
    #: share of a candidate's time that does not shrink with the network (host work per launch,
    #: the loss and histogram kernels on the full-size logits, the unchanged layers' memory-bound
    #: kernels).  A least-squares fit of time = a + b * (relative multiply-accumulates) to the event-timed
    #: candidates of the DeepLabV3+-ResNet-50 sweep on MI355X gives a = 1.13 ms, b = 0.66 ms, a / (a + b) = 0.63
    #: (residual 0.04 ms of 1.6-2.5 ms; it was 0.76 of 3.4-3.9 ms while the convolutions ran on MIOpen and the
    #: candidate's tail took three launches): tools/partition_balance.py
    FIXED_COST_SHARE = 0.63

    def _sliced_conv_dims(self, g, n: int) -> Dict[nn.Module, tuple]:
        """(in_channels, out_channels, groups) of every convolution coupled to group `g` once `n`
        of its channels are gone."""
        dims: Dict[nn.Module, list] = {}
        for m in g.members:
            mod = m.module
            if not isinstance(mod, nn.Conv2d):
                continue
            d = dims.setdefault(mod, [mod.in_channels, mod.out_channels, mod.groups])
            if m.kind == "out":
                d[1] -= n
            elif m.kind == "in":
                d[0] -= n
            elif m.kind == "dw":
                d[0] -= n
                d[1] -= n
                d[2] -= n
        return {k: tuple(v) for k, v in dims.items()}

    def _weight_share_cost(self, g, n: int) -> float:
        """Before any geometry is known: the convolutions' multiply-accumulates per output POSITION
        (= weight elements) that remain, as a share of the un-pruned network's."""
        if not hasattr(self, "_weights_total"):
            self._weights_total = float(sum(m.weight.numel() for m in self.model.modules() if isinstance(m, nn.Conv2d)))
        gone = 0.0
        for mod, (cin, cout, groups) in self._sliced_conv_dims(g, n).items():
            k = mod.kernel_size[0] * mod.kernel_size[1]
            gone += mod.weight.numel() - cout * (cin // max(groups, 1)) * k
        return self.FIXED_COST_SHARE + (1.0 - self.FIXED_COST_SHARE) * (1.0 - gone / max(self._weights_total, 1.0))

    def candidate_costs(self, images: torch.Tensor) -> List[float]:
        """Cost of every candidate = FIXED_COST_SHARE + (1 - FIXED_COST_SHARE) * (multiply-accumulates
        of the SLICED network's convolutions on `images`) / (those of the un-pruned network), the
        model tp.utils.count_ops_and_params applies to convolutions (pruning/pruning_core.py:192):
        out positions x out channels x in channels / groups x kernel area.  Also stored on the
        candidates.  Needs one forward of the un-pruned network (to learn every layer's geometry)."""
        self._ensure_geometry(images)
        macs: Dict[nn.Module, float] = {}
        for mod, ((n, _, h, w), _) in self._geometry.items():
            ho = (h + 2 * mod.padding[0] - mod.dilation[0] * (mod.kernel_size[0] - 1) - 1) // mod.stride[0] + 1
            wo = (w + 2 * mod.padding[1] - mod.dilation[1] * (mod.kernel_size[1] - 1) - 1) // mod.stride[1] + 1
            macs[mod] = float(n * ho * wo)   # output positions; times weight elements below
        total = sum(p * mod.weight.numel() for mod, p in macs.items())
        # with the prefix taken from the un-pruned network's forward, the units in front of the group's first one
        # cost nothing: their convolutions' multiply-accumulates, by unit
        tape = self._ensure_tape() if self.prefix_reuse else None
        before = None
        if tape is not None:
            names = {m: n for n, m in self.model.named_modules()}
            per_unit = [0.0] * (len(tape) + 1)
            for mod, p in macs.items():
                per_unit[tape.unit_of(names[mod])] += p * mod.weight.numel()
            before = np.concatenate([[0.0], np.cumsum(per_unit)])   # before[k] = work of the units in front of k
        out = []
        fixed = self.FIXED_COST_SHARE if tape is None else self.FIXED_COST_SHARE_PREFIX
        for cand in self.candidates:
            g = self.groups[cand.group]
            gone = 0.0 if before is None else float(before[self.first_unit(cand.index)])
            for mod, (cin, cout, groups) in self._sliced_conv_dims(g, len(cand.pruned)).items():
                if mod in macs:
                    k = mod.kernel_size[0] * mod.kernel_size[1]
                    gone += macs[mod] * (mod.weight.numel() - cout * (cin // max(groups, 1)) * k)
            cand.cost = fixed + (1.0 - fixed) * (1.0 - gone / max(total, 1.0))
            out.append(cand.cost)
        return out

    # -- the un-pruned network's activations in front of a candidate's first block ---------------------------
    #: FIXED_COST_SHARE when candidates start at their group's first block: the time of a candidate is then nearly
    #: proportional to the multiply-accumulates it still runs (fit on MI355X: 0.17 ms + 1.40 ms x relative
    #: multiply-accumulates, residual 0.08 ms of 0.16-1.9 ms): tools/partition_balance.py
    FIXED_COST_SHARE_PREFIX = 0.11
    GRAPHS_FROM_CANDIDATES = 150

    def _ensure_tape(self) -> Optional[PrefixTape]:
        """The tape of the primary network's units (None: the network names no units)."""
        if self._tape is None:
            units = (self.model.prefix_units() if hasattr(self.model, "prefix_units") and hasattr(self.model, "attach_tape")
                     else [])
            self._tape = PrefixTape(units) if units else None
        return self._tape

    def first_unit(self, index: int) -> int:
        """Index of the first unit (execution order) that runs a module of candidate `index`'s group: the units in
        front of it see the same inputs and weights as in the un-pruned network."""
        k = self._first_unit.get(index)
        if k is None:
            tape = self._ensure_tape()
            g = self.groups[self.candidates[index].group]
            k = 0 if tape is None else tape.first_touched(m.name for m in g.members)
            self._first_unit[index] = k
        return k

    def last_unit(self, index: int) -> int:
        """Index of the last unit that runs a module of candidate `index`'s group (the number of units when one of
        them lies behind the last unit): the units behind it run the un-pruned network's launches on new values."""
        k = self._last_unit.get(index)
        if k is None:
            tape = self._ensure_tape()
            g = self.groups[self.candidates[index].group]
            k = 0 if tape is None else tape.last_touched(m.name for m in g.members)
            self._last_unit[index] = k
        return k

    def teacher_forward(self, images: torch.Tensor) -> torch.Tensor:
        """The un-pruned network's logits on `images`; with prefix reuse on, the pass also records every unit's
        output, for the candidates evaluated on the same batch tensor afterwards."""
        tape = self._ensure_tape() if self.prefix_reuse else None
        if tape is None or not images.is_cuda:
            return self.forward(images)
        self.join_lanes()   # candidates still in flight on the lanes' streams read the activations dropped here
        self.model.attach_tape(tape.recorder())
        try:
            logits = self.forward(images)
        finally:
            self.model.attach_tape(None)
        # the tensor object is part of the key and is kept alive with the activations (an address alone can come
        # back from the allocator holding another batch)
        self._tape_src, self._tape_key = images, (images._version, tuple(images.shape))
        if not tape.complete():
            self._tape_src = None
        for lane in self._lanes:   # weights changed in place since the graphs were captured: their fp16 copies moved
            lane._weights_version = sum(int(p._version) for p in lane.model.parameters()) + \
                sum(int(b._version) for b in lane.model.buffers())
        return logits

    def _tape_for(self, images: torch.Tensor) -> Optional[PrefixTape]:
        if (self.prefix_reuse and self._tape is not None and self._tape_src is images
                and self._tape_key == (images._version, tuple(images.shape))):
            return self._tape
        return None

    def _ensure_geometry(self, images: torch.Tensor) -> None:
        """Input shape and dtype of every convolution for this batch geometry (one hooked forward of the
        un-pruned network; hooked modules run one by one on torch's kernels, so nothing else is kept from it)."""
        shape = tuple(images.shape)
        if self._geometry_for == shape:
            return
        self._geometry = {}
        hooks = [m.register_forward_hook(
            lambda mod, inp, out: self._geometry.__setitem__(mod, (tuple(inp[0].shape), inp[0].dtype)))
            for m in self.model.modules() if isinstance(m, nn.Conv2d)]
        try:
            with self.session():
                self.forward(images)
        finally:
            for h in hooks:
                h.remove()
        self._geometry_for = shape

    def may_need_solver_lookups(self, images: torch.Tensor) -> bool:
        """Whether a candidate of this sweep can bring MIOpen a convolution problem it has not seen: true as soon as
        one convolution a channel group touches goes to torch's conv2d on this batch geometry (depthwise ones aside:
        they do not go through MIOpen).  The library's own kernels take every channel count, so where they serve all
        those convolutions un-pruned they serve every candidate's too, and nothing needs looking up ahead."""
        self._ensure_geometry(images)
        for g in self.groups:
            for m in g.members:
                mod = m.module
                if not isinstance(mod, nn.Conv2d) or mod not in self._geometry or mod.groups == mod.in_channels > 1:
                    continue
                (_, _, h, w), in_dtype = self._geometry[mod]
                dt = self.autocast_dtype or in_dtype
                if self._served_inside_a_block(mod, dt):
                    continue
                if isinstance(mod, (Conv1x1, Conv1x1Bias, Conv3x3Bias, StemConv7x7)):
                    if mod.served_by_library(dt, h, w):
                        continue
                elif conv3x3_served_by_library(mod, dt, h, w):
                    continue
                return True
        return False

    def _served_inside_a_block(self, mod: nn.Module, dtype: torch.dtype) -> bool:
        """The two 1x1 convolutions of a squeeze-and-excite block never run as convolutions where the block's forward
        is the library's (nets/efficientnet.py::SqueezeExcite)."""
        owners = getattr(self, "_block_owner", None)
        if owners is None:
            owners = self._block_owner = {}
            for m in self.model.modules():
                if hasattr(m, "conv_reduce") and hasattr(m, "conv_expand") and hasattr(m, "served_by_library"):
                    owners[m.conv_reduce] = owners[m.conv_expand] = m
        owner = owners.get(mod)
        return owner is not None and owner.served_by_library(dtype)

    # -- host preparation for a set of candidates: plans + one shared index upload
    # -- lanes: further copies of the network, each with its own launch plans and its own stream -------------
    def enable_lanes(self, count: int) -> int:
        """Evaluate `count` candidates at a time: the candidates are dealt to `count` lanes in the order they are
        first named (prepare / evaluate), a lane being a copy of the network with launch plans of its own, whose
        launches go to a stream of its own.  One host thread feeds all of
        them; on the device the lanes' kernels fill each other's ramps, tails and half-empty launches (two
        processes sharing an MI355X evaluate 16 % more candidates per second than one; this is the same overlap
        without the second process).  Results are those of one lane: every candidate still runs the same kernels
        on the same values.  Returns the number of lanes in use (1 on a device without streams)."""
        count = max(int(count), 1)
        if self.device.type != "cuda" or count == 1:
            self._lanes = [self]
            return 1
        imp = self._ctor["importance"]
        names = {m: n for n, m in self.model.named_modules()}
        while len(self._lanes) < count:
            replica = copy.deepcopy(self.model)
            by_name = dict(replica.named_modules())
            ignored = self._ctor["ignored_layers"]
            lane = SensitivitySweep(replica, self._ctor["num_classes"], self._ctor["ratios"],
                                    importance=None if imp is None else copy.deepcopy(imp),
                                    ignored_layers=None if ignored is None else [by_name[names[m]] for m in ignored],
                                    autocast_dtype=self._ctor["autocast_dtype"],
                                    kd_temperature=self._ctor["kd_temperature"], kd_alpha=self._ctor["kd_alpha"])
            same = len(lane.candidates) == len(self.candidates) and all(
                a.group == b.group and np.array_equal(a.pruned, b.pruned) for a, b in zip(lane.candidates, self.candidates))
            if not same:
                raise RuntimeError("a copy of the network does not give the same candidates")
            lane._twin_of = {by_name[n]: m for m, n in names.items()}   # its modules -> this network's
            lane._stream = torch.cuda.Stream(device=self.device)
            lane._stream.wait_stream(torch.cuda.current_stream(self.device))
            self._lanes.append(lane)
        del self._lanes[count:]
        self._lane_assign = {}
        return len(self._lanes)

    def lane_index(self, index: int) -> int:
        """The lane candidate `index` belongs to (dealt round-robin at its first mention, so that a rank's share is
        spread evenly whatever its indices are)."""
        k = self._lane_assign.get(index)
        if k is None or k >= len(self._lanes):
            k = len(self._lane_assign) % len(self._lanes)
            self._lane_assign[index] = k
        return k

    def _lane_of(self, index: int) -> "SensitivitySweep":
        return self._lanes[self.lane_index(index)] if len(self._lanes) > 1 else self

    @contextlib.contextmanager
    def lane_context(self, index: int):
        """The stream candidate `index` is evaluated on, as the current stream (for what the caller launches around
        evaluate() on the candidate's own outputs, such as clearing its confusion matrix)."""
        lane = self._lane_of(index)
        if lane._stream is None:
            yield
        else:
            with torch.cuda.stream(lane._stream):
                yield

    def fork_lanes(self) -> None:
        """Everything queued on the current stream so far (inputs, teacher logits, cleared outputs) comes before what
        the other lanes launch from here on."""
        for lane in self._lanes[1:]:
            lane._stream.wait_stream(torch.cuda.current_stream(self.device))

    def join_lanes(self) -> None:
        """What the other lanes have launched so far comes before what the current stream launches from here on."""
        for lane in self._lanes[1:]:
            torch.cuda.current_stream(self.device).wait_stream(lane._stream)

    def prepare(self, indices: Sequence[int]) -> None:
        if len(self._lanes) > 1:
            for k, lane in enumerate(self._lanes):
                mine = [i for i in indices if self.lane_index(i) == k]
                if lane._stream is None:
                    lane._prepare(mine)
                else:
                    with torch.cuda.stream(lane._stream):   # the stream that will read the plans
                        lane._prepare(mine)
        else:
            self._prepare(indices)

    def _prepare(self, indices: Sequence[int]) -> None:
        todo = [i for i in indices if i not in self.jobs]
        jobs = [SliceJob([(self.groups[self.candidates[i].group], self.candidates[i].pruned)]) for i in todo]
        arrays, starts, pos = [], [], 0
        for j in jobs:
            starts.append(pos)
            for a in j.index_arrays():
                arrays.append(a)
                pos += len(a)
        # the plans first (host work only), then the index lists and every plan's work items as ONE asynchronous
        # copy on the current stream - the stream that will launch them
        for i, j, st in zip(todo, jobs, starts):
            self.jobs[i] = j.build(idx_device=_LATER, idx_start=st, with_grads=False, defer_upload=True)
        built = [j.plan for j in jobs if j.plan is not None]
        parts = [np.concatenate(arrays) if arrays else np.zeros(1, dtype=np.int64)]
        parts += [p.items_host.numpy() for p in built]
        dev = ops.upload_staged(parts, self.device)
        for j in jobs:
            j.idx = dev[0]
        for p, items in zip(built, dev[1:]):
            p.items, p.items_host = items, None

    @contextlib.contextmanager
    def session(self):
        """One autocast region around a whole run of candidates.  Autocast keeps the fp16 copies of
        fp32 parameters it has cast until its OUTERMOST region exits; the per-forward regions nest
        inside this one, so the ~70 unchanged convolution weights are cast once per sweep instead of
        once per candidate (the same fp16 values either way)."""
        self._session_depth += 1
        try:
            if self.autocast_dtype is None:
                yield self
            else:
                with torch.autocast("cuda", dtype=self.autocast_dtype):
                    yield self
        finally:
            self._session_depth -= 1
            if self._session_depth == 0:
                # captured graphs read the fp16 copies of weights that autocast made inside this region and frees
                # with it
                self.drop_suffix_graphs()

    def warm_shapes(self, images: torch.Tensor, indices: Sequence[int]) -> int:
        """Run one throw-away convolution for every NEW convolution problem (changed channel
        counts) the candidates in `indices` will bring, before any of them is evaluated.
        MIOpen's solver lookup for a problem it has not seen in this process synchronises the
        stream; inside the candidate loop that drains the queue of launches the host has built up
        (about 1.5 ms per candidate on MI355X), here nothing is queued yet (about 0.4 ms per problem).
        Doing the lookups on a helper thread instead does not pay: they wait for the device queue the
        evaluation loop keeps full, and the loop ends up waiting for the helper.  Returns the number
        of problems warmed."""
        if len(self._lanes) > 1:   # each lane for its own candidates (its plans name its own copy's modules)
            self._ensure_geometry(images)
            for lane in self._lanes[1:]:   # the copies see the same geometry: no forward of their own to learn it
                lane._geometry = {m: self._geometry[t] for m, t in lane._twin_of.items() if t in self._geometry}
                lane._geometry_for = self._geometry_for
            return sum(lane._warm_shapes(images, [i for i in indices if self.lane_index(i) == k])
                       for k, lane in enumerate(self._lanes))
        return self._warm_shapes(images, indices)

    def _warm_shapes(self, images: torch.Tensor, indices: Sequence[int]) -> int:
        if not indices:
            return 0
        self._ensure_geometry(images)
        warmed = 0
        for i in indices:
            if i not in self.jobs:
                self.prepare([i])
            warmed += self._warm_candidate(i)
        return warmed

    def _warm_candidate(self, i: int) -> int:
        warmed = 0
        for mod, upd in self.jobs[i].attr_updates:
            if not isinstance(mod, nn.Conv2d) or mod not in self._geometry:
                continue
            (n, _, h, w), in_dtype = self._geometry[mod]
            cin = upd.get("in_channels", mod.in_channels)
            cout = upd.get("out_channels", mod.out_channels)
            groups = upd.get("groups", mod.groups)
            key = (n, cin, h, w, cout, groups, mod.kernel_size, mod.stride, mod.padding, mod.dilation)
            if key in self._warmed or groups == cin:  # depthwise convolutions do not go through MIOpen
                continue
            if self._served_inside_a_block(mod, self.autocast_dtype or in_dtype):
                continue  # a squeeze-and-excite block's products: inside the library's own launches
            if isinstance(mod, (Conv1x1, Conv1x1Bias, Conv3x3Bias, StemConv7x7)) and mod.served_by_library(self.autocast_dtype or in_dtype, h, w):
                continue  # a plain matrix product: no solver lookup to get out of the way
            if conv3x3_served_by_library(mod, self.autocast_dtype or in_dtype, h, w):
                continue  # the library's own 3x3 kernel (the bottlenecks' convolutions): likewise
            self._warmed.add(key)
            dt = self.autocast_dtype or in_dtype
            x = torch.empty(n, cin, h, w, device=self.device, dtype=dt)
            wt = torch.empty(cout, cin // groups, *mod.kernel_size, device=self.device, dtype=dt)
            with torch.no_grad():
                F.conv2d(x, wt, None, mod.stride, mod.padding, mod.dilation, groups)
            warmed += 1
        return warmed

    def _network_input(self, images: torch.Tensor) -> torch.Tensor:
        """The batch as the first convolution reads it under autocast.  Every candidate sees the same batch, so the
        cast is made once per batch tensor (object and version), not once per candidate."""
        if (self.autocast_dtype is None or not images.is_floating_point() or images.dtype == self.autocast_dtype
                or not images.is_cuda):
            return images
        # keyed by the tensor object (kept alive with its cast) and its version: an address alone can come back from
        # the allocator holding another batch
        if getattr(self, "_cast_src", None) is not images or self._cast_key != images._version:
            old = getattr(self, "_cast", None)
            if old is not None and old.shape == images.shape and old.dtype == self.autocast_dtype and old.device == images.device:
                old.copy_(images)   # the same address for every batch of this geometry (captured graphs read it)
            else:
                self._cast = images.to(self.autocast_dtype)
            self._cast_src, self._cast_key = images, images._version
        return self._cast

    def forward(self, images: torch.Tensor) -> torch.Tensor:
        with torch.no_grad():
            if self.autocast_dtype is not None:
                with torch.autocast("cuda", dtype=self.autocast_dtype):
                    return self.model(self._network_input(images))
            return self.model(images)

    def evaluate(self, index: int, images: torch.Tensor, labels: torch.Tensor,
                 teacher_logits: Optional[torch.Tensor], conf_out: torch.Tensor,
                 kd_out: Optional[torch.Tensor] = None, first_lane: bool = False) -> int:
        """Enqueue one candidate; conf_out [C,C] int64 (+=), kd_out [3] fp32 (=).  Returns the
        algorithmic bytes of the gather launch.  first_lane: on this network and the current stream whatever lane
        the candidate belongs to (measurements that want one candidate on the device at a time).

        Ordering with lanes enabled (enable_lanes(k > 1)): a candidate of another lane is launched on that lane's
        stream, which is ordered behind the current stream only at fork_lanes().  The caller queues `images`,
        `labels`, `teacher_logits` and the cleared `conf_out` / `kd_out` on the current stream, calls fork_lanes(),
        then evaluate() as often as it likes, and calls join_lanes() before it reads the outputs or frees / rewrites
        the inputs on the current stream (run() and lane_context() do exactly this).  evaluate() does not order the
        lane's stream behind the current one itself: that would put every lane behind the first lane's candidates
        and undo the overlap the lanes exist for.  A new teacher tensor is the one exception: its prepared parts
        are made on the current stream between a join and a fork of their own."""
        lane = self if first_lane else self._lane_of(index)
        tape = self._tape_for(images)
        replay = None
        if tape is not None:
            first, behind = self.first_unit(index), self.last_unit(index) + 1
            graph = lane._graph_from(behind, tape, images) if (self.suffix_graphs and self._session_depth > 0) else None
            if graph is not None:
                replay = tape.replay(first, stop=behind, keep={})
                replay.graph = graph
            else:
                replay = tape.replay(first)
        prepared = None
        if self.fused_head and self.prepared_teacher and teacher_logits is not None and kd_out is not None:
            prepared = self._teacher_parts(teacher_logits)
        if lane is not self:
            lane.fused_head = self.fused_head
            with torch.cuda.stream(lane._stream):
                moved = lane._evaluate(index, images, labels, teacher_logits, conf_out, kd_out, self.kd_events, prepared, replay)
        else:
            moved = self._evaluate(index, images, labels, teacher_logits, conf_out, kd_out, self.kd_events, prepared, replay)
        if replay is not None:
            self.units_reused += replay.reused
        return moved

    # -- the unchanged units behind a candidate's group as one captured launch -------------------------------
    def capture_suffix_graphs(self, images: torch.Tensor) -> Dict[str, float]:
        """Capture, for every unit k the network names, the forward from unit k to the head's convolution as a HIP
        graph on the un-pruned network (every lane its own, on its own copy).  A candidate whose group touches units
        u .. k-1 then runs those eagerly on its sliced tensors and the rest as one graph launch: the host queues one
        launch where it queued twenty-odd, which is what bounds the sweep (the host, not the device, sets the pace of
        the candidate loop).  Needs teacher_forward(images) before (the graphs read the recorded activations of the
        units in front of them where the candidate leaves them unchanged) and an enclosing session(); the graphs hold
        until that session ends, the next teacher_forward, or drop_suffix_graphs().  Returns what it did."""
        import time
        info = {"graphs": 0, "seconds": 0.0, "lanes": len(self._lanes)}
        tape = self._tape_for(images)
        if (not self.suffix_graphs or tape is None or self._session_depth == 0 or not self.fused_head
                or getattr(self.model, "prefix_skip_units", lambda: None)() is None):
            return info
        t0 = time.perf_counter()
        self.join_lanes()
        torch.cuda.synchronize(self.device)
        # a network whose forward cannot be taken apart from its tail (a head of another form, hooks): nothing to capture
        if self._forward_to_head(images) is None:
            return info
        for lane in self._lanes:
            lane.fused_head = self.fused_head
            for k in range(1, len(tape)):
                if lane._capture_graph(k, tape, images) is not None:
                    info["graphs"] += 1
        torch.cuda.synchronize(self.device)
        self.fork_lanes()
        info["seconds"] = time.perf_counter() - t0
        return info

    def drop_suffix_graphs(self) -> None:
        for lane in self._lanes:
            lane._graphs, lane._graphs_for, lane._graph_pool = {}, None, None

    def _graph_from(self, k: int, tape: PrefixTape, images: torch.Tensor):
        """The captured graph that starts at unit k, if there is one and it was captured on these activations."""
        if not self._graphs or self._graphs_for != self._graph_key(tape, images):
            return None
        return self._graphs.get(k)

    def _graph_key(self, tape: PrefixTape, images: torch.Tensor):
        # the addresses the graphs hold: the recorded activations (their generation), the cast batch (its geometry),
        # the weights (their versions, as of the last teacher_forward)
        return (id(tape), getattr(tape, "generation", 0), tuple(images.shape), images.dtype, self._weights_version)

    def _capture_graph(self, k: int, tape: PrefixTape, images: torch.Tensor):
        key = self._graph_key(tape, images)
        if self._graphs_for != key:
            self._graphs, self._graphs_for = {}, key
        names = [n for n, _ in tape.units]
        skips = self.model.prefix_skip_units()
        # what the captured units read of the units in front of them: the output of the unit just before, and the
        # feature maps that left the encoder earlier
        inputs = [names[k - 1]] + [n for n in skips if tape.index[n] < k - 1]
        clone = lambda v: tuple(None if t is None else t.clone() for t in v) if isinstance(v, (tuple, list)) else v.clone()
        statics = {n: clone(tape.acts[n]) for n in inputs}
        stream = self._stream if self._stream is not None else torch.cuda.current_stream(self.device)
        side = torch.cuda.Stream(device=self.device)
        graph = torch.cuda.CUDAGraph()
        parts = None
        try:
            with torch.cuda.stream(stream):
                self.model.attach_tape(tape.replay(k, given=statics))
                try:
                    warm = self._forward_to_head(images)   # once eagerly: whatever the launches set up lazily exists before the capture
                    if warm is None:
                        return None   # (the forward cannot be taken apart from its tail: nothing to capture)
                    del warm
                    side.wait_stream(stream)
                    if self._graph_pool is None:
                        # one memory pool for all graphs of a lane: they never run at the same time, and each keeps
                        # its output tensor alive, so a later capture reuses only what earlier ones have let go of
                        self._graph_pool = torch.cuda.graph_pool_handle()
                    with torch.cuda.stream(side):
                        graph.capture_begin(pool=self._graph_pool)
                        try:
                            parts = self._forward_to_head(images)
                        finally:
                            graph.capture_end()
                    stream.wait_stream(side)
                finally:
                    self.model.attach_tape(None)
        except Exception as e:   # a launch that cannot be captured: this unit's candidates stay eager
            import warnings
            warnings.warn(f"no launch graph from unit {names[k]}: {e}")
            torch.cuda.synchronize(self.device)
            return None
        if parts is None:
            return None
        # (parts[1] is None where the head has no up-sampling: the graph ends in the logits, and the candidate's tail
        # is the three launches behind it)
        g = _SuffixGraph(graph, statics, inputs, parts[0], parts[1])
        g.recording = getattr(tape, "recordings", 0)
        self._graphs[k] = g
        return g

    def _teacher_parts(self, teacher_logits: torch.Tensor):
        """(probabilities at the distillation temperature, per-pixel sum p ln p) of the teacher's logits, taken once
        per teacher tensor (object and version) on the current stream: every candidate of a sweep is compared
        with the same un-pruned network.  None when the library does not cover the layout."""
        # the tensor object itself is part of the key and is kept alive with the result: an address alone can come
        # back from the allocator holding another teacher's logits
        key = (teacher_logits._version, float(self.kd["temperature"]))
        if self._teacher_src is not teacher_logits or self._teacher_key != key:
            if self._teacher_prepared is not None:
                self.join_lanes()   # head launches still queued on the lanes' streams read the parts released here
            self._teacher_src, self._teacher_key, self._teacher_prepared = teacher_logits, key, None
            if teacher_logits.is_cuda and teacher_logits.dtype == torch.float32 and teacher_logits.is_contiguous():
                self._teacher_prepared = ops.kd_teacher_prepare(teacher_logits, self.kd["temperature"])
                self.fork_lanes()   # the other lanes' streams wait for it
        return self._teacher_prepared

    def _evaluate(self, index: int, images: torch.Tensor, labels: torch.Tensor, teacher_logits: Optional[torch.Tensor],
                  conf_out: torch.Tensor, kd_out: Optional[torch.Tensor], kd_events: Optional[list],
                  prepared=None, replay=None) -> int:
        if index not in self.jobs:
            self._prepare([index])
        job = self.jobs[index]
        moved = job.run(keep_undo=True)
        if replay is not None:
            # the units in front of the group's first one return what the un-pruned network produced there
            self.model.attach_tape(replay)
        try:
            with_kd = teacher_logits is not None and kd_out is not None
            e0 = e1 = None
            if with_kd and kd_events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            graph = getattr(replay, "graph", None) if replay is not None else None
            if graph is not None and not (with_kd and self.fused_head):
                raise RuntimeError("a pass that stops in front of a captured graph needs the one-launch tail")
            if graph is not None:
                # the units the group touches, then the unchanged rest of the network as ONE launch of the graph
                # captured on the un-pruned network (the same kernels on the same shapes, reading this candidate's
                # values from the graph's own input buffers)
                try:
                    self._forward_to_head(images)
                    raise RuntimeError("the forward did not stop in front of the captured units")
                except StopForward:
                    pass
                parts = graph.launch(replay.keep, replay.tape)
                self.graph_launches += 1
            else:
                parts = self._forward_to_head(images) if (with_kd and self.fused_head) else None
            logits = None
            if parts is not None and parts[1] is None:
                logits, parts = parts[0].contiguous(), None
            if parts is not None:
                # up-sampling, loss and arg-max histogram in one launch over the head's small fp16 output
                low, up = parts
                if e0 is not None:
                    e0.record()
                direct = kd_out if (kd_out.is_contiguous() and kd_out.dtype == torch.float32) else None
                if prepared is not None:
                    out3 = ops.sweep_head(low, prepared[0], labels, conf_out, align_corners=True, out=direct,
                                          scale_factor=float(up.scale_factor), teacher_hneg=prepared[1], **self.kd)
                else:
                    out3 = ops.sweep_head(low, teacher_logits, labels, conf_out, align_corners=True, out=direct,
                                          scale_factor=float(up.scale_factor), **self.kd)
                if out3 is None:   # a shape the one-launch kernel does not cover
                    with torch.no_grad(), torch.autocast("cuda", dtype=self.autocast_dtype):
                        logits = up(low).contiguous()
                else:
                    if e1 is not None:
                        e1.record()
                        kd_events.append((e0, e1))
                    if direct is None:
                        kd_out.copy_(out3)
                    self.head_launches += 1
            elif logits is None:
                logits = self.forward(images).contiguous()
            if logits is not None:
                ops.argmax_confusion_update(conf_out, logits, labels, self.num_classes)
                if with_kd:
                    student = logits.float()
                    if e0 is not None:
                        e0.record()
                    out3, _, _ = ops.kdlite_forward(student, teacher_logits, labels, **self.kd)
                    if e1 is not None:
                        e1.record()
                        kd_events.append((e0, e1))
                    kd_out.copy_(out3)
        finally:
            if replay is not None:
                self.model.attach_tape(None)
            job.undo()
        return moved

    def _forward_to_head(self, images: torch.Tensor):
        """SegmentationModel.forward_to_head on the batch under this sweep's fp16 autocast: (low, up), (logits, None),
        or None when the network cannot stop at its head's convolution (the caller then runs the whole forward)."""
        m = self.model
        if self.autocast_dtype != torch.float16 or not images.is_cuda or not hasattr(m, "forward_to_head"):
            return None
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            return m.forward_to_head(self._network_input(images))

    def run(self, images: torch.Tensor, labels: torch.Tensor, indices: Optional[Sequence[int]] = None,
            with_kd: bool = True) -> Dict[str, torch.Tensor]:
        """Evaluate this rank's share of `indices` (default: all) and combine over ranks.
        Returns {"confusion": [n,C,C] int64, "kd": [n,3] fp32, "miou": [n] fp64} (device tensors)."""
        n = len(self.candidates)
        indices = list(range(n)) if indices is None else list(indices)
        import time
        rank, world = sdist.world()

        def dev_sync():
            if self.device.type == "cuda":
                torch.cuda.synchronize(self.device)

        ph: Dict[str, float] = {}
        t = time.perf_counter()
        all_costs = self.candidate_costs(images)
        costs = [all_costs[i] for i in indices]
        parts = sdist.partition_by_cost(costs, world)
        mine = [indices[k] for k in parts[rank]]
        dev_sync()
        ph["geometry_and_costs_s"], t = time.perf_counter() - t, time.perf_counter()
        # Launch plans: made in front of each candidate's launches (evaluate does that for a candidate it has no plan
        # for), while the device works on the candidates before - their bytes travel as a launch, so the host is not
        # held.  Only where a candidate may bring MIOpen a new convolution problem are plans and look-ups made up
        # front: a look-up synchronises the stream, which is cheap while nothing is queued.
        # (without captured graphs for the unchanged units the host, not the device, sets the pace of the loop: plans
        # made in a row up front then cost less than plans made between the launches)
        # Captured graphs for the unchanged units: a single sweep gets their capture time back about evenly, so run()
        # captures them only on request (graphs_in_run) - or from the SECOND run on a geometry inside a session the
        # caller holds open (an analysis over several batches: the graphs live until that session ends, and the first
        # run has shown that more are coming).
        shape = tuple(images.shape)
        self._runs_on[shape] = self._runs_on.get(shape, 0) + 1
        have = any(l._graphs for l in self._lanes)
        capture = self.suffix_graphs and not have and len(mine) >= self.GRAPHS_FROM_CANDIDATES and (
            self.graphs_in_run or (self._session_depth > 0 and self._runs_on[shape] >= 2))
        graphs = self.suffix_graphs and (have or capture)
        if self.plans_up_front or not graphs or self.may_need_solver_lookups(images):
            self.prepare(mine)
            dev_sync()
            ph["plans_s"], t = time.perf_counter() - t, time.perf_counter()
            self.warm_shapes(images, mine)
            dev_sync()
            ph["solver_lookups_s"], t = time.perf_counter() - t, time.perf_counter()
        else:
            ph["plans_s"] = ph["solver_lookups_s"] = 0.0
            t = time.perf_counter()
        self.last_share = {"candidates": len(mine), "cost": float(sum(all_costs[i] for i in mine)), "phases": ph}
        c = self.num_classes
        table = torch.zeros(n, c, c, dtype=torch.int64, device=self.device)
        kd = torch.zeros(n, 3, dtype=torch.float32, device=self.device)
        t0 = time.perf_counter()
        with self.session():
            teacher = None
            # the un-pruned network's forward: the teacher's logits, and the activations the candidates start from
            unpruned = self.teacher_forward(images)
            if with_kd:
                teacher = unpruned.float().contiguous()
            del unpruned
            dev_sync()
            ph["teacher_forward_s"], t = time.perf_counter() - t, time.perf_counter()
            if capture:
                self.capture_suffix_graphs(images)
                dev_sync()
                ph["graph_capture_s"], t = time.perf_counter() - t, time.perf_counter()
            self.fork_lanes()
            for lane in self._lanes[1:]:
                # one forward of the un-pruned copy on the lane's stream: its allocator pool and its fp16 copies of the
                # weights exist before the first candidate needs them (a first allocation stalls every stream)
                with torch.cuda.stream(lane._stream):
                    lane.forward(images)
            for i in mine:   # no communication before the final all-reduce
                self.evaluate(i, images, labels, teacher, table[i], kd[i] if with_kd else None)
            self.join_lanes()
            dev_sync()   # this rank's own work, for the busy / idle split
            ph["candidates_s"] = time.perf_counter() - t
            self.last_share["busy_s"] = time.perf_counter() - t0
            for tb in ([table, kd] if with_kd else [table]):
                sdist.all_reduce_sum_(tb)
        return {"confusion": table, "kd": kd, "miou": miou_from_confusion(table)}


class _SuffixGraph:
    """A captured forward from some unit to the head's convolution, its input buffers and its output."""

    def __init__(self, graph, statics, inputs, low, up):
        self.graph, self.statics, self.inputs, self.low, self.up = graph, statics, inputs, low, up
        self.foreign = set()   # inputs whose buffer holds a candidate's values, not the un-pruned network's
        self.recording = None  # the recording of the un-pruned network the other buffers hold

    @staticmethod
    def _copy(dst, src) -> bool:
        """False (nothing copied) where a shape differs: the candidate took channels out of this tensor, so every
        module that reads it belongs to the candidate's group and ran eagerly - the captured units do not read it."""
        pairs = list(zip(dst, src)) if isinstance(dst, (tuple, list)) else [(dst, src)]
        if any(d is not None and (v is None or d.shape != v.shape) for d, v in pairs):
            return False
        for d, v in pairs:
            if d is not None and d is not v:   # (a unit that wrote into the buffer itself hands the buffer back)
                d.copy_(v)
        return True

    def launch(self, kept: Dict[str, object], tape: PrefixTape):
        """Replay on the current stream with the outputs `kept` of the units a candidate ran itself; the other
        inputs are the un-pruned network's (copied back only where the last candidate left its own)."""
        if self.recording != getattr(tape, "recordings", 0):   # another batch since: every buffer is out of date
            self.foreign = set(self.inputs)
            self.recording = getattr(tape, "recordings", 0)
        for n in self.inputs:
            if n in kept:
                if self._copy(self.statics[n], kept[n]):
                    self.foreign.add(n)
            elif n in self.foreign:
                self._copy(self.statics[n], tape.acts[n])
                self.foreign.discard(n)
        self.graph.replay()
        return self.low, self.up


def miou_from_confusion(table: torch.Tensor) -> torch.Tensor:
    """mean over ALL classes of diag / (row + col - diag + 1e-10), per table (the in-tree
    definition: baseline/eval/metrics.py:29-36), in fp64 on the device."""
    m = table.double()
    hit = torch.diagonal(m, dim1=-2, dim2=-1)
    union = m.sum(-1) + m.sum(-2) - hit
    return (hit / (union + 1e-10)).mean(-1)
