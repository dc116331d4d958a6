"""Activations of a network at unit boundaries, kept from one forward and handed back in later ones.

A sensitivity-sweep candidate slices ONE channel group; every layer in front of the first module that group
touches computes, from the same batch, what the un-pruned network computed there, bit for bit (the same kernels on
the same values).  A network that supports it names its *units* - blocks in execution order, each a function of
the units before it (`prefix_units()`) - and routes each unit through the tape that is attached to it for the
duration of a forward:

    recording   every unit runs; its output is kept            (the un-pruned network's forward of a sweep)
    replaying   units in front of `first` return what was kept; the others run   (a candidate's forward)

    replaying up to a stop   the same, and the forward ends (StopForward) where unit `stop` would start: the caller
                             has the outputs of the units in between (`keep`) and continues elsewhere - a captured
                             launch graph of the unchanged units behind them, which reads its inputs from buffers of
                             its own (`given`: what stands in for the recorded output of a unit in front of `first`)

With no tape attached the forward is the plain one.  Unit outputs must never be written in place by a later unit.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple


import os

# SPICE_UNIT_DEST=0: units never write into a captured graph's input buffers themselves (A/B measurements)
_DIRECT = os.environ.get("SPICE_UNIT_DEST", "1") != "0"


class StopForward(Exception):
    """Raised by a pass at the unit it was told to stop in front of."""


class PrefixTape:
    def __init__(self, units: Sequence[Tuple[str, Sequence[str]]]):
        """`units`: (unit name, prefixes of the names of the modules it runs), in execution order."""
        self.units: List[Tuple[str, Tuple[str, ...]]] = [(n, tuple(p)) for n, p in units]
        self.index: Dict[str, int] = {n: i for i, (n, _) in enumerate(self.units)}
        self.acts: Dict[str, object] = {}
        self.key = None   # what the activations were recorded from (the owner's business)

    def __len__(self) -> int:
        return len(self.units)

    def unit_of(self, module_name: str) -> int:
        """Index of the unit that runs the module of that name; len(self) for modules behind the last unit."""
        for i, (_, prefixes) in enumerate(self.units):
            for p in prefixes:
                if module_name == p or module_name.startswith(p + "."):
                    return i
        return len(self.units)

    def first_touched(self, module_names) -> int:
        """The first unit (in execution order) that runs any of the named modules."""
        return min((self.unit_of(n) for n in module_names), default=len(self.units))

    def last_touched(self, module_names) -> int:
        """The last unit that runs any of the named modules (len(self) when one of them lies behind the last unit)."""
        return max((self.unit_of(n) for n in module_names), default=len(self.units))

    def complete(self) -> bool:
        return len(self.acts) == len(self.units)

    def nbytes(self) -> int:
        total = 0
        for v in self.acts.values():
            for t in (v if isinstance(v, (tuple, list)) else (v,)):
                if t is not None:
                    total += t.numel() * t.element_size()
        return total

    def recorder(self) -> "_Pass":
        """A recording pass.  Where a unit's output has the layout of the one recorded before, the new values are
        copied into the old tensor, so that the addresses of the recorded activations outlive the batch (launch graphs
        captured on them stay valid); `generation` counts the times an address changed."""
        self._old, self.acts = self.acts, {}
        self.recordings = getattr(self, "recordings", 0) + 1
        return _Pass(self, 0, True)

    def _store(self, name: str, out) -> None:
        old = getattr(self, "_old", {}).pop(name, None)
        a = out if isinstance(out, (tuple, list)) else (out,)
        b = old if isinstance(old, (tuple, list)) else (old,)
        same = old is not None and len(a) == len(b) and all(
            (x is None and y is None) or (x is not None and y is not None and x.shape == y.shape and x.dtype == y.dtype
                                          and x.device == y.device and y.is_contiguous() == x.is_contiguous())
            for x, y in zip(a, b))
        if same:
            for x, y in zip(a, b):
                if y is not None:
                    y.copy_(x)
            self.acts[name] = old
        else:
            self.acts[name] = out
            self.generation = getattr(self, "generation", 0) + 1

    def replay(self, first: int, stop: Optional[int] = None, given: Optional[Dict[str, object]] = None,
               keep: Optional[Dict[str, object]] = None) -> "_Pass":
        """A pass in which the units in front of index `first` return the recorded activations (`given[name]` where
        the caller supplies one); with `stop`, the forward ends with StopForward in front of unit `stop`, the outputs
        of the units that ran left in `keep`."""
        p = _Pass(self, first, False)
        p.stop, p.given, p.keep = stop, given, keep
        return p


class _Pass:
    """What a network's forward sees: run(unit name, function, *arguments)."""

    __slots__ = ("tape", "first", "recording", "reused", "stop", "given", "keep", "graph")

    def __init__(self, tape: PrefixTape, first: int, recording: bool):
        self.tape, self.first, self.recording, self.reused = tape, first, recording, 0
        self.stop = self.given = self.keep = self.graph = None

    def skips(self, name: str) -> bool:
        return not self.recording and self.tape.index[name] < self.first

    def run(self, name: str, fn: Callable, *args):
        tape = self.tape
        k = tape.index[name]
        if not self.recording and k < self.first:
            self.reused += 1
            if self.given is not None and name in self.given:
                return self.given[name]
            return tape.acts[name]
        if self.stop is not None and k >= self.stop:
            raise StopForward(name)
        dest = None
        if _DIRECT and self.graph is not None and self.keep is not None:
            dest = self.graph.statics.get(name)
        if dest is not None and not isinstance(dest, (tuple, list)):
            from . import layers
            layers.set_unit_destination(dest)   # the unit may write its result into the graph's input buffer
            try:
                out = fn(*args)
            finally:
                layers.set_unit_destination(None)
        else:
            out = fn(*args)
        if self.recording:
            tape._store(name, out)
        if self.keep is not None:
            self.keep[name] = out
        return out
