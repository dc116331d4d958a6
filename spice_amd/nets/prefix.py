"""Activations of a network at unit boundaries, kept from one forward and handed back in later ones.

A sensitivity-sweep candidate slices ONE channel group; every layer in front of the first module that group
touches computes, from the same batch, what the un-pruned network computed there, bit for bit (the same kernels on
the same values).  A network that supports it names its *units* - blocks in execution order, each a function of
the units before it (`prefix_units()`) - and routes each unit through the tape that is attached to it for the
duration of a forward:

    recording   every unit runs; its output is kept            (the un-pruned network's forward of a sweep)
    replaying   units in front of `first` return what was kept; the others run   (a candidate's forward)

With no tape attached the forward is the plain one.  Unit outputs must never be written in place by a later unit.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Sequence, Tuple


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
        self.acts = {}
        return _Pass(self, 0, True)

    def replay(self, first: int) -> "_Pass":
        """A pass in which the units in front of index `first` return the recorded activations."""
        return _Pass(self, first, False)


class _Pass:
    """What a network's forward sees: run(unit name, function, *arguments)."""

    __slots__ = ("tape", "first", "recording", "reused")

    def __init__(self, tape: PrefixTape, first: int, recording: bool):
        self.tape, self.first, self.recording, self.reused = tape, first, recording, 0

    def skips(self, name: str) -> bool:
        return not self.recording and self.tape.index[name] < self.first

    def run(self, name: str, fn: Callable, *args):
        tape = self.tape
        if not self.recording and tape.index[name] < self.first:
            self.reused += 1
            return tape.acts[name]
        out = fn(*args)
        if self.recording:
            tape.acts[name] = out
        return out
