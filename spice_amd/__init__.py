"""MI355X-native pruning hot path (importance, channel gather, KD-Lite loss, mIoU confusion).

Sub-modules:
  spice_amd.ops       tensor-level entry points over the C ABI (libspice_amd.so)
  spice_amd.pruning   drop-in mirror of the reference toolkit's `pruning` package API
"""
__version__ = "0.1.0"

import os as _os

# MIOpen runs the backbone convolutions.  Pruning produces many convolution shapes that are each
# used only a few times (every sweep candidate, every pruning step); MIOpen's default find mode
# benchmarks solvers the first time it meets a shape (about 35 ms per shape on MI355X), its FAST
# mode picks by heuristic in well under 2 ms at the same steady-state speed.  Respect an
# explicit user setting.
_os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
_os.environ.setdefault("MIOPEN_LOG_LEVEL", "3")  # heuristic-fallback warnings per new shape are expected
