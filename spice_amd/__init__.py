"""MI355X-native pruning hot path (importance, channel gather, KD-Lite loss, mIoU confusion).

Sub-modules:
  spice_amd.ops       tensor-level entry points over the C ABI (libspice_amd.so)
  spice_amd.pruning   drop-in mirror of the reference toolkit's `pruning` package API
"""
__version__ = "0.1.0"
