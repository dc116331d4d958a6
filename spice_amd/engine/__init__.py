"""Structured-pruning engine: coupled-channel analysis, importance criteria and the pruner
object whose `.step()` the reference obtains from torch-pruning 1.6.0 (absent here).
All tensor arithmetic goes through the HIP kernels in spice_amd.ops."""
from .depgraph import ChannelGroup, Member, build_groups  # noqa: F401
from .importance import (Importance, LAMPImportance, MagnitudeImportance, RandomImportance,  # noqa: F401
                         TaylorImportance)
from .pruner import MagnitudePruner, count_ops_and_params  # noqa: F401
