"""cProfile of warm_shapes inside the bench's whole-sweep block."""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from spice_amd.sweep import SensitivitySweep
orig = SensitivitySweep.warm_shapes
def wrapped(self, images, indices):
    if len(indices) > 100:
        pr = cProfile.Profile(); pr.enable()
        r = orig(self, images, indices)
        pr.disable(); pstats.Stats(pr, stream=sys.stderr).sort_stats("cumtime").print_stats(30)
        return r
    return orig(self, images, indices)
SensitivitySweep.warm_shapes = wrapped
sys.argv = ["bench.py", "--no-kernel-table", "--no-cpu-baseline"]
bench.main()
