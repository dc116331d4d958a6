"""Where the host time of in-loop plan-making goes: accumulated wall time of the pieces of a prepare call."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from spice_amd import ops
from spice_amd.engine import pruner
from spice_amd import sweep as sw

acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
def wrap(owner, name, label):
    orig = getattr(owner, name)
    def f(*a, **k):
        t = time.perf_counter()
        try:
            return orig(*a, **k)
        finally:
            d = time.perf_counter() - t
            e = acc[label]; e[0] += 1; e[1] += d; e[2] = max(e[2], d)
    setattr(owner, name, f)
wrap(sw.SensitivitySweep, "_prepare", "_prepare")
wrap(sw.SensitivitySweep, "prepare", "prepare")
wrap(ops, "upload_staged", "upload_staged")
wrap(pruner.SliceJob, "build", "SliceJob.build")
wrap(pruner.SliceJob, "__init__", "SliceJob.__init__")
wrap(ops.GatherPlan, "__init__", "GatherPlan.__init__")
wrap(ops._StagingRing, "take", "ring.take")
wrap(ops, "_device_bytes", "device_bytes")
wrap(ops, "_queue_copy", "queue_copy")
wrap(sw.SensitivitySweep, "evaluate", "evaluate")
import gc
if os.environ.get("NOGC"):
    gc.disable()
bench.main()
for k, (n, tot, mx) in acc.items():
    print(f"{k:24s} calls {n:5d} total {1e3*tot:8.2f} ms  mean {1e6*tot/max(n,1):8.1f} us  max {1e3*mx:7.2f} ms", file=sys.stderr)
