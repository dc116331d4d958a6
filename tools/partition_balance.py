"""Balance of the cost-model partition, judged by MEASURED per-candidate times on one GPU.

Every candidate of the sweep is evaluated once with a HIP event pair around it (queue kept full, no
host synchronisation inside the loop); the partition for 2/4/8 ranks is then scored by the measured
time of each share: efficiency = mean share time / largest share time.  Also fits the fixed share of
the cost model (time = a + b * relative multiply-accumulates) to the measurement."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from spice_amd import nets, dist as sdist
from spice_amd.sweep import SensitivitySweep, DEFAULT_RATIOS

dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = nets.create_model("DeepLabV3Plus", "resnet50", bench.CLASSES).to(dev).eval()
images, labels = bench.build_inputs(dev)
ignored = [m for m in model.segmentation_head.modules() if isinstance(m, torch.nn.Conv2d)]
sw = SensitivitySweep(model, bench.CLASSES, DEFAULT_RATIOS, ignored_layers=ignored)
costs = np.array(sw.candidate_costs(images))
n = len(costs)
every = list(range(n))
sw.prepare(every)
sw.warm_shapes(images, every)
table = torch.zeros(n, bench.CLASSES, bench.CLASSES, dtype=torch.int64, device=dev)
kd = torch.zeros(n, 3, device=dev)
times = np.zeros((3, n))
with sw.session():
    teacher = sw.teacher_forward(images).float().contiguous()   # (records the activations the candidates start from)
    for rep in range(3):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
        torch.cuda.synchronize()
        ev[0].record()
        for i in every:
            sw.evaluate(i, images, labels, teacher, table[i], kd[i])
            ev[i + 1].record()
        torch.cuda.synchronize()
        times[rep] = [ev[i].elapsed_time(ev[i + 1]) for i in range(n)]
t = np.median(times[1:], axis=0)   # first repetition warms every shape
fs = SensitivitySweep.FIXED_COST_SHARE_PREFIX if sw.prefix_reuse and sw._tape is not None else SensitivitySweep.FIXED_COST_SHARE
rel_macs = (costs - fs) / (1.0 - fs)
A = np.stack([np.ones(n), rel_macs], 1)
(a, b), *_ = np.linalg.lstsq(A, t, rcond=None)
out = {"candidates": n, "ms_total": float(t.sum()), "ms_min": float(t.min()), "ms_max": float(t.max()),
       "fit_ms": {"fixed": float(a), "per_full_network_macs": float(b), "fixed_share": float(a / (a + b)),
                  "rms_residual_ms": float(np.sqrt(np.mean((A @ np.array([a, b]) - t) ** 2)))},
       "model_fixed_share": fs, "prefix_reuse": bool(sw.prefix_reuse and sw._tape is not None), "partitions": {}}
for world in (2, 4, 8):
    for name, c in (("cost_model", costs), ("measured_time_as_cost", t), ("equal_counts_round_robin", None)):
        parts = ([list(range(r, n, world)) for r in range(world)] if c is None else sdist.partition_by_cost(list(c), world))
        share = np.array([t[p].sum() for p in parts])
        out["partitions"][f"{world}_ranks_{name}"] = {"share_ms": [round(float(x), 1) for x in share],
                                                      "efficiency": round(float(share.mean() / share.max()), 4)}
# the same measurement scored under other fixed shares of the cost model
out["efficiency_by_fixed_share"] = {}
for share in (0.0, 0.1, 0.2, 0.3, 0.45, 0.6, 0.8):
    c = share + (1.0 - share) * rel_macs
    out["efficiency_by_fixed_share"][str(share)] = {
        str(world): round(float(np.mean([t[p].sum() for p in sdist.partition_by_cost(list(c), world)]) /
                                np.max([t[p].sum() for p in sdist.partition_by_cost(list(c), world)])), 4)
        for world in (2, 4, 8)}
print(json.dumps(out, indent=1))
