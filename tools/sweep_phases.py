"""Finer split of the replicated set-up of SensitivitySweep.run() (what every rank repeats before its share)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from spice_amd import nets, dist as sdist
from spice_amd.sweep import SensitivitySweep, DEFAULT_RATIOS

dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = nets.create_model("DeepLabV3Plus", "resnet50", bench.CLASSES).to(dev).eval()
images, labels = bench.build_inputs(dev)
ignored = [m for m in model.segmentation_head.modules() if isinstance(m, torch.nn.Conv2d)]

def sync():
    torch.cuda.synchronize()

def fresh():
    return SensitivitySweep(model, bench.CLASSES, DEFAULT_RATIOS, ignored_layers=ignored)

sw = fresh()
with sw.session():
    sw.forward(images)
sync()
for rep in range(2):
    sw = fresh()
    sync(); t = time.perf_counter()
    sw._ensure_geometry(images); sync()
    t1 = time.perf_counter()
    costs = sw.candidate_costs(images)
    t2 = time.perf_counter()
    parts = sdist.partition_by_cost(costs, 8)
    t3 = time.perf_counter()
    mine = parts[0]
    sw.prepare(mine); sync()
    t4 = time.perf_counter()
    sw.warm_shapes(images, mine); sync()
    t5 = time.perf_counter()
    print(f"rep {rep}: hooked forward {1e3*(t1-t):.1f} ms, costs {1e3*(t2-t1):.1f} ms, partition(8) {1e3*(t3-t2):.1f} ms, "
          f"plans for {len(mine)} candidates {1e3*(t4-t3):.1f} ms, solver lookups {1e3*(t5-t4):.1f} ms ({len(sw._warmed)} problems)", flush=True)
sw = fresh()
sync()
prof = cProfile.Profile(); prof.enable()
sw._ensure_geometry(images); sync()
sw.candidate_costs(images)
prof.disable()
pstats.Stats(prof).sort_stats("cumulative").print_stats(25)
t = time.perf_counter(); s2 = fresh(); sync(); print(f"SensitivitySweep construction {1e3*(time.perf_counter()-t):.1f} ms")
