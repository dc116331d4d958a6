"""Which convolution problems of the whole sweep still go to MIOpen (and what their look-ups cost)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from spice_amd import nets, miopen_tuning
from spice_amd.sweep import SensitivitySweep, DEFAULT_RATIOS

miopen_tuning.ensure_tuned("DeepLabV3Plus", "resnet50", bench.CLASSES, bench.BATCH, bench.SIZE, "fp16", device=0)
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = nets.create_model("DeepLabV3Plus", "resnet50", bench.CLASSES).to(dev).eval()
images, labels = bench.build_inputs(dev)
ignored = [m for m in model.segmentation_head.modules() if isinstance(m, torch.nn.Conv2d)]
sw = SensitivitySweep(model, bench.CLASSES, DEFAULT_RATIOS, ignored_layers=ignored, autocast_dtype=torch.float16)
idx = list(range(len(sw.candidates)))
sw.candidate_costs(images)
sw.prepare(idx)
names = {m: n for n, m in model.named_modules()}
with torch.autocast("cuda", dtype=torch.float16):
    for i in idx:
        before = set(sw._warmed)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        k = sw._warm_candidate(i)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        if k:
            c = sw.candidates[i]
            print(i, "group", c.group, "pruned", len(c.pruned), "lookups", k, f"{1e3*dt:.1f} ms", sorted(set(sw._warmed) - before),
                  [names.get(m) for m, _ in sw.jobs[i].attr_updates])
