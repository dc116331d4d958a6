"""Per-problem cost of SensitivitySweep.warm_shapes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spice_amd import miopen_tuning
miopen_tuning.ensure_tuned("DeepLabV3Plus", "resnet50", 21, 8, 512, "fp16")
import torch, torch.nn.functional as F
from spice_amd import nets, synth
from spice_amd.sweep import DEFAULT_RATIOS, SensitivitySweep
dev = torch.device("cuda:0"); C = 21
torch.manual_seed(1234)
model = nets.create_model("DeepLabV3Plus", "resnet50", C).to(dev).eval()
img = torch.from_numpy(synth.normal(1234, 50, 8 * 3 * 512 * 512).reshape(8, 3, 512, 512)).to(dev)
ignored = [m for m in model.segmentation_head.modules() if isinstance(m, torch.nn.Conv2d)]
sweep = SensitivitySweep(model, C, DEFAULT_RATIOS, ignored_layers=ignored)
todo = list(range(0, len(sweep.candidates), 7))
sweep.prepare(todo)
sweep.warm_shapes(img, [])   # geometry pass only
torch.cuda.synchronize()
orig = F.conv2d
rows = []
def timed(x, w, *a, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter(); y = orig(x, w, *a, **k); t1 = time.perf_counter()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    y2 = orig(x, w, *a, **k); torch.cuda.synchronize(); t3 = time.perf_counter()
    rows.append((1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), tuple(x.shape), tuple(w.shape), a[1:]))
    return y
import spice_amd.sweep as sw
sw.F.conv2d = timed
t0 = time.perf_counter(); n = sweep.warm_shapes(img, todo); torch.cuda.synchronize()
print("warmed", n, "in %.3f s" % (time.perf_counter() - t0))
rows.sort(reverse=True)
for r in rows[:12] + rows[-5:]:
    print("host %.2f ms, drain %.2f ms, second call %.2f ms  x%s w%s %s" % r)
print("sum host %.1f ms, sum drain %.1f ms" % (sum(r[0] for r in rows), sum(r[1] for r in rows)))
