"""Host time per leaf module in the first forward of a few sweep candidates: which modules pay for
a new shape?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spice_amd import miopen_tuning
miopen_tuning.ensure_tuned("DeepLabV3Plus", "resnet50", 21, 8, 512, "fp16")
import torch
from spice_amd import nets, synth
from spice_amd.sweep import DEFAULT_RATIOS, SensitivitySweep
dev = torch.device("cuda:0"); C = 21
torch.manual_seed(1234)
model = nets.create_model("DeepLabV3Plus", "resnet50", C).to(dev).eval()
img = torch.from_numpy(synth.normal(1234, 50, 8 * 3 * 512 * 512).reshape(8, 3, 512, 512)).to(dev)
ignored = [m for m in model.segmentation_head.modules() if isinstance(m, torch.nn.Conv2d)]
sweep = SensitivitySweep(model, C, DEFAULT_RATIOS, ignored_layers=ignored)
names = {m: n for n, m in model.named_modules()}
t_in, acc = {}, {}
def pre(m, i): t_in[m] = time.perf_counter()
def post(m, i, o): acc[m] = acc.get(m, 0.0) + time.perf_counter() - t_in[m]
for m in model.modules():
    if not list(m.children()):
        m.register_forward_pre_hook(pre); m.register_forward_hook(post)
todo = [3, 47, 101, 160, 222, 290, 333, 400]
sweep.prepare(todo)
with sweep.session():
    for _ in range(3):
        sweep.forward(img)
    torch.cuda.synchronize(); acc.clear()
    t0 = time.perf_counter(); sweep.forward(img); base_total = time.perf_counter() - t0
    base = dict(acc); torch.cuda.synchronize()
    print("base forward host ms %.2f, sum of leaves %.2f" % (1e3 * base_total, 1e3 * sum(base.values())))
    for i in todo:
        job = sweep.jobs[i]; job.run(keep_undo=True); acc.clear()
        t0 = time.perf_counter(); sweep.forward(img); tot = time.perf_counter() - t0
        first = dict(acc); acc.clear()
        t0 = time.perf_counter(); sweep.forward(img); tot2 = time.perf_counter() - t0
        job.undo(); torch.cuda.synchronize()
        slow = sorted(((first[m] - base.get(m, 0), m) for m in first), key=lambda t: -t[0])[:6]
        cand = sweep.candidates[i]
        print("cand %d group %s ratio %.1f: first %.2f ms, second %.2f ms; extra: %s" % (
            i, sweep.groups[cand.group].root_name, cand.ratio, 1e3 * tot, 1e3 * tot2,
            ", ".join("%s(%s) +%.2f" % (names[m], type(m).__name__, 1e3 * d) for d, m in slow if d > 5e-5)), flush=True)
