"""Host (enqueue) time of one forward of the un-pruned DeepLabV3+-R50 under the sweep's autocast
session, by leaf module type, and of one whole sweep step."""
import os, sys, time, collections
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
lab = torch.from_numpy(synth.label_map(1234, 8, 512, 512, C)).to(dev)
ignored = [m for m in model.segmentation_head.modules() if isinstance(m, torch.nn.Conv2d)]
sweep = SensitivitySweep(model, C, DEFAULT_RATIOS, ignored_layers=ignored)
t_in, acc, cnt = {}, collections.Counter(), collections.Counter()
def pre(m, i): t_in[m] = time.perf_counter()
def post(m, i, o):
    acc[type(m).__name__] += time.perf_counter() - t_in[m]; cnt[type(m).__name__] += 1
hooks = []
for m in model.modules():
    if not list(m.children()):
        hooks += [m.register_forward_pre_hook(pre), m.register_forward_hook(post)]
with sweep.session():
    for _ in range(5):
        sweep.forward(img)
    torch.cuda.synchronize(); acc.clear(); cnt.clear()
    R = 20
    t0 = time.perf_counter()
    for _ in range(R):
        sweep.forward(img)
    tot = time.perf_counter() - t0
    torch.cuda.synchronize()
    print("hooked forward host ms %.3f; leaves %.3f" % (1e3 * tot / R, 1e3 * sum(acc.values()) / R))
    for k, v in acc.most_common():
        print("  %-22s %5.1f calls  %.3f ms  (%.1f us each)" % (k, cnt[k] / R, 1e3 * v / R, 1e6 * v / cnt[k]))
    for h in hooks:
        h.remove()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(R):
        sweep.forward(img)
    tot = time.perf_counter() - t0
    torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    print("plain forward: host enqueue %.3f ms, with final sync %.3f ms" % (1e3 * tot / R, 1e3 * t1 / R))
    # whole steps
    idx = list(range(100, 140))
    sweep.prepare(idx); sweep.warm_shapes(img, idx)
    teacher = sweep.forward(img).float().contiguous()
    table = torch.zeros(len(sweep.candidates), C, C, dtype=torch.int64, device=dev)
    kd = torch.zeros(len(sweep.candidates), 3, device=dev)
    for i in idx[:10]:
        sweep.evaluate(i, img, lab, teacher, table[i], kd[i])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in idx[10:]:
        sweep.evaluate(i, img, lab, teacher, table[i], kd[i])
    tot = time.perf_counter() - t0
    torch.cuda.synchronize(); t1 = time.perf_counter() - t0
    print("sweep step: host enqueue %.3f ms, wall %.3f ms" % (1e3 * tot / 30, 1e3 * t1 / 30))
