"""Does MIOpen re-read MIOPEN_FIND_MODE inside one process?  Phase 1 searches solvers for the
un-pruned model with NORMAL, phase 2 switches the variable to FAST and evaluates candidates."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MIOPEN_USER_DB_PATH"] = "/tmp/spice_miopen_switch_test"
os.makedirs(os.environ["MIOPEN_USER_DB_PATH"], exist_ok=True)
os.environ["MIOPEN_FIND_MODE"] = "NORMAL"
import torch
from spice_amd import nets, synth
from spice_amd.sweep import DEFAULT_RATIOS, SensitivitySweep
dev = torch.device("cuda:0"); C = 21
torch.manual_seed(1234)
model = nets.create_model("DeepLabV3Plus", "resnet50", C).to(dev).eval()
img = torch.from_numpy(synth.normal(1234, 50, 8 * 3 * 512 * 512).reshape(8, 3, 512, 512)).to(dev)
lab = torch.from_numpy(synth.label_map(1234, 8, 512, 512, C, "blocky", ignore_share=0.05)).to(dev)
ignored = [m for m in model.segmentation_head.modules() if isinstance(m, torch.nn.Conv2d)]
sweep = SensitivitySweep(model, C, DEFAULT_RATIOS, ignored_layers=ignored)
torch.cuda.synchronize(); t0 = time.perf_counter()
with sweep.session():
    sweep.forward(img)
torch.cuda.synchronize(); print("phase 1 (NORMAL) first forward: %.2f s" % (time.perf_counter() - t0), flush=True)
t0 = time.perf_counter()
with sweep.session():
    sweep.forward(img)
torch.cuda.synchronize(); print("second forward: %.1f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
os.environ["MIOPEN_FIND_MODE"] = "FAST"
todo = list(range(0, len(sweep.candidates), 7))
t0 = time.perf_counter()
sweep.prepare(todo); n = sweep.warm_shapes(img, todo); torch.cuda.synchronize()
print("phase 2 (FAST): warmed %d problems in %.3f s (%.1f ms each)" % (n, time.perf_counter() - t0, 1e3 * (time.perf_counter() - t0) / max(n, 1)), flush=True)
conf = torch.zeros(C, C, dtype=torch.int64, device=dev)
with sweep.session():
    teacher = sweep.forward(img).float().contiguous()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in todo:
        sweep.evaluate(i, img, lab, teacher, conf, None)
    torch.cuda.synchronize()
print("candidates: %.2f ms each" % (1e3 * (time.perf_counter() - t0) / len(todo)))
