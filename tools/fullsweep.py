"""Wall time of a WHOLE sensitivity sweep (every candidate, preparation included) on one GPU."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spice_amd import miopen_tuning
t0 = time.perf_counter()
miopen_tuning.ensure_tuned("DeepLabV3Plus", "resnet50", 21, 8, 512, "fp16")
tune_s = time.perf_counter() - t0
import torch
from spice_amd import nets, synth
from spice_amd.sweep import DEFAULT_RATIOS, SensitivitySweep
dev = torch.device("cuda:0"); C = 21
torch.manual_seed(1234)
model = nets.create_model("DeepLabV3Plus", "resnet50", C).to(dev).eval()
img = torch.from_numpy(synth.normal(1234, 50, 8 * 3 * 512 * 512).reshape(8, 3, 512, 512)).to(dev)
lab = torch.from_numpy(synth.label_map(1234, 8, 512, 512, C, "blocky", ignore_share=0.05)).to(dev)
ignored = [m for m in model.segmentation_head.modules() if isinstance(m, torch.nn.Conv2d)]
torch.cuda.synchronize(); t0 = time.perf_counter()
sweep = SensitivitySweep(model, C, DEFAULT_RATIOS, ignored_layers=ignored)
torch.cuda.synchronize(); build_s = time.perf_counter() - t0
with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    model(img)  # first forward of the process (library initialisation) outside the timed sweep
torch.cuda.synchronize(); t0 = time.perf_counter()
res = sweep.run(img, lab)
torch.cuda.synchronize(); run_s = time.perf_counter() - t0
n = len(sweep.candidates)
miou = res["miou"].cpu().numpy()
print(json.dumps({"candidates": n, "groups": len(sweep.groups), "find_db_warmup_s": round(tune_s, 2),
                  "sweep_object_s": round(build_s, 3), "sweep_run_s": round(run_s, 3),
                  "ms_per_candidate": round(1e3 * run_s / n, 3), "candidates_per_s": round(n / run_s, 1),
                  "new_conv_problems": len(sweep._warmed), "miou_min_max": [float(miou.min()), float(miou.max())],
                  "pixels_counted_ok": bool((res["confusion"].sum((1, 2)) == int((lab != 255).sum())).all())}))
