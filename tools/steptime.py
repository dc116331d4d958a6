"""Host-side time of each section of one sweep candidate (no device synchronisation inside the
loop, so the figures are enqueue costs), and the device-side total."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from spice_amd import miopen_tuning
if os.environ.get("TUNE", "1") == "1":
    print("tune s", miopen_tuning.ensure_tuned("DeepLabV3Plus", "resnet50", 21, 8, 512, "fp16"), flush=True)
from spice_amd import nets, ops, synth
from spice_amd.sweep import DEFAULT_RATIOS, SensitivitySweep

if os.environ.get("IMMEDIATE") == "1":
    torch.backends.miopen.immediate = True
dev = torch.device("cuda:0")
torch.manual_seed(1234)
C = 21
model = nets.create_model("DeepLabV3Plus", "resnet50", C).to(dev).eval()
img = torch.from_numpy(synth.normal(1234, 50, 8 * 3 * 512 * 512).reshape(8, 3, 512, 512)).to(dev)
lab = torch.from_numpy(synth.label_map(1234, 8, 512, 512, C, "blocky", ignore_share=0.05)).to(dev)
ignored = [m for m in model.segmentation_head.modules() if isinstance(m, torch.nn.Conv2d)]
sweep = SensitivitySweep(model, C, DEFAULT_RATIOS, ignored_layers=ignored)
todo = list(range(0, len(sweep.candidates), 5))[:80]
sweep.prepare(todo)
acc = {k: 0.0 for k in ("run", "forward", "contig", "argmax", "float", "kd", "copy", "undo")}
with sweep.session():
    teacher = sweep.forward(img).float().contiguous()
    conf = torch.zeros(C, C, dtype=torch.int64, device=dev); kd = torch.zeros(3, device=dev)
    for phase in ("warm", "timed"):
        torch.cuda.synchronize(); t_all = time.perf_counter()
        for i in (todo[:20] if phase == "warm" else todo[20:]):
            job = sweep.jobs[i]
            t = time.perf_counter(); job.run(keep_undo=True); t1 = time.perf_counter(); acc["run"] += t1 - t
            logits = sweep.forward(img); t2 = time.perf_counter(); acc["forward"] += t2 - t1
            logits = logits.contiguous(); t3 = time.perf_counter(); acc["contig"] += t3 - t2
            ops.argmax_confusion_update(conf, logits, lab, C); t4 = time.perf_counter(); acc["argmax"] += t4 - t3
            st = logits.float(); t5 = time.perf_counter(); acc["float"] += t5 - t4
            out3, _, _ = ops.kdlite_forward(st, teacher, lab, temperature=4.0, alpha=0.7); t6 = time.perf_counter(); acc["kd"] += t6 - t5
            kd.copy_(out3); t7 = time.perf_counter(); acc["copy"] += t7 - t6
            job.undo(); t8 = time.perf_counter(); acc["undo"] += t8 - t7
        torch.cuda.synchronize(); wall = time.perf_counter() - t_all
        n = 20 if phase == "warm" else len(todo) - 20
        print(phase, "wall ms/cand %.2f" % (1e3 * wall / n), {k: round(1e3 * v / n, 3) for k, v in acc.items()}, flush=True)
        for k in acc: acc[k] = 0.0
    # the same candidates again: every convolution shape has been seen once
    torch.cuda.synchronize(); t_all = time.perf_counter()
    for i in todo[20:]:
        job = sweep.jobs[i]
        job.run(keep_undo=True); logits = sweep.forward(img); job.undo()
    torch.cuda.synchronize(); wall = time.perf_counter() - t_all
    print("repeat (known shapes) wall ms/cand %.2f" % (1e3 * wall / (len(todo) - 20)), flush=True)
print("logits dtype", logits.dtype, {k: v for k, v in os.environ.items() if k.startswith("MIOPEN")})
