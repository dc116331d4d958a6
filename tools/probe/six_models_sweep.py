"""A sensitivity sweep (three ratios per group, second run timed) on each of the six model names: candidates/s, which
path the plans take, how many convolution problems still reach MIOpen."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spice_amd import nets, synth
from spice_amd.pruning.model_configs import MODEL_CONFIGS
from spice_amd.sweep import SensitivitySweep

dev = torch.device("cuda:0"); C = 21; N, S = 8, 512
img = torch.from_numpy(synth.normal(1234, 50, N * 3 * S * S).reshape(N, 3, S, S)).to(dev)
lab = torch.from_numpy(synth.label_map(1234, N, S, S, C, "blocky", ignore_share=0.05)).to(dev)
for name, cfg in MODEL_CONFIGS.items():
    torch.manual_seed(1)
    model = nets.create_model(cfg["architecture"], cfg["encoder_name"], C).to(dev).eval()
    ignored = [m for m in model.segmentation_head.modules() if isinstance(m, torch.nn.Conv2d)]
    sweep = SensitivitySweep(model, C, (0.25, 0.5, 0.75), ignored_layers=ignored)
    sweep.enable_lanes(2)
    need = sweep.may_need_solver_lookups(img)
    with sweep.session():
        sweep.run(img, lab)   # (first run: MIOpen's first calls, plans)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = sweep.run(img, lab)
        torch.cuda.synchronize(); t1 = time.perf_counter() - t0
        sweep.teacher_forward(img)
        g = sweep.capture_suffix_graphs(img)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res2 = sweep.run(img, lab)
        torch.cuda.synchronize(); t2 = time.perf_counter() - t0
    n = len(sweep.candidates)
    print(json.dumps({"model": name, "candidates": n, "groups": len(sweep.groups), "may_need_solver_lookups": need,
                      "problems_looked_up": sum(len(l._warmed) for l in sweep._lanes),
                      "eager_candidates_per_s": round(n / t1, 1), "graphs": g["graphs"],
                      "graph_candidates_per_s": round(n / t2, 1),
                      "same_tables": bool(torch.equal(res["confusion"], res2["confusion"])),
                      "max_cells_moved": int((res["confusion"] - res2["confusion"]).abs().sum((1, 2)).max())}), flush=True)
    del sweep, model
    torch.cuda.empty_cache()
