"""Wall time of MagnitudePruner construction and of each pruner.step() for the six model names."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spice_amd import engine
from spice_amd.pruning import PruningConfig
from spice_amd.pruning.models import ModelFactory
from spice_amd.pruning.model_configs import MODEL_CONFIGS
dev = torch.device("cuda:0")
for name in MODEL_CONFIGS:
    for mode in (False, "serial", True):
        torch.manual_seed(0)
        cfg = PruningConfig(model_name=name, use_wandb=False, output_dir="/tmp/unused_spice")
        model = ModelFactory.create_model(cfg).to(dev).eval()
        ignored = ModelFactory.get_ignored_layers(model, cfg)
        x = torch.randn(1, 3, 576, 576, device=dev)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        imp = engine.MagnitudeImportance(p=2)
        if mode == "serial":  # a plug-in importance without batch(): the group-after-group local path
            inner = imp
            class OneByOne(engine.Importance):
                def __call__(self, group):
                    return inner(group)
            imp = OneByOne()
        pr = engine.MagnitudePruner(model, example_inputs=x, importance=imp, pruning_ratio=0.5,
                                    iterative_steps=3, ignored_layers=ignored, global_pruning=(mode is True))
        torch.cuda.synchronize(); t1 = time.perf_counter()
        steps = []
        for s in range(3):
            torch.cuda.synchronize(); a = time.perf_counter()
            pr.step()
            torch.cuda.synchronize(); steps.append(1e3 * (time.perf_counter() - a))
        n = sum(p.numel() for p in model.parameters())
        print(f"{name:28s} global={mode!s:5}: construct {1e3*(t1-t0):7.1f} ms, steps " + " ".join(f"{v:6.1f}" for v in steps) +
              f" ms, groups {len(pr.prunable_groups())}, params left {n/1e6:.2f} M", flush=True)
