"""Wall time of MagnitudePruner construction and of each pruner.step() on DeepLabV3+-R50."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spice_amd import engine, nets
dev = torch.device("cuda:0")
for mode in (False, True):
    torch.manual_seed(0)
    model = nets.create_model("DeepLabV3Plus", "resnet50", 21).to(dev).eval()
    ignored = [m for m in model.segmentation_head.modules() if isinstance(m, torch.nn.Conv2d)]
    x = torch.randn(1, 3, 512, 512, device=dev)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pr = engine.MagnitudePruner(model, example_inputs=x, importance=engine.MagnitudeImportance(p=2), pruning_ratio=0.5,
                                iterative_steps=5, ignored_layers=ignored, global_pruning=mode)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"global={mode}: construct {1e3*(t1-t0):.1f} ms", flush=True)
    for s in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        pr.step()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        n = sum(p.numel() for p in model.parameters())
        print(f"  step {s+1}: {1e3*(t1-t0):.1f} ms, params {n/1e6:.2f} M", flush=True)
import cProfile, pstats
torch.manual_seed(0)
model = nets.create_model("DeepLabV3Plus", "resnet50", 21).to(dev).eval()
ignored = [m for m in model.segmentation_head.modules() if isinstance(m, torch.nn.Conv2d)]
pr = engine.MagnitudePruner(model, example_inputs=x, importance=engine.MagnitudeImportance(p=2), pruning_ratio=0.5,
                            iterative_steps=5, ignored_layers=ignored, global_pruning=False)
pr.step()
prof = cProfile.Profile(); prof.enable(); pr.step(); torch.cuda.synchronize(); prof.disable()
pstats.Stats(prof).sort_stats("cumulative").print_stats(25)
