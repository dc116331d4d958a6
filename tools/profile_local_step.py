"""cProfile of one local-mode MagnitudePruner.step() (host side) on one model name."""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spice_amd import engine
from spice_amd.pruning import PruningConfig
from spice_amd.pruning.models import ModelFactory

name = sys.argv[1] if len(sys.argv) > 1 else "DeepLabV3Plus_resnet"
dev = torch.device("cuda:0")
cfg = PruningConfig(model_name=name, use_wandb=False, output_dir="/tmp/unused_spice")
model = ModelFactory.create_model(cfg).to(dev).eval()
ignored = ModelFactory.get_ignored_layers(model, cfg)
x = torch.randn(1, 3, 576, 576, device=dev)
pr = engine.MagnitudePruner(model, example_inputs=x, importance=engine.MagnitudeImportance(p=2), pruning_ratio=0.5,
                            iterative_steps=4, ignored_layers=ignored, global_pruning=False)
pr.step()
torch.cuda.synchronize()
prof = cProfile.Profile()
prof.enable()
pr.step()
torch.cuda.synchronize()
prof.disable()
pstats.Stats(prof).sort_stats("cumulative").print_stats(35)
