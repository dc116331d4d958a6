"""Throughput of the evaluation loop (pruning/eval/components/core_evaluator.py:101-182: autocast forward,
arg-max, confusion update) on a model (default DeepLabV3+-R50; `python tools/evalbench.py Unet timm-efficientnet-b3` for another) at the reference's
evaluation geometry (576x576, batch 32, 5 classes), with the models' modules on the library's kernels and on torch's."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spice_amd import nets
from spice_amd.nets import layers
from spice_amd.pruning.eval import SegmentationMetrics
from spice_amd.pruning.eval.components import CoreEvaluator
dev = torch.device("cuda:0")
torch.manual_seed(0)
arch, enc = (sys.argv[1:] + ["DeepLabV3Plus", "resnet50"])[:2]
model = nets.create_model(arch, enc, 5).to(dev).eval()
loader = [(torch.randn(32, 3, 576, 576, device=dev), torch.randint(0, 5, (32, 576, 576), device=dev)) for _ in range(4)]
ev = CoreEvaluator(dev, 5, 576)
out = {}
for name, flag in (("library_forward", True), ("torch_forward", False)):
    layers.set_library_forward(flag)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = ev.evaluate_single_model(model, loader, SegmentationMetrics(5, dev), None)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    out[name] = {"images_per_s": round(128 / dt, 1), "ms_per_batch_of_32": round(1e3 * dt / 4, 2), "miou": res["miou"]}
layers.set_library_forward(True)
out["model"] = f"{arch}/{enc}"
print(json.dumps(out))
