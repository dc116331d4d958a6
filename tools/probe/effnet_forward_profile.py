"""Forward of DeepLabV3Plus / timm-efficientnet-b3 at [8,3,512,512] under fp16 autocast, ten times: for a kernel trace
(where the device time of the EfficientNet encoder goes) and the host time per forward."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spice_amd import nets
dev = torch.device("cuda:0")
torch.manual_seed(0)
arch, enc = (sys.argv[1:] + ["DeepLabV3Plus", "timm-efficientnet-b3"])[:2]
model = nets.create_model(arch, enc, 21).to(dev).eval()
x = torch.randn(8, 3, 512, 512, device=dev)
with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    for _ in range(3):
        model(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        model(x)
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize(); t_all = time.perf_counter() - t0
print(f"{arch}/{enc}: host {1e3 * t_host / 10:.2f} ms per forward, with device {1e3 * t_all / 10:.2f} ms")
