"""First-call vs steady-state forward time of the DeepLabV3+ model under the current MIOpen env."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spice_amd import nets
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = nets.create_model("DeepLabV3Plus", "resnet50", 21).to(dev).eval()
x = torch.randn(8, 3, 512, 512, device=dev)
print("env", {k: v for k, v in os.environ.items() if k.startswith("MIOPEN")}, flush=True)
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        y = m(x)
    torch.cuda.synchronize(); print(f"  iter {it}: {1e3*(time.perf_counter()-t0):.1f} ms", flush=True)
# a new shape: slice one conv and time again
c = m.encoder.layer3[2].conv2
for n_keep in (230, 204, 179):
    w = c.weight.data[:n_keep].clone()
    conv = torch.nn.Conv2d(256, n_keep, 3, padding=1, bias=False).to(dev)
    conv.weight.data.copy_(w)
    xin = torch.randn(8, 256, 32, 32, device=dev)
    for it in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            y = conv(xin)
        torch.cuda.synchronize(); print(f"  conv out={n_keep} iter {it}: {1e3*(time.perf_counter()-t0):.2f} ms", flush=True)
