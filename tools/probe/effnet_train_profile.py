"""Forward + backward of UNET / timm-efficientnet-b3 in fp32 (as the reference's fine-tuning runs it), five times: for a
kernel trace of where the training step's device time goes."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spice_amd import nets
dev = torch.device("cuda:0")
torch.manual_seed(0)
arch, enc = (sys.argv[1:] + ["Unet", "timm-efficientnet-b3"])[:2]
model = nets.create_model(arch, enc, 21).to(dev).train()
x = torch.randn(8, 3, 512, 512, device=dev)
y = torch.randint(0, 21, (8, 512, 512), device=dev)
for _ in range(2):
    model.zero_grad(); torch.nn.functional.cross_entropy(model(x), y).backward()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    model.zero_grad(); torch.nn.functional.cross_entropy(model(x), y).backward()
torch.cuda.synchronize()
print(f"{arch}/{enc}: {1e3 * (time.perf_counter() - t0) / 5:.1f} ms per forward + backward (fp32, batch 8, 512 x 512)")
