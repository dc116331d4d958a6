import sys; sys.path.insert(0, "/root/repo")
import torch
from spice_amd import nets
dev = torch.device("cuda:0")
m = nets.create_model("Unet", "resnet50", 21).to(dev).eval()
x = torch.randn(2, 3, 256, 256, device=dev)
def hook(name):
    def f(mod, inp, out):
        i = inp[0]
        print(name, type(mod).__name__, "in", i.dtype, tuple(i.shape), "out", out.dtype if torch.is_tensor(out) else type(out))
    return f
for n, mod in m.named_modules():
    if n.startswith("decoder.blocks") and n.count(".") == 2 or n.startswith("segmentation_head"):
        mod.register_forward_hook(hook(n))
with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
    feats = m.encoder(x)
    print([f.dtype for f in feats])
    m(x)
