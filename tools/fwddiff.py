"""First module whose output differs between the no_grad (library kernels) and the grad-mode
(torch kernels) forward of a model, and whether each mode repeats itself."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spice_amd import nets
dev = torch.device("cuda:0")
torch.manual_seed(11)
model = nets.create_model("DeepLabV3Plus", "resnet50", 5).to(dev).eval()
for m in model.modules():
    if isinstance(m, torch.nn.BatchNorm2d):
        m.running_mean.normal_(0.0, 0.2); m.running_var.uniform_(0.5, 1.5)
        m.weight.data.uniform_(0.5, 1.5); m.bias.data.normal_(0.0, 0.2)
x = torch.randn(2, 3, 128, 128, device=dev)
names = {m: n for n, m in model.named_modules()}
def run(grad):
    outs = {}
    hooks = [m.register_forward_hook(lambda mod, i, o: outs.__setitem__(names[mod], o.detach().clone()))
             for m in model.modules() if not list(m.children())]
    with torch.autocast("cuda", dtype=torch.float16):
        if grad:
            y = model(x.clone().requires_grad_(True)).detach()
        else:
            with torch.no_grad():
                y = model(x)
    for h in hooks:
        h.remove()
    return y, outs
a1, oa1 = run(False); a2, oa2 = run(False); b1, ob1 = run(True); b2, ob2 = run(True)
print("no_grad repeat equal:", torch.equal(a1, a2), " grad repeat equal:", torch.equal(b1, b2), " modes equal:", torch.equal(a1, b1))
for tag, p, q in (("no_grad vs no_grad", oa1, oa2), ("grad vs grad", ob1, ob2), ("no_grad vs grad", oa1, ob1)):
    for n in p:
        if p[n].dtype != q[n].dtype or not torch.equal(p[n].float(), q[n].float()):
            d = (p[n].float() - q[n].float()).abs()
            print(tag, "first difference at", n, type(dict(model.named_modules())[n]).__name__, p[n].dtype, q[n].dtype,
                  "max", float(d.max()), "count", int((d > 0).sum()), "of", d.numel())
            break
    else:
        print(tag, "all module outputs equal")
