"""Where the fp16-output up-sampling differs from torch's fp32 kernel followed by a cast."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spice_amd import ops
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(2 + 7 + 36 + 36 + 7)
x = (3.0 * torch.randn(2, 7, 36, 36, generator=g)).half().to(dev)
got = ops.upsample_bilinear(x, scale_factor=4.0, align_corners=True, out_dtype=torch.float16)
f32 = ops.upsample_bilinear(x, scale_factor=4.0, align_corners=True, out_dtype=torch.float32)
there32 = torch.nn.functional.interpolate(x.float(), scale_factor=4.0, mode="bilinear", align_corners=True)
print("fp32 equal:", torch.equal(f32, there32))
there = there32.half()
bad = (got != there).nonzero()
print("mismatches:", len(bad), "of", got.numel())
for i in bad[:10]:
    i = tuple(i.tolist())
    print(i, float(got[i]), float(there[i]), repr(float(there32[i])), got[i].view(torch.int16).item(), there[i].view(torch.int16).item())
