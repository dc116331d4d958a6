"""Details of the elements where formula 0 differs from torch's inference batch-norm (fp16)."""
import ctypes, os
import numpy as np
import torch
import torch.nn.functional as F
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "bnprobe.so"))
dev = torch.device("cuda:0")
P = ctypes.c_void_p
def run(x, mean, var, wt, b, form=0, inv_kind=0):
    y = torch.empty_like(x)
    n, c, h, w_ = x.shape
    lib.bn_probe(P(x.data_ptr()), P(y.data_ptr()), P(mean.data_ptr()), P(var.data_ptr()), P(wt.data_ptr()), P(b.data_ptr()),
                 ctypes.c_float(1e-5), ctypes.c_int64(x.numel()), ctypes.c_int(c), ctypes.c_int64(h * w_),
                 ctypes.c_int(1 if x.dtype == torch.float16 else 0), ctypes.c_int(inv_kind), ctypes.c_int(form),
                 P(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return y
tot = 0
for seed in range(40):
    torch.manual_seed(seed)
    shape = (2, 304, 9, 7)
    n, c, h, w_ = shape
    x = (2.0 * torch.randn(*shape, device=dev)).half()
    mean = torch.randn(c, device=dev); var = torch.rand(c, device=dev) * 2 + 0.05
    wt = torch.rand(c, device=dev) + 0.5; b = torch.randn(c, device=dev)
    want = F.batch_norm(x, mean, var, wt, b, False, 0.1, 1e-5)
    got = run(x, mean, var, wt, b)
    want32 = F.batch_norm(x.float(), mean, var, wt, b, False, 0.1, 1e-5)
    got32 = run(x.float(), mean, var, wt, b)
    bad = (got != want).nonzero()
    tot += len(bad)
    for i in bad[:3]:
        i = tuple(i.tolist())
        print(seed, i, "x", float(x[i]), "want", float(want[i]), "got", float(got[i]), "fp32 want", repr(float(want32[i])), "fp32 got", repr(float(got32[i])),
              "flat", (i[0] * c + i[1]) * h * w_ + i[2] * w_ + i[3])
print("total mismatches", tot)
