"""For which plane sizes torch's inference batch-norm (fp16) equals formula 0 with a separate
conversion, over enough elements to meet fp32 ties."""
import ctypes, os
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
for (h, w_) in ((1, 1), (1, 2), (1, 3), (2, 2), (1, 5), (2, 3), (1, 7), (3, 3), (2, 5), (3, 4), (3, 5), (4, 4), (3, 6), (5, 5), (6, 6),
                (7, 7), (9, 7), (8, 8), (9, 9), (10, 10), (11, 11), (18, 18), (36, 36), (33, 33), (17, 2), (17, 6), (72, 72), (144, 144)):
    hw = h * w_
    c = 96
    n = max(1, min(64, (1 << 21) // (c * hw)))
    bad = tot = 0
    for seed in range(3):
        torch.manual_seed(seed)
        x = (2.0 * torch.randn(n, c, h, w_, device=dev)).half()
        mean = torch.randn(c, device=dev); var = torch.rand(c, device=dev) * 2 + 0.05
        wt = torch.rand(c, device=dev) + 0.5; b = torch.randn(c, device=dev)
        want = F.batch_norm(x, mean, var, wt, b, False, 0.1, 1e-5)
        got = run(x, mean, var, wt, b)
        bad += int((got != want).sum()); tot += x.numel()
    print(f"hw={hw:6d} ({h}x{w_}) n={n:3d} elements={tot:9d} mismatches={bad}", flush=True)
