"""Host enqueue time and device time of a 1x1 convolution through F.conv2d (MIOpen) and through
torch.matmul (hipBLASLt) on fp16 NCHW activations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spice_amd import miopen_tuning
miopen_tuning.ensure_tuned("DeepLabV3Plus", "resnet50", 21, 8, 512, "fp16")
import torch
import torch.nn.functional as F
dev = torch.device("cuda:0")
shapes = [(8, 64, 128, 128, 256), (8, 256, 128, 128, 64), (8, 512, 64, 64, 128), (8, 1024, 32, 32, 256), (8, 2048, 32, 32, 512),
          (8, 256, 32, 32, 1024), (8, 1280, 32, 32, 256), (8, 304, 128, 128, 256), (8, 100, 128, 128, 256)]
def bench(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    host = time.perf_counter() - t0
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    return 1e6 * host / reps, 1e6 * wall / reps
for n, cin, h, w, cout in shapes:
    x = torch.randn(n, cin, h, w, device=dev).half()
    wt = (torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5).half()
    w2 = wt.view(cout, cin)
    a = F.conv2d(x, wt)
    b = torch.matmul(w2, x.view(n, cin, h * w)).view(n, cout, h, w)
    err = float((a.float() - b.float()).abs().max()); ref = float(a.float().abs().max())
    hc, wc = bench(lambda: F.conv2d(x, wt))
    hm, wm = bench(lambda: torch.matmul(w2, x.view(n, cin, h * w)).view(n, cout, h, w))
    print(f"[{n},{cin},{h},{w}]->{cout}: conv2d host {hc:5.1f} us wall {wc:6.1f} us | matmul host {hm:5.1f} us wall {wm:6.1f} us | max diff {err:.2e} of {ref:.2f}", flush=True)
