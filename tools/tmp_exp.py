import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spice_amd import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
def timeit(fn, iters=20):
    for i in range(3): fn(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for i in range(iters): fn(i)
    e1.record(); torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / iters
with torch.no_grad():
    for c, h, w, dil in [(512, 32, 32, 2), (520, 32, 32, 2), (504, 32, 32, 2), (512, 32, 40, 2), (512, 40, 32, 2), (256, 32, 32, 1), (264, 32, 32, 1), (256, 32, 40, 1)]:
        xs = [torch.randn(8, c, h, w, device=dev).half() for _ in range(4)]
        w9 = (torch.randn(c, 9 * c, device=dev) / (9 * c) ** 0.5).half()
        bn = torch.nn.BatchNorm2d(c).eval().to(dev)
        us = timeit(lambda i: ops.conv3x3_bn_act(xs[i % 4], w9, bn, True, 1, dil))
        gf = 2 * 8 * h * w * c * c * 9 / 1e9
        print(json.dumps({"op": "conv3x3", "c": c, "h": h, "w": w, "us": round(us, 1), "TF": round(gf / us / 1e3, 1)}), flush=True)
    for cin, cout, h, w in [(2048, 256, 32, 32), (2056, 256, 32, 32), (2048, 256, 32, 40), (1024, 256, 32, 32), (1032, 256, 32, 32), (512, 2048, 32, 32), (520, 2048, 32, 32)]:
        xs = [torch.randn(8, cin, h, w, device=dev).half() for _ in range(4)]
        w2 = (torch.randn(cout, cin, device=dev) / cin ** 0.5).half()
        bn = torch.nn.BatchNorm2d(cout).eval().to(dev)
        us = timeit(lambda i: ops.conv1x1_bn_act(xs[i % 4], w2, bn, None, None, True))
        gf = 2 * 8 * h * w * cin * cout / 1e9
        print(json.dumps({"op": "conv1x1", "cin": cin, "cout": cout, "h": h, "w": w, "us": round(us, 1), "TF": round(gf / us / 1e3, 1)}), flush=True)
