"""Host time per call of the library's entry points: through torch.ops.spice.* (what spice_amd.ops uses)
against a ctypes binding of the same C ABI (what it used before).  Tiny tensors, so the device is never the
bottleneck and the wall time of a loop of calls is host time.  Usage: python tools/callcost.py [--n 3000]"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spice_amd import _native as nat  # noqa: E402
from spice_amd import ops  # noqa: E402


def loop_us(fn, n):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    dt = time.perf_counter() - t0
    torch.cuda.synchronize()
    return 1e6 * dt / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=3000)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    L = nat.lib()
    raw_stream = torch._C._cuda_getCurrentRawStream
    out = {}

    x = torch.randn(2, 64, 8, 8, device=dev).half()
    mean, var, w, b = (torch.rand(64, device=dev) + 0.5 for _ in range(4))

    def bn_ctypes():
        y = torch.empty_like(x)
        L.spice_bn_act(x.data_ptr(), None, y.data_ptr(), 1, mean.data_ptr(), var.data_ptr(), w.data_ptr(), b.data_ptr(),
                       1e-5, 2, 64, 64, 1, raw_stream(0))
        return y

    out["bn_act"] = {"torch_ops_us": loop_us(lambda: ops.bn_act(x, mean, var, w, b, 1e-5, relu=True), args.n),
                     "ctypes_us": loop_us(bn_ctypes, args.n)}

    w2 = torch.randn(64, 64, device=dev).half()

    def c1_ctypes():
        y = torch.empty((2, 64, 8, 8), dtype=torch.float16, device=dev)
        L.spice_conv1x1_f16(x.data_ptr(), w2.data_ptr(), y.data_ptr(), 2, 64, 64, 64, raw_stream(0))
        return y

    out["conv1x1_f16"] = {"torch_ops_us": loop_us(lambda: ops.conv1x1_f16(x, w2), args.n),
                          "ctypes_us": loop_us(c1_ctypes, args.n)}

    c = 21
    logits = torch.randn(1, c, 16, 16, device=dev)
    lab = torch.randint(0, c, (1, 16, 16), device=dev)
    conf = torch.zeros(c, c, dtype=torch.int64, device=dev)
    ws = torch.zeros(L.spice_confusion_workspace_bytes(256, c), dtype=torch.uint8, device=dev)

    def am_ctypes():
        rc = L.spice_argmax_confusion_update(logits.data_ptr(), 0, lab.data_ptr(), 1, 256, c, conf.data_ptr(), ws.data_ptr(),
                                             ws.numel(), torch.cuda.current_stream(dev).cuda_stream)
        nat.check(rc, "x")

    out["argmax_confusion_update"] = {"torch_ops_us": loop_us(lambda: ops.argmax_confusion_update(conf, logits, lab, c), args.n),
                                      "ctypes_us": loop_us(am_ctypes, args.n)}

    kws = torch.zeros(L.spice_kdlite_workspace_bytes(1, c, 16, 16), dtype=torch.uint8, device=dev)
    teacher = logits + 1

    def kd_ctypes():
        o3 = torch.empty(3, device=dev)
        p = nat.KdliteParams(4.0, 0.7, 1.0, 0, 255, None)
        rc = L.spice_kdlite_forward(logits.data_ptr(), teacher.data_ptr(), lab.data_ptr(), 1, c, 16, 16, ctypes.byref(p),
                                    o3.data_ptr(), None, 0.7, 0.3, None, kws.data_ptr(), kws.numel(),
                                    torch.cuda.current_stream(dev).cuda_stream)
        nat.check(rc, "x")
        return o3

    out["kdlite_forward"] = {"torch_ops_us": loop_us(lambda: ops.kdlite_forward(logits, teacher, lab, temperature=4.0, alpha=0.7), args.n),
                             "ctypes_us": loop_us(kd_ctypes, args.n)}
    for v in out.values():
        for k in v:
            v[k] = round(v[k], 2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
