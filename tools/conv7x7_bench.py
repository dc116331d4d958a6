"""The ResNet stem's 7x7 stride-2 convolution on an 8x3x512x512 batch: one launch of the library's matrix-core kernel
against torch's convolution (MIOpen, as it picks without tuning).  Usage: python tools/conv7x7_bench.py [--iters 20]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from spice_amd import ops


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    xs = [torch.randn(8, 3, 512, 512, device=dev).half() for _ in range(4)]
    w = (torch.randn(64, 3, 7, 7, device=dev) / 12).half()
    w2 = w.reshape(64, -1).contiguous()
    fns = {"library": lambda i: ops.conv7x7_s2(xs[i % 4], w2), "miopen": lambda i: F.conv2d(xs[i % 4], w, None, 2, 3)}
    row = {"conv": "stem 7x7/2, 3 -> 64, [8,3,512,512]", "GF": round(2 * 8 * 256 * 256 * 64 * 147 / 1e9, 1)}
    with torch.no_grad():
        for key, fn in fns.items():
            if args.only and key != args.only:
                continue
            for i in range(3):
                fn(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for i in range(args.iters):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            row[key + "_us"] = round(1e3 * e0.elapsed_time(e1) / args.iters, 1)
    print(json.dumps(row))


if __name__ == "__main__":
    main()
