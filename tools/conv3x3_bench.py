"""3x3 convolution + batch-norm + ReLU: one launch of the library's matrix-core kernel against torch's convolution
(MIOpen) + the library's batch-norm launch, for the bottleneck convolutions of the DeepLabV3+-ResNet-50 forward on an
8x3x512x512 batch (--unet: the narrow convolutions at the end of the U-Net decoder instead).
Usage: python tools/conv3x3_bench.py [--iters 20] [--unet]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from spice_amd import ops

# (name, count per forward, channels, input side, stride, dilation)
CONVS = [("layer1.x.conv2", 3, 64, 128, 1, 1), ("layer2.0.conv2", 1, 128, 128, 2, 1), ("layer2.x.conv2", 3, 128, 64, 1, 1),
         ("layer3.0.conv2", 1, 256, 64, 2, 1), ("layer3.x.conv2", 5, 256, 32, 1, 1), ("layer4.x.conv2", 3, 512, 32, 1, 2)]
# the same with (input channels, output channels): U-Net decoder blocks 3 and 4 and the head on a ResNet-50 encoder
UNET = [("decoder.3.conv1", 1, (128, 32), 256, 1, 1), ("decoder.3.conv2", 1, (32, 32), 256, 1, 1),
        ("decoder.4.conv1", 1, (32, 16), 512, 1, 1), ("decoder.4.conv2", 1, (16, 16), 512, 1, 1),
        ("head", 1, (16, 21), 512, 1, 1)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--only", default="", help="library | miopen_plus_bn: time that side only (counter passes)")
    ap.add_argument("--convs", default="", help="comma-separated names: those convolutions only")
    ap.add_argument("--unet", action="store_true", help="the U-Net decoder's narrow convolutions")
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    total = {"library": 0.0, "miopen_plus_bn": 0.0}
    for name, count, c, side, stride, dil in (UNET if args.unet else CONVS):
        if args.convs and name not in args.convs.split(","):
            continue
        cin, cout = c if isinstance(c, tuple) else (c, c)
        xs = [torch.randn(args.batch, cin, side, side, device=dev).half() for _ in range(4)]
        w = (torch.randn(cout, cin, 3, 3, device=dev) / (9 * cin) ** 0.5).half()
        w9 = w.permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous()
        bn = torch.nn.BatchNorm2d(cout).eval().to(dev)
        out_px = args.batch * (side // stride) ** 2
        row = {"conv": name, "count": count, "channels": c, "in": side, "stride": stride, "dilation": dil,
               "GF": round(2 * out_px * cin * cout * 9 / 1e9, 1),
               "MB": round(2 * (args.batch * cin * side * side + out_px * cout) / 1e6, 1)}
        fns = {"library": lambda i: ops.conv3x3_bn_act(xs[i % 4], w9, bn, True, stride, dil),
               "miopen_plus_bn": lambda i: ops.bn_act(F.conv2d(xs[i % 4], w, None, stride, dil, dil), bn.running_mean,
                                                      bn.running_var, bn.weight, bn.bias, bn.eps, relu=True)}
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
                us = 1e3 * e0.elapsed_time(e1) / args.iters
                row[key + "_us"] = round(us, 1)
                total[key] += us * count
        print(json.dumps(row), flush=True)
    print(json.dumps({"per_forward_us": {k: round(v, 1) for k, v in total.items()}}))


if __name__ == "__main__":
    main()
