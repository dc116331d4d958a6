"""1x1 convolution + batch-norm tail, one launch (matrix-core kernel) against two launches (hipBLASLt product +
batch-norm pass), for the pairs of the DeepLabV3+-ResNet-50 forward on an 8x3x512x512 batch.
Usage: python tools/conv1x1_bench.py [--iters 30] [--routes 1,2]"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from spice_amd import ops

# (name, count per forward, Cin, Cout, H*W side, tail)
PAIRS = [
    ("layer1.0.conv1", 1, 64, 64, 128, "bn_relu"), ("layer1.x.conv1", 2, 256, 64, 128, "bn_relu"),
    ("layer1.0.conv3", 1, 64, 256, 128, "dual"), ("layer1.x.conv3", 2, 64, 256, 128, "res"),
    ("layer2.0.conv1", 1, 256, 128, 128, "bn_relu"), ("layer2.x.conv1", 3, 512, 128, 64, "bn_relu"),
    ("layer2.0.conv3", 1, 128, 512, 64, "dual"), ("layer2.x.conv3", 3, 128, 512, 64, "res"),
    ("layer3.0.conv1", 1, 512, 256, 64, "bn_relu"), ("layer3.x.conv1", 5, 1024, 256, 32, "bn_relu"),
    ("layer3.0.conv3", 1, 256, 1024, 32, "dual"), ("layer3.x.conv3", 5, 256, 1024, 32, "res"),
    ("layer4.0.conv1", 1, 1024, 512, 32, "bn_relu"), ("layer4.x.conv1", 2, 2048, 512, 32, "bn_relu"),
    ("layer4.0.conv3", 1, 512, 2048, 32, "dual"), ("layer4.x.conv3", 2, 512, 2048, 32, "res"),
    ("aspp.branch", 4, 2048, 256, 32, "bn_relu"), ("aspp.project", 1, 1280, 256, 32, "bn_relu"),
    ("aspp.tail", 1, 256, 256, 32, "bn_relu"), ("decoder.block1", 1, 256, 48, 128, "bn_relu"),
    ("decoder.block2", 1, 304, 256, 128, "bn_relu"),
]


def norm(c, dev):
    bn = torch.nn.BatchNorm2d(c).eval().to(dev)
    bn.running_mean.normal_(0, 0.2); bn.running_var.uniform_(0.5, 1.5)
    bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.normal_(0, 0.2)
    return bn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--routes", default="1,2")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--plain", action="store_true")
    args = ap.parse_args()
    routes = [int(r) for r in args.routes.split(",")]
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    rows, total = [], {r: 0.0 for r in routes}
    for name, count, cin, cout, side, tail in PAIRS:
        n_buf = 4
        xs = [torch.randn(args.batch, cin, side, side, device=dev).half() for _ in range(n_buf)]
        w = (torch.randn(cout, cin, device=dev) / cin ** 0.5).half()
        res = [torch.randn(args.batch, cout, side, side, device=dev).half() for _ in range(n_buf)] if tail != "bn_relu" else None
        bn, rbn = norm(cout, dev), (norm(cout, dev) if tail == "dual" else None)
        p = args.batch * side * side
        mbytes = 2 * p * (cin + cout + (cout if res else 0)) / 1e6
        row = {"pair": name, "count": count, "cin": cin, "cout": cout, "pixels": p, "tail": tail, "MB": round(mbytes, 1),
               "GF": round(2 * p * cin * cout / 1e9, 1)}
        with torch.no_grad():
            for r in routes:
                for i in range(3):
                    ops.conv1x1_bn_act(xs[i % n_buf], w, bn, res[i % n_buf] if res else None, rbn, True, route=r)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                for i in range(args.iters):
                    ops.conv1x1_bn_act(xs[i % n_buf], w, bn, res[i % n_buf] if res else None, rbn, True, route=r)
                e1.record()
                torch.cuda.synchronize()
                us = 1e3 * e0.elapsed_time(e1) / args.iters
                row[f"route{r}_us"] = round(us, 1)
                row[f"route{r}_GBps"] = round(mbytes / us * 1e3, 0)
                total[r] += us * count
        if args.plain:   # the product alone through the same kernel (no tail, no residual): what the tail costs
            import ctypes
            from spice_amd import _native as nat
            null = ctypes.c_void_p(0)
            y = torch.empty(args.batch, cout, side, side, device=dev, dtype=torch.float16)
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            call = lambda i: nat.lib().spice_conv1x1_bn_act_f16(xs[i % n_buf].data_ptr(), w.data_ptr(), y.data_ptr(), args.batch, cin,
                                                                 side * side, cout, null, null, null, null, 0.0, null, null, null,
                                                                 null, null, 0.0, 0, 0, null, null, null, null, 0.0, 0, stream)
            for i in range(3):
                call(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for i in range(args.iters):
                call(i)
            e1.record()
            torch.cuda.synchronize()
            row["plain_us"] = round(1e3 * e0.elapsed_time(e1) / args.iters, 1)
            row["plain_MB"] = round(2 * p * (cin + cout) / 1e6, 1)
            for i in range(3):
                ops.conv1x1_f16(xs[i % n_buf], w)
            torch.cuda.synchronize()
            e0.record()
            for i in range(args.iters):
                ops.conv1x1_f16(xs[i % n_buf], w)
            e1.record()
            torch.cuda.synchronize()
            row["blaslt_us"] = round(1e3 * e0.elapsed_time(e1) / args.iters, 1)
        rows.append(row)
        print(json.dumps(row), flush=True)
    print(json.dumps({"per_forward_us": {f"route{r}": round(v, 1) for r, v in total.items()}}))


if __name__ == "__main__":
    main()
