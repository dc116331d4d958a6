"""The 1x1 convolution kernel's forms on the shapes of a DeepLabV3+-ResNet-50 sweep (pruned channel counts included),
each launched behind a kernel that fills the LDS of every compute unit with a pattern (quiet NaNs, signalling NaNs,
ones): a kernel that reads LDS it has not written gives another result then.  Build tools/probe/poison_lds.so first
(tools/README.md)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from spice_amd import ops
dev = torch.device("cuda:0")
P = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "poison_lds.so"))
P.poison_lds.argtypes = [ctypes.c_uint, ctypes.c_void_p]
def poison(pattern=0x7fc07fc0):
    assert P.poison_lds(pattern, torch.cuda.current_stream().cuda_stream) == 0
torch.manual_seed(0)
def bn(c):
    b = torch.nn.BatchNorm2d(c).eval().to(dev)
    b.running_mean.normal_(0, 0.2); b.running_var.uniform_(0.5, 1.5); b.weight.data.uniform_(0.5, 1.5); b.bias.data.normal_(0, 0.2)
    return b
cases = []
shapes = [(512, 820, 32, "dual"), (512, 615, 32, "dual"), (512, 410, 32, "res"), (512, 205, 32, "dual"), (820, 512, 32, "bn"),
          (205, 512, 32, "bn"), (615, 256, 32, "bn"), (2048, 231, 32, "bn"), (256, 231, 32, "bn"), (820, 256, 32, "bn"), (410, 256, 32, "bn"),
          (64, 256, 128, "res"), (256, 64, 128, "bn"), (1024, 256, 32, "bn"), (256, 1024, 32, "dual"), (304, 256, 128, "bn"), (256, 21, 128, "bn"),
          (231, 256, 32, "bn"), (1280, 256, 32, "bn"), (1255, 256, 32, "bn")]
for (cin, cout, hw, tail) in shapes:
    x = torch.randn(8, cin, hw, hw, device=dev).half()
    w = (torch.randn(cout, cin, device=dev) / cin ** 0.5).half()
    b1, b2 = bn(cout), bn(cout)
    r = torch.randn(8, cout, hw, hw, device=dev).half() if tail != "bn" else None
    def call(x=x, w=w, b1=b1, b2=b2, r=r, tail=tail):
        return ops.conv1x1_bn_act(x, w, b1, residual=r, res_bn=b2 if tail == "dual" else None, relu=True)
    cases.append(((cin, cout, hw, tail), call))
with torch.no_grad():
    for name, call in cases:
        outs = []
        for pattern in (None, 0x7fc07fc0, 0x7f800001, 0x3c003c00):
            if pattern is not None:
                poison(pattern)
            outs.append(call().clone())
            torch.cuda.synchronize()
        same = [bool(torch.equal(outs[0], o)) for o in outs[1:]]
        nan = [int(torch.isnan(o).sum()) for o in outs]
        print(name, "equal to the unpoisoned result:", same, "NaNs:", nan)

# the 3x3 modes (pruned, odd channel counts; stride 2; dilation 2)
# cin, cout, side, stride, dilation
for (cin, cout, side, stride, dil) in ((461, 512, 32, 1, 2), (512, 512, 32, 1, 2), (410, 512, 32, 1, 2), (205, 256, 32, 1, 1), (64, 64, 128, 1, 1),
                                      (57, 64, 128, 1, 1), (128, 128, 128, 2, 1), (115, 128, 128, 2, 1), (256, 256, 32, 1, 1), (231, 256, 32, 1, 1),
                                      (128, 128, 64, 1, 1), (256, 256, 64, 2, 1), (230, 256, 64, 2, 1)):
    x = torch.randn(8, cin, side, side, device=dev).half()
    w9 = (torch.randn(cout, 9 * cin, device=dev) / (9 * cin) ** 0.5).half()
    b = bn(cout)
    with torch.no_grad():
        outs = []
        for pattern in (None, 0x7fc07fc0, 0x7f800001, 0x3c003c00):
            if pattern is not None:
                poison(pattern)
            outs.append(ops.conv3x3_bn_act(x, w9, b, True, stride, dil).clone())
            torch.cuda.synchronize()
    print((cin, cout, side, stride, dil), [bool(torch.equal(outs[0], o)) for o in outs[1:]], [int(torch.isnan(o).sum()) for o in outs])
