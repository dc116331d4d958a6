"""The ResNet stem's 7x7 stride-2 convolution in one launch of the matrix-core kernel (spice_conv7x7_stride2_f16): the
49 taps are picked from x while it is staged.  Exact on integer-valued data against torch's convolution in fp64; on
random data within 2^-9 of the largest result of the fp64 sum (fp32 sums rounded once), and within 2^-8 of MIOpen's."""
import pytest
import torch

from spice_amd import ops
from spice_amd.nets import layers, resnet

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


# (n, cin, h, w, cout): image edges in both directions, one and several 64-pixel tiles per row, M and K tails
CASES = [(1, 3, 8, 16, 64), (2, 3, 32, 32, 64), (1, 3, 6, 48, 40), (2, 1, 16, 16, 8), (1, 4, 8, 32, 130), (2, 3, 64, 128, 64),
         (8, 3, 128, 128, 64)]


@pytest.mark.parametrize("case", CASES)
def test_conv7x7_exact_on_integer_data(dev, case):
    n, cin, h, w, cout = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randint(-3, 4, (n, cin, h, w), generator=g).half()
    wt = torch.randint(-2, 3, (cout, cin, 7, 7), generator=g).half()
    exact = torch.nn.functional.conv2d(x.double(), wt.double(), None, 2, 3).half().to(dev)
    got = ops.conv7x7_s2(x.to(dev), wt.reshape(cout, -1).to(dev))
    assert got.shape == exact.shape and got.dtype == torch.float16
    assert torch.equal(got, exact)


def test_conv7x7_on_random_data(dev):
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 64, 96, generator=g).half()
    wt = (torch.randn(64, 3, 7, 7, generator=g) / 12).half()
    exact = torch.nn.functional.conv2d(x.double(), wt.double(), None, 2, 3)
    got = ops.conv7x7_s2(x.to(dev), wt.reshape(64, -1).to(dev)).cpu().double()
    there = torch.nn.functional.conv2d(x.to(dev), wt.to(dev), None, 2, 3).cpu().double()   # MIOpen
    scale = float(exact.abs().max())
    assert float((got - exact).abs().max()) <= 2.0 ** -9 * scale + 1e-6     # fp32 sums rounded once to fp16
    assert float((got - there).abs().max()) <= 2.0 ** -8 * scale + 1e-6


def test_conv7x7_limits(dev):
    wt = torch.zeros(8, 3 * 49, device=dev).half()
    with pytest.raises(ValueError):
        ops.conv7x7_s2(torch.zeros(1, 3, 7, 16, device=dev).half(), wt)    # odd height
    with pytest.raises(ValueError):
        ops.conv7x7_s2(torch.zeros(1, 3, 8, 24, device=dev).half(), wt)    # width not a multiple of 16
    with pytest.raises(TypeError):
        ops.conv7x7_s2(torch.zeros(1, 3, 8, 16, device=dev), wt)
    assert ops.conv7x7_s2(torch.zeros(0, 3, 8, 16, device=dev).half(), wt).shape == (0, 8, 4, 8)


def test_stem_module_uses_the_library_and_equals_torch(dev):
    torch.manual_seed(3)
    conv = layers.StemConv7x7(3, 64).to(dev).eval()
    x = torch.randint(-3, 4, (2, 3, 64, 64), device=dev).float()
    conv.weight.data = torch.randint(-2, 3, conv.weight.shape, device=dev).float()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        lib = conv(x)
        layers.set_library_forward(False)
        try:
            ref = conv(x)
        finally:
            layers.set_library_forward(True)
    assert lib.dtype == torch.float16 and torch.equal(lib, ref)
    assert conv.served_by_library(torch.float16, 64, 64) and not conv.served_by_library(torch.float16, 64, 40)
    with torch.no_grad():   # fp32 without autocast: torch's convolution
        assert conv(x).dtype == torch.float32
    enc = resnet.ResNet50Encoder() if hasattr(resnet, "ResNet50Encoder") else None
    assert enc is None or isinstance(enc.conv1, layers.StemConv7x7)
