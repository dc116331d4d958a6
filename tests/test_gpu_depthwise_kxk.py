"""The tiled depthwise k x k kernel (EfficientNet blocks): exact on integer data, fp64 sums on random data, the
module against torch's convolution, what the entry point declines."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from spice_amd import _native as nat
from spice_amd import ops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    return torch.device("cuda:0")


CASES = [(2, 5, 16, 16), (1, 3, 34, 70), (2, 4, 64, 64), (1, 2, 130, 256), (1, 7, 9, 18), (3, 6, 33, 32), (1, 2, 2, 2)]


@pytest.mark.parametrize("k", [3, 5])
@pytest.mark.parametrize("stride", [1, 2])
@pytest.mark.parametrize("dilation", [1, 2])
@pytest.mark.parametrize("shape", CASES)
def test_depthwise_kxk_exact_on_integer_data(dev, k, stride, dilation, shape):
    """Small integers: every product and every partial sum is exact in fp32 and in fp16, whatever the order - the
    kernel must give torch's CPU convolution bit for bit, image edges, tile edges and odd output widths included;
    fp32 and fp16 weights alike."""
    n, c, h, w = shape
    g = torch.Generator().manual_seed(k * 100 + stride * 10 + dilation + h)
    x = torch.randint(-4, 5, shape, generator=g).float()
    wt = torch.randint(-3, 4, (c, 1, k, k), generator=g).float()
    want = F.conv2d(x, wt, None, stride, (k // 2) * dilation, dilation, groups=c).half()
    assert ops.depthwise_conv_supported(x.half().to(dev), wt.to(dev), stride, dilation)
    for wdev in (wt.to(dev), wt.half().to(dev)):
        got = ops.depthwise_conv(x.half().to(dev), wdev, stride, dilation)
        assert got.dtype == torch.float16 and got.shape == want.shape
        assert torch.equal(got.cpu(), want)


@pytest.mark.parametrize("k,stride,dilation,shape", [(3, 2, 1, (8, 144, 64, 64)), (5, 2, 1, (4, 192, 64, 64)),
                                                      (5, 1, 1, (4, 288, 32, 32)), (5, 1, 2, (2, 816, 32, 32)),
                                                      (3, 1, 2, (2, 1392, 32, 32)), (3, 1, 1, (2, 40, 128, 128))])
def test_depthwise_kxk_random_data_against_fp64_and_torch(dev, k, stride, dilation, shape):
    """Random data at the block shapes of the B3 encoder: within one fp16 step of the fp64 sums, and of torch's own
    device convolution."""
    n, c, h, w = shape
    g = torch.Generator().manual_seed(k + stride + dilation)
    x = torch.randn(shape, generator=g).half()
    wt = (torch.randn(c, 1, k, k, generator=g) / k).half()
    exact = F.conv2d(x.double(), wt.double(), None, stride, (k // 2) * dilation, dilation, groups=c)
    got = ops.depthwise_conv(x.to(dev), wt.to(dev), stride, dilation)
    step = 1e-3 * max(1.0, float(exact.abs().max()))
    assert float((got.cpu().double() - exact).abs().max()) <= step
    there = F.conv2d(x.to(dev), wt.to(dev), None, stride, (k // 2) * dilation, dilation, groups=c)
    assert float((got.float() - there.float()).abs().max()) <= 2 * step


def test_efficientnet_blocks_take_the_library_depthwise_kernel(dev):
    """The encoder under fp16 autocast with and without the library forward: every feature map within fp16 rounding
    of torch's modules; the depthwise module falls back to torch's convolution where the kernel does not apply."""
    from spice_amd.nets import layers
    from spice_amd.nets.efficientnet import EfficientNetB3Encoder
    torch.manual_seed(5)
    enc = EfficientNetB3Encoder(output_stride=16).to(dev).eval()
    assert sum(isinstance(m, layers.DepthwiseConvKxK) for m in enc.modules()) == 26
    x = torch.randn(2, 3, 128, 128, device=dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        lib = enc(x)
        layers.set_library_forward(False)
        try:
            ref = enc(x)
        finally:
            layers.set_library_forward(True)
    for a, b in zip(lib[1:], ref[1:]):
        assert a.shape == b.shape and a.dtype == b.dtype
        scale = max(1.0, float(b.float().abs().max()))
        assert float((a.float() - b.float()).abs().max()) <= 2e-2 * scale
    dw = layers.DepthwiseConvKxK(6, 5, stride=2).to(dev)
    odd = torch.randn(1, 6, 9, 9, device=dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        y = dw(odd)    # odd width: torch's convolution
    assert y.shape == (1, 6, 5, 5)
    with torch.autocast("cuda", dtype=torch.float16):
        assert dw(torch.randn(1, 6, 8, 8, device=dev)).requires_grad   # gradients wanted: torch's convolution


def test_depthwise_kxk_entry_point_declines(dev):
    L = nat.lib()
    x = torch.zeros(1, 2, 8, 8, dtype=torch.float16, device=dev)
    w = torch.zeros(2, 1, 7, 7, dtype=torch.float16, device=dev)
    y = torch.zeros(1, 2, 8, 8, dtype=torch.float16, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream

    def call(k, s, d, wd=8):
        return L.spice_depthwise_conv_f16(x.data_ptr(), w.data_ptr(), 1, y.data_ptr(), 1, 2, 8, wd, k, s, d,
                                          None, None, None, None, 0.0, 0, st)
    assert call(7, 1, 1) == nat.ERR_UNSUPPORTED and call(3, 3, 1) == nat.ERR_UNSUPPORTED and call(3, 1, 4) == nat.ERR_UNSUPPORTED
    assert call(3, 1, 1, wd=7) == nat.ERR_UNSUPPORTED
    assert call(3, 1, 1) == 0
    assert not ops.depthwise_conv_supported(x[:, :, :, :7].contiguous(), w[:, :, :3, :3].contiguous(), 1, 1)
    with pytest.raises(RuntimeError):
        ops.depthwise_conv(x, w, 1, 1)


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
@pytest.mark.parametrize("with_res", [False, True])
def test_bn_act_with_silu_tail(dev, dtype, with_res):
    """silu(bn(x) [+ residual]) in one launch against torch's batch-norm, add and SiLU kernels: the same rounding points
    (batch-norm and sum rounded to the dtype before the activation), so at most one step of the dtype apart (the
    exponential and the reciprocal are the hardware's 1-ulp instructions), and equal for nearly all elements."""
    g = torch.Generator().manual_seed(7)
    c = 24
    x = (3.0 * torch.randn(4, c, 16, 20, generator=g)).to(dtype).to(dev)
    res = (2.0 * torch.randn(4, c, 16, 20, generator=g)).to(dtype).to(dev) if with_res else None
    mean, var = torch.randn(c, generator=g).to(dev), (torch.rand(c, generator=g) * 2 + 0.05).to(dev)
    weight, bias = (torch.rand(c, generator=g) + 0.5).to(dev), torch.randn(c, generator=g).to(dev)
    got = ops.bn_act(x, mean, var, weight, bias, 1e-5, residual=res, act=2)
    there = F.batch_norm(x, mean, var, weight, bias, False, 0.0, 1e-5)
    if with_res:
        there = there + res
    there = F.silu(there)
    assert got.dtype == dtype
    # (fp32: the three 1-ulp steps - exponential, reciprocal, product - add up to a few units of the last place)
    tol = (2.0 ** -10 if dtype == torch.float16 else 2.0 ** -20) * there.float().abs().clamp_min(2.0 ** -14)
    assert bool(((got.float() - there.float()).abs() <= tol).all())
    assert float((got == there).float().mean()) > (0.99 if dtype == torch.float16 else 0.5)
    # the ReLU code of the same argument still is ReLU
    assert torch.equal(ops.bn_act(x, mean, var, weight, bias, 1e-5, act=1), ops.bn_act(x, mean, var, weight, bias, 1e-5, relu=True))


@pytest.mark.parametrize("cin,cout,hw,with_res", [(40, 240, (32, 32), False), (240, 40, (32, 32), True),
                                                    (136, 816, (16, 16), False), (33, 17, (8, 8), False)])
def test_conv1x1_pair_with_silu_tail(dev, cin, cout, hw, with_res):
    """The matrix-core pair kernel with the SiLU tail (EfficientNet's expansion) and with a plain residual behind the
    batch-norm (its projection + skip): the SiLU of the kernel's own ReLU-free result, element for element."""
    from spice_amd.nets import layers
    torch.manual_seed(cin + cout)
    conv = layers.Conv1x1(cin, cout).to(dev)
    bn = layers.BatchNormAct2d(cout, silu=not with_res).to(dev).eval()
    plain = layers.BatchNormAct2d(cout).to(dev).eval()
    for m in (bn, plain):
        m.running_mean.copy_(torch.linspace(-0.5, 0.5, cout)); m.running_var.copy_(torch.linspace(0.5, 1.5, cout))
        m.weight.data.copy_(torch.linspace(0.7, 1.3, cout)); m.bias.data.copy_(torch.linspace(-0.3, 0.3, cout))
    x = torch.randn(2, cin, *hw, device=dev).half()
    res = torch.randn(2, cout, *hw, device=dev).half() if with_res else None
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        got = layers.conv1x1_bn(conv, bn, x, res)
        base = layers.conv1x1_bn(conv, plain, x, res)
    want = base if with_res else F.silu(base)
    assert got.dtype == torch.float16 and got.shape == want.shape
    tol = 2.0 ** -10 * want.float().abs().clamp_min(2.0 ** -14)
    assert bool(((got.float() - want.float()).abs() <= tol).all())
    if with_res:
        assert torch.equal(got, want)


@pytest.mark.parametrize("c,r,hw", [(40, 10, (64, 64)), (144, 6, (32, 32)), (816, 34, (16, 16)), (24, 6, (5, 7))])
def test_squeeze_excite_on_the_library_kernels(dev, c, r, hw):
    """The block's forward through the library against torch's modules under fp16 autocast (mean, two convolutions
    through MIOpen, SiLU, sigmoid, product): the gate within a few fp16 steps, the product with it; and against the
    fp64 value of the same expression."""
    from spice_amd.nets import layers
    from spice_amd.nets.efficientnet import SqueezeExcite
    torch.manual_seed(c + r)
    se = SqueezeExcite(c, r).to(dev).eval()
    se.conv_reduce.bias.data.normal_(0.0, 0.3); se.conv_expand.bias.data.normal_(0.0, 0.3)
    x = torch.randn(3, c, *hw, device=dev).half()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        assert se.served_by_library(x.dtype)
        got = se(x)
        layers.set_library_forward(False)
        try:
            assert not se.served_by_library(x.dtype)
            there = se(x)
        finally:
            layers.set_library_forward(True)
    xd = x.double()
    s = xd.mean((2, 3))
    h = torch.nn.functional.silu(s @ se.conv_reduce.weight.double().reshape(r, c).t() + se.conv_reduce.bias.double())
    g = torch.sigmoid(h @ se.conv_expand.weight.double().reshape(c, r).t() + se.conv_expand.bias.double())
    exact = xd * g[:, :, None, None]
    assert got.dtype == torch.float16 and got.shape == x.shape
    scale = float(exact.abs().max())
    assert float((got.double() - exact).abs().max()) <= 6e-3 * scale
    assert float((got.float() - there.float()).abs().max()) <= 6e-3 * scale
    with torch.autocast("cuda", dtype=torch.float16):
        assert se(x.requires_grad_(True)).requires_grad   # gradients wanted: torch's modules


@pytest.mark.parametrize("k,stride,dilation,shape,act", [(3, 1, 1, (2, 40, 64, 64), 2), (5, 2, 1, (2, 144, 32, 32), 2),
                                                          (5, 1, 2, (1, 816, 16, 16), 2), (3, 2, 1, (2, 24, 20, 36), 1),
                                                          (3, 1, 1, (1, 8, 8, 8), 0)])
def test_depthwise_kxk_with_the_batch_norm_in_the_same_launch(dev, k, stride, dilation, shape, act):
    """Batch-norm (+ ReLU / SiLU) applied to the rounded convolution result before the store: bit for bit the
    convolution launch followed by spice_bn_act with the same tail code; and through the modules."""
    from spice_amd.nets import layers
    n, c, h, w = shape
    g = torch.Generator().manual_seed(k + stride + c)
    x = torch.randn(shape, generator=g).half().to(dev)
    conv = layers.DepthwiseConvKxK(c, k, stride, dilation).to(dev)
    bn = layers.BatchNormAct2d(c, relu=act == 1, silu=act == 2).to(dev).eval()
    bn.running_mean.copy_(torch.randn(c, generator=g)); bn.running_var.copy_(torch.rand(c, generator=g) + 0.5)
    bn.weight.data.copy_(torch.rand(c, generator=g) + 0.5); bn.bias.data.copy_(torch.randn(c, generator=g))
    plain = ops.depthwise_conv(x, conv.weight.detach(), stride, dilation)
    want = ops.bn_act(plain, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps, act=act)
    got = ops.depthwise_conv(x, conv.weight.detach(), stride, dilation, bn, act)
    assert torch.equal(got, want)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        assert torch.equal(layers.depthwise_bn(conv, bn, x), want)
        assert torch.equal(bn(conv(x)), want)
