"""The one-launch 1x1 convolution + batch-norm tail (matrix-core kernel, spice_conv1x1_bn_act_f16) against the
two-launch route it replaces and against fp64 sums.

On integer-valued fp16 data every product and every fp32 partial sum is exact, so the stored product does not
depend on the order of the sums: there the fused kernel must equal the two launches BIT FOR BIT (this pins the
operand layouts, the tile edges and the tail).  On random data the two routes may round the product to different
fp16 neighbours where fp32 sums of another order land on the other side of a tie: a bounded number of elements,
each within the tail's amplification of one fp16 step."""
import ctypes

import pytest
import torch

from spice_amd import _native as nat
from spice_amd import ops

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    nat.lib()
    return torch.device("cuda:0")


def _norm(c, g, dev, eps=1e-5):
    bn = torch.nn.BatchNorm2d(c, eps=eps).eval()
    bn.running_mean.copy_(torch.randn(c, generator=g))
    bn.running_var.copy_(torch.rand(c, generator=g) * 2 + 0.05)
    bn.weight.data.copy_(torch.rand(c, generator=g) + 0.5)
    bn.bias.data.copy_(torch.randn(c, generator=g))
    return bn.to(dev)


# n, cin, h, w, cout: whole tiles; K, M and pixel tails; a K that is not a multiple of 8 (unaligned weight rows);
# tiny; deep K (the variant with three K steps in flight: K >= 256 on few tiles), also with a K tail and with
# unaligned weight rows; a tile count that is not a multiple of 8; several row tiles
SHAPES = [(2, 64, 16, 16, 256), (1, 40, 8, 24, 130), (3, 230, 4, 6, 77), (1, 8, 2, 4, 5), (2, 2048, 32, 32, 512),
          (3, 96, 8, 40, 300), (1, 33, 16, 8, 64), (8, 256, 32, 32, 48), (1, 328, 8, 8, 40), (2, 301, 4, 8, 140),
          (1, 640, 8, 16, 129),
          # many tiles and more than 128 channels; with an M tail, a K tail, unaligned rows
          (8, 64, 64, 64, 300), (8, 72, 64, 64, 256), (8, 33, 64, 64, 257),
          # deep K on two 128-channel tiles per compute unit: the eight-wave form with 256-channel tiles, K tail
          (8, 264, 32, 32, 512),
          # deep K on many tiles: the eight-wave form with 128-pixel tiles; M tail and K tail
          (8, 264, 64, 64, 300), (8, 512, 32, 32, 512),
          # deep K on one round of 128-pixel tiles (192 and 256 of them): the sixteen-wave form; M tail and K tail
          (8, 264, 64, 64, 100), (6, 512, 32, 32, 512),
          # at most 64 output channels on whole 128-pixel tiles: the 64-channel form; one K step, K tail + M tail
          (8, 64, 64, 64, 64), (2, 200, 16, 16, 33)]


def _integer_case(shape, dev):
    n, cin, h, w, cout = shape
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randint(-3, 4, (n, cin, h, w), generator=g).half()
    wt = torch.randint(-2, 3, (cout, cin), generator=g).half()
    exact = torch.einsum("oc,nchw->nohw", wt.double(), x.double())
    return g, x.to(dev), wt.to(dev), exact.half().contiguous().to(dev)   # |sums| <= 6 * 2048: exact in fp32, one rounding to fp16


@pytest.mark.parametrize("shape", SHAPES)
def test_fused_plain_product_is_exact_on_integer_data(dev, shape):
    """mean == NULL through the raw C ABI: the product alone, every tile edge."""
    n, cin, h, w, cout = shape
    _, x, wt, product = _integer_case(shape, dev)
    y = torch.full((n, cout, h, w), float("nan"), dtype=torch.float16, device=dev)
    guard = torch.full((4096,), 7.0, dtype=torch.float16, device=dev)   # allocated right behind y: must stay untouched
    null = ctypes.c_void_p(0)
    rc = nat.lib().spice_conv1x1_bn_act_f16(x.data_ptr(), wt.data_ptr(), y.data_ptr(), n, cin, h * w, cout, null, null,
                                            null, null, 0.0, null, null, null, null, null, 0.0, 0, 0,
                                            null, null, null, null, 0.0, 0,
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.equal(y, product)
    assert bool((guard == 7.0).all())
    assert torch.equal(ops.conv1x1_f16(x, wt), product)   # the hipBLASLt product is exact on this data as well


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("mode", ["bn", "bn_relu", "res_relu", "dual_relu", "dual"])
def test_fused_equals_two_launches_on_integer_data(dev, shape, mode):
    n, cin, h, w, cout = shape
    g, x, wt, product = _integer_case(shape, dev)
    bn = _norm(cout, g, dev)
    res = (2.0 * torch.randn(n, cout, h, w, generator=g)).half().to(dev) if mode != "bn" and mode != "bn_relu" else None
    rbn = _norm(cout, g, dev, eps=1e-3) if mode.startswith("dual") else None
    relu = mode.endswith("relu")
    with torch.no_grad():
        got = ops.conv1x1_bn_act(x, wt, bn, res, rbn, relu, route=1)
        two = ops.conv1x1_bn_act(x, wt, bn, res, rbn, relu, route=2)
        want = ops.bn_act_dual(product, bn, res, rbn, relu=relu) if rbn is not None else \
            ops.bn_act(product, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps, residual=res, relu=relu)
    assert got.dtype == torch.float16 and got.shape == (n, cout, h, w)
    assert torch.equal(got, want)
    assert torch.equal(two, want)


@pytest.mark.parametrize("shape", [(2, 64, 16, 16, 256), (8, 256, 128, 128, 64), (3, 230, 4, 6, 77), (2, 2048, 32, 32, 512),
                                   (2, 512, 32, 32, 2048), (8, 304, 128, 128, 256)])
def test_fused_on_random_data(dev, shape):
    """Against fp64 sums of the fp16 values through the oracle's tail, and against the two-launch route."""
    from oracle import bnact as ob
    n, cin, h, w, cout = shape
    g = torch.Generator().manual_seed(sum(shape) + 5)
    x = torch.randn(n, cin, h, w, generator=g).half()
    wt = (torch.randn(cout, cin, generator=g) / cin ** 0.5).half()
    res = torch.randn(n, cout, h, w, generator=g).half()
    bn = _norm(cout, g, "cpu")
    exact = torch.einsum("oc,nchw->nohw", wt.double(), x.double()).half().contiguous()
    want = ob.bn_act(exact, bn.running_mean, bn.running_var, bn.weight.data, bn.bias.data, bn.eps, residual=res, relu=True)
    bn = bn.to(dev)
    with torch.no_grad():
        got = ops.conv1x1_bn_act(x.to(dev), wt.to(dev), bn, res.to(dev), None, True, route=1)
        two = ops.conv1x1_bn_act(x.to(dev), wt.to(dev), bn, res.to(dev), None, True, route=2)
    # one fp16 step of the product (|product| <~ 5) times the largest scale of the norm (1.5 / sqrt(0.05) = 6.7), plus the
    # steps of the two roundings of the tail
    bound = 2.0 ** -9 * 6.7 + 2.0 ** -10 * float(want.float().abs().max()) * 2
    assert float((got.float().cpu() - want.float()).abs().max()) <= bound
    assert float((got.float() - two.float()).abs().max()) <= bound
    assert float((got != two).float().mean()) < 0.02   # the routes differ only where a sum sits on a rounding tie


def test_fused_rejects_what_it_cannot_take(dev):
    x = torch.randn(1, 8, 2, 3, device=dev).half()      # 6 pixels per plane: not a multiple of 8
    wt = torch.randn(4, 8, device=dev).half()
    bn = _norm(4, torch.Generator().manual_seed(0), dev)
    with torch.no_grad():
        with pytest.raises(RuntimeError):
            ops.conv1x1_bn_act(x, wt, bn, None, None, True, route=1)
        two = ops.conv1x1_bn_act(x, wt, bn, None, None, True, route=2)
        auto = ops.conv1x1_bn_act(x, wt, bn, None, None, True)     # falls back to the two launches by itself
        assert torch.equal(auto, two)
        with pytest.raises(TypeError):
            ops.conv1x1_bn_act(x.float(), wt, bn, None, None, True)
        with pytest.raises(ValueError):
            ops.conv1x1_bn_act(x, wt[:, :3].contiguous(), bn, None, None, True)
        with pytest.raises(ValueError):
            ops.conv1x1_bn_act(x, wt, bn, None, None, True, route=5)
        assert ops.conv1x1_bn_act(x[:0], wt, bn, None, None, True).shape == (0, 4, 2, 3)
    # through the C ABI: a weight matrix that does not start on 4 bytes (the base of a buffer descriptor) is refused
    x = torch.randn(1, 8, 2, 4, device=dev).half()
    flat = torch.randn(4 * 8 + 1, device=dev).half()
    y = torch.zeros(1, 4, 2, 4, device=dev).half()
    null = ctypes.c_void_p(0)
    rc = nat.lib().spice_conv1x1_bn_act_f16(x.data_ptr(), flat.data_ptr() + 2, y.data_ptr(), 1, 8, 8, 4, null, null, null, null,
                                            0.0, null, null, null, null, null, 0.0, 0, 0, null, null, null, null, 0.0, 0,
                                            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == nat.ERR_UNSUPPORTED


@pytest.mark.parametrize("route", [1, 2])
def test_fused_writes_into_a_channel_slice(dev, route):
    """out= a channel slice of a concatenation buffer: the slice holds what a separate tensor would, its neighbours stay."""
    shape = (3, 40, 8, 24, 50)
    n, cin, h, w, cout = shape
    g, x, wt, _ = _integer_case(shape, dev)
    bn = _norm(cout, g, dev)
    res = torch.randn(n, cout, h, w, generator=g).half().to(dev)
    both = torch.full((n, 16 + cout + 24, h, w), 3.0, dtype=torch.float16, device=dev)
    with torch.no_grad():
        want = ops.conv1x1_bn_act(x, wt, bn, res, None, True, route=route)
        got = ops.conv1x1_bn_act(x, wt, bn, res, None, True, route=route, out=both[:, 16:16 + cout])
    assert got.data_ptr() == both[:, 16:16 + cout].data_ptr()
    assert torch.equal(both[:, 16:16 + cout], want)
    assert bool((both[:, :16] == 3.0).all()) and bool((both[:, 16 + cout:] == 3.0).all())
    with torch.no_grad(), pytest.raises(ValueError):
        ops.conv1x1_bn_act(x, wt, bn, res, None, True, out=both[:, :cout, :, ::2])
    with torch.no_grad(), pytest.raises(TypeError):
        ops.conv1x1_bn_act(x, wt, bn, res, None, True, out=both.float()[:, :cout])


@pytest.mark.parametrize("shape", [(2, 64, 16, 32, 96), (1, 40, 8, 16, 130), (3, 230, 4, 16, 77), (8, 256, 128, 128, 512),
                                   (1, 512, 16, 32, 64), (2, 8, 2, 16, 5)])
def test_stride2_product(dev, shape):
    """The 1x1 convolution of stride 2 (projection shortcuts): exact on integer data against the product over the
    even pixels, and within an fp16 step of torch's strided convolution on random data."""
    n, cin, h, w, cout = shape
    g = torch.Generator().manual_seed(sum(shape) + 2)
    x = torch.randint(-3, 4, (n, cin, h, w), generator=g).half()
    wt = torch.randint(-2, 3, (cout, cin), generator=g).half()
    exact = torch.einsum("oc,nchw->nohw", wt.double(), x[:, :, ::2, ::2].double()).half().contiguous()
    got = ops.conv1x1_f16(x.to(dev), wt.to(dev), stride=2)
    assert got.shape == (n, cout, h // 2, w // 2) and got.dtype == torch.float16 and torch.equal(got.cpu(), exact)
    x = torch.randn(n, cin, h, w, generator=g).half().to(dev)
    wt = (torch.randn(cout, cin, generator=g) / cin ** 0.5).half().to(dev)
    got = ops.conv1x1_f16(x, wt, stride=2)
    there = torch.nn.functional.conv2d(x, wt.view(cout, cin, 1, 1), stride=2)
    scale = float(there.float().abs().max())
    assert float((got.float() - there.float()).abs().max()) <= 2.0 ** -9 * scale + 1e-6


def test_stride2_module_and_limits(dev):
    from spice_amd.nets.layers import Conv1x1
    conv = Conv1x1(16, 24, stride=2).to(dev)
    x = torch.randn(2, 16, 8, 32, device=dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        assert conv.served_by_library(x.dtype, 8, 32) and not conv.served_by_library(x.dtype, 8, 24)
        got = conv(x)
        handle = conv.register_forward_hook(lambda m, i, o: None)   # does not change the route of a plain module call
        again = conv(x)
        handle.remove()
        there = torch.nn.functional.conv2d(x.half(), conv.weight.half(), stride=2)
        odd = conv(torch.randn(2, 16, 8, 24, device=dev))   # width not a multiple of 16: torch's convolution
    assert got.dtype == torch.float16 and torch.equal(got, again)
    assert float((got.float() - there.float()).abs().max()) <= 2.0 ** -9 * float(there.float().abs().max()) + 1e-6
    assert odd.shape == (2, 24, 4, 12)
    with pytest.raises(ValueError):
        ops.conv1x1_f16(x.half(), conv.weight.half().view(24, 16), stride=3)
    with pytest.raises(ValueError):
        ops.conv1x1_f16(x.half()[..., :24].contiguous(), conv.weight.half().view(24, 16), stride=2)


@pytest.mark.parametrize("shape", [(2, 64, 16, 16, 256), (1, 40, 8, 24, 130), (3, 230, 4, 6, 77), (8, 128, 64, 64, 512),
                                   (2, 512, 32, 32, 2048), (1, 328, 8, 8, 40)])
@pytest.mark.parametrize("pre_relu", [True, False])
def test_pending_input_batch_norm_rides_along(dev, shape, pre_relu):
    """x raw, its batch-norm (+ ReLU) applied by the kernel on the way in: bit for bit what the same kernel gives on
    the tensor a batch-norm launch has stored (same values in, same order of sums), on random data."""
    n, cin, h, w, cout = shape
    g = torch.Generator().manual_seed(sum(shape) + 9)
    x = (2.0 * torch.randn(n, cin, h, w, generator=g)).half().to(dev)
    wt = (torch.randn(cout, cin, generator=g) / cin ** 0.5).half().to(dev)
    res = torch.randn(n, cout, h, w, generator=g).half().to(dev)
    bn, pbn = _norm(cout, g, dev), _norm(cin, g, dev, eps=1e-3)
    with torch.no_grad():
        stored = ops.bn_act(x, pbn.running_mean, pbn.running_var, pbn.weight, pbn.bias, pbn.eps, relu=pre_relu)
        want = ops.conv1x1_bn_act(stored, wt, bn, res, None, True, route=1)
        got = ops.conv1x1_bn_act(x, wt, bn, res, None, True, route=1, pre_bn=pbn, pre_relu=pre_relu)
        two = ops.conv1x1_bn_act(x, wt, bn, res, None, True, route=2, pre_bn=pbn, pre_relu=pre_relu)
        two_want = ops.conv1x1_bn_act(stored, wt, bn, res, None, True, route=2)
    assert torch.equal(got, want)
    assert torch.equal(two, two_want)


def test_pending_input_batch_norm_limits_and_bottleneck(dev):
    g = torch.Generator().manual_seed(4)
    x = torch.randn(1, 520, 4, 8, generator=g).half().to(dev)       # more input channels than the kernel holds parameters for
    wt = (torch.randn(24, 520, generator=g) / 23.0).half().to(dev)
    bn, pbn = _norm(24, g, dev), _norm(520, g, dev)
    with torch.no_grad():
        with pytest.raises(RuntimeError):
            ops.conv1x1_bn_act(x, wt, bn, None, None, True, route=1, pre_bn=pbn, pre_relu=True)
        auto = ops.conv1x1_bn_act(x, wt, bn, None, None, True, pre_bn=pbn, pre_relu=True)    # separate launches by itself
        stored = ops.bn_act(x, pbn.running_mean, pbn.running_var, pbn.weight, pbn.bias, pbn.eps, relu=True)
        assert torch.equal(auto, ops.conv1x1_bn_act(stored, wt, bn, None, None, True, route=2))
    # a bottleneck on the library path against the same block with a hook on bn2 (which sends bn2 through its module)
    from spice_amd.nets.resnet import Bottleneck
    torch.manual_seed(6)
    blk = Bottleneck(64, 16, stride=1, downsample=False).to(dev).eval()
    for m in blk.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0.0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
    xin = torch.randn(2, 64, 16, 24, device=dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        a = blk(xin)
        handle = blk.bn2.register_forward_hook(lambda m, i, o: None)
        b = blk(xin)
        handle.remove()
    # the 3x3 convolution in between is MIOpen's: the same algorithm twice in a row on one input in practice, but
    # not by contract, hence a tolerance of two fp16 steps instead of equality
    assert float((a.float() - b.float()).abs().max()) <= 2.0 ** -9 * float(b.float().abs().max())
