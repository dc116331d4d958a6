"""The 3x3 convolution + batch-norm (+ ReLU) in one launch of the matrix-core kernel (spice_conv3x3_bn_act_f16): nine
shifted 1x1 products in one sum.  On integer-valued fp16 data every product and partial sum is exact, so the result
must equal an exact convolution followed by the batch-norm launch BIT FOR BIT (this pins the tap order of the weights,
the shifts, the image edges, stride and dilation).  On random data: within an fp16 step of fp64 sums and of torch's
device convolution (MIOpen)."""
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


def _norm(c, g, dev):
    bn = torch.nn.BatchNorm2d(c).eval()
    bn.running_mean.copy_(torch.randn(c, generator=g))
    bn.running_var.copy_(torch.rand(c, generator=g) * 2 + 0.05)
    bn.weight.data.copy_(torch.rand(c, generator=g) + 0.5)
    bn.bias.data.copy_(torch.randn(c, generator=g))
    return bn.to(dev)


def _taps(w):
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous()


# n, cin, h, w, cout, stride, dilation
CASES = [(2, 16, 8, 16, 24, 1, 1), (1, 64, 16, 16, 64, 1, 1), (1, 40, 8, 8, 130, 1, 1), (3, 23, 4, 8, 7, 1, 1),
         (2, 32, 16, 16, 48, 1, 2), (1, 8, 8, 24, 5, 1, 3), (1, 16, 8, 8, 16, 1, 7), (2, 16, 8, 16, 24, 2, 1),
         (1, 64, 32, 32, 64, 2, 1), (2, 21, 6, 32, 9, 2, 1), (4, 64, 64, 64, 64, 1, 1), (2, 256, 32, 32, 256, 1, 1),
         (1, 128, 16, 16, 300, 1, 2),
         # the eight-wave forms: 256-channel tiles (two 128-channel tiles per compute unit), and 128-channel tiles with a K tail
         (8, 32, 32, 32, 512, 1, 1), (4, 40, 32, 32, 256, 1, 2),
         # ... and 128-pixel tiles (deep K, many tiles), M tail
         (8, 32, 64, 64, 130, 1, 1),
         # one round of 128-pixel tiles: the sixteen-wave form; M tail, K tail, dilation
         (8, 32, 64, 64, 128, 1, 1), (6, 40, 64, 64, 100, 1, 2),
         # the patch form (x staged once per 64 input channels, nine taps out of it; taken from 512 input channels on):
         # dilation 2 on 32 x 32; an M tail; rows that do not fill the last tile, dilation 3; one 128-pixel row per tile
         (2, 512, 32, 32, 512, 1, 2), (1, 512, 16, 16, 200, 1, 1), (1, 576, 10, 64, 130, 1, 3), (1, 512, 5, 128, 96, 1, 1),
         # the form for at most 32 output channels (4 x 128-pixel tiles of whole patches, blocks of 16 or 32 input
         # channels): columns and rows that do not fill the last tile; several blocks; one tile row; 21 and 32 channels
         (1, 32, 12, 136, 16, 1, 1), (2, 128, 8, 256, 32, 1, 1), (1, 16, 9, 512, 21, 1, 1), (1, 48, 6, 128, 20, 1, 1),
         (3, 64, 3, 264, 32, 1, 1), (2, 16, 64, 64, 1, 1, 1)]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("relu", [True, False])
def test_conv3x3_exact_on_integer_data(dev, case, relu):
    n, cin, h, w, cout, stride, dil = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randint(-3, 4, (n, cin, h, w), generator=g).half()
    wt = torch.randint(-2, 3, (cout, cin, 3, 3), generator=g).half()
    exact = torch.nn.functional.conv2d(x.double(), wt.double(), None, stride, dil, dil).half().contiguous().to(dev)
    bn = _norm(cout, g, dev)
    with torch.no_grad():
        got = ops.conv3x3_bn_act(x.to(dev), _taps(wt).to(dev), bn, relu, stride, dil)
        want = ops.bn_act(exact, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps, relu=relu)
    assert got.shape == want.shape and got.dtype == torch.float16
    assert torch.equal(got, want)


@pytest.mark.parametrize("case", [(2, 64, 32, 32, 64, 1, 1), (2, 128, 32, 64, 128, 2, 1), (1, 256, 32, 32, 512, 1, 2),
                                  (8, 64, 128, 128, 64, 1, 1), (2, 512, 32, 32, 512, 1, 2), (2, 32, 256, 256, 16, 1, 1),
                                  (1, 128, 64, 128, 32, 1, 1)])
def test_conv3x3_on_random_data(dev, case):
    n, cin, h, w, cout, stride, dil = case
    g = torch.Generator().manual_seed(sum(case) + 3)
    x = torch.randn(n, cin, h, w, generator=g).half().to(dev)
    wt = (torch.randn(cout, cin, 3, 3, generator=g) / (9 * cin) ** 0.5).half().to(dev)
    bn = torch.nn.BatchNorm2d(cout).eval().to(dev)   # identity statistics: the product itself, up to bn's own rounding
    with torch.no_grad():
        got = ops.conv3x3_bn_act(x, _taps(wt), bn, False, stride, dil)
        there = torch.nn.functional.conv2d(x, wt, None, stride, dil, dil)
        exact = torch.nn.functional.conv2d(x.double(), wt.double(), None, stride, dil, dil)
    scale = float(exact.abs().max())
    assert float((got.double() - exact).abs().max()) <= 2.0 ** -9 * scale + 1e-6
    assert float((got.float() - there.float()).abs().max()) <= 2.0 ** -8 * scale + 1e-6


def test_conv3x3_limits_and_bottleneck(dev):
    g = torch.Generator().manual_seed(1)
    bn = _norm(8, g, dev)
    wt = torch.randn(8, 4, 3, 3, generator=g).half().to(dev)
    x = torch.randn(1, 4, 8, 12, generator=g).half().to(dev)      # 12 columns: rows are not whole 16-byte pieces
    assert not ops.conv3x3_supported(x, 1, 1) and ops.conv3x3_supported(x[..., :8].contiguous(), 1, 1)
    with torch.no_grad():
        with pytest.raises(RuntimeError):
            ops.conv3x3_bn_act(x, _taps(wt), bn, True, 1, 1)
        with pytest.raises(TypeError):
            ops.conv3x3_bn_act(x.float(), _taps(wt), bn, True, 1, 1)
        with pytest.raises(ValueError):
            ops.conv3x3_bn_act(x, _taps(wt)[:, :27].contiguous(), bn, True, 1, 1)
        with pytest.raises(ValueError):
            ops.conv3x3_bn_act(x, _taps(wt), bn, True, 3, 1)
        assert ops.conv3x3_bn_act(x[:0, :, :, :8].contiguous(), _taps(wt), bn, True, 1, 1).shape == (0, 8, 8, 8)
    # a bottleneck with its 3x3 convolution on the library's kernel against the same block on torch's route (a hook on
    # conv2 sends it there): MIOpen sums in another order (Winograd), hence a few fp16 steps
    from spice_amd.nets.resnet import Bottleneck
    torch.manual_seed(6)
    for stride, dilation in ((1, 1), (2, 1), (1, 2)):
        blk = Bottleneck(64, 16, stride=stride, dilation=dilation, downsample=True).to(dev).eval()
        for m in blk.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0.0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
        xin = torch.randn(2, 64, 16, 32, device=dev)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            a = blk(xin)
            handle = blk.conv2.register_forward_hook(lambda m, i, o: None)
            b = blk(xin)
            handle.remove()
        assert a.shape == b.shape
        assert float((a.float() - b.float()).abs().max()) <= 2.0 ** -7 * float(b.float().abs().max())


_FORCED = """
import sys, torch
sys.path.insert(0, %r)
from spice_amd import ops
dev = torch.device("cuda:0")
for n, cin, h, w, cout, dil in ((2, 64, 16, 128, 64, 1), (1, 128, 8, 32, 40, 2), (2, 192, 12, 64, 70, 1), (1, 64, 4, 8, 16, 7)):
    g = torch.Generator().manual_seed(n + cin + h + w + cout + dil)
    x = torch.randint(-3, 4, (n, cin, h, w), generator=g).half()
    wt = torch.randint(-2, 3, (cout, cin, 3, 3), generator=g).half()
    exact = torch.nn.functional.conv2d(x.double(), wt.double(), None, 1, dil, dil).half().contiguous().to(dev)
    bn = torch.nn.BatchNorm2d(cout).eval()
    bn.running_mean.copy_(torch.randn(cout, generator=g)); bn.running_var.copy_(torch.rand(cout, generator=g) * 2 + 0.05)
    bn = bn.to(dev)
    with torch.no_grad():
        got = ops.conv3x3_bn_act(x.to(dev), wt.permute(0, 2, 3, 1).reshape(cout, -1).contiguous().to(dev), bn, True, 1, dil)
        want = ops.bn_act(exact, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps, relu=True)
    assert torch.equal(got, want), (n, cin, h, w, cout, dil)
print("ok")
"""


def test_conv3x3_narrow_form_forced_on_few_tiles():
    """The form for at most 32 output channels on layers it leaves to the general kernel by itself (deep sums on
    fewer tiles than compute units): blocks of 32 input channels, several of them, M tails: exact on integer data."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = _FORCED.replace("((2, 64, 16, 128, 64, 1), (1, 128, 8, 32, 40, 2), (2, 192, 12, 64, 70, 1), (1, 64, 4, 8, 16, 7))",
                             "((2, 128, 8, 256, 32, 1), (1, 96, 6, 136, 21, 1), (1, 256, 5, 16, 7, 1), (3, 160, 4, 128, 32, 1))")
    assert script != _FORCED
    env = dict(os.environ, SPICE_CONV3X3_NARROW="2")
    out = subprocess.run([sys.executable, "-c", script % root], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr


@pytest.mark.parametrize("form", ["2", "3"])
def test_conv3x3_patch_form_forced_on_small_layers(form):
    """Both tile shapes of the patch form on layers it is not taken for by itself (SPICE_CONV3X3_HALO_FORM is read
    once per process, hence a process of its own): exact on integer data."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, SPICE_CONV3X3_HALO_FORM=form)
    out = subprocess.run([sys.executable, "-c", _FORCED % root], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
