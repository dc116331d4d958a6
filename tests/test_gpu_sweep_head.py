"""The sweep candidate's tail in one launch (spice_sweep_head_f16): up-sampling + KD-Lite loss +
arg-max histogram.  Checked against the three separate launches it replaces (interpolated logits
bit for bit, so the confusion matrix is identical; the loss differs only by the order in which
workgroups' fp64 partial sums are added) and against the CPU oracle (torch's interpolation in
fp32, first-maximum arg-max, histogram, fp64 loss)."""
import numpy as np
import pytest
import torch

from oracle import confusion as oc
from oracle import kdlite as ok
from oracle import upsample as ou
from spice_amd import _native as nat
from spice_amd import ops, synth

pytestmark = pytest.mark.gpu

REL = 1e-5  # tolerance of the north-star statement for the fp32 KD loss


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    nat.lib()
    return torch.device("cuda:0")


def _inputs(seed, n, c, h, w, scale, ignore_share=0.1):
    g = np.random.default_rng(seed)
    low = (g.standard_normal((n, c, h, w)) * 3).astype(np.float16)
    teacher = (g.standard_normal((n, c, h * scale, w * scale)) * 3).astype(np.float32)
    lab = synth.label_map(seed, n, h * scale, w * scale, c, "uniform")
    lab[g.random(lab.shape) < ignore_share] = 255
    return low, teacher, lab


def _three_launches(dev, low, teacher, lab, c, scale, align, kw):
    up = ops.upsample_bilinear(low, scale_factor=float(scale), align_corners=align, out_dtype=torch.float32)
    conf = torch.zeros(c, c, dtype=torch.int64, device=dev)
    ops.argmax_confusion_update(conf, up, lab, c)
    out3, _, _ = ops.kdlite_forward(up, teacher, lab, **kw)
    return conf, out3, up


CASES = [
    # n, c, h, w (source), scale, align_corners
    (8, 21, 128, 128, 4, True),     # the bench's shape: [8,21,512,512]
    (2, 21, 32, 32, 4, True),
    (2, 5, 24, 40, 4, False),       # W = 160: tiles straddle rows; class count below one step
    (1, 19, 16, 8, 2, True),        # x2, W = 16
    (3, 3, 8, 16, 8, False),
    (1, 6, 40, 24, 4, True),        # W = 96 does not divide a tile
    (2, 9, 16, 264, 4, True),       # long rows: W = 1056 > one tile
]


@pytest.mark.parametrize("n,c,h,w,scale,align", CASES)
def test_sweep_head_equals_three_launches(dev, n, c, h, w, scale, align):
    low, teacher, lab = _inputs(11, n, c, h, w, scale)
    kw = dict(temperature=4.0, alpha=0.7)
    low_d, t_d, lab_d = torch.from_numpy(low).to(dev), torch.from_numpy(teacher).to(dev), torch.from_numpy(lab).to(dev)
    ref_conf, ref_out3, _ = _three_launches(dev, low_d, t_d, lab_d, c, scale, align, kw)
    conf = torch.zeros(c, c, dtype=torch.int64, device=dev)
    for rep in (1, 2):   # the second call finds the tickets returned to zero and adds to the same matrix
        out3 = ops.sweep_head(low_d, t_d, lab_d, conf, align_corners=align, scale_factor=float(scale), **kw)
        assert out3 is not None, "the library must cover this shape"
        np.testing.assert_array_equal(conf.cpu().numpy(), rep * ref_conf.cpu().numpy())
        np.testing.assert_allclose(out3.cpu().numpy(), ref_out3.cpu().numpy(), rtol=1e-6, atol=0)
    # the teacher's softmax taken once beforehand: the same matrix, the same loss within fp32 rounding of the
    # per-pixel terms (another, algebraically equal, form of the KL sum)
    parts = ops.kd_teacher_prepare(t_d, kw["temperature"])
    assert parts is not None
    conf.zero_()
    out3 = ops.sweep_head(low_d, parts[0], lab_d, conf, align_corners=align, scale_factor=float(scale),
                          teacher_hneg=parts[1], **kw)
    assert out3 is not None
    np.testing.assert_array_equal(conf.cpu().numpy(), ref_conf.cpu().numpy())
    np.testing.assert_allclose(out3.cpu().numpy(), ref_out3.cpu().numpy(), rtol=REL, atol=0)


def test_teacher_parts_vs_float64(dev):
    n, c, h, w = 2, 21, 24, 40
    t = (np.random.default_rng(9).standard_normal((n, c, h, w)) * 4).astype(np.float32)
    probs, hneg = ops.kd_teacher_prepare(torch.from_numpy(t).to(dev), 4.0)
    z = t.astype(np.float64) / 4.0
    lp = z - z.max(1, keepdims=True)
    lp = lp - np.log(np.exp(lp).sum(1, keepdims=True))
    np.testing.assert_allclose(probs.cpu().numpy(), np.exp(lp), rtol=2e-6, atol=1e-9)
    np.testing.assert_allclose(hneg.cpu().numpy(), (np.exp(lp) * lp).sum(1), rtol=2e-6, atol=1e-6)
    assert ops.kd_teacher_prepare(torch.zeros(1, 3, 3, 3, device=dev), 4.0) is None   # H*W % 4 != 0


@pytest.mark.parametrize("n,c,h,w,scale,align", [(2, 21, 32, 32, 4, True), (2, 5, 24, 40, 4, False),
                                                  (1, 19, 16, 8, 2, True)])
def test_sweep_head_vs_oracle(dev, n, c, h, w, scale, align):
    low, teacher, lab = _inputs(5, n, c, h, w, scale)
    weights = torch.tensor(ok.FIVE_CLASS_WEIGHTS, dtype=torch.float32) if c == 5 else None
    up = ou.upsample_bilinear(torch.from_numpy(low), scale_factor=float(scale), align_corners=align).numpy()
    ref_conf = np.zeros((c, c), dtype=np.int64)
    oc.update_confusion(ref_conf, oc.argmax_first(up), lab, c)
    ref = ok.kdlite_loss_f64(up, teacher, lab, temperature=4.0, alpha=0.7, num_classes=c)
    conf = torch.zeros(c, c, dtype=torch.int64, device=dev)
    out3 = ops.sweep_head(torch.from_numpy(low).to(dev), torch.from_numpy(teacher).to(dev), torch.from_numpy(lab).to(dev),
                          conf, align_corners=align, scale_factor=float(scale), temperature=4.0, alpha=0.7,
                          class_weight=None if weights is None else weights.to(dev))
    assert out3 is not None
    got = out3.cpu().numpy().astype(np.float64)
    parts = ops.kd_teacher_prepare(torch.from_numpy(teacher).to(dev), 4.0)
    conf2 = torch.zeros(c, c, dtype=torch.int64, device=dev)
    out3p = ops.sweep_head(torch.from_numpy(low).to(dev), parts[0], torch.from_numpy(lab).to(dev), conf2,
                           align_corners=align, scale_factor=float(scale), temperature=4.0, alpha=0.7,
                           class_weight=None if weights is None else weights.to(dev), teacher_hneg=parts[1])
    assert torch.equal(conf, conf2)
    if weights is None:
        want = np.array([ref["total"], ref["ce"], ref["kd"]])
        np.testing.assert_allclose(got, want, rtol=REL)
        np.testing.assert_allclose(out3p.cpu().numpy().astype(np.float64), want, rtol=REL)
    # torch's CPU interpolation and the device kernel agree to the last bit on almost every value; an arg-max
    # can only differ where two interpolated logits are within one rounding of each other
    diff = np.abs(conf.cpu().numpy() - ref_conf).sum()
    assert diff <= 2 * max(1, int(1e-5 * lab.size)), diff


def test_sweep_head_ties_nan_and_infinities(dev):
    """First maximum wins on exact ties; a NaN is the maximum (first NaN wins); rows of -inf pick class 0;
    out-of-range labels are skipped by the histogram and counted by the loss."""
    n, c, h, w, scale = 2, 21, 16, 16, 4
    g = np.random.default_rng(3)
    low = (np.round(g.standard_normal((n, c, h, w)) * 2) / 2).astype(np.float16)   # many exact ties
    low[0, :, 3:6, 2:9] = 1.0                                                      # whole rows of ties -> class 0
    low[1, 7, 5, 5] = np.nan
    low[1, 12, 5, 5] = np.nan
    low[1, :, 10, 10] = -np.inf
    low[0, 4, 12, 3] = np.inf
    teacher = (g.standard_normal((n, c, h * scale, w * scale)) * 2).astype(np.float32)
    lab = synth.label_map(3, n, h * scale, w * scale, c, "uniform")
    lab[0, :4, :] = 255
    lab[1, 5, :7] = -3
    lab[1, 6, :7] = c + 2
    kw = dict(temperature=2.0, alpha=0.5)
    low_d, t_d, lab_d = torch.from_numpy(low).to(dev), torch.from_numpy(teacher).to(dev), torch.from_numpy(lab).to(dev)
    ref_conf, ref_out3, up = _three_launches(dev, low_d, t_d, lab_d, c, scale, True, kw)
    # the separate launches themselves against torch's arg-max on the same interpolated logits
    pred = up.argmax(1)
    want = np.zeros((c, c), dtype=np.int64)
    oc.update_confusion(want, pred.cpu().numpy(), lab, c)
    np.testing.assert_array_equal(ref_conf.cpu().numpy(), want)
    before = ops.kdlite_bad_label_count(dev)
    conf = torch.zeros(c, c, dtype=torch.int64, device=dev)
    out3 = ops.sweep_head(low_d, t_d, lab_d, conf, align_corners=True, scale_factor=float(scale), **kw)
    assert out3 is not None
    np.testing.assert_array_equal(conf.cpu().numpy(), want)
    a, b = out3.cpu().numpy(), ref_out3.cpu().numpy()
    assert np.array_equal(np.isnan(a), np.isnan(b))
    np.testing.assert_allclose(a[~np.isnan(a)], b[~np.isnan(b)], rtol=1e-6)
    assert ops.kdlite_bad_label_count(dev) - before == 14
    parts = ops.kd_teacher_prepare(t_d, kw["temperature"])
    conf.zero_()
    out3 = ops.sweep_head(low_d, parts[0], lab_d, conf, align_corners=True, scale_factor=float(scale),
                          teacher_hneg=parts[1], **kw)
    np.testing.assert_array_equal(conf.cpu().numpy(), want)
    a = out3.cpu().numpy()
    assert np.array_equal(np.isnan(a), np.isnan(b))
    np.testing.assert_allclose(a[~np.isnan(a)], b[~np.isnan(b)], rtol=REL)


def test_sweep_head_declines_what_it_does_not_cover(dev):
    c = 4
    t = torch.zeros(1, c, 12, 12, device=dev)
    lab = torch.zeros(1, 12, 12, dtype=torch.int64, device=dev)
    conf = torch.zeros(c, c, dtype=torch.int64, device=dev)
    low = torch.zeros(1, c, 3, 3, dtype=torch.float16, device=dev)   # in_w % 8 != 0
    assert ops.sweep_head(low, t, lab, conf, scale_factor=4.0, temperature=4.0, alpha=0.5) is None
    assert int(conf.sum()) == 0
    with pytest.raises((TypeError, RuntimeError)):
        ops.sweep_head(low.float(), t, lab, conf, scale_factor=4.0, temperature=4.0, alpha=0.5)
    with pytest.raises((ValueError, RuntimeError)):
        ops.sweep_head(low, t, lab[:, :5], conf, scale_factor=4.0, temperature=4.0, alpha=0.5)


def test_head_entries_decline_stages_near_the_lds_limit(dev):
    """Class counts and source widths whose LDS stage walks across the kernels' limits (the dynamic part plus the
    kernels' own static arrays must fit the 64 KiB a launch may ask for): every shape is either served, with the
    tables of the separate launches, or declined with SPICE_ERR_UNSUPPORTED (None here) - never a failed launch."""
    kw = dict(temperature=4.0, alpha=0.7)
    served = declined = 0
    for c in (17, 21, 24, 29):
        for in_w in range(8, 264, 8):
            h, scale = 6, 2
            low, teacher, lab = _inputs(c * 1000 + in_w, 1, c, h, in_w, scale)
            low_d, t_d, lab_d = (torch.from_numpy(x).to(dev) for x in (low, teacher, lab))
            ref_conf, ref_out3, _ = _three_launches(dev, low_d, t_d, lab_d, c, scale, False, kw)
            parts = ops.kd_teacher_prepare(t_d, kw["temperature"])
            for teacher_arg, extra in ((t_d, {}), (parts[0], {"teacher_hneg": parts[1]})):
                conf = torch.zeros(c, c, dtype=torch.int64, device=dev)
                out3 = ops.sweep_head(low_d, teacher_arg, lab_d, conf, align_corners=False, scale_factor=float(scale),
                                      **extra, **kw)
                if out3 is None:
                    declined += 1
                    assert int(conf.sum()) == 0
                    continue
                served += 1
                np.testing.assert_array_equal(conf.cpu().numpy(), ref_conf.cpu().numpy())
                np.testing.assert_allclose(out3.cpu().numpy(), ref_out3.cpu().numpy(), rtol=REL, atol=0)
            conf = torch.zeros(c, c, dtype=torch.int64, device=dev)
            if ops.upsample_argmax_confusion(conf, low_d, lab_d, scale_factor=float(scale), align_corners=False):
                np.testing.assert_array_equal(conf.cpu().numpy(), ref_conf.cpu().numpy())
    torch.cuda.synchronize()
    assert served > 0 and declined > 0


def test_sweep_uses_the_head_launch_and_gives_the_same_tables(dev):
    """SensitivitySweep with the one-launch head against the same sweep on the three separate launches:
    identical confusion matrices, losses to 1e-6."""
    import torch.nn as nn
    from spice_amd import nets
    from spice_amd.sweep import SensitivitySweep
    torch.manual_seed(0)
    model = nets.create_model("DeepLabV3Plus", "resnet50", 21).to(dev).eval()
    ignored = [m for m in model.segmentation_head.modules() if isinstance(m, nn.Conv2d)]
    sweep = SensitivitySweep(model, 21, ratios=(0.3,), ignored_layers=ignored)
    images = torch.randn(2, 3, 128, 128, device=dev)
    labels = torch.from_numpy(synth.label_map(1, 2, 128, 128, 21, "blocky")).to(dev)
    idx = list(range(0, len(sweep.candidates), max(1, len(sweep.candidates) // 5)))[:5]
    sweep.fused_head = True
    got = sweep.run(images, labels, indices=idx)
    assert sweep.head_launches == len(idx) and sweep._teacher_prepared is not None
    sweep.prepared_teacher = False
    mid = sweep.run(images, labels, indices=idx)
    assert sweep.head_launches == 2 * len(idx)
    np.testing.assert_array_equal(got["confusion"].cpu().numpy(), mid["confusion"].cpu().numpy())
    np.testing.assert_allclose(got["kd"].cpu().numpy(), mid["kd"].cpu().numpy(), rtol=1e-5, atol=1e-6)
    sweep.fused_head = False
    ref = sweep.run(images, labels, indices=idx)
    assert sweep.head_launches == 2 * len(idx)
    np.testing.assert_array_equal(got["confusion"].cpu().numpy(), ref["confusion"].cpu().numpy())
    # the un-pruned network is the teacher: the KL term is small, fp32 noise of the per-pixel terms around it
    np.testing.assert_allclose(mid["kd"].cpu().numpy(), ref["kd"].cpu().numpy(), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(got["kd"].cpu().numpy(), ref["kd"].cpu().numpy(), rtol=1e-5, atol=1e-6)
    assert int(got["confusion"][idx[0]].sum()) == int((labels != 255).sum())


def test_sweep_run_again_on_another_batch(dev):
    """A second run() on a new batch whose tensor may land on the first one's address (same shape, same
    version): the cast of the batch and the prepared teacher must be those of the new batch."""
    import torch.nn as nn
    from spice_amd import nets
    from spice_amd.sweep import SensitivitySweep
    torch.manual_seed(1)
    model = nets.create_model("DeepLabV3Plus", "resnet50", 21).to(dev).eval()
    ignored = [m for m in model.segmentation_head.modules() if isinstance(m, nn.Conv2d)]
    labels = torch.from_numpy(synth.label_map(4, 2, 128, 128, 21, "blocky")).to(dev)
    sweep = SensitivitySweep(model, 21, ratios=(0.5,), ignored_layers=ignored)
    idx = [0, len(sweep.candidates) // 2, len(sweep.candidates) - 1]
    images = torch.randn(2, 3, 128, 128, device=dev)
    first = sweep.run(images, labels, indices=idx)
    del images
    images = torch.randn(2, 3, 128, 128, device=dev) * 2 + 1
    again = sweep.run(images, labels, indices=idx)
    fresh = SensitivitySweep(model, 21, ratios=(0.5,), ignored_layers=ignored).run(images, labels, indices=idx)
    assert not torch.equal(first["confusion"], again["confusion"])
    assert torch.equal(again["confusion"], fresh["confusion"])
    np.testing.assert_allclose(again["kd"].cpu().numpy(), fresh["kd"].cpu().numpy(), rtol=1e-5, atol=1e-6)


def test_raw_ctypes_binding_of_the_head_entry_points(dev):
    """spice_kd_teacher_prepare and spice_sweep_head_f16 called as a maintainer's ctypes stub would call them (no
    spice_amd.ops, no torch extension): argument errors are return codes, an uncovered shape is
    SPICE_ERR_UNSUPPORTED with nothing written, and the results are the oracle's."""
    import ctypes
    import os
    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                   "spice_amd", "lib", "libspice_amd.so"))

    class Params(ctypes.Structure):
        _fields_ = [("temperature", ctypes.c_float), ("alpha", ctypes.c_float), ("boundary_weight", ctypes.c_float),
                    ("confidence_weight", ctypes.c_int32), ("ignore_index", ctypes.c_int64),
                    ("class_weight", ctypes.c_void_p)]

    V, I64, F = ctypes.c_void_p, ctypes.c_int64, ctypes.c_float
    lib.spice_kdlite_workspace_bytes.restype = ctypes.c_size_t
    lib.spice_kdlite_workspace_bytes.argtypes = [I64] * 4
    lib.spice_confusion_workspace_bytes.restype = ctypes.c_size_t
    lib.spice_confusion_workspace_bytes.argtypes = [I64, ctypes.c_int32]
    lib.spice_kd_teacher_prepare.restype = ctypes.c_int32
    lib.spice_kd_teacher_prepare.argtypes = [V, I64, I64, I64, I64, F, V, V, V]
    lib.spice_sweep_head_f16.restype = ctypes.c_int32
    lib.spice_sweep_head_f16.argtypes = [V, V, V, V, I64, I64, I64, I64, I64, I64, ctypes.c_int32, F, F,
                                         ctypes.POINTER(Params), V, V, V, ctypes.c_size_t, V, ctypes.c_size_t, V]
    n, c, h, w, scale = 2, 21, 16, 24, 4
    low, teacher, lab = _inputs(21, n, c, h, w, scale)
    H, W = h * scale, w * scale
    low_d, t_d, lab_d = torch.from_numpy(low).to(dev), torch.from_numpy(teacher).to(dev), torch.from_numpy(lab).to(dev)
    kd_ws = torch.zeros(lib.spice_kdlite_workspace_bytes(n, c, H, W), dtype=torch.uint8, device=dev)
    conf_ws = torch.zeros(lib.spice_confusion_workspace_bytes(n * H * W, c), dtype=torch.uint8, device=dev)
    probs, hneg = torch.empty_like(t_d), torch.empty(n, H, W, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    assert lib.spice_kd_teacher_prepare(t_d.data_ptr(), n, c, H, W, 4.0, probs.data_ptr(), hneg.data_ptr(), stream) == 0
    params = Params(4.0, 0.7, 1.0, 0, 255, None)

    def call(teacher_ptr, hneg_ptr, low_t=low_d, in_w=w):
        out3 = torch.full((3,), -7.0, device=dev)
        conf = torch.zeros(c, c, dtype=torch.int64, device=dev)
        rc = lib.spice_sweep_head_f16(low_t.data_ptr(), teacher_ptr, hneg_ptr, lab_d.data_ptr(), n, c, h, in_w, H, W, 1,
                                      float(scale), float(scale), ctypes.byref(params), out3.data_ptr(), conf.data_ptr(),
                                      kd_ws.data_ptr(), kd_ws.numel(), conf_ws.data_ptr(), conf_ws.numel(), stream)
        return rc, out3.cpu().numpy().astype(np.float64), conf.cpu().numpy()

    up = ou.upsample_bilinear(torch.from_numpy(low), scale_factor=float(scale), align_corners=True).numpy()
    ref = ok.kdlite_loss_f64(up, teacher, lab, temperature=4.0, alpha=0.7, num_classes=c)
    want = np.array([ref["total"], ref["ce"], ref["kd"]])
    want_conf = np.zeros((c, c), dtype=np.int64)
    oc.update_confusion(want_conf, oc.argmax_first(up), lab, c)
    for rc, out3, conf in (call(t_d.data_ptr(), None), call(probs.data_ptr(), hneg.data_ptr())):
        assert rc == 0
        np.testing.assert_allclose(out3, want, rtol=REL)
        assert np.abs(conf - want_conf).sum() <= 2
    # a source width that is no multiple of 8: declined, nothing written
    rc, out3, conf = call(t_d.data_ptr(), None, in_w=w - 1)
    assert rc == 10002 and (out3 == -7.0).all() and conf.sum() == 0
    # argument errors are return codes
    rc = lib.spice_sweep_head_f16(None, None, None, None, n, c, h, w, H, W, 1, 4.0, 4.0, ctypes.byref(params), None, None,
                                  None, 0, None, 0, None)
    assert rc != 0
    assert lib.spice_kd_teacher_prepare(None, n, c, H, W, 4.0, None, None, None) != 0
    small = torch.zeros(lib.spice_kdlite_workspace_bytes(n, c, H, W) - 16, dtype=torch.uint8, device=dev)
    out3 = torch.zeros(3, device=dev)
    conf = torch.zeros(c, c, dtype=torch.int64, device=dev)
    rc = lib.spice_sweep_head_f16(low_d.data_ptr(), t_d.data_ptr(), None, lab_d.data_ptr(), n, c, h, w, H, W, 1, 4.0, 4.0,
                                  ctypes.byref(params), out3.data_ptr(), conf.data_ptr(), small.data_ptr(), small.numel(),
                                  conf_ws.data_ptr(), conf_ws.numel(), stream)
    assert rc != 0 and conf.sum().item() == 0   # workspace too small


@pytest.mark.parametrize("n,c,h,w,scale,align", CASES)
def test_upsample_argmax_confusion_equals_two_launches(dev, n, c, h, w, scale, align):
    """The evaluation loops' form of the kernel (no teacher, no loss): identical to up-sampling followed by the
    fused arg-max + histogram launch, on exact ties too."""
    low, _, lab = _inputs(13, n, c, h, w, scale)
    low = (np.round(low.astype(np.float32)) / 2).astype(np.float16)   # many exact ties between classes
    low[0, :, 1, 1:3] = np.nan if c > 4 else low[0, :, 1, 1:3]
    low_d, lab_d = torch.from_numpy(low).to(dev), torch.from_numpy(lab).to(dev)
    up = ops.upsample_bilinear(low_d, scale_factor=float(scale), align_corners=align, out_dtype=torch.float32)
    want = torch.zeros(c, c, dtype=torch.int64, device=dev)
    ops.argmax_confusion_update(want, up, lab_d, c)
    conf = torch.zeros(c, c, dtype=torch.int64, device=dev)
    for rep in (1, 2):
        assert ops.upsample_argmax_confusion(conf, low_d, lab_d, align_corners=align, scale_factor=float(scale))
        assert torch.equal(conf, rep * want)
    assert int(conf.sum()) == 2 * int(((lab >= 0) & (lab < c)).sum())
    # declined shapes launch nothing
    odd = torch.zeros(1, c, 3, 3, dtype=torch.float16, device=dev)
    assert not ops.upsample_argmax_confusion(conf, odd, torch.zeros(1, 12, 12, dtype=torch.int64, device=dev),
                                             scale_factor=4.0)
    assert torch.equal(conf, 2 * want)


def test_evaluation_loops_with_the_upsampling_inside_the_histogram_launch(dev):
    """inference() and SegmentationMetrics.update_from_model on a DeepLabV3+ (head = convolution -> x4 bilinear
    up-sampling): the same results with the up-sampling inside the histogram launch as with the model's own forward
    followed by the fused arg-max + histogram launch."""
    from spice_amd import nets
    import importlib
    inf = importlib.import_module("spice_amd.baseline.eval.inference")
    bm = importlib.import_module("spice_amd.baseline.eval.metrics")
    from spice_amd.pruning.eval.metrics import SegmentationMetrics
    torch.manual_seed(5)
    c = 5
    model = nets.create_model("DeepLabV3Plus", "resnet50", c).to(dev).eval()
    # 128 x 128: every convolution of the forward is one of the library's kernels, which give the same values on every
    # call (MIOpen may pick another algorithm for the same problem on a later call)
    loader = [(torch.randn(2, 3, 128, 128), torch.from_numpy(synth.label_map(k, 2, 128, 128, c, "blocky"))) for k in range(3)]
    calls = []
    real = ops.upsample_argmax_confusion
    try:
        ops.upsample_argmax_confusion = lambda *a, **k: calls.append(1) or real(*a, **k)
        bm.FUSED_UPSAMPLE = True
        got = inf.inference(model, loader, dev, num_classes=c, shard=False)
        assert len(calls) == len(loader), "the one-launch route must have been taken"
        sm = SegmentationMetrics(c, device=dev)
        with torch.no_grad(), torch.amp.autocast("cuda"):
            for x, y in loader:
                sm.update_from_model(model, x.to(dev), y.to(dev))
        bm.FUSED_UPSAMPLE = False
        ref = inf.inference(model, loader, dev, num_classes=c, shard=False)
        assert len(calls) == 2 * len(loader)
        sm2 = SegmentationMetrics(c, device=dev)
        with torch.no_grad(), torch.amp.autocast("cuda"):
            for x, y in loader:
                sm2.update_from_logits(model(x.to(dev)), y.to(dev))
    finally:
        ops.upsample_argmax_confusion = real
        bm.FUSED_UPSAMPLE = True
    assert got == ref
    assert torch.equal(sm.conf, sm2.conf) and int(sm.conf.sum()) > 0
