"""Parity of the HIP kernels (called through the C ABI) with the CPU oracle and the golden
fixtures generated from the reference.  Bars: bit-exact for confusion matrices and gathered
tensors; relative error <= 1e-5 (fp32) for importance scores and the KD-Lite loss."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import confusion as oc
from oracle import importance as oi
from oracle import kdlite as ok
from oracle import selection as osel
from spice_amd import _native as nat
from spice_amd import ops, synth

pytestmark = pytest.mark.gpu

REL = 1e-5  # tolerance of the north-star statement for fp32 importance scores and KD loss


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    nat.lib()
    return torch.device("cuda:0")


# ---------------------------------------------------------------------------------- confusion
CF = load_golden("confusion.npz")


@pytest.mark.parametrize("name", [str(x) for x in CF["names"]])
def test_confusion_golden(dev, name):
    c = int(CF[f"{name}/num_classes"][0])
    conf = torch.zeros(c, c, dtype=torch.int64, device=dev)
    for k in range(2):
        ops.confusion_update(conf, torch.from_numpy(CF[f"{name}/pred{k}"]).to(dev),
                             torch.from_numpy(CF[f"{name}/target{k}"]).to(dev), c)
        np.testing.assert_array_equal(conf.cpu().numpy(), CF[f"{name}/conf{k}"])


@pytest.mark.parametrize("c,shape,kind", [(21, (8, 512, 512), "uniform"), (21, (8, 512, 512), "blocky"),
                                          (19, (3, 333, 517), "uniform"), (2, (1, 1, 1), "uniform"),
                                          (64, (2, 256, 256), "uniform"), (130, (1, 64, 64), "uniform")])
def test_confusion_vs_oracle(dev, c, shape, kind):
    n, h, w = shape
    lab = synth.label_map(7, n, h, w, c, kind)
    pred = synth.predictions_from(lab, 7, c)
    ref = np.zeros((c, c), dtype=np.int64)
    oc.update_confusion(ref, pred, lab, c)
    conf = torch.zeros(c, c, dtype=torch.int64, device=dev)
    ops.confusion_update(conf, torch.from_numpy(pred).to(dev), torch.from_numpy(lab).to(dev), c)
    np.testing.assert_array_equal(conf.cpu().numpy(), ref)
    assert int(conf.sum()) == int(((lab >= 0) & (lab < c)).sum())
    # in-place accumulation: a second update doubles every bin
    ops.confusion_update(conf, torch.from_numpy(pred).to(dev), torch.from_numpy(lab).to(dev), c)
    np.testing.assert_array_equal(conf.cpu().numpy(), 2 * ref)


def test_confusion_unaligned_and_empty(dev):
    c = 5
    lab = synth.label_map(9, 1, 37, 41, c, "uniform")
    pred = synth.predictions_from(lab, 9, c)
    tl = torch.from_numpy(np.concatenate([[0], lab.reshape(-1)])).to(dev)[1:]   # 8-byte aligned only
    tp = torch.from_numpy(np.concatenate([[0], pred.reshape(-1)])).to(dev)[1:]
    ref = np.zeros((c, c), dtype=np.int64)
    oc.update_confusion(ref, pred, lab, c)
    conf = torch.zeros(c, c, dtype=torch.int64, device=dev)
    ops.confusion_update(conf, tp.contiguous() if not tp.is_contiguous() else tp, tl, c)
    np.testing.assert_array_equal(conf.cpu().numpy(), ref)
    empty = torch.zeros(0, dtype=torch.int64, device=dev)
    ops.confusion_update(conf, empty, empty, c)
    np.testing.assert_array_equal(conf.cpu().numpy(), ref)


def test_confusion_counts_out_of_range_predictions(dev):
    c = 4
    before = ops.invalid_prediction_count(dev, c)
    t = torch.tensor([0, 1, 2, 3, 255, -1], dtype=torch.int64, device=dev)
    p = torch.tensor([0, 9, -2, 3, 7, 7], dtype=torch.int64, device=dev)
    conf = torch.zeros(c, c, dtype=torch.int64, device=dev)
    ops.confusion_update(conf, p, t, c)
    assert conf.cpu().tolist() == [[1, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 0], [0, 0, 0, 1]]
    assert ops.invalid_prediction_count(dev, c) - before == 2


def test_confusion_config5_size_properties(dev):
    """[32,1024,2048] int64, 19 classes: checked against the oracle on the full input."""
    c, n, h, w = 19, 32, 1024, 2048
    lab = synth.label_map(11, n, h, w, c, "uniform", ignore_share=0.10)
    pred = synth.predictions_from(lab, 11, c)
    conf = torch.zeros(c, c, dtype=torch.int64, device=dev)
    ops.confusion_update(conf, torch.from_numpy(pred).to(dev), torch.from_numpy(lab).to(dev), c)
    ref = np.zeros((c, c), dtype=np.int64)
    oc.update_confusion(ref, pred, lab, c)
    np.testing.assert_array_equal(conf.cpu().numpy(), ref)
    assert int(conf.sum()) == int((lab != 255).sum())


def _config5_logits(dev, dtype, n, c, h, w):
    """Device logits in steps of 1/2 (many exact ties: the first maximum must win), and the oracle's first-maximum
    arg-max of the same values, image by image on the host."""
    g = torch.Generator(device=dev).manual_seed(5)
    logits = torch.empty(n, c, h, w, device=dev, dtype=dtype)
    for i in range(n):   # image by image: no second full-size temporary
        logits[i] = (torch.randn(c, h, w, device=dev, generator=g) * 3).mul_(2).round_().div_(2).to(dtype)
    pred = np.empty((n, h, w), dtype=np.int64)
    for i in range(n):
        pred[i] = oc.argmax_first(logits[i].float().cpu().numpy()[None])[0]
    return logits, pred


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32])
def test_argmax_confusion_config5_size(dev, dtype):
    """The evaluation loops' fused entry point at the size of the 1024x2048, 19-class, batch-32 evaluation
    (baseline/train_test_quant.py:205-229): [32,19,1024,2048] logits - 2.5 GB as fp16, past 2^31 bytes, 5.1 GB as
    fp32 - bit-exact against the oracle's arg-max + histogram and against confusion_update on torch's arg-max."""
    c, n, h, w = 19, 32, 1024, 2048
    lab = synth.label_map(12, n, h, w, c, "uniform", ignore_share=0.10)
    logits, pred = _config5_logits(dev, dtype, n, c, h, w)
    assert logits.numel() * logits.element_size() > 2 ** 31
    ref = np.zeros((c, c), dtype=np.int64)
    oc.update_confusion(ref, pred, lab, c)
    lab_d = torch.from_numpy(lab).to(dev)
    conf = torch.zeros(c, c, dtype=torch.int64, device=dev)
    ops.argmax_confusion_update(conf, logits, lab_d, c)
    np.testing.assert_array_equal(conf.cpu().numpy(), ref)
    two_step = torch.zeros(c, c, dtype=torch.int64, device=dev)
    ops.confusion_update(two_step, logits.argmax(1), lab_d, c)
    np.testing.assert_array_equal(two_step.cpu().numpy(), ref)
    assert int(conf.sum()) == int((lab != 255).sum())
    # the last image alone (its logits start past 2^31 bytes): a wrapped offset would read another image
    last = torch.zeros(c, c, dtype=torch.int64, device=dev)
    ops.argmax_confusion_update(last, logits[n - 1:], lab_d[n - 1:], c)
    ref_last = np.zeros((c, c), dtype=np.int64)
    oc.update_confusion(ref_last, pred[n - 1:], lab[n - 1:], c)
    np.testing.assert_array_equal(last.cpu().numpy(), ref_last)


def test_upsample_argmax_confusion_config5_size(dev):
    """[32,19,256,512] fp16 head logits -> 1024x2048 inside the histogram launch: bit-exact against the two
    launches it replaces (up-sampling kernel, then arg-max histogram), against confusion_update on torch's
    arg-max of the up-sampled logits and against the oracle's histogram of that arg-max."""
    c, n, h, w = 19, 32, 1024, 2048
    lab = synth.label_map(13, n, h, w, c, "blocky", ignore_share=0.10)
    lab_d = torch.from_numpy(lab).to(dev)
    g = torch.Generator(device=dev).manual_seed(6)
    low = (torch.randn(n, c, h // 4, w // 4, device=dev, generator=g) * 3).half()
    conf = torch.zeros(c, c, dtype=torch.int64, device=dev)
    assert ops.upsample_argmax_confusion(conf, low, lab_d, align_corners=True, scale_factor=4.0), \
        "the one-launch form must cover the 1024x2048 evaluation"
    full = ops.upsample_bilinear(low, scale_factor=4.0, align_corners=True, out_dtype=torch.float32)
    assert full.numel() * 4 > 2 ** 31
    two = torch.zeros(c, c, dtype=torch.int64, device=dev)
    ops.argmax_confusion_update(two, full, lab_d, c)
    assert torch.equal(conf, two)
    pred = full.argmax(1)
    three = torch.zeros(c, c, dtype=torch.int64, device=dev)
    ops.confusion_update(three, pred, lab_d, c)
    assert torch.equal(conf, three)
    ref = np.zeros((c, c), dtype=np.int64)
    oc.update_confusion(ref, pred.cpu().numpy(), lab, c)
    np.testing.assert_array_equal(conf.cpu().numpy(), ref)
    assert int(conf.sum()) == int((lab != 255).sum())


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 21, 64, 64), (1, 5, 15, 17), (1, 19, 3, 8)])
def test_argmax_confusion_vs_oracle(dev, dtype, shape):
    n, c, h, w = shape
    s, _ = synth.logits_pair(3, n, c, h, w)
    s = np.round(s * 2) / 2  # many exact ties: the first maximum must win
    lab = synth.label_map(3, n, h, w, c, "uniform")
    logits = torch.from_numpy(s).to(dtype)
    pred = oc.argmax_first(logits.float().numpy())
    ref = np.zeros((c, c), dtype=np.int64)
    oc.update_confusion(ref, pred, lab, c)
    conf = torch.zeros(c, c, dtype=torch.int64, device=dev)
    ops.argmax_confusion_update(conf, logits.to(dev), torch.from_numpy(lab).to(dev), c)
    np.testing.assert_array_equal(conf.cpu().numpy(), ref)


# ---------------------------------------------------------------------------------- KD-Lite
KD = load_golden("kdlite.npz")


def _kd_kwargs(name):
    T, alpha, bw, conf, c = KD[f"{name}/params"]
    return dict(temperature=float(T), alpha=float(alpha), boundary_weight=float(bw),
                confidence_weight=bool(conf)), int(c)


def _class_weight(c, dev):
    return torch.tensor(ok.FIVE_CLASS_WEIGHTS, dtype=torch.float32, device=dev) if c == 5 else None


def _close(got, ref, what):
    if np.isnan(ref):
        assert np.isnan(got), what
    else:
        assert abs(got - ref) <= REL * abs(ref) + 1e-6 * (abs(ref) < 1e-5), (what, got, ref)


@pytest.mark.parametrize("name", [str(x) for x in KD["names"]])
@pytest.mark.parametrize("want_grad", [False, True])
def test_kdlite_golden(dev, name, want_grad):
    kw, c = _kd_kwargs(name)
    s = torch.from_numpy(KD[f"{name}/student"]).to(dev)
    t = torch.from_numpy(KD[f"{name}/teacher"]).to(dev)
    lab = torch.from_numpy(KD[f"{name}/labels"]).to(dev)
    out3, grad, _ = ops.kdlite_forward(s, t, lab, class_weight=_class_weight(c, dev),
                                       want_grad=want_grad, **kw)
    got = out3.cpu().numpy()
    ref = KD[f"{name}/out3"]
    # the exact value (fp64) is the judge where the reference's own fp32 rounding is visible
    exact = ok.kdlite_loss_f64(KD[f"{name}/student"], KD[f"{name}/teacher"], KD[f"{name}/labels"],
                               num_classes=c, **kw)
    for k, key in enumerate(("total", "ce", "kd")):
        if name == "c21_same_logits" and key != "ce":
            assert abs(got[k] - exact[key]) <= 1e-6, (key, got[k], exact[key])  # kd ~ 1e-7 in the reference
            continue
        _close(got[k], ref[k], f"{name}:{key} vs reference")
        if not np.isnan(ref[k]):
            assert abs(got[k] - exact[key]) <= REL * abs(exact[key]) + 1e-7
    if want_grad and np.isfinite(ref[0]):
        g_ref = KD[f"{name}/grad"]
        err = np.abs(grad.cpu().numpy() - g_ref).max()
        assert err <= REL * np.abs(g_ref).max(), (name, err, np.abs(g_ref).max())


@pytest.mark.parametrize("variant", ["default", "boundary_conf"])
def test_kdlite_config3_size(dev, variant):
    """[8,21,512,512], T=4, alpha=0.7: HIP vs the fp32 op-sequence oracle and the fp64 value."""
    n, c, h, w = 8, 21, 512, 512
    bad_before = ops.kdlite_bad_label_count(dev)   # (a count per process: tests before this one may have fed bad labels)
    s, t = synth.logits_pair(1234, n, c, h, w)
    lab = synth.label_map(1234, n, h, w, c, "blocky" if variant != "default" else "uniform")
    kw = dict(temperature=4.0, alpha=0.7)
    if variant != "default":
        kw.update(boundary_weight=2.0, confidence_weight=True)
    out3, grad, sums = ops.kdlite_forward(torch.from_numpy(s).to(dev), torch.from_numpy(t).to(dev),
                                          torch.from_numpy(lab).to(dev), want_grad=True, **kw)
    sums = sums.clone()  # a view of the per-stream workspace: the next call overwrites it
    got = out3.cpu().numpy()
    ref, g_ref = ok.kdlite_loss_and_grad(torch.from_numpy(s), torch.from_numpy(t), torch.from_numpy(lab),
                                         num_classes=c, **kw)
    for k, key in enumerate(("total", "ce", "kd")):
        assert abs(got[k] - ref[key].item()) <= REL * abs(ref[key].item()), (key, got[k], ref[key].item())
    # fp64 on a quarter of the batch keeps the check inside a few seconds
    exact = ok.kdlite_loss_f64(s[:2], t[:2], lab[:2], num_classes=c, **kw)
    out3q, _, _ = ops.kdlite_forward(torch.from_numpy(s[:2]).to(dev), torch.from_numpy(t[:2]).to(dev),
                                     torch.from_numpy(lab[:2]).to(dev), **kw)
    for k, key in enumerate(("total", "ce", "kd")):
        assert abs(out3q[k].item() - exact[key]) <= REL * abs(exact[key])
    g = grad.cpu()
    assert (g - g_ref).abs().max().item() <= REL * g_ref.abs().max().item()
    assert int(sums[1].item()) == int((lab != 255).sum())
    assert ops.kdlite_bad_label_count(dev) == bad_before


@pytest.mark.parametrize("c,hw,variant", [(24, (20, 28), "default"), (25, (20, 28), "default"), (40, (9, 13), "boundary_conf"),
                                          (2, (7, 5), "default"), (24, (6, 10), "boundary_conf"), (1, (8, 8), "default"),
                                          (8, (12, 16), "default"), (8, (12, 16), "boundary_conf"), (9, (12, 16), "default"),
                                          (5, (36, 36), "boundary_conf"), (3, (4, 4), "default"), (7, (10, 6), "default")])
def test_kdlite_gradient_both_sides_of_the_lds_limit(dev, c, hw, variant):
    """The gradient pass keeps the logits in registers up to 8 classes (16-byte layouts), parks them in
    LDS up to 24 and re-reads them above that: every variant against the op-sequence oracle, aligned
    and odd geometries, 1 and 2 classes."""
    h, w = hw
    g = torch.Generator().manual_seed(100 + c)
    s = 3.0 * torch.randn(2, c, h, w, generator=g)
    t = s + torch.randn(2, c, h, w, generator=g)
    lab = torch.randint(0, c, (2, h, w), generator=g)
    lab[torch.rand(2, h, w, generator=g) < 0.1] = 255
    kw = dict(temperature=3.0, alpha=0.6)
    if variant != "default":
        kw.update(boundary_weight=2.5, confidence_weight=True)
    cw = torch.tensor(ok.FIVE_CLASS_WEIGHTS).to(dev) if c == 5 else None   # the reference weights exactly five classes
    out3, grad, _ = ops.kdlite_forward(s.to(dev), t.to(dev), lab.to(dev), want_grad=True, class_weight=cw, **kw)
    ref, g_ref = ok.kdlite_loss_and_grad(s, t, lab, num_classes=c, **kw)
    for k, key in enumerate(("total", "ce", "kd")):
        assert abs(out3[k].item() - ref[key].item()) <= REL * abs(ref[key].item()) + 1e-7, (key, out3[k].item(), ref[key].item())
    assert (grad.cpu() - g_ref).abs().max().item() <= REL * g_ref.abs().max().item() + 1e-9


def test_kdlite_coefficients_and_upstream(dev):
    kw, c = _kd_kwargs("c21_default")
    s = torch.from_numpy(KD["c21_default/student"]).to(dev)
    t = torch.from_numpy(KD["c21_default/teacher"]).to(dev)
    lab = torch.from_numpy(KD["c21_default/labels"]).to(dev)
    _, g_total, _ = ops.kdlite_forward(s, t, lab, want_grad=True, **kw)
    _, g_ce, _ = ops.kdlite_forward(s, t, lab, want_grad=True, coef=(1.0, 0.0), **kw)
    _, g_kd, _ = ops.kdlite_forward(s, t, lab, want_grad=True, coef=(0.0, 1.0), **kw)
    a = kw["alpha"]
    torch.testing.assert_close(g_total, a * g_ce + (1 - a) * g_kd, rtol=1e-5, atol=1e-9)
    up = torch.tensor([2.0, 0.5, -1.0], device=dev)
    _, g_up, _ = ops.kdlite_forward(s, t, lab, want_grad=True, upstream=up, **kw)
    torch.testing.assert_close(g_up, (2 * a + 0.5) * g_ce + (2 * (1 - a) - 1.0) * g_kd, rtol=1e-5, atol=1e-9)
    one = torch.ones(1, device=dev)
    before = g_total.clone()
    ops.scale_inplace(g_total, one)
    assert torch.equal(g_total, before)
    ops.scale_inplace(g_total, torch.full((1,), 3.0, device=dev))
    torch.testing.assert_close(g_total, 3 * before)


def test_kdlite_counts_out_of_range_labels(dev):
    s = torch.zeros(1, 3, 2, 2, device=dev)
    lab = torch.tensor([[[0, 7], [255, 2]]], dtype=torch.int64, device=dev)
    before = ops.kdlite_bad_label_count(dev)
    ops.kdlite_forward(s, s.clone(), lab, temperature=4.0, alpha=0.5)
    assert ops.kdlite_bad_label_count(dev) - before == 1


# ---------------------------------------------------------------------------------- importance
def _device_pairs(pairs, dev):
    return ([torch.from_numpy(w).to(dev) for w, _ in pairs], [torch.from_numpy(g).to(dev) for _, g in pairs])


@pytest.mark.parametrize("mode,p", [(nat.IMP_L1, 1), (nat.IMP_L2, 2)])
def test_magnitude_importance_resnet50_shapes(dev, mode, p):
    shapes = synth.deeplab_resnet50_conv_shapes(21)
    pairs = synth.conv_pairs(1234, shapes)
    ws, _ = _device_pairs(pairs, dev)
    flat, offs = ops.channel_importance(ws, mode)
    flat = flat.cpu().numpy()
    for (w, _), off in zip(pairs, offs):
        got = flat[off: off + w.shape[0]]
        exact = oi.magnitude_rows_f64(w, p)
        ref = oi.magnitude_rows(torch.from_numpy(w), p).numpy()
        assert np.abs(got - exact).max() <= REL * np.abs(exact).max()
        np.testing.assert_allclose(got, ref, rtol=REL)
    # deterministic: a second launch gives the same bits
    flat2, _ = ops.channel_importance(ws, mode)
    assert np.array_equal(flat2.cpu().numpy(), flat)


def test_taylor_importance_accumulates_over_batches(dev):
    shapes = synth.deeplab_resnet50_conv_shapes(21)
    pairs = synth.conv_pairs(77, shapes)
    ws, gs = _device_pairs(pairs, dev)
    scores, offs = ops.channel_importance(ws, nat.IMP_TAYLOR, grads=gs)
    first = scores.clone()
    ops.channel_importance(ws, nat.IMP_TAYLOR, grads=gs, out=scores, accumulate=True)
    ops.channel_importance(ws, nat.IMP_TAYLOR, grads=gs, out=scores, accumulate=True)
    got1, got3 = first.cpu().numpy(), scores.cpu().numpy()
    for (w, g), off in zip(pairs, offs):
        exact = oi.taylor_rows_f64(w, g)
        ref = oi.taylor_rows(torch.from_numpy(w), torch.from_numpy(g)).numpy()
        seg = slice(off, off + w.shape[0])
        assert np.abs(got1[seg] - exact).max() <= REL * np.abs(exact).max()
        np.testing.assert_allclose(got1[seg], ref, rtol=REL)
        np.testing.assert_allclose(got3[seg], 3 * exact, rtol=REL)  # linear in the batch count
        assert (got1[seg] >= 0).all()


def test_importance_odd_shapes(dev):
    shapes = [(7, 3, 5, 5), (1, 1, 1, 1), (33, 10, 1, 1), (5, 9, 3, 3), (130, 1, 3, 3), (3, 1001, 1, 1)]
    pairs = synth.conv_pairs(5, shapes)
    # one view whose storage is only 4-byte aligned relative to the others
    big = torch.from_numpy(np.concatenate([[0.0], pairs[0][0].reshape(-1)]).astype(np.float32)).to(dev)
    ws, gs = _device_pairs(pairs, dev)
    ws[0] = big[1:].view(7, 3, 5, 5)
    for mode, fn in ((nat.IMP_L2, lambda w, g: oi.magnitude_rows_f64(w, 2)),
                     (nat.IMP_TAYLOR, oi.taylor_rows_f64)):
        flat, offs = ops.channel_importance(ws, mode, grads=gs)
        for (w, g), off in zip(pairs, offs):
            np.testing.assert_allclose(flat.cpu().numpy()[off: off + w.shape[0]], fn(w, g), rtol=REL)


def test_in_channel_importance(dev):
    shapes = [(512, 512, 3, 3), (256, 2048, 1, 1), (48, 7, 5, 5), (37, 10, 1, 1), (2048, 64, 1, 1)]
    pairs = synth.conv_pairs(21, shapes)
    flat = torch.cat([torch.from_numpy(w).reshape(-1) for w, _ in pairs]).to(dev)
    layers, pos, opos = [], 0, 0
    for (o, i, kh, kw) in shapes:
        layers.append(ops.ColsLayout(pos, opos, o, i, kh * kw))
        pos += o * i * kh * kw
        opos += i
    plan = ops.ColsPlan(layers, dev)
    for mode, p in ((nat.IMP_L1, 1), (nat.IMP_L2, 2)):
        out = torch.empty(opos, dtype=torch.float32, device=dev)
        plan.run([flat], out, mode)
        got = out.cpu().numpy()
        for (w, _), lay in zip(pairs, layers):
            exact = oi.magnitude_cols_f64(w, p)
            np.testing.assert_allclose(got[lay.out_off: lay.out_off + lay.cols], exact, rtol=REL)
            np.testing.assert_allclose(got[lay.out_off: lay.out_off + lay.cols],
                                       oi.magnitude_cols(torch.from_numpy(w), p).numpy(), rtol=REL)


# ---------------------------------------------------------------------------------- gather
@pytest.mark.parametrize("shape", [(512, 512, 3, 3), (256, 2048, 1, 1), (2048, 1, 3, 3), (64, 3, 7, 7),
                                   (304,), (21, 256, 1, 1), (10, 6, 5, 5)])
def test_gather_is_bit_exact(dev, shape):
    n = int(np.prod(shape))
    x = synth.normal(3, 9, n).reshape(shape)
    x.reshape(-1)[::97] = np.float32("nan")  # payloads must survive: a word copy, not arithmetic
    rows = shape[0]
    keep_out = osel.keep_indices(rows, osel.select_local(synth.uniform(1, 1, rows), rows // 3))
    cases = [(keep_out, None)]
    if len(shape) > 1 and shape[1] > 1:
        cols = shape[1]
        keep_in = osel.keep_indices(cols, osel.select_local(synth.uniform(1, 2, cols), cols // 2))
        cases += [(None, keep_in), (keep_out, keep_in)]
    xd = torch.from_numpy(x).to(dev)
    for ko, ki in cases:
        ref = osel.slice_tensor(torch.from_numpy(x), ko, ki).numpy()
        got = ops.gather_channels(xd, None if ko is None else torch.from_numpy(ko).to(dev),
                                  None if ki is None else torch.from_numpy(ki).to(dev)).cpu().numpy()
        assert got.shape == ref.shape
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_gather_whole_model_one_launch(dev):
    shapes = synth.deeplab_resnet50_conv_shapes(21)[:30]
    pairs = synth.conv_pairs(8, shapes)
    src = torch.cat([torch.from_numpy(w).reshape(-1) for w, _ in pairs]).to(dev)
    specs, idx_parts, refs = [], [], []
    spos = dpos = ipos = 0
    for (w, _), (o, i, kh, kw) in zip(pairs, shapes):
        ko = osel.keep_indices(o, osel.select_local(oi.magnitude_rows_f64(w, 2), o // 2))
        ki = osel.keep_indices(i, np.arange(0, i, 3)) if i > 3 else None
        refs.append(osel.slice_tensor(torch.from_numpy(w), ko, ki).numpy())
        o_off, ipos = ipos, ipos + len(ko)
        idx_parts.append(ko)
        i_off = -1
        if ki is not None:
            i_off, ipos = ipos, ipos + len(ki)
            idx_parts.append(ki)
        di = len(ki) if ki is not None else i
        specs.append(ops.GatherSpec(spos, dpos, o_off, i_off, len(ko), di, i, kh * kw))
        spos += w.size
        dpos += (len(ko) * di * kh * kw + 3) // 4 * 4
    plan = ops.GatherPlan(specs, dev)
    dst = torch.zeros(dpos, dtype=torch.float32, device=dev)
    idx = torch.from_numpy(np.concatenate(idx_parts)).to(dev)
    plan.run([src], dst, idx)
    out = dst.cpu().numpy()
    for spec, ref in zip(specs, refs):
        got = out[spec.dst_off: spec.dst_off + ref.size].reshape(ref.shape)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_selected_indices_from_device_scores_match_oracle(dev):
    """Channel indices chosen from the HIP scores equal those chosen from the reference-order CPU
    scores, including a layer with duplicated rows (exact ties -> lower index first)."""
    shapes = [(64, 32, 3, 3), (128, 64, 1, 1)]
    pairs = synth.conv_pairs(31, shapes)
    w0 = pairs[0][0]
    w0[10] = w0[3]
    w0[40] = w0[3]
    ws, _ = _device_pairs(pairs, dev)
    flat, offs = ops.channel_importance(ws, nat.IMP_L2)
    flat = flat.cpu().numpy()
    for (w, _), off in zip(pairs, offs):
        got = flat[off: off + w.shape[0]]
        assert got[3] == got[10] == got[40] or w is not w0
        for n_pruned in (1, w.shape[0] // 2, w.shape[0] - 8):
            mine = osel.select_local(got, n_pruned)
            theirs = osel.select_local(oi.magnitude_rows_f64(w, 2), n_pruned)
            np.testing.assert_array_equal(mine, theirs)


def test_gather_staged_and_fallback_items_in_one_launch(dev):
    """In-channel lists through the LDS-staged path and past its limits, mixed in ONE launch: several batches of
    rows per item with a short last batch, identity rows with a column list, 3x3 kernels with few columns, a
    source row too long for the stage, more kept columns than the column table holds, a row length that is not a
    multiple of four, and a rows-only tensor.  Guard words around every destination stay untouched."""
    rng = np.random.default_rng(17)
    # (rows, cols, inner, keep_out?, kept columns)
    tensors = [(37, 2048, 1, True, 1024),    # staged: 2 rows per batch, 25 kept rows -> 13 batches per item group
               (50, 512, 1, False, 256),     # staged, identity rows
               (24, 16, 9, True, 8),         # staged, 3x3, tiny rows: many rows per batch
               (6, 1300, 4, True, 648),      # source row of 5200 words: too long for the stage -> element path
               (3, 4800, 1, True, 2400),     # 2400 kept columns: more than the column table -> element path
               (9, 64, 9, True, 31),         # row length 279: not a multiple of four -> element path
               (128, 96, 1, True, None)]     # rows only
    srcs, specs, idx_parts, refs = [], [], [], []
    spos, dpos, ipos = 0, 0, 0
    for rows, cols, inner, use_out, kept in tensors:
        w = rng.standard_normal((rows, cols, inner)).astype(np.float32)
        ko = np.sort(rng.choice(rows, max(1, rows * 2 // 3), replace=False)).astype(np.int64) if use_out else None
        ki = np.sort(rng.choice(cols, kept, replace=False)).astype(np.int64) if kept else None
        ref = w[ko] if ko is not None else w
        ref = ref[:, ki] if ki is not None else ref
        o_off = i_off = -1
        if ko is not None:
            o_off, ipos = ipos, ipos + len(ko)
            idx_parts.append(ko)
        if ki is not None:
            i_off, ipos = ipos, ipos + len(ki)
            idx_parts.append(ki)
        specs.append(ops.GatherSpec(spos, dpos, o_off, i_off, ref.shape[0], ref.shape[1], cols, inner))
        srcs.append((spos, w))
        refs.append(ref)
        spos += (w.size + 3) // 4 * 4
        dpos += (ref.size + 3) // 4 * 4 + 4      # four guard words behind every destination
    src = np.zeros(spos, dtype=np.float32)
    for off, w in srcs:
        src[off: off + w.size] = w.reshape(-1)
    plan = ops.GatherPlan(specs, dev)
    assert plan.plan_flags & 4                    # the launch carries LDS for its staged items
    guard = np.float32(-3.5)
    dst = torch.full((dpos,), float(guard), dtype=torch.float32, device=dev)
    idx = torch.from_numpy(np.concatenate(idx_parts)).to(dev)
    plan.run([torch.from_numpy(src).to(dev)], dst, idx)
    out = dst.cpu().numpy()
    for spec, ref in zip(specs, refs):
        got = out[spec.dst_off: spec.dst_off + ref.size]
        assert np.array_equal(got.view(np.uint32), ref.reshape(-1).view(np.uint32)), (spec.dst_rows, spec.dst_cols)
        assert (out[spec.dst_off + ref.size: spec.dst_off + (ref.size + 3) // 4 * 4 + 4] == guard).all()


def test_gather_unaligned_offsets_and_odd_rows(dev):
    """Layouts that rule out 16-byte moves: destinations and sources at offsets that are not a
    multiple of four words, rows of odd length, one-row and one-column tensors, identity row
    lists with a column list, and items larger than one work item."""
    rng = np.random.default_rng(5)
    # (rows, cols, inner, keep_out?, keep_in?)
    tensors = [(7, 5, 9, True, True), (33, 1, 1, True, False), (1, 40, 1, False, True), (19, 3, 49, True, False),
               (300, 17, 1, True, True), (4, 4, 4, False, False), (65, 130, 9, False, True), (2048, 3, 1, True, False)]
    srcs, specs, idx_parts, refs = [], [], [], []
    spos, dpos, ipos = 1, 3, 0  # deliberately unaligned starts
    for rows, cols, inner, use_out, use_in in tensors:
        w = rng.standard_normal((rows, cols, inner)).astype(np.float32)
        ko = np.sort(rng.choice(rows, max(1, rows * 2 // 3), replace=False)).astype(np.int64) if use_out else None
        ki = np.sort(rng.choice(cols, max(1, cols // 2), replace=False)).astype(np.int64) if use_in else None
        ref = w[ko] if ko is not None else w
        ref = ref[:, ki] if ki is not None else ref
        o_off = i_off = -1
        if ko is not None:
            o_off, ipos = ipos, ipos + len(ko)
            idx_parts.append(ko)
        if ki is not None:
            i_off, ipos = ipos, ipos + len(ki)
            idx_parts.append(ki)
        specs.append(ops.GatherSpec(spos, dpos, o_off, i_off, ref.shape[0], ref.shape[1], cols, inner))
        srcs.append((spos, w))
        refs.append(ref)
        spos += w.size + int(rng.integers(0, 3))
        dpos += ref.size + int(rng.integers(0, 3))
    src = np.zeros(spos + 8, dtype=np.float32)
    for off, w in srcs:
        src[off: off + w.size] = w.reshape(-1)
    plan = ops.GatherPlan(specs, dev)
    guard = np.float32(-7.25)
    dst = torch.full((dpos + 8,), float(guard), dtype=torch.float32, device=dev)
    idx = torch.from_numpy(np.concatenate(idx_parts)).to(dev)
    plan.run([torch.from_numpy(src).to(dev)], dst, idx)
    out = dst.cpu().numpy()
    written = np.zeros(out.shape, dtype=bool)
    for spec, ref in zip(specs, refs):
        got = out[spec.dst_off: spec.dst_off + ref.size].reshape(ref.shape)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (spec, ref.shape)
        written[spec.dst_off: spec.dst_off + ref.size] = True
    assert (out[~written] == guard).all()  # nothing outside the destinations was touched


UPSAMPLE_CASES = [
    # shape, kwargs, input dtype, output dtype
    ((2, 5, 16, 16), dict(scale_factor=4.0, align_corners=True), torch.float32, torch.float32),
    ((1, 3, 7, 9), dict(scale_factor=2.0, align_corners=True), torch.float32, torch.float32),    # out_w % 4 != 0
    ((2, 4, 13, 17), dict(size=(40, 52), align_corners=False), torch.float32, torch.float32),
    ((2, 4, 12, 20), dict(scale_factor=2.0, align_corners=False), torch.float32, torch.float32),
    ((1, 2, 1, 1), dict(size=(8, 12), align_corners=False), torch.float32, torch.float32),       # a pooled 1x1 map
    ((1, 2, 1, 1), dict(size=(8, 12), align_corners=True), torch.float32, torch.float32),
    ((1, 2, 6, 6), dict(size=(6, 6), align_corners=True), torch.float32, torch.float32),         # identity size
    ((2, 7, 36, 36), dict(scale_factor=4.0, align_corners=True), torch.float16, torch.float32),  # autocast: fp16 in, fp32 out
    ((2, 7, 36, 36), dict(scale_factor=4.0, align_corners=True), torch.float16, torch.float16),
    ((8, 21, 128, 128), dict(scale_factor=4.0, align_corners=True), torch.float16, torch.float32),
    ((3, 2, 5, 300), dict(scale_factor=4.0, align_corners=True), torch.float32, torch.float32),  # one output row per workgroup
]


@pytest.mark.parametrize("shape,kw,in_dtype,out_dtype", UPSAMPLE_CASES)
def test_upsample_bilinear_vs_oracle_and_torch(dev, shape, kw, in_dtype, out_dtype):
    """The up-sampling kernel against torch's op on the CPU (the oracle: <= 1e-5 relative to the
    largest input) and against torch's own device kernel, whose arithmetic it repeats: bit for bit."""
    from oracle import upsample as ou
    g = torch.Generator().manual_seed(sum(shape) + 7)
    x = (3.0 * torch.randn(*shape, generator=g)).to(in_dtype)
    xd = x.to(dev)
    got = ops.upsample_bilinear(xd, out_dtype=out_dtype, **kw)
    want = ou.upsample_bilinear(x, **kw)
    assert got.shape == want.shape and got.dtype == out_dtype
    tol = 1e-5 * float(x.float().abs().max()) if out_dtype == torch.float32 else 2e-3 * float(x.float().abs().max())
    assert float((got.float().cpu() - want).abs().max()) <= tol
    there = torch.nn.functional.interpolate(xd.float(), mode="bilinear", **kw).to(out_dtype)
    assert torch.equal(got, there)


def test_upsample_rejects_what_it_does_not_implement(dev):
    x = torch.randn(1, 2, 8, 8, device=dev)
    with pytest.raises(RuntimeError):
        ops.upsample_bilinear(x, size=(4, 4))            # down-sampling
    with pytest.raises(RuntimeError):
        ops.upsample_bilinear(torch.randn(1, 2, 8, 8), scale_factor=2.0)   # host tensor
    with pytest.raises(TypeError):
        ops.upsample_bilinear(x.double(), scale_factor=2.0)
    with pytest.raises(ValueError):
        ops.upsample_bilinear(x, size=(16, 16), scale_factor=2.0)
    assert ops.upsample_bilinear(x[:0], scale_factor=2.0).shape == (0, 2, 16, 16)


def test_model_forward_on_library_kernels_equals_torch_forward(dev):
    """Every up-sampling and batch-norm module of a whole forward on the library's kernels (no_grad),
    re-run on its torch path (gradient mode) with the input it saw: identical output, under autocast
    and in fp32.  (Whole forwards are not compared: MIOpen may choose another convolution algorithm
    from one call to the next.)"""
    from spice_amd import nets
    from spice_amd.nets.decoders import BilinearLike, BilinearUp
    from spice_amd.nets.layers import BatchNormAct2d, Conv1x1, DepthwiseConv3x3
    torch.manual_seed(11)
    for arch in ("DeepLabV3Plus", "FPN", "Unet"):
        model = nets.create_model(arch, "resnet50", 5).to(dev).eval()
        for m in model.modules():   # running statistics and affine terms away from their initial 0 / 1
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_mean.normal_(0.0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.data.uniform_(0.5, 1.5)
                m.bias.data.normal_(0.0, 0.2)
        watched = [m for m in model.modules()
                   if isinstance(m, (BatchNormAct2d, BilinearUp, BilinearLike, Conv1x1, DepthwiseConv3x3))]
        x = torch.randn(2, 3, 128, 128, device=dev)
        for cast in (True, False):
            seen = []
            keep = lambda t: t.detach().clone() if isinstance(t, torch.Tensor) else t   # a shortcut's batch-norm rides along as a module
            hooks = [m.register_forward_hook(lambda mod, i, o: seen.append((mod, [keep(t) for t in i], o.detach().clone())))
                     for m in watched]
            with torch.autocast("cuda", dtype=torch.float16, enabled=cast):
                with torch.no_grad():
                    y = model(x)
                for h in hooks:   # before the modules are called again below
                    h.remove()
                assert y.shape == (2, 5, 128, 128) and len(seen) == len(watched) > 10
                for mod, inputs, out in seen:
                    again = mod(*inputs).detach()   # gradient mode: the torch ops
                    if isinstance(mod, BatchNormAct2d) or getattr(mod, "follow_input_dtype", False) or isinstance(mod, BilinearLike):
                        again = again.to(out.dtype)  # torch up-samples to fp32 under autocast; the consumer casts back
                    assert again.dtype == out.dtype, (arch, cast, type(mod).__name__)
                    if isinstance(mod, (Conv1x1, DepthwiseConv3x3)):
                        # another route to the same sums (fp32 accumulation either way): a few fp16 steps where terms cancel
                        scale = float(again.float().abs().max())
                        assert float((out.float() - again.float()).abs().max()) <= 4e-3 * scale, (arch, cast, type(mod).__name__)
                    else:
                        assert torch.equal(out, again), (arch, cast, type(mod).__name__)


@pytest.mark.parametrize("shape,w_dtype", [((2, 6, 8, 8), torch.float16), ((1, 3, 5, 16), torch.float32),
                                            ((2, 304, 36, 40), torch.float32), ((8, 304, 128, 128), torch.float32),
                                            ((1, 2, 1, 8), torch.float16), ((2, 5, 300, 24), torch.float16),
                                            # four output rows per lane: a height that is no multiple of four, rows of
                                            # 8 / 16 / 32 / 64 lanes, a last workgroup that is partly outside the plane
                                            ((1, 3, 130, 128), torch.float16), ((2, 4, 37, 512), torch.float32),
                                            ((1, 2, 257, 64), torch.float16), ((3, 7, 16, 256), torch.float32)])
def test_depthwise_conv3x3_vs_oracle_and_torch(dev, shape, w_dtype):
    """Bit for bit against the CPU oracle of torch's own depthwise kernel (fp32 sums in tap order, one rounding)
    and against that kernel itself on the device (torch with MIOpen switched off); within one fp16 step of the
    exact sum, which is what MIOpen's reference kernel (torch's default route for this call) computes."""
    from oracle import depthwise as od
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g).half()
    w = (0.5 * torch.randn(shape[1], 1, 3, 3, generator=g)).to(w_dtype)
    got = ops.depthwise_conv3x3(x.to(dev), w.to(dev))
    assert got.dtype == torch.float16 and torch.equal(got.cpu(), od.depthwise_conv3x3_fp32_chain(x, w, 1))
    with torch.backends.cudnn.flags(enabled=False):
        native = torch.nn.functional.conv2d(x.to(dev), w.to(dev).half(), None, 1, 1, 1, shape[1])
    assert torch.equal(got, native)
    exact = od.depthwise_conv3x3(x, w)
    # one fp16 step of the result, plus fp32 accumulation error of nine terms where they cancel
    terms = 9.0 * float(x.abs().max()) * float(w.abs().max())
    step = exact.float().abs() * 2.0 ** -10 + terms * 2.0 ** -22
    assert bool(((got.float().cpu() - exact.float()).abs() <= step).all())
    assert int((got.cpu() != exact).sum()) <= max(2, got.numel() // 1000)
    there = torch.nn.functional.conv2d(x.to(dev), w.to(dev).half(), None, 1, 1, 1, shape[1])   # torch's default route
    assert bool(((there.float() - got.float()).abs() <= step.to(dev)).all())


def test_depthwise_conv3x3_rejects_other_layouts(dev):
    x = torch.randn(1, 4, 8, 12, device=dev).half()    # width not a multiple of 8
    w = torch.randn(4, 1, 3, 3, device=dev)
    assert not ops.depthwise_conv3x3_supported(x, w)
    with pytest.raises(RuntimeError):
        ops.depthwise_conv3x3(x, w)
    with pytest.raises(TypeError):
        ops.depthwise_conv3x3(x.float(), w)
    with pytest.raises(ValueError):
        ops.depthwise_conv3x3(x[..., :8].contiguous(), w[:3])


@pytest.mark.parametrize("shape,dilation", [((2, 16, 40, 48), 4), ((1, 8, 16, 24), 8), ((4, 2048, 32, 32), 12),
                                             ((2, 512, 32, 32), 24), ((2, 512, 32, 32), 36), ((1, 3, 9, 8), 12)])
def test_dilated_depthwise_conv3x3_vs_oracle_and_torch(dev, shape, dilation):
    """Bit for bit against the oracle's fp32 tap-order sum and against torch's device result (its own
    depthwise kernel serves dilated calls and sums the same way)."""
    from oracle import depthwise as od
    g = torch.Generator().manual_seed(sum(shape) + dilation)
    x = torch.randn(*shape, generator=g).half()
    w = (0.5 * torch.randn(shape[1], 1, 3, 3, generator=g))
    got = ops.depthwise_conv3x3(x.to(dev), w.to(dev), dilation)
    assert torch.equal(got.cpu(), od.depthwise_conv3x3_fp32_chain(x, w, dilation))
    there = torch.nn.functional.conv2d(x.to(dev), w.to(dev).half(), None, 1, dilation, dilation, shape[1])
    assert torch.equal(got, there)
    with pytest.raises(RuntimeError):
        ops.depthwise_conv3x3(x.to(dev), w.to(dev), 6)


@pytest.mark.parametrize("shape,cout", [((2, 64, 16, 16), 256), ((8, 256, 128, 128), 64), ((3, 230, 9, 7), 1000), ((1, 3, 5, 5), 7),
                                         ((2, 2048, 32, 32), 512), ((2, 100, 1, 1), 33)])
def test_conv1x1_f16_vs_oracle_and_torch(dev, shape, cout):
    """The 1x1 convolution as a matrix product: against fp64 sums of the fp16 values (one fp16 step where
    terms cancel) and against torch's conv2d on the device (fp32 accumulation as well, other kernel)."""
    g = torch.Generator().manual_seed(sum(shape) + cout)
    x = torch.randn(*shape, generator=g).half()
    w = (torch.randn(cout, shape[1], generator=g) / shape[1] ** 0.5).half()
    got = ops.conv1x1_f16(x.to(dev), w.to(dev))
    exact = torch.einsum("oc,nchw->nohw", w.double(), x.double())
    scale = float(exact.abs().max())
    assert got.dtype == torch.float16 and got.shape == exact.shape
    assert float((got.double().cpu() - exact).abs().max()) <= 2.0 ** -10 * scale + 1e-6
    there = torch.nn.functional.conv2d(x.to(dev), w.to(dev).view(cout, shape[1], 1, 1))
    assert float((got.float() - there.float()).abs().max()) <= 2.0 ** -9 * scale + 1e-6
    again = ops.conv1x1_f16(x.to(dev), w.to(dev))     # the cached plan
    assert torch.equal(got, again)


def test_conv1x1_f16_rejects_other_layouts(dev):
    x = torch.randn(1, 4, 3, 3, device=dev).half()
    w = torch.randn(5, 4, device=dev).half()
    with pytest.raises(TypeError):
        ops.conv1x1_f16(x.float(), w)
    with pytest.raises(ValueError):
        ops.conv1x1_f16(x, w[:, :3].contiguous())
    with pytest.raises(RuntimeError):
        ops.conv1x1_f16(x.cpu(), w.cpu())
    assert ops.conv1x1_f16(x[:0], w).shape == (0, 5, 3, 3)


def test_torch_forward_switch(dev):
    """set_library_forward(False) keeps the modules on torch's kernels under no_grad as well; the
    bit-identical ones (batch-norm, up-sampling, dilated depthwise) give the same tensors either way."""
    from spice_amd.nets import layers
    from spice_amd.nets.decoders import BilinearUp
    torch.manual_seed(2)
    x = torch.randn(2, 16, 12, 16, device=dev).half()
    bn = layers.BatchNormAct2d(16, relu=True).to(dev).eval()
    bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0)
    up, dw = BilinearUp(4), layers.DepthwiseConv3x3(16, dilation=4).to(dev).half()
    assert layers.library_forward()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):   # as the sweep and the evaluation loops run
        lib = (bn(x, x), up(x), dw(x))
        layers.set_library_forward(False)
        try:
            assert not layers.Conv1x1(16, 8).takes_matrix_path(torch.float16)
            ref = (bn(x, x), up(x), dw(x))
        finally:
            layers.set_library_forward(True)
    for a, b in zip(lib, ref):
        assert a.dtype == b.dtype and torch.equal(a, b)


def test_half_precision_model_runs_through_the_modules(dev):
    """model.half() (no autocast): batch-norm parameters are fp16 there, which the fused kernel does not
    take - the modules must fall back to torch's ops instead of raising."""
    from spice_amd import nets
    torch.manual_seed(4)
    model = nets.create_model("DeepLabV3Plus", "resnet50", 5).to(dev).eval().half()
    with torch.no_grad():
        y = model(torch.randn(1, 3, 64, 64, device=dev).half())
    assert y.shape == (1, 5, 64, 64) and y.dtype == torch.float16 and bool(torch.isfinite(y).all())


BN_CASES = [
    # shape, dtype, residual, relu
    ((2, 16, 8, 8), torch.float32, False, False),
    ((2, 16, 8, 8), torch.float32, True, True),
    ((3, 7, 5, 3), torch.float32, True, True),        # odd plane, scalar path
    ((2, 24, 6, 6), torch.float16, False, True),      # planes of 36: vectors straddle planes
    ((2, 24, 6, 6), torch.float16, True, True),
    ((4, 10, 3, 2), torch.float16, True, False),      # planes shorter than a vector
    ((8, 64, 36, 36), torch.float16, True, True),
    ((8, 256, 64, 64), torch.float16, False, True),
    ((1, 5, 1, 2), torch.float16, False, True),
]


@pytest.mark.parametrize("shape,dtype,with_res,relu", BN_CASES)
def test_bn_act_vs_oracle_and_torch(dev, shape, dtype, with_res, relu):
    """The fused kernel against the CPU oracle (one rounding step of the dtype at most) and against
    the three device kernels it replaces (MIOpen batch-norm, add, clamp): bit for bit."""
    from oracle import bnact as ob
    g = torch.Generator().manual_seed(sum(shape))
    c = shape[1]
    x = (2.0 * torch.randn(*shape, generator=g)).to(dtype)
    res = (2.0 * torch.randn(*shape, generator=g)).to(dtype) if with_res else None
    mean, var = torch.randn(c, generator=g), torch.rand(c, generator=g) * 2 + 0.05
    weight, bias = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g)
    d = lambda t: None if t is None else t.to(dev)
    got = ops.bn_act(d(x), d(mean), d(var), d(weight), d(bias), 1e-5, residual=d(res), relu=relu)
    want = ob.bn_act(x, mean, var, weight, bias, 1e-5, residual=res, relu=relu)
    assert got.dtype == dtype and got.shape == want.shape
    step = 2e-3 if dtype == torch.float16 else 2e-6
    assert float((got.float().cpu() - want.float()).abs().max()) <= step * max(1.0, float(want.float().abs().max()))
    there = torch.nn.functional.batch_norm(d(x), d(mean), d(var), d(weight), d(bias), False, 0.0, 1e-5)
    if with_res:
        there = there + d(res)
    if relu:
        there = torch.relu(there)
    assert torch.equal(got, there)


@pytest.mark.parametrize("shape,dtype,relu", [((2, 16, 8, 8), torch.float32, True), ((3, 7, 5, 3), torch.float32, False),
                                               ((2, 24, 6, 6), torch.float16, True), ((8, 256, 64, 64), torch.float16, True),
                                               ((4, 10, 3, 2), torch.float16, True)])
def test_bn_act_dual_vs_oracle_and_torch(dev, shape, dtype, relu):
    """relu(bn(x) + bn_r(residual)) in one launch against the CPU oracle and against the four device kernels
    it replaces (MIOpen batch-norm twice, add, clamp): bit for bit; and through the ResNet bottleneck that uses it."""
    from oracle import bnact as ob
    g = torch.Generator().manual_seed(sum(shape) + 1)
    c = shape[1]
    bn, rbn = torch.nn.BatchNorm2d(c).eval(), torch.nn.BatchNorm2d(c, eps=1e-3).eval()
    for m in (bn, rbn):
        m.running_mean.copy_(torch.randn(c, generator=g))
        m.running_var.copy_(torch.rand(c, generator=g) * 2 + 0.05)
        m.weight.data.copy_(torch.rand(c, generator=g) + 0.5)
        m.bias.data.copy_(torch.randn(c, generator=g))
    x = (2.0 * torch.randn(*shape, generator=g)).to(dtype)
    res = (2.0 * torch.randn(*shape, generator=g)).to(dtype)
    want = ob.bn_act_dual(x, (bn.running_mean, bn.running_var, bn.weight.data, bn.bias.data, bn.eps), res,
                          (rbn.running_mean, rbn.running_var, rbn.weight.data, rbn.bias.data, rbn.eps), relu=relu)
    bn, rbn = bn.to(dev), rbn.to(dev)
    with torch.no_grad():
        got = ops.bn_act_dual(x.to(dev), bn, res.to(dev), rbn, relu=relu)
        f = torch.nn.functional.batch_norm
        there = (f(x.to(dev), bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps) +
                 f(res.to(dev), rbn.running_mean, rbn.running_var, rbn.weight, rbn.bias, False, 0.0, rbn.eps))
        if relu:
            there = torch.relu(there)
    assert got.dtype == dtype and torch.equal(got, there)
    step = 4e-3 if dtype == torch.float16 else 4e-6
    assert float((got.float().cpu() - want.float()).abs().max()) <= step * max(1.0, float(want.float().abs().max()))


def test_bottleneck_with_projection_uses_one_tail_launch(dev):
    """The first bottleneck of a ResNet stage: library forward (shortcut batch-norm fused into the tail) equals the
    module-by-module torch forward of the same block, under autocast and in fp32."""
    from spice_amd.nets import layers
    from spice_amd.nets.resnet import Bottleneck
    torch.manual_seed(3)
    blk = Bottleneck(64, 32, stride=2, downsample=True).to(dev).eval()
    for m in blk.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0.0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0.0, 0.2)
    x = torch.randn(2, 64, 32, 32, device=dev)
    for cast in (False, True):
        with torch.autocast("cuda", dtype=torch.float16, enabled=cast), torch.no_grad():
            mid = blk.conv3(blk.bn2(blk.conv2(blk.bn1(blk.conv1(x)))))
            short = blk.downsample[0](x)
            got = blk.bn3(mid, short, blk.downsample[1])
            layers.set_library_forward(False)
            try:
                there = torch.relu(torch.nn.functional.batch_norm(
                    mid, blk.bn3.running_mean, blk.bn3.running_var, blk.bn3.weight, blk.bn3.bias, False, 0.0, blk.bn3.eps) +
                    blk.downsample[1](short))
            finally:
                layers.set_library_forward(True)
            assert torch.equal(got, there.to(got.dtype)), cast
            assert blk(x).shape == (2, 128, 16, 16)


def test_bn_act_rejects_what_it_does_not_reproduce(dev):
    x = torch.randn(2, 4, 3, 3, device=dev).half()
    p = [torch.ones(4, device=dev) for _ in range(4)]
    assert not ops.bn_act_supported(x) and ops.bn_act_supported(x.float()) and ops.bn_act_supported(x[:, :, :, :2].contiguous())
    with pytest.raises(RuntimeError):
        ops.bn_act(x, *p, 1e-5)                      # fp16 with an odd plane size
    with pytest.raises(ValueError):
        ops.bn_act(x.float(), *p, 1e-5, residual=torch.zeros(2, 4, 3, 2, device=dev))
    with pytest.raises(TypeError):
        ops.bn_act(x.float(), p[0].half(), *p[1:], 1e-5)
    assert ops.bn_act(x.float()[:0], *p, 1e-5).shape == (0, 4, 3, 3)


@pytest.mark.parametrize("shape,relu,want_full", [((2, 5, 8, 16), True, True), ((1, 3, 2, 16), True, True),
                                                   ((3, 7, 34, 48), False, True), ((8, 64, 256, 256), True, True),
                                                   ((2, 64, 64, 128), True, False), ((1, 2, 6, 32), True, True)])
def test_bn_act_maxpool_vs_oracle_and_torch(dev, shape, relu, want_full):
    """The ResNet stem behind its first convolution in one launch: both outputs bit for bit what the batch-norm
    launch and torch's max-pooling give on the device, and the CPU oracle's within the batch-norm's fp16 step.
    NaN and infinities travel through the pooling as through torch's."""
    from oracle import bnact as ob
    g = torch.Generator().manual_seed(sum(shape))
    c = shape[1]
    bn = torch.nn.BatchNorm2d(c).eval()
    bn.running_mean.copy_(torch.randn(c, generator=g))
    bn.running_var.copy_(torch.rand(c, generator=g) * 2 + 0.05)
    bn.weight.data.copy_(torch.rand(c, generator=g) + 0.5)
    bn.bias.data.copy_(torch.randn(c, generator=g))
    x = (2.0 * torch.randn(*shape, generator=g)).half()
    x.view(-1)[::97] = float("nan")
    x.view(-1)[5::193] = float("inf")
    x.view(-1)[7::211] = float("-inf")
    want_f, want_p = ob.bn_act_maxpool(x, bn.running_mean, bn.running_var, bn.weight.data, bn.bias.data, bn.eps, relu=relu)
    bn = bn.to(dev)
    with torch.no_grad():
        full, pooled = ops.bn_act_maxpool(x.to(dev), bn, relu=relu, want_full=want_full)
        there_f = ops.bn_act(x.to(dev), bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps, relu=relu)
        there_p = torch.nn.functional.max_pool2d(there_f, 3, 2, 1)
    assert (full is None) == (not want_full)
    same = lambda a, b: torch.equal(torch.nan_to_num(a.float(), nan=12345.0), torch.nan_to_num(b.float(), nan=12345.0))
    assert pooled.shape == there_p.shape and pooled.dtype == torch.float16 and same(pooled, there_p)
    if want_full:
        assert same(full, there_f)
    finite = torch.isfinite(want_p.float()) & torch.isfinite(pooled.float().cpu())
    assert bool((torch.isnan(want_p) == torch.isnan(pooled.cpu())).all())
    diff = (pooled.float().cpu() - want_p.float()).abs()[finite]
    assert float(diff.max()) <= 4e-3 * max(1.0, float(want_p.float()[finite].abs().max()))


def test_bn_act_maxpool_rejects_other_layouts(dev):
    bn = torch.nn.BatchNorm2d(4).eval().to(dev)
    assert not ops.bn_act_maxpool_supported(torch.zeros(1, 4, 7, 16, device=dev).half())     # odd height
    assert not ops.bn_act_maxpool_supported(torch.zeros(1, 4, 8, 24, device=dev).half())     # width not a multiple of 16
    with pytest.raises(RuntimeError):
        ops.bn_act_maxpool(torch.zeros(1, 4, 7, 16, device=dev).half(), bn)
    with pytest.raises(TypeError):
        ops.bn_act_maxpool(torch.zeros(1, 4, 8, 16, device=dev), bn)
    full, pooled = ops.bn_act_maxpool(torch.zeros(0, 4, 8, 16, device=dev).half(), bn)
    assert full.shape == (0, 4, 8, 16) and pooled.shape == (0, 4, 4, 8)


def test_resnet_stem_uses_one_launch_and_equals_the_modules(dev):
    """The encoder's stem on the library path against its modules one after the other (hooks force that path)."""
    from spice_amd.nets.resnet import ResNet50Encoder
    torch.manual_seed(5)
    enc = ResNet50Encoder(3, output_stride=16).to(dev).eval()
    enc.bn1.running_mean.normal_(0.0, 0.2)
    enc.bn1.running_var.uniform_(0.5, 1.5)
    x = torch.randn(2, 3, 64, 96, device=dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        f1, pooled = enc._stem(x)
        handle = enc.maxpool.register_forward_hook(lambda m, i, o: None)
        g1, gp = enc._stem(x)
        handle.remove()
        enc.skip_stem_feature = True
        none, pooled2 = enc._stem(x)
    assert none is None and torch.equal(pooled, gp) and torch.equal(f1, g1) and torch.equal(pooled2, gp)


@pytest.mark.parametrize("dtype,out_dtype,align", [(torch.float16, torch.float16, True), (torch.float32, torch.float32, False),
                                                    (torch.float16, torch.float32, True)])
def test_upsample_into_a_channel_slice(dev, dtype, out_dtype, align):
    """out= a channel slice of a larger tensor: same values as a separate tensor, neighbours untouched."""
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 10, 6, 8, generator=g).to(dtype).to(dev)
    both = torch.full((3, 4 + 10 + 6, 24, 32), 3.0, dtype=out_dtype, device=dev)
    want = ops.upsample_bilinear(x, scale_factor=4.0, align_corners=align, out_dtype=out_dtype)
    got = ops.upsample_bilinear(x, scale_factor=4.0, align_corners=align, out=both[:, 4:14])
    assert got.data_ptr() == both[:, 4:14].data_ptr() and torch.equal(both[:, 4:14], want)
    assert bool((both[:, :4] == 3.0).all()) and bool((both[:, 14:] == 3.0).all())
    with pytest.raises(ValueError):
        ops.upsample_bilinear(x, scale_factor=4.0, align_corners=align, out=both[:, 4:14, ::2])


def test_deeplab_decoder_without_concatenation_pass(dev):
    """Producers writing their slices of the concatenations directly give what the decoder that concatenates gives,
    on the same encoder features (whole forwards are not comparable bit for bit: MIOpen may pick another algorithm
    for a 3x3 convolution from one call to the next)."""
    from spice_amd import nets
    from spice_amd.nets import decoders
    torch.manual_seed(9)
    model = nets.create_model("DeepLabV3Plus", "resnet50", 7).to(dev).eval()
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0.0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
    x = torch.randn(2, 3, 128, 160, device=dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        features = model.encoder(x)
        direct = model.decoder(features)
        keep = decoders._writes_in_place
        decoders._writes_in_place = lambda t: False
        try:
            concatenated = model.decoder(features)
        finally:
            decoders._writes_in_place = keep
    assert direct.dtype == torch.float16 and torch.equal(direct, concatenated)


@pytest.mark.parametrize("cin,cout,h,w", [(256, 21, 32, 32), (205, 21, 16, 24), (77, 5, 8, 8), (26, 19, 12, 10)])
def test_head_convolution_with_bias_in_one_launch(dev, cin, cout, h, w):
    """Conv1x1Bias under fp16 autocast: the matrix-core kernel's product rounded to fp16, plus the bias in fp32,
    rounded to fp16 - bit for bit what the bias-free product and a separate addition give; within fp16 rounding of
    torch's conv2d (which adds the bias before its one rounding); torch's conv2d itself where the plane is not a
    multiple of 8 pixels or gradients are wanted."""
    from spice_amd.nets import layers
    torch.manual_seed(cin + cout)
    conv = layers.Conv1x1Bias(cin, cout).to(dev)
    conv.bias.data.normal_(0.0, 0.5)
    x = torch.randn(2, cin, h, w, device=dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        assert conv.served_by_library(x.dtype, h, w) == ((h * w) % 8 == 0)
        got = conv(x)
        plain = ops.conv1x1_f16(x.half(), conv.weight.detach().half().reshape(cout, cin))
        want = (plain.float() + conv.bias.view(1, -1, 1, 1)).half()
        there = torch.nn.functional.conv2d(x, conv.weight, conv.bias)
    assert got.dtype == torch.float16 and got.shape == (2, cout, h, w)
    if (h * w) % 8 == 0:
        assert torch.equal(got, want)
    step = 2e-3 * max(1.0, float(there.float().abs().max()))
    assert float((got.float() - there.float()).abs().max()) <= step
    with torch.autocast("cuda", dtype=torch.float16):   # gradients wanted: torch's convolution
        y = conv(x)
    assert y.requires_grad and torch.equal(y.detach(), there)


def test_plan_upload_as_a_launch(dev):
    """spice_gather_upload through the raw C-ABI and ops.upload_staged above it: bytes from page-locked host memory
    arrive on the device in stream order without a copy command; several typed parts in one launch, more than the
    staging memory holds in a row (both halves reused), a part larger than a half by an ordinary copy."""
    L = nat.lib()
    host = torch.arange(4096, dtype=torch.int64).pin_memory()
    dst = torch.zeros(4096, dtype=torch.int64, device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    assert L.spice_gather_upload(dst.data_ptr(), host.data_ptr(), host.numel() * 8, st) == 0
    assert torch.equal(dst.cpu(), host)
    assert L.spice_gather_upload(dst.data_ptr(), host.data_ptr(), 0, st) == 0
    assert L.spice_gather_upload(dst.data_ptr(), host.data_ptr(), 24, st) != 0                       # not whole 16-byte words
    assert L.spice_gather_upload(dst.data_ptr() + 8, host.data_ptr(), 64, st) != 0                   # misaligned
    pageable = torch.arange(64, dtype=torch.int64)
    assert L.spice_gather_upload(dst.data_ptr(), pageable.data_ptr(), 512, st) != 0                  # not page-locked
    with pytest.raises(ValueError):
        ops.spice_ops().gather_upload(dst.view(torch.uint8), pageable.view(torch.uint8))
    rng = np.random.default_rng(3)
    parts = [rng.integers(0, 1 << 40, 1001).astype(np.int64), rng.integers(0, 255, 77).astype(np.uint8),
             np.zeros(0, dtype=np.int64), rng.standard_normal(300).astype(np.float32)]
    got = ops.upload_staged(parts, dev)
    for a, t in zip(parts, got):
        assert t.is_cuda and tuple(t.shape) == a.shape and np.array_equal(t.cpu().numpy(), a)
    kept = []
    for k in range(24):     # 24 x 300 KB: both halves of the staging memory are written several times over
        a = rng.integers(0, 1 << 30, 300 * 128).astype(np.int64)
        kept.append((a, ops.upload_staged([a], dev)[0]))
    big = rng.integers(0, 255, (1 << 21) + 4096).astype(np.uint8)
    kept.append((big, ops.upload_staged([big], dev)[0]))
    for a, t in kept:
        assert np.array_equal(t.cpu().numpy(), a)


@pytest.mark.parametrize("cin,cout,h,w", [(16, 21, 64, 64), (13, 5, 16, 24), (16, 19, 10, 12)])
def test_head_3x3_convolution_with_bias_in_one_launch(dev, cin, cout, h, w):
    """Conv3x3Bias (the U-Nets' head) under fp16 autocast: the matrix-core kernel's sums rounded to fp16, plus the bias
    in fp32, rounded to fp16 - bit for bit the bias-free launch and a separate addition; within fp16 rounding of
    torch's conv2d; torch's conv2d itself where the width is not a multiple of 8 or gradients are wanted."""
    from spice_amd.nets import layers
    torch.manual_seed(cin + cout + w)
    conv = layers.Conv3x3Bias(cin, cout).to(dev)
    conv.bias.data.normal_(0.0, 0.5)
    x = torch.randn(2, cin, h, w, device=dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        served = conv.served_by_library(x.dtype, h, w)
        assert served == (w % 8 == 0)
        got = conv(x)
        there = torch.nn.functional.conv2d(x, conv.weight, conv.bias, padding=1)
        if served:
            w9 = conv.weight.detach().half().permute(0, 2, 3, 1).reshape(cout, 9 * cin).contiguous()
            want = (ops.conv3x3_plain(x.half(), w9).float() + conv.bias.view(1, -1, 1, 1)).half()
            assert torch.equal(got, want)
    assert got.dtype == torch.float16 and got.shape == (2, cout, h, w)
    step = 2e-3 * max(1.0, float(there.float().abs().max()))
    assert float((got.float() - there.float()).abs().max()) <= step
    with torch.autocast("cuda", dtype=torch.float16):
        assert conv(x).requires_grad


@pytest.mark.parametrize("n,cin,h,w,cout", [(8, 2048, 32, 32, 256), (2, 205, 8, 24, 37), (1, 64, 5, 7, 3), (3, 320, 64, 64, 256)])
def test_aspp_pooling_branch_in_two_launches(dev, n, cin, h, w, cout):
    """ops.pooled_branch (plane means -> 1x1 convolution -> batch-norm -> ReLU -> spread over the plane) against the
    module sequence it replaces under fp16 autocast (decoders.ASPPPooling on torch's route): the means bit for bit,
    the result within one fp16 step of the fp64 evaluation of the same rounded steps and of torch's modules (their
    convolution sums in another order); into a channel slice of a larger tensor; and the module takes the op."""
    from spice_amd.nets import decoders, layers
    torch.manual_seed(n + cin + cout)
    m = decoders.ASPPPooling(cin, cout).to(dev).eval()
    m.bn.running_mean.normal_(0.0, 0.2); m.bn.running_var.uniform_(0.5, 1.5)
    m.bn.weight.data.uniform_(0.5, 1.5); m.bn.bias.data.normal_(0.0, 0.2)
    x = torch.randn(n, cin, h, w, device=dev).half()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        layers.set_library_forward(False)
        try:
            there = m(x)
        finally:
            layers.set_library_forward(True)
        assert m.served_by_library(x)
        got = m(x)
        big = torch.full((n, cout + 5, h, w), 7.0, device=dev, dtype=torch.float16)
        m(x, out=big[:, 2:2 + cout])
    assert got.shape == there.shape == (n, cout, h, w) and got.dtype == torch.float16
    assert torch.equal(big[:, 2:2 + cout], got) and bool((big[:, :2] == 7).all()) and bool((big[:, 2 + cout:] == 7).all())
    # every pixel of a plane holds the same value
    assert torch.equal(got, got[:, :, :1, :1].expand_as(got))
    # the same rounded steps in fp64
    mean16 = x.double().mean((2, 3)).half()
    w16 = m.conv.weight.detach().reshape(cout, cin).half()
    conv16 = (mean16.double() @ w16.double().t()).half()
    bn = m.bn
    y = ((conv16.double() - bn.running_mean.double()) / torch.sqrt(bn.running_var.double() + bn.eps)
         * bn.weight.detach().double() + bn.bias.detach().double()).half().clamp_min(0)
    step = 2.0 ** -10 * max(float(y.abs().max()), 1.0)
    assert float((got[:, :, 0, 0].double() - y.double()).abs().max()) <= 2 * step
    assert float((got.double() - there.double()).abs().max()) <= 3 * step


@pytest.mark.parametrize("encoder", ["resnet50", "timm-efficientnet-b3"])
def test_unet_decoder_copies_in_fp16_change_no_value(dev, encoder):
    """The U-Net decoder's nearest up-sampling and concatenation written in fp16 straight into the concatenation
    buffer against autocast's way (up-sampling and concatenation in fp32, cast back by the convolution): copies of
    fp16 values either way, so the logits are the same bit for bit; odd-sized and skip-less blocks included."""
    from spice_amd import nets
    from spice_amd.nets import decoders
    torch.manual_seed(4)
    model = nets.create_model("Unet", encoder, 7).to(dev).eval()
    x = torch.randn(2, 3, 96, 160, device=dev)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        fast = model(x)
        keep = decoders._fp16_inference
        decoders._fp16_inference = lambda t: False
        try:
            plain = model(x)
        finally:
            decoders._fp16_inference = keep
    assert fast.dtype == plain.dtype and torch.equal(fast, plain)
    blk = decoders.UnetDecoderBlock(16, 8, 8).to(dev).eval()
    a, s = torch.randn(3, 16, 5, 7, device=dev).half(), torch.randn(3, 8, 10, 14, device=dev).half()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        got = blk(a, s)
        decoders._fp16_inference = lambda t: False
        try:
            want = blk(a, s)
        finally:
            decoders._fp16_inference = keep
    assert torch.equal(got, want)
