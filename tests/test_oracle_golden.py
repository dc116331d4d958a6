"""The CPU oracle against the fixtures generated from the reference (tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import load_golden
from oracle import confusion as oc
from oracle import importance as oi
from oracle import kdlite as ok
from oracle import quota as oq
from toy_models import ToyDictNet, ToyNet

KD = load_golden("kdlite.npz")
KD_NAMES = [str(x) for x in KD["names"]]


def kd_case(name):
    T, alpha, bw, conf, c = KD[f"{name}/params"]
    kw = dict(temperature=float(T), alpha=float(alpha), boundary_weight=float(bw),
              confidence_weight=bool(conf), num_classes=int(c), ignore_index=255)
    return (torch.from_numpy(KD[f"{name}/student"]), torch.from_numpy(KD[f"{name}/teacher"]),
            torch.from_numpy(KD[f"{name}/labels"]), kw)


@pytest.mark.parametrize("name", KD_NAMES)
def test_kdlite_oracle_matches_reference(name):
    s, t, lab, kw = kd_case(name)
    out = ok.kdlite_loss(s, t, lab, **kw)
    got = np.array([out["total"].item(), out["ce"].item(), out["kd"].item()], dtype=np.float32)
    # same torch op sequence as the reference: equal up to the thread-count dependent order of
    # torch's CPU reductions (a few ulp)
    np.testing.assert_allclose(got, KD[f"{name}/out3"], rtol=2e-6, atol=2e-7, equal_nan=True)
    if np.isfinite(KD[f"{name}/out3"][0]):
        _, grad = ok.kdlite_loss_and_grad(s, t, lab, **kw)
        g_ref = KD[f"{name}/grad"]
        assert np.abs(grad.numpy() - g_ref).max() <= 1e-6 * np.abs(g_ref).max()


@pytest.mark.parametrize("name", KD_NAMES)
def test_kdlite_f64_matches_reference(name):
    s, t, lab, kw = kd_case(name)
    out, grad = ok.kdlite_loss_f64(s.numpy(), t.numpy(), lab.numpy(), want_grad=True, **kw)
    ref = KD[f"{name}/out3"]
    for k, key in enumerate(("total", "ce", "kd")):
        if np.isnan(ref[k]):
            assert np.isnan(out[key])
        elif name == "c21_same_logits" and key == "kd":
            assert abs(out[key] - ref[k]) < 1e-6  # the reference's own fp32 rounding (about 1e-7)
        else:
            assert abs(out[key] - ref[k]) <= 2e-6 * max(abs(ref[k]), 1e-3), (key, out[key], ref[k])
    if np.isfinite(ref[0]):
        g_ref = KD[f"{name}/grad"]
        assert np.abs(grad - g_ref).max() <= 1e-6 * max(np.abs(g_ref).max(), 1e-12) + 1e-9


def test_kdlite_special_cases():
    out = KD["c21_all_ignored/out3"]
    assert np.isnan(out[0]) and np.isnan(out[1]) and out[2] == 0.0
    assert abs(KD["c21_same_logits/out3"][2]) < 1e-6
    mask = KD["c3_single255/boundary_mask"]
    assert mask.sum() == 9 and mask[0, 3:6, 3:6].all()
    s, t, lab, kw = kd_case("c3_single255")
    np.testing.assert_array_equal(ok.boundary_mask(lab).numpy(), mask)


CF = load_golden("confusion.npz")


@pytest.mark.parametrize("name", [str(x) for x in CF["names"]])
def test_confusion_oracle_matches_reference(name):
    c = int(CF[f"{name}/num_classes"][0])
    conf = np.zeros((c, c), dtype=np.int64)
    for k in range(2):
        oc.update_confusion(conf, CF[f"{name}/pred{k}"], CF[f"{name}/target{k}"], c)
        np.testing.assert_array_equal(conf, CF[f"{name}/conf{k}"])
        valid = (CF[f"{name}/target{k}"] >= 0) & (CF[f"{name}/target{k}"] < c)
    iou, acc, miou = oc.metrics_from_confusion(torch.from_numpy(conf))
    np.testing.assert_array_equal(iou, CF[f"{name}/iou"])
    np.testing.assert_array_equal(acc, CF[f"{name}/acc"])
    assert miou == CF[f"{name}/miou"][0]
    m = oc.per_class_metrics(torch.from_numpy(conf))
    np.testing.assert_array_equal(np.array(m["per_class_iou"]), CF[f"{name}/pc_iou"])
    np.testing.assert_array_equal(np.array(m["per_class_acc"]), CF[f"{name}/pc_acc"])
    assert m["miou"] == CF[f"{name}/pc_means"][0] and m["mean_acc"] == CF[f"{name}/pc_means"][1]


@pytest.mark.parametrize("name", [str(x) for x in CF["names"]])
def test_confusion_torch_op_sequence_matches_reference(name):
    """The torch-op form the benchmark's CPU leg times: same matrices as the reference's function.  Cases whose
    predictions fall outside the table make the reference's bincount view raise; so must this."""
    c = int(CF[f"{name}/num_classes"][0])
    conf = torch.zeros(c, c, dtype=torch.int64)
    for k in range(2):
        oc.update_confusion_torch(conf, torch.from_numpy(CF[f"{name}/pred{k}"]), torch.from_numpy(CF[f"{name}/target{k}"]), c)
        np.testing.assert_array_equal(conf.numpy(), CF[f"{name}/conf{k}"])


def test_confusion_sum_is_valid_pixel_count():
    name = "c21_uniform"
    c = 21
    t = np.concatenate([CF[f"{name}/target0"], CF[f"{name}/target1"]])
    assert CF[f"{name}/conf1"].sum() == ((t >= 0) & (t < c)).sum()


TY = load_golden("taylor.npz")


def toy(cls):
    net = cls()
    sd = {k[len("state/"):]: torch.from_numpy(TY[k]) for k in TY.files if k.startswith("state/")}
    net.load_state_dict(sd)
    return net


def toy_batches():
    return [(torch.from_numpy(TY[f"batch{b}/images"]), torch.from_numpy(TY[f"batch{b}/targets"]))
            for b in range(3)]


@pytest.mark.parametrize("tag,cls,bare,max_batches", [
    ("tuple_max2", ToyNet, False, 2), ("tuple_all", ToyNet, False, 100),
    ("bare", ToyNet, True, 100), ("dict_out", ToyDictNet, False, 100)])
def test_taylor_oracle_matches_reference(tag, cls, bare, max_batches):
    net = toy(cls)
    net.train()
    batches = toy_batches()
    loader = [b[0] for b in batches] if bare else batches
    scores = oi.compute_taylor_scores(net, loader, "cpu", max_batches=max_batches)
    assert net.training
    by_name = dict(net.named_modules())
    for n in [str(x) for x in TY["conv_names"]]:
        # conv backward on CPU reduces in a thread-count dependent order: a few ulp, not bitwise
        np.testing.assert_allclose(scores[by_name[n]].numpy(), TY[f"{tag}/{n}"], rtol=1e-5, atol=1e-9)
        assert (scores[by_name[n]] >= 0).all()


QT = load_golden("quota.npz")


@pytest.mark.parametrize("name", [str(x) for x in QT["names"]])
def test_quota_oracle_matches_reference(name):
    ratio, alpha, min_ch = QT[f"{name}/params"]
    layers = QT["layers"]
    sens, flops, chans, pos = [], [], [], []
    for li, (o, i, kh, kw) in enumerate(layers):
        if o > int(min_ch):
            sens.append(torch.from_numpy(QT[f"{name}/scores{li}"]).mean().item())
            flops.append(oq.conv_flops(int(o), int(i), int(kh), int(kw)))
            chans.append(int(o))
            pos.append(li)
    taken, quotas = oq.layer_quotas(sens, flops, chans, float(ratio), float(alpha), int(min_ch))
    ref_q = QT[f"{name}/quota"]
    ref_s = QT[f"{name}/sensitivity"]
    for li in range(len(layers)):
        if li not in pos:
            assert ref_q[li] == -1.0  # layers at/below min_channels get no quota entry
    for k, li in enumerate(pos):
        assert quotas[k] == ref_q[li], (name, li, quotas[k], ref_q[li])
        assert sens[k] == ref_s[li]
    assert sum(taken) <= int(sum(chans) * float(ratio))
    assert all(c - t >= int(min_ch) for c, t in zip(chans, taken))
