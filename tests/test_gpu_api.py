"""The drop-in Python surfaces on the GPU: loss modules (autograd), Taylor pass, metric classes,
sweep and the end-to-end `prune_model` run, checked against the oracle / golden fixtures."""
import copy
import json
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import load_golden
from oracle import confusion as oc
from oracle import kdlite as ok
from spice_amd import engine, nets, synth
from spice_amd.baseline.eval.metrics import (compute_metrics_from_confusion, update_confusion_matrix,
                                             update_confusion_matrix_from_logits)
from spice_amd.pruning import PruningConfig, prune_model
from spice_amd.pruning.eval.metrics import SegmentationMetrics
from spice_amd.pruning.losses import KDLiteLoss, PixelKDLoss, SegmentationLoss
from spice_amd.pruning.sensitivity_utils import compute_taylor_scores
from spice_amd.sweep import SensitivitySweep
from toy_models import ToyDictNet, ToyNet

pytestmark = pytest.mark.gpu
KD = load_golden("kdlite.npz")
TY = load_golden("taylor.npz")


def _free_port():
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.mark.parametrize("name", ["c21_default", "c5_weighted", "c5_boundary_conf", "c19_odd_hw"])
def test_kdlite_module_forward_backward(dev, name):
    T, alpha, bw, conf, c = KD[f"{name}/params"]
    loss = KDLiteLoss(temperature=float(T), alpha=float(alpha), boundary_weight=float(bw),
                      confidence_weight=bool(conf), num_classes=int(c)).to(dev)
    s = torch.from_numpy(KD[f"{name}/student"]).to(dev).requires_grad_(True)
    t = torch.from_numpy(KD[f"{name}/teacher"]).to(dev)
    lab = torch.from_numpy(KD[f"{name}/labels"]).to(dev)
    out = loss(s, t, lab)
    assert set(out) == {"total", "ce", "kd"} and all(v.dim() == 0 and v.device == s.device for v in out.values())
    ref = KD[f"{name}/out3"]
    for k, key in enumerate(("total", "ce", "kd")):
        assert abs(out[key].item() - ref[k]) <= 1e-5 * abs(ref[k])
    (3.0 * out["total"]).backward()  # upstream factor is applied on the device
    g_ref = 3.0 * KD[f"{name}/grad"]
    assert np.abs(s.grad.cpu().numpy() - g_ref).max() <= 1e-5 * np.abs(g_ref).max()
    assert t.grad is None


@pytest.mark.parametrize("name", ["c21_default", "c5_weighted", "c19_odd_hw"])
def test_kdlite_label_pass_beside_the_forwards(dev, name):
    """KDLiteLoss.prepare_labels: the loss's label pass on a stream of its own before the networks' forwards
    (pruning/trainer.py:172-183 computes the loss behind both); the loss launch then skips the pass.  Same kernels on
    the same values: loss and gradient are those of the plain call bit for bit, and the reference's within 1e-5."""
    from spice_amd import ops
    T, alpha, bw, conf, c = KD[f"{name}/params"]
    t = torch.from_numpy(KD[f"{name}/teacher"]).to(dev)
    lab = torch.from_numpy(KD[f"{name}/labels"]).to(dev)
    plain = KDLiteLoss(temperature=float(T), alpha=float(alpha), num_classes=int(c)).to(dev)
    s0 = torch.from_numpy(KD[f"{name}/student"]).to(dev).requires_grad_(True)
    want = plain(s0, t, lab)
    want["total"].backward()
    loss = KDLiteLoss(temperature=float(T), alpha=float(alpha), num_classes=int(c)).to(dev)
    calls = []
    real = ops.kdlite_forward
    try:
        ops.kdlite_forward = lambda *a, **k: (calls.append(bool(k.get("normalisers_ready"))), real(*a, **k))[1]
        for step in range(2):   # the second step finds the workspace the first one used
            s = torch.from_numpy(KD[f"{name}/student"]).to(dev).requires_grad_(True)
            assert loss.prepare_labels(lab)
            _ = (s * 2).sum()   # (the forwards of a step go here)
            got = loss(s, t, lab)
            got["total"].backward()
            for k in ("total", "ce", "kd"):
                assert got[k].item() == want[k].item()
            assert torch.equal(s.grad, s0.grad)
        assert calls == [True, True]
        # labels that were not prepared, and a forward without gradient: the plain call
        calls.clear()
        loss.prepare_labels(lab)
        got = loss(s0.detach().requires_grad_(True), t, lab.clone())
        with torch.no_grad():
            loss.prepare_labels(lab)
            loss(s0, t, lab)
        assert calls == [False, False] and got["total"].item() == want["total"].item()
    finally:
        ops.kdlite_forward = real
    ref = KD[f"{name}/out3"]
    assert abs(want["total"].item() - ref[0]) <= 1e-5 * abs(ref[0])
    # the options whose pass reads the teacher have no separate label pass
    assert not KDLiteLoss(boundary_weight=2.0, num_classes=int(c)).to(dev).prepare_labels(lab)
    assert not loss.prepare_labels(lab.cpu())


def test_loss_components_differentiate_separately(dev):
    name = "c21_default"
    s0 = torch.from_numpy(KD[f"{name}/student"])
    t0 = torch.from_numpy(KD[f"{name}/teacher"])
    lab0 = torch.from_numpy(KD[f"{name}/labels"])
    loss = KDLiteLoss(temperature=4.0, alpha=0.7, num_classes=21).to(dev)
    s = s0.to(dev).requires_grad_(True)
    out = loss(s, t0.to(dev), lab0.to(dev))
    (g_kd,) = torch.autograd.grad(out["kd"], s, retain_graph=True)
    (g_ce,) = torch.autograd.grad(out["ce"], s)
    sc = s0.clone().requires_grad_(True)
    ref = ok.kdlite_loss(sc, t0, lab0, temperature=4.0, alpha=0.7, num_classes=21)
    (r_kd,) = torch.autograd.grad(ref["kd"], sc, retain_graph=True)
    (r_ce,) = torch.autograd.grad(ref["ce"], sc)
    assert (g_kd.cpu() - r_kd).abs().max() <= 1e-5 * r_kd.abs().max()
    assert (g_ce.cpu() - r_ce).abs().max() <= 1e-5 * r_ce.abs().max()
    # the two stand-alone loss modules
    seg = SegmentationLoss(21).to(dev)
    s2 = s0.to(dev).requires_grad_(True)
    ce = seg(s2, lab0.to(dev))
    ce.backward()
    assert abs(ce.item() - ref["ce"].item()) <= 1e-5 * abs(ref["ce"].item())
    assert (s2.grad.cpu() - r_ce).abs().max() <= 1e-5 * r_ce.abs().max()
    kd = PixelKDLoss(4.0)(s0.to(dev), t0.to(dev), (lab0 != 255).to(dev))
    assert abs(kd.item() - ref["kd"].item()) <= 1e-5 * abs(ref["kd"].item())
    with torch.no_grad():
        out_ng = loss(s0.to(dev), t0.to(dev), lab0.to(dev))
    assert not out_ng["total"].requires_grad


def test_training_step_launches_the_fused_loss_once(dev, monkeypatch):
    """Forward + backward of a training step is ONE spice_kdlite_forward call that writes the gradient
    too (pruning/trainer.py:179-182); an evaluation forward asks for no gradient; a backward through
    ce or kd alone recomputes with the actual upstream gradients."""
    from spice_amd import ops
    calls = []
    real = ops.kdlite_forward

    def counting(*a, **k):
        calls.append(bool(k.get("want_grad", False)))
        return real(*a, **k)

    monkeypatch.setattr(ops, "kdlite_forward", counting)
    name = "c21_default"
    t = torch.from_numpy(KD[f"{name}/teacher"]).to(dev)
    lab = torch.from_numpy(KD[f"{name}/labels"]).to(dev)
    loss = KDLiteLoss(temperature=4.0, alpha=0.7, num_classes=21).to(dev)
    s = torch.from_numpy(KD[f"{name}/student"]).to(dev).requires_grad_(True)
    out = loss(s, t, lab)
    out["total"].backward()
    assert calls == [True]
    g_ref = KD[f"{name}/grad"]
    assert np.abs(s.grad.cpu().numpy() - g_ref).max() <= 1e-5 * np.abs(g_ref).max()
    calls.clear()
    with torch.no_grad():
        loss(s, t, lab)
    loss(s.detach(), t, lab)
    assert calls == [False, False]
    calls.clear()
    out = loss(s, t, lab)
    torch.autograd.grad(out["kd"], s)
    assert calls == [True, True]


def test_loss_rejects_cpu_tensors():
    loss = KDLiteLoss(num_classes=21)
    x = torch.zeros(1, 21, 4, 4)
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        loss(x, x, torch.zeros(1, 4, 4, dtype=torch.long))


def _toy(cls, dev):
    net = cls()
    net.load_state_dict({k[len("state/"):]: torch.from_numpy(TY[k]) for k in TY.files if k.startswith("state/")})
    return net.to(dev)


@pytest.mark.parametrize("tag,cls,bare,max_batches", [("tuple_max2", ToyNet, False, 2), ("tuple_all", ToyNet, False, 100),
                                                      ("bare", ToyNet, True, 100), ("dict_out", ToyDictNet, False, 100)])
def test_compute_taylor_scores_matches_reference(dev, tag, cls, bare, max_batches):
    net = _toy(cls, dev)
    net.train()
    batches = [(torch.from_numpy(TY[f"batch{b}/images"]), torch.from_numpy(TY[f"batch{b}/targets"])) for b in range(3)]
    loader = [b[0] for b in batches] if bare else batches
    # Two bars.  (1) The part this library computes - |g|*|w| summed per channel, accumulated over the
    # batches, divided by their number - against an fp64 evaluation on the SAME device gradients: 1e-5.
    # (2) The whole function against the scores the reference produced on the CPU: its forward and
    # backward ran on torch's CPU kernels, here they run on MIOpen, so the gradients themselves differ
    # in the last bits before any reduction happens: 2e-5.
    from spice_amd.pruning import sensitivity_utils as su
    exact = {}

    def reduce_and_record(weights, grads, out, accumulate):
        for k, (w, g) in enumerate(zip(weights, grads)):
            term = (g.detach().double().abs() * w.detach().double().abs()).flatten(1).sum(1).cpu().numpy()
            exact[k] = exact.get(k, 0) + term
        exact["batches"] = exact.get("batches", 0) + 1
        return su._hip_row_reduce(weights, grads, out, accumulate)

    scores = compute_taylor_scores(net, loader, str(dev), max_batches=max_batches, _row_reduce=reduce_and_record)
    assert net.training and all(p.grad is None for p in net.parameters())
    by_name = dict(net.named_modules())
    assert len(scores) == 4
    convs = [m for m in net.modules() if isinstance(m, torch.nn.Conv2d)]
    for k, m in enumerate(convs):
        np.testing.assert_allclose(scores[m].cpu().numpy(), exact[k] / exact["batches"], rtol=1e-5, atol=1e-12)
    for n in [str(x) for x in TY["conv_names"]]:
        got = scores[by_name[n]].cpu().numpy()
        np.testing.assert_allclose(got, TY[f"{tag}/{n}"], rtol=2e-5, atol=1e-9)
        assert (got >= 0).all()


def test_metric_surfaces(dev):
    c = 21
    lab = synth.label_map(4, 2, 64, 96, c, "blocky")
    s, _ = synth.logits_pair(4, 2, c, 64, 96)
    pred = oc.argmax_first(s)
    want = np.zeros((c, c), dtype=np.int64)
    oc.update_confusion(want, pred, lab, c)
    conf = torch.zeros(c, c, dtype=torch.int64, device=dev)
    update_confusion_matrix(conf, torch.from_numpy(pred).to(dev), torch.from_numpy(lab).to(dev), c)
    update_confusion_matrix_from_logits(conf, torch.from_numpy(s).to(dev), torch.from_numpy(lab).to(dev), c)
    np.testing.assert_array_equal(conf.cpu().numpy(), 2 * want)
    iou, acc, miou = compute_metrics_from_confusion(conf)
    r_iou, r_acc, r_miou = oc.metrics_from_confusion(torch.from_numpy(2 * want))
    np.testing.assert_array_equal(iou, r_iou)
    assert miou == r_miou
    sm = SegmentationMetrics(c, device=str(dev))
    sm.update(torch.from_numpy(pred).to(dev), torch.from_numpy(lab).to(dev))
    sm.update_from_logits(torch.from_numpy(s).to(dev), torch.from_numpy(lab).to(dev))
    m = sm.compute_metrics()
    ref = oc.macro_metrics_present_classes(2 * want)
    assert m["miou"] == pytest.approx(ref["miou"], rel=1e-12) and m["mean_acc"] == pytest.approx(ref["mean_acc"], rel=1e-12)
    np.testing.assert_array_equal(m["confusion_matrix"], 2 * want)
    sm.reset()
    assert int(sm.conf.sum()) == 0


def test_sweep_candidate_equals_separately_pruned_model(dev):
    torch.manual_seed(3)
    model = nets.create_model("DeepLabV3Plus", "resnet50", 7).to(dev).eval()
    ignored = [m for m in model.segmentation_head.modules() if isinstance(m, nn.Conv2d)]
    before = copy.deepcopy(model.state_dict())
    sweep = SensitivitySweep(model, 7, ratios=(0.25, 0.5), ignored_layers=ignored, autocast_dtype=None)
    assert len(sweep.candidates) == 2 * 46
    x = torch.randn(2, 3, 64, 64, device=dev)
    lab = torch.from_numpy(synth.label_map(1, 2, 64, 64, 7, "blocky")).to(dev)
    pick = [0, 5, 40, len(sweep.candidates) - 1]
    res = sweep.run(x, lab, indices=pick)
    # the base model is untouched after the sweep
    for k, v in model.state_dict().items():
        assert torch.equal(v, before[k]), k
    for i in pick:
        cand = sweep.candidates[i]
        clone = copy.deepcopy(model)
        groups = engine.MagnitudePruner(clone, ignored_layers=[m for m in clone.segmentation_head.modules()
                                                               if isinstance(m, nn.Conv2d)]).prunable_groups()
        engine.pruner.apply_pruning([(groups[cand.group], cand.pruned)])
        with torch.no_grad():
            logits = clone(x)
        want = np.zeros((7, 7), dtype=np.int64)
        oc.update_confusion(want, oc.argmax_first(logits.cpu().numpy()), lab.cpu().numpy(), 7)
        # the same network evaluated twice: MIOpen may pick another algorithm for a convolution on a later call,
        # which can flip the arg-max of a pixel whose two best logits are a rounding error apart (2 cells per pixel)
        assert int(np.abs(res["confusion"][i].cpu().numpy() - want).sum()) <= 4
        assert int(res["confusion"][i].sum()) == int((lab != 255).sum())
    untouched = [i for i in range(len(sweep.candidates)) if i not in pick]
    assert int(res["confusion"][untouched].sum()) == 0
    assert torch.isfinite(res["kd"][pick]).all() and (res["kd"][pick][:, 2] >= -1e-6).all()  # KL of nearly equal logits: fp32 noise around 0


def test_sweep_lanes_give_the_results_of_one_lane(dev):
    """Two and three candidates in flight (copies of the network on streams of their own) against one: the same
    confusion matrices bit for bit (fp16 autocast forward: every convolution on the library's kernels, which give the
    same values on every call), the same losses to 1e-5, and the network itself untouched."""
    torch.manual_seed(11)
    model = nets.create_model("DeepLabV3Plus", "resnet50", 7).to(dev).eval()
    ignored = [m for m in model.segmentation_head.modules() if isinstance(m, nn.Conv2d)]
    before = copy.deepcopy(model.state_dict())
    x = torch.randn(2, 3, 128, 128, device=dev)
    lab = torch.from_numpy(synth.label_map(2, 2, 128, 128, 7, "blocky")).to(dev)
    pick = sorted(set(range(0, 92, 7)) | {1, 2, 90, 91})
    results = []
    for lanes in (1, 2, 3):
        sweep = SensitivitySweep(model, 7, ratios=(0.25, 0.5), ignored_layers=ignored)
        assert sweep.enable_lanes(lanes) == lanes
        results.append(sweep.run(x, lab, indices=pick))
        for k, v in model.state_dict().items():
            assert torch.equal(v, before[k]), k
    for res in results[1:]:
        assert torch.equal(res["confusion"], results[0]["confusion"])
        # the loss kernel sums its workgroups' partial sums in the order they finish: 1e-5 relative, and 1e-7 absolute
        # for the KL term, which is ~1e-6 here (the un-pruned network is the teacher)
        assert torch.allclose(res["kd"], results[0]["kd"], rtol=1e-5, atol=1e-7)
    assert int(results[0]["confusion"][pick].sum()) == len(pick) * int((lab != 255).sum())


def test_two_lane_sweep_repeats_the_one_lane_tables_at_the_benchmark_size(dev):
    """Candidates 270-320 of the 414 of the DeepLabV3Plus_resnet sweep on the benchmark's batch (the groups of layer4,
    whose pruned channel counts send the 1x1 kernel through its rarely taken forms and partly filled tiles), forwarded
    whole, two at a time, six times over: every run gives the confusion matrices of the one-lane run.  (A form of the
    1x1 kernel that fetched its batch-norm constants in front of the K loop broke exactly this, in about half of the
    runs, while each kernel alone repeated itself.)"""
    from spice_amd.sweep import DEFAULT_RATIOS
    torch.manual_seed(1234)
    model = nets.create_model("DeepLabV3Plus", "resnet50", 21).to(dev).eval()
    ignored = [m for m in model.segmentation_head.modules() if isinstance(m, nn.Conv2d)]
    x = torch.from_numpy(synth.normal(1234, 50, 8 * 3 * 512 * 512).reshape(8, 3, 512, 512)).to(dev)
    lab = torch.from_numpy(synth.label_map(1234, 8, 512, 512, 21, "blocky", ignore_share=0.05)).to(dev)
    pick = list(range(270, 320))

    def tables(lanes):
        sweep = SensitivitySweep(model, 21, DEFAULT_RATIOS, ignored_layers=ignored)
        sweep.prefix_reuse = False
        assert sweep.enable_lanes(lanes) == lanes
        return sweep.run(x, lab, indices=pick)["confusion"][pick].clone()

    one = tables(1)
    assert torch.equal(one, tables(1))
    for _ in range(6):
        two = tables(2)
        moved = (two - one).abs().sum((1, 2))
        assert int(moved.max()) == 0, {pick[i]: int(moved[i]) for i in moved.nonzero().flatten().tolist()}


def test_prune_model_end_to_end(dev, tmp_path):
    cfg = PruningConfig(model_name="DeepLabV3Plus_resnet", output_dir=str(tmp_path), data_root=str(tmp_path / "nodata"),
                        num_classes=5, input_size=(64, 64), example_input_size=(1, 3, 64, 64),
                        pruning_ratio=0.5, iterative_steps=2, pruning_strategy="magnitude_taylor",
                        device="cuda", batch_size=4, fine_tune_epochs=1, final_fine_tune_epochs=1,
                        calibration_batches=2, enable_kd_lite=True, kd_temperature=4.0, kd_alpha=0.7,
                        kd_data_ratio=0.5, use_wandb=False)
    res = prune_model(cfg)
    assert res["final_params"] < res["initial_params"] and 0 < res["params_reduction"] < 1
    run_dir = tmp_path / "DeepLabV3Plus_resnet_magnitude_taylor_0.50_kd_refine"
    for f in ("final_model.pth", "metrics.json", "model_info.txt", "hybrid_pruning_report.json"):
        assert (run_dir / f).exists(), f
    assert len(list((run_dir / "intermediate").glob("*_step_*_of_2.pth"))) == 2
    rep = json.load(open(run_dir / "hybrid_pruning_report.json"))
    assert rep["num_layers"] == 68 and "quota_allocation" in rep
    hist = json.load(open(run_dir / "metrics.json"))
    assert len(hist) == 2 and hist[1]["params"] < hist[0]["params"]
    pruned = torch.load(run_dir / "final_model.pth", weights_only=False)  # written by this test run
    with torch.no_grad():
        y = pruned.to(dev).eval()(torch.randn(1, 3, 64, 64, device=dev))
    assert y.shape == (1, 5, 64, 64)


def test_inference_loop_matches_reference_fixture(dev):
    """baseline/eval/inference.py:16-57 on a network whose logits are exact small integers."""
    from spice_amd.baseline.eval import inference
    with open(os.path.join(os.path.dirname(__file__), "golden", "formats.json")) as f:
        g = json.load(f)["inference"]
    net = nn.Conv2d(3, g["classes"], 1)
    with torch.no_grad():
        net.weight.copy_(torch.tensor(g["weight"]))
        net.bias.copy_(torch.tensor(g["bias"]))
    batches = [(torch.tensor(x), torch.tensor(y)) for x, y in zip(g["images"], g["masks"])]
    for loader in (batches, [{"image": x, "mask": y} for x, y in batches]):
        res = inference(net, loader, dev, g["classes"])
        assert set(res) == set(g["result"])
        for k, want in g["result"].items():
            np.testing.assert_allclose(res[k], want, rtol=2e-6, atol=1e-9, err_msg=k)


def test_core_evaluator_fused_equals_unfused(dev):
    from spice_amd.pruning.eval import SegmentationMetrics
    from spice_amd.pruning.eval.components import CoreEvaluator
    torch.manual_seed(3)
    model = nets.create_model("Unet", "resnet50", 5).to(dev).eval()
    loader = [(torch.randn(2, 3, 64, 64), torch.randint(0, 5, (2, 64, 64))) for _ in range(2)]
    ev = CoreEvaluator(dev, 5, 64)
    fused = ev.evaluate_single_model(model, loader, SegmentationMetrics(5, dev), {"model_name": "UNET_resnet"})

    class Unfused:  # a metrics object without the logits entry point takes the arg-max route
        def __init__(self, inner):
            self.inner = inner

        def update(self, predictions, targets):
            self.inner.update(predictions, targets)

        def compute_metrics(self):
            return self.inner.compute_metrics()

    plain = ev.evaluate_single_model(model, loader, Unfused(SegmentationMetrics(5, dev)), {"model_name": "UNET_resnet"})
    # two evaluations = two forwards: MIOpen may pick another convolution algorithm on the later call, which can
    # flip the arg-max of a pixel whose best logits are a rounding error apart (two cells per pixel)
    assert int(np.abs(np.asarray(fused["confusion_matrix"]) - np.asarray(plain["confusion_matrix"])).sum()) <= 8
    assert fused["miou"] == pytest.approx(plain["miou"], abs=2e-3) and int(fused["confusion_matrix"].sum()) == 2 * 2 * 64 * 64
    assert fused["params_percentage"] == pytest.approx(100.0) and fused["mac_reduction"] == pytest.approx(0.0, abs=1e-9)
    assert fused["total_params"] == sum(p.numel() for p in model.parameters()) and fused["gmacs"] > 0


def test_two_rank_prune_and_finetune(dev, tmp_path):
    """Two ranks sharing this GPU (gloo rehearsal of the one-process-per-GPU path): sharded Taylor
    analysis, identical slicing, bucketed gradient averaging -> bit-identical replicas."""
    import subprocess
    import sys
    env = dict(os.environ, SPICE_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(os.path.dirname(__file__), "ddp_prune_worker.py"), str(tmp_path)]
    done = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    assert done.returncode == 0, done.stdout[-3000:] + done.stderr[-3000:]
    r0, r1 = (json.load(open(tmp_path / f"rank{r}.json")) for r in range(2))
    assert r0["world"] == 2 and r0["digest"] == r1["digest"]
    assert r0["final_params"] == r1["final_params"] < r0["initial_params"]
    run_dir = tmp_path / "UNET_resnet_magnitude_taylor_0.40_kd"
    for f in ("final_model.pth", "metrics.json", "model_info.txt", "hybrid_pruning_report.json"):
        assert (run_dir / f).exists(), f


def test_two_rank_sharded_evaluation(dev, tmp_path):
    """Evaluation batches dealt to two ranks + one all-reduce of the int64 confusion matrix give
    every rank exactly the single-process metrics."""
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(__file__), "ddp_eval_worker.py")
    one = subprocess.run([sys.executable, worker, str(tmp_path)], capture_output=True, text=True, timeout=300,
                         env={k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert one.returncode == 0, one.stdout[-2000:] + one.stderr[-2000:]
    env = dict(os.environ, SPICE_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), worker, str(tmp_path)],
                         env=env, capture_output=True, text=True, timeout=300)
    assert two.returncode == 0, two.stdout[-2000:] + two.stderr[-2000:]
    ref = json.load(open(tmp_path / "eval_rank0_of1.json"))
    for r in range(2):
        got = json.load(open(tmp_path / f"eval_rank{r}_of2.json"))
        assert got == ref
    assert 0.0 <= ref["miou"] <= 1.0 and len(ref["per_class_iou"]) == 6


def test_rccl_single_rank_collectives(dev):
    """backend "nccl" (RCCL) on the one GPU at hand: process-group set-up through dist.init_from_env, the
    int64 / fp32 / fp64 all-reduces of the sweep and the Taylor pass on device buffers, the asynchronous
    bucketed gradient all-reduce.  With one rank the sums are identities; what is exercised is that RCCL
    loads, initialises and runs this path's collectives on HIP memory (several GPUs are not available here)."""
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(__file__), "rccl_worker.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "SPICE_DIST_BACKEND")}
    env["MASTER_PORT"] = str(_free_port())
    out = subprocess.run([sys.executable, worker], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "rccl ok" in out.stdout


def test_cli_run_prints_summary_and_writes_run_directory(dev, tmp_path, capsys):
    """`python -m pruning.cli --model M --strategy S --ratio R ...` end to end at the reference's
    default geometry (5 classes, 576x576), one step, one epoch each."""
    from spice_amd.pruning import cli
    cli.main(["--model", "FPN_resnet", "--strategy", "magnitude", "--pruning-ratio", "0.3", "--steps", "1",
              "--fine-tune-epochs", "1", "--final-fine-tune-epochs", "1", "--batch-size", "4",
              "--output-dir", str(tmp_path), "--data-root", str(tmp_path / "nodata"), "--seed", "3"])
    out = capsys.readouterr().out
    assert "PRUNING COMPLETE" in out and "Model: FPN_resnet" in out and "Strategy: magnitude" in out
    run = tmp_path / "FPN_resnet_magnitude_0.30"
    assert (run / "final_model.pth").exists() and (run / "metrics.json").exists()
    assert (run / "intermediate" / "FPN_resnet_step_1_of_1.pth").exists()
    assert "Pruning Ratio: 0.3" in (run / "model_info.txt").read_text()


def test_raw_ctypes_binding_of_the_library(dev):
    """A maintainer's ctypes binding against the raw library (no spice_amd.ops): the reference's
    update_confusion_matrix body replaced by one spice_confusion_update call."""
    import ctypes
    lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                   "spice_amd", "lib", "libspice_amd.so"))
    lib.spice_confusion_workspace_bytes.restype = ctypes.c_size_t
    lib.spice_confusion_workspace_bytes.argtypes = [ctypes.c_int64, ctypes.c_int32]
    lib.spice_confusion_update.restype = ctypes.c_int32
    lib.spice_confusion_update.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_int32,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    ws_cache = {}

    def update_confusion_matrix(conf, preds, target, num_classes):
        p, t = preds.reshape(-1).contiguous(), target.reshape(-1).contiguous()
        key = (conf.device.index, num_classes)
        if key not in ws_cache:
            n = lib.spice_confusion_workspace_bytes(p.numel(), num_classes)
            ws_cache[key] = torch.zeros(n, dtype=torch.uint8, device=conf.device)
        ws = ws_cache[key]
        rc = lib.spice_confusion_update(p.data_ptr(), t.data_ptr(), p.numel(), num_classes, conf.data_ptr(),
                                        ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream)
        if rc:
            raise RuntimeError(f"spice_confusion_update failed: {rc}")

    g = torch.Generator().manual_seed(9)
    c = 19
    conf = torch.zeros(c, c, dtype=torch.int64, device=dev)
    want = np.zeros((c, c), dtype=np.int64)
    for _ in range(2):
        t = torch.randint(0, c, (3, 50, 70), generator=g)
        p = torch.randint(0, c, (3, 50, 70), generator=g)
        t[torch.rand(3, 50, 70, generator=g) < 0.1] = 255
        update_confusion_matrix(conf, p.to(dev), t.to(dev), c)
        oc.update_confusion(want, p.numpy(), t.numpy(), c)
    np.testing.assert_array_equal(conf.cpu().numpy(), want)
    assert lib.spice_confusion_update(None, None, 5, c, None, None, 0, None) != 0  # argument errors are return codes


def test_pruned_model_evaluator_walks_a_directory_of_runs(dev, tmp_path, capsys):
    """pruning/eval/evaluator.py:40-192: every run directory is found, loaded, evaluated, printed
    and saved; a second evaluator over the same directories takes every result from the cache."""
    import pandas as pd
    from spice_amd.pruning.eval import EvaluationConfig, PrunedModelEvaluator
    from spice_amd.pruning.models import ModelFactory
    torch.manual_seed(5)
    runs = tmp_path / "pruned"
    full = nets.create_model("Unet", "resnet50", 5).to(dev).eval()
    slim = copy.deepcopy(full)
    cfg = PruningConfig(model_name="UNET_resnet", num_classes=5, use_wandb=False, output_dir=str(tmp_path / "unused"))
    engine.MagnitudePruner(slim, torch.randn(1, 3, 64, 64, device=dev), pruning_ratio=0.5,
                           ignored_layers=ModelFactory.get_ignored_layers(slim, cfg)).step()
    for name, model in (("UNET_resnet_magnitude_0.50", slim), ("UNET_resnet_random_0.00_kd", full)):
        (runs / name).mkdir(parents=True)
        torch.save(model, runs / name / "final_model.pth")
    (runs / "UNET_resnet_taylor_0.20").mkdir()   # no final model: not a run
    (runs / "UNET_resnet_lamp_0.30").mkdir()
    (runs / "UNET_resnet_lamp_0.30" / "final_model.pth").write_bytes(b"not a checkpoint")  # fails to load, run goes on
    loader = [(torch.randn(2, 3, 64, 64), torch.randint(0, 5, (2, 64, 64))) for _ in range(2)]
    config = EvaluationConfig(pruned_models_dir=str(runs), output_dir=str(tmp_path / "results"),
                              visualization_dir=str(tmp_path / "plots"), img_size=64, batch_size=2, num_classes=5,
                              device="cuda", use_wandb=False, generate_visualizations=False)
    PrunedModelEvaluator(config).evaluate_all_models(eval_loader=loader)
    out = capsys.readouterr().out
    assert "Evaluating [1/3]" in out and "EVALUATION SUMMARY" in out and out.count("Per-class IoU:") == 2
    table = pd.read_csv(tmp_path / "results" / "evaluation_summary.csv")
    assert list(table["strategy"]) == ["random", "magnitude"] and list(table["pruning_ratio"]) == [0.0, 0.5]
    assert list(table["fine_tune_method"]) == ["KD-Lite", "Standard"]
    by = {r["strategy"]: r for _, r in table.iterrows()}
    assert by["random"]["params_percentage"] == pytest.approx(100.0) and by["magnitude"]["params_percentage"] < 60.0
    assert by["magnitude"]["mac_reduction"] > 30.0
    for name, model in (("UNET_resnet_magnitude_0.50", slim), ("UNET_resnet_random_0.00_kd", full)):
        saved = json.load(open(tmp_path / "results" / name / "metrics.json"))
        conf = np.asarray(saved["confusion_matrix"])
        want = np.zeros((5, 5), dtype=np.int64)
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            for x, y in loader:
                oc.update_confusion(want, oc.argmax_first(model(x.to(dev)).float().cpu().numpy()), y.numpy(), 5)
        assert int(np.abs(conf - want).sum()) <= 8   # a second forward: see test_core_evaluator_fused_equals_unfused
        assert conf.sum() == want.sum() and np.array_equal(conf.sum(1), want.sum(1))   # same pixels, same label histogram
        assert saved["miou"] == pytest.approx(oc.macro_metrics_present_classes(conf)["miou"], rel=1e-5)
        assert saved["total_params"] == sum(p.numel() for p in model.parameters())
    assert not (tmp_path / "results" / "UNET_resnet_lamp_0.30").exists()

    class NoData:
        def __iter__(self):
            raise AssertionError("a cached run must not be evaluated again")

    again = PrunedModelEvaluator(config)
    again.evaluate_all_models(eval_loader=NoData())
    rerun = pd.read_csv(tmp_path / "results" / "evaluation_summary.csv")
    assert list(rerun["miou"]) == list(table["miou"]) and len(again.result_processor.results) == 2


def test_run_eval_command_line_one_and_two_ranks(dev, tmp_path, capsys):
    """`python -m pruning.eval.run_eval --pruned-dir D ...` (pruning/eval/run_eval.py:155-245) in
    this process, then under torch.distributed.run with the models dealt to two ranks: same table."""
    import subprocess
    import sys
    import pandas as pd
    from spice_amd.pruning.eval import run_eval
    torch.manual_seed(6)
    runs = tmp_path / "pruned"
    for k, name in enumerate(("FPN_resnet_magnitude_0.10", "FPN_resnet_taylor_0.20_kd", "FPN_resnet_lamp_0.30_kd_refine")):
        (runs / name).mkdir(parents=True)
        torch.save(nets.create_model("FPN", "resnet50", 5), runs / name / "final_model.pth")
    common = ["--pruned-dir", str(runs), "--data-root", str(tmp_path / "nodata"), "--batch-size", "8", "--device", "cuda",
              "--no-visualize", "--no-wandb", "--vis-dir", str(tmp_path / "plots")]
    run_eval.main(common + ["--output-dir", str(tmp_path / "one")])
    out = capsys.readouterr().out
    assert "Evaluating [3/3]" in out and "Evaluation completed!" in out
    one = pd.read_csv(tmp_path / "one" / "evaluation_summary.csv")
    assert list(one["strategy"]) == ["magnitude", "taylor", "lamp"] and (one["params_percentage"] == 100.0).all()
    assert list(one["fine_tune_method"]) == ["Standard", "KD-Lite", "Standard+KD"]

    env = dict(os.environ, SPICE_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "-m", "spice_amd.pruning.eval.run_eval"] + common + ["--output-dir", str(tmp_path / "two")]
    done = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420,
                          cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert done.returncode == 0, done.stdout[-3000:] + done.stderr[-3000:]
    assert done.stdout.count("Evaluation completed!") == 1 and done.stdout.count("EVALUATION SUMMARY") == 1
    two = pd.read_csv(tmp_path / "two" / "evaluation_summary.csv")
    assert list(two["dir_name"]) == list(one["dir_name"]) and list(two["total_params"]) == list(one["total_params"])
    assert list(two["total_macs"]) == list(one["total_macs"])
    # separate processes may pick different convolution algorithms: predictions agree up to rounding flips
    np.testing.assert_allclose(two["miou"], one["miou"], atol=5e-3)
    for name in one["dir_name"]:
        a = np.asarray(json.load(open(tmp_path / "one" / name / "metrics.json"))["confusion_matrix"])
        b = np.asarray(json.load(open(tmp_path / "two" / name / "metrics.json"))["confusion_matrix"])
        assert a.sum() == b.sum() and np.array_equal(a.sum(1), b.sum(1))   # same pixels, same label histogram
    # a second run takes everything from the cache written by rank 0
    run_eval.main(common + ["--output-dir", str(tmp_path / "two")])
    assert "Evaluating [" not in capsys.readouterr().out
    run_eval.main(common + ["--output-dir", str(tmp_path / "two"), "--clear-cache", "--single-model", "FPN_resnet_taylor_0.20_kd"])
    out = capsys.readouterr().out
    assert "Evaluation cache cleared" in out and "Model: FPN_resnet_taylor_0.20_kd" in out


def test_baseline_run_eval_prints_the_table(dev, tmp_path, capsys):
    """baseline/eval/run_eval.py:31-115: models whose checkpoint exists are evaluated, the others
    skipped; the table is printed in the reference's layout."""
    from spice_amd.baseline.eval import run_eval
    from spice_amd.baseline.eval.model_factory import get_model_by_name
    torch.manual_seed(9)
    for name in ("UNET_resnet", "FPN_resnet"):
        model, rel = get_model_by_name(name)
        (tmp_path / rel).parent.mkdir(parents=True)
        torch.save(model.state_dict(), tmp_path / rel)
    loader = [(torch.randn(2, 3, 64, 64), torch.randint(0, 5, (2, 64, 64))) for _ in range(2)]
    results = run_eval.main(root=str(tmp_path), eval_loader=loader)
    out = capsys.readouterr().out
    assert [r["model"] for r in results] == ["UNET_resnet", "FPN_resnet"]
    assert f"{'Model':<30} {'mIoU':<10} {'Mean Acc':<10}" in out
    for r in results:
        assert f"{r['model']:<30} {r['miou']:<10.4f} {r['mean_acc']:<10.4f}" in out
        assert 0.0 <= r["miou"] <= 1.0 and len(r["per_class_iou"]) == 5


def test_sweep_prefix_reuse_gives_the_tables_of_the_whole_forward(dev):
    """Every candidate of the DeepLabV3Plus_resnet 21-class sweep on the benchmark's 8x3x512x512 batch, forwarded from
    the first block its group touches (the activations in front of it taken from the un-pruned network's forward of
    the same run) and forwarded whole: all 414 confusion matrices identical, losses to 1e-5 (the loss kernel sums
    its workgroups' partial sums in the order they finish).  The blocks in front of a group's first module see the
    same inputs and weights as in the un-pruned network and run the same deterministic kernels."""
    from spice_amd.sweep import DEFAULT_RATIOS
    torch.manual_seed(1234)
    model = nets.create_model("DeepLabV3Plus", "resnet50", 21).to(dev).eval()
    ignored = [m for m in model.segmentation_head.modules() if isinstance(m, nn.Conv2d)]
    x = torch.from_numpy(synth.normal(1234, 50, 8 * 3 * 512 * 512).reshape(8, 3, 512, 512)).to(dev)
    lab = torch.from_numpy(synth.label_map(1234, 8, 512, 512, 21, "blocky", ignore_share=0.05)).to(dev)
    results, reused = [], []
    for reuse in (True, False):
        sweep = SensitivitySweep(model, 21, DEFAULT_RATIOS, ignored_layers=ignored)
        sweep.prefix_reuse = reuse
        assert sweep.enable_lanes(2) == 2
        results.append(sweep.run(x, lab))
        reused.append(sweep.units_reused)
        assert len(sweep.candidates) == 414
    assert reused[1] == 0 and reused[0] > 414 * 5          # 19 units; the average candidate starts past the middle
    assert torch.equal(results[0]["confusion"], results[1]["confusion"])
    assert torch.allclose(results[0]["kd"], results[1]["kd"], rtol=1e-5, atol=1e-7)
    assert int(results[0]["confusion"].sum()) == 414 * int((lab != 255).sum())
    # first units: the stem's group starts at unit 0, a decoder group behind every encoder unit
    first = [sweep.first_unit(c.index) for c in sweep.candidates]
    assert first[0] == 0 and max(first) >= 17


def test_sweep_suffix_graphs_give_the_tables_of_the_eager_forward(dev):
    """The unchanged blocks BEHIND a candidate's group as one launch of a HIP graph captured on the un-pruned network:
    every candidate of the 414-candidate sweep on the benchmark's batch, graphs captured on ANOTHER batch of the same
    geometry (the recorded activations keep their addresses from batch to batch), against the same sweep with every
    launch queued by the host: identical confusion matrices, losses to 1e-5.  The graphs go with the session."""
    from spice_amd.sweep import DEFAULT_RATIOS
    torch.manual_seed(1234)
    model = nets.create_model("DeepLabV3Plus", "resnet50", 21).to(dev).eval()
    ignored = [m for m in model.segmentation_head.modules() if isinstance(m, nn.Conv2d)]
    x = torch.from_numpy(synth.normal(1234, 50, 8 * 3 * 512 * 512).reshape(8, 3, 512, 512)).to(dev)
    other = torch.from_numpy(synth.normal(99, 50, 8 * 3 * 512 * 512).reshape(8, 3, 512, 512)).to(dev)
    lab = torch.from_numpy(synth.label_map(1234, 8, 512, 512, 21, "blocky", ignore_share=0.05)).to(dev)
    sweep = SensitivitySweep(model, 21, DEFAULT_RATIOS, ignored_layers=ignored)
    assert sweep.enable_lanes(2) == 2
    with sweep.session():
        eager = sweep.run(x, lab)
        assert sum(l.graph_launches for l in sweep._lanes) == 0
        sweep.teacher_forward(other)
        info = sweep.capture_suffix_graphs(other)
        assert info["graphs"] == 2 * (len(sweep._tape) - 1)
        # (a bottleneck in front of a graph writes its result into the graph's input buffer itself: counted here)
        from spice_amd.nets import layers as net_layers
        taken, take = [], net_layers.take_unit_destination
        net_layers.take_unit_destination = lambda *a: (taken.append(1), take(*a))[1]
        try:
            got = sweep.run(x, lab)          # its teacher forward writes x's activations where the other batch's were
        finally:
            net_layers.take_unit_destination = take
        launches = sum(l.graph_launches for l in sweep._lanes)
        assert launches > 100, launches
        assert len(taken) > 100, len(taken)
        # a weight changed in place: the graphs (which hold the address of its old fp16 copy) are not used any more
        with torch.no_grad():
            model.encoder.layer3[0].conv1.weight.mul_(1.0)
        again = sweep.run(x, lab, indices=list(range(0, 414, 40)))
        assert sum(l.graph_launches for l in sweep._lanes) == launches
    assert not sweep._graphs and not sweep._lanes[1]._graphs
    assert torch.equal(got["confusion"], eager["confusion"])
    assert torch.allclose(got["kd"], eager["kd"], rtol=1e-5, atol=1e-7)
    pick = list(range(0, 414, 40))
    assert torch.equal(again["confusion"][pick], eager["confusion"][pick])
    # which candidates go through a graph: those whose group ends in front of the last unit
    spans = [(sweep.first_unit(c.index), sweep.last_unit(c.index)) for c in sweep.candidates]
    assert all(a <= b for a, b in spans) and any(b < len(sweep._tape) - 1 for _, b in spans)


def test_sweep_suffix_graphs_on_a_pyramid_decoder(dev):
    """FPN reads four of the encoder's feature maps far behind the blocks that made them: the graphs' inputs are the
    unit just in front of them and every feature map that left the encoder earlier."""
    torch.manual_seed(3)
    model = nets.create_model("FPN", "resnet50", 7).to(dev).eval()
    ignored = [m for m in model.segmentation_head.modules() if isinstance(m, nn.Conv2d)]
    x = torch.randn(2, 3, 256, 256, device=dev)
    lab = torch.from_numpy(synth.label_map(4, 2, 256, 256, 7, "blocky")).to(dev)
    sweep = SensitivitySweep(model, 7, ratios=(0.3, 0.6), ignored_layers=ignored)
    pick = list(range(0, len(sweep.candidates), 2))
    with sweep.session():
        eager = sweep.run(x, lab, indices=pick)
        sweep.teacher_forward(x)
        info = sweep.capture_suffix_graphs(x)
        if info["graphs"] == 0:
            pytest.skip("this network's tail does not go through the one-launch head")
        got = sweep.run(x, lab, indices=pick)
        assert sweep.graph_launches > 0
    diff = (got["confusion"] - eager["confusion"]).abs().sum(dim=(1, 2))
    assert int(diff.max()) <= 4          # (MIOpen serves part of this network and may pick another algorithm on a later call)
    assert torch.allclose(got["kd"], eager["kd"], rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("arch,enc", [("Unet", "timm-efficientnet-b3"), ("FPN", "resnet50"), ("Unet", "resnet50")])
def test_sweep_prefix_reuse_on_the_other_networks(dev, arch, enc):
    """The encoders' and the U-Net decoder's units on networks whose forward goes through MIOpen (which may pick
    another algorithm on a later call): candidates with and without prefix reuse agree up to a pixel or two on an
    arg-max tie, and a replayed forward returns the recorded activations themselves."""
    torch.manual_seed(5)
    model = nets.create_model(arch, enc, 7).to(dev).eval()
    ignored = [m for m in model.segmentation_head.modules() if isinstance(m, nn.Conv2d)]
    x = torch.randn(2, 3, 64, 64, device=dev)
    lab = torch.from_numpy(synth.label_map(4, 2, 64, 64, 7, "blocky")).to(dev)
    out = []
    for reuse in (True, False):
        sweep = SensitivitySweep(model, 7, ratios=(0.5,), ignored_layers=ignored, autocast_dtype=None)
        sweep.prefix_reuse = reuse
        pick = list(range(0, len(sweep.candidates), 3))
        out.append(sweep.run(x, lab, indices=pick))
        assert (sweep.units_reused > 0) == reuse
    diff = (out[0]["confusion"] - out[1]["confusion"]).abs().sum(dim=(1, 2))
    assert int(diff.max()) <= 4
    assert torch.allclose(out[0]["kd"], out[1]["kd"], rtol=1e-3, atol=1e-5)
    from spice_amd.nets.prefix import PrefixTape
    tape = PrefixTape(model.prefix_units())
    with torch.no_grad():
        model.attach_tape(tape.recorder())
        y0 = model(x)
        model.attach_tape(tape.replay(len(tape)))
        y1 = model(x)
        model.attach_tape(None)
    assert tape.complete() and torch.allclose(y0, y1, rtol=1e-4, atol=1e-5)


def test_sweep_plans_made_between_the_launches(dev):
    """With captured graphs run() makes a candidate's launch plan in front of its launches (uploaded by a launch of
    spice_gather_upload, which does not hold the host) instead of all plans up front: same tables; and the choice
    falls back to plans + solver look-ups up front where a candidate may bring MIOpen a new convolution problem."""
    torch.manual_seed(7)
    model = nets.create_model("DeepLabV3Plus", "resnet50", 21).to(dev).eval()
    ignored = [m for m in model.segmentation_head.modules() if isinstance(m, nn.Conv2d)]
    x = torch.randn(2, 3, 256, 256, device=dev)
    lab = torch.from_numpy(synth.label_map(5, 2, 256, 256, 21, "blocky", ignore_share=0.05)).to(dev)
    tables = []
    for up_front in (True, False):
        sweep = SensitivitySweep(model, 21, ratios=(0.25, 0.5, 0.75), ignored_layers=ignored)
        sweep.enable_lanes(2)
        sweep.plans_up_front = up_front
        assert not sweep.may_need_solver_lookups(x)        # every convolution a group touches is the library's
        with sweep.session():
            sweep.teacher_forward(x)
            assert sweep.capture_suffix_graphs(x)["graphs"] > 0
            assert not sweep.jobs and not sweep._lanes[1].jobs
            out = sweep.run(x, lab)
            plans_s = sweep.last_share["phases"]["plans_s"]
            assert (plans_s > 0.0) == up_front
            assert len(sweep.jobs) + len(sweep._lanes[1].jobs) == len(sweep.candidates)
            assert sum(l.graph_launches for l in sweep._lanes) > 0
        tables.append(out)
    assert torch.equal(tables[0]["confusion"], tables[1]["confusion"])
    # (the loss kernel sums its workgroups' partial sums in the order they finish)
    assert torch.allclose(tables[0]["kd"], tables[1]["kd"], rtol=1e-5, atol=1e-7)
    fp32 = SensitivitySweep(model, 21, ratios=(0.5,), ignored_layers=ignored, autocast_dtype=None)
    assert fp32.may_need_solver_lookups(x)                  # the fp32 forward runs on torch's convolutions


def test_sweep_captures_its_graphs_from_the_second_run_of_a_held_session(dev):
    """An analysis over several batches inside one session: the first run() queues every launch itself, the second
    one captures the graphs (they live until the session ends) and runs on them; tables equal a sweep that never does."""
    from spice_amd.sweep import DEFAULT_RATIOS
    torch.manual_seed(11)
    model = nets.create_model("DeepLabV3Plus", "resnet50", 21).to(dev).eval()
    ignored = [m for m in model.segmentation_head.modules() if isinstance(m, nn.Conv2d)]
    xs = [torch.randn(2, 3, 256, 256, device=dev) for _ in range(2)]
    labs = [torch.from_numpy(synth.label_map(s, 2, 256, 256, 21, "blocky", ignore_share=0.05)).to(dev) for s in (1, 2)]
    plain = SensitivitySweep(model, 21, DEFAULT_RATIOS, ignored_layers=ignored)
    plain.suffix_graphs = False
    want = [plain.run(x, l) for x, l in zip(xs, labs)]
    sweep = SensitivitySweep(model, 21, DEFAULT_RATIOS, ignored_layers=ignored)
    sweep.run(xs[0], labs[0])                       # no session held: nothing captured, however often it runs
    sweep.run(xs[0], labs[0])
    assert sweep.graph_launches == 0 and not sweep._graphs
    sweep._runs_on.clear()
    with sweep.session():
        first = sweep.run(xs[0], labs[0])
        assert sweep.graph_launches == 0 and not sweep._graphs
        second = sweep.run(xs[1], labs[1])
        assert sweep.graph_launches > 100 and sweep._graphs
        assert "graph_capture_s" in sweep.last_share["phases"] and sweep.last_share["phases"]["plans_s"] == 0.0
    assert not sweep._graphs
    for got, ref in ((first, want[0]), (second, want[1])):
        assert torch.equal(got["confusion"], ref["confusion"])
        assert torch.allclose(got["kd"], ref["kd"], rtol=1e-5, atol=1e-7)
