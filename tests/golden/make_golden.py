"""Generate the golden fixtures in this directory from the reference implementation.

Run in the build container only (the reference does not travel):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference's hot-path modules are imported from /root/reference by file path.  Third-party
packages the reference imports at module top but that are absent here (torch_pruning, torchmetrics,
segmentation_models_pytorch, wandb, seaborn, torchvision, tqdm if missing) are satisfied with EMPTY
placeholder modules so that the import statements succeed; no arithmetic of those packages is
emulated and no fixture depends on them: every stored output is computed by in-tree reference code
(pruning/losses.py, pruning/sensitivity_utils.py, pruning/sensitivity.py, pruning/pruner.py
LayerQuotaCalculator, baseline/eval/metrics.py, pruning/eval/metrics.py::compute_metrics).

Each .npz stores inputs AND outputs, so the tests never depend on RNG streams.
"""
from __future__ import annotations

import importlib
import importlib.util
import os
import sys
import types

sys.dont_write_bytecode = True

import numpy as np
import torch
import torch.nn as nn

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))  # tests/
from toy_models import ToyDictNet, ToyNet  # noqa: E402


def load_by_path(name: str, rel: str):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def install_placeholders():
    def empty(name):
        m = types.ModuleType(name)
        m.__path__ = []
        sys.modules[name] = m
        return m

    tp = empty("torch_pruning")
    tp.importance = empty("torch_pruning.importance")
    tp.importance.Importance = type("Importance", (), {})
    tp.dependency = empty("torch_pruning.dependency")
    tp.dependency.Group = list
    tp.pruner = empty("torch_pruning.pruner")
    tp.pruner.MagnitudePruner = type("MagnitudePruner", (), {})
    tp.utils = empty("torch_pruning.utils")
    for name in ("torchmetrics", "segmentation_models_pytorch", "wandb", "seaborn"):
        if importlib.util.find_spec(name) is None:
            empty(name)
    if importlib.util.find_spec("torchvision") is None:
        tv = empty("torchvision")
        tv.transforms = empty("torchvision.transforms")
        tv.transforms.functional = empty("torchvision.transforms.functional")


def gen(seed):
    return torch.Generator().manual_seed(seed)


# ------------------------------------------------------------------------------------------
def make_kdlite(losses):
    cases = [
        # name, N, C, H, W, T, alpha, bw, conf, special
        ("c21_default", 2, 21, 16, 16, 4.0, 0.7, 1.0, False, None),
        ("c21_t3_a05", 2, 21, 16, 16, 3.0, 0.5, 1.0, False, None),
        ("c5_weighted", 2, 5, 16, 16, 3.0, 0.5, 1.0, False, None),
        ("c5_boundary", 2, 5, 16, 16, 4.0, 0.7, 2.0, False, "blocky"),
        ("c21_boundary", 2, 21, 16, 16, 4.0, 0.7, 2.5, False, "blocky"),
        ("c21_conf", 2, 21, 16, 16, 4.0, 0.7, 1.0, True, None),
        ("c5_boundary_conf", 1, 5, 12, 20, 3.0, 0.5, 2.0, True, "blocky"),
        ("c19_odd_hw", 1, 19, 15, 17, 4.0, 0.7, 1.0, False, None),
        ("c19_odd_hw_bconf", 1, 19, 15, 17, 2.0, 0.3, 3.0, True, "blocky"),
        ("c21_all_ignored", 1, 21, 8, 8, 4.0, 0.7, 1.0, False, "all_ignored"),
        ("c21_same_logits", 1, 21, 8, 8, 4.0, 0.7, 1.0, False, "same"),
        ("c3_single255", 1, 3, 8, 8, 4.0, 0.5, 2.0, False, "single255"),
    ]
    out = {}
    names = []
    for i, (name, n, c, h, w, T, alpha, bw, conf, special) in enumerate(cases):
        g = gen(1000 + i)
        s = 3.0 * torch.randn(n, c, h, w, generator=g)
        t = s + torch.randn(n, c, h, w, generator=g)
        if special == "blocky":
            yy, xx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing="ij")
            lab = (((yy // 4) * 7 + (xx // 4) * 3) % c).expand(n, h, w).clone()
        elif special == "single255":
            lab = torch.ones(n, h, w, dtype=torch.long)
        else:
            lab = torch.randint(0, c, (n, h, w), generator=g)
        if special == "all_ignored":
            lab[:] = 255
        elif special == "single255":
            lab[0, 4, 4] = 255
        else:
            drop = torch.rand(n, h, w, generator=g) < 0.08
            lab[drop] = 255
        if special == "same":
            t = s.clone()
        sg = s.clone().requires_grad_(True)
        fn = losses.KDLiteLoss(temperature=T, alpha=alpha, ignore_index=255, boundary_weight=bw,
                               confidence_weight=conf, num_classes=c)
        res = fn(sg, t, lab)
        if torch.isfinite(res["total"]):
            (grad,) = torch.autograd.grad(res["total"], sg)
        else:
            grad = torch.zeros_like(s)
        if special == "single255":
            out[f"{name}/boundary_mask"] = fn._get_boundary_mask(lab).numpy()
        names.append(name)
        out[f"{name}/student"] = s.numpy()
        out[f"{name}/teacher"] = t.numpy()
        out[f"{name}/labels"] = lab.numpy()
        out[f"{name}/params"] = np.array([T, alpha, bw, float(conf), float(c)], dtype=np.float64)
        out[f"{name}/out3"] = np.array([res["total"].item(), res["ce"].item(), res["kd"].item()],
                                       dtype=np.float32)
        out[f"{name}/grad"] = grad.numpy()
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "kdlite.npz"), **out)
    print("kdlite.npz", len(names), "cases")


# ------------------------------------------------------------------------------------------
def make_confusion(base_metrics, eval_metrics):
    out, names = {}, []
    specs = [("c5_uniform", 5, 4000, "uniform"), ("c19_uniform", 19, 6001, "uniform"),
             ("c21_blocky", 21, 4096, "blocky"), ("c21_uniform", 21, 5003, "uniform"),
             ("c2_tiny", 2, 7, "uniform")]
    for i, (name, c, n, kind) in enumerate(specs):
        g = gen(2000 + i)
        conf = torch.zeros(c, c, dtype=torch.long)
        steps = []
        for k in range(2):
            if kind == "blocky":
                idx = torch.arange(n)
                tgt = ((idx // 64) * 7 + k) % c
            else:
                tgt = torch.randint(0, c, (n,), generator=g)
            pred = tgt.clone()
            flip = torch.rand(n, generator=g) < 0.2
            pred[flip] = torch.randint(0, c, (int(flip.sum()),), generator=g)
            # invalid targets of every kind the reference filters: -1, 255, >= C
            bad = torch.rand(n, generator=g)
            tgt = tgt.clone()
            tgt[bad < 0.03] = -1
            tgt[(bad >= 0.03) & (bad < 0.08)] = 255
            tgt[(bad >= 0.08) & (bad < 0.10)] = c
            base_metrics.update_confusion_matrix(conf, pred.view(2, -1) if n % 2 == 0 else pred,
                                                 tgt.view(2, -1) if n % 2 == 0 else tgt, c)
            out[f"{name}/pred{k}"] = pred.numpy()
            out[f"{name}/target{k}"] = tgt.numpy()
            out[f"{name}/conf{k}"] = conf.clone().numpy()
            steps.append(k)
        iou, acc, miou = base_metrics.compute_metrics_from_confusion(conf)
        out[f"{name}/iou"] = iou
        out[f"{name}/acc"] = acc
        out[f"{name}/miou"] = np.array([miou], dtype=np.float64)
        m = eval_metrics.compute_metrics(conf)
        out[f"{name}/pc_iou"] = np.array(m["per_class_iou"], dtype=np.float64)
        out[f"{name}/pc_acc"] = np.array(m["per_class_acc"], dtype=np.float64)
        out[f"{name}/pc_means"] = np.array([m["miou"], m["mean_acc"]], dtype=np.float64)
        out[f"{name}/num_classes"] = np.array([c])
        names.append(name)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "confusion.npz"), **out)
    print("confusion.npz", len(names), "cases")


# ------------------------------------------------------------------------------------------
def make_taylor(sens_utils):
    out = {}
    g = gen(3000)
    net = ToyNet()
    with torch.no_grad():
        for p in net.parameters():
            p.copy_(torch.randn(p.shape, generator=g) * 0.3)
        for b in (net.bn1, net.bn2):
            b.running_mean.copy_(torch.randn(b.running_mean.shape, generator=g) * 0.1)
            b.running_var.copy_(torch.rand(b.running_var.shape, generator=g) + 0.5)
    for k, v in net.state_dict().items():
        out[f"state/{k}"] = v.numpy()
    batches = []
    for b in range(3):
        x = torch.randn(2, 3, 12, 12, generator=g)
        y = torch.randint(0, 4, (2, 12, 12), generator=g)
        batches.append((x, y))
        out[f"batch{b}/images"] = x.numpy()
        out[f"batch{b}/targets"] = y.numpy()
    conv_names = [n for n, m in net.named_modules() if isinstance(m, nn.Conv2d)]
    out["conv_names"] = np.array(conv_names)

    def run(model, loader, max_batches, tag):
        model.train()  # the pass must restore this
        scores = sens_utils.compute_taylor_scores(model, loader, "cpu", max_batches=max_batches)
        assert model.training
        by_name = {n: m for n, m in model.named_modules()}
        for n in conv_names:
            out[f"{tag}/{n}"] = scores[by_name[n]].detach().numpy()
        rep = sens_utils.get_sensitivity_report(scores)
        out[f"{tag}/report_means"] = np.array([s["mean_sensitivity"] for s in rep["layer_stats"]])
        out[f"{tag}/report_stds"] = np.array([s["std_sensitivity"] for s in rep["layer_stats"]])

    run(net, batches, 2, "tuple_max2")          # (images, targets) tuples, stops after 2 of 3
    run(net, batches, 100, "tuple_all")         # all 3
    run(net, [b[0] for b in batches], 100, "bare")  # bare tensors -> |out|.mean() loss
    dnet = ToyDictNet()
    dnet.load_state_dict(net.state_dict())
    run(dnet, batches, 100, "dict_out")         # model returning {"out": ...}
    np.savez_compressed(os.path.join(HERE, "taylor.npz"), **out)
    print("taylor.npz")


# ------------------------------------------------------------------------------------------
RESNETISH = [  # (out, in, kh, kw) of a ResNet-50-like stack plus tiny layers at/below min_channels
    (64, 3, 7, 7), (64, 64, 1, 1), (64, 64, 3, 3), (256, 64, 1, 1), (256, 64, 1, 1),
    (128, 256, 1, 1), (128, 128, 3, 3), (512, 128, 1, 1), (512, 256, 1, 1),
    (256, 512, 1, 1), (256, 256, 3, 3), (1024, 256, 1, 1), (1024, 512, 1, 1),
    (512, 1024, 1, 1), (512, 512, 3, 3), (2048, 512, 1, 1), (2048, 1024, 1, 1),
    (256, 2048, 1, 1), (48, 256, 1, 1), (256, 304, 3, 3), (8, 256, 1, 1), (5, 256, 1, 1),
    (12, 16, 3, 3), (9, 9, 1, 1),
]


def make_quota(pruner_mod):
    out, names = {}, []
    out["layers"] = np.array(RESNETISH, dtype=np.int64)
    model = nn.Sequential(*[nn.Conv2d(i, o, (kh, kw), bias=False) for (o, i, kh, kw) in RESNETISH])
    convs = list(model)
    configs = [("r05_a07", 0.5, 0.7, 8, 0), ("r07_a07", 0.7, 0.7, 8, 1), ("r05_a0", 0.5, 0.0, 8, 2),
               ("r09_a07", 0.9, 0.7, 8, 3), ("r03_a1_m16", 0.3, 1.0, 16, 4),
               ("r05_spiky", 0.5, 0.7, 8, 5)]
    for name, ratio, alpha, min_ch, k in configs:
        g = gen(4000 + k)
        scores = {}
        for li, m in enumerate(convs):
            v = torch.rand(m.out_channels, generator=g) * (10.0 ** float(torch.randint(-4, 1, (1,), generator=g)))
            if name == "r05_spiky" and li % 5 == 0:
                v = v * 1e-6  # nearly insensitive layers soak up their whole capacity
            scores[m] = v
        fake = types.SimpleNamespace(sensitivity_scores=scores)
        quotas, sens = pruner_mod.LayerQuotaCalculator.compute_layer_quotas(
            model, fake, ratio, alpha=alpha, min_channels=min_ch)
        out[f"{name}/params"] = np.array([ratio, alpha, float(min_ch)], dtype=np.float64)
        for li, m in enumerate(convs):
            out[f"{name}/scores{li}"] = scores[m].numpy()
        out[f"{name}/quota"] = np.array([quotas.get(m, -1.0) for m in convs], dtype=np.float64)
        out[f"{name}/sensitivity"] = np.array([sens.get(m, -1.0) for m in convs], dtype=np.float64)
        # the hybrid pruner's report rows (pruning/pruner.py:290-306)
        hyb = pruner_mod.HybridTaylorMagnitudePruner(types.SimpleNamespace())
        hyb.layer_quotas, hyb.layer_sensitivities = quotas, sens
        summ = hyb.get_pruning_summary()
        out[f"{name}/summary_prune"] = np.array([d["channels_to_prune"] for d in summ["layer_details"]])
        out[f"{name}/summary_out"] = np.array([d["out_channels"] for d in summ["layer_details"]])
        names.append(name)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "quota.npz"), **out)
    print("quota.npz", len(names), "cases")


# ------------------------------------------------------------------------------------------
def make_group_importance(sens_mod):
    """TaylorWeightImportance.__call__ / GroupSyncTaylorImportance.__call__ on hand-made groups:
    a group is a list of (dep, idxs) with dep.target.module (pruning/sensitivity.py:70-112,139-171)."""
    out = {}
    g = gen(5000)
    mods = {"a": nn.Conv2d(4, 16, 1), "b": nn.Conv2d(16, 16, 3), "c": nn.Conv2d(16, 24, 1),
            "short": nn.Conv2d(4, 10, 1), "bn": nn.BatchNorm2d(16), "unscored": nn.Conv2d(4, 16, 1)}
    scores = {mods[k]: torch.rand(mods[k].out_channels, generator=g) for k in ("a", "b", "c", "short")}
    for k in ("a", "b", "c", "short"):
        out[f"scores/{k}"] = scores[mods[k]].numpy()

    def dep(key):
        return types.SimpleNamespace(target=types.SimpleNamespace(module=mods[key]))

    groups = {
        "full": [("a", list(range(16))), ("bn", list(range(16))), ("b", list(range(16)))],
        "subset": [("a", [0, 3, 5, 15]), ("b", [0, 3, 5, 15])],
        "long_and_short": [("b", list(range(16))), ("c", list(range(16))), ("short", list(range(16)))],
        "bn_first": [("bn", list(range(16))), ("a", list(range(16)))],
        "no_conv": [("bn", list(range(16)))],
        "unscored_only": [("unscored", list(range(16)))],
        "unscored_then_scored": [("unscored", list(range(16))), ("b", list(range(16)))],
    }
    for cls_name in ("TaylorWeightImportance", "GroupSyncTaylorImportance"):
        cls = getattr(sens_mod, cls_name)
        obj = cls.__new__(cls)  # skip __init__ (it would run a calibration pass)
        obj.sensitivity_scores = scores
        for gname, members in groups.items():
            group = [(dep(k), idxs) for k, idxs in members]
            res = obj(group)
            out[f"{cls_name}/{gname}"] = res.detach().numpy().astype(np.float32)
        out[f"{cls_name}/empty"] = obj([]).numpy().astype(np.float32)
    out["group_names"] = np.array(list(groups.keys()))
    for gname, members in groups.items():
        out[f"group/{gname}/keys"] = np.array([k for k, _ in members])
        out[f"group/{gname}/idx0"] = np.array(members[0][1], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "group_importance.npz"), **out)
    print("group_importance.npz")


def main():
    torch.set_num_threads(4)
    losses = load_by_path("ref_losses", "pruning/losses.py")
    sens_utils = load_by_path("ref_sens_utils", "pruning/sensitivity_utils.py")
    base_metrics = load_by_path("ref_base_metrics", "baseline/eval/metrics.py")
    install_placeholders()
    eval_metrics = load_by_path("ref_eval_metrics", "pruning/eval/metrics.py")
    sys.path.insert(0, REF)
    sens_mod = importlib.import_module("pruning.sensitivity")
    pruner_mod = importlib.import_module("pruning.pruner")
    make_kdlite(losses)
    make_confusion(base_metrics, eval_metrics)
    make_taylor(sens_utils)
    make_quota(pruner_mod)
    make_group_importance(sens_mod)


if __name__ == "__main__":
    main()
