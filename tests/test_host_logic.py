"""Host-side logic of the drop-in API (no GPU): quota allocation, importance plug-ins, CLI/config,
strategy factory, channel coupling, partitioning, metric formulas -- against the fixtures generated
from the reference."""
import argparse
import types

import numpy as np
import pytest
import torch
import torch.nn as nn

from conftest import load_golden
from spice_amd import dist as sdist
from spice_amd import nets
from spice_amd.baseline.eval.metrics import compute_metrics_from_confusion
from spice_amd.engine.depgraph import build_groups
from spice_amd.pruning import (AVAILABLE_STRATEGIES, MODEL_CONFIGS, PruningConfig, PruningStrategyFactory)
from spice_amd.pruning.cli_helpers import build_config_from_args, create_parser, validate_args
from spice_amd.pruning.eval.metrics import compute_metrics
from spice_amd.pruning.pruner import HybridTaylorMagnitudePruner, LayerQuotaCalculator
from spice_amd.pruning.sensitivity import GroupSyncTaylorImportance, TaylorWeightImportance
from spice_amd.pruning.sensitivity_utils import get_layer_sensitivities, get_sensitivity_report
from spice_amd.pruning.strategies import ProgressivePruner

QT = load_golden("quota.npz")


@pytest.mark.parametrize("name", [str(x) for x in QT["names"]])
def test_layer_quota_calculator_matches_reference(name):
    ratio, alpha, min_ch = QT[f"{name}/params"]
    layers = QT["layers"]
    convs = [nn.Conv2d(int(i), int(o), (int(kh), int(kw)), bias=False) for o, i, kh, kw in layers]
    model = nn.Sequential(*convs)
    fake = types.SimpleNamespace(sensitivity_scores={m: torch.from_numpy(QT[f"{name}/scores{k}"])
                                                     for k, m in enumerate(convs)})
    quotas, sens = LayerQuotaCalculator.compute_layer_quotas(model, fake, float(ratio), alpha=float(alpha),
                                                             min_channels=int(min_ch))
    got_q = np.array([quotas.get(m, -1.0) for m in convs])
    got_s = np.array([sens.get(m, -1.0) for m in convs])
    np.testing.assert_array_equal(got_q, QT[f"{name}/quota"])  # exact: it decides channel counts
    np.testing.assert_array_equal(got_s, QT[f"{name}/sensitivity"])
    hyb = HybridTaylorMagnitudePruner(types.SimpleNamespace())
    hyb.layer_quotas, hyb.layer_sensitivities = quotas, sens
    rows = hyb.get_pruning_summary()["layer_details"]
    np.testing.assert_array_equal([r["channels_to_prune"] for r in rows], QT[f"{name}/summary_prune"])
    np.testing.assert_array_equal([r["out_channels"] for r in rows], QT[f"{name}/summary_out"])
    total = sum(m.out_channels for m in convs if m.out_channels > int(min_ch))
    assert sum(int(round(q * m.out_channels)) for m, q in quotas.items()) <= int(total * float(ratio))
    assert all(m.out_channels - round(q * m.out_channels) >= int(min_ch) for m, q in quotas.items())


GI = load_golden("group_importance.npz")


def _fake_groups():
    mods = {"a": nn.Conv2d(4, 16, 1), "b": nn.Conv2d(16, 16, 3), "c": nn.Conv2d(16, 24, 1),
            "short": nn.Conv2d(4, 10, 1), "bn": nn.BatchNorm2d(16), "unscored": nn.Conv2d(4, 16, 1)}
    scores = {mods[k]: torch.from_numpy(GI[f"scores/{k}"]) for k in ("a", "b", "c", "short")}
    groups = {}
    for g in [str(x) for x in GI["group_names"]]:
        keys = [str(k) for k in GI[f"group/{g}/keys"]]
        idx0 = GI[f"group/{g}/idx0"].tolist()
        groups[g] = [(types.SimpleNamespace(target=types.SimpleNamespace(module=mods[k])),
                      idx0 if i == 0 else list(range(16))) for i, k in enumerate(keys)]
    return scores, groups


@pytest.mark.parametrize("cls", [TaylorWeightImportance, GroupSyncTaylorImportance])
def test_taylor_importance_plugins_match_reference(cls):
    scores, groups = _fake_groups()
    obj = cls.__new__(cls)
    obj.sensitivity_scores = scores
    for name, group in groups.items():
        got = obj(group).detach().numpy().astype(np.float32)
        np.testing.assert_array_equal(got, GI[f"{cls.__name__}/{name}"], err_msg=name)
    assert obj([]).numel() == 0


def test_sensitivity_report_matches_reference():
    ty = load_golden("taylor.npz")
    names = [str(x) for x in ty["conv_names"]]
    convs = {"conv1": nn.Conv2d(3, 8, 3), "dw": nn.Conv2d(8, 8, 3, groups=8), "pw": nn.Conv2d(8, 12, 1),
             "head": nn.Conv2d(12, 4, 1)}
    scores = {convs[n]: torch.from_numpy(ty[f"tuple_all/{n}"]) for n in names}
    rep = get_sensitivity_report(scores)
    assert rep["num_layers"] == len(names)
    np.testing.assert_allclose([r["mean_sensitivity"] for r in rep["layer_stats"]], ty["tuple_all/report_means"], rtol=1e-6)
    np.testing.assert_allclose([r["std_sensitivity"] for r in rep["layer_stats"]], ty["tuple_all/report_stds"], rtol=1e-6)
    means = get_layer_sensitivities(scores)
    assert means[convs["pw"]] == pytest.approx(float(ty["tuple_all/pw"].mean()), rel=1e-6)


def test_confusion_metric_formulas_match_reference():
    cf = load_golden("confusion.npz")
    for name in [str(x) for x in cf["names"]]:
        conf = torch.from_numpy(cf[f"{name}/conf1"])
        iou, acc, miou = compute_metrics_from_confusion(conf)
        np.testing.assert_array_equal(iou, cf[f"{name}/iou"])
        np.testing.assert_array_equal(acc, cf[f"{name}/acc"])
        assert miou == cf[f"{name}/miou"][0]
        m = compute_metrics(conf)
        np.testing.assert_array_equal(m["per_class_iou"], cf[f"{name}/pc_iou"])
        assert m["miou"] == cf[f"{name}/pc_means"][0] and m["mean_acc"] == cf[f"{name}/pc_means"][1]


def test_cli_flags_defaults_and_alias(tmp_path):
    p = create_parser()
    a = p.parse_args(["--model", "UNET_efficientnet", "--strategy", "magnitude_taylor", "--pruning-ratio", "0.7",
                      "--enable-kd", "--kd-temperature", "4", "--kd-alpha", "0.7", "--steps", "10",
                      "--output-dir", str(tmp_path), "--norm", "1", "--flops-alpha", "0.3",
                      "--use-timestamp", "yes", "--confidence-weight", "--boundary-weight", "2.0"])
    validate_args(a)
    c = build_config_from_args(a)
    assert (c.model_name, c.pruning_strategy, c.pruning_ratio) == ("UNET_efficientnet", "magnitude_taylor", 0.7)
    assert c.enable_kd_lite and c.kd_temperature == 4.0 and c.kd_alpha == 0.7 and c.iterative_steps == 10
    assert c.importance_norm == 1 and c.flops_alpha == 0.3 and c.sensitivity_alpha == 0.7  # flag leaves alpha alone
    assert c.use_timestamp is True and c.confidence_weight is True and c.boundary_weight == 2.0
    # untouched defaults of the reference dataclass
    assert (c.num_classes, c.batch_size, c.calibration_batches, c.min_out_channels) == (5, 8, 100, 8)
    assert c.example_input_size == (1, 3, 576, 576) and c.kd_data_ratio == 0.3 and c.seed == 42
    assert c.output_path.endswith("pruned_UNET_efficientnet_magnitude_taylor_0.70_kd.pt")
    b = p.parse_args(["--model", "FPN_resnet", "--strategy", "lamp", "--ratio", "0.5"])
    assert b.ratio == 0.5
    for bad in ("0", "1", "1.5"):
        with pytest.raises(ValueError):
            validate_args(p.parse_args(["--model", "FPN_resnet", "--strategy", "lamp", "--ratio", bad]))
    with pytest.raises(SystemExit):
        p.parse_args(["--model", "nope", "--strategy", "lamp", "--ratio", "0.5"])
    with pytest.raises(ValueError, match="Unknown model"):
        PruningConfig(model_name="nope", output_dir=str(tmp_path))
    assert len(MODEL_CONFIGS) == 6 and len(AVAILABLE_STRATEGIES) == 6


def test_strategy_factory_and_progressive_pruner(tmp_path):
    assert PruningStrategyFactory.list_strategies() == ["magnitude", "random", "taylor", "lamp"]
    cfg = PruningConfig(model_name="UNET_resnet", output_dir=str(tmp_path), pruning_strategy="MAGNITUDE")
    assert type(PruningStrategyFactory.create_strategy(cfg)).__name__ == "MagnitudePruningStrategy"
    cfg.pruning_strategy = "taylor_weight"
    with pytest.raises(ValueError, match="Unknown pruning strategy"):
        PruningStrategyFactory.create_strategy(cfg)
    cfg.pruning_strategy = "magnitude"
    with pytest.raises(RuntimeError, match="Pruner not setup"):
        ProgressivePruner(PruningStrategyFactory.create_strategy(cfg), cfg).prune_step()


def test_channel_groups_of_the_two_benchmark_models():
    m = nets.create_model("DeepLabV3Plus", "resnet50", 21)
    gs = build_groups(m)
    by_root = {g.root_name: g for g in gs}
    res = by_root["encoder.layer1.0.conv3"]  # the 256-wide residual stream of layer1
    kinds = sorted((mm.name, mm.kind) for mm in res.members)
    assert ("encoder.layer1.0.downsample.0", "out") in kinds and ("encoder.layer2.0.conv1", "in") in kinds
    assert ("encoder.layer2.0.downsample.0", "in") in kinds and ("decoder.block1.0", "in") in kinds
    assert res.size == 256 and len([k for k in kinds if k[1] == "out"]) == 4
    cat = by_root["decoder.block1.0"]  # second segment of the 304-wide concatenation
    dw = [mm for mm in cat.members if mm.kind == "dw"][0]
    assert dw.seg_index == 1 and dw.offset() == 256
    assert by_root["segmentation_head.0"].frozen
    assert sum(1 for g in gs if not g.frozen) == 46
    u = nets.create_model("Unet", "timm-efficientnet-b3", 21)
    gu = {g.root_name: g for g in build_groups(u)}
    stem = gu["encoder.conv_stem"]  # stem -> depthwise -> squeeze-excite gate share one group
    assert {mm.kind for mm in stem.members} == {"out", "bn", "dw", "in"}
    assert any(mm.name.endswith("se.conv_expand") and mm.kind == "out" for mm in stem.members)


def test_partition_by_cost_properties():
    costs = [5, 1, 1, 1, 4, 4, 2, 2, 9, 3]
    for n in (1, 2, 3, 8):
        parts = sdist.partition_by_cost(costs, n)
        assert sorted(i for p in parts for i in p) == list(range(len(costs)))
        loads = [sum(costs[i] for i in p) for p in parts]
        assert max(loads) - min(loads) <= max(costs)
    assert sdist.partition_by_cost(costs, 2) == sdist.partition_by_cost(costs, 2)


def test_cli_exits_with_status_one_on_failure(tmp_path, capsys):
    """Any exception ends the command with exit status 1 (pruning/cli.py:43-48); here the
    failure is the refusal to run on a CPU device."""
    from spice_amd.pruning import cli
    with pytest.raises(SystemExit) as e:
        cli.main(["--model", "UNET_resnet", "--strategy", "magnitude", "--ratio", "0.3", "--device", "cpu",
                  "--output-dir", str(tmp_path), "--data-root", str(tmp_path / "nodata")])
    assert e.value.code == 1
    assert "PRUNING COMPLETE" not in capsys.readouterr().out


def test_run_eval_parser_defaults_and_config(tmp_path, monkeypatch):
    """pruning/eval/run_eval.py:25-177: flag names, defaults (device auto, figure dpi 600, both
    switches on) and how they land in the evaluation configuration."""
    from spice_amd.pruning.eval import run_eval
    monkeypatch.chdir(tmp_path)
    a = run_eval.parse_args([])
    assert (a.pruned_dir, a.data_root, a.output_dir, a.vis_dir) == ("pruned_models", "data", "pruning/eval/results", "pruning/eval/outputs")
    assert (a.batch_size, a.device, a.figure_dpi, a.figure_formats) == (32, "auto", 600, ["png", "pdf"])
    assert a.visualize and a.wandb and a.advanced_plots and not a.no_eval and not a.clear_cache and a.single_model is None
    assert (a.wandb_project, a.wandb_tags) == ("epic-v2", ["evaluation", "pruning"])
    b = run_eval.parse_args(["--pruned-dir", "p", "--output-dir", "o", "--vis-dir", "v", "--no-visualize", "--no-wandb",
                             "--batch-size", "4", "--device", "cuda", "--single-model", "m", "--no-eval", "--clear-cache"])
    cfg = run_eval.config_from_args(b)
    assert (cfg.pruned_models_dir, cfg.output_dir, cfg.batch_size, cfg.device) == ("p", "o", 4, "cuda")
    assert not cfg.generate_visualizations and not cfg.use_wandb and cfg.figure_dpi == 600
    assert (tmp_path / "o").is_dir() and not (tmp_path / "v").exists()
    with pytest.raises(SystemExit):
        run_eval.parse_args(["--device", "tpu"])
    run_eval.main(["--output-dir", str(tmp_path / "none"), "--no-eval", "--no-visualize", "--device", "cuda"])  # no CSV yet: message, no work


def test_baseline_eval_configuration_matches_reference():
    """baseline/eval/config.py:5-65 and model_factory.py:35-52: constants, the six entries, their
    checkpoint paths, and the error for an unknown name."""
    from spice_amd.baseline.eval import config as bc
    from spice_amd.baseline.eval.model_factory import create_model_from_config, get_model_by_name
    assert (bc.NUM_CLASSES, bc.IMG_SIZE, bc.BATCH_SIZE, bc.NUM_WORKERS) == (5, 576, 64, 16)
    assert list(bc.MODEL_CONFIGS) == ["DeepLabV3Plus_resnet", "DeepLabV3Plus_efficientnet", "UNET_resnet", "UNET_efficientnet",
                                      "FPN_resnet", "FPN_efficientnet"]
    e = bc.MODEL_CONFIGS["UNET_efficientnet"]
    assert e == {"architecture": "Unet", "encoder_name": "timm-efficientnet-b3", "encoder_weights": None, "in_channels": 3,
                 "classes": 5, "activation": None, "checkpoint_path": "checkpoints/fullsize/UNET_efficientnet/UNET_efficientnet.pt"}
    model, path = get_model_by_name("FPN_resnet")
    assert path == "checkpoints/fullsize/FPN_resnet/FPN_resnet.pt" and hasattr(model, "segmentation_head")
    with pytest.raises(ValueError, match="Unknown model"):
        get_model_by_name("nope")
    with pytest.raises(ValueError, match="Unknown architecture"):
        create_model_from_config(dict(e, architecture="PSPNet"))


def test_library_predicates_mirror_the_kernels_size_limits():
    """served_by_library / conv3x3_served_by_library say "torch's conv2d" for planes past the matrix-core kernel's
    32-bit byte offsets (one image of x and the weights below 2^30 - 2^20 bytes), where the C entry points decline."""
    from spice_amd.nets import layers
    f16 = torch.float16
    c3 = nn.Conv2d(64, 64, 3, padding=1, bias=False)
    assert layers.conv3x3_served_by_library(c3, f16, 512, 512)
    assert layers.conv3x3_served_by_library(c3, f16, 2048, 4088)        # 64 * 2048 * 4088 * 2 B just below the limit
    assert not layers.conv3x3_served_by_library(c3, f16, 2048, 4096)    # ... and just past it
    stem = layers.StemConv7x7(3, 64)
    assert stem.served_by_library(f16, 1024, 2048) and not stem.served_by_library(f16, 16384, 16384)
    down = layers.Conv1x1(256, 512, stride=2)
    assert down.served_by_library(f16, 128, 128) and not down.served_by_library(f16, 2048, 1024)
    assert not layers._within_kernel_offsets(1 << 17, 8, 9, 8, 8, 1)      # more input channels than the kernel indexes
    assert not layers._within_kernel_offsets(4096, 1 << 16, 9, 8, 8, 1)   # a weight matrix past the offsets


def test_fused_modules_keep_torch_semantics_and_channel_coupling():
    """The inference modules of the model zoo on their torch path (CPU): BatchNormAct2d == batch-norm, add,
    ReLU; Conv1x1 / DepthwiseConv3x3 == conv2d; BilinearUp == UpsamplingBilinear2d; and the dependency
    graph couples the channels of a residual block the same way whether the add is written out or sits
    inside the batch-norm module."""
    import torch.nn.functional as F
    from spice_amd.nets.decoders import BilinearUp
    from spice_amd.nets.layers import BatchNormAct2d, Conv1x1, DepthwiseConv3x3
    torch.manual_seed(0)
    x, r = torch.randn(2, 6, 5, 7), torch.randn(2, 6, 5, 7)
    bn = BatchNormAct2d(6, relu=True).eval()
    bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2.0); bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.normal_()
    plain = F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)
    assert torch.equal(bn(x, r), F.relu(plain + r)) and torch.equal(bn(x), F.relu(plain))
    assert torch.equal(BatchNormAct2d(6).eval()(x), F.batch_norm(x, torch.zeros(6), torch.ones(6), torch.ones(6), torch.zeros(6), False, 0.0, 1e-5))
    c1 = Conv1x1(6, 4)
    assert torch.equal(c1(x), F.conv2d(x, c1.weight)) and c1.takes_matrix_path(torch.float16) and not c1.takes_matrix_path(torch.float32)
    assert not Conv1x1(6, 4, stride=2).takes_matrix_path(torch.float16)
    dw = DepthwiseConv3x3(6, dilation=4)
    assert torch.equal(dw(x), F.conv2d(x, dw.weight, None, 1, 4, 4, 6))
    assert torch.equal(BilinearUp(2)(x), nn.UpsamplingBilinear2d(scale_factor=2)(x))

    class Written(nn.Module):   # the residual add written out, stock modules
        def __init__(self):
            super().__init__()
            self.stem = nn.Conv2d(3, 8, 3, padding=1)
            self.c1, self.b1 = nn.Conv2d(8, 4, 1, bias=False), nn.BatchNorm2d(4)
            self.c2, self.b2 = nn.Conv2d(4, 8, 1, bias=False), nn.BatchNorm2d(8)
            self.head = nn.Conv2d(8, 2, 1)

        def forward(self, x):
            s = self.stem(x)
            y = self.b2(self.c2(torch.relu(self.b1(self.c1(s)))))
            return self.head(torch.relu(y + s))

    class Folded(nn.Module):    # the same block with the model zoo's modules
        def __init__(self):
            super().__init__()
            self.stem = nn.Conv2d(3, 8, 3, padding=1)
            self.c1, self.b1 = Conv1x1(8, 4), BatchNormAct2d(4, relu=True)
            self.c2, self.b2 = Conv1x1(4, 8), BatchNormAct2d(8, relu=True)
            self.head = nn.Conv2d(8, 2, 1)

        def forward(self, x):
            s = self.stem(x)
            return self.head(self.b2(self.c2(self.b1(self.c1(s))), s))

    def shape(model):
        return sorted((g.size, g.frozen, sorted((m.name, m.kind) for m in g.members)) for g in build_groups(model))

    assert shape(Written()) == shape(Folded())
    w, f = Written().eval(), Folded().eval()
    f.load_state_dict(w.state_dict())
    img = torch.randn(1, 3, 6, 6)
    assert torch.equal(w(img), f(img))


@pytest.mark.parametrize("arch,enc", [("DeepLabV3Plus", "resnet50"), ("Unet", "timm-efficientnet-b3"),
                                      ("DeepLabV3Plus", "timm-efficientnet-b3"), ("Unet", "resnet50"),
                                      ("FPN", "resnet50"), ("FPN", "timm-efficientnet-b3")])
@pytest.mark.parametrize("classes", [21, 5])
def test_graph_analysis_equals_hand_declared_groups(arch, enc, classes):
    """engine.depgraph.build_groups against oracle/groups_ref.py, the coupled-channel groups of the six
    model names written down from their architectures: same roots, sizes, frozen flags, members,
    member kinds and positions inside concatenations.  An over-merged or missing coupling shows up here."""
    from oracle import groups_ref
    from spice_amd import nets
    model = nets.create_model(arch, enc, classes)
    got = groups_ref.canonical(build_groups(model))
    want = groups_ref.canonical(groups_ref.instantiate(model, groups_ref.TABLES[(arch, enc)](classes)))
    assert sorted(got) == sorted(want)
    for root in want:
        assert got[root][:2] == want[root][:2], root
        assert got[root][2] == want[root][2], (root, sorted(got[root][2] ^ want[root][2]))
    # module order of the roots is the iteration order of a pruning step
    assert [g.root_name for g in build_groups(model)] == [g.root_name for g in groups_ref.instantiate(
        model, groups_ref.TABLES[(arch, enc)](classes))]
    # offsets inside concatenations follow the sizes of the parts before them
    inst = {g.root_name: g for g in groups_ref.instantiate(model, groups_ref.TABLES[(arch, enc)](classes))}
    if arch == "DeepLabV3Plus":
        m = [x for x in inst["decoder.block1.0"].members if x.name == "decoder.block2.0.1"][0]
        assert m.offset() == 256
        inst["decoder.aspp.1.1"].size = 100
        assert m.offset() == 100
