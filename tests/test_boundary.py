"""The drop-in boundary: the reference's package names resolve here unedited, and the torch C++
extension (torch.ops.spice.*) is what the Python surfaces call."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import confusion as oc
from spice_amd import _native as nat
from spice_amd import ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

OPS = ["importance_run", "importance_cols_run", "gather_run", "confusion_update", "confusion_update_static",
       "argmax_confusion_update", "kdlite_forward", "kdlite_label_pass", "gather_upload", "depthwise_conv", "squeeze_excite", "pooled_branch", "scale_inplace", "upsample_bilinear", "bn_act", "bn_act_dual",
       "bn_act_maxpool", "depthwise_conv3x3", "conv1x1_f16", "conv1x1_bn_act", "conv3x3_bn_act", "conv7x7_s2"]


def test_reference_package_names_resolve_unedited():
    """pruning/__init__.py:18-31, baseline/eval/metrics.py:8, pruning/eval/metrics.py:11: same names, and
    each `pruning.X` IS the module spice_amd.pruning.X (one module object under two names)."""
    import baseline.eval.metrics
    import pruning
    import pruning.losses
    import spice_amd.baseline.eval.metrics
    import spice_amd.pruning.losses
    from baseline.eval.metrics import compute_metrics_from_confusion, update_confusion_matrix  # noqa: F401
    from pruning import (AVAILABLE_STRATEGIES, MODEL_CONFIGS, GroupSyncTaylorImportance, ModelFactory,  # noqa: F401
                         PruningConfig, PruningStrategyFactory, PruningTrainer, StructuredPruner,
                         TaylorWeightImportance, create_taylor_importance, prune_model, set_random_seeds)
    from pruning.eval.metrics import SegmentationMetrics, compute_metrics  # noqa: F401
    from pruning.losses import KDLiteLoss, PixelKDLoss, SegmentationLoss  # noqa: F401
    from pruning.sensitivity_utils import compute_taylor_scores  # noqa: F401
    assert pruning.losses is spice_amd.pruning.losses
    assert baseline.eval.metrics is spice_amd.baseline.eval.metrics
    assert len(pruning.__all__) == 12
    with pytest.raises(ImportError):
        import pruning.no_such_module  # noqa: F401


def test_cli_runs_under_its_reference_name():
    """`python -m pruning.cli` (pruning/cli.py:23): the parser answers, --ratio is validated (exit code != 0)."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-m", "pruning.cli", "--help"], cwd=ROOT, env=env, capture_output=True,
                         text=True, timeout=300)
    assert out.returncode == 0 and "--strategy" in out.stdout and "--pruning-ratio" in out.stdout
    bad = subprocess.run([sys.executable, "-m", "pruning.cli", "--model", "FPN_resnet", "--strategy", "magnitude",
                          "--ratio", "1.5"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert bad.returncode != 0
    if not torch.cuda.is_available():   # the un-pruned evaluation script takes no flags: it starts, then asks for a device
        out = subprocess.run([sys.executable, "-m", "baseline.eval.run_eval"], cwd=ROOT, env=env,
                             capture_output=True, text=True, timeout=300)
        assert "No module named" not in out.stderr and "HIP device" in out.stderr


def test_torch_extension_registers_every_operator():
    spice = nat.load_torch_ops()
    assert int(spice.abi_version()) == nat.ABI_VERSION
    for name in OPS:
        schema = getattr(spice, name).default._schema
        assert "Tensor" in str(schema), name
    # device errors come from TORCH_CHECK inside the extension, as exceptions
    conf = torch.zeros(3, 3, dtype=torch.int64)
    x = torch.zeros(10, dtype=torch.int64)
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        spice.confusion_update(conf, x, x, 3, torch.zeros(16, dtype=torch.uint8))
    with pytest.raises(RuntimeError, match="no CPU implementation"):
        spice.bn_act(torch.zeros(1, 2, 2, 2), *(torch.zeros(2),) * 4, 1e-5, None, True)


@pytest.mark.gpu
def test_torch_ops_spice_confusion_update_direct():
    """torch.ops.spice.confusion_update called as a maintainer would from update_confusion_matrix
    (baseline/eval/metrics.py:8-19): bit-exact against the oracle, on the current stream of a side stream too."""
    dev = torch.device("cuda:0")
    spice = nat.load_torch_ops()
    c = 21
    g = torch.Generator().manual_seed(11)
    t = torch.randint(0, c, (2, 96, 130), generator=g)
    p = torch.randint(0, c, (2, 96, 130), generator=g)
    t[torch.rand(2, 96, 130, generator=g) < 0.1] = 255
    want = np.zeros((c, c), dtype=np.int64)
    oc.update_confusion(want, p.numpy(), t.numpy(), c)
    ws = torch.zeros(nat.lib().spice_confusion_workspace_bytes(t.numel(), c), dtype=torch.uint8, device=dev)
    conf = torch.zeros(c, c, dtype=torch.int64, device=dev)
    pd, td = p.to(dev).reshape(-1), t.to(dev).reshape(-1)
    spice.confusion_update(conf, pd, td, c, ws)
    np.testing.assert_array_equal(conf.cpu().numpy(), want)
    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        spice.confusion_update(conf, pd, td, c, ws)
        spice.confusion_update_static(conf, pd, td, c, None)
    side.synchronize()
    np.testing.assert_array_equal(conf.cpu().numpy(), 3 * want)
    # TORCH_CHECK failures: dtype -> TypeError, shape -> ValueError, nothing enqueued
    with pytest.raises(TypeError):
        spice.confusion_update(conf, pd.int(), td, c, ws)
    with pytest.raises(ValueError):
        spice.confusion_update(conf, pd[:-1], td, c, ws)
    with pytest.raises(ValueError):
        spice.confusion_update(conf, pd[::2], td[::2], c, ws)
    with pytest.raises(RuntimeError):
        spice.confusion_update(conf, pd, td, c, torch.zeros(64, dtype=torch.uint8, device=dev))
    np.testing.assert_array_equal(conf.cpu().numpy(), 3 * want)
    # and the Python surface routes through it
    conf.zero_()
    ops.confusion_update(conf, pd, td, c)
    np.testing.assert_array_equal(conf.cpu().numpy(), want)


@pytest.mark.gpu
def test_cli_prunes_under_its_reference_name(tmp_path):
    """`python -m pruning.cli --model ... --strategy ... --ratio ...` end to end on the GPU, by the reference's
    module name, writing the reference's run directory (pruning/pruner.py:425-475)."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, "-m", "pruning.cli", "--model", "FPN_resnet", "--strategy", "magnitude",
                          "--ratio", "0.3", "--steps", "1", "--fine-tune-epochs", "0", "--final-fine-tune-epochs", "0",
                          "--output-dir", str(tmp_path), "--data-root", str(tmp_path / "nodata"), "--seed", "3",
                          "--device", "cuda"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "PRUNING COMPLETE" in out.stdout
    run = tmp_path / "FPN_resnet_magnitude_0.30"
    assert (run / "final_model.pth").exists() and (run / "metrics.json").exists()
