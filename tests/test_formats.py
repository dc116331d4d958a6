"""On-disk formats and host-side rules around the hot path, against tests/golden/formats.json
(generated from the reference by tests/golden/make_golden_formats.py)."""
import json
import os
import types
import warnings
from pathlib import Path

import pytest
import torch
import torch.nn as nn

from spice_amd.pruning import PruningConfig
from spice_amd.pruning.eval.loader import PrunedModelLoader, parse_run_directory_name
from spice_amd.pruning.lr_scheduler import create_scheduler
from spice_amd.pruning.models import ModelFactory
from spice_amd.pruning.pruner import ModelManager, StructuredPruner, TrainingStrategyFactory
from spice_amd.pruning.trainer import PruningMetrics, PruningTrainer
from spice_amd.pruning.utils import calculate_model_size_mb, count_model_parameters, generate_config_hash

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(HERE, "golden", "formats.json")) as f:
        return json.load(f)


def test_find_pruned_models_matches_reference(gold, tmp_path):
    g = gold["loader"]
    for d in g["run_dirs"]:
        (tmp_path / d).mkdir()
        if not d.startswith("nofinal"):
            (tmp_path / d / "final_model.pth").write_bytes(b"")
    (tmp_path / "a_file_not_a_dir_magnitude_0.5").write_bytes(b"")
    rows = PrunedModelLoader.find_pruned_models(str(tmp_path))
    for r in rows:
        r["path"] = os.path.relpath(r["path"], str(tmp_path))
    assert rows == g["found"]
    assert PrunedModelLoader.find_pruned_models(str(tmp_path / "absent")) == g["missing_root"] == []


def test_directory_named_like_a_suffix_is_skipped():
    # the reference raises IndexError on a run directory named just "kd" (pruning/eval/loader.py:82-94);
    # this parser reports "not a run" instead
    assert parse_run_directory_name("kd") is None
    assert parse_run_directory_name("kd_refine") is None


def test_intermediate_listing_matches_reference(gold, tmp_path):
    g = gold["loader"]
    inter = tmp_path / "run" / "intermediate"
    inter.mkdir(parents=True)
    for f in g["step_files"]:
        (inter / f).write_bytes(b"")
    rows = PrunedModelLoader.load_intermediate_models(str(tmp_path / "run"))
    for r in rows:
        r["path"] = os.path.relpath(r["path"], str(tmp_path / "run"))
    assert rows == g["intermediate"]
    assert PrunedModelLoader.load_intermediate_models(str(tmp_path)) == g["intermediate_missing"] == []


def test_load_pruned_model_missing_file(tmp_path):
    with pytest.raises(FileNotFoundError):
        PrunedModelLoader.load_pruned_model(str(tmp_path / "none.pth"))


def test_lr_schedules_match_reference(gold):
    for c in gold["schedules"]:
        cfg = types.SimpleNamespace(fine_tune_lr=1e-4, use_cosine_annealing=c["use_cosine_annealing"],
                                    lr_warmup_epochs=c["lr_warmup_epochs"], lr_min_factor=c["lr_min_factor"])
        opt = torch.optim.Adam([nn.Parameter(torch.zeros(1))], lr=cfg.fine_tune_lr)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            s = create_scheduler(cfg, opt, c["epochs"])
            lrs = [opt.param_groups[0]["lr"]]
            for e in range(c["epochs"]):
                opt.step()
                s.step(c["val"][e]) if "val" in c else s.step()
                lrs.append(opt.param_groups[0]["lr"])
        assert type(s).__name__ == c["scheduler"], c["name"]
        assert getattr(s, "warmup_epochs", None) == c["warmup_epochs"], c["name"]
        assert lrs == pytest.approx(c["lrs"], rel=1e-12, abs=0), c["name"]


def test_run_directory_layout_matches_reference(gold, tmp_path):
    for row in gold["pruner"]["run_dir_names"]:
        cfg = PruningConfig(model_name=row["model"], output_dir=str(tmp_path), pruning_strategy=row["strategy"],
                            pruning_ratio=row["ratio"], enable_kd_lite=row["kd"], kd_mode=row["kd_mode"], device="cpu")
        mm = ModelManager(cfg)
        assert os.path.relpath(str(mm.pruning_dir), str(tmp_path)) == row["dir"]
        assert mm.intermediate_dir.is_dir() == row["has_intermediate"]
        assert os.path.relpath(cfg.output_path, str(tmp_path)) == row["output_path"]
        info = parse_run_directory_name(row["dir"])  # the evaluator's parser reads back what the pruner names
        assert info["model_name"] == row["model"] and info["strategy"] == row["strategy"]
    stamped = ModelManager(PruningConfig(model_name="FPN_resnet", output_dir=str(tmp_path), use_timestamp=True,
                                         device="cpu")).pruning_dir.name
    assert parse_run_directory_name(stamped)["model_name"] == "FPN_resnet"
    assert stamped.endswith("_FPN_resnet_magnitude_0.50") and stamped[:8].isdigit() and stamped[9:15].isdigit()


def _tiny_module():
    torch.manual_seed(0)
    m = nn.Sequential(nn.Conv2d(3, 8, 3, bias=False), nn.BatchNorm2d(8), nn.Conv2d(8, 5, 1))
    for p in m[2].parameters():
        p.requires_grad = False
    return m


def test_model_info_file_matches_reference(gold, tmp_path):
    g = gold["pruner"]
    cfg = PruningConfig(model_name="DeepLabV3Plus_resnet", output_dir=str(tmp_path), pruning_strategy="magnitude_taylor",
                        pruning_ratio=0.5, enable_kd_lite=True, device="cpu")
    m = _tiny_module()
    ModelFactory.save_model(m, tmp_path / "final_model.pth", cfg)
    assert (tmp_path / "model_info.txt").read_text() == g["model_info_txt"]
    assert ModelFactory.get_model_info(m) == g["model_info"]
    assert calculate_model_size_mb(m) == g["size_mb"]
    assert count_model_parameters(m, only_trainable=True) == g["params_trainable"]
    back = torch.load(tmp_path / "final_model.pth", weights_only=False)  # written three lines above
    assert isinstance(back, nn.Sequential) and torch.equal(back[0].weight, m[0].weight)


def test_metrics_history_matches_reference(gold, tmp_path):
    g = gold["pruner"]
    pm = PruningMetrics()
    assert pm.get_final_metrics() == g["final_metrics_empty"] == {}
    pm.add_step(step=1, train_loss=0.0, val_loss=0.0, params=26_000_000, macs=36_000_000_000)
    assert pm.get_final_metrics() == g["final_metrics_one"]
    pm.add_step(step=2, train_loss=0.25, val_loss=0.5, params=13_500_000, macs=17_000_000_000)
    pm.save_history(str(tmp_path / "metrics.json"))
    assert (tmp_path / "metrics.json").read_text() == g["metrics_json_text"]
    assert pm.get_final_metrics() == g["final_metrics_two"]


def test_results_dictionary_matches_reference(gold, tmp_path):
    g = gold["pruner"]
    init = {"total_params": 1000, "trainable_params": 1000, "model_size_mb": 4.0, "macs": 5000, "macs_million": 0.005}
    fin = {"total_params": 400, "trainable_params": 400, "model_size_mb": 1.6, "macs": 1500, "macs_million": 0.0015}
    cfg = PruningConfig(model_name="DeepLabV3Plus_resnet", output_dir=str(tmp_path), pruning_strategy="magnitude_taylor",
                        pruning_ratio=0.5, enable_kd_lite=True, device="cpu")
    fake = types.SimpleNamespace(config=cfg, model_manager=types.SimpleNamespace(pruning_dir=Path("RUN")))
    got = StructuredPruner._build_results_dict(fake, init, fin, Path("RUN/final_model.pth"), Path("RUN/metrics.json"))
    assert got == g["results_kd"] and list(got) != []
    cfg2 = PruningConfig(model_name="UNET_resnet", output_dir=str(tmp_path), device="cpu")
    fake2 = types.SimpleNamespace(config=cfg2, model_manager=types.SimpleNamespace(pruning_dir=Path("RUN")))
    got2 = StructuredPruner._build_results_dict(fake2, dict(init, macs=0), fin, Path("RUN/final_model.pth"),
                                                Path("RUN/metrics.json"))
    assert got2 == g["results_plain_no_macs"]


def test_training_strategy_choice_matches_reference(gold, tmp_path):
    for row in gold["pruner"]["training_strategies"]:
        cfg = PruningConfig(model_name="UNET_resnet", output_dir=str(tmp_path), enable_kd_lite=row["kd"],
                            kd_mode=row["kd_mode"], device="cpu")
        assert type(TrainingStrategyFactory.create_strategy(cfg)).__name__ == row["strategy"]


def test_kd_subset_matches_reference(gold):
    for row in gold["pruner"]["kd_subsets"]:
        tr = PruningTrainer.__new__(PruningTrainer)
        tr.config = types.SimpleNamespace(kd_data_ratio=row["ratio"], seed=row["seed"], batch_size=4, num_workers=0)
        ds = torch.utils.data.TensorDataset(torch.arange(row["n"]))
        ld = tr._create_kd_subset_loader(torch.utils.data.DataLoader(ds, batch_size=4))
        assert list(ld.dataset.indices) == row["indices"]
    tr.config = types.SimpleNamespace(kd_data_ratio=0.05, seed=1, batch_size=4, num_workers=0)
    with pytest.raises(ValueError):  # an empty subset is refused, as by the reference's DataLoader
        tr._create_kd_subset_loader(torch.utils.data.DataLoader(torch.utils.data.TensorDataset(torch.arange(10))))


def test_config_hash_matches_reference(gold):
    for s, h in gold["pruner"]["config_hash"].items():
        assert generate_config_hash(s) == h


def _eval_rows(gold):
    import numpy as np
    g = gold["result_processor"]
    rows = [{k: r[k] for k in g["row_keys"]} for r in g["rows"]]
    return [dict(r, confusion_matrix=np.asarray(r["confusion_matrix"], dtype=np.int64)) for r in rows]


def test_result_processor_matches_reference(gold, tmp_path, capsys):
    """Console blocks, per-model metrics.json and evaluation_summary.csv, byte for byte
    (pruning/eval/components/result_processor.py:12-185)."""
    from spice_amd.pruning.eval.components import ResultProcessor
    g = gold["result_processor"]
    proc = ResultProcessor(str(tmp_path))
    capsys.readouterr()
    for row in _eval_rows(gold):
        proc.process_single_result(row, row["dir_name"])
    csv_path = proc.generate_summary(save_csv=True)
    assert capsys.readouterr().out == g["stdout"]
    assert os.path.relpath(csv_path, tmp_path) == g["csv_name"]
    assert Path(csv_path).read_text() == g["csv_text"]
    for name, text in g["metrics_json"].items():
        assert (tmp_path / name / "metrics.json").read_text() == text
    assert ResultProcessor(str(tmp_path / "empty")).generate_summary() == g["empty_summary"]
    assert len(proc.get_results_dataframe()) == len(g["rows"])


def test_evaluation_cache_round_trip(gold, tmp_path):
    """A stored result comes back for the same run directory while its model file is unchanged,
    and not after the file changes or for another configuration of the same path."""
    import numpy as np
    from spice_amd.pruning.eval.components import CacheManager
    rows = _eval_rows(gold)
    model = tmp_path / "final_model.pth"
    model.write_bytes(b"abc")
    info = {"path": str(model), "model_name": "UNET_resnet", "strategy": "magnitude", "pruning_ratio": 0.5,
            "dir_name": "UNET_resnet_magnitude_0.50", "fine_tune_method": "Standard"}
    cm = CacheManager(cache_dir=str(tmp_path), enabled=True)
    cache = cm.load_cache()
    assert cache == {} and cm.get_cached_result(info, cache) is None
    cm.store_result(info, rows[0], cache)
    again = CacheManager(cache_dir=str(tmp_path), enabled=True)
    hit = again.get_cached_result(info, again.load_cache())
    assert hit is not None and hit["miou"] == rows[0]["miou"]
    assert np.array_equal(np.asarray(hit["confusion_matrix"]), rows[0]["confusion_matrix"])
    assert again.get_cached_result(dict(info, pruning_ratio=0.6), again.load_cache()) is None
    os.utime(model, (1, 1))
    assert again.get_cached_result(info, again.load_cache()) is None
    off = CacheManager(cache_dir=str(tmp_path), enabled=False)
    assert off.load_cache() == {} and off.get_cached_result(info, {"x": 1}) is None


def test_evaluation_config_defaults_match_reference(gold):
    from spice_amd.pruning.eval import EvaluationConfig
    assert EvaluationConfig().__dict__ == gold["evaluation_config_defaults"]
