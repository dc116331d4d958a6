"""Generate tests/golden/formats.json from the reference: on-disk formats and host-side rules
around the hot path (run directory names and their parser, intermediate checkpoint names,
model_info.txt, metrics.json rows, the results dictionary, learning-rate schedules, the KD data
subset, the evaluation loop's result dictionary).

Run in the build container only (the reference does not travel):
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_formats.py

Imports follow make_golden.py: reference modules by file path or as the `pruning` package with
empty placeholder modules for absent third-party packages; every stored value is computed by
in-tree reference code.
"""
from __future__ import annotations

import importlib
import json
import os
import sys
import tempfile
import types
from pathlib import Path

sys.dont_write_bytecode = True

import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, install_placeholders, load_by_path  # noqa: E402

RUN_DIRS = [
    "DeepLabV3Plus_resnet_magnitude_0.50", "DeepLabV3Plus_resnet_magnitude_taylor_0.50_kd_refine",
    "UNET_efficientnet_taylor_weight_0.30_kd", "FPN_resnet_lamp_0.70", "UNET_resnet_random_0.25_kd_refine",
    "20250101_120000_DeepLabV3Plus_efficientnet_taylor_0.40", "20240229_235959_FPN_efficientnet_magnitude_taylor_0.90_kd",
    "DeepLabV3Plus_resnet_magnitude_abc", "short_0.5", "x_y_0.5", "magnitude_taylor_0.50", "a_taylor_weight_0.10",
    "UNET_resnet_magnitude_1", "DeepLabV3Plus_resnet_magnitude_0.50_refine", "nofinal_resnet_magnitude_0.50",
]
STEP_FILES = ["DeepLabV3Plus_resnet_step_1_of_5.pth", "DeepLabV3Plus_resnet_step_2_of_5.pth",
              "UNET_resnet_step_10_of_12.pth", "notes.txt", "weird_step_.pth", "a_step_1_of_2_step_3_of_4.pth",
              "final.pth"]


def loader_cases(loader_mod):
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        root = Path(tmp)
        for d in RUN_DIRS:
            (root / d).mkdir()
            if not d.startswith("nofinal"):
                (root / d / "final_model.pth").write_bytes(b"")
        (root / "a_file_not_a_dir_magnitude_0.5").write_bytes(b"")
        rows = loader_mod.PrunedModelLoader.find_pruned_models(str(root))
        for r in rows:
            r["path"] = os.path.relpath(r["path"], tmp)
        out["run_dirs"] = RUN_DIRS
        out["found"] = rows
        out["missing_root"] = loader_mod.PrunedModelLoader.find_pruned_models(str(root / "absent"))
        run = root / RUN_DIRS[0]
        (run / "intermediate").mkdir()
        for f in STEP_FILES:
            (run / "intermediate" / f).write_bytes(b"")
        rows = loader_mod.PrunedModelLoader.load_intermediate_models(str(run))
        for r in rows:
            r["path"] = os.path.relpath(r["path"], str(run))
        out["step_files"] = STEP_FILES
        out["intermediate"] = rows
        out["intermediate_missing"] = loader_mod.PrunedModelLoader.load_intermediate_models(str(root / RUN_DIRS[1]))
    return out


def schedule_cases(sched_mod):
    cases = [
        dict(name="cosine_warm2_e15", use_cosine_annealing=True, lr_warmup_epochs=2, lr_min_factor=0.01, epochs=15),
        dict(name="cosine_warm2_e8", use_cosine_annealing=True, lr_warmup_epochs=2, lr_min_factor=0.01, epochs=8),
        dict(name="cosine_warm2_e5", use_cosine_annealing=True, lr_warmup_epochs=2, lr_min_factor=0.01, epochs=5),
        dict(name="cosine_warm2_e2", use_cosine_annealing=True, lr_warmup_epochs=2, lr_min_factor=0.01, epochs=2),
        dict(name="cosine_warm0_e6", use_cosine_annealing=True, lr_warmup_epochs=0, lr_min_factor=0.1, epochs=6),
        dict(name="cosine_warm5_e30", use_cosine_annealing=True, lr_warmup_epochs=5, lr_min_factor=0.01, epochs=30),
        dict(name="plateau_e12", use_cosine_annealing=False, lr_warmup_epochs=2, lr_min_factor=0.01, epochs=12,
             val=[1.0, 0.9, 0.95, 0.96, 0.97, 0.98, 0.99, 1.0, 1.1, 1.2, 1.3, 1.4]),
    ]
    out = []
    for c in cases:
        cfg = types.SimpleNamespace(fine_tune_lr=1e-4, **{k: c[k] for k in
                                                          ("use_cosine_annealing", "lr_warmup_epochs", "lr_min_factor")})
        p = nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([p], lr=cfg.fine_tune_lr)
        s = sched_mod.create_scheduler(cfg, opt, c["epochs"])
        lrs = [opt.param_groups[0]["lr"]]
        for e in range(c["epochs"]):
            opt.step()
            if "val" in c:
                s.step(c["val"][e])
            else:
                s.step()
            lrs.append(opt.param_groups[0]["lr"])
        out.append({**c, "lrs": lrs, "scheduler": type(s).__name__,
                    "warmup_epochs": getattr(s, "warmup_epochs", None)})
    return out


def tiny_module():
    torch.manual_seed(0)
    m = nn.Sequential(nn.Conv2d(3, 8, 3, bias=False), nn.BatchNorm2d(8), nn.Conv2d(8, 5, 1))
    for p in m[2].parameters():
        p.requires_grad = False
    return m


def pruner_cases(pruner_mod, config_mod, models_mod, trainer_mod, utils_mod):
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        names = []
        for model, strat, ratio, kd, mode in [
            ("DeepLabV3Plus_resnet", "magnitude", 0.5, False, "refine"),
            ("UNET_efficientnet", "magnitude_taylor", 0.3, True, "refine"),
            ("FPN_resnet", "taylor_weight", 0.25, True, "replace"),
            ("UNET_resnet", "lamp", 1 / 3, False, "replace"),
        ]:
            cfg = config_mod.PruningConfig(model_name=model, output_dir=tmp, pruning_strategy=strat,
                                           pruning_ratio=ratio, enable_kd_lite=kd, kd_mode=mode, device="cpu")
            mm = pruner_mod.ModelManager(cfg)
            names.append({"model": model, "strategy": strat, "ratio": ratio, "kd": kd, "kd_mode": mode,
                          "dir": os.path.relpath(str(mm.pruning_dir), tmp),
                          "has_intermediate": mm.intermediate_dir.is_dir(),
                          "output_path": os.path.relpath(cfg.output_path, tmp)})
        out["run_dir_names"] = names

        cfg = config_mod.PruningConfig(model_name="DeepLabV3Plus_resnet", output_dir=tmp, pruning_strategy="magnitude_taylor",
                                       pruning_ratio=0.5, enable_kd_lite=True, device="cpu")
        m = tiny_module()
        models_mod.ModelFactory.save_model(m, os.path.join(tmp, "sub_final_model.pth"), cfg)
        out["model_info_txt"] = Path(tmp, "model_info.txt").read_text()
        out["model_info"] = models_mod.ModelFactory.get_model_info(m)
        out["size_mb"] = utils_mod.calculate_model_size_mb(m)
        out["params_trainable"] = utils_mod.count_model_parameters(m, only_trainable=True)

        pm = trainer_mod.PruningMetrics()
        out["final_metrics_empty"] = pm.get_final_metrics()
        pm.add_step(step=1, train_loss=0.0, val_loss=0.0, params=26_000_000, macs=36_000_000_000)
        out["final_metrics_one"] = pm.get_final_metrics()
        pm.add_step(step=2, train_loss=0.25, val_loss=0.5, params=13_500_000, macs=17_000_000_000)
        pm.save_history(os.path.join(tmp, "metrics.json"))
        out["metrics_json_text"] = Path(tmp, "metrics.json").read_text()
        out["final_metrics_two"] = pm.get_final_metrics()

        fake = types.SimpleNamespace(config=cfg, model_manager=types.SimpleNamespace(pruning_dir=Path("RUN")))
        init = {"total_params": 1000, "trainable_params": 1000, "model_size_mb": 4.0, "macs": 5000, "macs_million": 0.005}
        fin = {"total_params": 400, "trainable_params": 400, "model_size_mb": 1.6, "macs": 1500, "macs_million": 0.0015}
        out["results_kd"] = pruner_mod.StructuredPruner._build_results_dict(
            fake, init, fin, Path("RUN/final_model.pth"), Path("RUN/metrics.json"))
        cfg2 = config_mod.PruningConfig(model_name="UNET_resnet", output_dir=tmp, device="cpu")
        fake2 = types.SimpleNamespace(config=cfg2, model_manager=types.SimpleNamespace(pruning_dir=Path("RUN")))
        init0 = dict(init, macs=0)
        out["results_plain_no_macs"] = pruner_mod.StructuredPruner._build_results_dict(
            fake2, init0, fin, Path("RUN/final_model.pth"), Path("RUN/metrics.json"))

        # strategies chosen by the factory
        picks = []
        for kd, mode in [(False, "refine"), (True, "refine"), (True, "replace")]:
            c = config_mod.PruningConfig(model_name="UNET_resnet", output_dir=tmp, enable_kd_lite=kd, kd_mode=mode,
                                         device="cpu")
            picks.append({"kd": kd, "kd_mode": mode,
                          "strategy": type(pruner_mod.TrainingStrategyFactory.create_strategy(c)).__name__})
        out["training_strategies"] = picks

        # KD data subset (pruning/trainer.py:69-94)
        subsets = []
        for n, ratio, seed in [(100, 0.3, 42), (32, 0.5, 42), (57, 0.1, 7), (10, 0.15, 1)]:
            tr = trainer_mod.PruningTrainer.__new__(trainer_mod.PruningTrainer)
            tr.config = types.SimpleNamespace(kd_data_ratio=ratio, seed=seed, batch_size=4, num_workers=0)
            ds = torch.utils.data.TensorDataset(torch.arange(n))
            import warnings
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                ld = tr._create_kd_subset_loader(torch.utils.data.DataLoader(ds, batch_size=4))
            subsets.append({"n": n, "ratio": ratio, "seed": seed, "indices": list(ld.dataset.indices)})
        out["kd_subsets"] = subsets
    out["config_hash"] = {s: utils_mod.generate_config_hash(s) for s in
                          ("DeepLabV3Plus_resnet_magnitude_0.5_42", "UNET_efficientnet_magnitude_taylor_0.3_7", "")}
    return out


class ExactNet(nn.Module):
    """1x1 convolution with small integer weights on integer-valued images: every logit is an
    exactly representable small integer in fp32 and fp16, so the arg-max (ties included) is the
    same on any device and under autocast."""

    def __init__(self, classes):
        super().__init__()
        self.conv = nn.Conv2d(3, classes, 1, bias=True)
        g = torch.Generator().manual_seed(11)
        with torch.no_grad():
            self.conv.weight.copy_(torch.randint(-2, 3, self.conv.weight.shape, generator=g).float())
            self.conv.bias.copy_(torch.randint(-1, 2, self.conv.bias.shape, generator=g).float())

    def forward(self, x):
        return self.conv(x)


def inference_case(inference_mod):
    classes = 5
    g = torch.Generator().manual_seed(12)
    batches = []
    for b in range(3):
        x = torch.randint(-3, 4, (2, 3, 16, 16), generator=g).float()
        y = torch.randint(0, classes, (2, 16, 16), generator=g)
        y[torch.rand(2, 16, 16, generator=g) < 0.1] = 255
        batches.append((x, y))
    net = ExactNet(classes)
    res_tuple = inference_mod.inference(net, batches, torch.device("cpu"), classes)
    res_dict = inference_mod.inference(net, [{"image": x, "mask": y} for x, y in batches], torch.device("cpu"), classes)
    assert res_tuple == res_dict
    return {"classes": classes, "images": [x.tolist() for x, _ in batches], "masks": [y.tolist() for _, y in batches],
            "weight": net.conv.weight.tolist(), "bias": net.conv.bias.tolist(), "result": res_tuple}


def eval_result_rows():
    """Evaluation results as CoreEvaluator returns them (lists, numpy arrays, numpy scalars)."""
    import numpy as np
    rows = []
    for k, (model, strat, method, ratio) in enumerate([("UNET_resnet", "magnitude", "Standard", 0.5),
                                                       ("DeepLabV3Plus_resnet", "magnitude_taylor", "Standard+KD", 0.7),
                                                       ("DeepLabV3Plus_resnet", "taylor_weight", "KD-Lite", 0.3)]):
        conf = (np.arange(25, dtype=np.int64).reshape(5, 5) * (k + 1)) % 17
        rows.append({"miou": 0.5 + 0.1 * k, "mean_acc": float(np.float32(0.61 + 0.05 * k)),
                     "per_class_iou": [0.1 * (i + k) for i in range(5)],
                     "per_class_acc": np.linspace(0.2, 0.9, 5).astype(np.float32).tolist(),
                     "confusion_matrix": conf, "path": f"runs/{model}/final_model.pth", "model_name": model,
                     "strategy": strat, "pruning_ratio": ratio, "dir_name": f"{model}_{strat}_{ratio:.2f}",
                     "fine_tune_method": method, "total_params": 26_000_000 - 7_000_000 * k,
                     "model_size_mb": 99.18 - 26.7 * k, "total_macs": 36_000_000_000 - 9_000_000_000 * k,
                     "gmacs": 36.0 - 9.0 * k, "params_percentage": 100.0 - 26.9 * k, "size_percentage": 100.0 - 26.9 * k,
                     "mac_percentage": 100.0 - 25.0 * k, "mac_reduction": 25.0 * k})
    return rows


def result_processor_case(rp_mod):
    import contextlib
    import io
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        proc = rp_mod.ResultProcessor(tmp)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            for row in eval_result_rows():
                proc.process_single_result(row, row["dir_name"])
            csv_path = proc.generate_summary(save_csv=True)
        out["stdout"] = buf.getvalue()
        # the inputs, for the test to rebuild (confusion_matrix goes back to an int64 array there)
        out["rows"] = [dict(r, confusion_matrix=r["confusion_matrix"].tolist()) for r in eval_result_rows()]
        out["row_keys"] = list(eval_result_rows()[0])  # the file is written with sorted keys; columns follow this order
        out["csv_name"] = os.path.relpath(csv_path, tmp)
        out["csv_text"] = Path(csv_path).read_text()
        out["metrics_json"] = {r["dir_name"]: Path(tmp, r["dir_name"], "metrics.json").read_text() for r in eval_result_rows()}
        empty = rp_mod.ResultProcessor(os.path.join(tmp, "empty"))
        out["empty_summary"] = empty.generate_summary()
    return out


def evaluation_config_defaults():
    """Field defaults of the evaluation configuration (its constructor makes its output
    directories relative to the working directory, hence the temporary one)."""
    cfg_mod = load_by_path("ref_eval_config", "pruning/eval/config.py")
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            return dict(cfg_mod.EvaluationConfig().__dict__)
        finally:
            os.chdir(cwd)


def main():
    torch.set_num_threads(2)
    loader_mod = load_by_path("ref_eval_loader", "pruning/eval/loader.py")
    sched_mod = load_by_path("ref_lr_scheduler", "pruning/lr_scheduler.py")
    install_placeholders()
    if "tqdm" not in sys.modules:
        importlib.import_module("tqdm")
    sys.path.insert(0, REF)
    pruner_mod = importlib.import_module("pruning.pruner")
    config_mod = importlib.import_module("pruning.config")
    models_mod = importlib.import_module("pruning.models")
    trainer_mod = importlib.import_module("pruning.trainer")
    utils_mod = importlib.import_module("pruning.utils")
    inference_mod = importlib.import_module("baseline.eval.inference")
    rp_mod = load_by_path("ref_result_processor", "pruning/eval/components/result_processor.py")
    out = {"loader": loader_cases(loader_mod), "schedules": schedule_cases(sched_mod),
           "result_processor": result_processor_case(rp_mod),
           "evaluation_config_defaults": evaluation_config_defaults(),
           "pruner": pruner_cases(pruner_mod, config_mod, models_mod, trainer_mod, utils_mod),
           "inference": inference_case(inference_mod)}
    with open(os.path.join(HERE, "formats.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("formats.json written")


if __name__ == "__main__":
    main()
