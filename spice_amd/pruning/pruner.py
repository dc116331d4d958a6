"""Quota allocation and the end-to-end pruning run (pruning/pruner.py:26-812)."""
from __future__ import annotations

import copy
import datetime
import logging
from abc import ABC, abstractmethod
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from .. import engine
from .config import PruningConfig
from .dataset import PruningDataHandler
from .models import ModelFactory
from .pruning_core import PruningOrchestrator, compute_model_statistics
from .sensitivity import TaylorWeightImportance
from .tracking import WandBTracker
from .trainer import PruningMetrics, PruningTrainer
from .utils import calculate_model_size_mb, count_model_parameters, set_random_seeds

logger = logging.getLogger(__name__)


class LayerQuotaCalculator:
    """Taylor sensitivity -> per-layer pruning ratios (pruning/pruner.py:26-209).

    This arithmetic decides channel counts, so it follows the reference operation for operation:
    fp32 tensors for the weights, `.item()` to a Python float, `int()` truncation, and a budget that
    shrinks inside the allocation loop."""

    @staticmethod
    def compute_layer_quotas(model: nn.Module, taylor_importance: TaylorWeightImportance,
                             global_ratio: float, alpha: float = 0.7, min_channels: int = 8
                             ) -> Tuple[Dict[nn.Module, float], Dict[nn.Module, float]]:
        layers = LayerQuotaCalculator._collect_prunable_layers(
            model, taylor_importance.sensitivity_scores, min_channels)
        if not layers:
            logger.warning("No prunable layers found")
            return {}, {}
        weights = LayerQuotaCalculator._compute_pruning_weights(layers, alpha)
        return LayerQuotaCalculator._allocate_layer_quotas(layers, weights, global_ratio, min_channels)

    @staticmethod
    def _collect_prunable_layers(model: nn.Module, sensitivity_scores: Dict, min_channels: int) -> List[Dict]:
        found = []
        for m in model.modules():
            if isinstance(m, nn.Conv2d) and m in sensitivity_scores and m.out_channels > min_channels:
                found.append({"module": m, "sensitivity": sensitivity_scores[m].mean().item(),
                              "flops": LayerQuotaCalculator._estimate_flops_contribution(m),
                              "channels": m.out_channels})
        return found

    @staticmethod
    def _estimate_flops_contribution(m: nn.Conv2d) -> int:
        return m.out_channels * m.in_channels * m.kernel_size[0] * m.kernel_size[1]

    @staticmethod
    def _compute_pruning_weights(layers: List[Dict], alpha: float) -> torch.Tensor:
        s = torch.tensor([d["sensitivity"] for d in layers])
        s = s / (s.mean() + 1e-8)
        inv = 1.0 / (s + 1e-6)
        if alpha > 0:
            f = torch.tensor([d["flops"] for d in layers], dtype=torch.float32)
            w = inv * ((f / f.max()) ** alpha)
        else:
            w = inv
        return w / w.sum()

    @staticmethod
    def _allocate_layer_quotas(layers: List[Dict], weights: torch.Tensor, global_ratio: float,
                               min_channels: int) -> Tuple[Dict[nn.Module, float], Dict[nn.Module, float]]:
        total = sum(d["channels"] for d in layers)
        remaining = int(total * global_ratio)
        taken = [0] * len(layers)
        for _ in range(10):
            if remaining <= 0:
                break
            live = [i for i, d in enumerate(layers) if taken[i] < d["channels"] - min_channels]
            if not live:
                break
            share = [weights[i].item() for i in live]
            tot = sum(share)
            share = [1.0 / len(share)] * len(share) if tot == 0 else [x / tot for x in share]
            for i, sh in zip(live, share):
                room = layers[i]["channels"] - min_channels - taken[i]
                give = min(int(remaining * sh), room)
                taken[i] += give
                remaining -= give
        quotas, sens = {}, {}
        for i, d in enumerate(layers):
            quotas[d["module"]] = taken[i] / d["channels"] if d["channels"] > 0 else 0
            sens[d["module"]] = d["sensitivity"]
        allocated = sum(taken)
        logger.info(f"Quota allocation: target={global_ratio:.2%}, "
                    f"allocated={(allocated / total if total else 0):.2%}, pruned={allocated}/{total} channels")
        return quotas, sens


class HybridTaylorMagnitudePruner:
    """Taylor sensitivity sets each layer's quota, magnitude picks the channels
    (pruning/pruner.py:212-306)."""

    def __init__(self, config: PruningConfig):
        self.config = config
        self.layer_sensitivities: Dict[nn.Module, float] = {}
        self.layer_quotas: Dict[nn.Module, float] = {}

    def compute_layer_quotas(self, model, taylor_importance, global_ratio, alpha=0.7, min_channels=8):
        self.layer_quotas, self.layer_sensitivities = LayerQuotaCalculator.compute_layer_quotas(
            model, taylor_importance, global_ratio, alpha, min_channels)
        return self.layer_quotas

    def create_hybrid_pruner(self, model, taylor_importance, example_inputs, ignored_layers):
        quotas = self.compute_layer_quotas(
            model=model, taylor_importance=taylor_importance, global_ratio=self.config.pruning_ratio,
            alpha=getattr(self.config, "sensitivity_alpha", 0.7),
            min_channels=getattr(self.config, "min_out_channels", 8))
        pruner = engine.MagnitudePruner(
            model=model, example_inputs=example_inputs,
            importance=engine.MagnitudeImportance(p=self.config.importance_norm),
            pruning_ratio=self.config.pruning_ratio, pruning_ratio_dict=quotas,
            iterative_steps=self.config.iterative_steps, ignored_layers=ignored_layers,
            global_pruning=False)
        logger.info(f"Created hybrid pruner: Taylor sensitivity for {len(quotas)} layer quotas, "
                    "Magnitude for channel selection")
        return pruner

    def get_pruning_summary(self) -> Dict:
        rows = [{"layer_type": m.__class__.__name__, "out_channels": m.out_channels,
                 "sensitivity": float(self.layer_sensitivities.get(m, 0)), "pruning_quota": float(q),
                 "channels_to_prune": int(m.out_channels * q)} for m, q in self.layer_quotas.items()]
        rows.sort(key=lambda r: r["sensitivity"], reverse=True)
        return {"num_layers": len(self.layer_quotas), "layer_details": rows}


def create_hybrid_pruner(model, config, taylor_importance, example_inputs, ignored_layers):
    hybrid = HybridTaylorMagnitudePruner(config)
    pruner = hybrid.create_hybrid_pruner(model=model, taylor_importance=taylor_importance,
                                         example_inputs=example_inputs, ignored_layers=ignored_layers)
    return pruner, hybrid.get_pruning_summary()


class TrainingStrategy(ABC):
    @abstractmethod
    def train_model(self, trainer, model, loaders, epochs, teacher=None) -> Dict:
        ...


class StandardTrainingStrategy(TrainingStrategy):
    def train_model(self, trainer, model, loaders, epochs, teacher=None):
        return trainer.fine_tune(model, loaders[0], loaders[1], epochs, teacher=None)


class KDTrainingStrategy(TrainingStrategy):
    def train_model(self, trainer, model, loaders, epochs, teacher=None):
        return trainer.fine_tune(model, loaders[0], loaders[1], epochs, teacher=teacher)


class ModelManager:
    """Model loading and the output directory layout
    {model}_{strategy}_{ratio:.2f}[_kd|_kd_refine] (pruning/pruner.py:415-475)."""

    def __init__(self, config: PruningConfig):
        self.config = config

    def run_directory(self) -> Path:
        c = self.config
        name = f"{c.model_name}_{c.pruning_strategy}_{c.pruning_ratio:.2f}"
        if c.enable_kd_lite:
            name += "_kd_refine" if c.kd_mode == "refine" else "_kd"
        if c.use_timestamp:
            name = datetime.datetime.now().strftime("%Y%m%d_%H%M%S_") + name
        d = Path(c.output_dir) / name
        (d / "intermediate").mkdir(parents=True, exist_ok=True)
        return d

    def load(self) -> nn.Module:
        return ModelFactory.load_pretrained_model(self.config)

    def save_intermediate_model(self, model, run_dir: Path, step: int, total: int) -> Path:
        path = run_dir / "intermediate" / f"{self.config.model_name}_step_{step}_of_{total}.pth"
        torch.save(model, path)
        return path


class PruningStepExecutor:
    def __init__(self, config, trainer, model_manager, training_strategy, tracker):
        self.config, self.trainer, self.models = config, trainer, model_manager
        self.training_strategy, self.tracker = training_strategy, tracker

    def execute_pruning_step(self, pruner, model, example_inputs, loaders, step, total, run_dir, teacher):
        pruner.step()
        stats = compute_model_statistics(model, example_inputs)
        logger.info(f"Step {step}/{total}: {stats['params_million']:.2f}M params, {stats['macs_million']:.1f}M MACs")
        train = self.training_strategy.train_model(self.trainer, model, loaders, self.config.fine_tune_epochs,
                                                   teacher=teacher)
        self.models.save_intermediate_model(model, run_dir, step, total)
        return {**stats, **train}

    def execute_final_training(self, model, loaders, teacher):
        return self.training_strategy.train_model(self.trainer, model, loaders,
                                                  self.config.final_fine_tune_epochs, teacher=teacher)


class StructuredPruner:
    def __init__(self, config: PruningConfig):
        self.config = config
        set_random_seeds(config.seed)
        self.tracker = WandBTracker(config, enabled=config.use_wandb)
        self.model_manager = ModelManager(config)
        self.trainer = PruningTrainer(config, self.tracker)
        self.orchestrator = PruningOrchestrator(config)
        self.orchestrator.set_wandb_tracker(self.tracker)
        self.metrics = PruningMetrics()

    def prune(self) -> Dict[str, Any]:
        try:
            return self._run()
        except Exception as e:
            logger.error(f"Pruning failed: {e}")
            raise
        finally:
            self.trainer.cleanup()

    def _run(self) -> Dict[str, Any]:
        c = self.config
        if torch.device(c.device).type != "cuda":
            raise RuntimeError(f"device '{c.device}': this implementation runs on HIP devices only")
        model = self.model_manager.load()
        teacher = copy.deepcopy(model).eval() if c.enable_kd_lite else None
        torch.manual_seed(c.seed)
        example_inputs = torch.randn(*c.example_input_size).to(c.device)
        initial = compute_model_statistics(model, example_inputs)
        initial_size = calculate_model_size_mb(model)
        ignored = ModelFactory.get_ignored_layers(model)
        pruner, info = self.orchestrator.setup_importance_and_pruner(model, example_inputs, ignored)
        run_dir = self.model_manager.run_directory()
        self.orchestrator.save_reports(run_dir, info)
        handler = PruningDataHandler(c.data_root, c.batch_size, c.num_workers, c.seed, c.num_classes, c.input_size)
        loaders = handler.get_loaders()
        strategy = KDTrainingStrategy() if c.enable_kd_lite else StandardTrainingStrategy()
        executor = PruningStepExecutor(c, self.trainer, self.model_manager, strategy, self.tracker)
        for step in range(1, c.iterative_steps + 1):
            row = executor.execute_pruning_step(pruner, model, example_inputs, loaders, step,
                                                c.iterative_steps, run_dir, teacher)
            self.metrics.add_step(step, **row)
        final_train = executor.execute_final_training(model, loaders, teacher)
        final = compute_model_statistics(model, example_inputs)
        out_path = run_dir / "final_model.pth"
        ModelFactory.save_model(model, out_path, {"strategy": c.pruning_strategy, "ratio": c.pruning_ratio})
        self.metrics.save(run_dir / "metrics.json")
        final_size = calculate_model_size_mb(model)
        return {
            "model_name": c.model_name, "pruning_strategy": c.pruning_strategy,
            "pruning_ratio": c.pruning_ratio, "kd_lite_enabled": c.enable_kd_lite,
            "kd_temperature": c.kd_temperature, "kd_alpha": c.kd_alpha,
            "initial_params": initial["params"], "final_params": final["params"],
            "params_reduction": 1 - final["params"] / max(initial["params"], 1),
            "initial_macs": initial["macs"], "final_macs": final["macs"],
            "initial_size_mb": initial_size, "final_size_mb": final_size,
            "size_reduction": 1 - final_size / max(initial_size, 1e-12),
            "output_path": str(out_path), "final_training": final_train,
            "steps": self.metrics.history,
        }


def prune_model(config: PruningConfig) -> Dict[str, Any]:
    return StructuredPruner(config).prune()
