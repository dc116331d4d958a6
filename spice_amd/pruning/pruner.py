"""Quota allocation and the end-to-end pruning run (pruning/pruner.py:26-812)."""
from __future__ import annotations

import datetime
import logging
from abc import ABC, abstractmethod
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from .. import dist as sdist
from .. import engine
from .config import PruningConfig
from .models import ModelFactory
from .pruning_core import PruningOrchestrator, compute_model_statistics
from .sensitivity import TaylorWeightImportance
from .tracking import WandBTracker
from .trainer import PruningMetrics, PruningTrainer
from .utils import set_random_seeds

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
    def train_model(self, trainer: "PruningTrainer", model: nn.Module, epochs: int, step_info: str,
                    teacher_model: Optional[nn.Module] = None) -> nn.Module:
        ...


class StandardTrainingStrategy(TrainingStrategy):
    def train_model(self, trainer, model, epochs, step_info, teacher_model=None):
        return trainer.fine_tune(model, epochs, step_info, teacher_model=None)


class KDTrainingStrategy(TrainingStrategy):
    def train_model(self, trainer, model, epochs, step_info, teacher_model=None):
        return trainer.fine_tune(model, epochs, step_info, teacher_model=teacher_model)


class RefinedKDTrainingStrategy(TrainingStrategy):
    """Standard fine-tuning for `epochs`, then kd_refine_epochs of distillation
    (pruning/pruner.py:371-398)."""

    def __init__(self, config: PruningConfig):
        self.config = config
        self.standard_strategy = StandardTrainingStrategy()
        self.kd_strategy = KDTrainingStrategy()

    def train_model(self, trainer, model, epochs, step_info, teacher_model=None):
        logger.info("Stage 1: Standard fine-tuning")
        model = self.standard_strategy.train_model(trainer, model, epochs, f"{step_info} - Standard")
        logger.info("Stage 2: KD refinement")
        return self.kd_strategy.train_model(trainer, model, getattr(self.config, "kd_refine_epochs", epochs),
                                            f"{step_info} - KD Refine", teacher_model)


class TrainingStrategyFactory:
    @staticmethod
    def create_strategy(config: PruningConfig) -> TrainingStrategy:
        if not config.enable_kd_lite:
            return StandardTrainingStrategy()
        if getattr(config, "kd_mode", "") == "refine":
            return RefinedKDTrainingStrategy(config)
        return KDTrainingStrategy()


class ModelManager:
    """Model loading and the output directory layout (pruning/pruner.py:415-475):
    <output_dir>/[<YYYYmmdd_HHMMSS>_]<model>_<strategy>_<ratio:.2f>[_kd|_kd_refine]/
        final_model.pth  model_info.txt  metrics.json  [sensitivity_report.json | hybrid_pruning_report.json]
        intermediate/<model>_step_<i>_of_<n>.pth"""

    def __init__(self, config: PruningConfig):
        self.config = config
        self.model_factory = ModelFactory()
        self.pruning_dir = self._create_output_directory()
        self.intermediate_dir = self.pruning_dir / "intermediate"
        if sdist.world()[0] == 0:
            self.intermediate_dir.mkdir(parents=True, exist_ok=True)

    def _create_output_directory(self) -> Path:
        c = self.config
        name = f"{c.model_name}_{c.pruning_strategy}_{c.pruning_ratio:.2f}"
        if c.enable_kd_lite:
            name += "_kd_refine" if getattr(c, "kd_mode", "") == "refine" else "_kd"
        if c.use_timestamp:
            name = datetime.datetime.now().strftime("%Y%m%d_%H%M%S") + "_" + name
        rank, world = sdist.world()
        if world > 1:   # one run directory for all ranks: rank 0 names it (its clock), the others follow
            import torch.distributed as tdist
            box = [name]
            tdist.broadcast_object_list(box, src=0)
            name = box[0]
        d = Path(c.output_dir) / name
        if rank == 0:
            d.mkdir(parents=True, exist_ok=True)
        if world > 1:
            sdist.barrier()
        return d

    def load_model(self) -> nn.Module:
        set_random_seeds(self.config.seed)
        model = self.model_factory.load_pretrained_model(self.config)
        model.to(self.config.device)
        return model

    def save_intermediate_model(self, model: nn.Module, step: int, total_steps: int) -> None:
        if sdist.world()[0] != 0:  # one writer per run
            return
        path = self.intermediate_dir / f"{self.config.model_name}_step_{step}_of_{total_steps}.pth"
        model.zero_grad()
        torch.save(model, path)
        logger.debug(f"Saved intermediate pruned model: {path}")

    def save_final_model(self, model: nn.Module) -> Path:
        model.cpu()  # the saved module loads on hosts without a GPU (pruning/pruner.py:467)
        path = self.pruning_dir / "final_model.pth"
        self.model_factory.save_model(model, str(path), self.config)
        return path

    def save_metrics(self, metrics: "PruningMetrics") -> Path:
        path = self.pruning_dir / "metrics.json"
        metrics.save_history(str(path))
        return path


class PruningStepExecutor:
    def __init__(self, config: PruningConfig, model_manager: ModelManager):
        self.config = config
        self.model_manager = model_manager
        self.training_strategy = TrainingStrategyFactory.create_strategy(config)

    def execute_pruning_step(self, step: int, pruner, model: nn.Module, trainer: "PruningTrainer",
                             metrics: "PruningMetrics", teacher_model: Optional[nn.Module] = None) -> nn.Module:
        """`step` counts from 0.  One pruner.step() (importance -> selection -> physical slicing),
        statistics, fine-tuning, intermediate checkpoint, one metrics row.  The row's loss fields
        are written as 0.0, as the reference writes them (pruning/pruner.py:513-519); the real
        losses are on the trainer (last_train_loss / last_val_loss) and go to the tracker."""
        pruner.step()
        stats = compute_model_statistics(model, self._generate_example_inputs(step))
        if step == 0:
            self.initial_params = stats["params"]
        stats["compression_ratio"] = stats["params"] / self.initial_params
        model = self.training_strategy.train_model(trainer, model, self.config.fine_tune_epochs,
                                                   f"Step {step + 1}/{self.config.iterative_steps}", teacher_model)
        self.model_manager.save_intermediate_model(model, step + 1, self.config.iterative_steps)
        metrics.add_step(step=step + 1, train_loss=0.0, val_loss=0.0, params=stats["params"], macs=stats["macs"])
        tracker = getattr(trainer, "wandb_tracker", None)
        if tracker:
            tracker.log_pruning_step(step=step + 1, model_stats=stats,
                                     train_loss=getattr(trainer, "last_train_loss", 0.0),
                                     val_loss=getattr(trainer, "last_val_loss", 0.0))
        return model

    def execute_final_training(self, model: nn.Module, trainer: "PruningTrainer",
                               teacher_model: Optional[nn.Module] = None) -> nn.Module:
        logger.info(f"\n--- Final Fine-tuning ({self.config.final_fine_tune_epochs} epochs) ---")
        return self.training_strategy.train_model(trainer, model, self.config.final_fine_tune_epochs, "Final",
                                                  teacher_model)

    def _generate_example_inputs(self, step: int) -> torch.Tensor:
        torch.manual_seed(self.config.seed + step)
        return torch.randn(*self.config.example_input_size).to(self.config.device)


class StructuredPruner:
    """The whole run (pruning/pruner.py:552-806): load, analyse, prune in steps with fine-tuning,
    final fine-tuning, artefacts.  Under several ranks every rank prunes its replica identically
    (the analysis is all-reduced, selection is deterministic) and rank 0 writes the files."""

    def __init__(self, config: PruningConfig):
        self.config = config
        set_random_seeds(config.seed)
        self.model_manager = ModelManager(config)
        self.trainer = PruningTrainer(config)
        self.metrics = PruningMetrics()
        self.orchestrator = PruningOrchestrator(config)
        self.step_executor = PruningStepExecutor(config, self.model_manager)
        self.wandb_tracker = WandBTracker(config=config, project=config.wandb_project, entity=config.wandb_entity,
                                          name=config.wandb_run_name, tags=config.wandb_tags,
                                          enabled=config.use_wandb)
        self.trainer.set_wandb_tracker(self.wandb_tracker)
        self.orchestrator.set_wandb_tracker(self.wandb_tracker)

    def prune(self) -> Dict[str, Any]:
        self._log_pruning_info()
        try:
            if torch.device(self.config.device).type != "cuda":
                raise RuntimeError(f"device '{self.config.device}': this implementation runs on HIP devices only")
            model, teacher_model, initial_info = self._initialize_models()
            pruner, additional_info = self._setup_pruning(model)
            if sdist.world()[0] == 0:   # one writer per run
                self.orchestrator.save_reports(self.model_manager.pruning_dir, additional_info)
            model = self._execute_progressive_pruning(model, pruner, teacher_model)
            model = self.step_executor.execute_final_training(model, self.trainer, teacher_model)
            return self._finalize_pruning(model, initial_info)
        except Exception as e:
            logger.error(f"Pruning failed: {e}")
            raise
        finally:
            self.trainer.cleanup()

    def _log_pruning_info(self) -> None:
        c = self.config
        logger.info(f"Starting structured pruning of {c.model_name}")
        logger.info(f"Strategy: {c.pruning_strategy}, Ratio: {c.pruning_ratio}, Steps: {c.iterative_steps}")
        if c.enable_kd_lite:
            logger.info(f"KD-Lite enabled: Temperature={c.kd_temperature}, Alpha={c.kd_alpha}, "
                        f"Data ratio={c.kd_data_ratio}")

    def _initialize_models(self) -> Tuple[nn.Module, Optional[nn.Module], Dict]:
        model = self.model_manager.load_model()
        teacher_model = None
        if self.config.enable_kd_lite:
            logger.info("Saving original model as teacher for KD-Lite")
            teacher_model = self.model_manager.load_model()
            teacher_model.eval()
            for p in teacher_model.parameters():
                p.requires_grad = False
        initial_info = self.model_manager.model_factory.get_model_info(model)
        example_inputs = torch.randn(*self.config.example_input_size).to(self.config.device)
        stats = compute_model_statistics(model, example_inputs)
        initial_info["macs"], initial_info["macs_million"] = stats["macs"], stats["macs_million"]
        logger.info(f"Initial model: {initial_info['total_params']:,} parameters, "
                    f"{initial_info['model_size_mb']:.2f} MB, {initial_info['macs_million']:.2f}M MACs")
        self.wandb_tracker.log_initial_model_info(initial_info)
        return model, teacher_model, initial_info

    def _setup_pruning(self, model: nn.Module):
        ignored = self.model_manager.model_factory.get_ignored_layers(model, self.config)
        torch.manual_seed(self.config.seed)
        example_inputs = torch.randn(*self.config.example_input_size).to(self.config.device)
        self.wandb_tracker.log_model_graph(model, example_inputs)
        pruner, info = self.orchestrator.setup_importance_and_pruner(model, example_inputs, ignored)
        if "sensitivity_report" in info:
            self.wandb_tracker.log_sensitivity_analysis(info["sensitivity_report"])
        elif "hybrid_report" in info:
            quota = info["hybrid_report"].get("quota_allocation")
            self.wandb_tracker.log_sensitivity_analysis(info["hybrid_report"], layer_quotas=quota)
            if quota and quota.get("layer_details"):
                self.wandb_tracker.log_layer_pruning_details(quota["layer_details"])
        return pruner, info

    def _execute_progressive_pruning(self, model: nn.Module, pruner, teacher_model: Optional[nn.Module]) -> nn.Module:
        logger.info("Starting progressive pruning...")
        for step in range(self.config.iterative_steps):
            model = self.step_executor.execute_pruning_step(step, pruner, model, self.trainer, self.metrics,
                                                            teacher_model)
            row = self.metrics.history[-1]
            logger.info(f"Pruning Step {step + 1}/{self.config.iterative_steps}: "
                        f"Params {row['params_million']:.2f}M, MACs {row['macs_million']:.2f}M")
        return model

    def _finalize_pruning(self, model: nn.Module, initial_info: Dict) -> Dict[str, Any]:
        model.to(self.config.device)
        example_inputs = torch.randn(*self.config.example_input_size).to(self.config.device)
        final_stats = compute_model_statistics(model, example_inputs)
        rank, _ = sdist.world()
        if rank == 0:
            final_path = self.model_manager.save_final_model(model)
            metrics_path = self.model_manager.save_metrics(self.metrics)
        else:
            final_path = self.model_manager.pruning_dir / "final_model.pth"
            metrics_path = self.model_manager.pruning_dir / "metrics.json"
        sdist.barrier()
        final_info = self.model_manager.model_factory.get_model_info(model)
        final_info["macs"], final_info["macs_million"] = final_stats["macs"], final_stats["macs_million"]
        results = self._build_results_dict(initial_info, final_info, final_path, metrics_path)
        self._log_final_results(results)
        self.wandb_tracker.log_final_results(results)
        self.wandb_tracker.create_comparison_chart(
            {"params": initial_info["total_params"], "macs": initial_info.get("macs", 0)},
            {"params": final_info["total_params"], "macs": final_info.get("macs", 0)})
        if self.config.wandb_save_model:
            self.wandb_tracker.save_model_checkpoint(str(final_path),
                                                     aliases=["latest", f"ratio_{self.config.pruning_ratio}"])
        self.wandb_tracker.finish()
        return results

    def _build_results_dict(self, initial_info: Dict, final_info: Dict, final_path: Path,
                            metrics_path: Path) -> Dict[str, Any]:
        c = self.config

        def drop(key):
            before = initial_info.get(key, 0)
            return (before - final_info.get(key, 0)) / before if before > 0 else 0

        results = {
            "model_name": c.model_name, "pruning_strategy": c.pruning_strategy, "pruning_ratio": c.pruning_ratio,
            "initial_params": initial_info["total_params"], "final_params": final_info["total_params"],
            "params_reduction": drop("total_params"),
            "initial_size_mb": initial_info["model_size_mb"], "final_size_mb": final_info["model_size_mb"],
            "size_reduction": drop("model_size_mb"),
            "initial_macs": initial_info.get("macs", 0), "final_macs": final_info.get("macs", 0),
            "macs_reduction": drop("macs"),
            "output_path": str(final_path), "metrics_path": str(metrics_path),
            "pruning_dir": str(self.model_manager.pruning_dir), "kd_lite_enabled": c.enable_kd_lite,
        }
        if c.enable_kd_lite:
            results.update({"kd_temperature": c.kd_temperature, "kd_alpha": c.kd_alpha,
                            "kd_data_ratio": c.kd_data_ratio})
        return results

    def _log_final_results(self, results: Dict[str, Any]) -> None:
        logger.info("\n--- Pruning Complete ---")
        logger.info(f"Parameters: {results['initial_params']:,} → {results['final_params']:,} "
                    f"({results['params_reduction']:.2%} reduction)")
        logger.info(f"Model size: {results['initial_size_mb']:.2f} MB → {results['final_size_mb']:.2f} MB "
                    f"({results['size_reduction']:.2%} reduction)")
        logger.info(f"Saved to: {results['output_path']}")


def prune_model(config: PruningConfig) -> Dict[str, Any]:
    set_random_seeds(config.seed)
    return StructuredPruner(config).prune()
