"""Strategy dispatch and reports (pruning/pruning_core.py:21-203)."""
from __future__ import annotations

import json
import logging
from pathlib import Path
from typing import Any, Dict, List, Tuple

import torch
import torch.nn as nn

from .. import dist as sdist
from .. import engine
from .config import PruningConfig
from .dataset import PruningDataHandler
from .sensitivity import create_taylor_importance

logger = logging.getLogger(__name__)


class PruningOrchestrator:
    def __init__(self, config: PruningConfig):
        self.config = config
        self.wandb_tracker = None

    def set_wandb_tracker(self, tracker):
        self.wandb_tracker = tracker

    def _calibration_loader(self):
        c = self.config
        handler = PruningDataHandler(data_root=c.data_root, batch_size=c.batch_size,
                                     num_workers=c.num_workers, seed=c.seed,
                                     num_classes=c.num_classes, input_size=c.input_size)
        return handler.get_calibration_dataloader()

    def setup_importance_and_pruner(self, model: nn.Module, example_inputs: torch.Tensor,
                                    ignored_layers: List[nn.Module]) -> Tuple[engine.MagnitudePruner, Dict]:
        strategy = self.config.pruning_strategy
        info: Dict[str, Any] = {}
        if strategy == "taylor_weight":
            logger.info("Using Taylor weight-based sensitivity analysis (pure Taylor mode)...")
            importance = create_taylor_importance(model, self._calibration_loader(), self.config)
            info["sensitivity_report"] = importance.get_sensitivity_report()
            pruner = engine.MagnitudePruner(
                model=model, example_inputs=example_inputs, importance=importance,
                pruning_ratio=self.config.pruning_ratio, iterative_steps=self.config.iterative_steps,
                ignored_layers=ignored_layers, global_pruning=True)
            return pruner, info
        if strategy == "magnitude_taylor":
            from .pruner import create_hybrid_pruner
            logger.info("Using hybrid Magnitude+Taylor pruning (Taylor for quotas, Magnitude for selection)...")
            taylor = create_taylor_importance(model, self._calibration_loader(), self.config)
            pruner, summary = create_hybrid_pruner(model=model, config=self.config, taylor_importance=taylor,
                                                   example_inputs=example_inputs, ignored_layers=ignored_layers)
            report = taylor.get_sensitivity_report()
            report["quota_allocation"] = summary
            info["hybrid_report"] = report
            return pruner, info
        return self._setup_standard_pruning(model, example_inputs, ignored_layers, strategy), info

    def _setup_standard_pruning(self, model, example_inputs, ignored_layers, strategy: str):
        makers = {"magnitude": lambda: engine.MagnitudeImportance(p=self.config.importance_norm),
                  "random": engine.RandomImportance, "taylor": engine.TaylorImportance,
                  "lamp": engine.LAMPImportance}
        if strategy not in makers:
            raise ValueError(f"Unknown pruning strategy: {strategy}")
        return engine.MagnitudePruner(
            model=model, example_inputs=example_inputs, importance=makers[strategy](),
            pruning_ratio=self.config.pruning_ratio, iterative_steps=self.config.iterative_steps,
            ignored_layers=ignored_layers, global_pruning=True)

    def save_reports(self, pruning_dir: Path, additional_info: Dict) -> None:
        if sdist.world()[0] != 0:  # one writer per run
            return
        for key, fname in (("sensitivity_report", "sensitivity_report.json"),
                           ("hybrid_report", "hybrid_pruning_report.json")):
            if key in additional_info:
                path = Path(pruning_dir) / fname
                with open(path, "w") as f:
                    json.dump(additional_info[key], f, indent=2)
                logger.info(f"Saved {key.replace('_', ' ')} to {path}")


def compute_model_statistics(model: nn.Module, example_inputs: torch.Tensor) -> Dict[str, Any]:
    try:
        macs, nparams = engine.count_ops_and_params(model, example_inputs)
    except Exception as e:  # mirrors the reference's fallback (pruning_core.py:193-196)
        logger.warning(f"Could not calculate MACs: {e}")
        macs, nparams = 0, sum(p.numel() for p in model.parameters())
    return {"params": nparams, "params_million": nparams / 1e6, "macs": macs, "macs_million": macs / 1e6}
