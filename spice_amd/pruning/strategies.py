"""Strategy objects: name -> importance criterion -> pruner (pruning/strategies.py:18-159)."""
from __future__ import annotations

import logging
from abc import ABC, abstractmethod
from typing import List

import torch
import torch.nn as nn

from .. import engine
from .config import PruningConfig

logger = logging.getLogger(__name__)


class BasePruningStrategy(ABC):
    def __init__(self, config: PruningConfig):
        self.config = config

    @abstractmethod
    def create_importance(self) -> engine.Importance:
        ...

    def create_pruner(self, model: nn.Module, example_inputs: torch.Tensor,
                      ignored_layers: List[nn.Module]) -> engine.MagnitudePruner:
        return engine.MagnitudePruner(
            model=model, example_inputs=example_inputs, importance=self.create_importance(),
            pruning_ratio=self.config.pruning_ratio, iterative_steps=self.config.iterative_steps,
            ignored_layers=ignored_layers, global_pruning=True)


class MagnitudePruningStrategy(BasePruningStrategy):
    def create_importance(self):
        return engine.MagnitudeImportance(p=self.config.importance_norm)


class RandomPruningStrategy(BasePruningStrategy):
    def create_importance(self):
        return engine.RandomImportance()


class TaylorPruningStrategy(BasePruningStrategy):
    def create_importance(self):
        return engine.TaylorImportance()


class LAMPPruningStrategy(BasePruningStrategy):
    def create_importance(self):
        return engine.LAMPImportance()


class PruningStrategyFactory:
    STRATEGIES = {"magnitude": MagnitudePruningStrategy, "random": RandomPruningStrategy,
                  "taylor": TaylorPruningStrategy, "lamp": LAMPPruningStrategy}

    @classmethod
    def create_strategy(cls, config: PruningConfig) -> BasePruningStrategy:
        name = config.pruning_strategy.lower()
        if name not in cls.STRATEGIES:
            raise ValueError(f"Unknown pruning strategy: {name}")
        return cls.STRATEGIES[name](config)

    @classmethod
    def list_strategies(cls) -> List[str]:
        return list(cls.STRATEGIES.keys())


class ProgressivePruner:
    """setup_pruner() then prune_step() per iteration (pruning/strategies.py:103-159)."""

    def __init__(self, strategy: BasePruningStrategy, config: PruningConfig):
        self.strategy = strategy
        self.config = config
        self.pruner = None

    def _example_inputs(self) -> torch.Tensor:
        torch.manual_seed(self.config.seed)
        return torch.randn(*self.config.example_input_size).to(self.config.device)

    def setup_pruner(self, model: nn.Module, ignored_layers: List[nn.Module]):
        self.pruner = self.strategy.create_pruner(model, self._example_inputs(), ignored_layers)
        return self.pruner

    def prune_step(self) -> dict:
        if self.pruner is None:
            raise RuntimeError("Pruner not setup. Call setup_pruner first.")
        self.pruner.step()
        from .pruning_core import compute_model_statistics
        stats = compute_model_statistics(self.pruner.model, self._example_inputs())
        logger.info(f"After pruning step: {stats['params_million']:.2f}M params, "
                    f"{stats['macs_million']:.2f}M MACs")
        return stats
