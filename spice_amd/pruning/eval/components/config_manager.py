"""Device, logging, seeds and output directories of an evaluation run
(pruning/eval/components/config_manager.py:13-66)."""
from __future__ import annotations

import logging
from pathlib import Path

import torch

from ...utils import set_random_seeds
from ..config import EvaluationConfig

logger = logging.getLogger(__name__)


class ConfigManager:
    def __init__(self, config: EvaluationConfig):
        self.config = config
        self.device = torch.device(config.device)
        logging.basicConfig(level=logging.INFO, format="%(asctime)s - %(name)s - %(levelname)s - %(message)s")
        set_random_seeds(42)

    def create_output_directories(self) -> None:
        Path(self.config.output_dir).mkdir(parents=True, exist_ok=True)
        if self.config.generate_visualizations:
            (Path(self.config.visualization_dir) / "advanced").mkdir(parents=True, exist_ok=True)

    def should_use_wandb(self) -> bool:
        return self.config.use_wandb

    def get_wandb_config(self) -> dict:
        c = self.config
        return {"project": c.wandb_project, "entity": c.wandb_entity, "name": c.wandb_run_name or "pruning_evaluation",
                "tags": c.wandb_tags, "config": c.__dict__}

    @property
    def visualization_enabled(self) -> bool:
        return self.config.generate_visualizations

    @property
    def cache_enabled(self) -> bool:
        return getattr(self.config, "use_cache", True)
