"""Model factory and persistence (pruning/models.py:19-138)."""
from __future__ import annotations

import logging
from pathlib import Path
from typing import List, Optional

import torch
import torch.nn as nn

from .. import nets
from .config import PruningConfig
from .utils import calculate_model_size_mb, count_model_parameters, set_random_seeds

logger = logging.getLogger(__name__)


class ModelFactory:
    @staticmethod
    def create_model(config: PruningConfig) -> nn.Module:
        """Seeded construction of the configured architecture (pruning/models.py:22-43); the class
        count is config.num_classes (5 by default, as in the model table)."""
        set_random_seeds(config.seed)
        mc = config.model_config
        if mc["architecture"] not in nets.ARCHITECTURES:
            raise ValueError(f"Unknown architecture: {mc['architecture']}")
        return nets.create_model(mc["architecture"], mc["encoder_name"], classes=config.num_classes,
                                 in_channels=mc["in_channels"])

    @staticmethod
    def load_pretrained_model(config: PruningConfig) -> nn.Module:
        """Missing checkpoint: continue with the seeded random weights; unreadable checkpoint:
        RuntimeError (pruning/models.py:46-72).  State dicts are read with weights_only=True."""
        model = ModelFactory.create_model(config)
        ckpt = Path(config.checkpoint_path)
        if not ckpt.exists():
            logger.warning(f"Checkpoint not found: {ckpt}")
            logger.info("Using model with random initialization")
            return model
        try:
            model.load_state_dict(torch.load(ckpt, map_location="cpu", weights_only=True))
            logger.info(f"Loaded pretrained weights from {ckpt}")
        except Exception as e:
            logger.error(f"Failed to load checkpoint {ckpt}: {e}")
            raise RuntimeError("Checkpoint loading failed - cannot proceed with random weights")
        return model

    @staticmethod
    def get_ignored_layers(model: nn.Module, config: Optional[PruningConfig] = None) -> List[nn.Module]:
        """Every (transposed) convolution of the segmentation head keeps its channels; a
        torchvision-style DeepLab without that head keeps its last classifier layer
        (pruning/models.py:75-101)."""
        ignored: List[nn.Module] = []
        head = getattr(model, "segmentation_head", None)
        if head is not None:
            ignored = [m for m in head.modules() if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d))]
        elif hasattr(model, "classifier"):
            ignored = [model.classifier[-1]]
        logger.info(f"Ignored layers: {len(ignored)}")
        return ignored

    @staticmethod
    def get_model_info(model: nn.Module) -> dict:
        return {"total_params": count_model_parameters(model),
                "trainable_params": count_model_parameters(model, only_trainable=True),
                "model_size_mb": calculate_model_size_mb(model)}

    @staticmethod
    def save_model(model: nn.Module, path, config: PruningConfig) -> None:
        """Whole-module pickle (the sliced architecture travels with the weights) plus the
        six-line model_info.txt beside it (pruning/models.py:116-138)."""
        path = Path(path)
        path.parent.mkdir(parents=True, exist_ok=True)
        model.zero_grad()
        torch.save(model, path)
        logger.info(f"Pruned model saved to {path}")
        info = ModelFactory.get_model_info(model)
        lines = [f"Model: {config.model_name}", f"Pruning Strategy: {config.pruning_strategy}",
                 f"Pruning Ratio: {config.pruning_ratio}", f"Total Parameters: {info['total_params']:,}",
                 f"Trainable Parameters: {info['trainable_params']:,}",
                 f"Model Size: {info['model_size_mb']:.2f} MB"]
        (path.parent / "model_info.txt").write_text("\n".join(lines) + "\n")
