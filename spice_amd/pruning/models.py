"""Model factory and persistence (pruning/models.py:20-138)."""
from __future__ import annotations

import logging
from pathlib import Path
from typing import List

import torch
import torch.nn as nn

from .. import nets
from .config import PruningConfig
from .utils import calculate_model_size_mb, count_model_parameters, model_checksum, set_random_seeds

logger = logging.getLogger(__name__)


class ModelFactory:
    @staticmethod
    def create_model(config: PruningConfig) -> nn.Module:
        mc = config.model_config
        if mc["architecture"] not in nets.ARCHITECTURES:
            raise ValueError(f"Unsupported architecture: {mc['architecture']}")
        return nets.create_model(mc["architecture"], mc["encoder_name"], classes=config.num_classes,
                                 in_channels=mc["in_channels"])

    @staticmethod
    def load_pretrained_model(config: PruningConfig) -> nn.Module:
        model = ModelFactory.create_model(config)
        ckpt = Path(config.checkpoint_path)
        if not ckpt.exists():  # the reference continues with random weights too (models.py:50-53)
            logger.warning(f"Checkpoint not found at {ckpt}; using randomly initialised weights")
        else:
            state = torch.load(ckpt, map_location="cpu", weights_only=True)
            if isinstance(state, dict) and "state_dict" in state:
                state = state["state_dict"]
            model.load_state_dict(state)
            logger.info(f"Loaded checkpoint {ckpt} (checksum {model_checksum(model)})")
        set_random_seeds(config.seed)
        return model.to(config.device)

    @staticmethod
    def get_ignored_layers(model: nn.Module) -> List[nn.Module]:
        """Every convolution of the segmentation head keeps its channels (models.py:75-101)."""
        head = getattr(model, "segmentation_head", None)
        if head is None:
            logger.warning("Model has no segmentation_head; nothing is ignored")
            return []
        return [m for m in head.modules() if isinstance(m, nn.Conv2d)]

    @staticmethod
    def get_model_info(model: nn.Module) -> dict:
        return {"parameters": count_model_parameters(model), "size_mb": calculate_model_size_mb(model)}

    @staticmethod
    def save_model(model: nn.Module, path, extra: dict = None) -> None:
        """Whole-module pickle, as the reference does, so the sliced architecture travels with it."""
        path = Path(path)
        path.parent.mkdir(parents=True, exist_ok=True)
        was = model.training
        model.eval()
        torch.save(model, path)
        model.train(was)
        info = ModelFactory.get_model_info(model)
        with open(path.parent / "model_info.txt", "w") as f:
            f.write(f"Parameters: {info['parameters']:,}\nSize (MB): {info['size_mb']:.2f}\n")
            for k, v in (extra or {}).items():
                f.write(f"{k}: {v}\n")
