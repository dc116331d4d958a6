"""Importance plug-ins that serve precomputed Taylor scores to the pruner
(pruning/sensitivity.py:23-215).  A group iterates as (dep, idxs) pairs with
`dep.target.module`, the protocol the reference's plug-ins are written against."""
from __future__ import annotations

import logging
from typing import Dict

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..engine.importance import Importance
from .sensitivity_utils import compute_taylor_scores, get_layer_sensitivities, get_sensitivity_report

logger = logging.getLogger(__name__)


def _members(group):
    """(module, idxs) pairs of either an engine ChannelGroup or a list of (dep, idxs)."""
    if hasattr(group, "members"):
        idxs = list(range(group.size))
        return [(m.module, idxs) for m in group.members]
    return [(dep.target.module, idxs) for dep, idxs in group]


class TaylorWeightImportance(Importance):
    """Scores of the group's first Conv2d, low = prune first (pruning/sensitivity.py:70-112)."""

    def __init__(self, model: nn.Module, dataloader, device: str, max_batches: int = 100,
                 min_channels: int = 8):
        self.model = model
        self.device = device
        self.min_channels = min_channels
        logger.info("Computing Taylor weight sensitivity scores...")
        self.sensitivity_scores = self._compute_sensitivity(model, dataloader, device, max_batches)
        logger.info(f"Computed sensitivity for {len(self.sensitivity_scores)} layers")

    def _compute_sensitivity(self, model, dataloader, device, max_batches) -> Dict[nn.Module, torch.Tensor]:
        # with several ranks: calibration batches round-robin, one all-reduce of the flat score buffer
        return compute_taylor_scores(model, dataloader, device, max_batches, shard=True)

    def __call__(self, group) -> torch.Tensor:
        members = _members(group) if group else []
        if not members:
            return torch.tensor([])
        first_idxs = members[0][1]
        primary = next((m for m, _ in members if isinstance(m, nn.Conv2d)), None)
        if primary is None or primary not in self.sensitivity_scores:
            return torch.ones(len(first_idxs))
        scores = self.sensitivity_scores[primary]
        if len(first_idxs) < len(scores):
            scores = scores[torch.tensor(first_idxs, device=scores.device)]
        return scores

    def get_sensitivity_report(self) -> Dict:
        return get_sensitivity_report(self.sensitivity_scores)

    def get_layer_sensitivities(self) -> Dict[nn.Module, float]:
        return get_layer_sensitivities(self.sensitivity_scores)


class GroupSyncTaylorImportance(TaylorWeightImportance):
    """Mean of the scores of every scored member, each truncated / padded with 1.0 to the group
    width (pruning/sensitivity.py:139-171)."""

    def __call__(self, group) -> torch.Tensor:
        members = _members(group) if group else []
        if not members:
            return torch.tensor([])
        width = len(members[0][1])
        rows = []
        for module, _ in members:
            if module in self.sensitivity_scores:
                s = self.sensitivity_scores[module]
                s = s[:width] if len(s) >= width else F.pad(s, (0, width - len(s)), value=1.0)
                rows.append(s)
        if not rows:
            return torch.ones(width)
        return torch.stack(rows, dim=0).mean(dim=0)


def create_taylor_importance(model: nn.Module, dataloader, config) -> TaylorWeightImportance:
    grouped = any(tag in config.model_name for tag in ("resnet", "FPN", "UNET"))
    cls = GroupSyncTaylorImportance if grouped else TaylorWeightImportance
    if grouped:
        logger.info("Using group-synchronized Taylor importance")
    return cls(model=model, dataloader=dataloader, device=config.device,
               max_batches=getattr(config, "calibration_batches", 100),
               min_channels=getattr(config, "min_out_channels", 8))
