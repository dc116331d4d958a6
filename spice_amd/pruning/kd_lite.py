"""Teacher/student set-up for KD-Lite (pruning/kd_lite.py:17-102)."""
from __future__ import annotations

import logging
from typing import Dict, Optional

import torch
import torch.nn as nn

from .losses import KDLiteLoss

logger = logging.getLogger(__name__)


class KnowledgeDistiller:
    def __init__(self, config, model_factory=None):
        self.config = config
        self.model_factory = model_factory
        self.teacher_model: Optional[nn.Module] = None
        self.loss_fn: Optional[KDLiteLoss] = None

    def _freeze_teacher(self) -> None:
        self.teacher_model.to(self.config.device)
        self.teacher_model.eval()
        for p in self.teacher_model.parameters():
            p.requires_grad = False

    def setup_teacher(self, teacher_path: Optional[str] = None) -> None:
        """Teacher = a whole-module file written by this toolkit, or the configured checkpoint
        (pruning/kd_lite.py:26-41).  A module object may be handed over through `teacher_model`."""
        if teacher_path:
            # a module pickle the user wrote with save_model(); unpickling runs code from that file
            self.teacher_model = torch.load(teacher_path, map_location=self.config.device, weights_only=False)
        else:
            if self.model_factory is None:
                raise RuntimeError("No model factory to build the teacher from")
            self.teacher_model = self.model_factory.load_pretrained_model(self.config)
        self._freeze_teacher()
        logger.info("Teacher model loaded and frozen")

    def setup_student(self, student_model: nn.Module) -> nn.Module:
        c = self.config
        student_model.to(c.device)
        if c.freeze_backbone:
            self._freeze_backbone(student_model)
        self.loss_fn = KDLiteLoss(temperature=c.kd_temperature, alpha=c.kd_alpha,
                                  boundary_weight=c.boundary_weight, confidence_weight=c.confidence_weight,
                                  num_classes=c.num_classes).to(c.device)
        return student_model

    def _freeze_backbone(self, model: nn.Module) -> None:
        if hasattr(model, "encoder"):
            for p in model.encoder.parameters():
                p.requires_grad = False
            for m in model.encoder.modules():
                if isinstance(m, nn.BatchNorm2d):
                    m.eval()
        trainable = sum(p.numel() for p in model.parameters() if p.requires_grad)
        total = sum(p.numel() for p in model.parameters())
        logger.info(f"Trainable parameters: {trainable:,} / {total:,} ({100 * trainable / max(total, 1):.2f}%)")

    def get_teacher_logits(self, inputs: torch.Tensor) -> torch.Tensor:
        if self.teacher_model is None:
            raise RuntimeError("Teacher model not initialized. Call setup_teacher() first.")
        with torch.no_grad():
            return self.teacher_model(inputs)

    def compute_loss(self, student_outputs: torch.Tensor, inputs: torch.Tensor,
                     labels: torch.Tensor) -> Dict[str, torch.Tensor]:
        """Teacher forward on `inputs` (the images), then the fused loss (pruning/kd_lite.py:79-91)."""
        return self.loss_fn(student_outputs, self.get_teacher_logits(inputs), labels)
