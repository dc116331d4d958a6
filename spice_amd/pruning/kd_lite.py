"""Teacher/student set-up for KD-Lite (pruning/kd_lite.py:17-102)."""
from __future__ import annotations

import logging
from typing import Dict, Optional

import torch
import torch.nn as nn

from .losses import KDLiteLoss

logger = logging.getLogger(__name__)


class KnowledgeDistiller:
    def __init__(self, config):
        self.config = config
        self.teacher: Optional[nn.Module] = None
        self.loss_fn: Optional[KDLiteLoss] = None

    def setup_teacher(self, teacher: nn.Module) -> None:
        self.teacher = teacher.eval()
        for p in self.teacher.parameters():
            p.requires_grad_(False)

    def setup_student(self, student: nn.Module) -> None:
        c = self.config
        if getattr(c, "freeze_backbone", False) and hasattr(student, "encoder"):
            for p in student.encoder.parameters():
                p.requires_grad_(False)
            logger.info("Encoder frozen for KD training")
        self.loss_fn = KDLiteLoss(temperature=c.kd_temperature, alpha=c.kd_alpha,
                                  boundary_weight=c.boundary_weight, confidence_weight=c.confidence_weight,
                                  num_classes=c.num_classes).to(c.device)

    def compute_loss(self, student_logits, teacher_logits, labels) -> Dict[str, torch.Tensor]:
        if self.loss_fn is None:
            raise RuntimeError("KD loss not initialised. Call setup_student first.")
        return self.loss_fn(student_logits, teacher_logits, labels)

    def teacher_forward(self, images):
        if self.teacher is None:
            raise RuntimeError("Teacher not initialised. Call setup_teacher first.")
        with torch.no_grad():
            return self.teacher(images)
