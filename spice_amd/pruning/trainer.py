"""Fine-tuning after each pruning step (pruning/trainer.py:30-410): Adam, warm-up + cosine,
early stopping with best-weights restore; with a teacher the loss is the fused KD-Lite kernel,
and validation accumulates the mIoU confusion matrix with the fused arg-max histogram."""
from __future__ import annotations

import copy
import json
import logging
from pathlib import Path
from typing import Dict, List, Optional

import torch
import torch.nn as nn
from torch.utils.data import DataLoader, Subset

from .. import ops
from .kd_lite import KnowledgeDistiller
from .losses import SegmentationLoss
from .lr_scheduler import create_scheduler

logger = logging.getLogger(__name__)


class PruningMetrics:
    """Per-step history written to metrics.json (pruning/trainer.py:361-410)."""

    def __init__(self):
        self.history: List[Dict] = []

    def add_step(self, step: int, **values) -> None:
        self.history.append({"step": step, **values})

    def save(self, path) -> None:
        with open(path, "w") as f:
            json.dump(self.history, f, indent=2, default=float)


class PruningTrainer:
    def __init__(self, config, wandb_tracker=None):
        self.config = config
        self.wandb_tracker = wandb_tracker
        self.criterion = SegmentationLoss(config.num_classes).to(config.device)
        self.distiller: Optional[KnowledgeDistiller] = None

    def setup_kd_training(self, teacher: nn.Module, student: nn.Module) -> None:
        self.distiller = KnowledgeDistiller(self.config)
        self.distiller.setup_teacher(teacher)
        self.distiller.setup_student(student)

    def _create_kd_subset_loader(self, loader: DataLoader) -> DataLoader:
        n = len(loader.dataset)
        keep = max(1, int(n * self.config.kd_data_ratio))
        g = torch.Generator().manual_seed(self.config.seed)
        idx = torch.randperm(n, generator=g)[:keep].tolist()
        return DataLoader(Subset(loader.dataset, idx), batch_size=loader.batch_size, shuffle=True)

    def _train_epoch(self, model, loader, optimizer) -> Dict[str, float]:
        model.train()
        dev = self.config.device
        sums = torch.zeros(3, device=dev)
        n = 0
        for images, masks in loader:
            images, masks = images.to(dev), masks.to(dev)
            optimizer.zero_grad(set_to_none=True)
            student_out = model(images)
            if self.distiller is not None:
                teacher_out = self.distiller.teacher_forward(images)
                d = self.distiller.compute_loss(student_out, teacher_out, masks)
                loss = d["total"]
                sums += torch.stack([d["total"].detach(), d["ce"].detach(), d["kd"].detach()])
            else:
                loss = self.criterion(student_out, masks)
                sums[0] += loss.detach()
                sums[1] += loss.detach()
            loss.backward()
            optimizer.step()
            n += 1
        t = (sums / max(n, 1)).tolist()  # one host sync per epoch
        return {"loss": t[0], "ce": t[1], "kd": t[2]}

    @torch.no_grad()
    def _validate_epoch(self, model, loader) -> Dict[str, float]:
        model.eval()
        dev = self.config.device
        c = self.config.num_classes
        conf = torch.zeros(c, c, dtype=torch.int64, device=dev)
        loss_sum = torch.zeros(1, device=dev)
        n = 0
        for images, masks in loader:
            images, masks = images.to(dev), masks.to(dev).long().contiguous()
            out = model(images)
            loss_sum += self.criterion(out, masks)
            ops.argmax_confusion_update(conf, out.contiguous(), masks, c)
            n += 1
        m = conf.double()
        hit = m.diag()
        iou = hit / (m.sum(0) + m.sum(1) - hit + 1e-10)
        return {"val_loss": (loss_sum / max(n, 1)).item(), "val_miou": iou.mean().item()}

    def fine_tune(self, model: nn.Module, train_loader: DataLoader, val_loader: DataLoader, epochs: int,
                  teacher: Optional[nn.Module] = None) -> Dict[str, float]:
        if epochs <= 0:
            return {}
        cfg = self.config
        if teacher is not None and cfg.enable_kd_lite:
            self.setup_kd_training(teacher, model)
            train_loader = self._create_kd_subset_loader(train_loader)
        else:
            self.distiller = None
        params = [p for p in model.parameters() if p.requires_grad]
        optimizer = torch.optim.Adam(params, lr=cfg.fine_tune_lr, weight_decay=cfg.weight_decay)
        scheduler = create_scheduler(optimizer, cfg, epochs)
        best, best_state, stale = float("inf"), None, 0
        last = {}
        for epoch in range(epochs):
            tr = self._train_epoch(model, train_loader, optimizer)
            va = self._validate_epoch(model, val_loader)
            last = {**tr, **va, "epoch": epoch}
            if isinstance(scheduler, torch.optim.lr_scheduler.ReduceLROnPlateau):
                scheduler.step(va["val_loss"])
            else:
                scheduler.step()
            if va["val_loss"] < best:
                best, stale = va["val_loss"], 0
                best_state = copy.deepcopy(model.state_dict())
            else:
                stale += 1
                if stale >= cfg.early_stop_patience:
                    logger.info(f"Early stop at epoch {epoch}")
                    break
        if best_state is not None:
            model.load_state_dict(best_state)
        return last

    def cleanup(self) -> None:
        self.distiller = None
