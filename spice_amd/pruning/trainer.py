"""Fine-tuning after each pruning step (pruning/trainer.py:27-410): Adam, warm-up + cosine
schedule, early stopping with best-weights restore.  With a teacher the loss is the fused KD-Lite
kernel (forward and gradient in one pass over the logits); validation also accumulates the mIoU
confusion matrix with the fused arg-max histogram.  With several ranks (one process per GPU) the
training split is sharded, gradients are averaged through flat buckets whose all-reduces overlap
the backward pass, and every decision (early stop, best epoch) is taken on all-reduced values so
the ranks stay in lock step.  Running losses are accumulated on the device: one host
synchronisation per epoch instead of three per batch."""
from __future__ import annotations

import json
import logging
import os
import random
import shutil
from pathlib import Path
from typing import Dict, Optional, Tuple

import torch
import torch.nn as nn
import torch.optim as optim
from torch.optim.lr_scheduler import ReduceLROnPlateau
from torch.utils.data import DataLoader, Subset

from .. import dist as sdist
from .. import ops
from .config import PruningConfig
from .dataset import PruningDataHandler
from .kd_lite import KnowledgeDistiller
from .losses import SegmentationLoss
from .lr_scheduler import create_scheduler
from .utils import generate_config_hash, set_random_seeds

logger = logging.getLogger(__name__)


class PruningTrainer:
    def __init__(self, config: PruningConfig):
        self.config = config
        self.data_handler = PruningDataHandler(
            data_root=config.data_root, batch_size=config.batch_size, num_workers=config.num_workers,
            seed=config.seed, num_classes=config.num_classes, input_size=config.input_size)
        self._setup_seeds()
        tag = generate_config_hash(f"{config.model_name}_{config.pruning_strategy}_{config.pruning_ratio}_{config.seed}")
        self.temp_dir = Path(config.output_dir) / "temp" / f"{config.model_name}_{tag}_{os.getpid()}"
        self.temp_dir.mkdir(parents=True, exist_ok=True)
        self.distiller: Optional[KnowledgeDistiller] = None
        self.kd_loss_fn = None
        self.wandb_tracker = None
        self.last_train_loss = 0.0
        self.last_val_loss = 0.0
        self.last_val_miou = 0.0
        self.last_kd_loss = 0.0
        self.last_ce_loss = 0.0
        self._grad_buckets: Optional[sdist.FlatGradBuckets] = None

    def _setup_seeds(self) -> None:
        set_random_seeds(self.config.seed)

    def set_wandb_tracker(self, tracker) -> None:
        self.wandb_tracker = tracker

    # ---------------------------------------------------------------------------- data
    def _get_dataloaders(self, use_subset: bool = False) -> Tuple[DataLoader, DataLoader]:
        train_loader, val_loader = self.data_handler.get_train_val_loaders()
        if use_subset and self.config.enable_kd_lite:
            train_loader = self._create_kd_subset_loader(train_loader)
        return train_loader, val_loader

    def _create_kd_subset_loader(self, train_loader: DataLoader) -> DataLoader:
        """A seeded sample of int(len * kd_data_ratio) training items, reshuffled every epoch
        (pruning/trainer.py:69-94).  Under several ranks every rank draws the same sample, the list is
        padded with its own first items to a multiple of the world size (as DistributedSampler pads),
        and each rank keeps its stride of it: every rank then sees the same number of samples and of
        batches, so the gradient all-reduces of the ranks pair up."""
        data = train_loader.dataset
        n = len(data)
        keep = int(n * self.config.kd_data_ratio)
        picked = random.Random(self.config.seed).sample(range(n), keep)
        rank, world = sdist.world()
        if world > 1 and picked:
            short = (-len(picked)) % world
            picked = picked + (picked * world)[:short]
            picked = picked[rank::world]
        g = torch.Generator()
        g.manual_seed(self.config.seed)
        logger.info(f"Using {keep}/{n} samples ({self.config.kd_data_ratio * 100:.0f}%) for KD-Lite training")
        return DataLoader(Subset(data, picked), batch_size=self.config.batch_size, shuffle=True, num_workers=0,
                          generator=g)

    # ---------------------------------------------------------------------------- set-up
    def _setup_criterion(self) -> nn.Module:
        return SegmentationLoss(self.config.num_classes).to(self.config.device)

    def _setup_optimizer(self, model: nn.Module) -> optim.Optimizer:
        return optim.Adam([p for p in model.parameters() if p.requires_grad], lr=self.config.fine_tune_lr,
                          weight_decay=self.config.weight_decay)

    def _setup_scheduler(self, optimizer: optim.Optimizer, num_epochs: int):
        return create_scheduler(self.config, optimizer, num_epochs)

    def setup_kd_training(self, student_model: nn.Module, teacher_model: Optional[nn.Module] = None,
                          teacher_path: Optional[str] = None) -> nn.Module:
        if not self.config.enable_kd_lite:
            logger.warning("KD-Lite is not enabled in config. Skipping KD setup.")
            return student_model
        from .models import ModelFactory
        self.distiller = KnowledgeDistiller(self.config, ModelFactory())
        if teacher_model is not None:
            self.distiller.teacher_model = teacher_model
            self.distiller._freeze_teacher()
        else:
            self.distiller.setup_teacher(teacher_path)
        student_model = self.distiller.setup_student(student_model)
        self.kd_loss_fn = self.distiller.loss_fn
        return student_model

    # ---------------------------------------------------------------------------- epochs
    def _zero_grad(self, optimizer) -> None:
        if self._grad_buckets is not None:
            self._grad_buckets.zero_grad()
        else:
            optimizer.zero_grad()

    def _step(self, optimizer) -> None:
        if self._grad_buckets is not None:
            self._grad_buckets.finish()
        optimizer.step()

    def _train_epoch(self, model: nn.Module, train_loader: DataLoader, criterion: nn.Module,
                     optimizer: optim.Optimizer) -> float:
        cfg = self.config
        model.train()
        if cfg.enable_kd_lite and cfg.freeze_backbone and hasattr(model, "encoder"):
            model.encoder.eval()
        dev = cfg.device
        sums = torch.zeros(4, device=dev)  # total, kd, ce, batches
        kd_active = cfg.enable_kd_lite and self.distiller is not None
        # batches of 64 and more are split into chunks of about 32 (pruning/trainer.py:167,197-216)
        chunks = max(1, cfg.batch_size // 32)
        for images, masks in train_loader:
            images, masks = images.to(dev), masks.to(dev)
            self._zero_grad(optimizer)
            if kd_active:
                # the loss's label pass beside the two forwards, on a stream of its own (the loss is computed
                # behind both: pruning/trainer.py:172-183)
                if hasattr(self.kd_loss_fn, "prepare_labels"):
                    self.kd_loss_fn.prepare_labels(masks)
                student_out = model(images)
                with torch.no_grad():
                    teacher_out = self.distiller.teacher_model(images)
                d = self.kd_loss_fn(student_out, teacher_out, masks)
                d["total"].backward()
                sums[:3] += torch.stack([d["total"].detach(), d["kd"].detach(), d["ce"].detach()])
            elif chunks > 1:
                size = images.size(0) // chunks
                for i in range(chunks):
                    lo = i * size
                    hi = lo + size if i < chunks - 1 else images.size(0)
                    if hi <= lo:
                        continue
                    loss = criterion(model(images[lo:hi]), masks[lo:hi]) / chunks
                    if self._grad_buckets is not None:
                        self._grad_buckets.hold = i < chunks - 1
                    loss.backward()
                    sums[0] += loss.detach()
            else:
                loss = criterion(model(images), masks)
                loss.backward()
                sums[0] += loss.detach()
            self._step(optimizer)
            sums[3] += 1
        sdist.all_reduce_sum_(sums)
        total, kd, ce, n = sums.tolist()  # the epoch's one host synchronisation
        if n == 0:
            return 0.0
        if kd_active:
            self.last_kd_loss, self.last_ce_loss = kd / n, ce / n
            logger.debug(f"KD Loss: {kd / n:.4f}, CE Loss: {ce / n:.4f}")
            if self.wandb_tracker:
                self.wandb_tracker.log_kd_metrics(kd_loss=kd / n, ce_loss=ce / n, temperature=cfg.kd_temperature,
                                                  alpha=cfg.kd_alpha, data_ratio=cfg.kd_data_ratio)
        return total / n

    @torch.no_grad()
    def _validate_epoch(self, model: nn.Module, val_loader: DataLoader, criterion: nn.Module) -> float:
        """Mean validation loss (cross-entropy; the CE part of KD-Lite when distilling,
        pruning/trainer.py:252-275).  The mIoU of the same pass is left in `last_val_miou`."""
        cfg = self.config
        model.eval()
        dev, c = cfg.device, cfg.num_classes
        conf = torch.zeros(c, c, dtype=torch.int64, device=dev)
        sums = torch.zeros(2, device=dev)
        kd_active = cfg.enable_kd_lite and self.distiller is not None
        for images, masks in val_loader:
            images, masks = images.to(dev), masks.to(dev).long().contiguous()
            out = model(images)
            if kd_active:
                loss = self.kd_loss_fn(out, self.distiller.teacher_model(images), masks)["ce"]
            else:
                loss = criterion(out, masks)
            sums += torch.stack([loss, torch.ones_like(loss)])
            ops.argmax_confusion_update(conf, out.contiguous(), masks, c)
        sdist.all_reduce_sum_(sums)
        sdist.all_reduce_sum_(conf)
        m = conf.double()
        hit = m.diag()
        miou = (hit / (m.sum(0) + m.sum(1) - hit + 1e-10)).mean()
        total, n = sums.tolist()
        self.last_val_miou = float(miou.item())
        return total / n if n > 0 else 0.0

    # ---------------------------------------------------------------------------- driver
    def fine_tune(self, model: nn.Module, num_epochs: int, step_info: Optional[str] = None,
                  teacher_model: Optional[nn.Module] = None) -> nn.Module:
        cfg = self.config
        self._setup_seeds()
        kind = "KD-Lite fine-tuning" if cfg.enable_kd_lite else "fine-tuning"
        logger.info(f"Starting {kind} for {num_epochs} epochs" + (f" ({step_info})" if step_info else ""))
        if cfg.enable_kd_lite and self.distiller is None:
            model = self.setup_kd_training(model, teacher_model)
        model.to(cfg.device)
        model.train()
        train_loader, val_loader = self._get_dataloaders(use_subset=cfg.enable_kd_lite)
        criterion = self._setup_criterion()
        optimizer = self._setup_optimizer(model)
        scheduler = self._setup_scheduler(optimizer, num_epochs)
        _, world = sdist.world()
        self._grad_buckets = (sdist.FlatGradBuckets([p for p in model.parameters() if p.requires_grad])
                              if world > 1 else None)
        best, stale, best_state = float("inf"), 0, None
        try:
            for epoch in range(num_epochs):
                if hasattr(getattr(train_loader, "sampler", None), "set_epoch"):
                    train_loader.sampler.set_epoch(epoch)
                train_loss = self._train_epoch(model, train_loader, criterion, optimizer)
                sdist.average_buffers_(model)
                val_loss = self._validate_epoch(model, val_loader, criterion)
                if isinstance(scheduler, ReduceLROnPlateau):
                    scheduler.step(val_loss)
                else:
                    scheduler.step()
                lr = optimizer.param_groups[0]["lr"]
                logger.info(f"Epoch {epoch + 1}/{num_epochs}: Train Loss: {train_loss:.4f}, "
                            f"Val Loss: {val_loss:.4f}, Val mIoU: {self.last_val_miou:.4f}, LR: {lr:.6f}")
                self.last_train_loss, self.last_val_loss = train_loss, val_loss
                if self.wandb_tracker:
                    self.wandb_tracker.log_training_epoch(epoch=epoch + 1, train_loss=train_loss, val_loss=val_loss,
                                                          learning_rate=lr, step_info=step_info)
                if val_loss < best:
                    # the best weights stay in HBM (a 107 MB copy) instead of a round trip through a file
                    best, stale = val_loss, 0
                    best_state = {k: v.detach().clone() for k, v in model.state_dict().items()}
                else:
                    stale += 1
                    if stale >= cfg.early_stop_patience:
                        logger.info(f"Early stopping at epoch {epoch + 1}")
                        break
        finally:
            if self._grad_buckets is not None:
                self._grad_buckets.close()
                self._grad_buckets = None
                for p in model.parameters():
                    p.grad = None
        if best_state is not None:
            model.load_state_dict(best_state)
            logger.info(f"Restored best model with val_loss: {best:.4f}")
        return model

    def cleanup(self) -> None:
        if self.temp_dir.exists():
            shutil.rmtree(self.temp_dir, ignore_errors=True)


class PruningMetrics:
    """Per-step history written to metrics.json (pruning/trainer.py:361-410)."""

    def __init__(self):
        self.history = []

    def add_step(self, step: int, train_loss: float, val_loss: float, params: int, macs: int) -> None:
        self.history.append({"step": step, "train_loss": train_loss, "val_loss": val_loss, "params": params,
                             "macs": macs, "params_million": params / 1e6, "macs_million": macs / 1e6})

    def save_history(self, filepath: str) -> None:
        with open(filepath, "w") as f:
            json.dump(self.history, f, indent=2)
        logger.info(f"Saved metrics history to {filepath}")

    def get_final_metrics(self) -> Dict:
        if not self.history:
            return {}
        last = self.history[-1]
        first = self.history[0] if len(self.history) > 1 else last
        return {"initial_params": first["params"], "final_params": last["params"],
                "params_reduction": (first["params"] - last["params"]) / first["params"],
                "initial_macs": first["macs"], "final_macs": last["macs"],
                "macs_reduction": (first["macs"] - last["macs"]) / first["macs"] if first["macs"] > 0 else 0,
                "final_val_loss": last["val_loss"]}
