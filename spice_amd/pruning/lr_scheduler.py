"""Warm-up + cosine (or plateau) schedule from stock torch schedulers (pruning/lr_scheduler.py:18-70)."""
from torch.optim.lr_scheduler import CosineAnnealingLR, LinearLR, ReduceLROnPlateau, SequentialLR


def create_scheduler(optimizer, config, epochs: int):
    if not getattr(config, "use_cosine_annealing", True):
        return ReduceLROnPlateau(optimizer, mode="min", factor=0.5, patience=2)
    warm = min(getattr(config, "lr_warmup_epochs", 2), max(epochs - 1, 0))
    cosine = CosineAnnealingLR(optimizer, T_max=max(epochs - warm, 1),
                               eta_min=config.fine_tune_lr * getattr(config, "lr_min_factor", 0.01))
    if warm <= 0:
        return cosine
    return SequentialLR(optimizer, [LinearLR(optimizer, start_factor=0.1, end_factor=1.0, total_iters=warm),
                                    cosine], milestones=[warm])
