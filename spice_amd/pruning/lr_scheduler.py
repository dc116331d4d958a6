"""Fine-tuning learning-rate schedule from stock torch schedulers
(pruning/lr_scheduler.py:18-70): linear warm-up into cosine decay, plain cosine decay, or
reduce-on-plateau; the floor is fine_tune_lr * lr_min_factor in every case."""
import logging

from torch.optim.lr_scheduler import CosineAnnealingLR, LinearLR, ReduceLROnPlateau, SequentialLR

logger = logging.getLogger(__name__)


def create_scheduler(config, optimizer, num_epochs):
    floor = config.fine_tune_lr * config.lr_min_factor
    if not config.use_cosine_annealing:
        logger.info("Using ReduceLROnPlateau")
        return ReduceLROnPlateau(optimizer, mode="min", factor=0.5, patience=3, min_lr=floor)
    if config.lr_warmup_epochs > 0 and num_epochs > config.lr_warmup_epochs:
        # warm-up never takes more than a quarter of the run (lr_scheduler.py:35)
        warm = min(config.lr_warmup_epochs, num_epochs // 4)
        sched = SequentialLR(
            optimizer,
            schedulers=[LinearLR(optimizer, start_factor=0.1, end_factor=1.0, total_iters=warm),
                        CosineAnnealingLR(optimizer, T_max=num_epochs - warm, eta_min=floor)],
            milestones=[warm])
        sched.warmup_epochs = warm
        logger.info(f"Using linear warm-up ({warm} epochs) + cosine decay (min_lr={floor:.6f})")
        return sched
    logger.info(f"Using cosine decay without warm-up (min_lr={floor:.6f})")
    return CosineAnnealingLR(optimizer, T_max=num_epochs, eta_min=floor)
