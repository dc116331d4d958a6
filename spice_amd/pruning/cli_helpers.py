"""Command-line flags -> PruningConfig.  Flag names, types and the flag -> field mapping are those
of pruning/cli_helpers.py:41-250; `--pruning-ratio` (the spelling the reference's README uses) is
accepted as an alias of `--ratio`."""
from __future__ import annotations

import argparse
import logging
from pathlib import Path

from .config import PruningConfig
from .model_configs import AVAILABLE_STRATEGIES, MODEL_CONFIGS


def _truthy(text: str) -> bool:
    return text.lower() in ("true", "1", "yes")


# (flag, config field or None, argparse keyword arguments)
_FLAGS = [
    ("--steps", "iterative_steps", dict(type=int, help="Number of iterative pruning steps")),
    ("--norm", "importance_norm", dict(type=int, choices=[1, 2], help="Norm for magnitude importance")),
    ("--fine-tune-epochs", "fine_tune_epochs", dict(type=int, help="Epochs for intermediate fine-tuning")),
    ("--final-fine-tune-epochs", "final_fine_tune_epochs", dict(type=int, help="Epochs for final fine-tuning")),
    ("--lr", "fine_tune_lr", dict(type=float, help="Learning rate for fine-tuning")),
    ("--batch-size", "batch_size", dict(type=int, help="Batch size for fine-tuning")),
    ("--early-stop-patience", "early_stop_patience", dict(type=int, help="Early stopping patience")),
    ("--data-root", "data_root", dict(type=str, help="Root directory for dataset")),
    ("--device", "device", dict(type=str, help="Device to use")),
    ("--num-workers", "num_workers", dict(type=int, help="Number of data loading workers")),
    ("--output-dir", "output_dir", dict(type=str, help="Output directory for pruned models")),
    ("--seed", "seed", dict(type=int, help="Random seed")),
    ("--calibration-batches", "calibration_batches", dict(type=int, help="Number of batches for sensitivity analysis")),
    ("--min-out-channels", "min_out_channels", dict(type=int, help="Minimum channels to keep per layer")),
    ("--flops-alpha", "flops_alpha", dict(type=float, help="Weight factor for FLOPs-based allocation")),
    ("--kd-mode", "kd_mode", dict(type=str, choices=["replace", "refine"], help="KD mode: 'replace' or 'refine'")),
    ("--kd-temperature", "kd_temperature", dict(type=float, help="Temperature for knowledge distillation")),
    ("--kd-alpha", "kd_alpha", dict(type=float, help="Weight for KD loss vs CE loss")),
    ("--kd-data-ratio", "kd_data_ratio", dict(type=float, help="Fraction of training data for KD")),
    ("--kd-refine-epochs", "kd_refine_epochs", dict(type=int, help="Additional KD refinement epochs")),
    ("--boundary-weight", "boundary_weight", dict(type=float, help="Weight multiplier for boundary pixels")),
    ("--use-timestamp", "use_timestamp", dict(type=_truthy, help="Timestamp prefix in the output directory name")),
]
_SWITCHES = [("--enable-kd", "enable_kd_lite", "Enable knowledge distillation"),
             ("--freeze-backbone", "freeze_backbone", "Freeze backbone during KD training"),
             ("--confidence-weight", "confidence_weight", "Enable confidence-based weighting")]


def create_parser() -> argparse.ArgumentParser:
    models = "\n".join(f"  - {m}" for m in MODEL_CONFIGS)
    strategies = "\n".join(f"  - {s}" for s in AVAILABLE_STRATEGIES)
    parser = argparse.ArgumentParser(
        description="Structured pruning for segmentation models",
        formatter_class=argparse.RawDescriptionHelpFormatter,
        epilog=f"Available models:\n{models}\n\nAvailable pruning strategies:\n{strategies}\n\n"
               "Example:\n  python -m spice_amd.pruning.cli --model UNET_resnet --strategy magnitude --ratio 0.5\n")
    parser.add_argument("--model", type=str, required=True, choices=list(MODEL_CONFIGS), help="Model to prune")
    parser.add_argument("--strategy", type=str, required=True, choices=AVAILABLE_STRATEGIES, help="Pruning strategy")
    parser.add_argument("--ratio", "--pruning-ratio", dest="ratio", type=float, required=True,
                        help="Target pruning ratio (0-1)")
    for flag, _field, kw in _FLAGS:
        parser.add_argument(flag, default=None, **kw)
    for flag, _field, text in _SWITCHES:
        parser.add_argument(flag, action="store_true", help=text)
    return parser


def validate_args(args: argparse.Namespace) -> None:
    if args.ratio <= 0 or args.ratio >= 1:
        raise ValueError("Pruning ratio must be between 0 and 1")
    if args.steps and args.steps <= 0:
        raise ValueError("Number of steps must be positive")
    if args.data_root and not Path(args.data_root).exists():
        logging.getLogger(__name__).warning(f"Data root directory does not exist: {args.data_root}")


def build_config_from_args(args: argparse.Namespace) -> PruningConfig:
    kwargs = {"model_name": args.model, "pruning_strategy": args.strategy, "pruning_ratio": args.ratio}
    for flag, field, _kw in _FLAGS:
        value = getattr(args, flag.lstrip("-").replace("-", "_"), None)
        if value is not None:
            kwargs[field] = value
    for flag, field, _text in _SWITCHES:
        if getattr(args, flag.lstrip("-").replace("-", "_")):
            kwargs[field] = True
    return PruningConfig(**kwargs)
