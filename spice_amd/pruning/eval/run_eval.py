"""`python -m spice_amd.pruning.eval.run_eval [--pruned-dir D] [--single-model NAME] ...`
(pruning/eval/run_eval.py:25-245: same flags and defaults).  The plotting and tracker flags are
accepted so that existing command lines keep working; this package produces the numbers
(`metrics.json`, `evaluation_summary.csv`) and leaves plots to whatever reads those files.
Under `torch.distributed.run` the models are dealt to one process per GPU."""
from __future__ import annotations

import argparse
from pathlib import Path

from ... import dist as sdist
from .config import EvaluationConfig
from .evaluator import PrunedModelEvaluator

DEFAULT_PROJECT = "epic-v2"


def create_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Evaluate pruned segmentation models")
    p.add_argument("--pruned-dir", type=str, default="pruned_models", help="Directory containing pruned models")
    p.add_argument("--data-root", type=str, default="data", help="Root directory of dataset")
    p.add_argument("--output-dir", type=str, default="pruning/eval/results", help="Directory to save evaluation results")
    p.add_argument("--batch-size", type=int, default=32, help="Batch size for evaluation")
    p.add_argument("--device", type=str, default="auto", choices=["cuda", "cpu", "auto"], help="Device to use for evaluation")
    p.add_argument("--single-model", type=str, help="Evaluate a single model by specifying its directory name")
    p.add_argument("--no-visualize", dest="visualize", action="store_false", default=True,
                   help="Disable visualization generation (enabled by default)")
    p.add_argument("--vis-dir", type=str, default="pruning/eval/outputs", help="Directory to save visualizations")
    p.add_argument("--figure-dpi", type=int, default=600, help="DPI for saved figures")
    p.add_argument("--figure-formats", nargs="+", default=["png", "pdf"], help="Formats to save figures in")
    p.add_argument("--advanced-plots", action="store_true", default=True, help="Generate advanced visualizations")
    p.add_argument("--no-wandb", dest="wandb", action="store_false", default=True,
                   help="Disable WandB tracking (enabled by default)")
    p.add_argument("--wandb-project", type=str, default=DEFAULT_PROJECT, help=f"WandB project name (default: {DEFAULT_PROJECT})")
    p.add_argument("--wandb-entity", type=str, default=None, help="WandB entity (team/user)")
    p.add_argument("--wandb-run-name", type=str, default=None, help="WandB run name")
    p.add_argument("--wandb-tags", nargs="+", default=["evaluation", "pruning"], help="WandB tags for the run")
    p.add_argument("--no-eval", action="store_true", help="Skip evaluation, only generate visualizations from existing CSV")
    p.add_argument("--clear-cache", action="store_true", help="Clear evaluation cache before running")
    return p


def parse_args(argv=None) -> argparse.Namespace:
    return create_parser().parse_args(argv)


def config_from_args(args: argparse.Namespace) -> EvaluationConfig:
    return EvaluationConfig(pruned_models_dir=args.pruned_dir, data_root=args.data_root, output_dir=args.output_dir,
                            batch_size=args.batch_size, device=args.device, generate_visualizations=args.visualize,
                            visualization_dir=args.vis_dir, figure_dpi=args.figure_dpi, figure_formats=args.figure_formats,
                            generate_advanced_plots=args.advanced_plots, use_wandb=args.wandb,
                            wandb_project=args.wandb_project, wandb_entity=args.wandb_entity,
                            wandb_run_name=args.wandb_run_name, wandb_tags=args.wandb_tags)


def main(argv=None) -> None:
    args = parse_args(argv)
    config = config_from_args(args)
    rank, world, local = sdist.init_from_env()
    if world > 1 and config.device == "cuda":
        config.device = f"cuda:{local}"
    if args.clear_cache and rank == 0:
        cache_file = Path(config.output_dir) / ".eval_cache.json"
        if cache_file.exists():
            cache_file.unlink()
            print("Evaluation cache cleared")
    sdist.barrier()
    if args.no_eval:
        csv_file = Path(config.output_dir) / "evaluation_summary.csv"
        if not csv_file.exists():
            print(f"Error: CSV file not found: {csv_file}")
            print("Run evaluation first to generate the CSV file")
            return
        print(f"Skipping evaluation; {csv_file} holds the data for plots (none are drawn by this package)")
        return
    evaluator = PrunedModelEvaluator(config)
    if args.single_model:
        model_path = Path(args.pruned_dir) / args.single_model / "final_model.pth"
        if not model_path.exists():
            print(f"Model not found: {model_path}")
            return
        if rank == 0:
            results = evaluator.evaluate_single_model(str(model_path), evaluator._evaluation_loader(),
                                                      {"model_name": args.single_model})
            evaluator.result_processor.process_single_result(results, args.single_model)
    else:
        evaluator.evaluate_all_models()
    if rank == 0:
        print("\nEvaluation completed!")


if __name__ == "__main__":
    main()
