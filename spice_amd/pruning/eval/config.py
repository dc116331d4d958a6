"""Configuration of the pruned-model evaluation (pruning/eval/config.py:11-74): same fields and
defaults.  Plotting and the remote experiment tracker are not part of this package; their switches
are accepted and their directories created as the reference does, nothing is drawn or sent."""
from __future__ import annotations

from dataclasses import dataclass, field
from pathlib import Path
from typing import List, Optional


@dataclass
class EvaluationConfig:
    pruned_models_dir: str = "pruned_models"
    data_root: str = "data"
    output_dir: str = "pruning/eval/results"
    visualization_dir: str = "pruning/eval/outputs"

    batch_size: int = 32
    num_workers: int = 4
    img_size: int = 576
    num_classes: int = 5

    device: str = "cuda"  # cuda, or auto

    save_predictions: bool = False
    max_vis_samples: int = 10
    use_cache: bool = True

    save_csv: bool = True
    save_plots: bool = True

    generate_visualizations: bool = True
    figure_dpi: int = 300
    figure_formats: List[str] = field(default_factory=lambda: ["png", "pdf"])
    generate_advanced_plots: bool = True

    use_wandb: bool = True
    wandb_project: Optional[str] = None
    wandb_entity: Optional[str] = None
    wandb_run_name: Optional[str] = None
    wandb_tags: List[str] = field(default_factory=lambda: ["evaluation", "pruning"])

    create_3d_plots: bool = True
    create_pareto_frontier: bool = True
    create_correlation_matrix: bool = True
    create_interactive_plots: bool = True

    def __post_init__(self):
        Path(self.output_dir).mkdir(parents=True, exist_ok=True)
        if self.generate_visualizations:
            (Path(self.visualization_dir) / "advanced").mkdir(parents=True, exist_ok=True)
        if self.device == "auto":
            import torch
            self.device = "cuda" if torch.cuda.is_available() else "cpu"
