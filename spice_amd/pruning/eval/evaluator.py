"""Evaluation of every pruned model found under a directory (pruning/eval/evaluator.py:40-192):
find the run directories, load each final model, run the evaluation loop (forward under autocast,
arg-max fused into the confusion histogram), print / save / cache the result, write the summary.
Plotting and the remote tracker of the reference are not part of this package.

With torch.distributed initialised (one process per GPU) the models still to be evaluated are
dealt to the ranks, every rank evaluates its share on the whole evaluation set, and rank 0
collects the result dictionaries and does all printing and writing: models are independent, so
no collective touches the data path."""
from __future__ import annotations

import logging
from typing import Any, Dict, Optional

import torch.distributed as tdist
from torch.utils.data import DataLoader

from ... import dist as sdist
from ..dataset import PruningDataHandler
from .components import CacheManager, ConfigManager, CoreEvaluator, ResultProcessor
from .config import EvaluationConfig
from .loader import PrunedModelLoader
from .metrics import SegmentationMetrics

logger = logging.getLogger(__name__)


class PrunedModelEvaluator:
    def __init__(self, config: EvaluationConfig):
        self.config = config
        self.config_manager = ConfigManager(config)
        self.device = self.config_manager.device
        self.config_manager.create_output_directories()
        self.loader = PrunedModelLoader()
        self.cache_manager = CacheManager(cache_dir=config.output_dir, enabled=self.config_manager.cache_enabled)
        self.core_evaluator = CoreEvaluator(device=self.device, num_classes=config.num_classes, img_size=config.img_size)
        self.result_processor = ResultProcessor(config.output_dir)
        if self.config_manager.visualization_enabled:
            logger.info("Plots are not produced by this package; evaluation_summary.csv holds their data")

    def evaluate_single_model(self, model_path: str, data_loader: DataLoader,
                              model_info: Optional[Dict] = None) -> Dict[str, Any]:
        model = self.loader.load_pruned_model(model_path, self.device)
        metrics = SegmentationMetrics(self.config.num_classes, device=self.device)
        return self.core_evaluator.evaluate_single_model(model=model, data_loader=data_loader,
                                                         metrics_calculator=metrics, model_info=model_info)

    def _evaluation_loader(self) -> DataLoader:
        c = self.config
        handler = PruningDataHandler(data_root=c.data_root, batch_size=c.batch_size, num_workers=c.num_workers,
                                     num_classes=c.num_classes, input_size=(c.img_size, c.img_size))
        # the whole test split on every rank: ranks divide the models, not the samples
        return handler._loader("test", False, 0.05, distribute=False)

    def evaluate_all_models(self, eval_loader: Optional[DataLoader] = None) -> None:
        cache = self.cache_manager.load_cache()
        if eval_loader is None:
            eval_loader = self._evaluation_loader()
        pruned_models = self.loader.find_pruned_models(self.config.pruned_models_dir)
        if not pruned_models:
            logger.warning(f"No pruned models found in {self.config.pruned_models_dir}")
            return
        todo = []
        for info in pruned_models:
            cached = self.cache_manager.get_cached_result(info, cache)
            if cached:
                self.result_processor.add_result(cached)
            else:
                todo.append(info)
        logger.info(f"Found {len(pruned_models)} models, {len(todo)} need evaluation")
        rank, world = sdist.world()
        if world == 1:
            for k, info in enumerate(todo, 1):
                try:
                    print(f"\nEvaluating [{k}/{len(todo)}]: {info['dir_name']}")
                    results = self.evaluate_single_model(info["path"], eval_loader, info)
                    self.result_processor.process_single_result(results, info["dir_name"])
                    self.cache_manager.store_result(info, results, cache)
                except Exception as e:  # one unreadable model does not end the run (evaluator.py:187-189)
                    logger.error(f"Failed to evaluate {info['dir_name']}: {e}")
                    continue
            self._finalize_evaluation()
            return
        mine = {}
        for k in range(rank, len(todo), world):
            info = todo[k]
            try:
                mine[k] = self.evaluate_single_model(info["path"], eval_loader, info)
            except Exception as e:
                mine[k] = f"{e}"
        shares = [None] * world
        tdist.all_gather_object(shares, mine)
        if rank != 0:
            return
        merged = {k: v for share in shares for k, v in share.items()}
        for k, info in enumerate(todo):
            print(f"\nEvaluating [{k + 1}/{len(todo)}]: {info['dir_name']}")
            if isinstance(merged[k], str):
                logger.error(f"Failed to evaluate {info['dir_name']}: {merged[k]}")
                continue
            self.result_processor.process_single_result(merged[k], info["dir_name"])
            self.cache_manager.store_result(info, merged[k], cache)
        self._finalize_evaluation()

    def _finalize_evaluation(self) -> str:
        return self.result_processor.generate_summary(save_csv=self.config.save_csv)
