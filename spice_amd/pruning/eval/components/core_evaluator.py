"""Evaluation of one pruned model (pruning/eval/components/core_evaluator.py:12-182).

The forward runs under autocast as in the reference; when the metrics object can take logits
(`update_from_logits`, the fused arg-max + histogram kernel) the [N,H,W] prediction map of
core_evaluator.py:141 is never materialised."""
from __future__ import annotations

import logging
from typing import Any, Dict, Optional

import torch
from torch.utils.data import DataLoader

from .... import engine, nets

logger = logging.getLogger(__name__)


class ModelInfoCalculator:
    def __init__(self, img_size: int):
        self._original_model_cache: Dict[str, Dict[str, float]] = {}
        self.img_size = img_size

    def calculate_model_info(self, model: torch.nn.Module) -> Dict[str, float]:
        total_params = sum(p.numel() for p in model.parameters())
        try:
            p = next(model.parameters())
            x = torch.zeros(1, 3, self.img_size, self.img_size, device=p.device, dtype=p.dtype)
            total_macs, _ = engine.count_ops_and_params(model, x)
        except Exception as e:
            logger.warning(f"Could not calculate MACs: {e}")
            total_macs = 0
        return {"total_params": total_params, "model_size_mb": total_params * 4 / (1024 * 1024),
                "total_macs": total_macs, "gmacs": total_macs / 1e9}

    def get_original_model_info(self, model_name: str) -> Dict[str, float]:
        if model_name in self._original_model_cache:
            return self._original_model_cache[model_name]
        try:
            from ...model_configs import MODEL_CONFIGS
            if model_name in MODEL_CONFIGS:
                mc = MODEL_CONFIGS[model_name]
                full = nets.create_model(mc["architecture"], mc["encoder_name"], classes=mc["classes"],
                                         in_channels=mc["in_channels"])
                info = self.calculate_model_info(full)
                out = {"original_params": info["total_params"], "original_size_mb": info["model_size_mb"],
                       "original_macs": info["total_macs"], "original_gmacs": info["gmacs"]}
                self._original_model_cache[model_name] = out
                return out
        except Exception as e:
            logger.warning(f"Could not get original model info for {model_name}: {e}")
        return {"original_params": 0, "original_size_mb": 0, "original_macs": 0, "original_gmacs": 0}

    def calculate_percentages(self, current_info: Dict, original_info: Dict) -> Dict[str, float]:
        out = {"params_percentage": 100.0, "size_percentage": 100.0, "mac_percentage": 100.0, "mac_reduction": 0.0}
        if original_info["original_params"] > 0:
            out["params_percentage"] = current_info["total_params"] / original_info["original_params"] * 100
            out["size_percentage"] = current_info["model_size_mb"] / original_info["original_size_mb"] * 100
        if original_info.get("original_macs", 0) > 0:
            out["mac_percentage"] = current_info["total_macs"] / original_info["original_macs"] * 100
            out["mac_reduction"] = 100 - out["mac_percentage"]
        return out


class CoreEvaluator:
    def __init__(self, device: torch.device, num_classes: int, img_size: int):
        self.device = torch.device(device)
        self.num_classes = num_classes
        self.img_size = img_size
        self.model_info_calc = ModelInfoCalculator(img_size=img_size)

    def evaluate_single_model(self, model: torch.nn.Module, data_loader: DataLoader, metrics_calculator,
                              model_info: Optional[Dict] = None) -> Dict[str, Any]:
        model.eval()
        fused = hasattr(metrics_calculator, "update_from_logits")
        whole = hasattr(metrics_calculator, "update_from_model")
        with torch.no_grad():
            for batch in data_loader:
                images, masks = self._extract_batch_data(batch)
                if whole:   # forward, arg-max and histogram behind one call (the up-sampling inside the last launch)
                    with torch.amp.autocast(self.device.type, enabled=(self.device.type == "cuda")):
                        metrics_calculator.update_from_model(model, images, masks)
                    continue
                outputs = self._forward_pass(model, images)
                if fused:
                    metrics_calculator.update_from_logits(outputs, masks)
                else:
                    metrics_calculator.update(outputs.argmax(dim=1), masks)
        results = metrics_calculator.compute_metrics()
        if model_info:
            results.update(model_info)
        current = self.model_info_calc.calculate_model_info(model)
        results.update(current)
        if model_info and "model_name" in model_info:
            original = self.model_info_calc.get_original_model_info(model_info["model_name"])
            results.update(self.model_info_calc.calculate_percentages(current, original))
        return results

    def _extract_batch_data(self, batch):
        if isinstance(batch, (list, tuple)):
            return batch[0].to(self.device), batch[1].to(self.device)
        return batch["image"].to(self.device), batch["mask"].to(self.device)

    def _forward_pass(self, model: torch.nn.Module, images: torch.Tensor) -> torch.Tensor:
        with torch.amp.autocast(self.device.type, enabled=(self.device.type == "cuda")):
            outputs = model(images)
        if isinstance(outputs, dict):
            outputs = outputs.get("out", outputs.get("logits", outputs))
        return outputs
