"""Printing and saving of evaluation results (pruning/eval/components/result_processor.py:12-185):
per model `<output_dir>/<run directory>/metrics.json`, over all models `evaluation_summary.csv`
sorted by (model_name, pruning_ratio) with the list-valued columns JSON-encoded, and the console
blocks of the reference."""
from __future__ import annotations

import json
import logging
from pathlib import Path
from typing import Any, Dict, List

import numpy as np
import pandas as pd

logger = logging.getLogger(__name__)

_SUMMARY_COLUMNS = ["model_name", "strategy", "fine_tune_method", "pruning_ratio", "miou", "mean_acc", "total_params",
                    "params_percentage", "model_size_mb", "size_percentage", "gmacs", "mac_reduction"]
_PERCENT_COLUMNS = {"params_percentage": "params_%", "size_percentage": "size_%", "mac_reduction": "mac_reduction_%"}
_LIST_COLUMNS = ("per_class_iou", "per_class_acc", "confusion_matrix")


class ResultFormatter:
    @staticmethod
    def print_single_result(results: Dict[str, Any]) -> None:
        print(f"\n{'=' * 50}")
        if "model_name" in results:
            name = results["model_name"]
            if "strategy" in results:
                name += f" ({results['strategy']})"
            print(f"Model: {name}")
        if "fine_tune_method" in results:
            print(f"Fine-tune Method: {results['fine_tune_method']}")
        if "pruning_ratio" in results:
            print(f"Pruning Ratio: {results['pruning_ratio']:.2f}")
        print(f"mIoU: {results['miou']:.4f}")
        print(f"Mean Accuracy: {results['mean_acc']:.4f}")
        line = f"Parameters: {results.get('total_params', 0):,}"
        if "params_percentage" in results:
            line += f" ({results['params_percentage']:.1f}% of original)"
        print(line)
        line = f"Model Size: {results.get('model_size_mb', 0):.2f} MB"
        if "size_percentage" in results:
            line += f" ({results['size_percentage']:.1f}% of original)"
        print(line)
        if "per_class_iou" in results:
            print("\nPer-class IoU:")
            for i, iou in enumerate(results["per_class_iou"]):
                print(f"  Class {i}: {iou:.4f}")

    @staticmethod
    def format_summary_table(df: pd.DataFrame) -> pd.DataFrame:
        out = df[[c for c in _SUMMARY_COLUMNS if c in df.columns]].copy()
        for col in _PERCENT_COLUMNS:
            if col in out.columns:
                out[col] = out[col].apply(lambda x: f"{x:.1f}%")
        if "gmacs" in out.columns:
            out["gmacs"] = out["gmacs"].apply(lambda x: f"{x:.2f}")
        return out.rename(columns=_PERCENT_COLUMNS)

    @staticmethod
    def print_summary_table(df: pd.DataFrame) -> None:
        bar = "=" * 140
        print("\n" + bar)
        print("EVALUATION SUMMARY")
        print(bar)
        print(ResultFormatter.format_summary_table(df).to_string(index=False))
        print(bar)


class ResultSaver:
    def __init__(self, output_dir: str):
        self.output_dir = Path(output_dir)
        self.output_dir.mkdir(parents=True, exist_ok=True)

    def save_single_model_results(self, results: Dict[str, Any], model_name: str) -> None:
        d = self.output_dir / model_name
        d.mkdir(parents=True, exist_ok=True)
        with open(d / "metrics.json", "w") as f:
            json.dump(self._serialize_for_json(results), f, indent=2)
        logger.info(f"Saved results to {d / 'metrics.json'}")

    def save_summary_csv(self, results: List[Dict[str, Any]]) -> str:
        if not results:
            return ""
        df = pd.DataFrame(results)
        if "model_name" in df.columns and "pruning_ratio" in df.columns:
            df = df.sort_values(["model_name", "pruning_ratio"])
        df = df.copy()
        for col in _LIST_COLUMNS:
            if col in df.columns:
                df[col] = df[col].apply(self._serialize_complex_field)
        path = self.output_dir / "evaluation_summary.csv"
        df.to_csv(path, index=False)
        logger.info(f"Saved summary to {path}")
        return str(path)

    def _serialize_for_json(self, data: Dict[str, Any]) -> Dict[str, Any]:
        return {k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in data.items()}

    def _serialize_complex_field(self, value):
        if isinstance(value, np.ndarray):
            return json.dumps(value.tolist())
        if isinstance(value, list):
            return json.dumps(value)
        return value


class ResultProcessor:
    def __init__(self, output_dir: str):
        self.formatter = ResultFormatter()
        self.saver = ResultSaver(output_dir)
        self.results: List[Dict[str, Any]] = []

    def add_result(self, result: Dict[str, Any]) -> None:
        self.results.append(result)

    def process_single_result(self, result: Dict[str, Any], model_name: str) -> None:
        self.formatter.print_single_result(result)
        self.saver.save_single_model_results(result, model_name)
        self.add_result(result)

    def generate_summary(self, save_csv: bool = True) -> str:
        if not self.results:
            logger.warning("No results to summarize")
            return ""
        self.formatter.print_summary_table(pd.DataFrame(self.results))
        return self.saver.save_summary_csv(self.results) if save_csv else ""

    def get_results_dataframe(self) -> pd.DataFrame:
        return pd.DataFrame(self.results) if self.results else pd.DataFrame()
