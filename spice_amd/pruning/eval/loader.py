"""Locating and loading the artefacts of pruning runs (pruning/eval/loader.py:14-181).

A run directory is named [<YYYYmmdd>_<HHMMSS>_]<model>_<strategy>_<ratio>[_kd|_kd_refine] and holds
final_model.pth plus intermediate/<model>_step_<i>_of_<n>.pth."""
from __future__ import annotations

import logging
import re
from pathlib import Path
from typing import Dict, List, Optional

import torch
import torch.nn as nn

logger = logging.getLogger(__name__)

# strategies whose names contain an underscore take two name fields (loader.py:101-114)
_TWO_WORD_STRATEGIES = (("taylor", "weight"), ("magnitude", "taylor"))
_STEP_FILE = re.compile(r"^(?P<model>.*)_step_(?P<step>[^_]+)_of_(?P<total>[^_]+)$")


def parse_run_directory_name(dir_name: str) -> Optional[Dict]:
    """Fields of a run directory name, or None when the name does not describe a run."""
    fields = dir_name.split("_")
    if len(fields) > 2 and fields[0].isdigit() and fields[1].isdigit():
        fields = fields[2:]  # time stamp
    method = "Standard"
    if fields[-1] == "kd":
        method, fields = "KD-Lite", fields[:-1]
    elif fields[-2:] == ["kd", "refine"]:
        method, fields = "Standard+KD", fields[:-2]
    try:
        ratio = float(fields[-1])
    except (ValueError, IndexError):
        return None
    if len(fields) < 3:
        return None
    words = 2 if len(fields) >= 4 and tuple(fields[-3:-1]) in _TWO_WORD_STRATEGIES else 1
    return {"model_name": "_".join(fields[:-1 - words]), "strategy": "_".join(fields[-1 - words:-1]),
            "pruning_ratio": ratio, "dir_name": dir_name, "fine_tune_method": method}


class PrunedModelLoader:
    @staticmethod
    def load_pruned_model(model_path: str, device: torch.device = torch.device("cpu")) -> nn.Module:
        """Whole-module file written by a pruning run -> module in eval mode on `device`.
        Unpickling a module runs code from the file: load only files you wrote."""
        if not Path(model_path).exists():
            raise FileNotFoundError(f"Model file not found: {model_path}")
        try:
            model = torch.load(model_path, map_location=device, weights_only=False)
        except Exception as e:
            logger.error(f"Failed to load model from {model_path}: {e}")
            raise
        model.to(device)
        model.eval()
        logger.info(f"Successfully loaded pruned model from {model_path}")
        return model

    @staticmethod
    def find_pruned_models(pruned_dir: str) -> List[Dict]:
        """One dict per run directory that holds a final_model.pth, ordered by (model, ratio)."""
        root = Path(pruned_dir)
        if not root.exists():
            logger.warning(f"Pruned models directory not found: {root}")
            return []
        found = []
        for d in root.iterdir():
            final = d / "final_model.pth"
            if not d.is_dir() or not final.exists():
                continue
            info = parse_run_directory_name(d.name)
            if info is None:
                continue
            found.append({"path": str(final), **info})
            logger.info(f"Found pruned model: {d.name} (model={info['model_name']}, "
                        f"strategy={info['strategy']}, fine-tune={info['fine_tune_method']})")
        found.sort(key=lambda r: (r["model_name"], r["pruning_ratio"]))
        return found

    @staticmethod
    def load_intermediate_models(model_dir: str) -> List[Dict]:
        """{"path", "step", "total_steps"} per intermediate checkpoint, in file-name order."""
        inter = Path(model_dir) / "intermediate"
        if not inter.exists():
            logger.warning(f"No intermediate models found in {model_dir}")
            return []
        rows = []
        for f in sorted(inter.glob("*.pth")):
            pieces = f.stem.split("_step_")
            if len(pieces) != 2:
                continue
            step_total = pieces[1].split("_of_")
            if len(step_total) != 2:
                continue
            rows.append({"path": str(f), "step": int(step_total[0]), "total_steps": int(step_total[1])})
        return rows
