"""Result cache of the evaluation run (pruning/eval/components/cache_manager.py:11-109): one
entry per (model file, its modification time, strategy, ratio); dropped when the evaluation code
changes.  Stored as JSON (`.eval_cache.json`) rather than the reference's pickle, so reading a
cache never executes anything from the file."""
from __future__ import annotations

import hashlib
import json
import logging
from pathlib import Path
from typing import Any, Dict, Optional

import numpy as np

logger = logging.getLogger(__name__)


def _plain(value):
    if isinstance(value, np.ndarray):
        return value.tolist()
    if isinstance(value, (np.floating, np.integer)):
        return value.item()
    if isinstance(value, dict):
        return {k: _plain(v) for k, v in value.items()}
    if isinstance(value, (list, tuple)):
        return [_plain(v) for v in value]
    return value


class CacheManager:
    def __init__(self, cache_dir: str, enabled: bool = True):
        self.cache_dir = Path(cache_dir)
        self.enabled = enabled
        self.cache_file = self.cache_dir / ".eval_cache.json"
        self.code_hash = self._compute_code_hash()
        if enabled:
            self.cache_dir.mkdir(parents=True, exist_ok=True)

    def _compute_code_hash(self) -> str:
        here = Path(__file__).parent
        stamp = "".join(str(p.stat().st_mtime) for p in (here / "core_evaluator.py", here.parent / "metrics.py")
                        if p.exists())
        return hashlib.md5(stamp.encode()).hexdigest()[:8]

    def generate_cache_key(self, model_info: Dict) -> str:
        if not self.enabled:
            return ""
        path = Path(model_info["path"])
        if not path.exists():
            return ""
        key = f"{model_info['path']}_{path.stat().st_mtime}_{model_info.get('strategy', '')}_{model_info.get('pruning_ratio', '')}"
        return hashlib.md5(key.encode()).hexdigest()

    def load_cache(self) -> Dict:
        if not self.enabled or not self.cache_file.exists():
            return {}
        try:
            with open(self.cache_file) as f:
                data = json.load(f)
        except Exception as e:
            logger.warning(f"Failed to load cache: {e}")
            return {}
        if not isinstance(data, dict) or data.get("code_hash", "") != self.code_hash:
            logger.info("Evaluation code changed or cache format outdated, clearing cache")
            return {}
        cache = data.get("results", {})
        logger.info(f"Loaded evaluation cache with {len(cache)} entries")
        return cache

    def save_cache(self, cache: Dict) -> None:
        if not self.enabled:
            return
        try:
            with open(self.cache_file, "w") as f:
                json.dump({"code_hash": self.code_hash, "results": _plain(cache)}, f)
        except Exception as e:
            logger.warning(f"Failed to save cache: {e}")

    def get_cached_result(self, model_info: Dict, cache: Dict) -> Optional[Dict[str, Any]]:
        if not self.enabled:
            return None
        key = self.generate_cache_key(model_info)
        if key and key in cache:
            logger.info(f"Using cached results for {model_info['dir_name']}")
            return cache[key]
        return None

    def store_result(self, model_info: Dict, result: Dict[str, Any], cache: Dict) -> None:
        if not self.enabled:
            return
        key = self.generate_cache_key(model_info)
        if key:
            cache[key] = result
            self.save_cache(cache)
