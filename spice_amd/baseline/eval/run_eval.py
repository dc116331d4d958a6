"""`python -m spice_amd.baseline.eval.run_eval`: evaluate every un-pruned model whose state dict
exists under `<root>/checkpoints/fullsize/<name>/<name>.pt` and print the mIoU table
(baseline/eval/run_eval.py:31-115; the remote tracker's calls are not made).  Under
`torch.distributed.run` the evaluation batches of each model are dealt to the ranks and rank 0 prints."""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Iterable, List, Optional

import torch

from ... import dist as sdist
from . import evaluate_model
from .config import BATCH_SIZE, IMG_SIZE, MODEL_CONFIGS, NUM_CLASSES, NUM_WORKERS
from .model_factory import get_model_by_name


def main(root: Optional[str] = None, eval_loader: Optional[Iterable] = None,
         models: Optional[List[str]] = None) -> List[Dict]:
    root = Path(root) if root is not None else Path.cwd()
    rank, world, local = sdist.init_from_env()
    if not torch.cuda.is_available():
        raise RuntimeError("the evaluation runs on a HIP device; none is visible")
    device = torch.device("cuda", local if world > 1 else 0)
    if eval_loader is None:
        from ...pruning.dataset import PruningDataHandler
        handler = PruningDataHandler(data_root=str(root / "data"), batch_size=BATCH_SIZE, num_workers=NUM_WORKERS,
                                     num_classes=NUM_CLASSES, input_size=(IMG_SIZE, IMG_SIZE))
        eval_loader = handler._loader("test", False, 0.05, distribute=False)  # inference() deals the batches itself
    results = []
    for model_name in (models if models is not None else list(MODEL_CONFIGS)):
        model, checkpoint_path = get_model_by_name(model_name)
        full_path = root / checkpoint_path
        if full_path.exists():
            metrics = evaluate_model(model, str(full_path), eval_loader, device, NUM_CLASSES)
            if metrics:
                results.append({"model": model_name, **metrics})
    if rank == 0:
        print("\n" + "=" * 50)
        print(f"{'Model':<30} {'mIoU':<10} {'Mean Acc':<10}")
        print("-" * 50)
        for r in results:
            print(f"{r['model']:<30} {r['miou']:<10.4f} {r['mean_acc']:<10.4f}")
        print("=" * 50)
    return results


if __name__ == "__main__":
    main()
