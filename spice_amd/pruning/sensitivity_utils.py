"""Taylor sensitivity scores (pruning/sensitivity_utils.py:15-144).

Per calibration batch the model's forward/backward runs on MIOpen as in the reference; the
per-output-channel reduction sum |grad|*|weight| of EVERY convolution is then ONE launch of the
HIP importance kernel accumulating into a flat fp32 buffer (the reference launches five small
ops per layer and materialises three temporaries each).  With torch.distributed initialised and
`shard=True`, ranks take calibration batches round-robin and the flat buffer is summed with a
single all-reduce.
"""
from __future__ import annotations

import logging
from typing import Dict, List

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import _native as nat
from .. import dist as sdist
from .. import ops

logger = logging.getLogger(__name__)


def _hip_row_reduce(weights, grads, out, accumulate):
    ops.channel_importance(weights, nat.IMP_TAYLOR, grads=grads, out=out, accumulate=accumulate)


def compute_taylor_scores(model: nn.Module, dataloader, device: str, max_batches: int = 100,
                          shard: bool = False, _row_reduce=_hip_row_reduce) -> Dict[nn.Module, torch.Tensor]:
    """`_row_reduce(weights, grads, out, accumulate)` is the per-channel reduction; the default and
    only product implementation is the HIP kernel (tests substitute a checker to exercise the
    batch sharding and the all-reduce on machines without a GPU)."""
    was_training = model.training
    model.eval()  # also puts every BatchNorm in eval mode
    convs = [m for m in model.modules() if isinstance(m, nn.Conv2d)]
    offsets, total = [], 0
    for m in convs:
        offsets.append(total)
        total += m.out_channels
    # scores of every convolution, then one extra slot that carries the batch count through the same
    # all-reduce (exact in fp32 far beyond any calibration set)
    flat = torch.zeros(total + 1, dtype=torch.float32, device=device)
    rank, world = 0, 1
    if shard and torch.distributed.is_available() and torch.distributed.is_initialized():
        rank, world = torch.distributed.get_rank(), torch.distributed.get_world_size()

    seen = 0
    for batch_idx, batch in enumerate(dataloader):
        if batch_idx >= max_batches:
            break
        if batch_idx % world != rank:
            continue
        if isinstance(batch, (list, tuple)):
            images = batch[0].to(device)
            targets = batch[1].to(device) if len(batch) > 1 else None
        else:
            images, targets = batch.to(device), None
        model.zero_grad(set_to_none=False)  # keeps gradient storage (and the launch plan) stable
        outputs = model(images)
        if isinstance(outputs, dict):
            outputs = outputs.get("out", list(outputs.values())[0])
        if targets is None:
            loss = outputs.abs().mean()
        elif targets.dim() == 3:
            loss = F.cross_entropy(outputs, targets.long())
        else:
            loss = F.cross_entropy(outputs, targets)
        loss.backward()
        live = [(m, o) for m, o in zip(convs, offsets) if m.weight.grad is not None]
        if live:
            if len(live) == len(convs):
                _row_reduce([m.weight.detach() for m in convs], [m.weight.grad for m in convs],
                            flat[:total], True)
            else:  # some convolutions took no part in the loss: reduce the others only
                n_live = sum(m.out_channels for m, _ in live)
                part = torch.zeros(n_live, dtype=torch.float32, device=device)
                _row_reduce([m.weight.detach() for m, _ in live], [m.weight.grad for m, _ in live], part, False)
                po = 0
                for m, o in live:
                    flat[o: o + m.out_channels] += part[po: po + m.out_channels]
                    po += m.out_channels
        seen += 1

    if world > 1:
        flat[total] = float(seen)
        sdist.all_reduce_sum_(flat)   # the one exchange of the calibration pass
        seen = int(round(float(flat[total].item())))
    flat = flat[:total] / max(seen, 1)
    model.zero_grad(set_to_none=True)
    model.train(was_training)
    return {m: flat[o: o + m.out_channels] for m, o in zip(convs, offsets)}


def get_sensitivity_report(scores: Dict[nn.Module, torch.Tensor]) -> Dict:
    report = {"num_layers": len(scores), "layer_stats": []}
    for module, s in scores.items():
        if isinstance(module, nn.Conv2d):
            report["layer_stats"].append({
                "out_channels": module.out_channels,
                "mean_sensitivity": s.mean().item(),
                "std_sensitivity": s.std().item(),
                "min_sensitivity": s.min().item(),
                "max_sensitivity": s.max().item(),
            })
    return report


def get_layer_sensitivities(scores: Dict[nn.Module, torch.Tensor]) -> Dict[nn.Module, float]:
    return {m: s.mean().item() for m, s in scores.items() if isinstance(m, nn.Conv2d)}


def identify_skip_connection_layers(model: nn.Module, model_name: str) -> List[nn.Module]:
    if any(tag in model_name for tag in ("resnet", "FPN", "UNET")):
        return [m for m in model.modules() if isinstance(m, nn.Conv2d)]
    return []
