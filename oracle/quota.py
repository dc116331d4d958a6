"""CPU oracle for the Taylor -> per-layer pruning-quota allocator (test infrastructure; never
imported by spice_amd/).  Restates LayerQuotaCalculator, pruning/pruner.py:26-209, on plain
lists; pinned by tests/golden/quota_*.npz generated from the reference class itself.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import torch


def conv_flops(out_channels: int, in_channels: int, kh: int, kw: int) -> int:
    """_estimate_flops_contribution, pruning/pruner.py:87-95."""
    return out_channels * in_channels * kh * kw


def pruning_weights(sensitivities: Sequence[float], flops: Sequence[int], alpha: float) -> torch.Tensor:
    """_compute_pruning_weights, pruning/pruner.py:97-117 (fp32 tensor arithmetic)."""
    s = torch.tensor(list(sensitivities))
    s = s / (s.mean() + 1e-8)
    inv = 1.0 / (s + 1e-6)
    if alpha > 0:
        f = torch.tensor(list(flops), dtype=torch.float32)
        f = f / f.max()
        w = inv * (f ** alpha)
    else:
        w = inv
    return w / w.sum()


def allocate(channels: Sequence[int], weights: torch.Tensor, global_ratio: float,
             min_channels: int) -> Tuple[List[int], List[float]]:
    """_allocate_layer_quotas, pruning/pruner.py:119-209: at most 10 rounds; inside a round the
    remaining budget shrinks as each layer takes int(remaining * share), capped at
    channels - min_channels.  Returns (channels pruned per layer, quota per layer)."""
    total = sum(channels)
    remaining = int(total * global_ratio)
    taken = [0] * len(channels)
    for _ in range(10):
        if remaining <= 0:
            break
        live = [i for i, ch in enumerate(channels) if taken[i] < ch - min_channels]
        if not live:
            break
        share = [weights[i].item() for i in live]
        tot = sum(share)
        share = [1.0 / len(share)] * len(share) if tot == 0 else [x / tot for x in share]
        for i, sh in zip(live, share):
            want = int(remaining * sh)
            room = channels[i] - min_channels - taken[i]
            give = min(want, room)
            taken[i] += give
            remaining -= give
    quotas = [taken[i] / ch if ch > 0 else 0 for i, ch in enumerate(channels)]
    return taken, quotas


def layer_quotas(sensitivities: Sequence[float], flops: Sequence[int], channels: Sequence[int],
                 global_ratio: float, alpha: float = 0.7, min_channels: int = 8):
    """compute_layer_quotas, pruning/pruner.py:29-57, for layers already filtered to
    out_channels > min_channels (pruning/pruner.py:66-71)."""
    if not channels:
        return [], []
    return allocate(channels, pruning_weights(sensitivities, flops, alpha), global_ratio, min_channels)
