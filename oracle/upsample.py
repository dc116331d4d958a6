"""CPU oracle for the bilinear up-sampling of the models' decoder and head (test infrastructure;
never imported by spice_amd/).

The models the reference builds (pruning/models.py:35-39) call torch's own
`F.interpolate(..., mode="bilinear")`; the oracle is that op on the CPU in fp32, plus a plain numpy
restatement of its arithmetic (source index, truncation, lambdas, two-level weighted sum) that the
tests hold against it.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F


def upsample_bilinear(x: torch.Tensor, size: Optional[Tuple[int, int]] = None, scale_factor: Optional[float] = None,
                      align_corners: bool = False) -> torch.Tensor:
    return F.interpolate(x.detach().float().cpu(), size=size, scale_factor=scale_factor, mode="bilinear",
                         align_corners=align_corners)


def _source(scale: np.float32, n_out: int, align_corners: bool) -> np.ndarray:
    d = np.arange(n_out, dtype=np.float32)
    if align_corners:
        return (scale * d).astype(np.float32)
    return np.maximum(scale * (d + np.float32(0.5)) - np.float32(0.5), np.float32(0)).astype(np.float32)


def upsample_bilinear_numpy(x: np.ndarray, out_hw: Tuple[int, int], align_corners: bool,
                            scale_factor: Optional[float] = None) -> np.ndarray:
    """[N,C,H,W] fp32 -> [N,C,oh,ow] fp32, every product and sum rounded to fp32 (no fused ops)."""
    x = x.astype(np.float32)
    h, w = x.shape[-2:]
    oh, ow = out_hw

    def scale(n_in, n_out):
        if align_corners:
            return np.float32(n_in - 1) / np.float32(n_out - 1) if n_out > 1 else np.float32(0)
        return np.float32(1.0) / np.float32(scale_factor) if scale_factor else np.float32(n_in) / np.float32(n_out)

    sy, sx = _source(scale(h, oh), oh, align_corners), _source(scale(w, ow), ow, align_corners)
    y0, x0 = sy.astype(np.int64), sx.astype(np.int64)
    y1, x1 = np.minimum(y0 + 1, h - 1), np.minimum(x0 + 1, w - 1)
    h1 = (sy - y0.astype(np.float32))[:, None]
    w1 = (sx - x0.astype(np.float32))[None, :]
    h0, w0 = np.float32(1) - h1, np.float32(1) - w1
    top = w0 * x[..., y0[:, None], x0[None, :]] + w1 * x[..., y0[:, None], x1[None, :]]
    bot = w0 * x[..., y1[:, None], x0[None, :]] + w1 * x[..., y1[:, None], x1[None, :]]
    return (h0 * top + h1 * bot).astype(np.float32)
