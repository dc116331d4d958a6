"""CPU oracle for the fused inference batch-norm + residual add + ReLU (test infrastructure; never
imported by spice_amd/): the three torch ops of the models' forward (the models of
pruning/models.py:35-39), run one after the other on the CPU in fp32, each result rounded to the
tensor's dtype as the separate device kernels store it."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F


def bn_act(x: torch.Tensor, mean: torch.Tensor, var: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float,
           residual: Optional[torch.Tensor] = None, relu: bool = False) -> torch.Tensor:
    dtype = x.dtype
    y = F.batch_norm(x.detach().float().cpu(), mean.float().cpu(), var.float().cpu(), weight.float().cpu(),
                     bias.float().cpu(), False, 0.0, eps).to(dtype)
    if residual is not None:
        y = (y.float() + residual.detach().float().cpu()).to(dtype)
    if relu:
        y = torch.relu(y)
    return y


def bn_act_dual(x: torch.Tensor, bn, residual: torch.Tensor, res_bn, relu: bool = False) -> torch.Tensor:
    """relu?(bn(x) + res_bn(residual)) as four separate ops, each result rounded to the tensor's dtype:
    the tail of a residual block whose shortcut is a projection followed by its own batch-norm.
    `bn`, `res_bn`: (mean, var, weight, bias, eps) tuples."""
    dtype = x.dtype
    r = bn_act(residual, *res_bn)
    return bn_act(x, *bn, residual=r.to(dtype), relu=relu)


def bn_act_maxpool(x: torch.Tensor, mean: torch.Tensor, var: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor,
                   eps: float, relu: bool = True):
    """(relu?(bn(x)), max_pool2d(that, 3, stride 2, padding 1)): the stem of the ResNet encoders behind their first
    convolution, as its separate ops."""
    full = bn_act(x, mean, var, weight, bias, eps, relu=relu)
    return full, F.max_pool2d(full.float(), 3, 2, 1).to(full.dtype)
