"""Building blocks shared by the encoders and decoders whose inference forward on a HIP device
is a launch of the library instead of a chain of torch kernels."""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops


class BatchNormAct2d(nn.BatchNorm2d):
    """nn.BatchNorm2d with what follows it in a residual network folded into the module:
    y = relu?( bn(x) [+ residual] ).

    In eval mode under no_grad on a HIP device that is ONE kernel launch (x and the residual read
    once, y written once) instead of MIOpen's batch-norm, torch's add and torch's clamp; the kernel
    rounds every intermediate where those would have stored it, so the values are the same.  In
    every other situation (training, gradients wanted, CPU, fp16 planes with an odd element count,
    mixed dtypes) the torch ops run one after the other."""

    def __init__(self, num_features: int, relu: bool = False, **kwargs):
        super().__init__(num_features, **kwargs)
        self.relu_after = relu

    def forward(self, x: torch.Tensor, residual: Optional[torch.Tensor] = None) -> torch.Tensor:
        if not (self.training or torch.is_grad_enabled()) and x.is_cuda and self.track_running_stats and self.affine:
            if ops.bn_act_supported(x) and (residual is None or (residual.dtype == x.dtype and residual.shape == x.shape
                                                                  and residual.is_contiguous())):
                return ops.bn_act(x, self.running_mean, self.running_var, self.weight, self.bias, self.eps,
                                  residual=residual, relu=self.relu_after)
        y = super().forward(x)
        if residual is not None:
            y = y + residual
        return F.relu(y) if self.relu_after else y

    def extra_repr(self) -> str:
        return super().extra_repr() + f", relu={self.relu_after}"
