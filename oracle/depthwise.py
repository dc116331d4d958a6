"""CPU oracle for the depthwise 3x3 convolution of the separable decoder blocks (test
infrastructure; never imported by spice_amd/): torch's conv2d on the CPU in fp64 on the fp16 values
(products and sums exact), rounded to fp16 ONCE (numpy's fp64 -> fp16 conversion; torch's goes through fp32 and rounds
twice).  torch's device result for the same call depends on the solver MIOpen picks and sits
within one fp16 step of this."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def depthwise_conv3x3(x: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    x64 = x.detach().cpu().to(torch.float16).double()
    w64 = weight.detach().cpu().to(torch.float16).double()
    return torch.from_numpy(F.conv2d(x64, w64, None, 1, 1, 1, x.shape[1]).numpy().astype(np.float16))
