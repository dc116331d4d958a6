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


def depthwise_conv3x3_fp32_chain(x: torch.Tensor, weight: torch.Tensor, dilation: int) -> torch.Tensor:
    """The dilated case as torch's own depthwise device kernel computes it: the nine products
    (exact in fp32: two fp16 factors) added to an fp32 sum in tap order, rows first, one rounding
    to fp16 at the end.  Padding = dilation."""
    xn = x.detach().cpu().to(torch.float16).numpy().astype(np.float32)
    wn = weight.detach().cpu().to(torch.float16).numpy().astype(np.float32)
    n, c, h, w = xn.shape
    d = int(dilation)
    pad = np.zeros((n, c, h + 2 * d, w + 2 * d), dtype=np.float32)
    pad[:, :, d:d + h, d:d + w] = xn
    acc = np.zeros((n, c, h, w), dtype=np.float32)
    for r in range(3):
        for j in range(3):
            acc = (acc + wn[None, :, 0, r, j, None, None] * pad[:, :, r * d:r * d + h, j * d:j * d + w]).astype(np.float32)
    return torch.from_numpy(acc.astype(np.float16))
