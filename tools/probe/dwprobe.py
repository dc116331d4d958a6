"""Depthwise 3x3: the library kernel and torch's device result against exact sums rounded once
(numpy fp64 -> fp16) and rounded twice (fp64 -> fp32 -> fp16)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import torch.nn.functional as F
from spice_amd import ops
dev = torch.device("cuda:0")
for shape in ((2, 304, 36, 40), (8, 304, 128, 128), (8, 256, 32, 32)):
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g).half()
    w = (0.5 * torch.randn(shape[1], 1, 3, 3, generator=g)).half()
    exact = F.conv2d(x.double(), w.double(), None, 1, 1, 1, shape[1]).numpy()
    once = torch.from_numpy(exact.astype(np.float16))
    twice = torch.from_numpy(exact.astype(np.float32).astype(np.float16))
    mine = ops.depthwise_conv3x3(x.to(dev), w.to(dev)).cpu()
    there = F.conv2d(x.to(dev), w.to(dev), None, 1, 1, 1, shape[1]).cpu()
    f32 = F.conv2d(x.float().to(dev), w.float().to(dev), None, 1, 1, 1, shape[1]).cpu()
    n = once.numel()
    print(shape, "elements", n, "| once vs twice", int((once != twice).sum()), "| mine vs once", int((mine != once).sum()),
          "| mine vs twice", int((mine != twice).sum()), "| torch-device vs once", int((there != once).sum()),
          "| torch-device vs twice", int((there != twice).sum()), "| torch-device vs mine", int((there != mine).sum()),
          "| fp32 device conv rounded vs once", int((f32.half() != once).sum()), flush=True)
