"""Small fixed CNNs shared by the golden-vector generator and the tests."""
from __future__ import annotations

import torch
import torch.nn as nn


class ToyNet(nn.Module):
    """Plain conv, depthwise conv, pointwise conv, BN and a 1x1 head: [N,3,H,W] -> [N,4,H,W]."""

    def __init__(self, classes: int = 4):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 8, 3, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(8)
        self.dw = nn.Conv2d(8, 8, 3, padding=1, groups=8, bias=False)
        self.pw = nn.Conv2d(8, 12, 1, bias=True)
        self.bn2 = nn.BatchNorm2d(12)
        self.head = nn.Conv2d(12, classes, 1)

    def forward(self, x):
        x = torch.relu(self.bn1(self.conv1(x)))
        x = self.dw(x)
        x = torch.relu(self.bn2(self.pw(x)))
        return self.head(x)


class ToyDictNet(ToyNet):
    """Same network returning {"out": logits} (the dict branch of the Taylor pass)."""

    def forward(self, x):
        return {"out": super().forward(x)}
