"""Portable synthetic inputs for the hot path: a counter-based generator (splitmix64 of
(seed, stream, element index)) evaluated in numpy, so the same arrays are produced on any
machine and torch version.  Shapes follow the benchmark configurations: fp32 NCHW logits,
int64 label maps (uniform or blocky, with a share of ignore labels), conv weight/grad pairs.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def bits(seed: int, stream: int, n: int) -> np.ndarray:
    """n 64-bit words for (seed, stream)."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.array([seed], dtype=np.uint64) * np.uint64(0x100000001B3)
                           + np.uint64(stream))[0]
        return _splitmix64(np.arange(n, dtype=np.uint64) + base)


def uniform(seed: int, stream: int, n: int) -> np.ndarray:
    """float64 in [0, 1)."""
    return (bits(seed, stream, n) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def normal(seed: int, stream: int, n: int, dtype=np.float32) -> np.ndarray:
    """Standard normal by Box-Muller on two uniform streams."""
    u1 = uniform(seed, 2 * stream + 1_000_003, n)
    u2 = uniform(seed, 2 * stream + 1_000_004, n)
    r = np.sqrt(-2.0 * np.log1p(-u1))
    return (r * np.cos(2.0 * np.pi * u2)).astype(dtype)


def randint(seed: int, stream: int, n: int, hi: int) -> np.ndarray:
    return (bits(seed, stream, n) % np.uint64(hi)).astype(np.int64)


def logits_pair(seed: int, n: int, c: int, h: int, w: int) -> Tuple[np.ndarray, np.ndarray]:
    """student ~ 3 N(0,1); teacher = student + N(0,1); fp32 [n,c,h,w]."""
    size = n * c * h * w
    s = 3.0 * normal(seed, 1, size)
    t = s + normal(seed, 2, size)
    return s.reshape(n, c, h, w).astype(np.float32), t.reshape(n, c, h, w).astype(np.float32)


def label_map(seed: int, n: int, h: int, w: int, c: int, kind: str = "uniform",
              ignore_share: float = 0.05, ignore_index: int = 255) -> np.ndarray:
    """int64 [n,h,w]: 'uniform' in [0,c) or 'blocky' 64x64 tiles (same-bin contention)."""
    if kind == "uniform":
        lab = randint(seed, 3, n * h * w, c).reshape(n, h, w)
    elif kind == "blocky":
        yy, xx = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        lab = np.broadcast_to(((yy // 64) * 7 + (xx // 64) * 3) % c, (n, h, w)).astype(np.int64).copy()
    else:
        raise ValueError(kind)
    if ignore_share > 0:
        drop = uniform(seed, 4, n * h * w).reshape(n, h, w) < ignore_share
        lab[drop] = ignore_index
    return lab


def predictions_from(labels: np.ndarray, seed: int, c: int, wrong_share: float = 0.2) -> np.ndarray:
    """Labels with a share re-drawn uniformly; ignore labels get a uniform class too."""
    flat = labels.reshape(-1)
    pred = flat.copy()
    redraw = (uniform(seed, 5, flat.size) < wrong_share) | (flat < 0) | (flat >= c)
    pred[redraw] = randint(seed, 6, flat.size, c)[redraw]
    return pred.reshape(labels.shape)


def conv_pairs(seed: int, shapes: Sequence[Tuple[int, int, int, int]]) -> List[Tuple[np.ndarray, np.ndarray]]:
    """(weight, grad) per [out, in, kh, kw]: w ~ N(0, 2/fan_in), g ~ N(0, 1e-3)."""
    out = []
    for k, (o, i, kh, kw) in enumerate(shapes):
        nel = o * i * kh * kw
        wgt = normal(seed, 100 + 2 * k, nel) * np.float32(np.sqrt(2.0 / (i * kh * kw)))
        grd = normal(seed, 101 + 2 * k, nel) * np.float32(1e-3)
        out.append((wgt.reshape(o, i, kh, kw), grd.reshape(o, i, kh, kw)))
    return out


# (out, in/groups, kh, kw) of the convolutions of a DeepLabV3+ with a ResNet-50 encoder
# (output stride 16, separable ASPP and decoder), 21-class head.
def deeplab_resnet50_conv_shapes(classes: int = 21) -> List[Tuple[int, int, int, int]]:
    shapes = [(64, 3, 7, 7)]
    inplanes = 64
    for planes, blocks in ((64, 3), (128, 4), (256, 6), (512, 3)):
        for b in range(blocks):
            shapes += [(planes, inplanes, 1, 1), (planes, planes, 3, 3), (planes * 4, planes, 1, 1)]
            if b == 0:
                shapes.append((planes * 4, inplanes, 1, 1))
            inplanes = planes * 4
    shapes.append((256, 2048, 1, 1))                       # ASPP 1x1
    for _ in range(3):                                     # separable atrous branches
        shapes += [(2048, 1, 3, 3), (256, 2048, 1, 1)]
    shapes.append((256, 2048, 1, 1))                       # image pooling branch
    shapes.append((256, 1280, 1, 1))                       # ASPP projection
    shapes += [(256, 1, 3, 3), (256, 256, 1, 1)]           # separable 3x3 after ASPP
    shapes.append((48, 256, 1, 1))                         # high-resolution branch
    shapes += [(304, 1, 3, 3), (256, 304, 1, 1)]           # fuse
    shapes.append((classes, 256, 1, 1))                    # head
    return shapes
