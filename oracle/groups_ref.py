"""Hand-declared coupled-channel groups of the six model names (test infrastructure; never
imported by spice_amd/).

What torch-pruning 1.6.0's DependencyGraph.get_all_groups(root_module_types=[Conv]) yields for
smp 0.5.0's DeepLabV3Plus / Unet / FPN on the resnet50 and timm-efficientnet-b3 encoders
(pruning/model_configs.py:5-60), as consumed by
`pruner.step()` (pruning/pruner.py:495), written down from the architectures themselves:
  * a convolution's output channels are tied to the batch-norm behind it, to every depthwise
    convolution and squeeze-excite gate applied to them, and to the input channels of every
    convolution that reads them;
  * tensors that meet in a residual addition share one set of channels;
  * a convolution reading a concatenation sees each part at its offset in the concatenation.
Nothing here traces a model or imports the code under test: the tables are spelled out from layer
names and widths, so that the product's graph analysis (spice_amd/engine/depgraph.py) can be checked
against an independent statement.  torch-pruning is absent from the reference tree and this image:
these tables are the build's definition of its groups (parity unpinned).

A table maps root layer name -> Group(size, frozen, members); a member is
(layer name, kind, segments, seg_index): kind in {"out","in","bn","dw"}, `segments` the root names
whose channels make up the member's channel axis in order (one entry unless it reads a concatenation).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List, Tuple

Member = Tuple[str, str, Tuple[str, ...], int]


@dataclass
class Group:
    size: int
    frozen: bool = False
    members: List[Member] = field(default_factory=list)

    def add(self, name: str, kind: str, segments: Tuple[str, ...] = (), index: int = 0) -> "Group":
        self.members.append((name, kind, segments, index))
        return self


def _own(table: Dict[str, Group]) -> None:
    """Members declared without segments sit alone on their axis: segments = (their group's root,)."""
    for root, g in table.items():
        g.members = [(n, k, s if s else (root,), i) for n, k, s, i in g.members]


# ------------------------------------------------------------------------------------------------
# encoders: each returns (table, taps) with taps[k] = root name of the channel set of feature k
# (k = 1 .. 5 at strides 2 .. 32; feature 0 is the image itself)

def _resnet50_encoder(t: Dict[str, Group]) -> Dict[int, str]:
    e = "encoder."
    # stem: conv1 -> bn1 -> (relu = feature 1, max-pool) -> layer1.0.conv1 and the layer1.0 shortcut
    t[e + "conv1"] = (Group(64).add(e + "conv1", "out").add(e + "bn1", "bn")
                      .add(e + "layer1.0.conv1", "in").add(e + "layer1.0.downsample.0", "in"))
    taps = {1: e + "conv1"}
    blocks = {1: (3, 64), 2: (4, 128), 3: (6, 256), 4: (3, 512)}
    for layer, (n_blocks, width) in blocks.items():
        p = f"{e}layer{layer}."
        for b in range(n_blocks):
            # bottleneck interior: conv1 -> bn1 -> conv2, conv2 -> bn2 -> conv3
            t[f"{p}{b}.conv1"] = (Group(width).add(f"{p}{b}.conv1", "out").add(f"{p}{b}.bn1", "bn")
                                  .add(f"{p}{b}.conv2", "in"))
            t[f"{p}{b}.conv2"] = (Group(width).add(f"{p}{b}.conv2", "out").add(f"{p}{b}.bn2", "bn")
                                  .add(f"{p}{b}.conv3", "in"))
        # the residual stream of the stage: every conv3, the projection shortcut, every reader
        res = Group(4 * width)
        res.add(f"{p}0.conv3", "out").add(f"{p}0.bn3", "bn")
        res.add(f"{p}0.downsample.0", "out").add(f"{p}0.downsample.1", "bn")
        for b in range(1, n_blocks):
            res.add(f"{p}{b}.conv1", "in").add(f"{p}{b}.conv3", "out").add(f"{p}{b}.bn3", "bn")
        if layer < 4:
            nxt = f"{e}layer{layer + 1}.0."
            res.add(nxt + "conv1", "in").add(nxt + "downsample.0", "in")
        t[f"{p}0.conv3"] = res
        taps[layer + 1] = f"{p}0.conv3"
    return taps


# stage: (block type, repeats, output width) after width 1.2 / depth 1.4 scaling of EfficientNet-B0
_B3_STAGES = (("ds", 2, 24), ("ir", 3, 32), ("ir", 3, 48), ("ir", 5, 96), ("ir", 5, 136), ("ir", 6, 232), ("ir", 2, 384))
_B3_STEM = 40


def _se_width(block_in: int) -> int:
    return max(1, round(block_in * 0.25))


def _efficientnet_b3_encoder(t: Dict[str, Group]) -> Dict[int, str]:
    e = "encoder."
    # the channel set every block's input lives in: (root name, Group)
    stream_root, stream = e + "conv_stem", Group(_B3_STEM)
    stream.add(e + "conv_stem", "out").add(e + "bn1", "bn")
    t[stream_root] = stream
    taps = {1: stream_root}
    width_in = _B3_STEM
    for s, (kind, repeats, width_out) in enumerate(_B3_STAGES):
        for b in range(repeats):
            p = f"{e}blocks.{s}.{b}."
            first = b == 0  # the first block of a stage changes width (or stride): no identity shortcut
            if kind == "ds":
                # depthwise-separable: the INPUT channels go through depthwise conv, bn, SE gate, then conv_pw reads them
                stream.add(p + "conv_dw", "dw").add(p + "bn1", "bn")
                stream.add(p + "se.conv_reduce", "in").add(p + "se.conv_expand", "out")
                stream.add(p + "conv_pw", "in")
                t[p + "se.conv_reduce"] = (Group(_se_width(width_in)).add(p + "se.conv_reduce", "out")
                                           .add(p + "se.conv_expand", "in"))
                if first:
                    stream_root, stream = p + "conv_pw", Group(width_out)
                    t[stream_root] = stream
                stream.add(p + "conv_pw", "out").add(p + "bn2", "bn")
            else:
                # inverted residual: the stream is read by conv_pw; the expanded set is its own group
                stream.add(p + "conv_pw", "in")
                mid = 6 * width_in
                t[p + "conv_pw"] = (Group(mid).add(p + "conv_pw", "out").add(p + "bn1", "bn")
                                    .add(p + "conv_dw", "dw").add(p + "bn2", "bn")
                                    .add(p + "se.conv_reduce", "in").add(p + "se.conv_expand", "out")
                                    .add(p + "conv_pwl", "in"))
                t[p + "se.conv_reduce"] = (Group(_se_width(width_in)).add(p + "se.conv_reduce", "out")
                                           .add(p + "se.conv_expand", "in"))
                if first:
                    stream_root, stream = p + "conv_pwl", Group(width_out)
                    t[stream_root] = stream
                stream.add(p + "conv_pwl", "out").add(p + "bn3", "bn")
            width_in = width_out
        # feature taps: stage 1 (/4), stage 2 (/8), stage 4 (/16), stage 6 (/32)
        if s in (1, 2, 4, 6):
            taps[{1: 2, 2: 3, 4: 4, 6: 5}[s]] = stream_root
    return taps


_ENCODERS = {"resnet50": _resnet50_encoder, "timm-efficientnet-b3": _efficientnet_b3_encoder}


# ------------------------------------------------------------------------------------------------
# decoders: attach the readers of the encoder's features, then their own groups

def _deeplabv3plus_decoder(t: Dict[str, Group], taps: Dict[int, str], classes: int) -> None:
    """smp DeepLabV3Plus: separable ASPP (12/24/36) on feature 5, 48-channel branch on feature 2, decoder 256."""
    a = "decoder.aspp.0."
    deep = t[taps[5]]
    deep.add(a + "convs.0.0", "in")                               # 1x1 branch
    for k in (1, 2, 3):                                           # separable atrous branches: depthwise, then pointwise
        deep.add(f"{a}convs.{k}.0.0", "dw").add(f"{a}convs.{k}.0.1", "in")
    deep.add(a + "convs.4.conv", "in")                            # image-pooling branch
    t[taps[2]].add("decoder.block1.0", "in")                      # high-resolution branch
    # five 256-channel branches concatenated into the projection
    branch_roots = (a + "convs.0.0", a + "convs.1.0.1", a + "convs.2.0.1", a + "convs.3.0.1", a + "convs.4.conv")
    branch_bn = (a + "convs.0.1", a + "convs.1.1", a + "convs.2.1", a + "convs.3.1", a + "convs.4.bn")
    for i, (root, bn) in enumerate(zip(branch_roots, branch_bn)):
        t[root] = Group(256).add(root, "out").add(bn, "bn").add(a + "project.0", "in", branch_roots, i)
    # projection -> bn -> (dropout) -> separable 3x3 (depthwise decoder.aspp.1.0, pointwise .1.1)
    t[a + "project.0"] = (Group(256).add(a + "project.0", "out").add(a + "project.1", "bn")
                          .add("decoder.aspp.1.0", "dw").add("decoder.aspp.1.1", "in"))
    # fuse: cat(up-sampled ASPP output [256], high-res branch [48]) -> separable 3x3 (block2.0.0 dw, block2.0.1 pw)
    cat = ("decoder.aspp.1.1", "decoder.block1.0")
    t["decoder.aspp.1.1"] = (Group(256).add("decoder.aspp.1.1", "out").add("decoder.aspp.2", "bn")
                             .add("decoder.block2.0.0", "dw", cat, 0).add("decoder.block2.0.1", "in", cat, 0))
    t["decoder.block1.0"] = (Group(48).add("decoder.block1.0", "out").add("decoder.block1.1", "bn")
                             .add("decoder.block2.0.0", "dw", cat, 1).add("decoder.block2.0.1", "in", cat, 1))
    t["decoder.block2.0.1"] = (Group(256).add("decoder.block2.0.1", "out").add("decoder.block2.1", "bn")
                               .add("segmentation_head.0", "in"))
    t["segmentation_head.0"] = Group(classes, frozen=True).add("segmentation_head.0", "out")


def _unet_decoder(t: Dict[str, Group], taps: Dict[int, str], classes: int) -> None:
    """smp Unet, decoder (256,128,64,32,16): block k = cat(up-sampled previous, skip) -> conv1(3x3)+bn -> conv2(3x3)+bn;
    skips are features 4, 3, 2, 1; the last block has none; head 3x3."""
    dec_widths = (256, 128, 64, 32, 16)
    skips = {0: taps[4], 1: taps[3], 2: taps[2], 3: taps[1]}
    prev_root = taps[5]
    for k, w in enumerate(dec_widths):
        d = f"decoder.blocks.{k}."
        if k in skips:
            cat = (prev_root, skips[k])
            t[prev_root].add(d + "conv1.0", "in", cat, 0)
            t[skips[k]].add(d + "conv1.0", "in", cat, 1)
        else:
            t[prev_root].add(d + "conv1.0", "in")
        t[d + "conv1.0"] = Group(w).add(d + "conv1.0", "out").add(d + "conv1.1", "bn").add(d + "conv2.0", "in")
        t[d + "conv2.0"] = Group(w).add(d + "conv2.0", "out").add(d + "conv2.1", "bn")
        prev_root = d + "conv2.0"
    t[prev_root].add("segmentation_head.0", "in")
    t["segmentation_head.0"] = Group(classes, frozen=True).add("segmentation_head.0", "out")


def _fpn_decoder(t: Dict[str, Group], taps: Dict[int, str], classes: int) -> None:
    """smp FPN, pyramid 256, segmentation 128: p5 = 1x1(feature 5); p4/p3/p2 = up(previous) + 1x1(feature 4/3/2), so the four
    lateral convolutions share one set of 256 channels; four segmentation branches of 3x3 conv + GroupNorm(32) (+ x2
    up-sampling), 3 / 2 / 1 / 1 convolutions deep, summed.  GroupNorm normalises channels in blocks, so every set of
    channels it sees is frozen (never pruned); the sum ties the last convolution of every branch together."""
    d = "decoder."
    t[taps[5]].add(d + "p5", "in")
    t[taps[4]].add(d + "p4.skip_conv", "in")
    t[taps[3]].add(d + "p3.skip_conv", "in")
    t[taps[2]].add(d + "p2.skip_conv", "in")
    pyr = Group(256).add(d + "p5", "out")
    for k in (4, 3, 2):
        pyr.add(f"{d}p{k}.skip_conv", "out")
    depth = (3, 2, 1, 1)   # convolutions in branch 0..3 (branch i reads p5, p4, p3, p2)
    for i in range(4):
        pyr.add(f"{d}seg_blocks.{i}.0.block.0", "in")
    t[d + "p5"] = pyr
    last = []
    for i, n in enumerate(depth):
        for j in range(n):
            name = f"{d}seg_blocks.{i}.{j}.block.0"
            if j == n - 1:
                last.append(name)
            else:
                t[name] = Group(128, frozen=True).add(name, "out").add(f"{d}seg_blocks.{i}.{j + 1}.block.0", "in")
    merged = Group(128, frozen=True)
    for name in last:
        merged.add(name, "out")
    merged.add("segmentation_head.0", "in")
    t[last[0]] = merged
    t["segmentation_head.0"] = Group(classes, frozen=True).add("segmentation_head.0", "out")


_DECODERS = {"DeepLabV3Plus": _deeplabv3plus_decoder, "Unet": _unet_decoder, "FPN": _fpn_decoder}


def table_for(arch: str, encoder: str, classes: int) -> Dict[str, Group]:
    t: Dict[str, Group] = {}
    taps = _ENCODERS[encoder](t)
    _DECODERS[arch](t, taps, classes)
    _own(t)
    return t


def deeplabv3plus_resnet50(classes: int) -> Dict[str, Group]:
    return table_for("DeepLabV3Plus", "resnet50", classes)


def unet_efficientnet_b3(classes: int) -> Dict[str, Group]:
    return table_for("Unet", "timm-efficientnet-b3", classes)


# every (architecture, encoder) pair of pruning/model_configs.py:5-60
TABLES = {(a, e): (lambda classes, a=a, e=e: table_for(a, e, classes)) for a in _DECODERS for e in _ENCODERS}


# ------------------------------------------------------------------------------------------------
# the tables as objects the pruning oracle (oracle/prune_ref.py) walks

class RefMember:
    def __init__(self, module, name: str, kind: str):
        self.module, self.name, self.kind = module, name, kind
        self.segments: List["RefGroup"] = []
        self.seg_index = 0

    def offset(self) -> int:
        return sum(g.size for g in self.segments[: self.seg_index])


class RefGroup:
    def __init__(self, root_name: str, size: int, frozen: bool):
        self.root_name, self.size, self.initial_size, self.frozen = root_name, size, size, frozen
        self.members: List[RefMember] = []
        self.root = None


def instantiate(model, table: Dict[str, Group]) -> List[RefGroup]:
    """Bind a table to the modules of `model`; groups in module order of their root."""
    modules = dict(model.named_modules())
    order = {name: i for i, name in enumerate(modules)}
    groups = {root: RefGroup(root, g.size, g.frozen) for root, g in table.items()}
    for root, g in table.items():
        rg = groups[root]
        rg.root = modules[root]
        for name, kind, segments, index in g.members:
            m = RefMember(modules[name], name, kind)
            m.segments = [groups[s] for s in segments]
            m.seg_index = index
            rg.members.append(m)
    return sorted(groups.values(), key=lambda g: order[g.root_name])


def canonical(groups) -> Dict[str, tuple]:
    """Order-free form of a list of groups (product's or instantiated): root -> (size, frozen, members)."""
    out = {}
    for g in groups:
        out[g.root_name] = (g.size, bool(g.frozen),
                            frozenset((m.name, m.kind, tuple(s.root_name for s in m.segments), m.seg_index)
                                      for m in g.members))
    return out
