"""The activation tape on the CPU: for all six model names a forward replayed from any unit returns the logits of
the plain forward bit for bit, units are named in execution order, and a channel group's first unit is the first
one that runs any of its modules."""
import pytest
import torch

from spice_amd import nets
from spice_amd.engine.depgraph import build_groups
from spice_amd.nets.prefix import PrefixTape

MODELS = [("DeepLabV3Plus", "resnet50"), ("Unet", "timm-efficientnet-b3"), ("FPN", "resnet50"), ("Unet", "resnet50"),
          ("DeepLabV3Plus", "timm-efficientnet-b3"), ("FPN", "timm-efficientnet-b3")]


@pytest.mark.parametrize("arch,enc", MODELS)
def test_replay_from_any_unit_equals_the_plain_forward(arch, enc):
    torch.manual_seed(0)
    m = nets.create_model(arch, enc, 5).eval()
    x = torch.randn(1, 3, 64, 64)
    with torch.no_grad():
        plain = m(x)
        tape = PrefixTape(m.prefix_units())
        assert len(tape) >= 17
        m.attach_tape(tape.recorder())
        recorded = m(x)
        assert tape.complete() and tape.nbytes() > 0
        for first in sorted({0, 1, len(tape) // 3, len(tape) // 2, len(tape) - 1, len(tape)}):
            p = tape.replay(first)
            m.attach_tape(p)
            y = m(x)
            assert torch.equal(y, plain) and p.reused == first
        m.attach_tape(None)
        assert torch.equal(recorded, plain) and torch.equal(m(x), plain)


def test_replay_really_skips_the_units_in_front():
    """Poisoned weights in front of the first unit do not reach the logits; poisoned weights behind it do."""
    torch.manual_seed(1)
    m = nets.create_model("DeepLabV3Plus", "resnet50", 5).eval()
    x = torch.randn(1, 3, 64, 64)
    with torch.no_grad():
        tape = PrefixTape(m.prefix_units())
        m.attach_tape(tape.recorder())
        plain = m(x)
        k = tape.index["encoder.layer3.0"]
        m.encoder.layer2[1].conv1.weight.mul_(0.0)          # in front of unit k
        m.attach_tape(tape.replay(k))
        assert torch.equal(m(x), plain)
        m.encoder.layer3[2].conv1.weight.mul_(0.0)          # behind it
        assert not torch.equal(m(x), plain)
        m.attach_tape(None)


@pytest.mark.parametrize("arch,enc", MODELS[:2])
def test_first_unit_of_every_group(arch, enc):
    m = nets.create_model(arch, enc, 5).eval()
    tape = PrefixTape(m.prefix_units())
    names = [n for n, _ in tape.units]
    assert len(set(names)) == len(names)
    order = {n: i for i, (n, _) in enumerate(m.named_modules())}
    # units are listed in execution order, which for these networks is module order of their first module
    firsts = [min(order[p] for p in prefixes) for _, prefixes in tape.units]
    assert firsts == sorted(firsts)
    for g in build_groups(m):
        k = tape.first_touched(mm.name for mm in g.members)
        touched = sorted(tape.unit_of(mm.name) for mm in g.members)
        assert k == touched[0]
        # no member of the group lives in a unit in front of k
        assert all(tape.unit_of(mm.name) >= k for mm in g.members)
    assert tape.unit_of("segmentation_head.0") == len(tape)
    assert tape.first_touched([]) == len(tape)


def test_models_stay_traceable_and_hook_aware():
    """The channel-coupling analysis traces the forward symbolically (no tape attached: the plain forward); a hook on
    a container the forward would otherwise look inside fires."""
    m = nets.create_model("DeepLabV3Plus", "resnet50", 5).eval()
    assert len(build_groups(m)) > 40
    seen = []
    hooks = [m.decoder.aspp.register_forward_hook(lambda *_: seen.append("aspp")),
             m.decoder.aspp[0].project.register_forward_hook(lambda *_: seen.append("project")),
             m.encoder.layer1[0].downsample.register_forward_hook(lambda *_: seen.append("downsample"))]
    x = torch.randn(1, 3, 64, 64)
    with torch.no_grad():
        hooked = m(x)
    for h in hooks:
        h.remove()
    with torch.no_grad():
        plain = m(x)
    assert sorted(set(seen)) == ["aspp", "downsample", "project"]
    assert torch.allclose(hooked, plain, rtol=1e-5, atol=1e-6)
