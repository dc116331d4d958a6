"""Do two-lane sweeps repeat the one-lane tables?  Candidates a .. b-1 of a sweep at [8,3,512,512] evaluated with one
lane (twice: the reference and its own repeat), then twelve times with two lanes; prints the candidates whose confusion
matrices differ and how often.  Usage: python3 tools/probe/lanes_repeat.py a b [whole|reuse] [architecture encoder]
(HOLD_DST=1: every candidate's sliced tensors stay allocated, so that no address is handed out twice)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch import nn
from spice_amd import nets, synth
from spice_amd.sweep import SensitivitySweep, DEFAULT_RATIOS
dev = torch.device("cuda:0")
torch.manual_seed(1234)
ARCH, ENC = (sys.argv[4:] + ["DeepLabV3Plus", "resnet50"])[:2]
model = nets.create_model(ARCH, ENC, 21).to(dev).eval()
ignored = [m for m in model.segmentation_head.modules() if isinstance(m, nn.Conv2d)]
x = torch.from_numpy(synth.normal(1234, 50, 8 * 3 * 512 * 512).reshape(8, 3, 512, 512)).to(dev)
lab = torch.from_numpy(synth.label_map(1234, 8, 512, 512, 21, "blocky", ignore_share=0.05)).to(dev)
RATIOS = DEFAULT_RATIOS if ARCH == "DeepLabV3Plus" and ENC == "resnet50" else (0.25, 0.5, 0.75)
idx = list(range(int(sys.argv[1]), int(sys.argv[2])))
reuse = len(sys.argv) > 3 and sys.argv[3] == "reuse"
HOLD = []
if os.environ.get("HOLD_DST") == "1":   # the sliced tensors of every candidate stay allocated: no address comes back
    from spice_amd.engine import pruner as _pr
    _orig_run = _pr.SliceJob.run
    def _run(self, keep_undo=False):
        r = _orig_run(self, keep_undo)
        for o, a, _, _ in self.entries:
            t = o._parameters.get(a, None)
            HOLD.append(t if t is not None else o._buffers.get(a))
        return r
    _pr.SliceJob.run = _run
def run(lanes):
    sweep = SensitivitySweep(model, 21, RATIOS, ignored_layers=ignored)
    sweep.prefix_reuse = reuse
    sweep.enable_lanes(lanes)
    return sweep.run(x, lab, indices=idx)["confusion"][idx].clone()
base = run(1)
print("lanes 1 repeat equal:", bool(torch.equal(base, run(1))))
bad = {}
for r in range(12):
    d = (run(2) - base).abs().sum((1, 2))
    for i in d.nonzero().flatten().tolist():
        bad[idx[i]] = bad.get(idx[i], 0) + 1
print("mismatching candidates over 12 two-lane runs:", bad)
