"""Experiment: sweep candidates evaluated from T host threads of ONE process, each with its own model
replica and HIP stream (torch releases the GIL inside its operators, so the host work of the
threads overlaps).  Prints candidates/s for T = 1, 2, 3."""
import copy, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spice_amd import miopen_tuning
miopen_tuning.ensure_tuned("DeepLabV3Plus", "resnet50", 21, 8, 512, "fp16")
import torch
from spice_amd import nets, synth
from spice_amd.sweep import DEFAULT_RATIOS, SensitivitySweep
dev = torch.device("cuda:0"); C = 21
torch.manual_seed(1234)
base = nets.create_model("DeepLabV3Plus", "resnet50", C).to(dev).eval()
img = torch.from_numpy(synth.normal(1234, 50, 8 * 3 * 512 * 512).reshape(8, 3, 512, 512)).to(dev)
lab = torch.from_numpy(synth.label_map(1234, 8, 512, 512, C)).to(dev)

def make(model):
    ignored = [m for m in model.segmentation_head.modules() if isinstance(m, torch.nn.Conv2d)]
    return SensitivitySweep(model, C, DEFAULT_RATIOS, ignored_layers=ignored)

for T in (1, 2, 3):
    sweeps = [make(base if t == 0 else copy.deepcopy(base)) for t in range(T)]
    n = len(sweeps[0].candidates)
    table = torch.zeros(n, C, C, dtype=torch.int64, device=dev)
    kd = torch.zeros(n, 3, device=dev)
    with sweeps[0].session():
        teacher = sweeps[0].forward(img).float().contiguous()
    todo = list(range(60, 60 + 40 * T))
    shares = [todo[t::T] for t in range(T)]
    for s, share in zip(sweeps, shares):
        s.prepare(share); s.warm_shapes(img, share)
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=dev) for _ in range(T)]
    def work(t, idx):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(streams[t]), sweeps[t].session():
            for i in idx:
                sweeps[t].evaluate(i, img, lab, teacher, table[i], kd[i])
            streams[t].synchronize()
    # warm-up: 10 candidates per thread, then the timed 30 per thread
    for phase, sl in (("warm", slice(0, 10)), ("timed", slice(10, 40))):
        threads = [threading.Thread(target=work, args=(t, shares[t][sl])) for t in range(T)]
        t0 = time.perf_counter()
        for th in threads: th.start()
        for th in threads: th.join()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if phase == "timed":
            cnt = 30 * T
            ok = int(table[todo].sum()) == len(todo) * int((lab != 255).sum())
            print(f"threads={T}: {cnt} candidates in {1e3 * dt:.1f} ms = {cnt / dt:.1f} candidates/s, pixel counts ok: {ok}", flush=True)
