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

for T in (0, 1, 2, 3):
  main_only = T == 0
  T = max(T, 1)
  if True:
    sweeps = [make(base if t == 0 else copy.deepcopy(base)) for t in range(T)]
    n = len(sweeps[0].candidates)
    table = torch.zeros(n, C, C, dtype=torch.int64, device=dev)
    kd = torch.zeros(n, 3, device=dev)
    with sweeps[0].session():
        teacher = sweeps[0].forward(img).float().contiguous()
    todo = list(range(2, n, 3))[:40 * T]   # spread over the whole network
    shares = [todo[t::T] for t in range(T)]
    for s, share in zip(sweeps, shares):
        s.prepare(share); s.warm_shapes(img, share)
    torch.cuda.synchronize()
    same = os.environ.get("SAME_STREAM", "0") == "1"   # every thread on the default stream: device work in order, host work overlapped
    streams = [torch.cuda.current_stream(dev) if same else torch.cuda.Stream(device=dev) for _ in range(T)]
    if main_only:   # the plain loop of SensitivitySweep.run / bench.py, no worker thread
        with sweeps[0].session():
            for i in shares[0][:10]:
                sweeps[0].evaluate(i, img, lab, teacher, table[i], kd[i])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in shares[0][10:40]:
                sweeps[0].evaluate(i, img, lab, teacher, table[i], kd[i])
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        print(f"main thread only: 30 candidates in {1e3 * dt:.1f} ms = {30 / dt:.1f} candidates/s", flush=True)
        continue
    gate = threading.Barrier(T + 1)
    def work(t):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(streams[t]), sweeps[t].session():
            for i in shares[t][:10]:     # warm-up in the same thread that runs the timed part
                sweeps[t].evaluate(i, img, lab, teacher, table[i], kd[i])
            streams[t].synchronize()
            gate.wait()
            for i in shares[t][10:40]:
                sweeps[t].evaluate(i, img, lab, teacher, table[i], kd[i])
            streams[t].synchronize()
            gate.wait()
    threads = [threading.Thread(target=work, args=(t,)) for t in range(T)]
    for th in threads: th.start()
    gate.wait()
    t0 = time.perf_counter()
    gate.wait()
    dt = time.perf_counter() - t0
    for th in threads: th.join()
    torch.cuda.synchronize()
    cnt = 30 * T
    ok = int(table[todo].sum()) == len(todo) * int((lab != 255).sum())
    print(f"threads={T}: {cnt} candidates in {1e3 * dt:.1f} ms = {cnt / dt:.1f} candidates/s, pixel counts ok: {ok}", flush=True)
