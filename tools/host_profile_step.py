"""cProfile of the candidate loop (host side): where the Python + dispatch time of a sweep step goes."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from spice_amd import nets, miopen_tuning
from spice_amd.sweep import SensitivitySweep, DEFAULT_RATIOS

miopen_tuning.ensure_tuned("DeepLabV3Plus", "resnet50", bench.CLASSES, bench.BATCH, bench.SIZE, "fp16", device=0)
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = nets.create_model("DeepLabV3Plus", "resnet50", bench.CLASSES).to(dev).eval()
images, labels = bench.build_inputs(dev)
ignored = [m for m in model.segmentation_head.modules() if isinstance(m, torch.nn.Conv2d)]
sw = SensitivitySweep(model, bench.CLASSES, DEFAULT_RATIOS, ignored_layers=ignored)
sw.enable_lanes(int(os.environ.get("LANES", "2")))
idx = list(range(0, len(sw.candidates), 5))[:80]
sw.prepare(idx); sw.warm_shapes(images, idx)
n = len(sw.candidates)
table = torch.zeros(n, bench.CLASSES, bench.CLASSES, dtype=torch.int64, device=dev)
kd = torch.zeros(n, 3, device=dev)
with sw.session():
    teacher = sw.teacher_forward(images).float().contiguous()   # (records the activations the candidates start from)
    sw.fork_lanes()
    for i in idx[:20]:
        sw.evaluate(i, images, labels, teacher, table[i], kd[i])
    sw.join_lanes()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in idx[20:]:
        sw.evaluate(i, images, labels, teacher, table[i], kd[i])
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(f"{len(idx)-20} candidates: enqueue {1e3*t_enq/(len(idx)-20):.3f} ms each, with device {1e3*t_all/(len(idx)-20):.3f} ms each")
    prof = cProfile.Profile()
    prof.enable()
    for i in idx[20:]:
        sw.evaluate(i, images, labels, teacher, table[i], kd[i])
    prof.disable()
    torch.cuda.synchronize()
st = pstats.Stats(prof)
st.sort_stats("tottime").print_stats(40)
st.sort_stats("cumtime").print_stats(30)

# the launch plans of 40 candidates not seen so far (what the bench's timed region holds up front)
new = [i for i in range(2, len(sw.candidates), 5)][:40]
torch.cuda.synchronize()
t0 = time.perf_counter()
sw.prepare(new[:20])
torch.cuda.synchronize()
print(f"plans of 20 new candidates: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms each")
prof = cProfile.Profile()
prof.enable()
sw.prepare(new[20:])
prof.disable()
torch.cuda.synchronize()
st = pstats.Stats(prof)
st.sort_stats("tottime").print_stats(35)
st.sort_stats("cumtime").print_stats(25)
