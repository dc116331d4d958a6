"""Benchmark of the MI355X pruning hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Either invocation works: started as a plain process with --gpus N > 1 (no WORLD_SIZE in the environment), this
file starts the N ranks itself - `python -m torch.distributed.run ... bench.py <same arguments>` as a CHILD process,
before anything here has imported torch or touched a GPU - lets rank 0's JSON line through and exits with the
child's status.

Workload: the magnitude_taylor sensitivity sweep of DeepLabV3Plus_resnet on one synthetic
21-class 8x3x512x512 evaluation batch.  One step = one sweep candidate per rank: slice one channel
group at one ratio (HIP gather), forward the sliced network from the first block the group touches
(fp16 autocast as the reference's evaluator; every convolution on the library's kernels; the
activations of the un-pruned network in front of that block come from the teacher forward of the
same sweep), and ONE launch for the head's up-sampling, the mIoU confusion matrix (arg-max
histogram) and the KD-Lite loss against the un-pruned network's logits.  Weak scaling: every rank
evaluates K candidates spread evenly over its cost-balanced share; the per-candidate tables are
combined with one all-reduce inside the timed region; the timed candidates' launch plans are made
inside it, each in front of its candidate's launches (uploaded by a launch).  Set-up outside the
timed region: model build, the Taylor calibration pass (HIP Taylor reduction) with quota allocation,
group magnitude importance (HIP), launch plans of the warm-up candidates, the un-pruned network's
forward from every block on as captured HIP graphs (once per geometry).  Rank 0 prints ONE JSON line.

    python bench.py --gpus N --strong
times the WHOLE sweep instead (strong scaling: the 414 candidates are fixed, ranks share them):
SensitivitySweep.run() wall-clock per rank, including the geometry pass, the cost model and the
cost-balanced partition, launch plans, solver lookups for the rank's new convolution shapes, the
teacher forward, every candidate and the all-reduce of the tables.  The find-database warm-up,
the model build and the sharded Taylor calibration are reported beside it under "setup".
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


def launch_ranks_if_needed(argv=None) -> None:
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: start the N ranks as a child
    `python -m torch.distributed.run` of this same file and exit with its status.  Runs before torch or the
    package is imported, so this process never initialises a GPU (it only waits for the child; the child is a
    new process, not an exec of this one).  Returns when the process is itself a rank, or N == 1."""
    argv = sys.argv[1:] if argv is None else list(argv)
    probe = argparse.ArgumentParser(add_help=False)
    probe.add_argument("--gpus", type=int, default=1)
    known, _ = probe.parse_known_args(argv)
    if known.gpus <= 1 or "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return
    import socket
    import subprocess
    with socket.socket() as s:   # a free rendezvous port on the loop-back interface
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={known.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs between the ranks' processes
    # torchrun gives every rank ONE host thread unless told otherwise; the ranks share the host's cores instead
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // known.gpus)))
    print(f"[bench] starting {known.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


if __name__ == "__main__":
    launch_ranks_if_needed()

import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, ROOT)

from spice_amd import _native as nat  # noqa: E402
from spice_amd import dist as sdist  # noqa: E402
from spice_amd import nets, ops, synth  # noqa: E402
from spice_amd.sweep import DEFAULT_RATIOS, SensitivitySweep  # noqa: E402

PEAK_HBM_GBPS = 8000.0  # MI355X HBM3E peak
PEAK_MFMA_F16_TFLOPS = 2500.0  # dense fp16 matrix-core peak
CLASSES, BATCH, SIZE = 21, 8, 512


def build_inputs(dev):
    img = synth.normal(1234, 50, BATCH * 3 * SIZE * SIZE).reshape(BATCH, 3, SIZE, SIZE)
    lab = synth.label_map(1234, BATCH, SIZE, SIZE, CLASSES, "blocky", ignore_share=0.05)
    return torch.from_numpy(img).to(dev), torch.from_numpy(lab).to(dev)


def taylor_and_quotas(model, images, labels, dev, shard=False):
    """The sensitivity analysis of the magnitude_taylor strategy: Taylor scores over calibration
    batches (labels must be in [0,C): ignore pixels are mapped to class 0) and layer quotas."""
    from spice_amd.pruning.pruner import LayerQuotaCalculator
    from spice_amd.pruning.sensitivity_utils import compute_taylor_scores
    cal = labels.clone()
    cal[cal == 255] = 0
    half = BATCH // 2
    loader = [(images[:half], cal[:half]), (images[half:], cal[half:])]
    t0 = time.perf_counter()
    scores = compute_taylor_scores(model, loader, str(dev), max_batches=2, shard=shard)
    torch.cuda.synchronize()
    t1 = time.perf_counter()

    class _Imp:
        sensitivity_scores = scores

    quotas, _ = LayerQuotaCalculator.compute_layer_quotas(model, _Imp, 0.5, alpha=0.7, min_channels=8)
    return scores, quotas, t1 - t0


def event_ms(fn, iters, warmup=3, batch=8):
    """Median time per launch over `iters` batches of `batch` back-to-back launches (the GPU stays
    busy inside a batch, so host launch latency is not part of the figure)."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    times = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(batch):
            fn()
        b.record()
        b.synchronize()
        times.append(a.elapsed_time(b) / batch)
    return float(np.median(times))


def pmc_traffic(kernel_key):
    """(HBM bytes per launch, file it was read from): the figure comes from the newest committed
    rocprofv3 --pmc summary under profiles/, not from this run (counters need a run of their own)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_traffic.json")), reverse=True):
        try:
            with open(path) as f:
                return json.load(f)["kernels"][kernel_key]["hbm_bytes_corrected"], os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def rocprof_median_us(kernel_key):
    """(median duration in us, file): the kernel's per-dispatch median in the newest committed rocprofv3 kernel trace of
    this command with one candidate on the device at a time (the profiler serialises dispatches; this run's own figure
    comes from HIP events)."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "bench_lanes1_kernel_medians.txt")), reverse=True):
        try:
            with open(path) as f:
                for line in f:
                    if kernel_key in line:
                        return float(line.split()[1]), os.path.relpath(path, ROOT)
        except (OSError, ValueError, IndexError):
            continue
    return None, None


def cpu_description():
    """(threads the CPU legs use, what the host is): the box's share of host cores is what OMP_NUM_THREADS says
    where it is set (os.cpu_count() names every core of the machine, many times the share of a one-GPU box)."""
    n_os = os.cpu_count() or 1
    try:
        share = int(os.environ.get("OMP_NUM_THREADS", "0"))
    except ValueError:
        share = 0
    threads = min(n_os, share) if share > 0 else n_os
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return threads, {"cpu_model": model, "os_cpu_count": n_os, "threads_used": threads,
                     "threads_from": "OMP_NUM_THREADS" if share > 0 else "os.cpu_count()"}


def cpu_min_ms(fn, reps=3):
    """min-of-`reps` wall-clock of a CPU leg, in ms (one untimed call first: allocator and thread pool)."""
    fn()
    best = float("inf")
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        best = min(best, time.perf_counter() - t0)
    return 1e3 * best


def kernel_table(dev, cpu_legs=True):
    """HIP-event time of every hand-written kernel at the benchmark sizes, cycling through more
    than 256 MiB of distinct buffers so the Infinity Cache cannot serve repeats.  The four path functions also
    carry `cpu_ms`: the reference's torch op sequence for that function (the oracle's restatement of it) on the
    same inputs copied to the host, min of 3, on the host cores of the same box."""
    out = []
    threads, _ = cpu_description()
    if cpu_legs:
        from oracle import confusion as oc
        from oracle import importance as oi
        from oracle import kdlite as ok
        from oracle import selection as osel
        torch.set_num_threads(threads)

    def row(name, ms, nbytes, cpu_ms=None, cpu_what=None):
        gbs = nbytes / ms / 1e6
        out.append({"kernel": name, "ms": round(ms, 4), "algorithmic_MB": round(nbytes / 1e6, 1),
                    "GBps": round(gbs, 1), "frac": round(gbs / PEAK_HBM_GBPS, 3)})
        if cpu_ms is not None:
            out[-1].update({"cpu_ms": round(cpu_ms, 2), "cpu_GBps": round(nbytes / cpu_ms / 1e6, 2), "cpu_threads": threads,
                            "cpu_leg": cpu_what, "gpu_over_cpu": round(cpu_ms / ms, 1)})

    def row_flops(name, ms, macs, nbytes):
        tf = 2 * macs / ms / 1e9
        out.append({"kernel": name, "ms": round(ms, 4), "GFLOP": round(2 * macs / 1e9, 1), "TFLOPs": round(tf, 1),
                    "frac_mfma": round(tf / PEAK_MFMA_F16_TFLOPS, 3), "algorithmic_MB": round(nbytes / 1e6, 1),
                    "GBps": round(nbytes / ms / 1e6, 1)})

    shapes = synth.deeplab_resnet50_conv_shapes(CLASSES)
    offs, pos = [], 0
    for o, i, kh, kw in shapes:
        offs.append(pos)
        pos += (o * i * kh * kw + 3) // 4 * 4
    layers, opos = [], 0
    for (o, i, kh, kw), off in zip(shapes, offs):
        layers.append(ops.RowsLayout(off, off, opos, o, i * kh * kw))
        opos += o
    plan = ops.RowsPlan(layers, dev)
    ws = [torch.randn(pos, device=dev) * 0.05 for _ in range(4)]
    gs = [torch.randn(pos, device=dev) * 1e-3 for _ in range(4)]
    score = torch.zeros(opos, device=dev)
    k = [0]

    def cyc(lst):
        k[0] += 1
        return lst[k[0] % len(lst)]

    cpu_w = cpu_g = None
    if cpu_legs:   # the same values, as the per-layer tensors the reference's loop walks
        wh, gh = ws[0].cpu(), gs[0].cpu()
        cpu_w = [wh[off: off + o * i * kh * kw].view(o, i, kh, kw) for (o, i, kh, kw), off in zip(shapes, offs)]
        cpu_g = [gh[off: off + o * i * kh * kw].view(o, i, kh, kw) for (o, i, kh, kw), off in zip(shapes, offs)]
        cpu_scores = [torch.zeros(o) for o, _, _, _ in shapes]

        def cpu_taylor():
            for w_, g_, sc in zip(cpu_w, cpu_g, cpu_scores):
                sc += oi.taylor_rows(w_, g_)
    row("importance_rows<L2> (DeepLabV3+-R50 conv weights)",
        event_ms(lambda: plan.run([cyc(ws)], (), score, nat.IMP_L2, False), 20),
        plan.algorithmic_bytes(nat.IMP_L2),
        cpu_min_ms(lambda: [oi.magnitude_rows(w_, 2) for w_ in cpu_w]) if cpu_legs else None,
        "w.flatten(1).abs().pow(2).sum(1) per layer (torch-pruning's magnitude form)")
    row("importance_rows<TAYLOR> (weights + grads, accumulate)",
        event_ms(lambda: plan.run([cyc(ws)], [gs[k[0] % 4]], score, nat.IMP_TAYLOR, True), 20),
        plan.algorithmic_bytes(nat.IMP_TAYLOR),
        cpu_min_ms(cpu_taylor) if cpu_legs else None,
        "scores += (grad.abs() * weight.abs()).sum((1,2,3)) per layer, pruning/sensitivity_utils.py:61-66")
    specs, idx, dpos, ipos = [], [], 0, 0
    for (o, i, kh, kw), off in zip(shapes, offs):
        ko = np.arange(0, o, 2, dtype=np.int64)
        specs.append(ops.GatherSpec(off, dpos, ipos, -1, len(ko), i, i, kh * kw))
        idx.append(ko)
        ipos += len(ko)
        dpos += (len(ko) * i * kh * kw + 3) // 4 * 4
    gplan = ops.GatherPlan(specs, dev)
    idx_t = torch.from_numpy(np.concatenate(idx)).to(dev)
    dsts = [torch.empty(dpos, device=dev) for _ in range(4)]
    row("gather (keep every 2nd out-channel of every conv)",
        event_ms(lambda: gplan.run([cyc(ws)], dsts[k[0] % 4], idx_t), 20),
        gplan.algorithmic_bytes(),
        cpu_min_ms(lambda: [osel.slice_tensor(w_, ko, None) for w_, ko in zip(cpu_w, idx)]) if cpu_legs else None,
        "index_select(weight, 0, keep).contiguous() per layer (torch-pruning's slicing)")
    del ws, gs, dsts, cpu_w, cpu_g

    sets = []
    for j in range(2):
        s = 3.0 * torch.randn(BATCH, CLASSES, SIZE, SIZE, device=dev)
        sets.append((s, s + torch.randn_like(s), torch.randint(0, CLASSES, (BATCH, SIZE, SIZE), device=dev)))
    P = BATCH * SIZE * SIZE
    cpu_kd = None
    if cpu_legs:
        hs, ht, hl = (x.cpu() for x in sets[0])
        with torch.no_grad():
            cpu_kd = cpu_min_ms(lambda: ok.kdlite_loss(hs, ht, hl, temperature=4.0, alpha=0.7, num_classes=CLASSES))
        del hs, ht, hl
    row("kdlite forward (T=4, alpha=0.7)",
        event_ms(lambda: ops.kdlite_forward(*cyc(sets), temperature=4.0, alpha=0.7), 20), P * (2 * CLASSES * 4 + 8),
        cpu_kd, "KDLiteLoss.forward op sequence, pruning/losses.py:127-162")
    row("kdlite forward+grad (label pre-pass + fused pass)",
        event_ms(lambda: ops.kdlite_forward(*cyc(sets), temperature=4.0, alpha=0.7, want_grad=True), 10, batch=4),
        P * (3 * CLASSES * 4 + 8))
    lg16 = [s[0].half() for s in sets] + [sets[0][1].half()]
    conf = torch.zeros(CLASSES, CLASSES, dtype=torch.int64, device=dev)
    row("argmax+confusion fp16 logits",
        event_ms(lambda: ops.argmax_confusion_update(conf, cyc(lg16), sets[0][2], CLASSES), 20),
        P * (CLASSES * 2 + 8))
    del lg16
    small = [(torch.randint(0, CLASSES, (P,), device=dev), torch.randint(0, CLASSES, (P,), device=dev)) for _ in range(10)]
    cpu_cf = None
    if cpu_legs:
        hp, ht = small[0][0].cpu(), small[0][1].cpu()
        hc = torch.zeros(CLASSES, CLASSES, dtype=torch.int64)
        cpu_cf = cpu_min_ms(lambda: oc.update_confusion_torch(hc, hp, ht, CLASSES))
    row("confusion [8,512,512] int64, 21 classes",
        event_ms(lambda: ops.confusion_update(conf, *cyc(small), CLASSES), 20), P * 16,
        cpu_cf, "update_confusion_matrix op sequence, baseline/eval/metrics.py:8-19")
    del small, sets
    P5 = 32 * 1024 * 2048
    t5 = torch.randint(0, 19, (P5,), device=dev)
    t5[torch.rand(P5, device=dev) < 0.1] = 255
    p5 = torch.randint(0, 19, (P5,), device=dev)
    conf5 = torch.zeros(19, 19, dtype=torch.int64, device=dev)
    cpu_cf5 = None
    if cpu_legs:
        hp, ht = p5.cpu(), t5.cpu()
        hc = torch.zeros(19, 19, dtype=torch.int64)
        cpu_cf5 = cpu_min_ms(lambda: oc.update_confusion_torch(hc, hp, ht, 19))
        del hp, ht
    row("confusion [32,1024,2048] int64, 19 classes",
        event_ms(lambda: ops.confusion_update(conf5, p5, t5, 19), 10, batch=4), P5 * 16,
        cpu_cf5, "update_confusion_matrix op sequence, baseline/eval/metrics.py:8-19")
    # the evaluation loops' fused entry points at the same size: arg-max over [32,19,1024,2048] logits inside the
    # histogram launch (fp16: 2.5 GB, past 2^31 bytes), and the x4 up-sampling of [32,19,256,512] on the fly
    for dt, nm in ((torch.float16, "fp16"), (torch.float32, "fp32")):
        lg5 = torch.randn(32, 19, 1024, 2048, device=dev, dtype=dt)
        row(f"argmax+confusion {nm} logits [32,19,1024,2048]",
            event_ms(lambda: ops.argmax_confusion_update(conf5, lg5, t5.view(32, 1024, 2048), 19), 5, warmup=2, batch=2),
            P5 * (19 * lg5.element_size() + 8))
        del lg5
    low5 = torch.randn(32, 19, 256, 512, device=dev).half()
    row("x4 up-sampling + arg-max/confusion in one launch, [32,19,256,512] -> 1024x2048",
        event_ms(lambda: ops.upsample_argmax_confusion(conf5, low5, t5.view(32, 1024, 2048), align_corners=True, scale_factor=4.0),
                 5, warmup=2, batch=2), low5.numel() * 2 + P5 * 8)
    del t5, p5, low5
    # the forward-side kernels at the shapes they meet in this workload (results dropped at once: the caching
    # allocator hands the same few output buffers round, the inputs rotate)
    heads = [torch.randn(BATCH, CLASSES, SIZE // 4, SIZE // 4, device=dev).half() for _ in range(3)]
    row("bilinear x4 up-sampling of the head, fp16 -> fp32 logits",
        event_ms(lambda: ops.upsample_bilinear(cyc(heads), scale_factor=4.0, align_corners=True, out_dtype=torch.float32), 20),
        heads[0].numel() * (2 + 16 * 4))
    # a sweep candidate's tail and an evaluation batch's, up-sampling done on the fly (teachers rotate: 2 x 176 MB)
    lab = torch.randint(0, CLASSES, (BATCH, SIZE, SIZE), device=dev)
    teach = [3.0 * torch.randn(BATCH, CLASSES, SIZE, SIZE, device=dev) for _ in range(2)]
    head_bytes = heads[0].numel() * 2 + P * (CLASSES * 4 + 8)
    row("head of a sweep candidate in one launch: x4 up-sampling + KD-Lite loss + arg-max/confusion (teacher as logits)",
        event_ms(lambda: ops.sweep_head(cyc(heads), cyc(teach), lab, conf, align_corners=True, scale_factor=4.0,
                                        temperature=4.0, alpha=0.7), 20), head_bytes)
    prep = [ops.kd_teacher_prepare(t, 4.0) for t in teach]
    del teach

    def head_prepared():
        pr = cyc(prep)
        return ops.sweep_head(cyc(heads), pr[0], lab, conf, align_corners=True, scale_factor=4.0, temperature=4.0, alpha=0.7,
                              teacher_hneg=pr[1])
    row("the same against the teacher's softmax prepared once per sweep (the form the sweep runs)",
        event_ms(head_prepared, 20), head_bytes + P * 4)
    del prep
    row("x4 up-sampling + arg-max/confusion in one launch (evaluation loops)",
        event_ms(lambda: ops.upsample_argmax_confusion(conf, cyc(heads), lab, align_corners=True, scale_factor=4.0), 20),
        heads[0].numel() * 2 + P * 8)
    acts = [torch.randn(BATCH, 256, SIZE // 4, SIZE // 4, device=dev).half() for _ in range(4)]
    bn = [torch.rand(256, device=dev) + 0.5 for _ in range(4)]
    row("batch-norm + residual add + ReLU, [8,256,128,128] fp16",
        event_ms(lambda: ops.bn_act(cyc(acts), bn[0], bn[1], bn[2], bn[3], 1e-5, residual=acts[k[0] % 4 - 1], relu=True), 20),
        acts[0].numel() * 2 * 3)
    wdw = torch.randn(256, 1, 3, 3, device=dev)
    row("depthwise 3x3, [8,256,128,128] fp16",
        event_ms(lambda: ops.depthwise_conv3x3(cyc(acts), wdw), 20), acts[0].numel() * 2 * 2)
    # the matrix-core pair kernel on a bottleneck's last convolution: 64 -> 256 channels on 8x128x128, batch-norm,
    # residual add and ReLU in the one launch; bytes = x + residual + y (the 32 KB of weights stay in L2)
    xin = [torch.randn(BATCH, 64, SIZE // 4, SIZE // 4, device=dev).half() for _ in range(4)]
    w64 = (torch.randn(256, 64, device=dev) / 8.0).half()
    norm = torch.nn.BatchNorm2d(256).eval().to(dev)
    with torch.no_grad():
        row("1x1 convolution 64->256 + batch-norm + residual + ReLU in one launch, [8,.,128,128] fp16",
            event_ms(lambda: ops.conv1x1_bn_act(cyc(xin), w64, norm, acts[k[0] % 4], None, True, route=1), 20),
            xin[0].numel() * 2 + acts[0].numel() * 2 * 2)
        stem_in = [torch.randn(BATCH, 64, SIZE // 2, SIZE // 2, device=dev).half() for _ in range(4)]
        norm64 = torch.nn.BatchNorm2d(64).eval().to(dev)
        row("ResNet stem: batch-norm + ReLU + 3x3/2 max-pooling in one launch, [8,64,256,256] fp16",
            event_ms(lambda: ops.bn_act_maxpool(cyc(stem_in), norm64, want_full=False), 20),
            stem_in[0].numel() * 2 * 5 // 4)
        # the same kernel where it is bound by the matrix cores and the tile traffic through L2 / LDS: deep K
        for name, c, side, dil in (("3x3 convolution 512->512, dilation 2, + batch-norm + ReLU in one launch, [8,.,32,32] fp16", 512, 32, 2),
                                   ("3x3 convolution 256->256 + batch-norm + ReLU in one launch, [8,.,32,32] fp16", 256, 32, 1)):
            x3 = [torch.randn(BATCH, c, side, side, device=dev).half() for _ in range(4)]
            w9 = (torch.randn(c, 9 * c, device=dev) / (9 * c) ** 0.5).half()
            nrm = torch.nn.BatchNorm2d(c).eval().to(dev)
            row_flops(name, event_ms(lambda: ops.conv3x3_bn_act(cyc(x3), w9, nrm, True, 1, dil), 20),
                      BATCH * side * side * c * c * 9, 2 * x3[0].numel() * 2)
        xa = [torch.randn(BATCH, 2048, 32, 32, device=dev).half() for _ in range(4)]
        wa = (torch.randn(256, 2048, device=dev) / 2048 ** 0.5).half()
        row_flops("1x1 convolution 2048->256 + batch-norm + ReLU in one launch, [8,.,32,32] fp16",
                  event_ms(lambda: ops.conv1x1_bn_act(cyc(xa), wa, norm, None, None, True, route=1), 20),
                  BATCH * 32 * 32 * 2048 * 256, 2 * (xa[0].numel() + BATCH * 256 * 32 * 32))
    return out


def forward_probe(sweep, candidates, images, labels, teacher, table, kd):
    """The matrix-core convolution kernel, which takes most of a step's device time, against the MFMA peak: a few
    more candidates behind the timed region, with an event pair round every convolution call (not inside the timed
    region: ~120 event records per step would be in its way)."""
    records = []
    ops.set_conv_probe(records)
    try:
        with sweep.session():
            for c in candidates:
                table[c].zero_()
                sweep.evaluate(c, images, labels, teacher, table[c], kd[c])
        torch.cuda.synchronize()
    finally:
        ops.set_conv_probe(None)
    mine = [r for r in records if r[0].startswith("matrix_core")]
    ms = sum(a.elapsed_time(b) for _, a, b, _, _ in mine)
    macs, nbytes = sum(r[3] for r in mine), sum(r[4] for r in mine)
    steps = max(len(candidates), 1)
    by_kind = {}
    for kind, a, b, m, _ in mine:
        d = by_kind.setdefault(kind, [0, 0.0, 0])
        d[0] += 1
        d[1] += a.elapsed_time(b)
        d[2] += m
    tflops = 2 * macs / ms / 1e9
    return {"bound": "mfma", "kernel": "spice::conv1x1_bn_act_mfma_kernel (every instantiation: 1x1 pairs, stride-2 "
            "shortcuts, 3x3 convolutions)", "achieved": round(tflops, 1), "peak": PEAK_MFMA_F16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(tflops / PEAK_MFMA_F16_TFLOPS, 4), "launches_per_step": round(len(mine) / steps, 1),
            "avg_launch_us": round(1e3 * ms / max(len(mine), 1), 2), "us_per_step": round(1e3 * ms / steps, 1),
            "GFLOP_per_step": round(2 * macs / steps / 1e9, 1),
            "algorithmic_GBps": round(nbytes / ms / 1e6, 1), "algorithmic_MB_per_step": round(nbytes / steps / 1e6, 1),
            "by_kind": {k: {"launches_per_step": round(v[0] / steps, 1), "us_per_step": round(1e3 * v[1] / steps, 1),
                            "TFLOPs": round(2 * v[2] / v[1] / 1e9, 1)} for k, v in by_kind.items()},
            "measured": f"{steps} candidates behind the timed region, HIP events round each call on its stream "
                        "(event pairs add gaps between launches, not to them)"}


def cpu_candidate_baseline(model, sweep, images, labels, ignored):
    """One sweep candidate through the CPU oracle on the host cores, min of 3: group magnitude importance,
    stable selection, index_select slicing, fp32 forward, arg-max, confusion, KD-Lite loss.  Beside the time: how
    far the HIP path's results for the same candidate are from the oracle's, with the fp32 forward and with the
    fp16 forward the benchmark runs."""
    import copy

    from oracle import confusion as oc
    from oracle import kdlite as ok
    from oracle import prune_ref
    from oracle import selection as osel
    from spice_amd.engine.depgraph import build_groups
    from spice_amd.sweep import miou_from_confusion

    threads, host = cpu_description()
    torch.set_num_threads(threads)
    base = copy.deepcopy(model).cpu().float().eval()
    x, y = images.cpu(), labels.cpu()
    with torch.no_grad():
        teacher = base(x)
    cand = sweep.candidates[len(sweep.candidates) // 2]
    root_name = sweep.groups[cand.group].root_name
    times, conf, loss, pruned = [], None, None, None
    for _ in range(3):
        cpu_model = copy.deepcopy(base)
        g = [gg for gg in build_groups(cpu_model) if gg.root_name == root_name][0]
        t0 = time.perf_counter()
        imp = prune_ref.group_importance(g, 2)
        pruned = osel.select_local(imp, len(cand.pruned))
        prune_ref.prune_group(g, pruned)
        with torch.no_grad():
            logits = cpu_model(x)
        conf = np.zeros((CLASSES, CLASSES), dtype=np.int64)
        oc.update_confusion(conf, oc.argmax_first(logits.numpy()), y.numpy(), CLASSES)
        loss = ok.kdlite_loss(logits, teacher, y, temperature=4.0, alpha=0.7, num_classes=CLASSES)
        times.append(time.perf_counter() - t0)
        del cpu_model, logits
    dt = min(times)
    same = bool(np.array_equal(pruned, cand.pruned))
    miou_cpu = oc.metrics_from_confusion(torch.from_numpy(conf))[2]

    def hip_candidate(autocast_dtype):
        sw = sweep if autocast_dtype is not None else SensitivitySweep(
            model, CLASSES, DEFAULT_RATIOS, ignored_layers=ignored, autocast_dtype=None)
        r = sw.run(images, labels, indices=[cand.index])
        cf = r["confusion"][cand.index]
        return cf.cpu().numpy(), float(miou_from_confusion(cf).item()), [float(v) for v in r["kd"][cand.index].cpu()]

    deltas = {}
    for name, dt_ in (("hip_fp32_forward", None), ("hip_fp16_forward", torch.float16)):
        cf, miou, kd3 = hip_candidate(dt_)
        deltas[name] = {"pixels_in_other_cell": int(np.abs(cf - conf).sum()) // 2, "abs_d_miou": abs(miou - miou_cpu),
                        "rel_d_loss_total": abs(kd3[0] - float(loss["total"])) / max(abs(float(loss["total"])), 1e-12),
                        "rel_d_loss_ce": abs(kd3[1] - float(loss["ce"])) / max(abs(float(loss["ce"])), 1e-12),
                        "abs_d_loss_kd": abs(kd3[2] - float(loss["kd"]))}
    return {"value": round(1.0 / dt, 4), "unit": "candidates/s", "cores": threads, "kind": "port", **host,
            "seconds_min_of_3": round(dt, 3), "seconds_all": [round(t, 3) for t in times],
            "selected_channels_equal": same, "valid_pixels": int(conf.sum()), "miou_cpu": miou_cpu,
            "hip_vs_cpu": deltas,
            "sample": f"1 candidate (group {root_name}, ratio {cand.ratio}) on the same 8x3x512x512 batch, 3 times, minimum: "
                      f"oracle importance+selection+index_select slicing, fp32 torch-CPU forward of the whole sliced network, "
                      f"arg-max, confusion, KD-Lite loss; {dt:.2f} s on {threads} threads"}


def whole_sweep(sweep, images, labels, dev, rank, world):
    """SensitivitySweep.run() over ALL candidates, wall-clock, max over ranks, with every rank's share."""
    n_cand = len(sweep.candidates)
    sdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = sweep.run(images, labels)
    torch.cuda.synchronize()
    mine_s = time.perf_counter() - t0
    sdist.barrier()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    sdist.all_reduce_max_(t)
    elapsed = float(t.item())
    # per-rank split: own work (set-up of the share + candidates) against waiting for the slowest rank
    ph = sweep.last_share.get("phases", {})
    keys = ["geometry_and_costs_s", "plans_s", "solver_lookups_s", "teacher_forward_s", "graph_capture_s", "candidates_s"]
    share = torch.zeros(world, 4 + len(keys), device=dev, dtype=torch.float64)
    share[rank] = torch.tensor([sweep.last_share["candidates"], sweep.last_share["cost"], sweep.last_share["busy_s"], mine_s] +
                               [ph.get(k, 0.0) for k in keys], dtype=torch.float64)
    sdist.all_reduce_sum_(share)
    share = share.cpu().numpy()
    miou = res["miou"].cpu().numpy()
    return {
        "candidates": n_cand, "sweep_s": round(elapsed, 4), "candidates_per_s": round(n_cand / elapsed, 3),
        "ranks": [dict({"rank": r, "candidates": int(share[r, 0]),
                        "cost_share": round(float(share[r, 1] / max(share[:, 1].sum(), 1e-9)), 4),
                        "run_s": round(float(share[r, 3]), 4), "idle_s": round(float(elapsed - share[r, 3]), 4)},
                       **{k: round(float(share[r, 4 + j]), 4) for j, k in enumerate(keys)}) for r in range(world)],
        "new_conv_problems_looked_up": len(sweep._warmed),
        "check": {"miou_min": round(float(miou.min()), 6), "miou_max": round(float(miou.max()), 6),
                  "confusion_sum_per_candidate": int(res["confusion"][0].sum().item())},
    }


FORWARD_DTYPE = "f16 forward (f32 accumulation), f32 loss and importance, int64 confusion"
FORWARD_DTYPE_FP32 = "f32 forward, f32 loss and importance, int64 confusion"
STRONG_WORKLOAD = ("DeepLabV3Plus_resnet magnitude_taylor sensitivity sweep, ALL candidates: geometry pass + cost model + "
                   "partition + launch plans + solver lookups + teacher forward + every candidate (gather, forward from the "
                   "first block its group touches, arg-max/confusion, KD-Lite loss) + all-reduce of the tables")


def strong_mode(args, sweep, images, labels, dev, rank, world, tune_s, taylor_s, build_s, first_s, lanes=1, lanes_s=0.0,
                dist_info=None):
    """The whole sweep, wall-clock, max over ranks; rank 0 prints the JSON line."""
    w = whole_sweep(sweep, images, labels, dev, rank, world)
    result = {
        "metric": "sensitivity-sweep candidates/sec",
        "value": w["candidates_per_s"], "unit": "candidates/s",
        "n_gpus": world, "steps": w["candidates"], "warmup": 0,
        "ms_per_step": round(1e3 * w["sweep_s"] / w["candidates"], 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": FORWARD_DTYPE_FP32 if args.fp32_forward else FORWARD_DTYPE, "data": "synthetic",
        "config": {"workload": STRONG_WORKLOAD,
                   "num_classes": CLASSES, "eval_batch": [BATCH, 3, SIZE, SIZE], "candidates": w["candidates"],
                   "groups": len(sweep.groups), "ratios": list(DEFAULT_RATIOS),
                   "parallelism": f"candidates sharded over {world} rank(s) by estimated cost (LPT); {lanes} candidate(s) "
                                  "in flight per rank"},
        "sweep_s": w["sweep_s"], "ranks": w["ranks"],
        "setup": {"miopen_find_db_warmup_s": round(tune_s, 1), "model_build_s": round(build_s, 3),
                  "unpruned_first_forward_s": round(first_s, 3),
                  "taylor_calibration_s": round(taylor_s, 3), "taylor_batches": 2, "taylor_sharded": world > 1,
                  "new_conv_problems_seen": w["new_conv_problems_looked_up"], "lanes": lanes,
                  "lanes_setup_s": round(lanes_s, 3)},
        "check": w["check"],
        "dist": dist_info,
    }
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


def rank_info(rank, local, dev):
    """What proves how many ranks the collective backend saw: backend name, its world size, the sum of a one per
    rank through the backend's all-reduce, and every rank's device (gathered through the backend)."""
    import torch.distributed as dist
    on_gpu = dev is not None and dev.type == "cuda"
    me = {"rank": rank, "local_rank": local, "pid": os.getpid(), "device": str(dev) if dev is not None else "cpu",
          "device_name": torch.cuda.get_device_name(dev) if on_gpu else "cpu",
          "visible_gpus": torch.cuda.device_count()}
    if on_gpu:
        props = torch.cuda.get_device_properties(dev)
        me["pci_bus_id"] = getattr(props, "pci_bus_id", None)
        me["uuid"] = str(getattr(props, "uuid", ""))
    if not (dist.is_available() and dist.is_initialized()):
        return {"backend": None, "world_size": 1, "all_reduce_of_ones": 1, "ranks": [me]}
    everyone = [None] * dist.get_world_size()
    dist.all_gather_object(everyone, me)
    one = torch.ones(1, dtype=torch.int64, device=dev if on_gpu else "cpu")
    sdist.all_reduce_sum_(one)
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "all_reduce_of_ones": int(one.item()),
            "ranks": everyone}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: time the whole sweep (all candidates shared by the ranks) instead of K per rank")
    ap.add_argument("--no-strong-block", action="store_true",
                    help="skip the whole-sweep (strong scaling) measurement that follows the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lanes", type=int, default=2,
                    help="candidates evaluated at a time per rank: copies of the network on streams of their own (1 = one after the other)")
    ap.add_argument("--no-kernel-table", action="store_true")
    ap.add_argument("--three-launch-head", action="store_true",
                    help="A/B run: up-sampling, arg-max histogram and loss of a candidate as three launches instead of one")
    ap.add_argument("--fp32-forward", action="store_true", help="forward without fp16 autocast")
    ap.add_argument("--no-graphs", action="store_true",
                    help="every launch of a candidate queued by the host (no captured HIP graphs of the unchanged blocks)")
    ap.add_argument("--plan-chunk", type=int, default=8,
                    help="--plans chunks: launch plans of this many candidates at a time, in front of the first of them")
    ap.add_argument("--plans", default="inloop", choices=["upfront", "inloop", "chunks"],
                    help="launch plans of the timed candidates: all before the first evaluation, or each in front of its "
                         "candidate's launches (both inside the clock)")
    ap.add_argument("--warm-mode", default="upfront", choices=["upfront", "none"],
                    help="how the solver lookups for new convolution shapes are taken out of the candidate loop")
    ap.add_argument("--no-tune", action="store_true", help="skip the MIOpen find-database warm-up of the backbone")
    ap.add_argument("--torch-forward", action="store_true",
                    help="A/B: keep the models' up-sampling, batch-norm, depthwise and 1x1 modules on their torch path")
    ap.add_argument("--no-prefix-reuse", action="store_true",
                    help="A/B: every candidate forwards the whole network instead of starting at the first block its group touches")
    ap.add_argument("--rank-probe", action="store_true",
                    help="start the ranks, report what the collective backend saw (one JSON line) and stop: no GPU work "
                         "beyond naming the devices; with no GPU visible the ranks meet over gloo")
    args = ap.parse_args()
    if args.rank_probe:
        backend = "nccl" if torch.cuda.device_count() > 0 else "gloo"
        rank, world, local = sdist.init_from_env(backend)
        if world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
        dev = torch.device("cuda", local) if torch.cuda.device_count() > 0 else None
        info = rank_info(rank, local, dev)
        if rank == 0:
            print(json.dumps({"probe": "ranks", "n_gpus": args.gpus, "dist": info}), flush=True)
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    if args.torch_forward:
        from spice_amd.nets import layers
        layers.set_library_forward(False)
        os.environ["SPICE_TORCH_FORWARD"] = "1"   # for the tuning child, which then tunes the 1x1 convolutions too

    # Set-up, before this process touches the GPU: a child process searches MIOpen solvers for the
    # un-pruned backbone's convolution shapes once and leaves them in this rank's find database.
    local_env = int(os.environ.get("LOCAL_RANK", "0"))
    n_dev = torch.cuda.device_count()
    tune_s = 0.0
    if not args.no_tune:
        from spice_amd import miopen_tuning
        try:
            tune_s = miopen_tuning.ensure_tuned("DeepLabV3Plus", "resnet50", CLASSES, BATCH, SIZE,
                                                "fp32" if args.fp32_forward else "fp16",
                                                device=local_env % max(n_dev, 1))
        except Exception as e:  # the sweep still runs, on MIOpen's heuristic solver picks
            print(f"[bench] MIOpen find-database warm-up skipped: {e}", file=sys.stderr, flush=True)
            tune_s = -1.0

    rank, world, local = sdist.init_from_env("nccl")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: leave the torchrun environment unset and this file starts "
                         "the ranks itself, or pass the same N to both")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    nat.lib()
    torch.backends.cudnn.benchmark = False
    dist_info = rank_info(rank, local, dev)
    if dist_info["world_size"] != args.gpus or dist_info["all_reduce_of_ones"] != args.gpus:
        raise SystemExit(f"the collective backend saw {dist_info['world_size']} rank(s), --gpus {args.gpus}")

    torch.manual_seed(1234)
    t_build = time.perf_counter()
    model = nets.create_model("DeepLabV3Plus", "resnet50", CLASSES).to(dev).eval()
    images, labels = build_inputs(dev)
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t_build
    scores, quotas, taylor_s = taylor_and_quotas(model, images, labels, dev, shard=args.strong)
    ignored = [m for m in model.segmentation_head.modules() if isinstance(m, torch.nn.Conv2d)]
    sweep = SensitivitySweep(model, CLASSES, DEFAULT_RATIOS, ignored_layers=ignored,
                             autocast_dtype=None if args.fp32_forward else torch.float16)
    n_cand = len(sweep.candidates)
    if args.three_launch_head:
        sweep.fused_head = False
    if args.no_prefix_reuse:
        sweep.prefix_reuse = False
    t_lanes = time.perf_counter()
    try:   # set-up like the construction of the sweep itself: copies of the network + their groups
        lanes = sweep.enable_lanes(args.lanes)
    except Exception as e:   # the sweep still runs, one candidate at a time
        print(f"[bench] lanes not available ({e}); one candidate at a time", file=sys.stderr, flush=True)
        lanes = sweep.enable_lanes(1)
    torch.cuda.synchronize()
    lanes_s = time.perf_counter() - t_lanes
    if args.strong:
        # per-process library initialisation, like the find-database warm-up: the first forward of the un-pruned
        # network at this geometry (MIOpen's first call per convolution problem, hipBLASLt plans, autocast copies)
        t_first = time.perf_counter()
        with sweep.session():
            sweep.forward(images)
        torch.cuda.synchronize()
        first_s = time.perf_counter() - t_first
        strong_mode(args, sweep, images, labels, dev, rank, world, tune_s, taylor_s, build_s, first_s, lanes, lanes_s, dist_info)
        return
    parts = sdist.partition_by_cost(sweep.candidate_costs(images), world)
    mine = parts[rank]
    # the K timed candidates are spread evenly over the rank's share (which runs from the stem's group to the
    # decoder's): a candidate's cost depends on where its group sits in the network, so the first K of the list
    # would time the early groups only.  The W warm-up candidates are spread over the rest of the share.
    n_mine = len(mine)
    timed = ([mine[(j * n_mine) // args.steps] for j in range(args.steps)] if args.steps <= n_mine
             else [mine[j % n_mine] for j in range(args.steps)])
    rest = [c for c in mine if c not in set(timed)] or mine
    warm = ([rest[(j * len(rest)) // args.warmup] for j in range(args.warmup)] if 0 < args.warmup <= len(rest)
            else [rest[j % len(rest)] for j in range(args.warmup)])
    todo = warm + timed
    total_steps = args.warmup + args.steps
    warm_list = sorted(set(warm))
    timed_list = [i for i in dict.fromkeys(timed) if i not in set(warm_list)]
    sweep.prepare(warm_list)
    sweep.warm_shapes(images, warm_list)

    host_s = [0.0]
    prep_split = [0.0]
    # SPICE_BENCH_MARK=1 (profiling runs): one launch of a kernel the sweep never uses on either side of the timed region
    mark = torch.ones(4, device=dev) if os.environ.get("SPICE_BENCH_MARK") == "1" else None
    mark_one = torch.full((1,), 2.0, device=dev)
    # plans in front of each candidate's launches unless a candidate may bring MIOpen a new problem (then plans and
    # look-ups up front, as SensitivitySweep.run does)
    plans_mode = "upfront" if sweep.may_need_solver_lookups(images) else args.plans

    def timed_region(first_time):
        """K candidates, wall-clock, barrier and device synchronisation on both sides; the first time it holds ALL
        per-candidate work of its K candidates: launch plans and their upload (in front of each candidate's launches,
        or all up front together with the solver lookups for new convolution shapes where the network has
        convolutions on MIOpen), evaluation, the all-reduce of the tables."""
        table.zero_()   # (outside the clock) every row is its owner's alone before the all-reduce sums the ranks' tables
        kd.zero_()
        sweep.fork_lanes()
        sdist.barrier()
        torch.cuda.synchronize()
        if first_time and mark is not None:
            ops.scale_inplace(mark, mark_one)   # (a kernel no sweep launches: tools/stepbreakdown.py finds the region by it)
        t0 = time.perf_counter()
        prep = 0.0
        if first_time and plans_mode == "upfront":
            sweep.prepare(timed_list)
            prep_split[0] = time.perf_counter() - t0
            if args.warm_mode == "upfront":
                sweep.warm_shapes(images, timed_list)
            prep = time.perf_counter() - t0
        moved = 0
        for i in range(args.warmup, total_steps):
            if first_time and plans_mode == "chunks" and (i - args.warmup) % args.plan_chunk == 0:
                sweep.prepare(todo[i:min(i + args.plan_chunk, total_steps)])
            with sweep.lane_context(todo[i]):
                table[todo[i]].zero_()
            moved += sweep.evaluate(todo[i], images, labels, teacher, table[todo[i]], kd[todo[i]])
        host_s[0] = time.perf_counter() - t0   # the host has queued everything; what is left is the device catching up
        sweep.join_lanes()
        sdist.all_reduce_sum_(table)
        sdist.all_reduce_sum_(kd)
        torch.cuda.synchronize()
        sdist.barrier()
        dt = time.perf_counter() - t0
        if first_time and mark is not None:
            ops.scale_inplace(mark, mark_one)
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        sdist.all_reduce_max_(t)
        return float(t.item()), prep, moved

    with sweep.session():
        # the un-pruned network's forward: the teacher's logits and, with prefix reuse, every block's activations
        # (recorded either way, so that the A/B region behind the timed one can run both forms)
        want_reuse, sweep.prefix_reuse = sweep.prefix_reuse, True
        # one-time set-up like the launch plans of the un-pruned network itself: its forward from every block on as a
        # captured HIP graph - captured on ANOTHER batch of the same geometry; they serve every batch and every sweep
        # on this geometry until the session ends (teacher_forward writes a new batch's activations to the same addresses)
        graph_info = {"graphs": 0, "seconds": 0.0}
        if want_reuse and not args.no_graphs:
            scratch = torch.randn_like(images)
            sweep.teacher_forward(scratch)
            graph_info = sweep.capture_suffix_graphs(scratch)
            del scratch
        teacher = sweep.teacher_forward(images).float().contiguous()
        sweep.prefix_reuse = want_reuse
        table = torch.zeros(n_cand, CLASSES, CLASSES, dtype=torch.int64, device=dev)
        kd = torch.zeros(n_cand, 3, dtype=torch.float32, device=dev)

        sweep.fork_lanes()
        for i in range(args.warmup):
            with sweep.lane_context(todo[i]):
                table[todo[i]].zero_()
            sweep.evaluate(todo[i], images, labels, teacher, table[todo[i]], kd[todo[i]])
        sweep.join_lanes()
        sweep.kd_events = []
        reused0 = sweep.units_reused
        elapsed, prep_s, moved = timed_region(True)
        host_first = host_s[0]
        units_reused = (sweep.units_reused - reused0) / max(args.steps, 1)
        kd_events = sweep.kd_events
        sweep.kd_events = None
        # the same K candidates once more the other way (whole forward / prefix taken from the teacher pass), launch
        # plans and solver lookups already made: the figure rounds before prefix reuse reported, kept comparable
        table_a = table[timed].clone()
        sweep.prefix_reuse = not sweep.prefix_reuse
        elapsed_other, _, _ = timed_region(False)
        sweep.prefix_reuse = not sweep.prefix_reuse
        same_tables = bool(torch.equal(table_a, table[timed]))
        elapsed_again, _, _ = timed_region(False)   # and the timed form again without its plans: like for like
        host_again = host_s[0]
        graph_launches = sum(l.graph_launches for l in sweep._lanes)
        elapsed_eager, same_tables_eager = None, None
        if graph_info["graphs"]:
            # ... and with every launch queued by the host (no captured graphs): the figure earlier rounds reported
            table_g = table[timed].clone()
            sweep.suffix_graphs = False
            elapsed_eager, _, _ = timed_region(False)
            sweep.suffix_graphs = True
            same_tables_eager = bool(torch.equal(table_g, table[timed]))
    n_warm = len(sweep._warmed)
    kd_events_alone = kd_events
    head_lows = []
    real_head = ops.sweep_head

    def spy_head(low, *a, **k):   # the head's inputs of the candidates below, for the back-to-back launches behind them
        head_lows.append(low.clone())
        return real_head(low, *a, **k)
    ops.sweep_head = spy_head
    if lanes > 1:
        # With several candidates in flight the loss kernel shares the device with the other lane's convolutions and
        # its duration says how they split the memory system, not how good the kernel is: the same K candidates once
        # more, one at a time on the current stream, for the duration of the kernel by itself.
        sweep.kd_events = []
        with sweep.session():
            for i in range(args.warmup, total_steps):
                table[todo[i]].zero_()
                sweep.evaluate(todo[i], images, labels, teacher, table[todo[i]], kd[todo[i]], first_lane=True)
        torch.cuda.synchronize()
        kd_events_alone, sweep.kd_events = sweep.kd_events, None
    ops.sweep_head = real_head
    kd_ms_shared = float(np.mean([a.elapsed_time(b) for a, b in kd_events])) if kd_events else float("nan")
    kd_ms = float(np.mean([a.elapsed_time(b) for a, b in kd_events_alone])) if kd_events_alone else float("nan")
    # the kernel those events bracket: the candidate's tail in one launch (up-sampling + loss + arg-max histogram over
    # the head's small fp16 output, the teacher's logits and the labels), or the loss kernel alone on full-size logits
    one_launch_head = any(l.head_launches for l in sweep._lanes)
    if one_launch_head:
        prepared = sweep._teacher_prepared is not None   # teacher's softmax taken once per sweep: + 4 bytes per pixel
        kd_bytes = (BATCH * CLASSES * (SIZE // 4) * (SIZE // 4) * 2 + BATCH * SIZE * SIZE * (CLASSES * 4 + 8)
                    + (BATCH * SIZE * SIZE * 4 if prepared else 0))
        # the class-innermost form takes the prepared teacher at the integer temperatures (SPICE_HEAD_FORM=1: the first form)
        second_form = prepared and os.environ.get("SPICE_HEAD_FORM", "0") != "1"
        if second_form:
            kd_key = "spice::kdlite_head2_kernel<28, 4, true, 512>"
            kd_name = "spice::kdlite_head2_kernel<28,4,true,512>"
        else:
            kd_key = f"spice::kdlite_head_kernel<4, 4, {str(prepared).lower()}, true>"
            kd_name = f"spice::kdlite_head_kernel<4,4,{str(prepared).lower()},true>"
        kd_kernel = (f"{kd_name} (one launch per candidate: bilinear "
                     "up-sampling of the head's fp16 logits + KD-Lite loss + arg-max + confusion histogram"
                     + (", against the teacher's probabilities prepared once per sweep" if prepared else "") +
                     "; replaces upsample_bilinear_kernel, argmax_confusion_kernel and kdlite_main_kernel = 128 us "
                     "and 550 MB of traffic per candidate)")
        kd_note = ("bound by vector arithmetic and by the dependent trips of its two tails (per-workgroup totals -> ticket -> "
                   "last arriver), not by memory: an exponential, the six-operation interpolation and the arg-max per logit; "
                   "the loss kernel alone on full-size logits reaches 0.66-0.71 (`kernels` table)")
    else:
        kd_bytes = BATCH * SIZE * SIZE * (2 * CLASSES * 4 + 8)
        kd_kernel = "spice::kdlite_main_kernel<4,false,false,false> (one launch per spice_kdlite_forward call)"
        kd_key = "spice::kdlite_main_kernel<4, false, false, false>"
        kd_note = None
    # A pair of events per launch adds the events' own trips to the figure (5-7 us on a 50 us kernel).  The duration
    # the roofline uses: the head launches of those K candidates again, on their own fp16 head outputs, back to back
    # between ONE pair of events (the teacher's probabilities rotate over three copies, 3 x 176 MB, so that no launch
    # finds them in the 256 MB memory-side cache); median of five such runs / K.
    kd_ms_pairs = kd_ms
    back_to_back = False
    if one_launch_head and head_lows and sweep._teacher_prepared is not None and lanes > 1:
        pr = [sweep._teacher_prepared] + [tuple(t.clone() for t in sweep._teacher_prepared[:2]) for _ in range(2)]
        scratch_conf = torch.zeros(CLASSES, CLASSES, dtype=torch.int64, device=dev)
        scratch_out = torch.zeros(3, dtype=torch.float32, device=dev)

        def heads_once():
            for j, low in enumerate(head_lows):
                t = pr[j % 3]
                real_head(low, t[0], labels, scratch_conf, align_corners=True, out=scratch_out, scale_factor=4.0,
                          teacher_hneg=t[1], **sweep.kd)
        heads_once()
        torch.cuda.synchronize()
        runs = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            heads_once()
            b.record()
            b.synchronize()
            runs.append(a.elapsed_time(b) / len(head_lows))
        kd_ms = float(np.median(runs))
        back_to_back = True
        del pr
    achieved = kd_bytes / kd_ms / 1e6
    result = {
        "metric": "sensitivity-sweep candidates/sec",
        "value": round(world * args.steps / elapsed, 3),
        "unit": "candidates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": FORWARD_DTYPE_FP32 if args.fp32_forward else FORWARD_DTYPE, "data": "synthetic",
        "config": {"workload": "DeepLabV3Plus_resnet magnitude_taylor sensitivity sweep: one (channel group x "
                               "ratio) candidate per step = gather + forward + arg-max/confusion + KD-Lite loss; "
                               + ("the forward starts at the first block the candidate's group touches, on the un-pruned "
                                  "network's activations of the same sweep (prefix reuse); the K timed candidates are spread "
                                  "evenly over the rank's share"
                                  + ("; the unchanged blocks behind the group's last one run as one launch of a HIP graph "
                                     "captured on the un-pruned network before the clock starts" if graph_info["graphs"] else "")
                                  if sweep.prefix_reuse else
                                  "every candidate forwards the whole network (--no-prefix-reuse); the K timed candidates are "
                                  "spread evenly over the rank's share"),
                   "prefix_reuse": sweep.prefix_reuse,
                   "num_classes": CLASSES, "eval_batch": [BATCH, 3, SIZE, SIZE], "candidates": n_cand,
                   "groups": len(sweep.groups), "ratios": list(DEFAULT_RATIOS),
                   "forward": "fp32" if args.fp32_forward else (
                       "fp16 autocast, every module on torch's kernels (A/B run)" if args.torch_forward else
                       "fp16 autocast; 1x1 and 3x3 convolutions with their batch-norm (+ residual + ReLU) in one launch of the "
                       "library's matrix-core kernel each, the stem's 7x7 convolution through the same kernel, stem batch-norm + "
                       "max-pooling, up-sampling and depthwise 3x3 on the library's kernels; three plain 1x1 products on "
                       "hipBLASLt; no MIOpen convolution"),
                   "head": ("one launch: up-sampling of the head's logits + KD-Lite loss + arg-max/confusion"
                            if one_launch_head else "three launches: up-sampling, arg-max/confusion, KD-Lite loss"),
                   "kd": {"T": 4.0, "alpha": 0.7},
                   "parallelism": f"candidates sharded over {world} rank(s); {lanes} candidate(s) in flight per rank "
                                  "(copies of the network on streams of their own, one host thread)"},
        "roofline": {"bound": "hbm", "kernel": kd_kernel, "achieved": round(achieved, 1), "peak": PEAK_HBM_GBPS,
                     "unit": "GB/s", "frac": round(achieved / PEAK_HBM_GBPS, 4),
                     "traffic": pmc_traffic(kd_key)[0],
                     "traffic_source": pmc_traffic(kd_key)[1],
                     "rocprofv3_median_us": rocprof_median_us(kd_key)[0],
                     "rocprofv3_frac": (round(kd_bytes / rocprof_median_us(kd_key)[0] / 1e3 / PEAK_HBM_GBPS, 4)
                                        if rocprof_median_us(kd_key)[0] else None),
                     "rocprofv3_source": rocprof_median_us(kd_key)[1],
                     **({"note": kd_note} if kd_note else {}),
                     "algorithmic_bytes_per_launch": kd_bytes, "avg_launch_us": round(1e3 * kd_ms, 2),
                     "launches_timed": len(head_lows) * 5 if back_to_back else len(kd_events_alone),
                     "measured": ("HIP events round the K launches of the timed region" if lanes == 1 else
                                  "behind the timed region: the head launches of the same K candidates (their own fp16 "
                                  "head outputs, the batch's labels and prepared teacher, the teacher rotating over three "
                                  "copies) back to back between ONE pair of HIP events, median of 5 runs / K" if back_to_back else
                                  "HIP events round the launches of the same K candidates evaluated once more behind the "
                                  "timed region, one candidate on the device at a time (as `--lanes 1` times them "
                                  "inside it)"),
                     **({"one_event_pair_per_launch": {
                         "avg_launch_us": round(1e3 * kd_ms_pairs, 2),
                         "frac": round(kd_bytes / kd_ms_pairs / 1e6 / PEAK_HBM_GBPS, 4),
                         "note": "the same K candidates evaluated once more, one at a time, each head launch between "
                                 "its own pair of events: the events' own trips are in the figure"}} if back_to_back else {}),
                     "in_timed_region": {"avg_launch_us": round(1e3 * kd_ms_shared, 2),
                                         "frac": round(kd_bytes / kd_ms_shared / 1e6 / PEAK_HBM_GBPS, 4),
                                         "candidates_in_flight": lanes,
                                         "note": "the kernel's launches overlap the other lane's convolutions and share "
                                                 "the memory system with them"}},
        "setup": {"miopen_find_db_warmup_s": round(tune_s, 1), "timed_prepare_s": round(prep_s, 4), "timed_prepare_plans_s": round(prep_split[0], 4),
                  "plans": {"upfront": "all before the first evaluation", "chunks": f"{args.plan_chunk} candidates at a time",
                            "inloop": "each in front of its candidate's launches, uploaded by a launch "
                                      "(spice_gather_upload)"}[plans_mode],
                  "host_queued_everything_after_ms_per_step": {"timed_region": round(1e3 * host_first / args.steps, 3),
                                                               "same_candidates_again": round(1e3 * host_again / args.steps, 3),
                                                               "note": "wall time until the host has launched the region's last "
                                                                       "kernel, per candidate; close to ms_per_step = the host, "
                                                                       "not the device, sets the pace"}, "warm_mode": args.warm_mode,
                  "new_conv_problems_seen": n_warm, "taylor_calibration_s": round(taylor_s, 3), "taylor_batches": 2,
                  "lanes": lanes, "lanes_setup_s": round(lanes_s, 3),
                  "quota_layers": len(quotas), "gather_MB_per_candidate": round(moved / max(args.steps, 1) / 1e6, 2)},
    }
    # the same K candidates the other way, and the timed form again, both with launch plans and lookups already made
    k_other = round(world * args.steps / elapsed_other, 3)
    k_again = round(world * args.steps / elapsed_again, 3)
    result["value_full_forward" if sweep.prefix_reuse else "value_prefix_reuse"] = k_other
    result["value_without_plans" ] = k_again
    result["prefix_reuse"] = {
        "on_in_timed_region": sweep.prefix_reuse, "units_reused_per_candidate": round(units_reused, 2),
        "units": len(sweep._tape) if sweep._tape is not None else 0,
        "activations_MB": round(sweep._tape.nbytes() / 1e6, 1) if sweep._tape is not None else 0.0,
        "same_K_candidates_again": {"prefix_reuse": k_again if sweep.prefix_reuse else k_other,
                                    "full_forward": k_other if sweep.prefix_reuse else k_again,
                                    "note": "launch plans and solver lookups already made in the timed region; value holds them"},
        "confusion_matrices_identical_both_ways": same_tables}
    result["suffix_graphs"] = {
        "graphs_captured": graph_info["graphs"], "capture_s_outside_the_clock": round(graph_info["seconds"], 3),
        "lanes": graph_info.get("lanes", lanes), "graph_launches_in_all_regions": graph_launches,
        "note": "the un-pruned network's forward from every block on to the head's convolution, captured once per lane "
                "on another batch of the same geometry before the timed batch's teacher forward; a candidate runs the "
                "blocks its group touches eagerly and the rest as one graph launch (same kernels, same shapes: identical "
                "tables).  One-time set-up per geometry, valid for every batch and every sweep until the autocast session "
                "ends; a single 414-candidate sweep that had to capture them itself would get the capture time back about "
                "evenly (SensitivitySweep.run does not capture by default).  value_without_suffix_graphs = the same K "
                "candidates, plans made, every launch queued by the host",
        **({"value_without_suffix_graphs": round(world * args.steps / elapsed_eager, 3),
            "confusion_matrices_identical_both_ways": same_tables_eager} if elapsed_eager else {})}
    result["dist"] = dist_info
    if rank == 0 and not args.torch_forward and not args.fp32_forward:
        # candidates of the first lane: one after the other on the current stream
        probe = [c for c in dict.fromkeys(todo[args.warmup:]) if sweep.lane_index(c) == 0][::4][:8]
        result["forward_roofline"] = forward_probe(sweep, probe, images, labels, teacher, table, kd)
    if not args.no_strong_block:
        # The strong-scaling figure beside the weak one: the whole sweep, all candidates shared by the ranks, on a
        # fresh SensitivitySweep (new launch plans).  It runs AFTER the timed region above, so the convolution
        # problems of the candidates evaluated there are already in MIOpen's in-process cache: their look-ups here
        # cost tens of microseconds instead of ~0.4 ms each (at most 70 candidates' worth per rank).
        fresh = SensitivitySweep(model, CLASSES, DEFAULT_RATIOS, ignored_layers=ignored,
                                 autocast_dtype=None if args.fp32_forward else torch.float16)
        fresh.enable_lanes(lanes)
        fresh.prefix_reuse = sweep.prefix_reuse
        w = whole_sweep(fresh, images, labels, dev, rank, world)
        w["scaling"] = "strong"
        w["workload"] = STRONG_WORKLOAD
        w["note"] = "measured after the weak-scaling region of the same process; candidates/s = candidates / sweep_s"
        result["strong_sweep"] = w
    if rank == 0 and world == 1:
        if not args.no_kernel_table:
            result["kernels"] = kernel_table(dev)
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_candidate_baseline(model, sweep, images, labels, ignored)
            result["cpu_baseline"]["gpu_over_cpu"] = round(result["value"] / result["cpu_baseline"]["value"], 1)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
