"""Per-kernel micro-benchmark: HIP-event duration of every hot-path kernel at the benchmark
sizes, cycling through enough distinct buffers to exceed the 256 MiB Infinity Cache.
Usage: python tools/kbench.py [--iters 20] [--only name,...]"""
import argparse
import json
import sys
import os

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spice_amd import _native as nat
from spice_amd import ops, synth

PEAK = 8000.0  # GB/s, MI355X HBM3E


def timed(fn_list, iters, warmup=5):
    """fn_list: callables cycled round-robin (one per rotating buffer set). Returns ms list."""
    k = len(fn_list)
    for i in range(warmup):
        fn_list[i % k]()
    torch.cuda.synchronize()
    out = []
    for i in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn_list[i % k]()
        b.record()
        b.synchronize()
        out.append(a.elapsed_time(b))
    return out


def report(name, ms, nbytes):
    med = float(np.median(ms))
    gbs = nbytes / med / 1e6
    print(json.dumps({"kernel": name, "ms_median": round(med, 4), "ms_min": round(min(ms), 4),
                      "algorithmic_MB": round(nbytes / 1e6, 1), "GBps": round(gbs, 1),
                      "frac_of_8TBps": round(gbs / PEAK, 3)}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--only", type=str, default="")
    ap.add_argument("--labels", type=str, default="uniform", choices=["uniform", "blocky"],
                    help="label map of the head rows: uniform random, or 64x64 blocks (as real label maps, and as bench.py)")
    ap.add_argument("--imp-targets", type=str, default="")
    args = ap.parse_args()
    only = set(x for x in args.only.split(",") if x)
    dev = torch.device("cuda:0")
    want = lambda n: not only or n in only

    cols_only = "gather_cols" in only   # the rows + columns case alone (its own kernel-name statistics)
    if want("importance") or want("taylor") or want("gather") or cols_only:
        shapes = synth.deeplab_resnet50_conv_shapes(21)
        sizes = [o * i * kh * kw for o, i, kh, kw in shapes]
        offs, pos = [], 0
        for s in sizes:
            offs.append(pos)
            pos += (s + 3) // 4 * 4
        total = pos
        base = torch.from_numpy(synth.normal(1, 1, total)).to(dev)
        layers, opos = [], 0
        for (o, i, kh, kw), off in zip(shapes, offs):
            layers.append(ops.RowsLayout(off, off, opos, o, i * kh * kw))
            opos += o
        plan = ops.RowsPlan(layers, dev)
        copies = 4
        if args.imp_targets:
            w_t = [base.clone() for _ in range(copies)]
            sc_t = torch.zeros(opos, device=dev)
            for tgt in args.imp_targets.split(","):
                os.environ["SPICE_IMP_TARGET_ELEMS"] = tgt
                pl = ops.RowsPlan(layers, dev)
                fns = [lambda w=w: pl.run([w], (), sc_t, nat.IMP_L2, False) for w in w_t]
                report(f"importance_l2 target={tgt} ({pl.n_items} items)", timed(fns, args.iters), pl.algorithmic_bytes(nat.IMP_L2))
            os.environ.pop("SPICE_IMP_TARGET_ELEMS")
            del w_t
        ws = [base.clone() for _ in range(copies)]
        gs = [base.clone() * 1e-3 for _ in range(copies)]
        score = torch.zeros(opos, device=dev)
        if want("importance"):
            for mode, nm in ((nat.IMP_L2, "importance_l2"), (nat.IMP_L1, "importance_l1")):
                fns = [lambda w=w: plan.run([w], (), score, mode, False) for w in ws]
                report(nm + f" ({plan.n_items} items)", timed(fns, args.iters), plan.algorithmic_bytes(mode))
        if want("taylor"):
            fns = [lambda w=w, g=g: plan.run([w], [g], score, nat.IMP_TAYLOR, True)
                   for w, g in zip(ws, gs)]
            report("importance_taylor", timed(fns, args.iters), plan.algorithmic_bytes(nat.IMP_TAYLOR))
        if want("gather") and not cols_only:
            for label, pick in (("gather_out50", lambda o: np.arange(0, o, 2, dtype=np.int64)),
                                ("gather_first_half", lambda o: np.arange(0, (o + 1) // 2, dtype=np.int64))):
                specs, idx, dpos, ipos = [], [], 0, 0
                for (o, i, kh, kw), off in zip(shapes, offs):
                    ko = pick(o)
                    specs.append(ops.GatherSpec(off, dpos, ipos, -1, len(ko), i, i, kh * kw))
                    idx.append(ko)
                    ipos += len(ko)
                    dpos += (len(ko) * i * kh * kw + 3) // 4 * 4
                gplan = ops.GatherPlan(specs, dev)
                idx_t = torch.from_numpy(np.concatenate(idx)).to(dev)
                dsts = [torch.empty(dpos, device=dev) for _ in range(copies)]
                fns = [lambda w=w, d=d: gplan.run([w], d, idx_t) for w, d in zip(ws, dsts)]
                report(f"{label} ({gplan.n_items} items)", timed(fns, args.iters), gplan.algorithmic_bytes())
                del dsts
        if want("gather") or cols_only:
            # both lists, as in a whole-model pruning step: every second out-channel AND in-channel
            specs, idx, dpos, ipos = [], [], 0, 0
            for (o, i, kh, kw), off in zip(shapes, offs):
                ko = np.arange(0, o, 2, dtype=np.int64)
                ki = np.arange(0, i, 2, dtype=np.int64) if i > 3 else np.arange(i, dtype=np.int64)
                specs.append(ops.GatherSpec(off, dpos, ipos, ipos + len(ko), len(ko), len(ki), i, kh * kw))
                idx += [ko, ki]
                ipos += len(ko) + len(ki)
                dpos += (len(ko) * len(ki) * kh * kw + 3) // 4 * 4
            gplan = ops.GatherPlan(specs, dev)
            idx_t = torch.from_numpy(np.concatenate(idx)).to(dev)
            dsts = [torch.empty(dpos, device=dev) for _ in range(copies)]
            fns = [lambda w=w, d=d: gplan.run([w], d, idx_t) for w, d in zip(ws, dsts)]
            report(f"gather_out50_in50 ({gplan.n_items} items)", timed(fns, args.iters), gplan.algorithmic_bytes())
            del dsts
        del ws, gs

    if want("kdlite") or want("kdlite_grad"):
        n, c, h, w = 8, 21, 512, 512
        s, t = synth.logits_pair(1234, n, c, h, w)
        lab = synth.label_map(1234, n, h, w, c)
        sets = []
        for k in range(2):
            sets.append((torch.from_numpy(s).to(dev) + k, torch.from_numpy(t).to(dev) + k, torch.from_numpy(lab).to(dev).clone()))
        P = n * h * w
        if want("kdlite"):
            fns = [lambda a=a: ops.kdlite_forward(a[0], a[1], a[2], temperature=4.0, alpha=0.7) for a in sets]
            report("kdlite_fwd (incl. finalize)", timed(fns, args.iters), P * (2 * c * 4 + 8))
            fns = [lambda a=a: ops.kdlite_forward(a[0], a[1], a[2], temperature=4.0, alpha=0.7, boundary_weight=2.0,
                                                  confidence_weight=True) for a in sets]
            report("kdlite_fwd_boundary_conf", timed(fns, args.iters), P * (2 * c * 4 + 8))
        if want("kdlite_grad"):
            fns = [lambda a=a: ops.kdlite_forward(a[0], a[1], a[2], temperature=4.0, alpha=0.7, want_grad=True) for a in sets]
            report("kdlite_fwd_bwd (prepass+main+finalize, incl. grad alloc)", timed(fns, args.iters), P * (3 * c * 4 + 8))
        del sets

    if "kdlite5" in only:   # opt-in: it launches the same kernel names at another size, which a per-name PMC summary would mix up
        # the reference's own default geometry: 5 classes, 576x576, batch 32 (pruning/config.py)
        n, c, h, w = 32, 5, 576, 576
        P = n * h * w
        sets = []
        for k in range(2):
            s5 = 3.0 * torch.randn(n, c, h, w, device=dev)
            lab5 = torch.randint(0, c, (n, h, w), device=dev)
            lab5[torch.rand(n, h, w, device=dev) < 0.05] = 255
            sets.append((s5, s5 + torch.randn_like(s5), lab5))
        cw = torch.tensor([0.5, 1.0, 1.5, 2.0, 1.0], device=dev)
        fns = [lambda a=a: ops.kdlite_forward(a[0], a[1], a[2], temperature=4.0, alpha=0.7, class_weight=cw) for a in sets]
        report("kdlite_fwd C=5 [32,5,576,576]", timed(fns, args.iters), P * (2 * c * 4 + 8))
        fns = [lambda a=a: ops.kdlite_forward(a[0], a[1], a[2], temperature=4.0, alpha=0.7, class_weight=cw, want_grad=True) for a in sets]
        report("kdlite_fwd_bwd C=5 (incl. grad alloc)", timed(fns, args.iters), P * (3 * c * 4 + 8))
        lg = [a[0] for a in sets]
        conf = torch.zeros(c, c, dtype=torch.int64, device=dev)
        fns = [lambda a=a: ops.argmax_confusion_update(conf, a[0], a[2], c) for a in sets]
        report("argmax_confusion fp32 C=5", timed(fns, args.iters), P * (c * 4 + 8))
        del sets

    if want("confusion") or "confusion_small" in only:
        for (n, h, w, c, copies, tag) in ((8, 512, 512, 21, 10, "cfg1"), (32, 1024, 2048, 19, 1, "cfg5")):
            if "confusion_small" in only and tag != "cfg1":
                continue
            for kind in ("uniform", "blocky"):
                lab = synth.label_map(5, n, h, w, c, kind, ignore_share=0.10 if tag == "cfg5" else 0.05)
                pred = synth.predictions_from(lab, 5, c)
                tl, tp = torch.from_numpy(lab).to(dev), torch.from_numpy(pred).to(dev)
                sets = [(tp.clone(), tl.clone()) for _ in range(copies)]
                conf = torch.zeros(c, c, dtype=torch.int64, device=dev)
                fns = [lambda a=a: ops.confusion_update(conf, a[0], a[1], c) for a in sets]
                report(f"confusion_{tag}_{kind} (one launch)", timed(fns, args.iters), n * h * w * 16)
                del sets

    if want("argmax"):
        n, c, h, w = 8, 21, 512, 512
        s, _ = synth.logits_pair(1234, n, c, h, w)
        lab = torch.from_numpy(synth.label_map(1234, n, h, w, c)).to(dev)
        for dt, sz in ((torch.float32, 4), (torch.float16, 2)):
            sets = [torch.from_numpy(s).to(dev).to(dt) + k for k in range(3)]
            conf = torch.zeros(c, c, dtype=torch.int64, device=dev)
            fns = [lambda a=a: ops.argmax_confusion_update(conf, a, lab, c) for a in sets]
            report(f"argmax_confusion_{str(dt)[6:]}", timed(fns, args.iters), n * h * w * (c * sz + 8))

    if want("head"):
        # a sweep candidate's tail at the benchmark geometry: head output [8,21,128,128] fp16, teacher logits and
        # labels at 512x512; one launch against the three it replaces (up-sampling, arg-max histogram, loss)
        n, c, h, w = 8, 21, 512, 512
        _, t = synth.logits_pair(77, n, c, h, w)
        lab = torch.from_numpy(synth.label_map(77, n, h, w, c, args.labels)).to(dev)
        teachers = [torch.from_numpy(t).to(dev) + k for k in range(3)]
        lows = [torch.randn(n, c, h // 4, w // 4, device=dev).half() for _ in range(3)]
        conf = torch.zeros(c, c, dtype=torch.int64, device=dev)
        nbytes = n * c * (h // 4) * (w // 4) * 2 + n * h * w * (c * 4 + 8)
        fns = [lambda a=a, b=b: ops.sweep_head(a, b, lab, conf, align_corners=True, scale_factor=4.0, temperature=4.0,
                                               alpha=0.7) for a, b in zip(lows, teachers)]
        report("sweep_head (up-sampling + loss + arg-max histogram, one launch)", timed(fns, args.iters), nbytes)

        prepared = [ops.kd_teacher_prepare(b, 4.0) for b in teachers]
        fns = [lambda a=a, pr=pr: ops.sweep_head(a, pr[0], lab, conf, align_corners=True, scale_factor=4.0, temperature=4.0,
                                                 alpha=0.7, teacher_hneg=pr[1]) for a, pr in zip(lows, prepared)]
        report("sweep_head, teacher's softmax prepared once (one launch)", timed(fns, args.iters), nbytes + n * h * w * 4)
        del prepared

        fns = [lambda a=a: ops.upsample_argmax_confusion(conf, a, lab, align_corners=True, scale_factor=4.0) for a in lows]
        report("upsample + argmax_confusion in one launch (evaluation loops)", timed(fns, args.iters),
               n * c * (h // 4) * (w // 4) * 2 + n * h * w * 8)

        def two(a):
            ops.argmax_confusion_update(conf, ops.upsample_bilinear(a, scale_factor=4.0, align_corners=True,
                                                                    out_dtype=torch.float32), lab, c)
        fns = [lambda a=a: two(a) for a in lows]
        report("the same in two launches (up-sampling, arg-max histogram)", timed(fns, args.iters),
               n * c * (h // 4) * (w // 4) * 2 + n * h * w * 8)

        def three(a, b):
            up = ops.upsample_bilinear(a, scale_factor=4.0, align_corners=True, out_dtype=torch.float32)
            ops.argmax_confusion_update(conf, up, lab, c)
            ops.kdlite_forward(up, b, lab, temperature=4.0, alpha=0.7)
        fns = [lambda a=a, b=b: three(a, b) for a, b in zip(lows, teachers)]
        report("the same in three launches (up-sampling, arg-max histogram, loss)", timed(fns, args.iters), nbytes)
        del teachers, lows

    if want("upsample"):
        # the two x4 steps of DeepLabV3+ at the benchmark geometry: head [8,21,128,128] fp16 -> fp32 logits,
        # decoder [8,256,32,32] fp16 -> fp16; torch's own kernel beside them
        import torch.nn.functional as F
        for tag, shp, out_dt in (("head_x4_f16_to_f32", (8, 21, 128, 128), torch.float32),
                                 ("decoder_x4_f16_to_f16", (8, 256, 32, 32), torch.float16)):
            xs = [torch.randn(*shp, device=dev).half() for _ in range(3)]
            outs = []   # the caching allocator hands the buffers out in rotation
            nbytes = xs[0].numel() * 2 + xs[0].numel() * 16 * (4 if out_dt == torch.float32 else 2)
            fns = [lambda a=a: outs.append(ops.upsample_bilinear(a, scale_factor=4.0, align_corners=True, out_dtype=out_dt)) or
                   (len(outs) > 2 and outs.pop(0)) for a in xs]
            report(f"upsample_{tag}", timed(fns, args.iters), nbytes)
            outs.clear()
            fns = [lambda a=a: outs.append(F.interpolate(a.float(), scale_factor=4.0, mode="bilinear", align_corners=True)) or
                   (len(outs) > 2 and outs.pop(0)) for a in xs]
            report(f"torch_upsample_{tag} (cast + fp32 kernel)", timed(fns, args.iters), nbytes)
            outs.clear()

    if want("depthwise"):
        import torch.nn.functional as F
        for shp in ((8, 304, 128, 128), (8, 256, 32, 32)):
            xs = [torch.randn(*shp, device=dev).half() for _ in range(3)]
            wt = torch.randn(shp[1], 1, 3, 3, device=dev)
            w16 = wt.half()
            outs = []
            nbytes = xs[0].numel() * 4
            fns = [lambda a=a: outs.append(ops.depthwise_conv3x3(a, wt)) or (len(outs) > 2 and outs.pop(0)) for a in xs]
            report(f"depthwise3x3 {shp}", timed(fns, args.iters), nbytes)
            outs.clear()
            fns = [lambda a=a: outs.append(F.conv2d(a, w16, None, 1, 1, 1, shp[1])) or (len(outs) > 2 and outs.pop(0)) for a in xs]
            report(f"torch conv2d depthwise {shp}", timed(fns, args.iters), nbytes)
            outs.clear()

    if want("depthwise") or "depthwise_dilated" in only:
        # the three dilated depthwise branches of the ASPP block: [8, 2048, 32, 32], dilation 12 / 24 / 36
        shp = (8, 2048, 32, 32)
        xs = [torch.randn(*shp, device=dev).half() for _ in range(5)]
        wt = torch.randn(shp[1], 1, 3, 3, device=dev)
        outs = []
        for dil in (12, 24, 36):
            fns = [lambda a=a: outs.append(ops.depthwise_conv3x3(a, wt, dil)) or (len(outs) > 2 and outs.pop(0)) for a in xs]
            report(f"depthwise3x3 dilation {dil} {shp}", timed(fns, args.iters), xs[0].numel() * 4)
            outs.clear()

    if want("bnact"):
        # the largest batch-norms of DeepLabV3+-R50 at the benchmark geometry: layer1 [8,256,128,128] fp16,
        # with and without the residual; torch's three kernels beside them
        import torch.nn.functional as F
        shp = (8, 256, 128, 128)
        xs = [torch.randn(*shp, device=dev).half() for _ in range(3)]
        rs = [torch.randn(*shp, device=dev).half() for _ in range(3)]
        c = shp[1]
        mean, var = torch.randn(c, device=dev), torch.rand(c, device=dev) + 0.5
        wt, b = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev)
        outs = []
        keep = lambda t: outs.append(t) or (len(outs) > 2 and outs.pop(0))
        nb = xs[0].numel() * 2
        report("bn_act relu", timed([lambda a=a: keep(ops.bn_act(a, mean, var, wt, b, 1e-5, relu=True)) for a in xs], args.iters), 2 * nb)
        report("torch batch_norm + relu", timed([lambda a=a: keep(F.relu(F.batch_norm(a, mean, var, wt, b, False, 0.0, 1e-5))) for a in xs],
                                                args.iters), 2 * nb)
        report("bn_act add relu", timed([lambda a=a, r=r: keep(ops.bn_act(a, mean, var, wt, b, 1e-5, residual=r, relu=True))
                                         for a, r in zip(xs, rs)], args.iters), 3 * nb)
        report("torch batch_norm + add + relu", timed([lambda a=a, r=r: keep(F.relu(F.batch_norm(a, mean, var, wt, b, False, 0.0, 1e-5) + r))
                                                      for a, r in zip(xs, rs)], args.iters), 3 * nb)
        outs.clear()

    if "conv1x1" in only or "conv1x1_pmc" in only:   # opt-in: one kernel name at several sizes
        # the one-launch 1x1 convolution + batch-norm tail at four pairs of the DeepLabV3+-R50 forward (batch 8, 512x512 input):
        # algorithmic bytes = x + y (+ residual), fp16; the weights (<= 0.5 MB) are not counted
        for name, cin, cout, side, res in (("layer1 conv3 + residual", 64, 256, 128, True), ("layer1 conv1", 256, 64, 128, False),
                                           ("layer2 conv3 + residual", 128, 512, 64, True), ("layer3 conv1", 1024, 256, 32, False)):
            if "conv1x1_pmc" in only and not name.startswith("layer1 conv3"):
                continue   # one size per counter pass: the counters are summed per kernel name
            n_buf = 4
            xs = [torch.randn(8, cin, side, side, device=dev).half() for _ in range(n_buf)]
            rs = [torch.randn(8, cout, side, side, device=dev).half() for _ in range(n_buf)] if res else [None] * n_buf
            w = (torch.randn(cout, cin, device=dev) / cin ** 0.5).half()
            bn = torch.nn.BatchNorm2d(cout).eval().to(dev)
            nb = 2 * 8 * side * side * (cin + cout + (cout if res else 0))
            outs = []
            keep = lambda t: outs.append(t) or (len(outs) > 2 and outs.pop(0))
            with torch.no_grad():
                report(f"conv1x1_bn_act one launch, {name} {cin}->{cout} @{side}x{side}",
                       timed([lambda a=a, r=r: keep(ops.conv1x1_bn_act(a, w, bn, r, None, True, route=1)) for a, r in zip(xs, rs)],
                             args.iters), nb)
                if "conv1x1" in only:
                    report(f"conv1x1_bn_act two launches, {name} {cin}->{cout} @{side}x{side}",
                           timed([lambda a=a, r=r: keep(ops.conv1x1_bn_act(a, w, bn, r, None, True, route=2)) for a, r in zip(xs, rs)],
                                 args.iters), nb)
            outs.clear()

    if "stem" in only:
        # ResNet stem behind the first convolution: [8, 64, 256, 256] fp16 -> pooled [8, 64, 128, 128] (the first
        # feature map not stored, as in the DeepLabV3+ model); algorithmic bytes = x + pooled
        import torch.nn.functional as F
        shp = (8, 64, 256, 256)
        xs = [torch.randn(*shp, device=dev).half() for _ in range(5)]
        bn = torch.nn.BatchNorm2d(64).eval().to(dev)
        nb = xs[0].numel() * 2 * 5 // 4
        outs = []
        keep = lambda t: outs.append(t) or (len(outs) > 2 and outs.pop(0))
        with torch.no_grad():
            report("bn_act_maxpool (pooled only)", timed([lambda a=a: keep(ops.bn_act_maxpool(a, bn, want_full=False)[1]) for a in xs],
                                                         args.iters), nb)
            if "conv1x1_pmc" not in only:
                report("bn_act + torch max_pool2d", timed([lambda a=a: keep(F.max_pool2d(
                    ops.bn_act(a, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.eps, relu=True), 3, 2, 1)) for a in xs],
                    args.iters), nb)
        outs.clear()


if __name__ == "__main__":
    main()
