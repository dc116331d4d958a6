"""Per-step GPU time of the sweep from a rocprofv3 kernel trace of bench.py: finds a run of
candidate-ending launches and sums kernel durations between them.  Usage: stepbreakdown.py <kernel_trace.csv> [steps]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# a candidate ends in one launch of the head kernel (up-sampling + loss + arg-max histogram), or, on the three-launch
# route, of the plain loss kernel
kd = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith(("void spice::kdlite_head_kernel", "void spice::kdlite_head2_kernel"))
      and ", true" in r["Kernel_Name"].split("(")[0]]   # (the loss forms; the arg-max-only form belongs to the evaluation loops)
if len(kd) <= steps:
    kd = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void spice::kdlite_main_kernel<4, false, false, false>")]
ts = [int(rows[i]["Start_Timestamp"]) for i in kd]
gaps = [(ts[k + 1] - ts[k]) / 1e6 for k in range(len(ts) - 1)]
best, s = (0, 0), None
for k, g in enumerate(gaps + [0]):
    if 0.3 < g < 12:
        if s is None:
            s = k
    else:
        if s is not None and k - s > best[1] - best[0]:
            best = (s, k)
        s = None
b = best[1]
i0, i1 = kd[b - steps], kd[b]
win = rows[i0 + 1:i1 + 1]
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
t0, t1 = int(rows[i0]["End_Timestamp"]), int(rows[i1]["End_Timestamp"])
print(f"wall per step {(t1 - t0) / steps / 1e6:.3f} ms (under the profiler), GPU busy per step {sum(map(dur, win)) / steps / 1e6:.3f} ms, "
      f"launches per step {len(win) / steps:.1f}")
agg = collections.defaultdict(lambda: [0, 0])
for r in win:
    a = agg[r["Kernel_Name"][:90]]
    a[0] += dur(r)
    a[1] += 1
for n, (t, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    print(f"{n:90s} {c / steps:6.1f}/step {t / steps / 1e3:8.1f} us/step")
