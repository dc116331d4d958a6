"""Device time of the bench's timed region by kernel, from a rocprofv3 kernel trace of `SPICE_BENCH_MARK=1 bench.py`:
the region lies between the two launches of scale_inplace_kernel (which no sweep launches).
Usage: stepbreakdown.py <kernel_trace.csv> [steps] [rows]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "scale_inplace_kernel" in r["Kernel_Name"]]
if len(marks) < 2:
    sys.exit("no marked region in this trace (run bench.py with SPICE_BENCH_MARK=1)")
win = rows[marks[0] + 1:marks[1]]
dur = lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
t0, t1 = int(rows[marks[0]]["End_Timestamp"]), int(rows[marks[1]]["Start_Timestamp"])
heads = sum(1 for r in win if "kdlite_head" in r["Kernel_Name"] or "kdlite_main_kernel" in r["Kernel_Name"])
print(f"timed region: {heads} candidates ({steps} expected), wall per candidate {(t1 - t0) / steps / 1e6:.3f} ms (under the "
      f"profiler), sum of kernel durations per candidate {sum(map(dur, win)) / steps / 1e6:.3f} ms, launches per candidate "
      f"{len(win) / steps:.1f}")
agg = collections.defaultdict(lambda: [0, 0])
for r in win:
    a = agg[r["Kernel_Name"][:90]]
    a[0] += dur(r)
    a[1] += 1
for n, (t, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:int(sys.argv[3]) if len(sys.argv) > 3 else 32]:
    print(f"{n:90s} {c / steps:6.1f}/candidate {t / steps / 1e3:8.1f} us/candidate")
