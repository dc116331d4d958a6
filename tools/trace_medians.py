"""Median / min / mean duration per kernel name from a rocprofv3 --kernel-trace CSV directory
(library kernels only; any second argument lists every kernel)."""
import csv
import glob
import os
import sys
from collections import defaultdict

import numpy as np


def main():
    root = sys.argv[1]
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        raise SystemExit(f"no *kernel_trace.csv under {root}")
    d = defaultdict(list)
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                d[r["Kernel_Name"] + f' grid={r.get("Grid_Size_X", "?")} wg={r.get("Workgroup_Size_X", "?")}'].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print(f"{'n':>5} {'median_us':>10} {'min_us':>9} {'mean_us':>9}  kernel")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        if len(sys.argv) < 3 and "spice::" not in k:
            continue
        a = np.array(v) / 1e3
        print(f"{len(a):5d} {np.median(a):10.2f} {a.min():9.2f} {a.mean():9.2f}  {k[:150]}")


if __name__ == "__main__":
    main()
