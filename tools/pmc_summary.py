"""Per-kernel HBM traffic from two rocprofv3 counter passes (run separately, one counter each):

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_fetch -- python3 tools/kbench.py --iters 6
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_write -- python3 tools/kbench.py --iters 6
    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/rNN/pmc_traffic.json

Both counters are in KB.  On gfx950 FETCH_SIZE tallies these streaming reads at half their size
(the 4-byte-per-lane KD-Lite gradient kernel included: its doubled figure equals the bytes it must
read), so the fetch figure is doubled.  One kernel name may serve several input sizes: median and
max over its launches are both kept, and hbm_bytes_corrected uses the max (the benchmark size).
"""
import csv
import glob
import json
import statistics
import sys


def collect(directory, counter):
    out = {}
    for path in glob.glob(f"{directory}/**/*counter_collection.csv", recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] == counter and "spice" in row["Kernel_Name"]:  # mangled names too (fp16 template arguments)
                    name = row["Kernel_Name"].replace("void ", "").split("(")[0]
                    out.setdefault(name, []).append(float(row["Counter_Value"]))
    return out


def main():
    fetch, write = collect(sys.argv[1], "FETCH_SIZE"), collect(sys.argv[2], "WRITE_SIZE")
    kernels = {}
    for name in sorted(set(fetch) | set(write)):
        f, w = fetch.get(name, [0.0]), write.get(name, [0.0])
        factor = 2
        kernels[name] = {"launches": len(f), "FETCH_SIZE_KB_median": statistics.median(f), "FETCH_SIZE_KB_max": max(f),
                         "WRITE_SIZE_KB_median": statistics.median(w), "WRITE_SIZE_KB_max": max(w),
                         "fetch_factor": factor,
                         "hbm_bytes_corrected": int(round((factor * max(f) + max(w)) * 1024))}
    json.dump({"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python3 tools/kbench.py --iters 6",
               "note": "per launch; KB counters; fetch doubled per the gfx950 correction",
               "kernels": kernels}, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
