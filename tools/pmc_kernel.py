"""Counter values per kernel name from one rocprofv3 --pmc pass: median over the launches of each kernel.
Usage: python tools/pmc_kernel.py <directory> [substring of the kernel name]"""
import csv
import glob
import statistics
import sys


def main():
    directory, needle = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "spice")
    vals = {}
    for path in glob.glob(f"{directory}/**/*counter_collection.csv", recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if needle in row["Kernel_Name"]:
                    name = row["Kernel_Name"].replace("void ", "").split("(")[0]
                    vals.setdefault((name, row["Counter_Name"]), []).append(float(row["Counter_Value"]))
    for (name, counter), v in sorted(vals.items()):
        print(f"{name}\t{counter}\t{statistics.median(v):.0f}\t(n={len(v)})")


if __name__ == "__main__":
    main()
