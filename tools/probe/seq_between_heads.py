"""Kernel sequence of one candidate: the launches between two consecutive head launches in a rocprofv3 kernel trace."""
import csv, glob, sys
path = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
heads = [i for i, r in enumerate(rows) if "kdlite_head2_kernel<28, 4" in r["Kernel_Name"]]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(heads) // 2
a, b = heads[which], heads[which + 1]
tot = 0
for r in rows[a + 1:b + 1]:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    tot += d
    print(f"{d:8.1f} us  {r['Kernel_Name'][:100]}  grid={r.get('Grid_Size_X', r.get('Grid_Size'))}")
print(f"sum {tot:.1f} us over {b - a} launches; wall {(int(rows[b]['End_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1e3:.1f} us")
