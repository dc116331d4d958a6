"""Pretty-print the JSON lines of tools/conv1x1_bench.py (stdin)."""
import json, sys
for line in sys.stdin:
    line = line.strip()
    if line.startswith('{"pair"'):
        d = json.loads(line)
        extra = "".join(f"  {k[:-3]} {d[k]:6.1f}" for k in d if k.endswith("_us") and not k.startswith("route"))
        print(f"{d['pair']:16s} x{d['count']} K={d['cin']:5d} M={d['cout']:5d} P={d['pixels']:7d} {d['tail']:8s} {d['MB']:6.1f} MB"
              + "".join(f"  r{k[5]} {d[k]:6.1f} us" for k in d if k.startswith("route") and k.endswith("_us")) + extra)
    elif line:
        print(line[:200])
