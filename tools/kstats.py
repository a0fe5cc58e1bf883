#!/usr/bin/env python3
"""Per-step kernel table from a rocprofv3 --kernel-trace --stats run of bench.py:  python tools/kstats.py <dir> [steps+warmup+1]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0     # optimizer steps the run held (warm-up + timed)
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per step {tot / n / 1e6:.2f} ms, {sum(int(r['Calls']) for r in rows) / n:.0f} launches")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 34]:
    print(f"{float(r['TotalDurationNs']) / n / 1e6:7.3f} ms  x{int(r['Calls']) / n:6.1f}  avg {float(r['AverageNs']) / 1e3:7.1f} us  {r['Name'][:110]}")
