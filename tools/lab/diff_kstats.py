#!/usr/bin/env python3
"""Lab: per-kernel-name difference of two rocprofv3 --kernel-trace --stats runs (total time and calls per step).
    python3 tools/lab/diff_kstats.py <dirA> <dirB> <steps run in each>"""
import csv, glob, sys
def load(d, n):
    f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
    return {r["Name"]: (float(r["TotalDurationNs"]) / n / 1e6, int(r["Calls"]) / n) for r in csv.DictReader(open(f))}
n = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
a, b = load(sys.argv[1], n), load(sys.argv[2], n)
rows = []
for k in set(a) | set(b):
    ta, ca = a.get(k, (0.0, 0.0)); tb, cb = b.get(k, (0.0, 0.0))
    rows.append((tb - ta, k, ta, ca, tb, cb))
rows.sort(key=lambda r: -abs(r[0]))
print(f"total A {sum(v[0] for v in a.values()):.3f} ms  B {sum(v[0] for v in b.values()):.3f} ms per step")
for d, k, ta, ca, tb, cb in rows[:28]:
    print(f"{d:+8.3f} ms  A {ta:7.3f} ms x{ca:6.1f}   B {tb:7.3f} ms x{cb:6.1f}   {k[:90]}")
