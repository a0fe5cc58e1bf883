#!/usr/bin/env python3
"""Lab: idle gaps of the GPU timeline in a rocprofv3 kernel trace of `bench.py` (hipGraph mode): every gap above 15 us
with the kernels on either side, and the sum of all gaps, inside the last N optimizer steps (delimited by adamw launches).
  python3 tools/lab/step_gaps.py <rocprofv3 -d dir> [steps] [steps to drop at the end]"""
import csv
import glob
import sys

d = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
skip = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # optimizer steps to drop at the end (bench.py ends with an eager, profiled step)
rows = []
for fn in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r[2]]
# optimizer steps end with a burst of adamw launches: boundaries = last adamw of each burst
ends = [ad[i] for i in range(len(ad)) if i + 1 == len(ad) or ad[i + 1] - ad[i] > 50]
ends = ends[:len(ends) - skip][-(nsteps + 1):]
lo, hi = ends[0] + 1, ends[-1] + 1
seg = rows[lo:hi]
span = seg[-1][1] - seg[0][0]
busy = sum(e - s for s, e, _ in seg)
print(f"{nsteps} steps: span {span / 1e6 / nsteps:.3f} ms per step, kernel time {busy / 1e6 / nsteps:.3f} ms, {len(seg) / nsteps:.0f} launches per step")
gaps = []
last_end = seg[0][1]
for i in range(1, len(seg)):
    g = seg[i][0] - last_end
    if g > 0:
        gaps.append((g, seg[i - 1][2][:60], seg[i][2][:60]))
    last_end = max(last_end, seg[i][1])
tot = sum(g for g, _, _ in gaps)
print(f"idle: {tot / 1e6 / nsteps:.3f} ms per step in {len(gaps) / nsteps:.0f} gaps; gaps > 15 us:")
big = sorted([g for g in gaps if g[0] > 15000], reverse=True)
print(f"  {len(big) / nsteps:.1f} per step, {sum(g[0] for g in big) / 1e6 / nsteps:.3f} ms per step")
for g, a, b in big[:12]:
    print(f"  {g / 1e3:8.1f} us  after {a}  before {b}")
small = [g[0] for g in gaps if g[0] <= 15000]
if small:
    print(f"gaps <= 15 us: {len(small) / nsteps:.0f} per step, mean {sum(small) / len(small) / 1e3:.2f} us, {sum(small) / 1e6 / nsteps:.3f} ms per step")
