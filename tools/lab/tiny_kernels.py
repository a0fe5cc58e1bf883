#!/usr/bin/env python3
"""Lab: the small launches of the replayed training step.  Reads a rocprofv3 kernel trace (csv) of
`python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline` and prints, for the LAST replayed optimizer step, every
kernel name whose average duration is below 12 us with its count and summed time."""
import collections, csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# one optimizer step ends with the adamw launches; take the span between the last two groups of them
ad = [i for i, r in enumerate(rows) if "adamw_kernel" in r["Kernel_Name"]]
groups = []
for i in ad:
    if groups and i - groups[-1][-1] < 40:
        groups[-1].append(i)
    else:
        groups.append([i])
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lo, hi = groups[-1 - nsteps][-1] + 1, groups[-1][-1] + 1
sel = rows[lo:hi]
span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e6 / nsteps
dur, cnt = collections.Counter(), collections.Counter()
for r in sel:
    n = r["Kernel_Name"]
    dur[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    cnt[n] += 1
tot = sum(dur.values()) / 1e6 / nsteps
print(f"{len(sel) / nsteps:.0f} kernels per step, span {span:.2f} ms, kernel time {tot:.2f} ms")
small = [(dur[n], cnt[n], n) for n in cnt if dur[n] / cnt[n] < 12e3]
small.sort(reverse=True)
print(f"kernels with average < 12 us: {sum(c for _, c, _ in small) / nsteps:.0f} launches, {sum(d for d, _, _ in small) / 1e6 / nsteps:.2f} ms per step")
for d, c, n in small[:70]:
    print(f"{c / nsteps:6.1f} x {d / c / 1e3:6.2f} us = {d / 1e6 / nsteps:6.3f} ms  {n[:120]}")
