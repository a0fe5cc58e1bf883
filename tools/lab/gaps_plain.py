#!/usr/bin/env python3
"""Lab: idle gaps in the tail of a rocprofv3 kernel trace (no step detection: the last WINDOW_MS of the trace, minus the
final TAIL_MS), every gap above 20 us with the kernels on either side.
  python3 tools/lab/gaps_plain.py <rocprofv3 -d dir> [window ms = 90] [tail ms = 40]"""
import csv
import glob
import sys

d = sys.argv[1]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 90.0
tail = float(sys.argv[3]) if len(sys.argv) > 3 else 40.0
rows = []
for fn in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
t_end = rows[-1][1] - tail * 1e6
t_beg = t_end - win * 1e6
seg = [r for r in rows if t_beg <= r[0] <= t_end]
busy = sum(e - s for s, e, _, _ in seg)
print(f"window {win} ms: {len(seg)} kernels, kernel time {busy / 1e6:.2f} ms")
clean = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "")[:60]
last_end, last = seg[0][1], seg[0]
tot = 0
for r in seg[1:]:
    g = r[0] - last_end
    if g > 20000:
        print(f"  gap {g / 1e3:7.1f} us  after {clean(last[2])} [q{last[3]}] | before {clean(r[2])} [q{r[3]}]")
    if g > 0:
        tot += g
    if r[1] > last_end:
        last_end, last = r[1], r
print(f"sum of gaps {tot / 1e6:.2f} ms")
names = {}
for s, e, n, q in seg:
    if "ccl" in n.lower() or "allreduce" in n.lower() or "AllReduce" in n:
        names.setdefault(clean(n), []).append((e - s) / 1e3)
for n, v in names.items():
    print(f"  collective kernel {n}: {len(v)} x avg {sum(v) / len(v):.1f} us")
