#!/usr/bin/env python3
"""Lab: per optimizer step of a rocprofv3 kernel trace (steps delimited by the first adamw launch after >= 5 ms without
one): span, kernel time, idle time, launches; and the gaps above 15 us of ONE chosen step with the kernels around them.
  python3 tools/lab/gaps_steps.py <rocprofv3 -d dir> [index of the step to list, negative from the end: -6]"""
import csv
import glob
import sys

d = sys.argv[1]
pick = int(sys.argv[2]) if len(sys.argv) > 2 else -6
rows = []
for fn in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
marks, last_ad = [], -10**18
for i, r in enumerate(rows):
    if "adamw_kernel" in r[2]:
        if r[0] - last_ad > 5e6:
            marks.append(i)
        last_ad = r[0]
clean = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")[:56]
steps = []
for a, b in zip(marks, marks[1:]):
    seg = rows[a:b]
    span = seg[-1][1] - seg[0][0]
    busy, idle, cur_end = 0, 0, seg[0][0]
    for s, e, n, q in seg:
        if s > cur_end:
            idle += s - cur_end
        if e > cur_end:
            busy += e - max(s, cur_end)
            cur_end = e
    steps.append((a, b, span, busy, idle, len(seg)))
for k, (a, b, span, busy, idle, n) in enumerate(steps):
    print(f"step {k:2d}: span {span / 1e6:7.3f} ms  busy {busy / 1e6:7.3f}  idle {idle / 1e6:6.3f}  kernels {n}")
a, b, *_ = steps[pick]
seg = rows[a:b]
print(f"-- gaps above 15 us in step {pick % len(steps)}:")
cur_end, last = seg[0][1], seg[0]
for r in seg[1:]:
    if r[0] - cur_end > 15000:
        print(f"  {(r[0] - cur_end) / 1e3:7.1f} us at +{(r[0] - seg[0][0]) / 1e6:6.2f} ms  after {clean(last[2])} [q{last[3]}] | before {clean(r[2])} [q{r[3]}]")
    if r[1] > cur_end:
        cur_end, last = r[1], r
qs = {}
for s, e, n, q in seg:
    qs.setdefault(q, [0, 0])
    qs[q][0] += 1
    qs[q][1] += e - s
print("queues:", {q: (c, round(t / 1e6, 2)) for q, (c, t) in qs.items()})
print("-- markers of the chosen step (ms from its first kernel):")
for s, e, n, q in seg:
    if any(k in n for k in ("oneRankReduce", "gemm_ring_group", "adamw_kernel", "ncclDevKernel", "AllReduce")):
        print(f"  +{(s - seg[0][0]) / 1e6:7.3f} .. +{(e - seg[0][0]) / 1e6:7.3f}  {clean(n)} [q{q}]")
