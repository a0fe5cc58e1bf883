#!/usr/bin/env python3
"""Lab: timeline of the replayed decode step from a rocprofv3 kernel trace.
    python3 tools/lab/decode_timeline.py <trace dir>
Groups the kernels of the LAST 100 replays: per kernel name the count per frame, the average duration and the average
gap to the previous kernel's end."""
import collections, csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# frames are delimited by sample_token_kernel
idx = [i for i, r in enumerate(rows) if "sample_token" in r["Kernel_Name"]]
idx = idx[-101:]
sel = rows[idx[0] + 1: idx[-1] + 1]
nfr = len(idx) - 1
dur, gap, cnt = collections.Counter(), collections.Counter(), collections.Counter()
prev_end = None
for r in sel:
    n = r["Kernel_Name"].split("(")[0][-50:]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    dur[n] += e - s
    cnt[n] += 1
    if prev_end is not None:
        gap[n] += s - prev_end
    prev_end = e
span = int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])
print(f"{nfr} frames, {len(sel) / nfr:.1f} kernels per frame, {span / nfr / 1e3:.1f} us per frame, "
      f"kernel time {sum(dur.values()) / nfr / 1e3:.1f} us, gaps {sum(gap.values()) / nfr / 1e3:.1f} us")
for n, c in cnt.most_common():
    print(f"{c / nfr:6.1f} per frame  avg {dur[n] / c / 1e3:6.2f} us  gap before {gap[n] / c / 1e3:6.2f} us  {n}")
