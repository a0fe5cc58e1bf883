#!/usr/bin/env python3
"""Lab: durations of the decode step's rows-GEMM launches by grid size (= output columns / 16) from a rocprofv3 kernel trace.
    python3 tools/lab/decode_gemm_shapes.py <trace dir>"""
import collections, csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]                       # replays only
acc = collections.defaultdict(list)
for r in rows:
    if "gemm_rows" in r["Kernel_Name"]:
        key = ("mfma" if "mfma" in r["Kernel_Name"] else "dot", int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1))
        acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(acc.items()):
    v.sort()
    print(f"{k[0]:4s} blocks {k[1]:5d}: {len(v):6d} launches, median {v[len(v)//2]:6.2f} us, mean {sum(v)/len(v):6.2f} us")
