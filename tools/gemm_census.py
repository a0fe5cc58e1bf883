#!/usr/bin/env python3
"""Census of the bf16 GEMM launches of one training step: shape, tile configuration, epilogue options and FLOPs
(VG_DEBUG_GEMM=2 makes the library print one line per launch).  GPU only.
  VG_DEBUG_GEMM=2 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --graph 0 2>&1 >/dev/null | python tools/gemm_census.py"""
import collections
import re
import sys

rows = collections.Counter()
for line in sys.stdin:
    if line.startswith("[vg_gemm] cfg="):
        rows[line.strip()[10:]] += 1
tot = 0.0
out = []
for sig, n in rows.items():
    kv = dict(re.findall(r"(\w+)=([-\w.+]+)", sig))
    fl = 2.0 * int(kv["M"]) * int(kv["N"]) * int(kv["K"]) * n
    tot += fl
    out.append((fl, n, sig))
for fl, n, sig in sorted(out, reverse=True):
    print(f"{fl / tot * 100:5.1f} %  x{n:3d}  {sig}")
