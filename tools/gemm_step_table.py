#!/usr/bin/env python3
"""Per-shape time table of the bf16 GEMM launches of one eager training step (VG_DEBUG_GEMM=3: the library times every
launch with a pair of events and prints it with the shape).  GPU only.
  VG_DEBUG_GEMM=3 python bench.py --steps 1 --warmup 1 --no-cpu-baseline --graph 0 2>&1 >/dev/null | python tools/gemm_step_table.py
Lines are (shape, operand mode, epilogue) classes sorted by time share, with the TFLOP/s of the class; only the LAST
step's launches are kept (the count of lines per step is taken from the repeat of the first signature)."""
import collections
import re
import sys

recs = []
for line in sys.stdin:
    if line.startswith("[vg_gemm_t] "):
        kv = dict(re.findall(r"(\w+)=([-\w.+]+)", line))
        recs.append(kv)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = len(recs) // steps
recs = recs[-n:]
agg = collections.OrderedDict()
for kv in recs:
    us = float(kv.pop("us"))
    gf = float(kv.pop("gflop")) if "gflop" in kv else 2e-9 * int(kv["M"]) * int(kv["N"]) * int(kv["K"])
    key = " ".join(f"{k}={v}" for k, v in kv.items())
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += us
    a[2] += gf
tot = sum(a[1] for a in agg.values())
totf = sum(a[2] for a in agg.values())
print(f"{len(recs)} launches, {tot / 1e3:.2f} ms, {totf / tot * 1e3:.1f} TFLOP/s")
for key, (cnt, us, gf) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{us / tot * 100:5.1f} %  x{cnt:3d}  avg {us / cnt:7.1f} us  {gf / us * 1e3:7.1f} TF/s  {key}")
