#!/usr/bin/env python3
"""Summarise a rocprofv3 *_kernel_stats.csv: per-family and top-N kernel time per micro-batch.
usage: tools/prof_summary.py <kernel_stats.csv> [micro_batches_in_trace=6] [top=45]"""
import csv
import sys

path = sys.argv[1]
nmb = float(sys.argv[2]) if len(sys.argv) > 2 else 6.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
rows = list(csv.DictReader(open(path)))


def family(n):
    if "gemm_dma_kernel" in n or ("gemm_kernel" in n and "float" not in n and "IfL" not in n):
        return "bf16 MFMA GEMM"
    if "gemm_kernel" in n:
        return "fp32 MFMA GEMM"
    if "attn_" in n:
        return "attention"
    if "dwnorm" in n:
        return "conv rows"
    if "GLOBAL__N" in n or "anonymous namespace)::c" in n or "anonymous namespace)::r" in n or "cast_kernel" in n \
            or "colsum" in n or "sum_kernel" in n:
        return "row kernels"
    if n.startswith("Cijk"):
        return "hipBLASLt"
    if "multi_tensor_apply" in n:
        return "AdamW / foreach"
    if "reduce_kernel" in n or "norm" in n.lower():
        return "stock reductions / norms"
    if "elementwise" in n or "copyBuffer" in n or "fillBuffer" in n or "CatArray" in n:
        return "stock elementwise / copies / fills"
    return "other"


fam = {}
for r in rows:
    f = family(r["Name"])
    t, c = fam.get(f, (0.0, 0))
    fam[f] = (t + float(r["TotalDurationNs"]), c + int(r["Calls"]))
tot = sum(t for t, _ in fam.values())
print(f"total GPU kernel time per micro-batch: {tot / nmb / 1e6:.2f} ms")
for f, (t, c) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
    print(f"  {f:38s} {t / nmb / 1e6:7.3f} ms  {100 * t / tot:5.1f} %  {c / nmb:7.1f} launches")
print()
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
    print(f"{float(r['TotalDurationNs']) / nmb / 1e6:8.3f} ms {int(r['Calls']) / nmb:7.1f} x {float(r['AverageNs']) / 1e3:8.1f} us  {r['Name'][:120]}")
