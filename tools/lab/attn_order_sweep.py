#!/usr/bin/env python3
"""Lab (round 6): launch order of the attention blocks by rank (sweep length), VG_ATTN_SCHED bit 8 + the packed order.
Per-kernel times from HIP events around each of the three launches are not available through the C ABI, so this times
forward and backward (dQ + dK/dV share one order of 8 ranks at T = 1000; the forward has 4 ranks)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def code(order):
    v = 8
    for j, r in enumerate(order):
        v |= r << (8 + 3 * j)
    return v


orders = {
    "lpt 0..7": None,
    "0 7 1 6 2 5 3 4": [0, 7, 1, 6, 2, 5, 3, 4],
    "0 4 1 5 2 6 3 7": [0, 4, 1, 5, 2, 6, 3, 7],
    "0 1 7 2 6 3 5 4": [0, 1, 7, 2, 6, 3, 5, 4],
    "0 2 4 6 1 3 5 7": [0, 2, 4, 6, 1, 3, 5, 7],
    "0 1 2 7 3 6 4 5": [0, 1, 2, 7, 3, 6, 4, 5],
    "1 0 2 3 4 5 6 7": [1, 0, 2, 3, 4, 5, 6, 7],
    "0 3 1 2 (fwd: 0 3 1 2)": [0, 3, 1, 2, 4, 5, 6, 7],
    "0 2 1 3 (fwd)": [0, 2, 1, 3, 4, 5, 6, 7],
}
for name, order in orders.items():
    env = dict(os.environ)
    if order is not None:
        env["VG_ATTN_SCHED"] = str(code(order))
    env["SHAPES"] = "16x1000"
    env["STD"] = "0.3"
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "attn_bench.py")], env=env, capture_output=True, text=True).stdout
    line = [l for l in out.splitlines() if l.startswith("B=")]
    print(f"{name:28s} {line[0] if line else out[-200:]}", flush=True)
