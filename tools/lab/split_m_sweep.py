#!/usr/bin/env python3
"""Lab (round 6): what a row band costs as a launch of its own -- the pieces of a product split over M (csrc/vg_gemm.hip:
split_rows).  Cold operands (R rotating buffer sets), plain bf16 epilogue.  SHAPES=M:N:K,...  python tools/lab/split_m_sweep.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch

import hipvg
from hipvg import functional as F

dev = torch.device("cuda:0")
R, ITERS = 4, 5


def run(fns):
    for f in fns:
        f()
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(ITERS):
            for f in fns:
                f()
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / (ITERS * len(fns)) * 1e3
        best = t if best is None else min(best, t)
    return best


hipvg.lib()
g = torch.Generator().manual_seed(0)
shapes = [tuple(int(v) for v in s.split(":")) for s in os.environ.get(
    "SHAPES", "13312:4096:1024,12288:4096:1024,1024:4096:1024,13312:3072:1024,10752:3072:1024,2560:3072:1024,"
              "10240:4096:1024,8192:4096:1024,2048:4096:1024,13312:1024:1024,13312:1024:4096,13312:2048:512,8192:2048:512,5120:2048:512").split(",")]
for (M, N, K) in shapes:
    xs = [torch.randn(M, K, generator=g).to(dev).bfloat16() for _ in range(R)]
    ws = [(torch.randn(N, K, generator=g) * K ** -0.5).to(dev).bfloat16() for _ in range(R)]
    ys = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(R)]
    row = []
    for cfg in (0, 13, 15, 1):
        try:
            t = run([(lambda i=i: F.gemm(xs[i], ws[i], M, N, K, out=ys[i], tile_cfg=cfg)) for i in range(R)])
            row.append(f"cfg{cfg} {t:6.1f}")
        except Exception as e:       # noqa: BLE001
            row.append(f"cfg{cfg}   n/a")
    print(f"M={M:6d} N={N:5d} K={K:5d} NT | " + " | ".join(row), flush=True)
    del xs, ws, ys
