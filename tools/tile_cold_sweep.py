#!/usr/bin/env python3
"""Tile-configuration sweep of the forward (NT) and dgrad (NN) products of the Transformer layer on COLD operands
(rotation over R operand sets), for a given row count M (env M, default 8000).  GPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch

import hipvg
from hipvg import functional as F

dev = torch.device("cuda:0")
M = int(os.environ.get("M", "8000"))
R, ITERS = 6, 4


def run(fns):
    for f in fns:
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(ITERS):
        for f in fns:
            f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (ITERS * len(fns)) * 1e3


def main():
    hipvg.lib()
    g = torch.Generator(device="cpu").manual_seed(0)
    shapes = [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096), (1024, 3072)]
    if os.environ.get("SHAPES"):      # SHAPES=512x2048,2048x512  (N x K)
        shapes = [tuple(int(v) for v in t.split("x")) for t in os.environ["SHAPES"].split(",")]
    for (N, K) in shapes:
        xs = [torch.randn(M, K, generator=g).to(dev).bfloat16() for _ in range(R)]
        ws = [(torch.randn(N, K, generator=g) * K ** -0.5).to(dev).bfloat16() for _ in range(R)]
        wt = [(torch.randn(K, N, generator=g) * K ** -0.5).to(dev).bfloat16() for _ in range(R)]
        ys = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(R)]
        for mode in ("NT", "NN"):
            row = []
            for cfg in [int(v) for v in os.environ.get("CFGS", "1,2,3,4,5").split(",")]:
                if mode == "NT":
                    fns = [(lambda i=i: F.gemm(xs[i], ws[i], M, N, K, out=ys[i], tile_cfg=cfg)) for i in range(R)]
                else:
                    fns = [(lambda i=i: F.gemm(xs[i], wt[i], M, N, K, b_tr=True, out=ys[i], tile_cfg=cfg)) for i in range(R)]
                row.append(f"cfg{cfg}: {run(fns):6.1f}")
            print(f"M={M} N={N:5d} K={K:5d} {mode} | " + " | ".join(row) + " us", flush=True)


if __name__ == "__main__":
    main()
