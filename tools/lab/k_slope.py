#!/usr/bin/env python3
"""Where does a short-K product spend its time?  Times M x N x K products at several K on cold operands
(rotation over R sets): the slope is the cost of a K-step, the intercept is prologue + epilogue.  GPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch

import hipvg
from hipvg import functional as F

dev = torch.device("cuda:0")
M = int(os.environ.get("M", "16000"))
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
    for N in (4096, 1024):
        for mode in ("plain", "gelu+deriv"):
            row = []
            for K in (512, 1024, 2048, 4096):
                xs = [torch.randn(M, K, generator=g).to(dev).bfloat16() for _ in range(R)]
                ws = [(torch.randn(N, K, generator=g) * K ** -0.5).to(dev).bfloat16() for _ in range(R)]
                ys = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(R)]
                us = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(R)]
                bias = torch.randn(N, generator=g).to(dev)
                if mode == "plain":
                    fns = [(lambda i=i: F.gemm(xs[i], ws[i], M, N, K, out=ys[i])) for i in range(R)]
                else:
                    fns = [(lambda i=i: F.gemm(xs[i], ws[i], M, N, K, bias=bias, act=2 | 16, aux_out=us[i], out=ys[i]))
                           for i in range(R)]
                row.append((K, run(fns)))
            slope = (row[-1][1] - row[1][1]) / ((row[-1][0] - row[1][0]) / 64)
            print(f"N={N} {mode:11s} " + "  ".join(f"K={k}: {t:6.1f} us" for k, t in row) +
                  f" | per 64-deep K-step {slope:.2f} us, intercept {row[1][1] - slope * row[1][0] / 64:.1f} us", flush=True)


if __name__ == "__main__":
    main()
