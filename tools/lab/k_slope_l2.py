#!/usr/bin/env python3
"""Per-K-step cost of the 256x256 tile when every operand line is an L2 hit: a 32-tile problem whose operands
fit the XCDs' L2 (same operands every launch), K = 512..2048.  GPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch

import hipvg
from hipvg import functional as F

dev = torch.device("cuda:0")


def run(f, iters=50):
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    hipvg.lib()
    g = torch.Generator(device="cpu").manual_seed(0)
    for (M, N, cfg) in ((4096, 512, 3), (16384, 4096, 3), (2048, 512, 1), (16384, 4096, 1)):
        row = []
        for K in (512, 1024, 2048):
            x = torch.randn(M, K, generator=g).to(dev).bfloat16()
            w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev).bfloat16()
            y = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            row.append((K, run(lambda: F.gemm(x, w, M, N, K, out=y, tile_cfg=cfg))))
        slope = (row[-1][1] - row[0][1]) / ((row[-1][0] - row[0][0]) / 64)
        print(f"M={M} N={N} cfg{cfg} " + "  ".join(f"K={k}: {t:6.1f} us" for k, t in row) + f" | per K-step {slope:.2f} us",
              flush=True)


if __name__ == "__main__":
    main()
