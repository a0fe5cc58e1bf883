#!/usr/bin/env python3
"""Split-K / tile sweep of the weight-gradient GEMMs on COLD operands (rotation over R operand sets, as inside
the training step), accumulate-into-gradient form.  GPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch

import hipvg
from hipvg import functional as F

dev = torch.device("cuda:0")
M, R, ITERS = int(os.environ.get("M", "8000")), 8, 4


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
    return a.elapsed_time(b) / (ITERS * len(fns)) * 1e-3


def main():
    hipvg.lib()
    g = torch.Generator(device="cpu").manual_seed(0)
    for (N, K) in [(4096, 1024), (1024, 4096), (3072, 1024), (1024, 1024)]:
        dys = [torch.randn(M, N, generator=g).to(dev).bfloat16() for _ in range(R)]
        xs = [torch.randn(M, K, generator=g).to(dev).bfloat16() for _ in range(R)]
        gws = [torch.zeros(N, K, device=dev) for _ in range(R)]
        row = []
        for cfg in [int(v) for v in os.environ.get("CFGS", "1,2,4").split(",")]:
            for s in [int(v) for v in os.environ.get("SPLITS", "1,2,3,4,6").split(",")]:
                t = run([(lambda i=i: F.gemm(dys[i], xs[i], N, K, M, a_tr=True, b_tr=True, out=gws[i], split_k=s,
                                             accumulate=(s == 1), tile_cfg=cfg)) for i in range(R)])
                row.append(f"cfg{cfg} s{s}: {t * 1e6:6.1f}")
        print(f"dW[{N}x{K}] (auto split {F.wgrad_splits(N, K, M, torch.bfloat16)}) | " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
