#!/usr/bin/env python3
"""Sweep the split-K factor of the weight-gradient GEMM (TN) on the layer shapes.  GPU only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch
import hipvg
from hipvg import functional as F

dev = torch.device("cuda:0")
M = 8000


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e-3


hipvg.lib()
for (N, K) in [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096), (2048, 512), (512, 2048), (512, 512)]:
    x = torch.randn(M, K, device=dev).bfloat16()
    dy = torch.randn(M, N, device=dev).bfloat16()
    out = torch.zeros(N, K, device=dev)
    row = []
    for s in (1, 2, 3, 4, 6, 8, 12, 16):
        t = timeit(lambda: F.gemm(dy, x, N, K, M, a_tr=True, b_tr=True, out=out, split_k=s, accumulate=(s == 1)))
        row.append(f"S={s}: {2.0 * M * N * K / t / 1e12:6.0f} TF {t * 1e6:6.1f}us")
    print(f"dW[{N}x{K}] | " + " | ".join(row), flush=True)
