#!/usr/bin/env python3
"""K-slope / intercept of the NT GEMM tile kernels: time against K at M=16000, N=1024 (252 tiles, one per CU), cold
rotating operands.  CFGS=3,10,12,13,14 (12: no DMA in loop, 13: no MFMA, 14: no fragment reads -- lab ablations)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch
import hipvg
from hipvg import functional as F
dev = torch.device("cuda:0")
CFGS = [int(c) for c in os.environ.get("CFGS", "3,10,12,13,14").split(",")]
M, N, R = int(os.environ.get("M", "16000")), int(os.environ.get("N", "1024")), 6
hipvg.lib()
g = torch.Generator().manual_seed(0)
for K in (256, 1024, 2048, 4096):
    A = [torch.randn(M, K, generator=g).to(dev).bfloat16() for _ in range(R)]
    B = [(torch.randn(N, K, generator=g) * K ** -0.5).to(dev).bfloat16() for _ in range(R)]
    C = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(R)]
    res = {c: [] for c in CFGS}
    for c in CFGS:
        for i in range(R):
            F.gemm(A[i], B[i], M, N, K, out=C[i], tile_cfg=c)
    torch.cuda.synchronize()
    for _ in range(6):
        for c in CFGS:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(R):
                F.gemm(A[i], B[i], M, N, K, out=C[i], tile_cfg=c)
            b.record(); torch.cuda.synchronize()
            res[c].append(a.elapsed_time(b) / R * 1e3)
    print(f"K={K:5d} " + " | ".join(f"cfg{c:2d} {sorted(res[c])[3]:7.1f} us" for c in CFGS), flush=True)
