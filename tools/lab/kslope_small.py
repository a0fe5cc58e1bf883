#!/usr/bin/env python3
"""Lab: K-slope of the long-phase 256x256 main loop when only a few CUs run and the operands stay cache-resident
(warm, same buffers) against the full-chip cold case: is the loop bound inside the CU or by operand delivery?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch
import hipvg
from hipvg import functional as F
dev = torch.device("cuda:0")
hipvg.lib()
g = torch.Generator().manual_seed(0)
def run(M, N, Ks, mode, rot):
    ts = []
    for K in Ks:
        R = rot
        As = [torch.randn(M, K, generator=g).to(dev).bfloat16() for _ in range(R)]
        Bs = [(torch.randn(N, K, generator=g)).to(dev).bfloat16() for _ in range(R)] if mode == "nt" else [torch.randn(K, N, generator=g).to(dev).bfloat16() for _ in range(R)]
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fn = lambda i: F.gemm(As[i], Bs[i], M, N, K, b_tr=(mode == "nn"), out=out, tile_cfg=13)
        for i in range(R): fn(i)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 0
        a.record()
        for _ in range(max(1, 24 // R)):
            for i in range(R):
                fn(i); n += 1
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / n * 1e3)
    slope = (ts[-1] - ts[0]) / ((Ks[-1] - Ks[0]) / 64)
    print(f"{mode} M={M:6d} N={N:5d} tiles={(M+255)//256*((N+255)//256):4d} rot={rot}: " + " ".join(f"K={k}:{t:7.1f}us" for k, t in zip(Ks, ts)) + f" | slope {slope:.3f} us per K tile", flush=True)
for mode in ("nt", "nn"):
    run(512, 512, (1024, 4096, 8192), mode, 1)       # 4 tiles, warm
    run(2048, 1024, (1024, 4096, 8192), mode, 1)     # 32 tiles, warm
    run(8192, 1024, (1024, 4096), mode, 1)           # 128 tiles, warm
    run(16000, 1024, (1024, 4096), mode, 1)          # 252 tiles, warm (64 MB+: Infinity Cache)
    run(16000, 1024, (1024, 4096), mode, 6)          # 252 tiles, cold
