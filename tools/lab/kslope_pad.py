#!/usr/bin/env python3
"""Lab: does the row stride of the operands matter at full chip?  Same NT product (M=16000, N=1024, K=4096) with
dense operands (ld = K: a power of two times 2 bytes) and with padded row strides (ld = K + pad)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch
import hipvg
from hipvg import functional as F
dev = torch.device("cuda:0")
hipvg.lib()
g = torch.Generator().manual_seed(0)
M, N = 16000, 1024
R = 6
for K in (1024, 4096):
    for pad in (0, 64, 128, 192, 256, 1024):
        As = [torch.randn(M, K + pad, generator=g).to(dev).bfloat16()[:, :K] for _ in range(R)]
        Bs = [torch.randn(N, K + pad, generator=g).to(dev).bfloat16()[:, :K] for _ in range(R)]
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        fn = lambda i: F.gemm(As[i], Bs[i], M, N, K, out=out, tile_cfg=13)
        for i in range(R): fn(i)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(4):
            for i in range(R): fn(i)
        b.record(); torch.cuda.synchronize()
        t = a.elapsed_time(b) / (4 * R) * 1e3
        print(f"NT K={K} row stride {K + pad:5d} elements: {t:7.1f} us  {2.0 * M * N * K / t / 1e6:6.0f} TF", flush=True)
