#!/usr/bin/env python3
"""Lab (variant builds -DVG_LAB_XCDMASK=m: blocks of the XCDs not in the mask exit at once): K-slope of the NT
product M=16000 N=1024 with all 8, 4 or 1 XCDs working at full occupancy."""
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
ts = []
KS = tuple(int(v) for v in os.environ.get("KS", "1024,4096").split(","))
for K in KS:
    A = torch.randn(M, K, generator=g).to(dev).bfloat16()
    B = torch.randn(N, K, generator=g).to(dev).bfloat16()
    out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    fn = lambda: F.gemm(A, B, M, N, K, out=out, tile_cfg=13)
    for _ in range(4): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): fn()
    b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) / 20 * 1e3)
print(os.environ.get("VG_LIB", "all XCDs").split("/")[-1], f"K={KS[0]} {ts[0]:.1f} us, K={KS[1]} {ts[1]:.1f} us, slope {(ts[1] - ts[0]) / ((KS[1] - KS[0]) / 64):.3f} us per K tile")
