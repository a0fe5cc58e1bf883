#!/usr/bin/env python3
"""Lab: a few launches of one layer GEMM (for rocprofv3 --pmc passes): SHAPE=dgrad_model|ffn_out|qkv|wgrad"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch
import hipvg
from hipvg import functional as F
dev = torch.device("cuda:0")
hipvg.lib()
g = torch.Generator().manual_seed(0)
M, D, Fd = 16000, 1024, 4096
mk = lambda *s: torch.randn(*s, generator=g).to(dev).bfloat16()
shape = os.environ.get("SHAPE", "dgrad_model")
if shape == "dgrad_model":
    a, w, out = mk(M, Fd), mk(Fd, D), torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    fn = lambda: F.gemm(a, w, M, D, Fd, b_tr=True, out=out, tile_cfg=13)
elif shape == "ffn_out":
    a, w, out = mk(M, Fd), mk(D, Fd), torch.empty(M, D, device=dev, dtype=torch.bfloat16)
    fn = lambda: F.gemm(a, w, M, D, Fd, out=out, tile_cfg=13)
elif shape == "qkv":
    a, w, out = mk(M, D), mk(3 * D, D), torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16)
    fn = lambda: F.gemm(a, w, M, 3 * D, D, out=out, tile_cfg=13)
else:
    a, b = mk(M, Fd), mk(M, D)
    out = torch.zeros(Fd, D, device=dev)
    fn = lambda: F.gemm(a, b, Fd, D, M, a_tr=True, b_tr=True, out=out, split_k=4, tile_cfg=13)
for _ in range(6):
    fn()
torch.cuda.synchronize()
