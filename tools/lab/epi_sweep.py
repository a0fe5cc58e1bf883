#!/usr/bin/env python3
"""Lab: cost of the heavy epilogues of the FFN-in products on cold operands, per tile configuration.
  forward  [M,1024] x [4096,1024]^T, GELU + stored derivative (two 131 MB results)
  dgrad    [M,1024] x [1024,4096] (NN), x stored derivative (131 MB extra read) + per-tile column sums
  plain    the same products without the extra operand
CFGS=13,14,15,3  M=16000  python tools/lab/epi_sweep.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch

import hipvg
from hipvg import functional as F

dev = torch.device("cuda:0")
M = int(os.environ.get("M", "16000"))
R, ITERS = 4, 5


def run(fns):
    for f in fns:
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = None
    for _ in range(2):           # twice, the faster one: the first timed pass over fresh buffers occasionally takes 5-50x
        a.record()               # (page mapping of the output tensors), for either library
        for _ in range(ITERS):
            for f in fns:
                f()
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / (ITERS * len(fns)) * 1e3
        best = t if best is None else min(best, t)
    return best


hipvg.lib()
g = torch.Generator(device="cpu").manual_seed(0)
N, K = 4096, 1024
xs = [torch.randn(M, K, generator=g).to(dev).bfloat16() for _ in range(R)]
ws = [(torch.randn(N, K, generator=g) * K ** -0.5).to(dev).bfloat16() for _ in range(R)]
wt = [(torch.randn(K, N, generator=g) * K ** -0.5).to(dev).bfloat16() for _ in range(R)]
bias = torch.randn(N, device=dev)
ys = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(R)]
ds = [torch.rand(M, N, device=dev).bfloat16() for _ in range(R)]
d8 = [torch.randint(0, 256, (M, N), device=dev, dtype=torch.int32).to(torch.uint8) for _ in range(R)]
for cfg in [int(v) for v in os.environ.get("CFGS", "13,14,15").split(",")]:
    t = {}
    t["fwd plain"] = run([(lambda i=i: F.gemm(xs[i], ws[i], M, N, K, out=ys[i], bias=bias, tile_cfg=cfg)) for i in range(R)])
    t["fwd gelu+deriv"] = run([(lambda i=i: F.gemm(xs[i], ws[i], M, N, K, out=ys[i], bias=bias, tile_cfg=cfg,
                                                    act=hipvg.ACT_GELU | hipvg.ACT_SAVE_DERIV, aux_out=ds[i])) for i in range(R)])
    if os.environ.get("U8", "1") == "1":         # round 6: the derivative as one byte per element (VG_ACT_DERIV_U8)
        t["fwd gelu+deriv u8"] = run([(lambda i=i: F.gemm(xs[i], ws[i], M, N, K, out=ys[i], bias=bias, tile_cfg=cfg,
                                                           act=hipvg.ACT_GELU | hipvg.ACT_SAVE_DERIV | hipvg.ACT_DERIV_U8, aux_out=d8[i])) for i in range(R)])
        t["dgrad x deriv u8"] = run([(lambda i=i: F.gemm(xs[i], wt[i], M, N, K, b_tr=True, out=ys[i], tile_cfg=cfg,
                                                          dact=hipvg.ACT_STORED | hipvg.ACT_DERIV_U8, aux_in=d8[i])) for i in range(R)])
        t["dgrad x deriv u8 + colpart"] = run([(lambda i=i: F.gemm(xs[i], wt[i], M, N, K, b_tr=True, out=ys[i], tile_cfg=cfg,
                                                                    dact=hipvg.ACT_STORED | hipvg.ACT_DERIV_U8, aux_in=d8[i], colpart=[])) for i in range(R)])
    t["dgrad plain"] = run([(lambda i=i: F.gemm(xs[i], wt[i], M, N, K, b_tr=True, out=ys[i], tile_cfg=cfg)) for i in range(R)])
    t["dgrad x deriv"] = run([(lambda i=i: F.gemm(xs[i], wt[i], M, N, K, b_tr=True, out=ys[i], tile_cfg=cfg,
                                                   dact=hipvg.ACT_STORED, aux_in=ds[i])) for i in range(R)])
    t["dgrad x deriv + colpart"] = run([(lambda i=i: F.gemm(xs[i], wt[i], M, N, K, b_tr=True, out=ys[i], tile_cfg=cfg,
                                                             dact=hipvg.ACT_STORED, aux_in=ds[i], colpart=[])) for i in range(R)])
    print(f"M={M} cfg{cfg}: " + " | ".join(f"{k} {v:6.1f}" for k, v in t.items()), flush=True)
