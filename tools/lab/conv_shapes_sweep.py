#!/usr/bin/env python3
"""Lab: the conv-stack products of the step (M = 16000 frames) on cold operands, per tile configuration, with the
epilogues they carry in the step: N=2048/K=512 SiLU + stored derivative (NT), x stored derivative (NN) and plain + ReLU,
N=512/K=2048 with a residual (NT) and plain (NN), N=512/K=512 plain.  CFGS=1,13,14,15 python tools/lab/conv_shapes_sweep.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch

import hipvg
from hipvg import functional as F

dev = torch.device("cuda:0")
M = int(os.environ.get("M", "16000"))
R, ITERS = 6, 5


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
cfgs = [int(v) for v in os.environ.get("CFGS", "0,1,13,14,15").split(",")]
for (N, K) in [(2048, 512), (512, 2048), (512, 512), (1024, 1024)]:
    xs = [torch.randn(M, K, generator=g).to(dev).bfloat16() for _ in range(R)]
    ws = [(torch.randn(N, K, generator=g) * K ** -0.5).to(dev).bfloat16() for _ in range(R)]
    wt = [(torch.randn(K, N, generator=g) * K ** -0.5).to(dev).bfloat16() for _ in range(R)]
    bias = torch.randn(N, device=dev)
    ys = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(R)]
    ds = [torch.rand(M, N, device=dev).bfloat16() for _ in range(R)]
    rs = [torch.randn(M, N, device=dev).bfloat16() for _ in range(R)]
    gf = 2e-9 * M * N * K
    for cfg in cfgs:
        t = {}
        try:
            t["NT plain"] = run([(lambda i=i: F.gemm(xs[i], ws[i], M, N, K, out=ys[i], bias=bias, tile_cfg=cfg)) for i in range(R)])
            t["NT +res"] = run([(lambda i=i: F.gemm(xs[i], ws[i], M, N, K, out=ys[i], bias=bias, residual=rs[i], tile_cfg=cfg)) for i in range(R)])
            t["NT silu+deriv"] = run([(lambda i=i: F.gemm(xs[i], ws[i], M, N, K, out=ys[i], bias=bias, tile_cfg=cfg,
                                                           act=hipvg.ACT_SILU | hipvg.ACT_SAVE_DERIV, aux_out=ds[i])) for i in range(R)])
            t["NN plain"] = run([(lambda i=i: F.gemm(xs[i], wt[i], M, N, K, b_tr=True, out=ys[i], tile_cfg=cfg)) for i in range(R)])
            t["NN x deriv+colpart"] = run([(lambda i=i: F.gemm(xs[i], wt[i], M, N, K, b_tr=True, out=ys[i], tile_cfg=cfg,
                                                                 dact=hipvg.ACT_STORED, aux_in=ds[i], colpart=[])) for i in range(R)])
        except Exception as e:          # a configuration that does not take this shape / epilogue
            t["error"] = 0.0
            print(f"N={N} K={K} cfg{cfg}: {str(e)[:100]}")
        print(f"N={N:5d} K={K:5d} cfg{cfg:2d}: " + " | ".join(f"{k} {v:6.1f}us {gf / v * 1e3 if v else 0:5.0f}TF" for k, v in t.items()), flush=True)
