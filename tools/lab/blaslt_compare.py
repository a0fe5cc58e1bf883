#!/usr/bin/env python3
"""Lab: the layer's forward / dgrad products on cold operands -- this library (tile_cfg auto) against torch.matmul
(hipBLASLt) on the same operands.  Under rocprofv3 --kernel-trace the hipBLASLt kernel names give its macro tile.
M=16000 python tools/lab/blaslt_compare.py"""
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
for (N, K) in [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096), (1024, 3072), (2048, 512), (512, 2048)]:
    xs = [torch.randn(M, K, generator=g).to(dev).bfloat16() for _ in range(R)]
    ws = [(torch.randn(N, K, generator=g) * K ** -0.5).to(dev).bfloat16() for _ in range(R)]
    wt = [(torch.randn(K, N, generator=g) * K ** -0.5).to(dev).bfloat16() for _ in range(R)]
    ys = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(R)]
    gf = 2e-9 * M * N * K
    mine_nt = run([(lambda i=i: F.gemm(xs[i], ws[i], M, N, K, out=ys[i])) for i in range(R)])
    lt_nt = run([(lambda i=i: torch.matmul(xs[i], ws[i].T, out=ys[i])) for i in range(R)])
    mine_nn = run([(lambda i=i: F.gemm(xs[i], wt[i], M, N, K, b_tr=True, out=ys[i])) for i in range(R)])
    lt_nn = run([(lambda i=i: torch.matmul(xs[i], wt[i], out=ys[i])) for i in range(R)])
    print(f"M={M} N={N:5d} K={K:5d} | NT mine {mine_nt:6.1f} us {gf / mine_nt * 1e3:6.0f} TF  hipBLASLt {lt_nt:6.1f} us {gf / lt_nt * 1e3:6.0f} TF"
          f" | NN mine {mine_nn:6.1f} us {gf / mine_nn * 1e3:6.0f} TF  hipBLASLt {lt_nn:6.1f} us {gf / lt_nn * 1e3:6.0f} TF", flush=True)
