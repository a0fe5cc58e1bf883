#!/usr/bin/env python3
"""Lab: does the RELATIVE placement of the epilogue's two streams matter?  The x stored-derivative dgrad of the FFN
(reads 131 MB of derivative, writes 131 MB of result, same (row, column) at the same time) measured 144 us in one
process and 157 us in another on the same box.  Here the derivative and the result live in one pool at a controlled
distance: result = derivative + 131 MB + delta.  Same for the GELU forward's two results.
    python tools/lab/offset_sweep.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch

import hipvg
from hipvg import functional as F

dev = torch.device("cuda:0")
M, N, K = 16000, 4096, 1024
R, ITERS = 3, 6
hipvg.lib()
g = torch.Generator(device="cpu").manual_seed(0)
xs = [torch.randn(M, K, generator=g).to(dev).bfloat16() for _ in range(R)]
ws = [(torch.randn(N, K, generator=g) * K ** -0.5).to(dev).bfloat16() for _ in range(R)]
wt = [(torch.randn(K, N, generator=g) * K ** -0.5).to(dev).bfloat16() for _ in range(R)]
bias = torch.randn(N, device=dev)
nbytes = M * N * 2
SLACK = 8 << 20
pools = [torch.empty(2 * nbytes + 2 * SLACK, dtype=torch.uint8, device=dev) for _ in range(R)]


def views(pool, delta):
    base = (-pool.data_ptr()) % (2 << 20)                 # 2 MB aligned start
    a = pool[base:base + nbytes].view(torch.bfloat16).view(M, N)
    off = base + nbytes + delta
    b = pool[off:off + nbytes].view(torch.bfloat16).view(M, N)
    return a, b


def run(fns):
    for f in fns:
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = None
    for _ in range(2):
        a.record()
        for _ in range(ITERS):
            for f in fns:
                f()
        b.record()
        torch.cuda.synchronize()
        t = a.elapsed_time(b) / (ITERS * len(fns)) * 1e3
        best = t if best is None else min(best, t)
    return best


deltas = [0, 256, 1024, 4096, 8192, 16384, 65536, 262144, 1 << 20, (1 << 20) + 4096, (2 << 20), (2 << 20) + 65536 + 4096, 3 << 20]
for delta in deltas + deltas[:3]:
    pairs = [views(p, delta) for p in pools]
    for d, _ in pairs:
        d.copy_(torch.rand(M, N, device=dev).bfloat16())
    t_d = run([(lambda i=i: F.gemm(xs[i], wt[i], M, N, K, b_tr=True, out=pairs[i][1], tile_cfg=13, dact=hipvg.ACT_STORED,
                                   aux_in=pairs[i][0], colpart=[])) for i in range(R)])
    t_g = run([(lambda i=i: F.gemm(xs[i], ws[i], M, N, K, out=pairs[i][1], bias=bias, tile_cfg=13,
                                   act=hipvg.ACT_GELU | hipvg.ACT_SAVE_DERIV, aux_out=pairs[i][0])) for i in range(R)])
    t_p = run([(lambda i=i: F.gemm(xs[i], ws[i], M, N, K, out=pairs[i][1], bias=bias, tile_cfg=13)) for i in range(R)])
    print(f"delta {delta:>9d} B: dgrad x deriv + colpart {t_d:6.1f} us | fwd GELU + deriv {t_g:6.1f} us | fwd plain {t_p:6.1f} us", flush=True)
