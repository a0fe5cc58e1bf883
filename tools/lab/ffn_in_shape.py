#!/usr/bin/env python3
"""Lab: the FFN-in shape (M x 4096 x 1024, NT and NN) on cold operands next to N = 3072 and hipBLASLt, per 256-tile round.
VG_GEMM_GROUP_M is read once per process: run once per value.  LT=1 adds torch.matmul (kernel names under rocprofv3)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch

import hipvg
from hipvg import functional as F

dev = torch.device("cuda:0")
M = int(os.environ.get("M", "16000"))
LT = int(os.environ.get("LT", "0"))
CFG = int(os.environ.get("CFG", "0"))
R, ITERS = 6, 5


def run(fns):
    for f in fns:
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        a.record()
        for _ in range(ITERS):
            for f in fns:
                f()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / (ITERS * len(fns)) * 1e3)
    return sorted(ts)[1]


hipvg.lib()
g = torch.Generator(device="cpu").manual_seed(0)
tag = os.environ.get("VG_GEMM_GROUP_M", "4")
for (N, K) in [(3072, 1024), (4096, 1024), (1024, 4096)]:
    xs = [torch.randn(M, K, generator=g).to(dev).bfloat16() for _ in range(R)]
    ws = [(torch.randn(N, K, generator=g) * K ** -0.5).to(dev).bfloat16() for _ in range(R)]
    wt = [(torch.randn(K, N, generator=g) * K ** -0.5).to(dev).bfloat16() for _ in range(R)]
    ys = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(R)]
    gf = 2e-9 * M * N * K
    rounds = -(-(-(-M // 256) * (N // 256)) // 256)
    nt = run([(lambda i=i: F.gemm(xs[i], ws[i], M, N, K, out=ys[i], tile_cfg=CFG)) for i in range(R)])
    nn = run([(lambda i=i: F.gemm(xs[i], wt[i], M, N, K, b_tr=True, out=ys[i], tile_cfg=CFG)) for i in range(R)])
    line = f"group_m={tag} M={M} N={N:5d} K={K:5d} rounds={rounds} | NT {nt:6.1f} us ({nt / rounds:5.1f}/round) {gf / nt * 1e3:6.0f} TF | NN {nn:6.1f} us {gf / nn * 1e3:6.0f} TF"
    if LT:
        l1 = run([(lambda i=i: torch.matmul(xs[i], ws[i].T, out=ys[i])) for i in range(R)])
        l2 = run([(lambda i=i: torch.matmul(xs[i], wt[i], out=ys[i])) for i in range(R)])
        line += f" | hipBLASLt NT {l1:6.1f} NN {l2:6.1f}"
    print(line, flush=True)
