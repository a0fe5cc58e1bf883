#!/usr/bin/env python3
"""tile_cfg 6 / 7 (32-deep tiles, 4-stage ring) against tile_cfg 3 / 1: same results on odd shapes, K tails and
split-K; time per launch on the layer shapes with cold operands.  GPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch

import hipvg
from hipvg import functional as F

dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(0)


def rnd(*shape):
    return torch.randn(*shape, generator=g).to(dev).bfloat16()


def check():
    bad = 0
    for (M, N, K) in [(512, 512, 64), (520, 264, 96), (1000, 776, 1000), (256, 256, 32), (300, 200, 40), (2048, 1024, 4096),
                      (777, 1032, 72)]:
        for mode in ("NT", "NN", "TN"):
            a_tr, b_tr = mode == "TN", mode != "NT"
            if a_tr and (M % 8 or N % 8):
                continue
            if b_tr and N % 8:
                continue
            A = rnd(K, M) if a_tr else rnd(M, K)
            B = rnd(K, N) if b_tr else rnd(N, K)
            for split in ((1, 3) if mode == "TN" else (1,)):
                outs = {}
                for cfg in (1, 3, 6, 7):
                    out = torch.zeros(M, N, device=dev) if split > 1 else torch.empty(M, N, device=dev, dtype=torch.bfloat16)
                    F.gemm(A, B, M, N, K, a_tr=a_tr, b_tr=b_tr, out=out, tile_cfg=cfg, split_k=split)
                    outs[cfg] = out.float()
                ref = (A.float().t() if a_tr else A.float()) @ (B.float() if b_tr else B.float().t())
                for cfg in (3, 6, 7):
                    err = (outs[cfg] - outs[1]).abs().max().item()
                    tol = 0.02 * ref.abs().max().item() if split > 1 else 0.0
                    exact = torch.equal(outs[cfg], outs[1]) if split == 1 else err <= max(tol, 1e-2)
                    if not exact:
                        bad += 1
                        print(f"MISMATCH {mode} M={M} N={N} K={K} split={split} cfg{cfg}: max diff {err:.4g}")
                e1 = (outs[1] - ref).abs().max().item() / ref.abs().max().item()
                assert e1 < 2e-2, (mode, M, N, K, e1)
    print("check:", "OK" if bad == 0 else f"{bad} mismatches", flush=True)


def timeit(fns, iters=4):
    for f in fns:
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        for f in fns:
            f()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / (iters * len(fns)) * 1e3


def bench():
    M, R = 16000, 6
    for (N, K) in [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096), (1024, 3072)]:
        for mode in ("NT", "NN"):
            As = [rnd(M, K) for _ in range(R)]
            Bs = [rnd(K, N) if mode == "NN" else rnd(N, K) for _ in range(R)]
            Cs = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(R)]
            row = []
            for cfg in (1, 3, 6, 7):
                t = timeit([(lambda i=i: F.gemm(As[i], Bs[i], M, N, K, b_tr=(mode == "NN"), out=Cs[i], tile_cfg=cfg))
                            for i in range(R)])
                row.append(f"cfg{cfg}: {t:6.1f}")
            print(f"M={M} N={N:5d} K={K:5d} {mode} | " + " | ".join(row) + " us", flush=True)
    for (N, K) in [(4096, 1024), (1024, 4096), (3072, 1024), (1024, 1024)]:      # weight gradients
        dys = [rnd(M, N) for _ in range(R)]
        xs = [rnd(M, K) for _ in range(R)]
        gws = [torch.zeros(N, K, device=dev) for _ in range(R)]
        row = []
        for cfg, s in ((1, 2), (7, 2), (6, 4), (3, 4), (1, 8), (7, 8)):
            t = timeit([(lambda i=i: F.gemm(dys[i], xs[i], N, K, M, a_tr=True, b_tr=True, out=gws[i], split_k=s,
                                            tile_cfg=cfg)) for i in range(R)])
            row.append(f"cfg{cfg} s{s}: {t:6.1f}")
        print(f"dW[{N}x{K}] | " + " | ".join(row) + " us", flush=True)


if __name__ == "__main__":
    hipvg.lib()
    check()
    if "--bench" in sys.argv:
        bench()
