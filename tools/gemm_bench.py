#!/usr/bin/env python3
"""Micro-benchmark of vg_gemm on the Transformer shapes of the full config
(M = 8000 frames): every operand mode x tile configuration, random data,
checked against the register-staged kernel.  GPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch

import hipvg
from hipvg import functional as F

dev = torch.device("cuda:0")
M = int(os.environ.get("M", "8000"))
SHAPES = [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)]     # (N_out_features, K_in_features)
CFGS = [int(c) for c in os.environ.get("CFGS", "-1,1,2,3,4").split(",")]
ITERS = int(os.environ.get("ITERS", "20"))


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(ITERS):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / ITERS * 1e-3


def main():
    hipvg.lib()
    g = torch.Generator(device="cpu").manual_seed(0)
    for (N, K) in SHAPES:
        x = torch.randn(M, K, generator=g).to(dev).bfloat16()
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev).bfloat16()
        dy = torch.randn(M, N, generator=g).to(dev).bfloat16()
        flops = 2.0 * M * N * K
        cases = {
            "NT fwd  ": lambda cfg: F.gemm(x, w, M, N, K, tile_cfg=cfg),
            "NN dgrad": lambda cfg: F.gemm(dy, w, M, K, N, b_tr=True, tile_cfg=cfg),
            "TN wgrad": lambda cfg: F.gemm(dy, x, N, K, M, a_tr=True, b_tr=True, out_f32=True,
                                           split_k=F.wgrad_splits(N, K, M, torch.bfloat16), tile_cfg=cfg),
        }
        if os.environ.get("TORCH_REF", "1") == "1":    # hipBLASLt through torch, same operands (reference point)
            tt = [timeit(lambda: torch.matmul(x, w.t())), timeit(lambda: torch.matmul(dy, w)),
                  timeit(lambda: torch.matmul(dy.t(), x))]
            print(f"N={N:5d} K={K:5d} hipBLASLt | " + " | ".join(
                f"{n}: {flops / t / 1e12:7.1f} TF ({t * 1e6:6.1f} us)" for n, t in zip(("NT", "NN", "TN"), tt)), flush=True)
        aux = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        bias = torch.randn(N, generator=g).to(dev)
        res = torch.randn(M, N, generator=g).to(dev).bfloat16()
        pre = torch.randn(M, K, generator=g).to(dev).bfloat16()
        if os.environ.get("EPI", "0") == "1":
            cases = {
                "NT +bias+GELU+aux ": lambda cfg: F.gemm(x, w, M, N, K, bias=bias, act=2, aux_out=aux, tile_cfg=cfg),
                "NT +bias+residual ": lambda cfg: F.gemm(x, w, M, N, K, bias=bias, residual=res, tile_cfg=cfg),
                "NN +dGELU(aux)    ": lambda cfg: F.gemm(dy, w, M, K, N, b_tr=True, dact=2, aux_in=pre, tile_cfg=cfg),
            }
        for name, fn in cases.items():
            ref = fn(-1).float()
            row = []
            for cfg in CFGS:
                try:
                    out = fn(cfg).float()
                    err = (out - ref).abs().max().item() / (ref.abs().max().item() + 1e-9)
                    t = timeit(lambda: fn(cfg))
                    row.append(f"cfg{cfg:>2}: {flops / t / 1e12:7.1f} TF ({t * 1e6:6.1f} us, err {err:.1e})")
                except Exception as e:  # noqa
                    row.append(f"cfg{cfg:>2}: FAIL {str(e)[:40]}")
            print(f"N={N:5d} K={K:5d} {name} | " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
