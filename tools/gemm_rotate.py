#!/usr/bin/env python3
"""Does a GEMM run slower on operands it has not touched recently?  Times the FFN-shaped launches with
one fixed operand set vs a rotation over R distinct sets (cold L2 / Infinity Cache / TLB), and with the
producer->consumer pattern of the step (the A operand is written by the preceding launch).  GPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch

import hipvg
from hipvg import functional as F

dev = torch.device("cuda:0")
M = int(os.environ.get("M", "8000"))
R = int(os.environ.get("R", "16"))
ITERS = int(os.environ.get("ITERS", "4"))


def run(label, fns):
    for f in fns:
        f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(ITERS):
        for f in fns:
            f()
    b.record()
    torch.cuda.synchronize()
    t = a.elapsed_time(b) / (ITERS * len(fns)) * 1e-3
    print(f"{label:58s} {t * 1e6:7.1f} us per launch", flush=True)
    return t


def main():
    hipvg.lib()
    g = torch.Generator(device="cpu").manual_seed(0)
    D, Fd = 1024, 4096
    xs = [torch.randn(M, D, generator=g).to(dev).bfloat16() for _ in range(R)]
    hs = [torch.randn(M, Fd, generator=g).to(dev).bfloat16() for _ in range(R)]
    us = [torch.empty(M, Fd, device=dev, dtype=torch.bfloat16) for _ in range(R)]
    ys = [torch.empty(M, D, device=dev, dtype=torch.bfloat16) for _ in range(R)]
    w1 = [(torch.randn(Fd, D, generator=g) * D ** -0.5).to(dev).bfloat16() for _ in range(R)]
    w2 = [(torch.randn(D, Fd, generator=g) * Fd ** -0.5).to(dev).bfloat16() for _ in range(R)]
    b1 = torch.randn(Fd, generator=g).to(dev)
    b2 = torch.randn(D, generator=g).to(dev)
    ffn_in = lambda i: (lambda: F.gemm(xs[i], w1[i], M, Fd, D, bias=b1, act=2 | 16, aux_out=us[i], out=hs[i]))
    ffn_out = lambda i: (lambda: F.gemm(hs[i], w2[i], M, D, Fd, bias=b2, residual=xs[i], out=ys[i]))
    dgrad_u = lambda i: (lambda: F.gemm(ys[i], w2[i], M, Fd, D, b_tr=True, dact=4, aux_in=us[i], out=hs[i]))
    wg = [torch.zeros(Fd, D, device=dev) for _ in range(R)]
    wgrad1 = lambda i: (lambda: F.gemm(hs[i], xs[i], Fd, D, M, a_tr=True, b_tr=True, out=wg[i], split_k=2))
    for name, mk in (("FFN-in  fwd (+GELU, stores h and GELU')", ffn_in), ("FFN-out fwd (+bias +residual)", ffn_out),
                     ("dgrad to the hidden width (* stored GELU')", dgrad_u), ("wgrad W1 (split-K 2, atomics)", wgrad1)):
        run(name + " | same operands", [mk(0)])
        run(name + f" | rotating over {R} sets", [mk(i) for i in range(R)])
    # tile configurations on cold operands (the heuristic in vg_gemm.hip was tuned on warm ones)
    for cfg in (1, 2, 3, 4, 5):
        run(f"FFN-out fwd cold, tile_cfg {cfg}", [(lambda i=i: F.gemm(hs[i], w2[i], M, D, Fd, bias=b2, residual=xs[i], out=ys[i], tile_cfg=cfg)) for i in range(R)])
    for cfg in (1, 2, 3, 4, 5):
        run(f"FFN-in fwd cold, tile_cfg {cfg}", [(lambda i=i: F.gemm(xs[i], w1[i], M, Fd, D, bias=b1, act=2 | 16, aux_out=us[i], out=hs[i], tile_cfg=cfg)) for i in range(R)])
    for cfg in (1, 2, 3, 4):
        run(f"dgrad->hidden cold, tile_cfg {cfg}", [(lambda i=i: F.gemm(ys[i], w2[i], M, Fd, D, b_tr=True, dact=4, aux_in=us[i], out=hs[i], tile_cfg=cfg)) for i in range(R)])
    for cfg in (1, 2, 3, 4):
        run(f"dgrad->model (K=4096) cold, tile_cfg {cfg}", [(lambda i=i: F.gemm(hs[i], w1[i], M, D, Fd, b_tr=True, out=ys[i], tile_cfg=cfg)) for i in range(R)])
    # the same question for hipBLASLt (through torch): is the cold-operand penalty a property of this kernel?
    run("hipBLASLt FFN-out (h @ W2^T) | same operands", [lambda: torch.matmul(hs[0], w2[0].t())])
    run(f"hipBLASLt FFN-out (h @ W2^T) | rotating over {R} sets", [(lambda i=i: torch.matmul(hs[i], w2[i].t())) for i in range(R)])
    run("hipBLASLt wgrad W1 (h^T @ x) | same operands", [lambda: torch.matmul(hs[0].t(), xs[0])])
    run(f"hipBLASLt wgrad W1 (h^T @ x) | rotating over {R} sets", [(lambda i=i: torch.matmul(hs[i].t(), xs[i])) for i in range(R)])
    seq = []
    for i in range(R):
        seq += [ffn_in(i), ffn_out(i)]
    t = run(f"FFN-in -> FFN-out chained, {R} layers (avg of both)", seq)


if __name__ == "__main__":
    main()
