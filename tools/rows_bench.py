#!/usr/bin/env python3
"""Micro-benchmark of the HBM-bound row kernels (rmsnorm, colsum, dwnorm) at the
full-config shape, against their algorithmic byte counts.  GPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch

import hipvg
from hipvg import functional as F

dev = torch.device("cuda:0")
ITERS = int(os.environ.get("ITERS", "50"))


def timeit(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(ITERS):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / ITERS * 1e-3


def report(name, t, nbytes):
    print(f"{name:34s} {t * 1e6:8.1f} us  {nbytes / t / 1e12:6.2f} TB/s", flush=True)


def main():
    L = hipvg.lib()
    st = hipvg.stream()
    p = hipvg.ptr
    M, T = int(os.environ.get("M", "8000")), 1000
    lengths = torch.full((M // T,), T, dtype=torch.int32, device=dev)
    for C in (1024, 512):
        x = torch.randn(M, C, device=dev).bfloat16()
        dy = torch.randn(M, C, device=dev).bfloat16()
        add = torch.randn(M, C, device=dev).bfloat16()
        sc = torch.rand(C, device=dev) + 0.5
        y = torch.empty_like(x)
        dx = torch.empty_like(x)
        rstd = torch.empty(M, dtype=torch.float32, device=dev)
        nb = L.vg_rmsnorm_bwd_blocks(M)
        part = torch.empty(nb, C, dtype=torch.float32, device=dev)
        out = torch.zeros(C, dtype=torch.float32, device=dev)
        ws = torch.empty(L.vg_colsum_blocks(M), C, dtype=torch.float32, device=dev)
        t = timeit(lambda: L.vg_rmsnorm_fwd(p(x), p(sc), p(y), p(rstd), M, C, 1e-6, p(lengths), T, 1, st))
        report(f"rmsnorm_fwd C={C}", t, 2 * M * C * 2)
        t = timeit(lambda: L.vg_rmsnorm_bwd(p(dy), p(x), p(sc), p(rstd), p(add), p(dx), p(part), M, C, p(lengths), T, 1, st))
        report(f"rmsnorm_bwd C={C} (+dx_add)", t, 4 * M * C * 2)
        # cold operands: rotate over R buffer sets (> the 256 MB Infinity Cache), as inside the training step
        R = 8
        xs, dys, adds, ys_, dxs = ([torch.randn(M, C, device=dev).bfloat16() for _ in range(R)] for _ in range(5))
        k = [0]
        def fwd_cold():
            i = k[0] = (k[0] + 1) % R
            L.vg_rmsnorm_fwd(p(xs[i]), p(sc), p(ys_[i]), p(rstd), M, C, 1e-6, p(lengths), T, 1, st)
        def bwd_cold():
            i = k[0] = (k[0] + 1) % R
            L.vg_rmsnorm_bwd(p(dys[i]), p(xs[i]), p(sc), p(rstd), p(adds[i]), p(dxs[i]), p(part), M, C, p(lengths), T, 1, st)
        report(f"rmsnorm_fwd C={C} cold", timeit(fwd_cold), 2 * M * C * 2)
        report(f"rmsnorm_bwd C={C} cold", timeit(bwd_cold), 4 * M * C * 2)
        del xs, dys, adds, ys_, dxs
        t = timeit(lambda: L.vg_colsum(p(part), nb, C, C, None, p(out), 0, 1, st))
        report(f"colsum partial [{nb}x{C}] f32", t, nb * C * 4)
        t = timeit(lambda: L.vg_colsum(p(dy), M, C, C, p(ws), p(out), 1, 1, st))
        report(f"colsum [{M}x{C}] bf16", t, M * C * 2)
    # depthwise conv (7 taps, causal) + channel norm of the posterior encoder / UNet blocks, C = 512
    C, taps = 512, 7
    x = torch.randn(M, C, device=dev).bfloat16()
    dy = torch.randn(M, C, device=dev).bfloat16()
    w = torch.randn(C, taps, device=dev) * 0.3
    cb, gamma, beta = torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1
    te = torch.randn(8, C, device=dev) * 0.1
    y, du, dx = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
    mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    nbk = L.vg_dwnorm_blocks(M)
    npart, wpart = torch.empty(nbk, 2 * C, device=dev), torch.empty(nbk, C * taps, device=dev)
    t = timeit(lambda: L.vg_dwnorm_fwd(p(x), p(w), p(cb), p(te), p(gamma), p(beta), p(y), p(mean), p(rstd), M, C, T,
                                       taps, taps - 1, 1e-5, 1, st))
    report("dwnorm_fwd C=512 k=7", t, 2 * M * C * 2)
    t = timeit(lambda: L.vg_dwnorm_bwd(p(dy), p(x), p(w), p(cb), p(te), p(gamma), p(mean), p(rstd), p(x), p(du), p(dx),
                                       p(npart), p(wpart), M, C, T, taps, taps - 1, 1, st))
    report("dwnorm_bwd (norm + conv) C=512 k=7", t, 7 * M * C * 2)
    C = 4096
    dy = torch.randn(M, C, device=dev).bfloat16()
    out = torch.zeros(C, dtype=torch.float32, device=dev)
    ws = torch.empty(L.vg_colsum_blocks(M), C, dtype=torch.float32, device=dev)
    t = timeit(lambda: L.vg_colsum(p(dy), M, C, C, p(ws), p(out), 1, 1, st))
    report(f"colsum [{M}x{C}] bf16", t, M * C * 2)
    # stock comparison points
    x = torch.randn(M, 1024, device=dev).bfloat16()
    y = torch.empty_like(x)
    t = timeit(lambda: y.copy_(x))
    report("torch copy_ [8000x1024] bf16", t, 2 * M * 1024 * 2)
    z = torch.empty(1024, dtype=torch.float32, device=dev)
    t = timeit(lambda: z.zero_())
    report("torch zero_ [1024] f32 (launch floor)", t, 4096)


if __name__ == "__main__":
    main()
