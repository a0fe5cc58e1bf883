#!/usr/bin/env python3
"""Lab: the conv block's (depthwise conv -> norm) backward as one launch (vg_dwnorm_bwd_fused) against the two run kernels, at
the step's shape (16 x 1000 frames, 512 channels), cold operands (8 rotating sets, > the 256 MB Infinity Cache).  GPU only.
    python tools/lab/dw_bwd_fused_ab.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch

import hipvg
from hipvg import functional as F

dev = torch.device("cuda:0")
hipvg.lib()
ITERS = int(os.environ.get("ITERS", "64"))
B, T, C = int(os.environ.get("B", "16")), int(os.environ.get("T", "1000")), 512
M = B * T
R = 8
torch.manual_seed(0)
w = torch.randn(C, 7, device=dev) * 0.3
cb, gamma, beta = torch.randn(C, device=dev) * 0.1, 1 + 0.1 * torch.randn(C, device=dev), 0.1 * torch.randn(C, device=dev)
te = torch.randn(B, C, device=dev) * 0.2
xs = [torch.randn(M, C, device=dev).bfloat16() for _ in range(R)]
dys = [torch.randn(M, C + 64, device=dev).bfloat16() for _ in range(R)]
adds = [torch.randn(M, C, device=dev).bfloat16() for _ in range(R)]
stats = [F.dwnorm_fwd_raw(x, w, cb, te, gamma, beta, T, 7, 3, 1e-6)[1:] for x in xs]


def run(fused, wide, want_du=True):
    F._DW_FUSED = fused
    k = [0]

    def once():
        i = k[0] % R
        k[0] += 1
        dy = dys[i][:, :C] if wide else dys[i][:, :C].contiguous()
        if fused and not want_du:
            return F._dwnorm_bwd_fused(dy, xs[i], w, cb, te, gamma, stats[i][0], stats[i][1], adds[i], T, 7, 3, want_du=False)
        return (F.dwnorm_bwd_ld_raw if wide else F.dwnorm_bwd_raw)(dy, xs[i], w, cb, te, gamma, stats[i][0], stats[i][1], adds[i], T, 7, 3)
    if not wide:       # (the contiguous copy is not part of the measurement: pre-make it)
        dense = [d[:, :C].contiguous() for d in dys]

        def once():
            i = k[0] % R
            k[0] += 1
            if fused and not want_du:
                return F._dwnorm_bwd_fused(dense[i], xs[i], w, cb, te, gamma, stats[i][0], stats[i][1], adds[i], T, 7, 3, want_du=False)
            return F.dwnorm_bwd_raw(dense[i], xs[i], w, cb, te, gamma, stats[i][0], stats[i][1], adds[i], T, 7, 3)
    for _ in range(8):
        once()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(ITERS):
        once()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / ITERS * 1e3


for rep in range(2):
    for wide in (False, True):
        t0 = run(False, wide)
        t1 = run(True, wide)
        t2 = run(True, wide, want_du=False)
        nbytes = M * C * 2 * 5
        print(f"B={B} T={T} {'ldy=576' if wide else 'dense  '}: two launches {t0:6.1f} us ({nbytes / t0 / 1e6:5.2f} TB/s on 5 streams) | "
              f"one launch {t1:6.1f} us ({nbytes / t1 / 1e6:5.2f}) | one launch, du not stored {t2:6.1f} us ({M * C * 2 * 4 / t2 / 1e6:5.2f} on 4 streams)",
              flush=True)
