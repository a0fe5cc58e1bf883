#!/usr/bin/env python3
"""Lab: the conv block's backward row kernels (dwnorm_bwd_norm_run + dwnorm_bwd_conv_run) at the step's shape on rotating
buffers; run under tools/lab/kstat.sh for per-kernel times (VG_LIB selects a variant library)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch, hipvg
from hipvg import functional as F
hipvg.lib()
d = torch.device("cuda:0")
M, C, T, R = 16000, 512, 1000, 8
xs = [torch.randn(M, C, device=d).bfloat16() for _ in range(R)]
dys = [torch.randn(M, C, device=d).bfloat16() for _ in range(R)]
adds = [torch.randn(M, C, device=d).bfloat16() for _ in range(R)]
w = torch.randn(C, 7, device=d) * 0.3
cb, gamma, beta = torch.randn(C, device=d) * 0.1, torch.rand(C, device=d) + 0.5, torch.randn(C, device=d)
te = torch.randn(M // T, C, device=d)
st = [F.dwnorm_fwd_raw(x, w, cb, te, gamma, beta, T, 7, 6, 1e-6) for x in xs]
for it in range(6):
    for i in range(R):
        F.dwnorm_bwd_raw(dys[i], xs[i], w, cb, te, gamma, st[i][1], st[i][2], adds[i], T, 7, 6)
torch.cuda.synchronize()
