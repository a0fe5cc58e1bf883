import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch
import hipvg
from hipvg import functional as F
hipvg.lib()
d = torch.device("cuda:0")
for (B, T, shift) in ((16, 1000, 3), (1, 1000, 3), (3, 120, 3)):
    torch.manual_seed(1)
    C, M = 512, B * T
    x = torch.randn(M, C, device=d).bfloat16()
    w = torch.randn(C, 7, device=d) * 0.3
    cb, gamma, beta = torch.randn(C, device=d) * 0.1, 1 + 0.1 * torch.randn(C, device=d), 0.1 * torch.randn(C, device=d)
    te = torch.randn(B, C, device=d) * 0.2
    _, mean, rstd = F.dwnorm_fwd_raw(x, w, cb, te, gamma, beta, T, 7, shift, 1e-6)
    dy = torch.randn(M, C, device=d).bfloat16()
    dxa = torch.randn(M, C, device=d).bfloat16()
    F._DW_FUSED = False
    du0, dx0, pg0, pb0, pw0 = F.dwnorm_bwd_raw(dy, x, w, cb, te, gamma, mean, rstd, dxa, T, 7, shift)
    F._DW_FUSED = True
    du1, dx1, pg1, pb1, pw1 = F.dwnorm_bwd_raw(dy, x, w, cb, te, gamma, mean, rstd, dxa, T, 7, shift)
    torch.cuda.synchronize()
    print(f"B={B} T={T} shift={shift}")
    for name, a, b in (("du", du0, du1), ("dx", dx0, dx1)):
        diff = (a.float() - b.float()).abs()
        rows = diff.amax(1)
        bad = torch.nonzero(rows > 0).flatten().tolist()
        print(f"  {name}: max diff {float(diff.max()):.4g} of {float(a.float().abs().max()):.3g}; bad rows {bad[:40]}{'...' if len(bad) > 40 else ''} ({len(bad)})")
        if bad:
            r = bad[0]
            cols = torch.nonzero(diff[r] > 0).flatten().tolist()
            print(f"     row {r}: bad cols {cols[:16]} ({len(cols)}); want {a[r, cols[:4]].tolist()} got {b[r, cols[:4]].tolist()}")
    for name, a, b in (("pg", pg0, pg1), ("pb", pb0, pb1), ("pw", pw0, pw1)):
        want, got = a.double().sum(0), b.double().sum(0)
        print(f"  {name}: max diff {float((want - got).abs().max()):.4g} of {float(want.abs().max()):.4g}")
