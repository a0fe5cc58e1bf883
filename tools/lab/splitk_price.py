#!/usr/bin/env python3
"""Lab: what would splitting K over two blocks buy the half-empty launches of the conv stacks (N = 512, K = 2048: 126 tiles
of 256 x 256 on 256 CUs)?  Priced with what exists: the fp32-accumulate split-K launch with its in-launch slab reduction."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch, hipvg
from hipvg import functional as F
hipvg.lib()
d = torch.device("cuda:0")
M, R = 16000, 6
def run(fns, it=5):
    for f in fns: f()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(3):
        a.record()
        for _ in range(it):
            for f in fns: f()
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / (it * len(fns)) * 1e3)
    return sorted(ts)[1]
g = torch.Generator().manual_seed(0)
for (N, K) in [(512, 2048), (512, 512), (1024, 1024)]:
    xs = [torch.randn(M, K, generator=g).to(d).bfloat16() for _ in range(R)]
    ws = [(torch.randn(N, K, generator=g) * K ** -0.5).to(d).bfloat16() for _ in range(R)]
    wt = [(torch.randn(K, N, generator=g) * K ** -0.5).to(d).bfloat16() for _ in range(R)]
    yb = [torch.empty(M, N, device=d, dtype=torch.bfloat16) for _ in range(R)]
    yf = [torch.zeros(M, N, device=d) for _ in range(R)]
    row = []
    for tag, kw in (("NT", {}), ("NN", {"b_tr": True})):
        B = ws if tag == "NT" else wt
        t0 = run([(lambda i=i: F.gemm(xs[i], B[i], M, N, K, out=yb[i], **kw)) for i in range(R)])
        t1 = run([(lambda i=i: F.gemm(xs[i], B[i], M, N, K, out=yf[i], accumulate=True, tile_cfg=13, **kw)) for i in range(R)])
        t2 = run([(lambda i=i: F.gemm(xs[i], B[i], M, N, K, out=yf[i], split_k=2, tile_cfg=13, **kw)) for i in range(R)])
        t4 = run([(lambda i=i: F.gemm(xs[i], B[i], M, N, K, out=yf[i], split_k=4, tile_cfg=13, **kw)) for i in range(R)])
        row.append(f"{tag}: bf16 out (auto cfg) {t0:5.1f} | fp32 += 1 split {t1:5.1f} | 2 splits {t2:5.1f} | 4 splits {t4:5.1f}")
    print(f"N={N} K={K}: " + "  ||  ".join(row), flush=True)
