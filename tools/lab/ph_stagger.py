#!/usr/bin/env python3
"""Effect of the first-round start stagger (VG_GEMM_STAGGER, ticks of 10 ns) on the multi-round layer GEMMs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch, hipvg
from hipvg import functional as F
dev = torch.device("cuda:0")
hipvg.lib()
g = torch.Generator().manual_seed(0)
M, D, Fd, R = 16000, 1024, 4096, 6
mk = lambda *s: [torch.randn(*s, generator=g).to(dev).bfloat16() for _ in range(R)]
xs, hs, c5 = mk(M, D), mk(M, Fd), mk(M, 512)
q3 = [torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16) for _ in range(R)]
us = [torch.empty(M, Fd, device=dev, dtype=torch.bfloat16) for _ in range(R)]
ys = [torch.empty(M, D, device=dev, dtype=torch.bfloat16) for _ in range(R)]
h2 = [torch.empty(M, 2048, device=dev, dtype=torch.bfloat16) for _ in range(R)]
u2 = [torch.empty(M, 2048, device=dev, dtype=torch.bfloat16) for _ in range(R)]
w1 = [(torch.randn(Fd, D, generator=g) * D ** -0.5).to(dev).bfloat16() for _ in range(R)]
w2 = [(torch.randn(D, Fd, generator=g) * Fd ** -0.5).to(dev).bfloat16() for _ in range(R)]
wq = [(torch.randn(3 * D, D, generator=g) * D ** -0.5).to(dev).bfloat16() for _ in range(R)]
wc = [(torch.randn(2048, 512, generator=g) * 512 ** -0.5).to(dev).bfloat16() for _ in range(R)]
b1 = torch.randn(Fd, generator=g).to(dev)
bc = torch.randn(2048, generator=g).to(dev)
cases = {
    "QKV fwd N=3072 K=1024": (2.0 * M * 3 * D * D, lambda i: F.gemm(xs[i], wq[i], M, 3 * D, D, out=q3[i])),
    "FFN-in fwd N=4096 K=1024 +GELU'": (2.0 * M * Fd * D, lambda i: F.gemm(xs[i], w1[i], M, Fd, D, bias=b1, act=2 | 16, aux_out=us[i], out=hs[i])),
    "FFN-out fwd N=1024 K=4096": (2.0 * M * Fd * D, lambda i: F.gemm(hs[i], w2[i], M, D, Fd, residual=xs[i], out=ys[i])),
    "dgrad->hid N=4096 K=1024 *GELU'": (2.0 * M * Fd * D, lambda i: F.gemm(ys[i], w2[i], M, Fd, D, b_tr=True, dact=4, aux_in=us[i], out=hs[i])),
    "conv fwd N=2048 K=512 +SiLU'": (2.0 * M * 2048 * 512, lambda i: F.gemm(c5[i], wc[i], M, 2048, 512, bias=bc, act=3 | 16, aux_out=u2[i], out=h2[i])),
    "conv dgrad N=2048 K=512": (2.0 * M * 2048 * 512, lambda i: F.gemm(c5[i], wc[i].t().contiguous(), M, 2048, 512, b_tr=True, dact=4, aux_in=u2[i], out=h2[i])),
}
line = f"stagger {os.environ.get('VG_GEMM_STAGGER', '0'):>5}:"
for name, (flop, fn) in cases.items():
    for i in range(R):
        fn(i)
    torch.cuda.synchronize()
    ts = []
    for _ in range(8):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for i in range(R):
            fn(i)
        b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / R * 1e3)
    line += f" | {name.split()[0]} {name.split()[1]} {sorted(ts)[4]:6.1f}"
print(line, flush=True)
