#!/usr/bin/env python3
"""Lab: the two weight gradients of a conv bottleneck block (512 <-> 2048 channels, K = all frames) one by one
(split-K launches) against one grouped launch.  VG_GROUP_MIN_TILES=1 lets the grouped path take them."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch
import hipvg
from hipvg import functional as F
dev = torch.device("cuda:0")
hipvg.lib()
g = torch.Generator().manual_seed(0)
M, R = int(os.environ.get("M", "16000")), 6
mk = lambda *s: [torch.randn(*s, generator=g).to(dev).bfloat16() for _ in range(R)]
for shapes in ([(512, 2048), (2048, 512)], [(512, 2048), (2048, 512), (2048, 32)], [(1024, 1024), (512, 1024)], [(512, 512), (512, 512)]):
    dys = [mk(M, n) for n, k in shapes]
    xs = [mk(M, k) for n, k in shapes]
    ws = [torch.nn.Parameter(torch.zeros(n, k, device=dev)) for n, k in shapes]
    for w in ws:
        w.grad = torch.zeros_like(w)
    def one(i):
        for w, dy, x in zip(ws, dys, xs):
            F.sink_wgrad(w, dy[i], x[i])
    def grp(i):
        F.sink_wgrad_group([(w, dy[i], x[i]) for w, dy, x in zip(ws, dys, xs)])
    fl = sum(2.0 * M * n * k for n, k in shapes)
    for name, fn in (("one by one", one), ("grouped", grp)):
        for i in range(R):
            fn(i)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for i in range(R):
                fn(i)
            b.record()
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) / R * 1e-3)
        t = sorted(ts)[2]
        print(f"{shapes} {name:11s} {t * 1e6:7.1f} us {fl / t / 1e12:6.0f} TF", flush=True)
    # exactness of the grouped result
    for w in ws:
        w.grad.zero_()
    grp(0)
    torch.cuda.synchronize()
    err = max(((w.grad.double() - dy[0].double().T @ x[0].double()).abs().max() / (dy[0].double().T @ x[0].double()).abs().max()).item() for w, dy, x in zip(ws, dys, xs))
    print(f"   grouped relative error {err:.2e}")
