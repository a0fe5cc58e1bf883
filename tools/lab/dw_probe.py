import os, sys
sys.path.insert(0, "/root/repo/vae-gslm_amd")
import torch, hipvg
L = hipvg.lib(); st = hipvg.stream(); p = hipvg.ptr
dev = torch.device("cuda:0")
def timeit(fn, it=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it * 1e3
C = 512
for (B, T, taps, use_te) in [(8, 1000, 7, True), (2, 1000, 7, True), (32, 1000, 7, True), (8, 1000, 0, False), (8, 1000, 7, False), (8, 1000, 1, True), (8, 1000, 3, True)]:
    M = B * T
    x = torch.randn(M, C, device=dev).bfloat16()
    w = torch.randn(C, max(taps, 1), device=dev) * 0.3
    cb, gamma, beta = torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev)
    te = torch.randn(B, C, device=dev)
    y = torch.empty_like(x); mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
    t = timeit(lambda: L.vg_dwnorm_fwd(p(x), p(w) if taps else None, p(cb) if taps else None, p(te) if use_te else None, p(gamma), p(beta), p(y), p(mean), p(rstd), M, C, T, taps, max(taps - 1, 0), 1e-5, 1, st))
    print(f"B={B} T={T} taps={taps} temb={use_te}: {t:7.1f} us")
