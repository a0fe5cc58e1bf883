#!/usr/bin/env python3
"""Where one barrier interval of the complementary GEMM loop goes (tile_cfg 13 = stamped lab build of cfg 12):
per wave, s_memtime sums of [MFMA segment, read+request segment, counted wait, barrier] over the K loop."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch, hipvg
from hipvg import GemmDesc, ptr, stream, lib, check
dev = torch.device("cuda:0")
M, N = 16000, 1024
g = torch.Generator().manual_seed(0)
for K in (1024, 4096):
    A = torch.randn(M, K, generator=g).to(dev).bfloat16()
    B = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev).bfloat16()
    Cc = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
    nblk = ((M + 255) // 256) * ((N + 255) // 256)
    ws = torch.zeros(nblk * 8 * 16, device=dev)
    d = GemmDesc()
    d.A, d.B, d.C = ptr(A), ptr(B), ptr(Cc)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = K, K, N
    d.dtype, d.alpha, d.tile_cfg, d.split_k = 1, 1.0, 13, 1
    d.split_ws, d.split_ws_floats = ptr(ws), ws.numel()
    for _ in range(3):
        check(lib().vg_gemm(C.byref(d), stream()), "vg_gemm")
    torch.cuda.synchronize()
    t = ws.view(nblk, 8, 16).cpu()
    nint = 4 * (K // 64)
    for name, sl in (("X (waves 0-3: MFMA then reads)", slice(0, 4)), ("Y (waves 4-7: reads then MFMA)", slice(4, 8))):
        r = t[:, sl].reshape(-1, 16).median(0).values
        m = r[:4] / nint
        clk = r[8] / r[7] * 100.0
        print(f"K={K} {name}: per interval cycles  MFMA {m[0]:.0f}  reads+request {m[1]:.0f}  vmcnt {m[2]:.0f}  barrier {m[3]:.0f}  total {m.sum():.0f}")
        print(f"      entry->loop {r[4]:.0f} cyc, K loop {r[5]:.0f} cyc ({r[5] / (K // 64):.0f} per K tile), epilogue+drain {r[6]:.0f} cyc, "
              f"kernel {r[8]:.0f} cyc = {r[7] / 100:.1f} us -> in-kernel clock {clk:.0f} MHz")
