#!/usr/bin/env python3
"""Lab (diagnostic build: tools/lab/variant.sh stamps vg_gemm_ph.hip -DVG_LAB_STAMPS; VG_LIB=tools/lab/lib_stamps.so):
in-kernel time from block entry to the end of the main loop as a function of the number of K tiles, 4-round launches
(M = 16000, N = 4096) and one-round launches (N = 1024): slope = steady-state cost of a K tile, intercept = what a
tile pays before its pipeline runs."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch
import hipvg
from hipvg import functional as F
dev = torch.device("cuda:0")
hipvg.lib()
raw = ctypes.CDLL(hipvg.LIB_PATH)
stamps = torch.zeros(4096 * 8, dtype=torch.int64, device=dev)
raw.vg_lab_set_stamps.argtypes = [ctypes.c_void_p]
assert raw.vg_lab_set_stamps(stamps.data_ptr()) == 0
g = torch.Generator().manual_seed(0)
M, R = 16000, 3
for N in (4096, 1024):
    for mode in ("nt", "nn"):
        for K in (256, 512, 1024, 2048, 4096):
            xs = [torch.randn(M, K, generator=g).to(dev).bfloat16() for _ in range(R)]
            ws = [(torch.randn(N, K, generator=g) * K ** -0.5).to(dev).bfloat16() for _ in range(R)]
            if mode == "nn":
                ws = [w.T.contiguous() for w in ws]
            ys = [torch.empty(M, N, device=dev, dtype=torch.bfloat16) for _ in range(R)]
            fn = lambda i: F.gemm(xs[i], ws[i], M, N, K, out=ys[i], tile_cfg=13, b_tr=(mode == "nn"))
            for _ in range(2):
                for i in range(R):
                    fn(i)
            torch.cuda.synchronize()
            stamps.zero_()
            for i in range(R):
                fn(i)
            torch.cuda.synchronize()
            nblk = ((M + 255) // 256) * (N // 256)
            s = stamps.view(4096, 8)[:nblk].cpu()
            t0 = s[:, 0].min().item()
            ent, le, ee = [(s[:, k] - t0).double() / 100.0 for k in range(3)]
            loop, epi = (le - ent), (ee - le)
            first = ent < 1.0                      # blocks of the first round
            print(f"N={N} {mode} K={K:5d} ({K // 64:3d} K tiles): entry->loop end median {loop.median().item():6.2f} us "
                  f"(first round {loop[first].median().item():6.2f}, later {loop[~first].median().item() if (~first).any() else float('nan'):6.2f}) "
                  f"| epilogue {epi.median().item():5.2f} | launch {ee.max().item():6.1f} us", flush=True)
