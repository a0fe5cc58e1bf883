#!/usr/bin/env python3
"""Lab (diagnostic build: tools/lab/variant.sh stamps vg_gemm_ph.hip -DVG_LAB_STAMPS; VG_LIB=tools/lab/lib_stamps.so):
per-block wall-clock stamps (entry, main loop end, epilogue end) and hardware ids of a multi-round GEMM launch --
how long a CU spends in the epilogue and between two blocks."""
import collections, ctypes, os, sys
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
M, D, Fd = 16000, 1024, 4096
R = 4
mk = lambda *s: [torch.randn(*s, generator=g).to(dev).bfloat16() for _ in range(R)]
xs, hs = mk(M, D), mk(M, Fd)
us = [torch.empty(M, Fd, device=dev, dtype=torch.bfloat16) for _ in range(R)]
q3 = [torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16) for _ in range(R)]
ys = [torch.empty(M, D, device=dev, dtype=torch.bfloat16) for _ in range(R)]
w1 = [(torch.randn(Fd, D, generator=g) * D ** -0.5).to(dev).bfloat16() for _ in range(R)]
w2 = [(torch.randn(D, Fd, generator=g) * Fd ** -0.5).to(dev).bfloat16() for _ in range(R)]
wq = [(torch.randn(3 * D, D, generator=g) * D ** -0.5).to(dev).bfloat16() for _ in range(R)]
b1 = torch.randn(Fd, generator=g).to(dev)
cases = {
    "FFN-in fwd NT N=4096 K=1024 +GELU' (cfg 13)": (lambda i: F.gemm(xs[i], w1[i], M, Fd, D, bias=b1, act=2 | 16, aux_out=us[i], out=hs[i], tile_cfg=13), 1008),
    "FFN-in fwd plain epilogue (cfg 13)": (lambda i: F.gemm(xs[i], w1[i], M, Fd, D, out=hs[i], tile_cfg=13), 1008),
    "QKV fwd NT N=3072 K=1024 (cfg 13)": (lambda i: F.gemm(xs[i], wq[i], M, 3 * D, D, out=q3[i], tile_cfg=13), 756),
    "dgrad->hid NN N=4096 K=1024 *GELU' (cfg 13)": (lambda i: F.gemm(ys[i], w2[i], M, Fd, D, b_tr=True, dact=4, aux_in=us[i], out=hs[i], tile_cfg=13), 1008),
    "FFN-out fwd NT N=1024 K=4096 (cfg 13)": (lambda i: F.gemm(hs[i], w2[i], M, D, Fd, out=ys[i], tile_cfg=13), 252),
}
for name, (fn, nblk) in cases.items():
    for _ in range(2):
        for i in range(R):
            fn(i)
    torch.cuda.synchronize()
    stamps.zero_()
    for i in range(R):
        fn(i)                       # back to back: the stamps of the last launch stay
    torch.cuda.synchronize()
    s = stamps.view(4096, 8)[:nblk].cpu()
    t0 = s[:, 0].min().item()
    ent, le, ee = [(s[:, k] - t0).double() / 100.0 for k in range(3)]
    cu = collections.defaultdict(list)
    for b in range(nblk):
        hw, xcc = s[b, 3].item(), s[b, 4].item() & 15
        key = (xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15)
        cu[key].append((ent[b].item(), le[b].item(), ee[b].item()))
    gaps, per_cu = [], []
    for key, v in cu.items():
        v.sort()
        per_cu.append(len(v))
        for a, b in zip(v, v[1:]):
            gaps.append(b[0] - a[2])
    med = lambda t: t.median().item()
    print(f"== {name}: {nblk} blocks on {len(cu)} CUs ({min(per_cu)}..{max(per_cu)} per CU), launch {ee.max().item():.1f} us")
    print(f"   entry->loop end  median {med(le - ent):6.1f} us   (min {(le - ent).min().item():.1f}, max {(le - ent).max().item():.1f})")
    print(f"   epilogue         median {med(ee - le):6.1f} us   (min {(ee - le).min().item():.1f}, max {(ee - le).max().item():.1f})")
    if gaps:
        gt = torch.tensor(gaps)
        print(f"   gap between a block's end and the next block's entry on the same CU: median {gt.median().item():.1f} us (min {gt.min().item():.1f}, max {gt.max().item():.1f})")
