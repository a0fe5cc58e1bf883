#!/usr/bin/env python3
"""Attention at the bench shape with the round-5 ALiBi window on / off: times, the window per head derived from the
forward's statistics, and the difference the window makes to the gradients (it must be below bf16 resolution).
GPU only.  SCALES=1.0,0.3 sets the standard deviation of the synthetic q / k / v."""
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch

import hipvg
from hipvg import functional as F

dev = torch.device("cuda:0")
ITERS = int(os.environ.get("ITERS", "20"))


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(ITERS):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / ITERS * 1e3


def main():
    L = hipvg.lib()
    H = 16
    D = H * 64
    shapes = [(16, 1000), (8, 2000), (16, 640)]
    if os.environ.get("SHAPES"):
        shapes = [tuple(int(v) for v in s.split("x")) for s in os.environ["SHAPES"].split(",")]
    scales = [float(v) for v in os.environ.get("SCALES", "1.0,0.3").split(",")]
    p, st = hipvg.ptr, hipvg.stream()
    for (B, T) in shapes:
        for sc in scales:
            g = torch.Generator(device="cpu").manual_seed(0)
            qkv = (torch.randn(B * T, 3 * D, generator=g) * sc).to(dev).bfloat16()
            dout = torch.randn(B * T, D, generator=g).to(dev).bfloat16()
            sl = F.alibi_slopes(H)
            if os.environ.get("SLOPE_ALL"):          # every head gets the slope of head SLOPE_ALL (cost of one head type)
                sl = [sl[int(os.environ["SLOPE_ALL"])]] * H
            slopes = torch.tensor(sl, dtype=torch.float32, device=dev)
            out = torch.empty(B * T, D, dtype=torch.bfloat16, device=dev)
            ws = F.attn_workspace(B, T, H, B * T, dev)
            dq = [torch.empty_like(qkv) for _ in range(2)]
            delta = torch.empty(H, B * T, dtype=torch.float32, device=dev)
            fwd = lambda: F.attn_fwd_raw(qkv, out, ws, slopes, B, T, H, None)
            bwd = lambda i: F.attn_bwd_raw(qkv, out, dout, ws, slopes, dq[i], delta, B, T, H, None)
            fl = 256.0 * B * H * 0.5 * T * (T + 1)
            tf = timeit(fwd)
            only = os.environ.get("ONLY")             # ONLY=1 / 0: time one of the two modes (for per-kernel traces)
            # A/B with alternating order (whatever runs second in a pair measures ~5 % faster: clocks, caches), medians
            t1, t0 = [], []
            for rep in range(6):
                for mode in (("0", "1") if rep % 2 == 0 else ("1", "0")):
                    os.environ["VG_ATTN_WINDOW"] = mode
                    if mode == "1" and only != "0":
                        t1.append(timeit(lambda: bwd(0)))
                    if mode == "0" and only != "1":
                        t0.append(timeit(lambda: bwd(1)))
            os.environ["VG_ATTN_WINDOW"] = "1"
            med = lambda v: sorted(v)[len(v) // 2] if v else 0.0
            tb1, tb0 = med(t1), med(t0)
            torch.cuda.synchronize()
            diff = (dq[0].float() - dq[1].float()).abs().max().item()
            ref = dq[1].float().abs().max().item()
            # the window per head of sequence 0, from the statistics
            n = L.vg_attn_stats_floats(1, T, 1)
            stats = ws[H * B * T:].view(B * H, n).cpu()
            n128 = (n - 4) // 8
            c2 = 0.125 * math.log2(math.e)
            wins = []
            for h in range(H):
                s = stats[h]
                k2, q2, nl = s[0].item(), s[4:4 + 4 * n128].max().item(), s[4 + 4 * n128:].max().item()
                wins.append((c2 * math.sqrt(q2) * math.sqrt(k2) * 1.01 + nl + 20.0) / (F.alibi_slopes(H)[h] * math.log2(math.e)))
            print(f"B={B} T={T} std={sc}: fwd {tf:6.1f} us {fl/tf/1e6:6.1f} TF | bwd window {tb1:6.1f} us {2*fl/tb1/1e6:6.1f} TF, "
                  f"no window {tb0:6.1f} us | max |d dqkv| {diff:.2e} of {ref:.2e}", flush=True)
            print("   window (frames) per head: " + " ".join(f"{min(w, 9999):.0f}" for w in wins), flush=True)


if __name__ == "__main__":
    main()
