#!/usr/bin/env python3
"""Micro-benchmark of the attention kernels (forward, backward) at the full-config
shape; algorithmic (causal-exact) FLOPs.  GPU only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
    return a.elapsed_time(b) / ITERS * 1e-3


def main():
    hipvg.lib()
    H = 16
    D = H * 64
    shapes = [(8, 250), (8, 500), (8, 1000), (16, 1000), (4, 2000), (8, 2000)]
    if os.environ.get("SHAPES"):          # e.g. SHAPES=16x1000,8x2000
        shapes = [tuple(int(v) for v in s.split("x")) for s in os.environ["SHAPES"].split(",")]
    for (B, T) in shapes:
        g = torch.Generator(device="cpu").manual_seed(0)
        qkv = (torch.randn(B * T, 3 * D, generator=g) * float(os.environ.get("STD", "1.0"))).to(dev).bfloat16()
        dout = torch.randn(B * T, D, generator=g).to(dev).bfloat16()
        slopes = torch.tensor(F.alibi_slopes(H), dtype=torch.float32, device=dev)
        out = torch.empty(B * T, D, dtype=torch.bfloat16, device=dev)
        ws = F.attn_workspace(B, T, H, B * T, dev)        # log-sum-exp rows + the statistics of the backward's ALiBi window
        dqkv = torch.empty_like(qkv)
        delta = torch.empty(H, B * T, dtype=torch.float32, device=dev)
        # the entry points the model uses (round 5): vg_attn_fwd_stats / vg_attn_bwd_stats; STD=<sigma> scales q / k / v
        fwd = lambda: F.attn_fwd_raw(qkv, out, ws, slopes, B, T, H, None)
        bwd = lambda: F.attn_bwd_raw(qkv, out, dout, ws, slopes, dqkv, delta, B, T, H, None)
        fl = 256.0 * B * H * 0.5 * T * (T + 1)
        tf, tb = timeit(fwd), timeit(bwd)
        print(f"B={B} T={T}: fwd {tf*1e6:7.1f} us {fl/tf/1e12:6.1f} TF | bwd {tb*1e6:7.1f} us {2.0*fl/tb/1e12:6.1f} TF (2 x forward, SURVEY 8d)", flush=True)


if __name__ == "__main__":
    main()
