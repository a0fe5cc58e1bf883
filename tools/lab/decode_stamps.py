#!/usr/bin/env python3
"""Lab: in-kernel phase stamps of vg_attn_layer_decode (build: tools/lab/variant.sh dstamp vg_decode.hip -DVG_LAB_DSTAMP).
VG_LIB=tools/lab/lib_dstamp.so python tools/lab/decode_stamps.py   -> mean time between stamps, in us (100 MHz clock)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch

import hipvg
from hipvg import functional as F

dev = torch.device("cuda:0")
L = hipvg.lib()
B, H, Tmax, n = 8, 16, 600, int(os.environ.get("FRAMES", "400"))
D = 64 * H
g = torch.Generator().manual_seed(0)
layers = 16
ws = [dict(wqkv=(torch.randn(3 * D, D, generator=g) * D ** -0.5).to(dev).bfloat16(), wo=(torch.randn(D, D, generator=g) * D ** -0.5).to(dev).bfloat16(),
           kc=torch.randn(B, Tmax, D, generator=g).to(dev).bfloat16(), vc=torch.randn(B, Tmax, D, generator=g).to(dev).bfloat16()) for _ in range(layers)]
bq, bo, g1 = torch.zeros(3 * D, device=dev), torch.zeros(D, device=dev), torch.ones(D, device=dev)
x = torch.randn(B, D, generator=g).to(dev)
slopes = torch.tensor(F.alibi_slopes(H), dtype=torch.float32, device=dev)
pos = torch.full((B,), n, dtype=torch.int32, device=dev)
x1 = torch.zeros(B, D, device=dev)
z = torch.zeros(B, D, device=dev)
stamps = torch.zeros(B * H * 16, dtype=torch.int64, device=dev)
have = hasattr(L, "vg_lab_set_dstamp")
if have:
    L.vg_lab_set_dstamp.argtypes = [ctypes.c_void_p]
    L.vg_lab_set_dstamp(ctypes.c_void_p(stamps.data_ptr()))
acc = torch.zeros(9, dtype=torch.float64)
reps = 0
for it in range(4):
    for w in ws:
        x1.zero_()
        F.attention_layer_decode(x, g1, 1e-6, w["wqkv"], bq, w["wo"], bo, w["kc"], w["vc"], slopes, pos, H, x1, zero=z)
        if have and it > 0:
            torch.cuda.synchronize()
            st = stamps.view(B * H, 16)[:, :9].double().cpu()
            acc += (st - st[:, :1]).mean(0)
            reps += 1
torch.cuda.synchronize()
if have:
    names = ["entry", "x + rstd", "QKV rows", "prefetch + barrier", "cache walk", "merge", "out-proj", "atomics issued", "drained"]
    t = acc / reps / 100.0
    for i in range(9):
        print(f"{names[i]:22s} at {t[i]:7.2f} us   (+{t[i] - (t[i - 1] if i else 0):6.2f})")
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for it in range(5):
    for w in ws:
        F.attention_layer_decode(x, g1, 1e-6, w["wqkv"], bq, w["wo"], bo, w["kc"], w["vc"], slopes, pos, H, x1, zero=z)
b.record()
torch.cuda.synchronize()
print(f"back-to-back launches: {a.elapsed_time(b) / (5 * layers) * 1e3:.1f} us per launch")
