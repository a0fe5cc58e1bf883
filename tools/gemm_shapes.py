#!/usr/bin/env python3
"""Diagnostic: per-shape time of every vg_gemm call in one eager training micro-step
(full config, B=8, T=1000).  GPU only."""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch

import hipvg
from hipvg import functional as F
from hparams.hp import Hparams
from trainers.speech.lvtr import LVTRTrainer
from training_lib.synthetic import make_batch

dev = torch.device("cuda:0")
hp = Hparams.from_yamlfile(os.path.join(ROOT, "vae-gslm_amd/configs/train/speech/vae-gslm.yaml"))
hp.hip.graph = False
trainer = LVTRTrainer(hp).to(dev)
trainer.configure_optimizers()
trainer.attach_reducer()
trainer.global_step = 30000
batches = [make_batch(8, 1000, dev, seed=i) for i in range(4)]
for i in range(2):
    trainer.training_step(batches[i], i)
torch.cuda.synchronize()

records = []
orig = F.gemm


def timed(A, B, M, N, K, **kw):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    out = orig(A, B, M, N, K, **kw)
    b.record()
    mode = ("T" if kw.get("a_tr") else "N") + ("T" if not kw.get("b_tr") else "N")
    mode = {"NT": "NT fwd", "NN": "NN dgrad", "TN": "TN wgrad"}.get(("T" if kw.get("a_tr") else "N") + ("N" if kw.get("b_tr") else "T"), mode)
    records.append((mode, M, N, K, str(A.dtype)[6:], kw.get("split_k", 1), a, b))
    return out


F.gemm = timed
for i in range(2, 4):
    trainer.training_step(batches[i], i)
torch.cuda.synchronize()
agg = collections.OrderedDict()
for mode, M, N, K, dt, s, a, b in records:
    key = (mode, M, N, K, dt, s)
    t = a.elapsed_time(b) * 1e-3
    e = agg.setdefault(key, [0, 0.0])
    e[0] += 1
    e[1] += t
tot = sum(v[1] for v in agg.values())
print(f"total gemm time per micro-step: {tot / 2 * 1e3:.2f} ms over {len(records) // 2} calls")
for key, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    mode, M, N, K, dt, s = key
    fl = 2.0 * M * N * K * n
    print(f"{t / 2 * 1e3:7.3f} ms/micro  {n // 2:3d} calls  {t / n * 1e6:7.1f} us  {fl / t / 1e12:7.1f} TF  {mode:9s} M={M:5d} N={N:5d} K={K:5d} {dt} split={s}")
