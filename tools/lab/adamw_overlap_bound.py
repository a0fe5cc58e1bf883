#!/usr/bin/env python3
"""Lab (TIMING ONLY, the update races with the next step's reads): upper bound of what overlapping the fused AdamW
with the next step's forward could give.  The optimizer's launches go to a side stream that the next graph replay does
NOT wait for (mode 1), or waits for only after `--delay-kernels`-worth of its own work cannot be expressed -- so just
the two extremes: serial (mode 0) and free-running (mode 1).
    python tools/lab/adamw_overlap_bound.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch

import hipvg
from hparams.hp import Hparams
from trainers.speech.lvtr import LVTRTrainer
from training_lib.synthetic import make_batch

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
hipvg.lib()
hp = Hparams.from_yamlfile(os.path.join(ROOT, "vae-gslm_amd", "configs", "train", "speech", "vae-gslm.yaml"))
hp.hip.precision = "bf16"
hp.hip.graph = True
torch.manual_seed(1234)
tr = LVTRTrainer(hp).to(dev)
tr.configure_optimizers()
tr.attach_reducer()
tr.global_step = hp.training.scheduler.warmup_kld
B, accum = hp.data.train.batch_size, tr.gradient_update_step
STEPS, WARM = 12, 3
batches = [make_batch(B, 1000, dev, seed=i) for i in range((STEPS + WARM) * accum)]
side = torch.cuda.Stream(device=dev)
orig = tr.optimizer.step
mode = [0]


def step(*a, **k):
    if mode[0] == 0:
        return orig(*a, **k)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        return orig(*a, **k)


tr.optimizer.step = step
for m in (0, 1, 0, 1):
    mode[0] = m
    it = 0
    for _ in range(WARM * accum):
        tr.training_step(batches[it], it); it += 1
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS * accum):
        tr.training_step(batches[it], it); it += 1
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / STEPS
    print(f"mode {m} ({'AdamW free-running on a side stream' if m else 'serial'}): {ms:.3f} ms per step", flush=True)
