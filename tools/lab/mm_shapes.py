#!/usr/bin/env python3
"""Lab: shapes and device time of the stock aten::mm / addmm / bmm calls of one eager optimizer step (torch profiler with
record_shapes): which products still run in the vendor library, and what they cost.  GPU only.
    python tools/lab/mm_shapes.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch
from torch.profiler import ProfilerActivity, profile

CONFIG = os.path.join(ROOT, "vae-gslm_amd", "configs", "train", "speech", "vae-gslm.yaml")
import hipvg
from hparams.hp import Hparams
from trainers.speech.lvtr import LVTRTrainer
from training_lib.synthetic import make_batch

hipvg.lib()
dev = torch.device("cuda:0")
hp = Hparams.from_yamlfile(CONFIG)
hp.hip.precision = "bf16"
hp.hip.graph = False
torch.manual_seed(1234)
tr = LVTRTrainer(hp).to(dev)
tr.configure_optimizers()
tr.attach_reducer()
tr.global_step = hp.training.scheduler.warmup_kld
B, accum = hp.data.train.batch_size, tr.gradient_update_step
batches = [make_batch(B, 1000, dev, seed=i) for i in range(2 * accum)]
for i in range(accum):
    tr.training_step(batches[i], i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for i in range(accum, 2 * accum):
        tr.training_step(batches[i], i)
    torch.cuda.synchronize()
for ev in prof.events():
    if ev.name in ("aten::mm", "aten::addmm", "aten::bmm", "aten::matmul", "aten::linear") and getattr(ev, "kernels", None):
        ks = ev.kernels
        par = ev.cpu_parent
        chain = []
        while par is not None and len(chain) < 3:
            chain.append(par.name)
            par = par.cpu_parent
        print(f"{ev.name:12s} shapes {ev.input_shapes}  device {sum(k.duration for k in ks):8.1f} us  kernels {[k.name[:40] for k in ks]}  < {' < '.join(chain)}")
