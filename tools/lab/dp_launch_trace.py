#!/usr/bin/env python3
"""Lab: host-side order and times of the graph replays and bucket launches of the single-rank data-parallel step
(VG_DP_SINGLE_RANK=1): which buckets go out after which graph, and when the host gets there."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
os.environ["VG_DP_SINGLE_RANK"] = "1"
import torch
import torch.distributed as dist

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
comm = sys.argv[1] if len(sys.argv) > 1 else "abi"
if comm == "torch":
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29517", rank=0, world_size=1, device_id=dev)
import hipvg
from hparams.hp import Hparams
from trainers.speech.lvtr import LVTRTrainer
from training_lib.synthetic import make_batch
from training_lib import dp

hipvg.lib()
hp = Hparams.from_yamlfile(os.path.join(ROOT, "vae-gslm_amd", "configs", "train", "speech", "vae-gslm.yaml"))
hp.hip.precision = "bf16"
hp.hip.graph = True
hp.hip.comm = comm
torch.manual_seed(1234)
tr = LVTRTrainer(hp).to(dev)
tr.configure_optimizers()
red = tr.attach_reducer()
tr.global_step = hp.training.scheduler.warmup_kld
print("segmented", tr._segmented, "cuts", tr._cut_layers, "early", tr._early_buckets, "buckets", len(red.buckets), flush=True)
log = []
t0 = [0.0]
orig_launch = dp.GradReducer._launch


def launch(self, b):
    log.append((time.perf_counter() - t0[0], "launch bucket %d" % [i for i, x in enumerate(self.buckets) if x is b][0]))
    return orig_launch(self, b)


dp.GradReducer._launch = launch
orig_replay = torch.cuda.CUDAGraph.replay


def replay(self):
    a = time.perf_counter()
    r = orig_replay(self)
    log.append((a - t0[0], "replay (host %.2f ms)" % (1e3 * (time.perf_counter() - a))))
    return r


torch.cuda.CUDAGraph.replay = replay
B, accum = hp.data.train.batch_size, tr.gradient_update_step
batches = [make_batch(B, 1000, dev, seed=i) for i in range(8 * accum)]
for i in range(8 * accum):
    if i == 6 * accum:
        torch.cuda.synchronize()
        log.clear()
        t0[0] = time.perf_counter()
    tr.training_step(batches[i], i)
torch.cuda.synchronize()
print("total %.2f ms for 2 steps" % (1e3 * (time.perf_counter() - t0[0])))
for t, what in log:
    print(f"  +{1e3 * t:8.3f} ms  {what}")
