#!/usr/bin/env python3
"""Where does the host input pipeline spend its time?  (make_batch on CPU, pin_memory, H2D copy, worker thread.)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch

from training_lib.prefetch import BackgroundLoader, DevicePrefetcher, pin_batch
from training_lib.synthetic import make_batch

dev = torch.device("cuda:0")
torch.zeros(1, device=dev)
N = 40


def timed(label, fn, n=N):
    t = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    print(f"{label:50s} {(time.perf_counter() - t) / n * 1e3:7.2f} ms per batch", flush=True)


print("threads", torch.get_num_threads())
timed("make_batch(8, 640, cpu)", lambda i: make_batch(8, 640, "cpu", seed=i))
b = make_batch(8, 640, "cpu", seed=0)
timed("pin_batch", lambda i: pin_batch(b))
pb = pin_batch(b)
timed("H2D of a pinned batch (non_blocking)", lambda i: [v.value.to(dev, non_blocking=True) for v in pb.values()])
timed("BackgroundLoader alone (pin)", lambda i: None, n=1)
t = time.perf_counter()
for _ in BackgroundLoader(lambda i: make_batch(8, 640, "cpu", seed=i), N):
    pass
print(f"{'BackgroundLoader(pin=True) drained':50s} {(time.perf_counter() - t) / N * 1e3:7.2f} ms per batch")
t = time.perf_counter()
for _ in DevicePrefetcher(BackgroundLoader(lambda i: make_batch(8, 640, "cpu", seed=i), N), dev):
    pass
torch.cuda.synchronize()
print(f"{'DevicePrefetcher(BackgroundLoader) drained':50s} {(time.perf_counter() - t) / N * 1e3:7.2f} ms per batch")
