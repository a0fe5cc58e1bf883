#!/usr/bin/env python3
"""Lab: reproducer of the hipGraph first-launch crash met in the GPU test suite (rocgdb: hip::Graph::UpdateStreams reads past
the end of the exec's parallel-stream vector).

What the disassembly of libamdhip64 (ROCm 7.0, the one torch 2.10 ships) says: hipGraphInstantiate creates max_streams
internal streams for a graph whose widest level has max_streams branches; the FIRST hipGraphLaunch walks them and hands
branch i the next internal stream whose virtual device differs from the launch stream's -- with no bound on the walk.  One
internal stream like the launch stream is tolerated (one spare), two run off the vector.  Streams share the device's few
hardware queues (least-loaded queue first), so in a long-lived process with an uneven population of streams two internal
streams can land where the launch stream sits.

This script makes that population on purpose: 4 k raw streams, then all of one residue class destroyed (one hardware queue
k lighter than the others), a launch stream created next (lands on the light queue), then a 3-branch graph instantiated
(its internal streams land there too) and launched.  MODE=normal: launch stream of normal priority (expected: SIGSEGV);
MODE=high: launch stream of high priority (other queue pool: expected to run); MODE=mask: launch stream created with
a full CU mask (a queue of its own: expected to run).
    MODE=normal|high|mask [K=8] python tools/lab/hipgraph_queue_collision.py"""
import ctypes
import os
import sys

import torch

mode, K = os.environ.get("MODE", "normal"), int(os.environ.get("K", "8"))
RES = int(os.environ.get("RES", "0"))
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
x = torch.zeros(1 << 20, device=dev)
pool = [torch.cuda.Stream() for _ in range(3)]          # torch's pool exists before the raw streams
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
raw = []
for i in range(4 * K):
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
    raw.append(s)
# touch every stream once (a stream takes its hardware queue at first use)
for s in raw:
    with torch.cuda.stream(torch.cuda.ExternalStream(s.value)):
        x.add_(1)
torch.cuda.synchronize()
for i, s in enumerate(raw):
    if i % 4 == RES:
        assert hip.hipStreamDestroy(s) == 0
launch = ctypes.c_void_p()
if mode == "mask":      # every CU, normal priority, a hardware queue of its own
    words = 8
    full = (ctypes.c_uint32 * words)(*([0xFFFFFFFF] * words))
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(launch), ctypes.c_uint32(words), full) == 0
elif mode == "high":
    lo, hi = ctypes.c_int(), ctypes.c_int()
    hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
    assert hip.hipStreamCreateWithPriority(ctypes.byref(launch), 1, hi.value) == 0
else:
    assert hip.hipStreamCreateWithFlags(ctypes.byref(launch), 1) == 0
L = torch.cuda.ExternalStream(launch.value)
with torch.cuda.stream(L):
    x.add_(1)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
a, b, c = (torch.zeros(1 << 20, device=dev) for _ in range(3))
with torch.cuda.graph(g):
    main = torch.cuda.current_stream()
    for s, t in zip(pool[:2], (a, b)):
        s.wait_stream(main)
        with torch.cuda.stream(s):
            for _ in range(4):
                t.add_(1)
    for _ in range(4):
        c.add_(1)
    for s in pool[:2]:
        main.wait_stream(s)
    c.add_(a).add_(b)
print(f"mode={mode}: graph instantiated, first launch ...", flush=True)
with torch.cuda.stream(L):
    g.replay()
torch.cuda.synchronize()
print(f"mode={mode}: replay ok, c[0] = {float(c[0])}", flush=True)
