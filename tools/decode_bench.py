#!/usr/bin/env python3
"""Decode-step benchmark (BASELINE config 4: 3 s prompt -> 10 s continuation, full vae-gslm.yaml model,
random weights): frames/s and ms per frame for (a) the hipGraph-replayed DecodeSession, (b) the same
session launched eagerly, (c) the reference-style ``model.step`` loop with grown KV tensors.  The step is
bound by streaming the bf16 weights once (403 MB): the implied GB/s is printed next to the HBM peak.
GPU only.   usage: python tools/decode_bench.py [batch=8] [prompt=150] [frames=500]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
import torch

import hipvg
from hparams.hp import Hparams
from inference.speech.session import DecodeSession
from models.speech.lvtr import LVTR

CONFIG = os.path.join(ROOT, "vae-gslm_amd", "configs", "train", "speech", "vae-gslm.yaml")


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    Tp = int(sys.argv[2]) if len(sys.argv) > 2 else 150
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 500
    hipvg.lib()
    hipvg.set_precision("bf16")
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = LVTR(Hparams.from_yamlfile(CONFIG).model, input_dim=80).to(dev).eval()
    stack_params = sum(p.numel() for p in model.transformer[0].parameters())
    head_params = sum(p.numel() for m in (model.q_spliter, model.token_spliter, model.token_predictor, model.transformer[1])
                      for p in m.parameters()) + sum(l.film.linear.weight.numel() for l in model.transformer_flow.layers)
    wbytes = 2 * (stack_params + head_params)
    prior = torch.cat([torch.randint(0, 200, (B, Tp, 1), device=dev).float(), torch.randn(B, Tp, 4, device=dev)], -1)
    res = {"batch": B, "prompt_frames": Tp, "generated_frames": n, "weight_bytes_per_step": wbytes}
    for name, graph in (("hipgraph", True), ("eager", False)):
        sess = DecodeSession(model, B, Tp + n + 8, temperature=0.85, token_temperature=0.85, use_graph=graph)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sess.prefill(prior)
        torch.cuda.synchronize()
        t_pre = time.perf_counter() - t0
        sess.step(); sess.step()                      # eager first frame + capture, first replay
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sess.generate(n - 2)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / (n - 2)
        res[name] = {"prefill_ms": 1e3 * t_pre, "ms_per_frame": 1e3 * dt, "frames_per_s": B / dt,
                     "weight_stream_GBps": wbytes / dt / 1e9}
    # reference-style loop (grown KV tensors, Python-launched): shorter run, it is slow
    m = min(n, 60)
    with torch.no_grad():
        out = model.step(prior, push_init_state=True, temperature=0.85, token_temperature=0.85)
        it = {"output": out["output"][:, -1:], "kv": out["kv"]}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(m):
            it = model.step(it["output"], past_kv=it["kv"], temperature=0.85, token_temperature=0.85)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / m
    res["step_loop"] = {"ms_per_frame": 1e3 * dt, "frames_per_s": B / dt}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
