#!/usr/bin/env python3
"""Benchmark of the VAE-GSLM training step on MI355X (driver contract: see the
task statement).  One "step" = one optimizer step of configs/train/speech/
vae-gslm.yaml = `gradient_accumulation` (2) micro-batches of `batch_size` (8)
sequences, each a full LVTR forward + backward (Transformer+VAE hot path on the
HIP kernels, conv encoder / diffusion decoder on stock ops), the RCCL gradient
all-reduce (N > 1) and the fused AdamW update.  seq_len = 1000 frames (BASELINE
configs[1]); synthetic 50 Hz token + mel batches resident in HBM before timing.

    python bench.py [--gpus N --steps K --warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Rank 0 prints ONE JSON line: tokens/s (whole job), plus
  roofline     -- the bf16 MFMA GEMM kernel family (dominant kernel): algorithmic
                  FLOPs / launch durations measured with HIP events on the launch
                  stream during the timed region, against the 2.5 PFLOP/s dense peak;
  cpu_baseline -- the CPU oracle (port of the reference's fp32 path) timed on this
                  host's cores on a bounded sample of the same workload (N = 1 only).
"""
from __future__ import annotations

import argparse
import copy
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "vae-gslm_amd"))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

CONFIG = os.path.join(ROOT, "vae-gslm_amd", "configs", "train", "speech", "vae-gslm.yaml")
SEQ_LEN = 1000
TRAIN_FLOP_PER_TOKEN = 1.3238e9      # BASELINE.md section 3 (hot path, causal-exact, 3 x forward)
PEAK_BF16 = 2.5e15                   # dense bf16 MFMA peak, MI355X_MICROARCH.md


def cpu_model_name() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(threads_note=True):
    """Oracle (CPU port of the reference fp32 path): full config, B=2, T=1000,
    forward+backward after a short warm-up at T=64.  `cores` reports the intra-op threads the sample used (at most 32;
    `host_cores` is the box's core count): on these shapes more threads only add synchronisation cost -- a 256-thread
    pool made the same sample take minutes instead of seconds."""
    import yaml
    from oracle import lvtr_oracle as O
    from oracle.weights import fill_like
    with open(CONFIG) as f:
        cfg = yaml.safe_load(f)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    mcfg = cfg["model"]
    sd = {k: torch.from_numpy(v).requires_grad_(True) for k, v in fill_like(O.param_shapes(mcfg), 1).items()}

    B = 2

    def one(T, seed):
        g = torch.Generator().manual_seed(seed)
        batch = dict(tokens=torch.randint(0, 200, (B, T), generator=g), mel=torch.randn(B, T, 80, generator=g),
                     lengths=torch.tensor([T] * B), utt=torch.randn(B, 150, 80, generator=g),
                     utt_lengths=torch.tensor([150] * B))
        noise = dict(eps_q=torch.randn(B, T, 4, generator=g), init_state=torch.rand(B, 1, 64, generator=g) * 2 - 1,
                     eps_p=torch.zeros(B, T, 4), t_diff=torch.randint(0, 1000, (B,), generator=g),
                     eps_diff=torch.randn(B, T, 80, generator=g))
        out = O.training_loss(sd, mcfg, cfg["training"], batch, noise)
        out["loss"].backward()
        for v in sd.values():
            v.grad = None
    one(64, 0)
    # bounded sample: a quarter-length sequence first; the full 1000-frame step only
    # if it is projected to stay within ~40 s of host time
    T = SEQ_LEN // 4
    t0 = time.perf_counter()
    one(T, 1)
    dt = time.perf_counter() - t0
    steps = 1
    if dt * 5 < 40.0:
        T = SEQ_LEN
        t0 = time.perf_counter()
        one(T, 2)
        dt = time.perf_counter() - t0
        # aim at 10-30 s of timed CPU work: repeat the full-length step while the projection stays under ~25 s
        more = max(0, min(6, int(20.0 / max(dt, 1e-3)) - 1))
        for i in range(more):
            one(T, 3 + i)
        steps += more
        dt = time.perf_counter() - t0
    # `cores` = the threads the sample actually ran on (the contract's meaning); the host's core count beside it
    return {"value": steps * B * T / dt, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "cpu_model": cpu_model_name(), "host_cores": os.cpu_count() or 1,
            "sample": f"oracle fp32 fwd+bwd, full config, B={B}, T={T}, {steps} step(s) ({dt:.1f} s) after a warm-up"}


def decode_main(args):
    """``--mode decode``: BASELINE config 4 (3 s prompt -> 10 s continuation, hipGraph-captured decode step, 1 GPU).
    A "step" is one generated frame for every sequence of the batch; the JSON line keeps the training line's shape
    with an HBM roofline: the step streams the bf16 weights of the stack and heads once plus the live key/value
    cache, so achieved = (weight bytes + mean cache bytes per frame) / measured time per frame."""
    import hipvg
    from hparams.hp import Hparams
    from inference.speech.session import DecodeSession
    from models.speech.lvtr import LVTR
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP hot path has no CPU fallback")
    if args.gpus != 1:
        raise SystemExit("--mode decode is a single-GPU measurement (replicas only: sequences are independent)")
    hipvg.lib()
    hipvg.set_precision("bf16")
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    model = LVTR(Hparams.from_yamlfile(CONFIG).model, input_dim=80).to(dev).eval()
    B, Tp = args.decode_batch, args.prompt_frames
    frames, warm = max(args.steps, 8), max(args.warmup, 2)
    stack = model.transformer[0]
    n_params = sum(p.numel() for p in stack.parameters()) + sum(
        p.numel() for m in (model.q_spliter, model.token_spliter, model.token_predictor, model.transformer[1])
        for p in m.parameters()) + sum(l.film.linear.weight.numel() for l in model.transformer_flow.layers)
    wbytes = 2.0 * n_params
    prior = torch.cat([torch.randint(0, 200, (B, Tp, 1), device=dev).float(), torch.randn(B, Tp, 4, device=dev)], -1)
    sess = DecodeSession(model, B, Tp + frames + warm + 8, temperature=0.85, token_temperature=0.85, use_graph=True)
    sess.prefill(prior)
    for _ in range(warm):                # eager first frame, capture, first replay
        sess.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    sess.generate(frames)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / frames
    L, D = len(stack.layers), stack.hp.layer.dim
    mean_ctx = Tp + 1 + warm + frames / 2.0
    kv_bytes = 2.0 * L * B * mean_ctx * D * 2            # K and V rows of every layer, bf16
    peaks = hipvg.probe_peaks(dev)
    achieved = (wbytes + kv_bytes) / dt / 1e9
    line = {"metric": "decode frames/sec (50 Hz frames), 3 s prompt -> continuation, hipGraph-captured step",
            "value": B / dt, "unit": "frames/s", "n_gpus": 1, "steps": frames, "warmup": warm, "ms_per_step": 1e3 * dt,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"vae-gslm.yaml full config autoregressive decode (LVTR.step path), batch {B}, "
                                   f"prompt {Tp} frames, {frames} generated frames", "batch": B, "prompt_frames": Tp},
            "roofline": {"bound": "hbm", "kernel": "decode step (gemm_rows weight streaming + cache attention)",
                         "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "frac_of_measured_copy": achieved / (1e3 * peaks["hbm_copy_tb_per_s"]),
                         "hbm_copy_measured_tb_per_s": peaks["hbm_copy_tb_per_s"],
                         "weight_bytes_per_frame": wbytes, "kv_cache_bytes_per_frame": kv_bytes, "traffic": None}}
    print(json.dumps(line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", default="train", choices=("train", "decode"),
                    help="train: the headline training step (default); decode: BASELINE config 4, the AR decode step")
    ap.add_argument("--decode-batch", type=int, default=8)
    ap.add_argument("--prompt-frames", type=int, default=150)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--single-rank-rccl", action="store_true",
                    help="N = 1 only: run the whole data-parallel step (segmented graphs, bucket all-reduces on the "
                         "communication stream, per-bucket optimizer launches) on a ONE-rank RCCL communicator -- "
                         "the machinery's cost without the wire, and a functional check of the RCCL path on one GPU")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--comm", default="torch", choices=("torch", "abi"),
                    help="gradient all-reduce through torch.distributed (nccl = RCCL) or through vg_allreduce_bucket")
    ap.add_argument("--graph", type=int, default=1, help="replay each micro-step as a hipGraph (1) or launch eagerly (0)")
    ap.add_argument("--graph-bucket-mb", type=float, default=None,
                    help="N > 1, hipGraph mode: size of the gradient messages (hip.graph_bucket_mb of the yaml, 256)")
    ap.add_argument("--graph-cut-layers", default=None,
                    help="N > 1, hipGraph mode: comma-separated Transformer layers below which the micro-step is cut into "
                         "separately replayed graphs (hip.graph_cut_layer, default 12,8,4); 'none' = one graph")
    ap.add_argument("--bucket-mb", type=float, default=None, help="N > 1, eager mode: gradient bucket size (hip.bucket_mb, 50)")
    ap.add_argument("--seq-len", type=int, default=SEQ_LEN, help="frames per sequence (BASELINE config 5: 2000)")
    ap.add_argument("--coalesce", type=int, default=1,
                    help="1 (default, = hip.coalesce_accumulation of the yaml): the micro-batches of an accumulation "
                         "window (8 sequences x 2) run as ONE launch sequence over 16 sequences -- same samples, same "
                         "summed loss, same gradient and optimizer step, taller GEMMs; 0: one pass per micro-batch")
    ap.add_argument("--host-batches", action="store_true",
                    help="hand the step HOST batches: pinned memory -> asynchronous copies two steps ahead "
                         "(training_lib.prefetch); the PCIe-inclusive rate quoted in DESIGN.md")
    ap.add_argument("--packed-rows", type=int, default=None,
                    help="hip.packed_rows of the yaml (1): the Transformer stack runs on the valid frames only "
                         "(row gather + varlen attention); 0 keeps the padded rows")
    ap.add_argument("--packed-granule", type=int, default=None, help="hip.packed_rows_granule (rows per bucket)")
    ap.add_argument("--side-unet", type=int, default=None,
                    help="hip.side_unet of the yaml (0): 1 runs the diffusion decoder beside the Transformer stack on the step's "
                         "side stream (faster step, but every kernel's own duration then includes sharing the chip)")
    ap.add_argument("--packed-step", type=int, default=None,
                    help="hip.packed_step of the yaml (1): the WHOLE step of a ragged batch runs on its valid frames (+ an 18-frame "
                         "halo per sequence for the UNet's look-ahead blocks); 0: only the Transformer stack (hip.packed_rows)")
    ap.add_argument("--ragged-range", type=str, default="0.5,1.0",
                    help="with --ragged: sequence lengths ~ U{lo T .. hi T} (default 0.5,1.0; one sequence of every batch keeps "
                         "the full length so that the padded shape is the same)")
    ap.add_argument("--ragged", action="store_true",
                    help="sequence lengths ~ U{T/2..T} (right-padded batches, SURVEY 8d): shows the cost of masking; "
                         "tokens/s then counts valid frames only")
    args = ap.parse_args()
    if args.mode == "decode":
        if args.steps == 5 and args.warmup == 2:        # the training defaults: a decode run wants more frames
            args.steps, args.warmup = 400, 4
        return decode_main(args)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus or world == 1 and args.gpus == 1, \
        f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP hot path has no CPU fallback")
    # functional dry run of the N > 1 path on a one-GPU box: VG_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and
    # exchanges gradients over gloo (timings are then meaningless; the driver's runs use one GPU per rank and RCCL)
    one_dev = os.environ.get("VG_BENCH_ONE_DEVICE", "0") == "1"
    dev_index = 0 if one_dev else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dp = world > 1 or args.single_rank_rccl        # collectives run
    if args.single_rank_rccl:
        assert world == 1 and args.gpus == 1, "--single-rank-rccl is the one-GPU exercise of the N > 1 path"
        os.environ["VG_DP_SINGLE_RANK"] = "1"
        if not dist.is_initialized():
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    rank = dist.get_rank() if world > 1 else 0

    import hipvg
    import hipvg.functional
    from hparams.hp import Hparams
    from trainers.speech.lvtr import LVTRTrainer
    from training_lib.synthetic import make_batch

    hipvg.lib()
    hp = Hparams.from_yamlfile(CONFIG)
    hp.hip.precision = args.precision
    hp.hip.graph = bool(args.graph)
    if args.packed_rows is not None:
        hp.hip.packed_rows = bool(args.packed_rows)
    if args.packed_granule is not None:
        hp.hip.packed_rows_granule = args.packed_granule
    if args.packed_step is not None:
        hp.hip.packed_step = bool(args.packed_step)
    if args.side_unet is not None:
        hp.hip.side_unet = bool(args.side_unet)
    hp.hip.coalesce_accumulation = bool(args.coalesce)
    hp.hip.comm = args.comm
    if args.graph_bucket_mb is not None:
        hp.hip.graph_bucket_mb = args.graph_bucket_mb
    if args.bucket_mb is not None:
        hp.hip.bucket_mb = args.bucket_mb
    if args.graph_cut_layers is not None:
        hp.hip.graph_cut_layer = [] if args.graph_cut_layers.lower() == "none" else \
            [int(v) for v in args.graph_cut_layers.split(",") if v]
    torch.manual_seed(1234)
    trainer = LVTRTrainer(hp).to(device)
    if world > 1:
        for p in trainer.model.parameters():
            dist.broadcast(p.data, 0)
    trainer.configure_optimizers()
    reducer = trainer.attach_reducer()
    ranks = reducer.communicator_ranks()
    if ranks != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the gradient communicator has {ranks} rank(s) "
                         f"(comm={args.comm}, backend={dist.get_backend() if dist.is_initialized() else 'none'})")
    reducer.time_collectives = dp
    trainer.global_step = hp.training.scheduler.warmup_kld      # past the KL warm-up
    # event pairs captured into the graph do not report on replay (and cost graph nodes): off unless asked for
    trainer.profile_in_graph = bool(args.graph) and os.environ.get("VG_PROF_IN_GRAPH", "0") == "1"
    B = hp.data.train.batch_size
    accum = trainer.gradient_update_step
    n_micro = (args.steps + args.warmup) * accum
    T_SEQ = args.seq_len
    def lens_for(i):
        if not args.ragged:
            return None
        g = torch.Generator().manual_seed(99 + rank * 1000 + i)
        lo, hi = (float(v) for v in args.ragged_range.split(","))
        ls = torch.randint(max(1, int(lo * T_SEQ)), max(2, int(hi * T_SEQ)) + 1, (B,), generator=g)
        ls[0] = T_SEQ                      # the batch keeps its padded length
        return ls.tolist()
    all_lens = [lens_for(i) for i in range(n_micro)]
    batches = [make_batch(B, T_SEQ, device, seed=1234 + rank * 1000 + i, lengths=all_lens[i]) for i in range(n_micro)]
    torch.cuda.synchronize()

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.host_batches:
        from training_lib.prefetch import DevicePrefetcher, pin_batch
        if args.graph:
            trainer.enter_compute_stream(device)
        host = [pin_batch(make_batch(B, T_SEQ, "cpu", seed=1234 + rank * 1000 + i, lengths=all_lens[i]))
                for i in range(n_micro)]
        feed = DevicePrefetcher(host, device)
        fetch = lambda i: next(feed)
    else:
        fetch = lambda i: batches[i]
    it = 0
    for _ in range(args.warmup * accum):
        trainer.training_step(fetch(it), it)
        it += 1
    if args.ragged and args.graph and not args.host_batches:
        # ragged batches fall into several row buckets of hip.packed_rows: every (padded length, bucket) shape of the
        # timed batches is captured here, untimed, so that the timed region measures replays and not captures
        # (a training run meets each shape once in its first minutes)
        lo = it
        for j in range(args.steps * accum):
            trainer.training_step(batches[lo + j], lo + j)
    sync()
    if dp:
        reducer.comm_stats()               # drop the warm-up's collectives from the record
    if not args.graph:
        hipvg.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps * accum):
        out = trainer.training_step(fetch(it), it)
        it += 1
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    comm = reducer.comm_stats() if dp else None
    comm_exposed_ms = None
    if dp:
        # the same steps without their collectives: the difference is the part of the exchange nothing hides
        reducer.time_collectives, reducer.stub_collectives = False, True
        n_stub = min(args.steps, 3)
        sync()
        t1 = time.perf_counter()
        for j in range(n_stub * accum):
            trainer.training_step(fetch(it - accum * n_stub + j) if not args.host_batches else batches[it - accum * n_stub + j],
                                  it - accum * n_stub + j)
        sync()
        stub = torch.tensor([(time.perf_counter() - t1) / n_stub], device=device, dtype=torch.float64)
        dist.all_reduce(stub, op=dist.ReduceOp.MAX)
        reducer.stub_collectives = False
        comm_exposed_ms = 1e3 * (elapsed / args.steps - float(stub.item()))
    tokens = args.steps * accum * B * T_SEQ * world
    if args.ragged:     # valid frames of the timed micro-batches (this rank's draw, times the ranks)
        tokens = world * sum(sum(l) for l in all_lens[args.warmup * accum:])
    value = tokens / elapsed
    if args.graph:
        # Kernels inside a replayed hipGraph cannot be bracketed by host-recorded events: the per-kernel durations are
        # measured on ONE extra optimizer step of the same workload launched eagerly -- BEHIND queued graph replays, so
        # that the host has enqueued the step before the GPU reaches it and its kernels run back to back, as they do in
        # the timed replays.  (Launched into an idle queue, as rounds 1-3 did, every kernel starts on a GPU that has been
        # waiting for the host: rocprofv3 shows the same kernels 8-9 % longer there than in the replayed graph --
        # grouped weight gradients 1,202 against 1,100 us -- and the event pairs add the dispatch latency on top.  Behind
        # the replays the step's GPU span is 31.6-32.1 ms against 29.6 ms replayed, with 361 event pairs in it: the
        # durations reported here still carry a few microseconds of marker time per launch, i.e. they err on the slow side.)
        # A first eager step behind its own replays takes the allocator's and the lazy initialisations' host time.
        def eager_step_behind_replays(profiled):
            trainer.use_graph = True
            for r in range(3):
                for j in range(accum):
                    trainer.training_step(batches[it - accum + j], it - accum + j)
            trainer.use_graph = False
            hipvg.prof_enable(profiled)
            for j in range(accum):
                trainer.training_step(batches[it - accum + j], it - accum + j)
        eager_step_behind_replays(False)
        eager_step_behind_replays(True)
        sync()

    if rank == 0:
        kinds = {}
        tot_ms = tot_work = tot_bytes = 0.0
        tot_n = 0
        for k in ("gemm_bf16_nt", "gemm_bf16_nn", "gemm_bf16_tn", "gemm_f32", "attn_fwd", "attn_bwd"):
            ms, work, n = hipvg.prof_read(k)
            if n:
                kinds[k] = {"launches": n, "avg_us": 1e3 * ms / n, "tflops": work / (ms * 1e-3) / 1e12}
            if k.startswith("gemm_bf16"):
                tot_ms += ms
                tot_work += work
                tot_bytes += hipvg.prof_read_bytes(k)
                tot_n += n
        # HBM-bound row kernels: algorithmic bytes / launch time, against the copy rate measured below
        hbm_kernels = {}
        for k in ("rmsnorm_fwd", "rmsnorm_bwd", "adamw", "dwnorm_fwd", "dwnorm_bwd"):
            ms, nbytes, n = hipvg.prof_read(k)
            if n:
                hbm_kernels[k] = {"launches": n, "avg_us": 1e3 * ms / n, "gb_per_s": nbytes / (ms * 1e-3) / 1e9}
        achieved = tot_work / (tot_ms * 1e-3) / 1e12 if tot_ms else 0.0
        attn_hbm = {}
        rows_bytes = float(B * accum if args.coalesce else B) * T_SEQ * 1024 * 2        # one [rows, d_model] bf16 stream
        for k, streams in (("attn_fwd", 4.0), ("attn_bwd", 8.0)):
            ms, _, n = hipvg.prof_read(k)
            if n and not args.ragged:
                nbytes = streams * rows_bytes
                attn_hbm[k] = {"compulsory_mb": nbytes / 1e6, "avg_us": 1e3 * ms / n, "gb_per_s": nbytes / (ms / n * 1e-3) / 1e9,
                               "frac_of_8tb": nbytes / (ms / n * 1e-3) / 8e12}
        # attention kernels: causal-exact FLOP, the backward counted as 2 x forward (no recompute credit)
        a_ms = a_work = 0.0
        for k in ("attn_fwd", "attn_bwd"):
            ms, work, n = hipvg.prof_read(k)
            a_ms += ms
            a_work += work
        path_tflops = (tot_work + a_work) / ((tot_ms + a_ms) * 1e-3) / 1e12 if tot_ms + a_ms else 0.0
        # the north_star's gate, "attention + FFN path": the Transformer layers' own GEMMs (QKV, out-projection, FFN x
        # {forward, dgrad, wgrad}: every launch recorded inside TransformerLayerFn) + the attention kernels, without
        # the conv stacks, heads and input projection that the family figure above also holds
        l_ms = l_work = 0.0
        for k in ("gemm_bf16_nt", "gemm_bf16_nn", "gemm_bf16_tn"):
            ms, work, n = hipvg.prof_read_tag(k, hipvg.PROF_TAG_LAYER)
            l_ms += ms
            l_work += work
        layer_tflops = (l_work + a_work) / ((l_ms + a_ms) * 1e-3) / 1e12 if l_ms + a_ms else 0.0
        hipvg.prof_enable(False)
        peaks = hipvg.probe_peaks(device)          # this box, this run: register-fed MFMA chains and a 1 GiB copy
        # HBM-side bytes per launch of the same kernel family: bench.py cannot collect PMC counters itself, so
        # this is the figure of the committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (corrected as
        # MI355X_MICROARCH.md prescribes; profiles/r01/pmc_traffic_v9.json), valid for the default workload only
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, "profiles", "r06", "pmc_traffic.json")
        if not os.path.exists(pmc):
            pmc = os.path.join(ROOT, "profiles", "r05", "pmc_traffic.json")
        if os.path.exists(pmc) and T_SEQ == SEQ_LEN and args.precision == "bf16" and args.coalesce and not args.ragged:
            with open(pmc) as f:
                traffic = json.load(f)["bf16_gemm_family"]["traffic_bytes_per_launch"]
            traffic_src = (os.path.relpath(pmc, ROOT) + ": STATIC figure from committed rocprofv3 --pmc FETCH_SIZE / "
                           "WRITE_SIZE passes over this command (bench.py cannot read PMC counters itself); it "
                           "describes the build that profile was taken on")
        # 25,165,824 GEMM + 2*2*1024*(T+1)/2 attention FLOP per layer, 16 layers, + heads; x3
        flop_per_token = 3.0 * (16 * (25165824 + 2 * 2 * 1024 * (T_SEQ + 1) / 2) + 131072 + 5671936)
        line = {
            "metric": f"train tokens/sec (50 Hz frames) at seq_len={T_SEQ}",
            "value": value, "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16" if args.precision == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": "vae-gslm.yaml full config (L=16, d=1024, H=16, ffd=4096, 227M params), "
                                   f"fwd+bwd+AdamW, micro-batch {B} x grad-accum {accum} per step, seq_len {T_SEQ}"
                                   + (f" (the {accum} micro-batches run as one pass over {B * accum} sequences)"
                                      if args.coalesce and accum > 1 else ""),
                       "micro_batch": B, "grad_accum": accum, "seq_len": T_SEQ,
                       "parallelism": f"dp{world}", "loss": float(out["loss"]),
                       "lengths": ("U{T/2..T}, valid frames counted; the timed batches were stepped once before the timed "
                                   "region (untimed) so that every (padded length, row bucket) hipGraph exists: "
                                   "the timed steps replay, on weights those extra steps have already updated")
                                  if args.ragged else "full",
                       "inputs": ("pinned host batches, asynchronous H2D two steps ahead (PCIe inside the timed region)"
                                  if args.host_batches else "resident in HBM before the timed region"),
                       "accumulation": "one launch sequence over B x accum sequences" if args.coalesce else "per micro-batch",
                       # kind of the stream the hipGraphs are launched from (VG_LAUNCH_STREAM)
                       "launch_stream": hipvg.functional.launch_stream_kind() if args.graph else "none"},
            "roofline": {"bound": "mfma", "kernel": "gemm_kernel<bf16> (NT fwd, NN dgrad, TN wgrad)",
                         "achieved": achieved, "peak": PEAK_BF16 / 1e12, "unit": "TFLOP/s",
                         "frac": achieved / (PEAK_BF16 / 1e12), "traffic": traffic, "traffic_static": traffic is not None,
                         "traffic_source": traffic_src,
                         # operands and results once each, summed over the launches the figure above averages
                         "algorithmic_bytes_per_launch": tot_bytes / tot_n if tot_n else None,
                         # measured in this run (hipvg.probe_peaks: register-fed v_mfma_f32_32x32x16_bf16 chains)
                         "peak_measured": peaks["mfma_bf16_dense_tflops"],
                         "frac_of_measured": achieved / peaks["mfma_bf16_dense_tflops"],
                         "hbm_copy_measured_tb_per_s": peaks["hbm_copy_tb_per_s"], "hbm_peak_tb_per_s": 8.0,
                         # the north_star's gate: bf16 GEMMs and attention kernels together (causal-exact FLOP)
                         "attn_gemm_path_tflops": path_tflops, "attn_gemm_path_frac": path_tflops / (PEAK_BF16 / 1e12),
                         # ... and the gate as the north_star words it: Transformer-layer GEMMs + attention only
                         "attn_ffn_path_tflops": layer_tflops, "attn_ffn_path_frac": layer_tflops / (PEAK_BF16 / 1e12),
                         "attn_ffn_path_ms": l_ms + a_ms,
                         "hbm_kernels": hbm_kernels,
                         "measured_on": ("one optimizer step launched eagerly (HIP event pair around every launch) behind "
                                         "queued hipGraph replays right after the timed region: its kernels run back to "
                                         "back, as in the replays" if args.graph else "the timed region"),
                         # round 6: the attention kernels against the bound that actually sits closest at this shape.
                         # Head dimension 64 at T = 1000: 250 causal-exact FLOP per compulsory byte (forward: q, k, v in, o
                         # out; backward: q, k, v, o, do in, dq, dk, dv out, each once), under the chip's ridge of 312 (2.5
                         # PFLOP/s / 8 TB/s) -- these launches are HBM-side kernels first
                         "attn_hbm": attn_hbm,
                         "step_model_tflops": value / world * flop_per_token / 1e12,
                         "step_model_frac": value / world * flop_per_token / PEAK_BF16,
                         "kernels": kinds},
        }
        if dp:
            nb = len(reducer.buckets)
            line["comm"] = {"rccl_ranks": ranks, "mode": args.comm,
                            "backend": dist.get_backend(), "buckets": nb,
                            "bucket_bytes": [int(b["flat"].numel() * 4) for b in reducer.buckets],
                            "allreduce_ms_per_step": comm["allreduce_ms"] / args.steps,
                            "allreduce_bytes_per_step": comm["allreduce_bytes"] / args.steps,
                            "collectives_per_step": comm["collectives"] / args.steps,
                            "comm_exposed_ms": comm_exposed_ms,
                            "graph_mode": bool(args.graph),
                            "graph_bucket_mb": hp.hip.get("graph_bucket_mb", None),
                            "graph_cut_layers": list(getattr(trainer, "graph_cuts", [])) if args.graph else None,
                            "eager_bucket_mb": hp.hip.get("bucket_mb", None),
                            "note": "allreduce_ms: events on the communication stream (rank 0); comm_exposed_ms: "
                                    "ms_per_step minus the same steps with the collectives skipped"}
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
    # The JSON line is the LAST thing on stdout: RCCL prints a version banner through C stdio, which (block-buffered
    # on a pipe) would otherwise surface after it, when the processes exit.  Every rank empties its C buffers, the
    # ranks meet, the communicator goes, and only then does rank 0 write the line.
    import ctypes
    libc = ctypes.CDLL(None)
    sys.stdout.flush()
    libc.fflush(None)
    if dist.is_initialized():
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()
        libc.fflush(None)
    if rank == 0:
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
