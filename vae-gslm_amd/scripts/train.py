"""Training entry point:  python -m scripts.train -c configs/train/speech/vae-gslm.yaml

Same flags as the reference CLI (scripts/train.py:14-24): -c config, -n name,
-p profile (2000 steps, per-phase timers), -s sanity, -d detect_anomaly,
-r resume checkpoint, -v version, -log level.  Instead of a Lightning
``Trainer`` it runs one process per GPU: launch with
``python -m torch.distributed.run --nproc-per-node N -m scripts.train -c ...``
for data parallelism over RCCL/xGMI (single process otherwise).

No dataset ships with the repository; when the configured metadata file is
missing (or ``--synthetic`` is given) synthetic 50 Hz token + mel batches of
the configured crop length are used, so the full step can be exercised.
"""
from __future__ import annotations

import argparse
import importlib
import logging
import os
import time

import torch
import torch.distributed as dist

from hparams.hp import Hparams
from training_lib.synthetic import make_batch


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("-c", "--config", type=str, required=True)
    ap.add_argument("-n", "--name", type=str, default=None)
    ap.add_argument("-p", "--profile", action="store_true")
    ap.add_argument("-s", "--sanity", action="store_true")
    ap.add_argument("-d", "--detect_anomaly", action="store_true")
    ap.add_argument("-r", "--resume", type=str, default=None)
    ap.add_argument("-v", "--version", type=str, default=None)
    ap.add_argument("-log", "--loglevel", type=str, default="info")
    ap.add_argument("--synthetic", action="store_true", help="force synthetic batches")
    ap.add_argument("--device_batches", action="store_true",
                    help="create the synthetic batches on the GPU instead of staging them through pinned host memory")
    ap.add_argument("--max_steps", type=int, default=None, help="optimizer steps to run")
    return ap.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    logging.basicConfig(level=getattr(logging, args.loglevel.upper()))
    log = logging.getLogger("train")
    hp = Hparams.from_yamlfile(args.config)
    hp.check_arg_in_hparams("trainer", "logging", "training", "data")
    hp.trainer.check_arg_in_hparams("identifier", "total_steps")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise RuntimeError("scripts.train needs an MI355X GPU: the HIP hot path has no CPU fallback")
    # functional dry run of the multi-rank path on a one-GPU box: VG_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and
    # exchanges gradients over gloo (same switch as bench.py; real runs use one GPU per rank and RCCL)
    one_dev = os.environ.get("VG_BENCH_ONE_DEVICE") == "1"
    dev_index = 0 if one_dev else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_dev:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    rank = dist.get_rank() if world > 1 else 0

    module_name, cls_name = hp.trainer.identifier.rsplit(".", 1)
    trainer = getattr(importlib.import_module(module_name), cls_name)(hp).to(device)
    resume_ck = None
    if args.resume:          # weights first, so that rank 0's broadcast below starts from them; the file is read once
        resume_ck = torch.load(args.resume, map_location=device)
        trainer.model.load_state_dict(resume_ck["state_dict"] if isinstance(resume_ck, dict) and "state_dict" in resume_ck
                                      else resume_ck)
    if world > 1:   # identical replicas: broadcast rank 0's initial weights
        for p in trainer.model.parameters():
            dist.broadcast(p.data, 0)
    trainer.configure_optimizers()
    trainer.attach_reducer()
    if args.resume and trainer.load_checkpoint(args.resume, map_location=device, ckpt=resume_ck):
        # full checkpoint: optimizer moments / step, LR schedule and global_step (hence the KL warm-up) continue
        log.info("resumed %s at optimizer step %d", args.resume, trainer.global_step)
    resume_ck = None
    torch.autograd.set_detect_anomaly(args.detect_anomaly)

    data_hp = hp.data.train
    synthetic = args.synthetic or not os.path.exists(data_hp.get("path", ""))
    if not synthetic:
        raise NotImplementedError("the on-disk token/mel dataset pipeline (reference data/*.py) is outside "
                                  "the hot path of this build; run with --synthetic")
    T = data_hp.get("token_segment_size", 640)
    B = data_hp.batch_size
    accum = trainer.gradient_update_step
    total = args.max_steps or (2000 if args.profile else hp.trainer.total_steps)
    outdir = os.path.join(hp.logging.log_dir, args.name or "default", args.version or "version_0")
    if rank == 0:
        os.makedirs(outdir, exist_ok=True)
        trainer.save_hparams(outdir)
    # Checkpoints: the reference saves a full and a compact checkpoint every `save_every_n_epoch` epochs and keeps
    # the newest `save_top_k` (scripts/train.py:59-75).  Synthetic data has no epochs: `steps_per_epoch` optimizer
    # steps (default 1000) stand in for one.
    every = int(hp.trainer.get("save_every_n_epoch", 1)) * int(hp.trainer.get("steps_per_epoch", 1000))
    keep = int(hp.trainer.get("save_top_k", 5))
    saved = []
    if args.resume and rank == 0 and os.path.isdir(outdir):
        # a resumed run keeps pruning the earlier run's checkpoints (Lightning's ModelCheckpoint restores its
        # top-k list from the checkpoint): seed the list from the files already there, oldest step first
        import re
        found = {}
        for name in os.listdir(outdir):
            m = re.fullmatch(r"(epoch=\d+-step=(\d+))(-cpt)?\.ckpt", name)
            if m:
                found[int(m.group(2))] = os.path.join(outdir, m.group(1))
        saved = [found[k] for k in sorted(found)]

    def checkpoint():
        if rank != 0:
            return
        epoch = trainer.global_step // max(1, int(hp.trainer.get("steps_per_epoch", 1000)))
        stem = os.path.join(outdir, f"epoch={epoch}-step={trainer.global_step}")
        trainer.save_full_checkpoint(stem + ".ckpt")
        trainer.save_checkpoint(stem + "-cpt.ckpt")
        saved.append(stem)
        while keep > 0 and len(saved) > keep:
            old = saved.pop(0)
            for suffix in (".ckpt", "-cpt.ckpt"):
                if os.path.exists(old + suffix):
                    os.remove(old + suffix)

    t0, frames = time.time(), 0
    start_step = trainer.global_step
    total = max(0, total - start_step)
    first = start_step * accum            # a resumed run continues the batch stream where it stopped
    if args.device_batches:
        feed = (make_batch(B, T, device, seed=1234 + rank * 1000 + first + it) for it in range(total * accum))
    else:   # host batches on a worker thread -> pinned memory -> asynchronous copies two steps ahead
        from training_lib.prefetch import BackgroundLoader, DevicePrefetcher
        if getattr(trainer, "use_graph", False):
            trainer.enter_compute_stream(device)     # the copies synchronise with the stream the step runs on
        feed = DevicePrefetcher(BackgroundLoader(lambda it: make_batch(B, T, "cpu", seed=1234 + rank * 1000 + first + it),
                                                 total * accum), device)
    for it, batch in enumerate(feed):
        before = trainer.global_step
        out = trainer.training_step(batch, first + it)
        frames += B * T * world
        if trainer.global_step != before and trainer.global_step % every == 0:
            checkpoint()
        if rank == 0 and (it + 1) % (50 * accum) == 0:
            torch.cuda.synchronize()
            dt = time.time() - t0
            log.info("step %d  loss %.4f  kld/frame %.4f  tokens/s %.0f", trainer.global_step,
                     float(out["loss"]), float(trainer.logged.get("train/kld", 0.0)), frames / dt)
    if not saved or not saved[-1].endswith(f"step={trainer.global_step}"):
        checkpoint()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
