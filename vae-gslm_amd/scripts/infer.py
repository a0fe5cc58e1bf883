"""``python -m scripts.infer -c configs/infer/speech/vae-gslm.yaml`` (reference scripts/infer.py).

The reference hands the inferer to a Lightning ``Trainer.test`` over its dataset; this build drives
:class:`inference.speech.inferer.SpeechInferer` directly.  ``--synthetic`` decodes random prompts with a
randomly initialised model of ``--train-config`` (no checkpoint / dataset needed): the path used to
measure the decode step on the MI355X."""
import argparse
import json
import logging
import os
import time
from pathlib import Path

import torch

from hparams.hp import Hparams

parser = argparse.ArgumentParser(prog="Infer a model with a given config")
parser.add_argument("-c", "--config", type=str, required=True)
parser.add_argument("-v", "--version", type=str, default=None)
parser.add_argument("-log", "--loglevel", type=str, default="WARNING", choices=logging._nameToLevel.keys())
parser.add_argument("--synthetic", action="store_true", help="random prompts, randomly initialised model")
parser.add_argument("--train-config", type=str, default=os.path.join("configs", "train", "speech", "vae-gslm.yaml"))
parser.add_argument("--batch", type=int, default=8)
parser.add_argument("--sampling-timesteps", type=int, default=None, help="override diffusion.sampling_timesteps")


def main():
    args = parser.parse_args()
    logging.basicConfig(level=args.loglevel.upper())
    hp = Hparams.from_yamlfile(args.config)
    if hp.has("output_dir"):
        Path(hp.output_dir).mkdir(parents=True, exist_ok=True)
    if args.version is not None:
        hp.check_arg_in_hparams("exp_dir")
        hp.ckpt_path = os.path.join(hp.exp_dir, "ckpt", f"version_{args.version}")
    if args.sampling_timesteps is not None:
        hp.diffusion.sampling_timesteps = args.sampling_timesteps
    from inference.speech.inferer import FRAME_RATE, SpeechInferer
    if not torch.cuda.is_available():
        raise SystemExit("scripts.infer needs an MI355X: the HIP decode path has no CPU fallback")
    hp_model = Hparams.from_yamlfile(args.train_config) if args.synthetic else None
    inferer = SpeechInferer(hp, hp_model=hp_model)
    if not args.synthetic:
        raise SystemExit("dataset readers are outside this build (SURVEY.md 8): call SpeechInferer.test_step with "
                         "{tokens, mel} batches, or use --synthetic")
    from training_lib.synthetic import make_batch
    T = int(hp.sample_prior_length * FRAME_RATE)
    batch = make_batch(args.batch, T, "cuda:0", seed=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = inferer.test_step(batch)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    frames = out["frames"].shape[1] - T
    torch.save({"mel": out["output"].cpu(), "frames": out["frames"].cpu()}, os.path.join(hp.output_dir, "synthetic.pt"))
    print(json.dumps({"sequences": args.batch, "prompt_frames": T, "generated_frames": frames,
                      "seconds_total": dt, "mel_shape": list(out["output"].shape)}))


if __name__ == "__main__":
    main()
