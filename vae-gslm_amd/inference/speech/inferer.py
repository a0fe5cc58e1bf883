"""Prompt -> continuation driver (reference inference/speech/inferer.py:113-178, the ``lvtr`` branch of
``SpeechInferer.test_step``), without the Lightning shell, dataset readers and vocoder: those are outside
the hot path, so the driver consumes batches of ``{tokens, mel}`` TensorMasks (synthetic or
from the caller) and returns mel spectrogram continuations; writing audio needs the HiFi-GAN vocoder of the
reference and is left to the caller."""
from __future__ import annotations

import importlib
import os
from typing import Mapping, Optional

import torch

import hipvg
from hparams.hp import Hparams
from trainers.speech.sampler import ARTRSampler
from utils.tensormask import TensorMask

FRAME_RATE = 50          # frames per second of the 16 kHz / hop 320 mel and of the HuBERT tokens


class SpeechInferer:
    def __init__(self, hp: Hparams, hp_model: Optional[Hparams] = None, device="cuda:0"):
        self.hp = hp
        hip = hp.get("hip", None)
        hipvg.set_precision(hip.get("precision", "bf16") if hip is not None else "bf16")
        self.use_graph = bool(hip.get("graph", True)) if hip is not None else True
        self.group = int(hip.get("sessions_of", 16)) if hip is not None else 16
        ckpt = None
        if hp_model is None:
            hp.check_arg_in_hparams("ckpt_path")
            hp_model = Hparams.from_yamlfile(os.path.join(hp.ckpt_path, "hp.yaml"))
            ckpt = os.path.join(hp.ckpt_path, "last-cpt.ckpt")
        self.hp_model = hp_model
        p, m = hp.model.identifier.rsplit(".", 1)
        cls = getattr(importlib.import_module(p), m, None)
        if cls is None:
            raise ValueError(f"{m} not found in {p}.")
        self.model = cls(hp_model.model, input_dim=80)
        if ckpt is not None:
            self.model.load_state_dict(torch.load(ckpt, map_location="cpu"), strict=False)
        self.model = self.model.to(device).eval()
        if hp.has("diffusion"):
            dec = self.model.decoder
            if hp.diffusion.has("sampling_timesteps"):
                dec.sampling_timesteps = hp.diffusion.sampling_timesteps
            if hp.diffusion.has("ddim_sampling_eta"):
                dec.ddim_sampling_eta = hp.diffusion.ddim_sampling_eta
        self.sampler = ARTRSampler(self.model, use_graph=self.use_graph)
        self.use_tokens = bool(getattr(self.model, "use_tokens", False))
        self.nsteps = 0

    @torch.no_grad()
    def test_step(self, batch: Mapping[str, TensorMask], batch_idx: int = 0) -> Mapping[str, torch.Tensor]:
        """batch: ``mel`` (B, T, 80) and, for the token model, ``tokens`` (B, T).  Returns the sampler's dict:
        ``output`` = mel of prompt + continuation (TensorMask), ``frames`` = (B, Tp + length, 1 + latent)."""
        prompt = int(self.hp.sample_prior_length * FRAME_RATE)
        length = int(self.hp.sample_length * FRAME_RATE * self.model.sample_ratio)
        prior = batch["mel"].value[:, :prompt]
        if self.use_tokens:
            prior = torch.cat([batch["tokens"].value[:, :prompt, None].to(prior.dtype), prior], -1)
        outs = []
        for s in range(0, prior.shape[0], self.group):
            outs.append(self.sampler(length, prior[s: s + self.group], temperature=self.hp.temperature,
                                     token_temperature=self.hp.get("token_temperature", 1.0),
                                     truncated_norm=self.hp.get("truncated_norm", None),
                                     encoder_temperature=self.hp.get("encoder_temperature", 1.0)))
            self.nsteps += 1
        return {"output": torch.cat([o["output"].value for o in outs], 0),
                "frames": torch.cat([o["frames"] for o in outs], 0)}
