"""Trainer base without Lightning (reference training_lib/trainer.py:11-125).

The reference subclasses ``pl.LightningModule`` and lets Lightning own the
loop, device placement, DDP and checkpoint I/O.  This build keeps the pieces
the training step actually needs -- the gradient-accumulation counter, weight
initialisation, optimizer construction, a ``log`` sink with the reference's
scalar names -- and drives them from ``scripts/train.py`` with one process per
GPU and the RCCL gradient reducer of ``training_lib/dp.py``.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn as nn

import hipvg
from hparams.hp import Hparams


class BaseTrainer(nn.Module):
    def __init__(self, hp: Hparams) -> None:
        super().__init__()
        hp.check_arg_in_hparams("training")
        self.hp = hp
        self.gradient_update_step = hp.training.get("gradient_accumulation", 1)
        self.global_step = 0                 # optimizer steps taken (Lightning semantics)
        self.logged: Dict[str, float] = {}
        hip = hp.get("hip", None)
        hipvg.set_precision(hip.get("precision", "bf16") if hip is not None else "bf16")

    def log(self, name: str, value, **_unused) -> None:
        self.logged[name] = value

    def init_weights(self, module) -> None:
        """Zero every bias, reset LayerNorm/GroupNorm affine, then let modules
        that define ``custom_weight_init`` override (reference :113-125)."""
        init_std = self.hp.training.get("init_std", 1.0)
        bias = getattr(module, "bias", None)
        if isinstance(bias, torch.Tensor):
            with torch.no_grad():
                bias.zero_()
        elif isinstance(module, (nn.LayerNorm, nn.GroupNorm)) and module.weight is not None:
            with torch.no_grad():
                module.weight.fill_(1.0)
        hook = getattr(module, "custom_weight_init", None)
        if callable(hook):
            hook(init_std)
