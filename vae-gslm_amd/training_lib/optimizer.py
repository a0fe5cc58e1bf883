"""Optimizer / LR-schedule factory (reference training_lib/optimizer.py:8-130).

Same yaml keys and the same resulting schedule (optional linear warm-up, flat
phase, then cosine / linear / constant, optional constant tail) built from
stock ``torch.optim`` pieces; on the GPU the fused (single-kernel, multi-tensor)
Adam/AdamW implementation is used.
"""
from __future__ import annotations

from collections.abc import Iterable
from typing import Tuple

import torch
from torch import optim
from torch.optim.lr_scheduler import (CosineAnnealingLR, LambdaLR, LRScheduler, SequentialLR)

from hparams.hp import Hparams


def _is_cuda(groups) -> bool:
    for g in groups:
        for p in (g["params"] if isinstance(g, dict) else [g]):
            return p.is_cuda
    return False


def optimizer_map(hp: Hparams, parameters) -> optim.Optimizer:
    hp.check_arg_in_hparams("identifier")
    if hp.identifier not in ("Adam", "AdamW"):
        raise NotImplementedError(f"The specified optimizer {hp.identifier} is not implemented yet.")
    hp.check_arg_in_hparams("lr", "beta1", "beta2")
    parameters = list(parameters)
    kw = dict(lr=hp.lr, betas=(hp.beta1, hp.beta2), eps=hp.get("eps", 1e-8))
    if _is_cuda(parameters):
        kw["fused"] = True
    if hp.identifier == "Adam":
        return optim.Adam(parameters, weight_decay=hp.get("weight_decay", 0), **kw)
    return optim.AdamW(parameters, weight_decay=hp.get("weight_decay", 0.01), **kw)


class ConstantLR(LRScheduler):
    def __init__(self, optimizer, lr, last_epoch=-1):
        self.lr = lr
        super().__init__(optimizer, last_epoch)

    def get_lr(self):
        if self.last_epoch == 0:
            return [g["lr"] for g in self.optimizer.param_groups]
        return [self.lr for _ in self.optimizer.param_groups]


def scheduler_map(hp: Hparams, optimizer: optim.Optimizer, total_steps: int) -> Tuple[LRScheduler, str]:
    hp.check_arg_in_hparams("identifier")
    phases, milestones, elapsed = [], [], 0
    if hp.has("warmup_steps"):
        n = hp.warmup_steps
        phases.append(LambdaLR(optimizer, lambda s: float(s) / float(max(1, n))))
        elapsed += n
        milestones.append(elapsed)
    if hp.has("flat_steps"):
        phases.append(LambdaLR(optimizer, lambda s: 1.0))
        elapsed += hp.flat_steps
        milestones.append(elapsed)
    assert total_steps > elapsed
    remaining = total_steps - elapsed - hp.get("finish_steps", 0)
    kind = hp.identifier
    if kind in ("linear_decay", "triangle"):
        phases.append(LambdaLR(optimizer, lambda s: max(0.0, float(remaining - s) / float(remaining))))
    elif kind == "constant":
        phases.append(LambdaLR(optimizer, lambda s: 1.0))
    elif kind == "cosine":
        phases.append(CosineAnnealingLR(optimizer, T_max=remaining, eta_min=hp.get("min_lr", 0)))
    else:
        raise NotImplementedError
    if hp.has("finish_steps"):
        assert hp.get("min_lr", 0) != 0
        phases.append(ConstantLR(optimizer, hp.min_lr))
        milestones.append(elapsed + remaining)
    if len(phases) > 1:
        return SequentialLR(optimizer, phases, milestones), "step"
    return phases[0], "step"


def create_optimizer(hp: Hparams, parameters: Iterable, total_steps: int):
    hp.check_arg_in_hparams("optimizer", "scheduler")
    parameters = list(parameters)
    if hp.optimizer.get("exclude_norm_and_bias_from_weight_decay", False):
        parameters = [{"params": [p for p in parameters if p.ndim != 1]},
                      {"params": [p for p in parameters if p.ndim == 1], "weight_decay": 0}]
    optimizer = optimizer_map(hp.optimizer, parameters)
    scheduler, interval = scheduler_map(hp.scheduler, optimizer, total_steps)
    return optimizer, {"scheduler": scheduler, "interval": interval}
