"""Masked losses of the training step.

Drop-in for the hot-path functions of the reference ``training_lib/losses.py``
(``masked_loss`` :9-27, ``cross_entropy_loss`` :30-31, ``masked_ce_loss``
:34-41, L1/L2 wrappers :44-73).  The token cross-entropy runs on the fused HIP
log-softmax + NLL (+arg-max) kernel.  ``masked_loss`` takes an arbitrary
Python callable, so it stays generic tensor code; the KL term of the training
loss does not go through it in this build -- ``LVTR.forward`` returns the
fused ``kld`` computed by the prior-density kernel, and
``masked_loss(log_q, log_p, fn=a-b)`` on its outputs gives the same number.
InfoNCE / CPC / eos_loss of the reference are unused by VAE-GSLM and not provided.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from hipvg import functional as HF
from utils.tensormask import TensorMask


def masked_loss(x: TensorMask, y: TensorMask, fn: Callable, time_reduction: bool = False,
                batch_reduction: bool = False,
                batch_weight: Optional[torch.Tensor] = None) -> torch.Tensor:
    a = x.flatten().apply_mask().value
    b = y.flatten().apply_mask().value
    per_seq = fn(a, b).mean(-1).sum(-1)
    if batch_weight is not None:
        per_seq = per_seq * batch_weight
    if time_reduction and batch_reduction:
        return per_seq.sum() / x.length.sum()
    if time_reduction:
        return (per_seq / x.length).mean()
    if batch_reduction:
        return per_seq.mean()
    return per_seq.sum()


def cross_entropy_loss(a: torch.Tensor, b: torch.Tensor, reduction: str) -> torch.Tensor:
    """``F.cross_entropy(a, b, reduction, ignore_index=-100)`` on the HIP kernel
    (rows whose target is negative are ignored)."""
    logits = a.reshape(-1, a.shape[-1]).float().contiguous()
    total, _ = HF.cross_entropy_sum(logits, b.reshape(-1))
    if reduction == "sum":
        return total
    if reduction == "mean":
        return total / (b >= 0).sum().clamp_min(1)
    raise NotImplementedError("HIP cross entropy supports reduction='sum' | 'mean'")


def masked_ce_loss(x: TensorMask, y: TensorMask, reduction: str = "sum") -> torch.Tensor:
    B, T, V = x.value.shape
    logits = x.value.reshape(B * T, V).float().contiguous()
    total, _ = HF.cross_entropy_sum(logits, y.value.reshape(-1), x.lengths32, T)
    if reduction == "sum":
        return total
    if reduction == "mean":
        return total / x.length.sum()
    raise NotImplementedError("HIP cross entropy supports reduction='sum' | 'mean'")


def l1_loss(a, b):
    return torch.abs(a - b)


def l2_loss(a, b):
    return torch.pow(a - b, 2)


def masked_l1_loss(x, y, time_reduction=False, batch_reduction=False, batch_weight=None):
    return masked_loss(x, y, l1_loss, time_reduction, batch_reduction, batch_weight)


def masked_l2_loss(x, y, time_reduction=False, batch_reduction=False, batch_weight=None):
    return masked_loss(x, y, l2_loss, time_reduction, batch_reduction, batch_weight)
