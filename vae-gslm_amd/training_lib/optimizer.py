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


class FlatAdamW(optim.AdamW):
    """``torch.optim.AdamW`` whose step is ONE HIP launch per gradient bucket (``vg_adamw``).

    Until :meth:`bind` is called it behaves exactly like the stock (fused multi-tensor) optimizer.
    ``bind(reducer)`` moves parameters, both moments and a bf16 copy of the weights into flat buffers
    laid out like the reducer's gradient buckets (``training_lib/dp.py``); ``param.data`` and the
    ``state[param]`` entries become views, so ``state_dict`` / LR schedulers / checkpoints see the usual
    structure.  The launch also refreshes the bf16 weights the MFMA GEMMs read and clears the gradient
    bucket, which replaces the per-parameter casts and the separate zero pass."""

    def __init__(self, params, **kw):
        super().__init__(params, **kw)
        self._flat = None
        self._steps = 0

    # ---- layout
    def bind(self, reducer) -> None:
        import ctypes as C
        import hipvg
        group_of = {}
        for gi, g in enumerate(self.param_groups):
            assert not g.get("amsgrad", False) and not g.get("maximize", False)
            for p in g["params"]:
                group_of[p] = gi
        assert len(self.param_groups) <= 4, "vg_adamw supports up to 4 parameter groups"
        self._flat = []
        for b in reducer.buckets:
            grad = b["flat"]
            n, dev = grad.numel(), grad.device
            P = torch.zeros(n, dtype=torch.float32, device=dev)
            M, V = torch.zeros_like(P), torch.zeros_like(P)
            S = torch.zeros(n, dtype=torch.bfloat16, device=dev)
            chunk_group = torch.zeros(n // 256, dtype=torch.uint8)
            for p, off in zip(b["params"], b["offsets"]):
                k = p.numel()
                P[off: off + k].copy_(p.data.reshape(-1))
                p.data = P[off: off + k].view(p.shape)
                p._vg_flat_shadow = S[off: off + k].view(p.shape)
                chunk_group[off // 256: (off + k + 255) // 256] = group_of[p]
                old = self.state.get(p)
                if old:                    # state loaded (or steps taken) before binding moves into the flat buffers
                    M[off: off + k].copy_(old["exp_avg"].reshape(-1))
                    V[off: off + k].copy_(old["exp_avg_sq"].reshape(-1))
                    self._steps = max(self._steps, int(float(old["step"])))
                self.state[p] = {"step": torch.tensor(float(self._steps)), "exp_avg": M[off: off + k].view(p.shape),
                                 "exp_avg_sq": V[off: off + k].view(p.shape)}
            S.copy_(P)
            self._flat.append(dict(P=P, G=grad, M=M, V=V, S=S, groups=chunk_group.to(dev)))
        self._ctypes = (C, hipvg)

    @torch.no_grad()
    def step(self, closure=None, grad_scale=None, bucket_wait=None):
        if self._flat is None:
            return super().step(closure)
        C, hipvg = self._ctypes
        self._steps += 1
        ng = len(self.param_groups)
        lr = (C.c_float * 4)(*[float(g["lr"]) for g in self.param_groups] + [0.0] * (4 - ng))
        wd = (C.c_float * 4)(*[float(g["weight_decay"]) for g in self.param_groups] + [0.0] * (4 - ng))
        b1, b2 = self.param_groups[0]["betas"]
        eps = self.param_groups[0]["eps"]
        for i, f in enumerate(self._flat):
            if bucket_wait is not None:
                bucket_wait(i)        # e.g. GradReducer.wait_bucket: update bucket i while i+1.. are still reducing
            hipvg.check(hipvg.lib().vg_adamw(hipvg.ptr(f["P"]), hipvg.ptr(f["G"]), hipvg.ptr(f["M"]), hipvg.ptr(f["V"]),
                                             hipvg.ptr(f["S"]), hipvg.ptr(f["groups"]), f["P"].numel(), lr, wd, ng,
                                             float(b1), float(b2), float(eps), self._steps, hipvg.ptr(grad_scale), 1,
                                             hipvg.stream()), "vg_adamw")
        return None

    @torch.no_grad()
    def sync_shadows(self) -> None:
        """Re-derive the bf16 weight copies from the fp32 masters (after the parameters were written by anything
        other than :meth:`step`, e.g. ``load_state_dict`` on the bound model)."""
        for f in self._flat or ():
            f["S"].copy_(f["P"])

    @property
    def clears_gradients(self) -> bool:
        """True once bound: ``step`` leaves the gradient buckets zeroed."""
        return self._flat is not None

    def state_dict(self):
        if self._flat is not None:
            for st in self.state.values():
                st["step"] = torch.tensor(float(self._steps))
        return super().state_dict()

    def load_state_dict(self, state_dict):
        if self._flat is None:
            return super().load_state_dict(state_dict)
        views = {p: (st["exp_avg"], st["exp_avg_sq"]) for p, st in self.state.items()}
        super().load_state_dict(state_dict)
        for p, (m, v) in views.items():          # keep the flat storage: copy the loaded moments into it
            st = self.state[p]
            m.copy_(st["exp_avg"])
            v.copy_(st["exp_avg_sq"])
            self._steps = int(float(st["step"]))
            st["exp_avg"], st["exp_avg_sq"] = m, v


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
    if kw.get("fused"):
        return FlatAdamW(parameters, weight_decay=hp.get("weight_decay", 0.01), **kw)
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
