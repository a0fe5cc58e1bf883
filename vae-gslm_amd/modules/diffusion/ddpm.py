"""1-D Gaussian diffusion decoder: training loss and samplers (reference
modules/diffusion/ddpm.py:127-374).  Stock PyTorch-ROCm ops; the thirteen fp32
schedule buffers keep the reference's names so checkpoints interchange.

``forward`` accepts the two random draws of a training step (``t``, ``noise``)
as optional arguments so a parity run can inject them.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import nn

from training_lib.losses import masked_l1_loss, masked_l2_loss
from utils.tensormask import TensorMask


def _betas(hp) -> torch.Tensor:
    kind, n = hp.beta_schedule.identifier, hp.timesteps
    if kind == "cosine":
        s = hp.beta_schedule.get("s", 0.008)
        grid = torch.linspace(0, n, n + 1, dtype=torch.float64) / n
        f = torch.cos((grid + s) / (1 + s) * math.pi * 0.5) ** 2
        f = f / f[0]
        return torch.clip(1 - f[1:] / f[:-1], 0, 0.999)
    if kind == "linear":
        k = 1000 / n
        return torch.linspace(k * 1e-4, k * 0.02, n, dtype=torch.float64)
    if kind == "scaled_linear":
        lo, hi = hp.beta_schedule.get("beta_start", 0.0015), hp.beta_schedule.get("beta_end", 0.0195)
        return torch.linspace(lo ** 0.5, hi ** 0.5, n, dtype=torch.float64) ** 2
    raise ValueError(f"unknown beta schedule {hp.beta_schedule}")


def _at(table: torch.Tensor, t: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    return table.gather(-1, t).reshape(t.shape[0], *((1,) * (like.dim() - 1)))


class GaussianDiffusion1D(nn.Module):
    def __init__(self, model, hp):
        super().__init__()
        self.hp, self.model = hp, model
        self.objective = hp.get("objective", "pred_noise")
        self.loss_type = hp.get("loss_type", "l1")
        self.clamp_range = hp.get("clamp_range", [-1, 1])
        self.ddim_sampling_eta = hp.get("ddim_sampling_eta", 1.0)
        self.sigma = 1.0
        betas = _betas(hp)
        self.num_timesteps = int(betas.shape[0])
        self.sampling_timesteps = hp.get("sampling_timesteps", None) or self.num_timesteps
        assert self.sampling_timesteps <= self.num_timesteps
        alphas = 1.0 - betas
        ac = torch.cumprod(alphas, 0)
        ac_prev = torch.cat([torch.ones(1, dtype=ac.dtype), ac[:-1]])
        post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
        tables = dict(
            betas=betas, alphas_cumprod=ac, alphas_cumprod_prev=ac_prev,
            sqrt_alphas_cumprod=ac.sqrt(), sqrt_one_minus_alphas_cumprod=(1 - ac).sqrt(),
            log_one_minus_alphas_cumprod=(1 - ac).log(), sqrt_recip_alphas_cumprod=(1 / ac).sqrt(),
            sqrt_recipm1_alphas_cumprod=(1 / ac - 1).sqrt(), posterior_variance=post_var,
            posterior_log_variance_clipped=post_var.clamp(min=1e-20).log(),
            posterior_mean_coef1=betas * ac_prev.sqrt() / (1 - ac),
            posterior_mean_coef2=(1 - ac_prev) * alphas.sqrt() / (1 - ac),
            p2_loss_weight=(1 + ac / (1 - ac)) ** -0.0)
        for name, val in tables.items():
            self.register_buffer(name, val.to(torch.float32))

    @property
    def is_ddim_sampling(self) -> bool:
        return self.sampling_timesteps < self.num_timesteps

    @property
    def loss_fn(self):
        if self.loss_type == "l1":
            return masked_l1_loss
        if self.loss_type == "l2":
            return masked_l2_loss
        raise ValueError(f"invalid loss type {self.loss_type}")

    # ------------------------------------------------------------ training
    def q_sample(self, x_start, t, noise=None):
        noise = torch.randn_like(x_start) if noise is None else noise
        return (_at(self.sqrt_alphas_cumprod, t, x_start) * x_start
                + _at(self.sqrt_one_minus_alphas_cumprod, t, x_start) * noise)

    def p_losses(self, x_start: TensorMask, t: torch.Tensor, cond: TensorMask,
                 noise: Optional[torch.Tensor] = None, loss_batch_weight=None, **kwargs):
        noise = torch.randn_like(x_start.value) if noise is None else noise
        xv = x_start.value
        if (xv.is_cuda and xv.dim() == 3 and self.objective == "pred_noise" and self.loss_type == "l1"
                and loss_batch_weight is None and not xv.requires_grad):
            # q_sample and the masked L1 as HIP row kernels (vg_qsample, vg_l1_rows_*): same arithmetic, 3 launches
            # forward and 1 backward instead of ~35 element-wise launches
            from hipvg import functional as HF
            B, T, C = xv.shape
            lens = x_start.lengths32
            plan = getattr(x_start.mask, "_vg_plan", None)
            # packed rows (a pseudo batch of one-frame sequences): the diffusion step of a row is its sequence's
            t_row = t if plan is None else t[plan.seq.clamp(max=plan.B - 1).long()]
            x_t = HF.qsample(xv.reshape(B * T, C), noise.reshape(B * T, C), self.sqrt_alphas_cumprod,
                             self.sqrt_one_minus_alphas_cumprod, t_row, lens, T)
            pred = self.model(TensorMask(x_t.view(B, T, C), x_start.mask), t, cond, **kwargs)
            return HF.masked_l1_sum(pred.value.reshape(B * T, C), noise.reshape(B * T, C), lens, T)
        x_t = TensorMask(self.q_sample(x_start.value, t, noise), x_start.mask).apply_mask()
        pred = self.model(x_t, t, cond, **kwargs)
        target = TensorMask(noise, x_start.mask).apply_mask() if self.objective == "pred_noise" else x_start
        pred = TensorMask(pred.value.float(), pred.mask)
        return self.loss_fn(pred, target, batch_weight=loss_batch_weight)

    def forward(self, img: TensorMask, cond: TensorMask, t: Optional[torch.Tensor] = None,
                noise: Optional[torch.Tensor] = None, **kwargs):
        if t is None:
            t = torch.randint(0, self.num_timesteps, (img.value.size(0),), device=img.device).long()
        return self.p_losses(img, t, cond, noise=noise, **kwargs)

    # ------------------------------------------------------------ sampling (inference side)
    def _predict(self, x: TensorMask, t: torch.Tensor, cond: TensorMask):
        out = self.model(x, t, cond)
        out = TensorMask(out.value.float(), out.mask)
        if self.objective == "pred_noise":
            x0 = (_at(self.sqrt_recip_alphas_cumprod, t, x.value) * x.value
                  - _at(self.sqrt_recipm1_alphas_cumprod, t, x.value) * out.value)
            return out, TensorMask(x0, out.mask).apply_mask()
        eps = ((_at(self.sqrt_recip_alphas_cumprod, t, x.value) * x.value - out.value)
               / _at(self.sqrt_recipm1_alphas_cumprod, t, x.value))
        return TensorMask(eps, out.mask).apply_mask(), out

    @torch.no_grad()
    def p_sample_loop(self, start: TensorMask, cond: TensorMask, **kwargs) -> TensorMask:
        img = start
        stride = self.num_timesteps // self.sampling_timesteps
        for step in reversed(range(0, self.num_timesteps, stride)):
            t = torch.full((img.value.shape[0],), step, device=img.device, dtype=torch.long)
            _, x0 = self._predict(img, t, cond)
            x0v = x0.value.clamp(self.clamp_range[0], self.clamp_range[1])
            mean = (_at(self.posterior_mean_coef1, t, x0v) * x0v
                    + _at(self.posterior_mean_coef2, t, x0v) * img.value)
            logvar = _at(self.posterior_log_variance_clipped, t, x0v)
            z = torch.randn_like(img.value) * self.sigma if step > 0 else 0.0
            img = TensorMask(mean + (0.5 * logvar).exp() * z, x0.mask).apply_mask()
        return img

    @torch.no_grad()
    def ddim_sample(self, start: TensorMask, cond: TensorMask, **kwargs) -> TensorMask:
        steps = torch.linspace(-1, self.num_timesteps - 1, steps=self.sampling_timesteps + 1)
        steps = list(reversed(steps.int().tolist()))
        img = start
        for cur, nxt in zip(steps[:-1], steps[1:]):
            t = torch.full((img.value.shape[0],), cur, device=img.device, dtype=torch.long)
            eps, x0 = self._predict(img, t, cond)
            x0 = TensorMask(x0.value.clamp(self.clamp_range[0], self.clamp_range[1]), x0.mask).apply_mask()
            if nxt < 0:
                img = x0
                continue
            a, a_next = self.alphas_cumprod[cur], self.alphas_cumprod[nxt]
            sig = self.ddim_sampling_eta * ((1 - a / a_next) * (1 - a_next) / (1 - a)).sqrt()
            keep = (1 - a_next - sig ** 2).sqrt()
            z = torch.randn_like(img.value) * self.sigma
            img = TensorMask(x0.value * a_next.sqrt() + keep * eps.value + sig * z, x0.mask).apply_mask()
        return img

    @torch.no_grad()
    def sample(self, start: TensorMask, cond: TensorMask, **kwargs) -> TensorMask:
        return (self.ddim_sample if self.is_ddim_sampling else self.p_sample_loop)(start, cond, **kwargs)
