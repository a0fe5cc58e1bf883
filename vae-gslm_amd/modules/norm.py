"""Normalisation layers (reference modules/norm.py:6-47).

``RMSNorm`` runs on the HIP row kernel (one wave64 per frame, fp32 statistics);
the Transformer layer calls the same kernel through ``hipvg.functional.rmsnorm``
with the padding mask fused in.  ``InstanceNorm`` is the reference's per-frame
normalisation over the channel axis of a (B, C, T) tensor with unbiased
variance; it belongs to the conv stacks, which run on stock PyTorch-ROCm ops.
"""
import torch
import torch.nn as nn

import hipvg
from hipvg import functional as HF
from hparams.hp import Hparams


class RMSNorm(nn.Module):
    def __init__(self, dim: int, eps: float = 1e-5) -> None:
        super().__init__()
        self.eps = eps
        self.scale = nn.Parameter(torch.ones(dim))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        shape = x.shape
        dt = x.dtype if x.dtype in (torch.float32, torch.bfloat16) else torch.float32
        y = HF.rmsnorm(x.reshape(-1, shape[-1]).to(dt).contiguous(), self.scale, self.eps)
        return y.float().reshape(shape)      # the reference up-casts: x.float()


class InstanceNorm(nn.Module):
    """Asserts B, C, T; statistics over C for every (b, t)."""

    def __init__(self, dim: int, eps: float = 1e-5) -> None:
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = x.float()
        var, mean = torch.var_mean(x, dim=1, keepdim=True)
        xn = (x - mean) * torch.rsqrt(var + self.eps)
        return torch.addcmul(self.bias[:, None], self.weight[:, None], xn)


def get_norm_fn(dim, hp: Hparams) -> nn.Module:
    kind = hp.identifier
    if kind == "RMSNorm":
        return RMSNorm(dim, eps=hp.eps)
    if kind == "InstanceNorm":
        return InstanceNorm(dim, eps=hp.eps)
    if kind == "LayerNorm":
        return nn.LayerNorm(dim, eps=hp.eps)
    if kind == "GroupNorm":
        return nn.GroupNorm(hp.num_groups, dim, eps=hp.eps)
    if kind == "Identity":
        return nn.Identity()
    raise ValueError(f"{kind} not in the usable normalization function lists.")
