"""Sinusoidal position table (reference modules/position/absolute.py:6-36);
the diffusion time embedding looks rows up with ``get``."""
import math

import torch
import torch.nn as nn


def sincos_table(maxpos: int, ndim: int) -> torch.Tensor:
    pos = torch.arange(maxpos, dtype=torch.float32)[:, None]
    freq = torch.exp(torch.arange(0, ndim, 2, dtype=torch.float32) * -(math.log(10000.0) / ndim))
    table = torch.zeros(maxpos, ndim)
    table[:, 0::2] = torch.sin(pos * freq)
    table[:, 1::2] = torch.cos(pos * freq)
    return table


class SinCos(nn.Module):
    def __init__(self, ndim: int, maxpos: int = 10000, fixed_pos: bool = False, scaled: bool = False):
        super().__init__()
        self.register_buffer("p", sincos_table(maxpos, ndim), persistent=False)
        self.scalar = nn.Parameter(torch.ones(1)) if scaled else 1.0
        self.fixed_pos = fixed_pos

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        p = self.p if self.fixed_pos else self.p[: x.size(1)]
        return x + self.scalar * p.unsqueeze(0)

    def get(self, x: torch.Tensor) -> torch.Tensor:
        return self.p[x]
