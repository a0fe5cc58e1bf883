"""ALiBi (reference modules/position/alibi.py:6-33) without the table.

The reference registers a dense ``(H, maxpos, maxpos)`` fp32 buffer (64 MB at
the yaml's H=16, maxpos=1024) and every layer adds a ``(B, H, T, T)`` slice of
it to a materialised mask.  Here the bias ``-slope_h * |i - j|`` is evaluated
inside the attention kernels from the ``H`` slopes alone, so ``maxpos`` no
longer bounds the sequence length.  ``forward`` still returns
the dense slice for debugging / ``return_attn`` callers.
"""
import math

import torch
import torch.nn as nn


class ALiBi(nn.Module):
    def __init__(self, nheads: int, maxpos: int = 10000) -> None:
        super().__init__()
        self.nheads, self.maxpos = nheads, maxpos
        self.register_buffer("slopes", torch.tensor(self.get_slopes(nheads), dtype=torch.float32),
                             persistent=False)

    def get_slopes(self, n):
        def geometric(m):
            first = 2.0 ** (-(2.0 ** -(math.log2(m) - 3)))
            return [first ** (i + 1) for i in range(m)]
        if math.log2(n).is_integer():
            return geometric(n)
        base = 2 ** math.floor(math.log2(n))
        return geometric(base) + self.get_slopes(2 * base)[0::2][: n - base]

    def dense_bias(self, tq: int, tk: int, device=None) -> torch.Tensor:
        """Dense (H, tq, tk) bias, rows = the LAST tq positions of a tk-long sequence."""
        dev = device or self.slopes.device
        i = torch.arange(tk - tq, tk, device=dev)[:, None]
        j = torch.arange(tk, device=dev)[None, :]
        return -self.slopes.to(dev)[:, None, None] * (i - j).abs().float()

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.dense_bias(x.size(2), x.size(3), x.device)[:, -x.size(2):]
