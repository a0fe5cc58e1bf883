"""T5 bucketed relative-position bias -- importable surface of the reference ``modules/position/t5.py`` (``T5RPE`` :7).

Not used by vae-gslm.yaml (ALiBi is; its bias is evaluated inside the HIP attention kernels).  A T5 bias is a learned
dense (H, Tq, Tk) table, which the in-kernel path does not take, so this module is plain tensor code for direct callers:
``forward(x)`` with ``x`` of shape (B, H, Tq, Tk) returns the (H, Tq, Tk) bias.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn


class T5RPE(nn.Module):
    def __init__(self, nheads: int, bidirectional: bool, num_buckets: int = 32, max_distance: int = 128) -> None:
        super().__init__()
        self.bidirectional = bidirectional
        self.num_buckets = num_buckets
        self.max_distance = max_distance
        self.relative_attention_bias = nn.Embedding(num_buckets, nheads)

    @staticmethod
    def _relative_position_bucket(relative_position, bidirectional=True, num_buckets=32, max_distance=128):
        """Bucket of ``key_pos - query_pos``: half of the buckets hold exact small distances, the other half
        logarithmically growing ranges up to ``max_distance``; a bidirectional table spends half of its buckets on
        each sign, a causal one clamps positive (future) offsets to distance zero."""
        rel = relative_position
        offset = torch.zeros_like(rel)
        if bidirectional:
            num_buckets //= 2
            offset = (rel > 0).long() * num_buckets
            dist = rel.abs()
        else:
            dist = (-rel).clamp_min(0)
        exact = num_buckets // 2
        log_ratio = torch.log(dist.float().clamp_min(1) / exact) / math.log(max_distance / exact)
        far = (exact + (log_ratio * (num_buckets - exact)).long()).clamp_max(num_buckets - 1)
        return offset + torch.where(dist < exact, dist, far)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        tq, tk = x.size(2), x.size(3)
        q = torch.arange(tq, dtype=torch.long, device=x.device)[:, None]
        k = torch.arange(tk, dtype=torch.long, device=x.device)[None, :]
        bucket = self._relative_position_bucket(k - q, self.bidirectional, self.num_buckets, self.max_distance)
        return self.relative_attention_bias(bucket).permute(2, 0, 1)
