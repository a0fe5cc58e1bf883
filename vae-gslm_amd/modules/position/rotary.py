"""Rotary position embedding -- importable surface of the reference ``modules/position/rotary.py`` (``Rotary`` :59).

Nothing in vae-gslm.yaml reaches this class (the reference's factory matches the misspelt name "Rotery" and its
attention path would feed activations in as positions), so there is no HIP kernel behind it: it is plain
tensor code for callers that import it directly.  Supported: the language-frequency table (``freqs_for='lang'``),
optional learned frequencies, position interpolation, and ``rotate_queries_or_keys``; the xpos variant and the pixel /
constant tables raise ``NotImplementedError``.
"""
from __future__ import annotations

import torch
import torch.nn as nn


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    """(x0, x1, x2, x3, ...) -> (-x1, x0, -x3, x2, ...) on the last axis."""
    pairs = x.reshape(x.shape[:-1] + (-1, 2))
    return torch.stack((-pairs[..., 1], pairs[..., 0]), -1).reshape(x.shape)


def apply_rotary_emb(freqs: torch.Tensor, t: torch.Tensor, start_index: int = 0, scale=1.0) -> torch.Tensor:
    """Rotate channels ``start_index : start_index + freqs.shape[-1]`` of ``t`` by the angles ``freqs``."""
    width = freqs.shape[-1]
    end = start_index + width
    assert end <= t.shape[-1], "not enough channels to rotate"
    freqs = freqs.to(t.dtype)
    mid = t[..., start_index:end]
    mid = mid * freqs.cos() * scale + rotate_half(mid) * freqs.sin() * scale
    return torch.cat((t[..., :start_index], mid, t[..., end:]), -1)


class Rotary(nn.Module):
    def __init__(self, dim, custom_freqs=None, freqs_for="lang", theta=10000, max_freq=10, num_freqs=1,
                 learned_freq=False, use_xpos=False, xpos_scale_base=512, interpolate_factor=1.0,
                 theta_rescale_factor=1.0):
        super().__init__()
        if use_xpos:
            raise NotImplementedError("Rotary(use_xpos=True) is not provided by the MI355X build")
        if custom_freqs is not None:
            freqs = torch.as_tensor(custom_freqs, dtype=torch.float32)
        elif freqs_for == "lang":
            theta = theta * theta_rescale_factor ** (dim / (dim - 2))
            freqs = theta ** (-torch.arange(0, dim, 2)[: dim // 2].float() / dim)
        elif freqs_for in ("pixel", "constant"):
            raise NotImplementedError(f"Rotary(freqs_for={freqs_for!r}) is not provided by the MI355X build")
        else:
            raise ValueError(f"unknown modality {freqs_for}")
        assert interpolate_factor >= 1.0
        self.interpolate_factor = interpolate_factor
        self.use_xpos = False
        self.freqs = nn.Parameter(freqs, requires_grad=learned_freq)
        self.register_buffer("scale", None)

    def get_seq_pos(self, seq_len, device, dtype, offset=0):
        return (torch.arange(seq_len, device=device, dtype=dtype) + offset) / self.interpolate_factor

    def forward(self, t, cache_key=None) -> torch.Tensor:
        """Angles for positions ``t`` (a tensor or a callable returning one): (..., 2 * len(freqs))."""
        if callable(t):
            t = t()
        ang = t.to(self.freqs.dtype)[..., None] * self.freqs
        return ang.repeat_interleave(2, -1)

    def rotate_queries_or_keys(self, t: torch.Tensor, seq_dim: int = -2, offset: int = 0) -> torch.Tensor:
        pos = self.get_seq_pos(t.shape[seq_dim], t.device, t.dtype, offset)
        freqs = self.forward(pos)
        if seq_dim not in (-2, t.dim() - 2):
            freqs = freqs.reshape((freqs.shape[0],) + (1,) * (t.dim() - 2 - (seq_dim % t.dim())) + (freqs.shape[1],))
        return apply_rotary_emb(freqs, t)
