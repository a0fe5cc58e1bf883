"""Conditional affine-coupling flow on the 4-d latent (reference
modules/flow/layers.py:15-98,199-245).  State-dict layout is the reference's
(``layers.N.{film.linear, linear1, norm, linear2}``).

MI355X note: the only FLOP-carrying part is the FiLM projection of the
1024-d Transformer state (4 x Linear(1024 -> 128)); the stack evaluates all
of them as ONE HIP GEMM (weights concatenated along N) and hands each layer
its slice.  The 2->64->4 coupling nets are tiny tensor ops.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from hparams.hp import Hparams
from modules.activations import get_activation
from modules.linear.layers import FiLM, dense_2d
from modules.norm import get_norm_fn
from utils.tensormask import TensorMask

from .utils import TensorLogdet


class LinearCoupling(nn.Module):
    def __init__(self, dim: int, flip: bool, hp: Hparams, condition_dim: Optional[int] = None):
        super().__init__()
        hp.check_arg_in_hparams("hidden_dim", "activation", "mean_only", "norm")
        self.mean_only, self.flip = hp.mean_only, flip
        self.condition_dim = condition_dim
        if condition_dim is not None:
            self.film = FiLM(hp.hidden_dim, in_dim=condition_dim)
        use_bias = hp.get("bias", True)
        self.linear1 = nn.Linear(dim // 2, hp.hidden_dim, bias=use_bias)
        self.linear2 = nn.Linear(hp.hidden_dim, dim // 2 if hp.mean_only else dim, bias=use_bias)
        self.norm = get_norm_fn(hp.hidden_dim, hp.norm)
        self.activation = get_activation(hp.activation)
        self.scale_range = hp.get("scale_range", None)
        self.detach_coupling = hp.get("detach_coupling", False)

    def _shift_logscale(self, keep: torch.Tensor, film_wb):
        s = self.norm(self.linear1(keep.detach() if self.detach_coupling else keep))
        if film_wb is not None:
            s = film_wb[0] * s + film_wb[1]
        s = self.linear2(self.activation(s))
        if self.mean_only:
            return s, torch.zeros_like(s)
        shift, logs = s.chunk(2, -1)
        if self.scale_range is not None:
            hi, lo = self.scale_range          # the reference unpacks (_max, _min) in this order
            logs = torch.log(torch.sigmoid(logs) * (hi - lo) + lo)
        return shift, logs

    def _film(self, c, film_wb):
        if film_wb is not None or c is None or self.condition_dim is None:
            return film_wb
        cv = c.value if isinstance(c, TensorMask) else c
        return tuple(t.float() for t in self.film.modulation(cv))

    def forward(self, x: TensorLogdet, c: Optional[TensorMask] = None, film_wb=None) -> TensorLogdet:
        tm = x.tensor
        a, b = tm.value.chunk(2, -1)
        keep, move = (b, a) if self.flip else (a, b)
        shift, logs = self._shift_logscale(keep, self._film(c, film_wb))
        out = torch.cat([keep, shift + move * torch.exp(logs)], -1)
        logdet = x.logdet + TensorMask.use_mask(logs, tm.mask)
        return TensorLogdet(TensorMask(out, tm.mask, axis=tm.axis), logdet)

    def reverse(self, x: TensorMask, c: Optional[TensorMask] = None, film_wb=None) -> TensorMask:
        keep, moved = x.value.chunk(2, -1)
        shift, logs = self._shift_logscale(keep, self._film(c, film_wb))
        orig = (moved - shift) * torch.exp(-logs)
        pair = (orig, keep) if self.flip else (keep, orig)
        return TensorMask(torch.cat(pair, -1), x.mask, axis=x.axis)


class CouplingStack(nn.Module):
    def __init__(self, dim: int, hp: Hparams, condition_dim: Optional[int] = None) -> None:
        super().__init__()
        hp.check_arg_in_hparams("num_layers", "layer")
        assert hp.num_layers % 2 == 0
        self.identifier = hp.get("identifier", "LinearCoupling")
        if self.identifier != "LinearCoupling":
            raise NotImplementedError(f"{self.identifier}: only LinearCoupling is used by vae-gslm.yaml")
        self.condition_dim, self.dim = condition_dim, dim
        self.layers = nn.ModuleList([LinearCoupling(dim, True, hp.layer, condition_dim=condition_dim)
                                     for _ in range(hp.num_layers)])

    def film_all(self, c):
        """All layers' FiLM projections as one GEMM -> list of (weight, bias) per layer."""
        if c is None or self.condition_dim is None:
            return [None] * len(self.layers)
        cv = c.value if isinstance(c, TensorMask) else c
        W = torch.cat([l.film.linear.weight for l in self.layers], 0)
        Bv = torch.cat([l.film.linear.bias for l in self.layers], 0)
        wb = dense_2d(cv, W, Bv, out_f32=True)
        return [tuple(part.chunk(2, -1)) for part in wb.chunk(len(self.layers), -1)]

    def forward(self, x: TensorLogdet, c: Optional[TensorMask] = None) -> TensorLogdet:
        for layer, wb in zip(self.layers, self.film_all(c)):
            x = layer(x, c=c, film_wb=wb)
        return x

    def reverse(self, x: TensorMask, c: Optional[TensorMask] = None) -> TensorMask:
        for layer, wb in zip(reversed(self.layers), reversed(self.film_all(c))):
            x = layer.reverse(x, c=c, film_wb=wb)
        return x
