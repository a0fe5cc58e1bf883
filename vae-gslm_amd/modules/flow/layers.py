"""Conditional affine-coupling flow on the 4-d latent (reference
modules/flow/layers.py:15-98,199-245).  State-dict layout is the reference's
(``layers.N.{film.linear, linear1, norm, linear2}``).

MI355X note: the only FLOP-carrying part is the FiLM projection of the
1024-d Transformer state (4 x Linear(1024 -> 128)); the stack evaluates all
of them as ONE HIP GEMM (weights concatenated along N) and hands each layer
its slice.  The 2->64->4 coupling nets are tiny tensor ops.
"""
from __future__ import annotations

import os as _os
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

    # ---- HIP row-kernel path (hipvg vg_flow_*): the vae-gslm.yaml shape of the stack
    def _hip_ok(self, value: torch.Tensor, c) -> bool:
        if _os.environ.get("VG_FLOW_STOCK", "0") == "1" or c is None or self.condition_dim is None:
            return False
        if not value.is_cuda or self.dim != 4 or len(self.layers) > 8:
            return False
        l0 = self.layers[0]
        return (l0.linear1.out_features == 64 and not l0.mean_only and l0.scale_range is not None
                and isinstance(l0.norm, nn.LayerNorm) and l0.norm.elementwise_affine
                and isinstance(l0.activation, nn.GELU) and getattr(l0.activation, "approximate", "none") == "none"
                and not l0.detach_coupling and l0.linear1.bias is not None and l0.linear2.bias is not None
                and all(l.flip for l in self.layers))

    def _hip_args(self, c):
        cv = c.value if isinstance(c, TensorMask) else c
        from hipvg import functional as HF
        import hipvg
        # the layers' FiLM projections as one product, each weight's gradient sunk on its own (HF.stacked_linear)
        x2 = cv.reshape(-1, cv.shape[-1]).to(hipvg.compute_dtype()).contiguous()
        wb = HF.stacked_linear(x2, [l.film.linear.weight for l in self.layers],
                               [l.film.linear.bias for l in self.layers], out_f32=True)
        wb = wb.reshape(-1, 128 * len(self.layers))
        params = []
        for l in self.layers:
            params += [l.linear1.weight, l.linear1.bias, l.norm.weight, l.norm.bias, l.linear2.weight, l.linear2.bias]
        hi, lo = self.layers[0].scale_range
        return wb, params, dict(eps=self.layers[0].norm.eps, hi=float(hi), lo=float(lo))

    def forward(self, x: TensorLogdet, c: Optional[TensorMask] = None) -> TensorLogdet:
        tm = x.tensor
        if isinstance(tm, TensorMask) and tm.axis == 1 and self._hip_ok(tm.value, c):
            from hipvg import functional as HF
            B, T = tm.value.shape[:2]
            wb, params, kw = self._hip_args(c)
            u, logdet = HF.coupling_flow(tm.value.reshape(-1, self.dim).float(), wb, params,
                                         lengths=tm.lengths32, T=T, **kw)
            return TensorLogdet(TensorMask(u.view(B, T, self.dim), tm.mask, axis=tm.axis),
                                x.logdet + logdet.view(B, T, 1))
        for layer, wb in zip(self.layers, self.film_all(c)):
            x = layer(x, c=c, film_wb=wb)
        return x

    def reverse(self, x: TensorMask, c: Optional[TensorMask] = None) -> TensorMask:
        if isinstance(x, TensorMask) and x.axis == 1 and self._hip_ok(x.value, c):
            from hipvg import functional as HF
            B, T = x.value.shape[:2]
            wb, params, kw = self._hip_args(c)
            z = HF.coupling_flow_reverse(x.value.reshape(-1, self.dim), wb, params, **kw)
            return TensorMask(z.view(B, T, self.dim).to(x.value.dtype), x.mask, axis=x.axis)
        for layer, wb in zip(reversed(self.layers), reversed(self.film_all(c))):
            x = layer.reverse(x, c=c, film_wb=wb)
        return x
