"""Causal multi-head self-attention on the MI355X HIP kernels.

Drop-in for the reference ``modules/attention/attention.py`` (``SelfAttention``
:21-98, ``reshape_head`` :12-18): same constructor, parameters
(``in_proj.weight`` (3d, d), ``out_proj.weight`` (d, d); biases only when the
hparams carry ``bias``), forward signature and returned dict.  What changes is
how the result is produced:

* ``in_proj`` / ``out_proj`` are MFMA GEMMs; the output re-mask
  (reference :80) is the GEMM epilogue's row predicate;
* no ``(B, H, T, T)`` mask or ALiBi tensor exists: the flash-style kernel
  applies ``key <= query`` and ``-slope_h (i - j)`` on the fly, so the
  ``rpe_bias`` handed from layer 0 to the later layers (reference
  modules/transformer/layers.py:163-175) is a tiny ``AlibiBias`` handle;
* heads are never split/merged in memory (reference :74,78): the kernel reads
  head ``h`` as columns ``64h..64h+63`` of the packed projection.
"""
from __future__ import annotations

import math
from typing import Any, Mapping, Optional, Tuple

import torch
import torch.nn as nn

import hipvg
from hipvg import functional as HF
from hparams.hp import Hparams
from utils.tensormask import TensorMask


def reshape_head(q, k, v, num_heads):
    """(B, T, C) -> (B, H, T, C/H) views (kept for API parity; debug paths only)."""
    def split(x):
        b, t, c = x.shape
        return x.view(b, t, num_heads, c // num_heads).transpose(1, 2)
    return split(q), split(k), split(v)


class AlibiBias(object):
    """Stands in for the dense ``rpe_bias`` tensor of the reference: carries the
    per-head slopes the kernels need (fp32, device)."""

    def __init__(self, slopes: torch.Tensor):
        self.slopes = slopes

    def dense(self, tq: int, tk: int) -> torch.Tensor:
        i = torch.arange(tk - tq, tk, device=self.slopes.device)[:, None]
        j = torch.arange(tk, device=self.slopes.device)[None, :]
        return (-self.slopes[:, None, None] * (i - j).abs().float())[None]


def _slopes_from(rpe_pair, rpe_bias, device) -> AlibiBias:
    if rpe_bias is not None:
        if not isinstance(rpe_bias, AlibiBias):
            raise NotImplementedError("dense attention bias tensors are not supported by the HIP path; "
                                      "pass the AlibiBias handle returned by the first layer")
        return rpe_bias
    if rpe_pair is not None and rpe_pair[0] == "ALiBi":
        return AlibiBias(rpe_pair[1].slopes.to(device=device, dtype=torch.float32).contiguous())
    if rpe_pair is not None and rpe_pair[0] is not None:
        raise NotImplementedError(f"positional encoding {rpe_pair[0]} has no HIP attention kernel")
    raise NotImplementedError("the HIP attention kernel implements causal ALiBi attention only")


class SelfAttention(nn.Module):
    def __init__(self, dim: int, hp: Hparams) -> None:
        super().__init__()
        hp.check_arg_in_hparams("nheads", "causal")
        self.hp = hp
        self.nheads, self.dim = hp.nheads, dim
        assert self.dim % self.nheads == 0
        self.head_dim = self.dim // self.nheads
        bias = bool(hp.get("bias", None))
        self.in_proj = nn.Linear(dim, dim * 3, bias=bias)
        self.out_proj = nn.Linear(dim, dim, bias=bias)
        self.dropout_p = hp.get("dropout", 0.0)
        if self.head_dim != 64 or not hp.causal or self.dropout_p:
            raise NotImplementedError("HIP attention: head_dim 64, causal, dropout 0 only "
                                      f"(got head_dim={self.head_dim}, causal={hp.causal}, p={self.dropout_p})")

    def forward(self, x: TensorMask,
                rpe_pair: Optional[Tuple[str, Any]] = None,
                rpe_bias=None,
                return_attn: bool = False,
                past_kv: Optional[Mapping[str, torch.Tensor]] = None,
                return_kv: bool = False) -> Mapping[str, Any]:
        B, Tq, D = x.value.shape
        dt = hipvg.compute_dtype()
        outputs = dict()
        bias_h = _slopes_from(rpe_pair, rpe_bias, x.value.device)
        if rpe_pair is not None and rpe_pair[0] == "ALiBi":
            outputs["rpe_bias"] = bias_h
        x2 = x.value.reshape(B * Tq, D).to(dt)
        lens = x.lengths32
        qkv = HF.linear(x2, self.in_proj.weight, self.in_proj.bias)
        if past_kv is None:
            ctx = HF.attention(qkv, bias_h.slopes, B, Tq, self.nheads, lens)
            k_all = v_all = None
        else:
            if Tq != 1:
                raise NotImplementedError("HIP decode path takes one new frame per step")
            q, k, v = qkv.view(B, 1, 3, D).unbind(2)
            k_all = torch.cat([past_kv["key"].to(dt), k], 1).contiguous()
            v_all = torch.cat([past_kv["value"].to(dt), v], 1).contiguous()
            pos = torch.full((B,), k_all.shape[1], dtype=torch.int32, device=x2.device)
            ctx = HF.attention_decode(q.reshape(B, D).contiguous(), k_all, v_all, bias_h.slopes, pos,
                                      self.nheads)
        out = HF.linear(ctx, self.out_proj.weight, self.out_proj.bias, lengths=lens, T=Tq)
        outputs["output"] = TensorMask(out.view(B, Tq, D), x.mask)
        if return_kv:
            if k_all is None:
                k_all = qkv.view(B, Tq, 3, D)[:, :, 1]
                v_all = qkv.view(B, Tq, 3, D)[:, :, 2]
            outputs["kv"] = {"key": k_all.detach(), "value": v_all.detach()}
        if return_attn:      # debugging only, plain torch math in fp32 (reference :86-92)
            with torch.no_grad():
                q3 = qkv.view(B, Tq, 3, D)[:, :, 0].float()
                kk = (k_all if k_all is not None else qkv.view(B, Tq, 3, D)[:, :, 1]).float()
                Tk = kk.shape[1]
                qh = q3.view(B, Tq, self.nheads, 64).transpose(1, 2)
                kh = kk.reshape(B, Tk, self.nheads, 64).transpose(1, 2)
                s = qh @ kh.transpose(-1, -2) / math.sqrt(64) + bias_h.dense(Tq, Tk)
                i = torch.arange(Tk - Tq, Tk, device=s.device)[:, None]
                j = torch.arange(Tk, device=s.device)[None, :]
                s = s.masked_fill(j > i, float("-inf"))
                outputs["attn"] = torch.softmax(s, -1)
        return outputs

    def custom_weight_init(self, init_std: float):
        bound = init_std / math.sqrt(self.dim / 3)
        nn.init.uniform_(self.in_proj.weight, -bound, bound)
        nn.init.uniform_(self.out_proj.weight, -bound, bound)


class CrossAttention(nn.Module):
    """The reference's cross-attention (:101-172) is only reachable from the
    text-conditioned LVTTS model, for which no config is shipped; it is outside
    the training hot path and has no HIP kernel."""

    def __init__(self, dim: int, hp: Hparams) -> None:
        super().__init__()
        raise NotImplementedError("CrossAttention is outside the VAE-GSLM training hot path "
                                  "(no shipped config uses it)")
