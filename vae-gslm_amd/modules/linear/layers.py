"""TensorMask-aware dense layers on the HIP GEMM.

Drop-in for the hot-path classes of the reference ``modules/linear/layers.py``:
``Linear`` (:184-193), ``Embedding`` (:150-157), ``GaussianParameterize``
(:54-148), ``TimeAggregation`` (:260-262), ``FiLM`` (:265-292).  Parameter
names are unchanged (``linear.weight``, ``mean.weight``, ``logstd.weight`` ...).
The Gumbel / RVQ / LinearBlock classes of that file (:13-51, :160-181,
:196-257) belong to other model families; they are kept importable with the
same constructors and result types but run on stock PyTorch ops (no HIP kernel).
"""
from __future__ import annotations

from typing import Optional, Tuple, Union

import torch
import torch.nn as nn
import torch.nn.functional as F

import hipvg
from hipvg import functional as HF
from hparams.hp import Hparams
from modules.activations import get_activation, hip_act_id
from modules.norm import get_norm_fn
from utils.attr import AttrDict
from utils.helpers import repeat_batch
from utils.tensormask import TensorMask


def _gemm_dtype(in_dim: int, out_dim: int) -> torch.dtype:
    """bf16 MFMA needs 16-byte (8-element) aligned rows; the tiny latent-side
    projections (K or N = 4) run on the exact-f32 MFMA kernel instead."""
    dt = hipvg.compute_dtype()
    if dt == torch.bfloat16 and (in_dim % 8 or out_dim % 8):
        return torch.float32
    return dt


def dense_2d(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor], act="none",
             out_f32: bool = False, lengths=None, T: int = 0) -> torch.Tensor:
    """(..., K) -> (..., N) through the HIP GEMM, keeping leading dims."""
    lead = x.shape[:-1]
    dt = _gemm_dtype(weight.shape[1], weight.shape[0])
    x2 = x.reshape(-1, x.shape[-1]).to(dt).contiguous()
    y = HF.linear(x2, weight, bias, act=act, out_f32=out_f32, lengths=lengths, T=T)
    return y.view(*lead, weight.shape[0])


class Linear(nn.Module):
    def __init__(self, in_dim: int, out_dim: int, bias: bool = True, activation=nn.Identity()) -> None:
        super().__init__()
        self.linear = nn.Linear(in_dim, out_dim, bias=bias)
        self.activation = activation

    def forward(self, x: TensorMask) -> TensorMask:
        fused = hip_act_id(self.activation)
        y = dense_2d(x.value, self.linear.weight, self.linear.bias, act=fused or "none")
        if fused is None:
            y = self.activation(y)
        return TensorMask(y, x.mask)      # deliberately NOT masked (reference :192-193)


class Embedding(nn.Embedding):
    def forward(self, x: TensorMask) -> TensorMask:
        return TensorMask(super().forward(x.value), x.mask).apply_mask()

    def custom_weight_init(self, init_std: float):
        self._fill_padding_idx_with_zero()
        nn.init.uniform_(self.weight, -1.0, 1.0)


class GaussianParameterize(nn.Module):
    """Diagonal-Gaussian head: ``mean``/``logstd`` projections + reparameterised
    sample.  The two projections run as ONE GEMM (weights concatenated along N,
    fp32 output) followed by the fused reparameterisation kernel."""

    def __init__(self, in_dim: int, dim: int, bias: bool = True, std: Optional[float] = None,
                 std_range: Optional[Tuple[float, float]] = None,
                 truncated_norm: Optional[Tuple[float, float]] = None,
                 total_std: Optional[float] = None, use_tanh: bool = False, use_relu: bool = False,
                 normalization: bool = False, mean: Optional[float] = None):
        super().__init__()
        self._mean, self.dim, self.std = mean, dim, std
        if mean is None:
            self.mean = nn.Linear(in_dim, dim, bias=bias)
        if std is None:
            self.logstd = nn.Linear(in_dim, dim, bias=bias)
        self.truncated_norm = truncated_norm
        self.std_range = None
        if std_range is not None:
            assert std is None and len(std_range) == 2
            self.std_range = std_range
        self.total_std = total_std
        if total_std is not None:
            assert std is None and std_range is None
        self.use_tanh, self.use_relu, self.normalization = use_tanh, use_relu, normalization

    # ---- the two projections as one fp32-output GEMM: (..., in) -> (..., 2*dim) = [mean | logstd]
    def project(self, h: torch.Tensor) -> torch.Tensor:
        assert self._mean is None and self.std is None
        w = torch.cat([self.mean.weight, self.logstd.weight], 0)
        b = None if self.mean.bias is None else torch.cat([self.mean.bias, self.logstd.bias], 0)
        return dense_2d(h, w, b, out_f32=True)

    @property
    def plain(self) -> bool:
        return (self._mean is None and self.std is None and self.std_range is None
                and self.total_std is None and not (self.use_tanh or self.use_relu or self.normalization))

    def forward(self, x: TensorMask, temperature: float = 1.0,
                truncated_norm: Optional[Tuple[float, float]] = None,
                noise: Optional[torch.Tensor] = None) -> AttrDict:
        v = x.value
        if self.plain:
            mu_ls = self.project(v)
            mean, logstd = mu_ls[..., :self.dim], mu_ls[..., self.dim:]
        else:   # rarely-used parameterisations: same GEMMs, small torch post-ops
            if self._mean is None:
                mean = dense_2d(v, self.mean.weight, self.mean.bias, out_f32=True)
            else:
                mean = torch.full(v.shape[:2] + (self.dim,), self._mean, device=v.device)
            if self.normalization:
                mean = F.normalize(mean, p=2.0, dim=-1)
            if self.use_relu:
                mean = F.relu(mean)
            if self.use_tanh:
                mean = torch.tanh(mean) * 0.5
            if self.std is None:
                logstd = dense_2d(v, self.logstd.weight, self.logstd.bias, out_f32=True)
                if self.std_range is not None:
                    hi, lo = self.std_range
                    logstd = torch.log(torch.sigmoid(logstd) * (hi - lo) + lo)
            else:
                logstd = torch.log(torch.full(mean.size(), self.std, device=v.device))
            if self.total_std is not None:
                std = torch.exp(logstd.float())
                std = std / std.sum(-1, keepdim=True) * self.total_std * std.size(-1)
                logstd = torch.log(std)
        if noise is None:
            noise = torch.randn_like(mean)
        tn = truncated_norm if truncated_norm is not None else self.truncated_norm
        if tn is not None:
            noise = nn.init.trunc_normal_(torch.empty_like(mean), a=tn[0], b=tn[1])
        D = self.dim
        sample, _ = HF.reparameterize(mean.reshape(-1, D), logstd.reshape(-1, D), noise.reshape(-1, D),
                                      temperature)
        return AttrDict(mean=TensorMask(mean, x.mask), logstd=TensorMask(logstd, x.mask),
                        sample=TensorMask(sample.view(mean.shape), x.mask))

    def sample(self, n: int, mean: TensorMask, logstd: TensorMask, temperature: float = 1.0) -> AttrDict:
        mean, logstd = repeat_batch(mean, n), repeat_batch(logstd, n)
        D = mean.value.shape[-1]
        eps = torch.randn_like(mean.value)
        s, _ = HF.reparameterize(mean.value.reshape(-1, D), logstd.value.reshape(-1, D),
                                 eps.reshape(-1, D), temperature)
        return AttrDict(mean=mean, logstd=logstd, sample=TensorMask(s.view(mean.value.shape), mean.mask))


class TimeAggregation(nn.Module):
    def forward(self, x: TensorMask) -> torch.Tensor:
        if getattr(x.mask, "_vg_full", False):       # no padding: the masked mean over time is the plain mean
            v = x.flatten().value
            return v.sum(1) / float(v.shape[1])
        return x.flatten().apply_mask().value.sum(1) / x.length[..., None]


class FiLM(nn.Module):
    """Feature-wise affine modulation ``y = w(c) * x + b(c)`` (conv stacks / flow)."""

    def __init__(self, dim: int, bias: bool = True, time_first: bool = True, in_dim: int = None):
        super().__init__()
        in_dim = dim if in_dim is None else in_dim
        self.linear = (nn.Linear(in_dim, dim * 2, bias=bias) if time_first
                       else nn.Conv1d(in_dim, dim * 2, 1, bias=bias))
        self.time_first = time_first

    def modulation(self, c: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        if self.time_first:
            wb = dense_2d(c, self.linear.weight, self.linear.bias)
            return wb.chunk(2, -1)
        return self.linear(c).chunk(2, 1)

    def forward(self, x: Union[torch.Tensor, TensorMask], c: Union[torch.Tensor, TensorMask]):
        y = x.value if isinstance(x, TensorMask) else x
        c = c.value if isinstance(c, TensorMask) else c
        w, b = self.modulation(c)
        y = w * y + b
        if isinstance(x, TensorMask):
            return TensorMask(y, x.mask, axis=1 if self.time_first else 2)
        return y


# ---------------------------------------------------------------------------------------------------------
# Not on the VAE-GSLM path (stock PyTorch ops): discrete / residual-VQ front ends and the residual MLP stack of the
# reference's other model families.  Same constructors, parameter names and result types.
class GumbelSoftMaxParameterize(nn.Module):
    """Straight-through Gumbel-softmax code selection (reference :13-51)."""

    def __init__(self, in_dim: int, num_codebooks: int, codebook_dim: int, temperature: float = 1.0):
        super().__init__()
        self.in_dim, self.temperature = in_dim, temperature
        self.in_linear = nn.Linear(in_dim, num_codebooks, bias=False)
        self.encode_linear = nn.Linear(num_codebooks, codebook_dim, bias=False)

    def gumbel_softmax_sample(self, logits: torch.Tensor, temperature: float, eps: float = 1e-20) -> torch.Tensor:
        noise = -torch.log(eps - torch.log(torch.rand_like(logits) + eps))
        return F.softmax((logits + noise) / temperature, dim=-1)

    def forward(self, x: TensorMask, temperature: Optional[float] = None) -> AttrDict:
        logits = self.in_linear(x.value) * self.in_dim ** -0.5
        soft = self.gumbel_softmax_sample(logits, self.temperature if temperature is None else temperature)
        hard = F.one_hot(soft.argmax(-1), soft.shape[-1]).to(soft.dtype)
        code = (hard - soft).detach() + soft                 # one-hot forward, soft gradient
        return AttrDict(logits=TensorMask(logits, x.mask).apply_mask(-1000),
                        output=TensorMask(self.encode_linear(code), x.mask).apply_mask(),
                        gumbel_prob=TensorMask(soft, x.mask).apply_mask())


class RVQEmbedding(nn.Module):
    """Sum of one embedding table per residual quantizer: (B, T, n) ids -> (B, T, C) (reference :160-181)."""

    def __init__(self, num_quantizers: int, codebook_size: int, dim: int) -> None:
        super().__init__()
        self.num_quantizers = num_quantizers
        self.embeddings = nn.ModuleList([nn.Embedding(codebook_size, dim) for _ in range(num_quantizers)])

    def forward(self, x: TensorMask) -> TensorMask:
        total = sum(table(x.value[..., i]) for i, table in enumerate(self.embeddings))
        return TensorMask(total, x.mask).apply_mask()


class LinearBlock(nn.Module):
    """Pre-norm residual MLP block: x + W2 act(norm2(W1 act(norm1 x))) (reference :196-228)."""

    def __init__(self, hp: Hparams):
        super().__init__()
        hp.check_arg_in_hparams("hidden_dim", "activation", "norm")
        bias, width = hp.get("bias", True), hp.hidden_dim
        self.linear1 = nn.Linear(width, width, bias=bias)
        self.linear2 = nn.Linear(width, width, bias=bias)
        self.dropout = nn.Dropout(hp.get("dropout", 0.0))
        self.norm1 = get_norm_fn(width, hp.norm)
        self.norm2 = get_norm_fn(width, hp.norm)
        self.activation = get_activation(hp.activation)

    def forward(self, x: TensorMask) -> TensorMask:
        r = self.linear1(self.activation(self.norm1(x.value)))
        r = self.linear2(self.activation(self.norm2(r)))
        return TensorMask(x.value + r, x.mask).apply_mask()


class LinearLayerStack(nn.Module):
    def __init__(self, hp: Hparams, input_dim: Optional[int] = None, output_dim: Optional[int] = None) -> None:
        super().__init__()
        hp.check_arg_in_hparams("num_layers", "layer")
        self.hp = hp
        self.layers = nn.ModuleList([LinearBlock(hp.layer) for _ in range(hp.num_layers)])
        self.linear = nn.Linear(input_dim, hp.layer.hidden_dim) if input_dim is not None else None
        self.out_linear = nn.Linear(hp.layer.hidden_dim, output_dim) if output_dim is not None else None

    def forward(self, x: TensorMask) -> TensorMask:
        if self.linear is not None:
            x = TensorMask(self.linear(x.value), x.mask).apply_mask()
        for layer in self.layers:
            x = layer(x)
        if self.out_linear is not None:
            x = TensorMask(self.out_linear(x.value), x.mask).apply_mask()
        return x
