"""Convolutional stacks that run INSIDE the training step but outside the HIP
hot path (SURVEY.md section 2, rows marked as stock PyTorch-ROCm): the
posterior encoder / diffusion UNet body (``BottleNeckResNet``) and the
utterance encoder (``CNNStack``).

Only the state-dict layout is dictated by the reference
(modules/conv/layers.py:70-135,231-295,386-652: ``linear``, ``layers.N.{norm,
conv1,conv2,conv3,time_emb}``, ``skip_conv.N``, ``final_norm``, ``out_linear``;
``layers.N.{conv,norm}`` for the CNN stack) so its checkpoints load with
``strict=True``.  The implementation is a single configurable bottleneck block
instead of the reference's four subclasses, and supports exactly what
``vae-gslm.yaml`` instantiates: resample rate 1, "concat" conditioning,
optional diffusion-time embedding, concat skip connections.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

import hipvg
from hipvg import functional as HF
from hparams.hp import Hparams
from modules.activations import get_activation, hip_act_id
from modules.linear.layers import dense_2d
from modules.norm import InstanceNorm, get_norm_fn
from utils.helpers import get_padding
from utils.tensormask import TensorMask


def channel_norm_rows(y: torch.Tensor, norm, T: int) -> torch.Tensor:
    """Per-frame channel norm on [B*T, C] rows: the fused HIP row kernel when the width fills whole
    wavefront vectors (512 bf16 / 256 fp32 channels per pass), plain tensor ops for the narrow
    utterance-encoder layers (a few hundred frames)."""
    width = 64 * (8 if y.dtype == torch.bfloat16 else 4)
    if y.shape[1] % width == 0 and y.shape[1] // width <= 2:
        return HF.channel_norm(y, norm.weight, norm.bias, T=T, eps=norm.eps)
    if y.is_cuda and 2 <= y.shape[1] <= 1024 and os.environ.get("VG_NARROW_NORM", "1") != "0":
        return HF.narrow_channel_norm(y, norm.weight, norm.bias, eps=norm.eps)      # vg_chnorm_*: any width
    x = y.float()
    var, mean = torch.var_mean(x, dim=-1, keepdim=True)
    return (norm.weight * ((x - mean) * torch.rsqrt(var + norm.eps)) + norm.bias).to(y.dtype)


class Conv1d(nn.Conv1d):
    """Conv1d that accepts an asymmetric ``padding=(left, right)`` tuple."""

    def __init__(self, *args, **kwargs):
        pad = kwargs.get("padding", 0)
        self.two_side_padding = None
        if isinstance(pad, tuple):
            assert len(pad) == 2
            self.two_side_padding = pad
            kwargs["padding"] = 0
        super().__init__(*args, **kwargs)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.two_side_padding is not None:
            x = F.pad(x, list(self.two_side_padding))
        return super().forward(x)


class BottleneckBlock(nn.Module):
    """x + conv3(act(conv2([norm(dwconv(x) + t_emb) ; cond])))   on (B, C, T)."""

    def __init__(self, channels: int, hidden: int, lhp: Hparams, cond_dim: int = 0,
                 time_dim: Optional[int] = None):
        super().__init__()
        lhp.check_arg_in_hparams("kernel_size", "norm", "activation")
        assert lhp.norm.identifier != "LayerNorm", "BCT format not supported"
        if lhp.get("shortcut", False) or lhp.has("layer_scale") or lhp.get("dropout", 0.0):
            raise NotImplementedError("shortcut / layer_scale / dropout variants are not used by vae-gslm.yaml")
        pad = get_padding(lhp.kernel_size, causal=lhp.get("causal_padding", False),
                          future=lhp.get("future_padding", False))
        self.norm = get_norm_fn(channels, lhp.norm)
        self.act = get_activation(lhp.activation)
        self.conv1 = Conv1d(channels, channels, kernel_size=lhp.kernel_size, padding=pad, groups=channels)
        self.conv2 = nn.Conv1d(channels + cond_dim, hidden, kernel_size=1)
        self.conv3 = nn.Conv1d(hidden, channels, kernel_size=1)
        if time_dim is not None:
            self.time_emb = nn.Linear(time_dim, channels)
        self.has_time, self.has_cond = time_dim is not None, cond_dim > 0
        k = lhp.kernel_size
        self.taps = k
        self.shift = (k - 1) if lhp.get("causal_padding", False) else (0 if lhp.get("future_padding", False)
                                                                      else (k - 1) // 2)

    def hip_ready(self) -> bool:
        return isinstance(self.norm, InstanceNorm) and hip_act_id(self.act) in ("relu", "silu")

    def forward_rows(self, x2: torch.Tensor, T, cond2: Optional[torch.Tensor],
                     temb: Optional[torch.Tensor], te: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Channels-last HIP path: x2 [B*T, C] in the compute dtype.  ``te``: this block's time-embedding
        projection if the caller already computed it (one batched GEMM for all blocks).  ``T``: frames per sequence,
        or a ``hipvg.functional.PackPlan`` (packed rows: ragged sequences laid end to end, ``te`` per real sequence)."""
        if te is None:
            te = self.time_emb(self.act(temb)) if self.has_time else None
        return HF.conv_block(x2, te, cond2 if self.has_cond else None, self.conv1.weight, self.conv1.bias,
                             self.norm.weight, self.norm.bias, self.conv2.weight, self.conv2.bias,
                             self.conv3.weight, self.conv3.bias, T=T, taps=self.taps, shift=self.shift,
                             eps=self.norm.eps, act=hip_act_id(self.act))

    def forward(self, x: torch.Tensor, cond: Optional[torch.Tensor] = None,
                temb: Optional[torch.Tensor] = None) -> torch.Tensor:
        h = self.conv1(x)
        if self.has_time:
            h = h + self.time_emb(self.act(temb)).unsqueeze(-1)
        h = self.norm(h)
        if self.has_cond:
            h = torch.cat([h, cond.to(h.dtype)], 1)
        return x + self.conv3(self.act(self.conv2(h)))


class BottleNeckResNet(nn.Module):
    def __init__(self, hp: Hparams, input_dim: Optional[int] = None,
                 output_dim: Optional[int] = None) -> None:
        super().__init__()
        hp.check_arg_in_hparams("num_layers", "layer", "init_channel", "out_channels",
                                "hidden_channels", "resample_rates", "resample_ksize")
        self.hp = hp
        L = hp.num_layers
        for name in ("resample_rates", "resample_ksize", "out_channels", "hidden_channels"):
            assert len(getattr(hp, name)) == L
        if any(r not in (1, -1) for r in hp.resample_rates):
            raise NotImplementedError("resampling BottleNeckResNet layers are not used by vae-gslm.yaml")
        boundary = hp.upward_layer.boundary if hp.has("upward_layer") else L
        assert boundary <= L
        self.conditional: List[bool] = list(hp.get("conditional", [False] * L))
        cond_dim = 0
        if hp.has("conditional"):
            hp.check_arg_in_hparams("condition_dim")
            cond_dim = hp.condition_dim
        self.time_dim = hp.get("time_dim", None)
        self.skip_connection = list(hp.get("skip_connection", [None] * L))
        self.skip_concat = hp.get("connection_type", None) == "concat"
        widths = [hp.init_channel] + list(hp.out_channels)
        blocks, skips = [], []
        for i in range(L):
            lhp = hp.layer if i < boundary else hp.upward_layer
            assert widths[i] == widths[i + 1]
            if self.conditional[i] and lhp.get("condition_type", "film") != "concat":
                raise NotImplementedError("only condition_type='concat' is used by vae-gslm.yaml")
            blocks.append(BottleneckBlock(widths[i], hp.hidden_channels[i], lhp,
                                          cond_dim=cond_dim if self.conditional[i] else 0,
                                          time_dim=self.time_dim))
            fuse = self.skip_connection[i] is not None and self.skip_concat
            skips.append(nn.Conv1d(2 * widths[i], widths[i], 1) if fuse else nn.Identity())
        self.layers = nn.ModuleList(blocks)
        self.skip_conv = nn.ModuleList(skips)
        self.linear = nn.Linear(input_dim, hp.init_channel) if input_dim is not None else None
        self.out_linear = nn.Linear(widths[-1], output_dim) if output_dim is not None else None
        self.final_norm = get_norm_fn(widths[-1], hp.layer.norm) if hp.get("final_norm", False) else None
        self.first_norm = get_norm_fn(widths[0], hp.layer.norm) if hp.get("first_norm", False) else None

    def packable(self) -> bool:
        """Can this stack run on packed rows (hipvg.functional.PackPlan with a halo)?  Every block on the run kernels
        (bf16, 512 channels, 7 taps), per-frame norms only."""
        norms_ok = all(n is None or isinstance(n, InstanceNorm) for n in (self.final_norm, self.first_norm))
        return (norms_ok and hipvg.compute_dtype() == torch.bfloat16
                and all(b.hip_ready() and b.taps == 7 and b.conv1.in_channels == 512 for b in self.layers))

    def lookahead_frames(self) -> int:
        """Frames after a sequence's end that its valid frames depend on (the halo a packed layout must carry)."""
        return sum(b.taps - 1 - b.shift for b in self.layers)

    def _hip_ok(self, x: TensorMask) -> bool:
        norms_ok = all(n is None or isinstance(n, InstanceNorm) for n in (self.final_norm, self.first_norm))
        return (x.value.is_cuda and os.environ.get("VG_CONV_STOCK", "0") != "1" and norms_ok
                and all(b.hip_ready() for b in self.layers))

    def time_projections(self, temb: Optional[torch.Tensor]) -> dict:
        """Time-embedding projections of all blocks as ONE Linear: same SiLU(t) for every block, the six (256 -> 512)
        weights concatenated along N (6 x ~14 tiny launches forward + backward become ~17).  {block index: [B, C]}."""
        tes = {}
        timed = [i for i, b in enumerate(self.layers) if b.has_time] if (temb is not None and self.time_dim is not None) else []
        if (len(timed) > 1 and len({type(self.layers[i].act) for i in timed}) == 1
                and os.environ.get("VG_BATCH_TEMB", "1") != "0"):
            a = self.layers[timed[0]].act(temb)
            W = torch.cat([self.layers[i].time_emb.weight for i in timed], 0)
            bvec = torch.cat([self.layers[i].time_emb.bias for i in timed], 0)
            sizes = [self.layers[i].time_emb.out_features for i in timed]
            proj = HF.small_linear(a, W, bvec) if (a.is_cuda and os.environ.get("VG_SMALL_LINEAR", "1") != "0") else F.linear(a, W, bvec)
            for i, piece in zip(timed, proj.split(sizes, dim=1)):
                tes[i] = piece
        return tes

    def forward_hip(self, x: TensorMask, c: Optional[TensorMask], t: Optional[torch.Tensor], tes: Optional[dict] = None) -> TensorMask:
        """Same computation on [B*T, C] rows: 1x1 convs on the MFMA GEMM, fused depthwise-conv+norm
        row kernels, skip connections as two accumulating GEMMs (no concat, no transposes)."""
        mask, lens = x.mask, x.lengths32
        B, T = mask.shape
        # packed rows (LVTR.forward's packed step): the batch is a pseudo batch of B = rows one-frame sequences whose
        # mask says which rows hold a frame; the time structure is the plan's
        plan = getattr(mask, "_vg_plan", None)
        Tseq = plan if plan is not None else T
        dt = hipvg.compute_dtype()
        h = x.value.reshape(B * T, -1)
        if self.linear is not None:
            h = dense_2d(h, self.linear.weight, self.linear.bias, lengths=lens, T=T)
        h = h.to(dt).contiguous()
        if self.first_norm is not None:
            h = channel_norm_rows(h, self.first_norm, T)
        cond2 = None if c is None else c.value.reshape(B * T, -1).to(dt).contiguous()
        temb = None if t is None else t.float()
        history = [h]
        if tes is None:
            tes = self.time_projections(temb)
        for i, block in enumerate(self.layers):
            h = block.forward_rows(h, Tseq, cond2 if self.conditional[i] else None,
                                   temb if self.time_dim is not None else None, tes.get(i))
            src = self.skip_connection[i]
            if src is not None:
                if self.skip_concat:
                    W = self.skip_conv[i].weight
                    Cw = h.shape[1]
                    # the k = 1 convolution over cat([h, history[src]]) as two products on column slices of its weight
                    part = HF.slice_linear(h, W, 0, self.skip_conv[i].bias)
                    h = HF.slice_linear(history[src], W, Cw, None, residual=part)
                else:
                    h = h + history[src]
            history.append(h)
        if self.final_norm is not None:
            h = channel_norm_rows(h, self.final_norm, T)
        if self.out_linear is not None:
            # (its epilogue's row predicate already zeroes the padded frames: no second mask pass)
            h = dense_2d(h, self.out_linear.weight, self.out_linear.bias, out_f32=True, lengths=lens, T=T)
            return TensorMask(h.view(B, T, -1), mask)
        return TensorMask(h.view(B, T, -1), mask).apply_mask()

    def forward(self, x: TensorMask, c: Optional[TensorMask] = None,
                t: Optional[torch.Tensor] = None, tes: Optional[dict] = None) -> TensorMask:
        """x: (B, T, C) TensorMask; c: (B, T, Cc) TensorMask; t: (B, time_dim); tes: ``time_projections(t)`` if the caller
        already has them (computed on the step's side branch)."""
        if self._hip_ok(x):
            return self.forward_hip(x, c, t, tes)
        mask = x.mask
        h = x.value
        if self.linear is not None:
            h = TensorMask(self.linear(h), mask).apply_mask().value
        h = h.transpose(1, 2)
        if self.first_norm is not None:
            h = self.first_norm(h)
        cond = None if c is None else c.value.transpose(1, 2)
        history = [h]
        for i, block in enumerate(self.layers):
            h = block(h, cond if self.conditional[i] else None, t if self.time_dim is not None else None)
            src = self.skip_connection[i]
            if src is not None:
                if self.skip_concat:
                    h = self.skip_conv[i](torch.cat([h, history[src].to(h.dtype)], 1))
                else:
                    h = h + history[src]
            history.append(h)
        if self.final_norm is not None:
            h = self.final_norm(h)
        h = h.transpose(1, 2)
        if self.out_linear is not None:
            h = self.out_linear(h)
        return TensorMask(h, mask).apply_mask()

    @property
    def sample_ratio(self) -> float:
        return 1.0


class ConvNormAct(nn.Module):
    """Strided conv -> per-frame channel norm -> activation on (B, C, T)."""

    def __init__(self, cin: int, cout: int, kernel: int, rate: int, lhp: Hparams):
        super().__init__()
        if rate >= 1 and rate != 1:
            raise NotImplementedError("up-sampling CNNStack layers are not used by vae-gslm.yaml")
        self.factor = 1 if rate == 1 else -rate
        self.conv = Conv1d(cin, cout, kernel_size=kernel, stride=self.factor,
                           padding=get_padding(kernel, causal=lhp.get("causal_padding", False),
                                               future=lhp.get("future_padding", False)))
        self.norm = get_norm_fn(cout, lhp.norm)
        self.act = get_activation(lhp.activation)

    def forward_rows(self, h3: torch.Tensor, length: torch.Tensor):
        """Channels-last HIP path on (B, T, C): the strided convolution is a window gather followed by
        one MFMA GEMM ([B*T_out, k*C] x [k*C, C_out]), then the channel-norm row kernel."""
        B, T, C = h3.shape
        k, stride = self.conv.kernel_size[0], self.conv.stride[0]
        pl, pr = self.conv.two_side_padding if self.conv.two_side_padding is not None else (self.conv.padding[0],) * 2
        w2 = self.conv.weight.permute(0, 2, 1).reshape(self.conv.out_channels, k * C)
        if h3.is_cuda and os.environ.get("VG_CONV_GATHER", "1") != "0":
            # window gather (padding, unfold, tap-major layout, dtype) as one HIP kernel; its adjoint in backward
            rows = HF.conv_gather(h3.to(hipvg.compute_dtype()), k, stride, pl, pr)
            t_out = rows.shape[0] // B
        else:
            win = F.pad(h3, (0, 0, pl, pr)).unfold(1, k, stride)            # (B, T_out, C, k) view
            t_out = win.shape[1]
            rows = win.permute(0, 1, 3, 2).reshape(B * t_out, k * C).to(hipvg.compute_dtype()).contiguous()
        y = HF.linear(rows, w2, self.conv.bias)
        width = 64 * (8 if y.dtype == torch.bfloat16 else 4)
        wide = y.shape[1] % width == 0 and y.shape[1] // width <= 2
        if (not wide and isinstance(self.act, nn.ReLU) and y.is_cuda and y.shape[1] <= 1024
                and os.environ.get("VG_NARROW_NORM", "1") != "0"):
            y = HF.narrow_channel_norm(y, self.norm.weight, self.norm.bias, eps=self.norm.eps, relu=True)
        else:
            y = self.act(channel_norm_rows(y, self.norm, t_out))
        y = y.view(B, t_out, -1)
        if self.factor != 1 and length is not None:      # None: every sequence is full and stays full (see CNNStack.forward)
            length = torch.clamp(TensorMask.resize_length(length, float(self.factor)), max=t_out)
        return y, length

    def forward(self, h: torch.Tensor, length: torch.Tensor):
        h = self.act(self.norm(self.conv(h)))
        if self.factor != 1:
            # the reference multiplies (not divides) the lengths of down-sampling
            # layers (modules/conv/layers.py:568,588-591); reproduced for parity
            length = TensorMask.resize_length(length, float(self.factor))
            length = torch.clamp(length, max=h.shape[-1])
        return h, length


class CNNStack(nn.Module):
    def __init__(self, hp: Hparams, input_dim: Optional[int] = None,
                 output_dim: Optional[int] = None) -> None:
        super().__init__()
        hp.check_arg_in_hparams("num_layers", "layer", "init_channel", "out_channels",
                                "resample_rates", "resample_ksize")
        self.hp = hp
        widths = [hp.init_channel] + list(hp.out_channels)
        assert len(hp.resample_rates) == len(hp.resample_ksize) == len(hp.out_channels) == hp.num_layers
        self.layers = nn.ModuleList([
            ConvNormAct(widths[i], widths[i + 1], hp.resample_ksize[i], hp.resample_rates[i], hp.layer)
            for i in range(hp.num_layers)])
        self.linear = nn.Linear(input_dim, hp.init_channel) if input_dim is not None else None
        self.out_linear = nn.Linear(widths[-1], output_dim) if output_dim is not None else None

    def forward(self, x: TensorMask) -> TensorMask:
        h = x.value
        hip = (h.is_cuda and os.environ.get("VG_CONV_STOCK", "0") != "1"
               and all(isinstance(l.norm, InstanceNorm) for l in self.layers))
        if hip:
            B, T = x.mask.shape
            if self.linear is not None:
                h = dense_2d(h, self.linear.weight, self.linear.bias, lengths=x.lengths32, T=T)
            # An input without padding (the fixed-length utterance crops) stays without padding: a down-sampling layer
            # MULTIPLIES the lengths by its factor (reference modules/conv/layers.py:568,588-591) and clamps them to the new
            # frame count, so full lengths map to full lengths at every level -- no length arithmetic (six tiny launches
            # per layer), no mask build and no re-mask of the output
            full = getattr(x.mask, "_vg_full", False) and all(l.factor >= 1 for l in self.layers)
            length = None if full else x.length
            for layer in self.layers:
                h, length = layer.forward_rows(h, length)
            if self.out_linear is not None:
                h = dense_2d(h, self.out_linear.weight, self.out_linear.bias, out_f32=True)
            if full:
                return TensorMask(h.float())
            return TensorMask.fromlength(h.float(), length).apply_mask()
        if self.linear is not None:
            h = TensorMask(self.linear(h), x.mask).apply_mask().value
        h, length = h.transpose(1, 2), x.length
        for layer in self.layers:
            h, length = layer(h, length)
        h = h.transpose(1, 2)
        if self.out_linear is not None:
            h = self.out_linear(h)
        return TensorMask.fromlength(h, length).apply_mask()

    @property
    def sample_ratio(self) -> float:
        r = 1.0
        for rate in self.hp.resample_rates:
            r = r * rate if rate > 0 else r / -rate
        return r
