"""Noise-prediction network of the diffusion decoder (reference
modules/diffusion/unet.py:10-27,67-93): sinusoidal time embedding -> 2-layer
MLP, a Linear that squeezes the frame condition, and the conditional
bottleneck ResNet.  Runs on stock PyTorch-ROCm ops (not part of the HIP list).
"""
import os

import torch
from torch import nn

from hparams.hp import Hparams
from modules.activations import get_activation
from modules.conv.layers import BottleNeckResNet
from modules.position.absolute import SinCos
from utils.tensormask import TensorMask


class TimeEmbedding(nn.Module):
    def __init__(self, hp: Hparams):
        super().__init__()
        hp.check_arg_in_hparams("activation", "maxpos", "dim")
        self.n_channels = hp.dim
        use_bias = hp.get("bias", True)
        self.lin1 = nn.Linear(hp.dim, hp.dim, bias=use_bias)
        self.act = get_activation(hp.activation)
        self.lin2 = nn.Linear(hp.dim, hp.dim, bias=use_bias)
        self.embedding = SinCos(hp.dim, maxpos=hp.maxpos)

    def forward(self, t: torch.Tensor) -> torch.Tensor:
        e = self.embedding.get(t)
        if e.is_cuda and e.dim() == 2 and os.environ.get("VG_SMALL_LINEAR", "1") != "0":
            # fp32 rows, one per sequence: the stock backward of these Linears is vendor-library launches of 15 - 40 µs each
            from hipvg import functional as HF
            return HF.small_linear(self.act(HF.small_linear(e, self.lin1.weight, self.lin1.bias)), self.lin2.weight, self.lin2.bias)
        return self.lin2(self.act(self.lin1(e)))


class ConditionalBottleNeckUNet(nn.Module):
    def __init__(self, cond_dim: int, noise_dim: int, hp: Hparams):
        super().__init__()
        hp.check_arg_in_hparams("unet", "time_embedding")
        hp.unet.check_arg_in_hparams("conditional")
        hp.unet.time_dim = hp.time_embedding.dim
        self.cond_net = nn.Linear(cond_dim, hp.unet.condition_dim)
        self.time_embedding = TimeEmbedding(hp.time_embedding)
        self.unet = BottleNeckResNet(hp.unet, input_dim=noise_dim, output_dim=noise_dim)

    def forward(self, noise: TensorMask, t: torch.Tensor, cond: TensorMask, temb=None, tes=None) -> TensorMask:
        """noise, cond: (B, T, C) TensorMasks; t: (B,) integer diffusion steps.  ``temb`` / ``tes``: the time embedding and
        the blocks' projections of it when the caller computed them ahead (LVTR.forward's side branch)."""
        if cond.value.is_cuda and cond.value.dim() == 3:
            # conditioning projection (cond_dim -> 32) through the HIP GEMM: the mask is its epilogue's row predicate
            from modules.linear.layers import dense_2d
            c = TensorMask(dense_2d(cond.value, self.cond_net.weight, self.cond_net.bias, lengths=cond.lengths32,
                                    T=cond.value.shape[1]), cond.mask)
        else:
            c = TensorMask(self.cond_net(cond.value), cond.mask).apply_mask()
        return self.unet(noise, c, self.time_embedding(t) if temb is None else temb, tes)
