"""Pre-LN Transformer layer and stack on the HIP kernels.

Drop-in for the reference ``modules/transformer/layers.py`` (``TransformerLayer``
:13-93, ``TransformerLayerStack`` :96-204): same constructors, parameter names
(``self_attn.*``, ``linear1/2``, ``norm1/3.scale``, ``final_norm.scale``,
``linear.weight``), ``forward`` / ``run`` signatures and result dicts.

One layer is six kernel launches forward:
    rmsnorm(+mask) -> QKV GEMM -> flash attention (causal + ALiBi in-kernel)
    -> out-proj GEMM (+residual, +mask) -> rmsnorm -> FFN GEMM pair
       (bias+GELU epilogue ; bias+residual+mask epilogue)
The activations stay 2-D ``[B*T, C]`` in the compute dtype between kernels;
``TensorMask`` objects are only created at the module boundary.
"""
from __future__ import annotations

import math
from typing import Any, List, Mapping, Optional, Tuple

import torch
import torch.nn as nn

import hipvg
from hipvg import functional as HF
from hparams.hp import Hparams
from modules.activations import get_activation, hip_act_id
from modules.attention.attention import AlibiBias, CrossAttention, SelfAttention, _slopes_from
from modules.norm import RMSNorm, get_norm_fn
from modules.position.embedding import get_positional_encoding
from utils.tensormask import TensorMask


class TransformerLayer(nn.Module):
    def __init__(self, hp: Hparams) -> None:
        super().__init__()
        hp.check_arg_in_hparams("ffd_size", "norm", "activation", "dim", "self_attn")
        self.hp = hp
        self.preln = hp.get("preln", True)
        if hp.get("dropout", 0.0) or hp.has("cross_attn") or not self.preln:
            raise NotImplementedError("HIP TransformerLayer: pre-LN, no dropout, no cross-attention")
        self.self_attn = SelfAttention(hp.dim, hp.self_attn)
        self.cross_attn = None
        use_bias = hp.get("bias", True)
        self.linear1 = nn.Linear(hp.dim, hp.ffd_size, bias=use_bias)
        self.linear2 = nn.Linear(hp.ffd_size, hp.dim, bias=use_bias)
        self.norm1 = get_norm_fn(hp.dim, hp.norm)
        self.norm3 = get_norm_fn(hp.dim, hp.norm)
        self.activation = get_activation(hp.activation)
        if not isinstance(self.norm1, RMSNorm) or hip_act_id(self.activation) != "gelu":
            raise NotImplementedError("HIP TransformerLayer implements RMSNorm + exact GELU "
                                      "(the vae-gslm.yaml layer)")

    # ---- the fused path used by training and by the stack: 2-D activations in, 2-D out
    def forward_2d(self, x2: torch.Tensor, B: int, T: int, lens: Optional[torch.Tensor],
                   slopes: torch.Tensor, pack=None) -> torch.Tensor:
        sa = self.self_attn
        return HF.transformer_layer(x2, self.norm1.scale, sa.in_proj.weight, sa.in_proj.bias,
                                    sa.out_proj.weight, sa.out_proj.bias, self.norm3.scale,
                                    self.linear1.weight, self.linear1.bias, self.linear2.weight,
                                    self.linear2.bias, slopes, lens, B, T, sa.nheads, self.norm1.eps, pack)

    def forward(self, tgt: TensorMask, memory: Optional[TensorMask] = None,
                rpe_pair: Optional[Tuple[str, Any]] = None, rpe_bias=None,
                past_kv: Optional[Mapping] = None, return_attn: bool = False,
                return_kv: bool = False) -> Mapping:
        B, T, D = tgt.value.shape
        dt = hipvg.compute_dtype()
        output = dict()
        x2 = tgt.value.reshape(B * T, D).to(dt).contiguous()
        lens = tgt.lengths32
        if past_kv is None and not return_attn and not return_kv:
            bias_h = _slopes_from(rpe_pair, rpe_bias, x2.device)
            if rpe_pair is not None and rpe_pair[0] == "ALiBi":
                output["rpe_bias"] = bias_h
            y = self.forward_2d(x2, B, T, lens, bias_h.slopes)
            output["output"] = TensorMask(y.view(B, T, D), tgt.mask)
            return output
        # decode / debug path: same kernels, through the SelfAttention module
        n1 = HF.rmsnorm(x2, self.norm1.scale, self.norm1.eps, lengths=lens, T=T)
        sa = self.self_attn(TensorMask(n1.view(B, T, D), tgt.mask), past_kv=past_kv, rpe_pair=rpe_pair,
                            rpe_bias=rpe_bias, return_attn=return_attn, return_kv=return_kv)
        if "rpe_bias" in sa:
            output["rpe_bias"] = sa["rpe_bias"]
        x1 = x2 + sa["output"].value.reshape(B * T, D)
        n3 = HF.rmsnorm(x1, self.norm3.scale, self.norm3.eps, lengths=lens, T=T)
        y = HF.ffn(n3, self.linear1.weight, self.linear1.bias, self.linear2.weight, self.linear2.bias,
                   residual=x1, lengths=lens, T=T)
        output["output"] = TensorMask(y.view(B, T, D), tgt.mask)
        if return_attn:
            output["self_attn"] = sa["attn"]
        if return_kv:
            output["kv"] = sa["kv"]
        return output


class TransformerLayerStack(nn.Module):
    def __init__(self, hp: Hparams, input_dim: Optional[int] = None,
                 output_dim: Optional[int] = None, memory_dim: Optional[int] = None) -> None:
        super().__init__()
        hp.check_arg_in_hparams("num_layers", "layer")
        self.hp = hp
        self.layers = nn.ModuleList([TransformerLayer(hp.layer) for _ in range(hp.num_layers)])
        use_bias = hp.get("bias", True)
        self.linear = nn.Linear(input_dim, hp.layer.dim, bias=use_bias) if input_dim is not None else None
        self.out = nn.Linear(hp.layer.dim, output_dim, bias=use_bias) if output_dim is not None else None
        self.memory_linear = None
        self.is_cross_attn = False
        self.final_norm = get_norm_fn(hp.layer.dim, hp.layer.norm) if hp.get("final_ln", True) else None
        self.first_norm = get_norm_fn(hp.layer.dim, hp.layer.norm) if hp.get("first_ln", False) else None
        # Packed rows (hip.packed_rows): None = padded rows; "auto" = pack whenever the batch has enough padding (reads the
        # lengths on the host: eager launches only); an int = pack into exactly that many rows (the trainer's hipGraph
        # path, which picks the bucket before it picks the graph).  Plans (static index buffers) are cached per shape.
        self.pack_rows = None
        self.pack_granule = 256
        self._pack_plans = {}
        self.rpe, self.rpe_id = None, None
        if hp.get("rpe", False):
            self.rpe_id = hp.rpe.identifier
            self.rpe = get_positional_encoding(self.rpe_id, hp.rpe, hp.layer.dim,
                                               hp.layer.self_attn.nheads)

    def _pack_plan(self, B, T, lens, x2):
        rows = self.pack_rows
        if rows is None or lens is None or self.first_norm is not None:
            return None
        M = B * T
        if rows == "auto":
            rows = HF.pack_rows_bucket(int(lens.sum().item()), self.pack_granule)
        else:
            # an explicit row count is the caller's promise that the batch fits it (the trainer's hipGraph path chooses
            # it from this batch and clears it afterwards).  Outside a capture the promise is cheap to check, and a
            # broken one would index out of bounds in the gathers (more valid frames than rows) or leave rows no
            # sequence covers (far fewer): fall back to padded rows instead.
            rows = int(rows)
            if not torch.cuda.is_current_stream_capturing():
                total = int(lens.sum().item())
                npseudo = -(-min(rows, self.pack_granule or rows) // T)
                if total > rows or rows - total > npseudo * T:
                    return None
        rows = int(rows)
        if rows > int(0.94 * M) or (x2.shape[1] * x2.element_size()) % 16:
            return None                    # not enough padding to pay for the two gathers
        key = (B, T, rows, x2.device)
        plan = self._pack_plans.get(key)
        if plan is None:
            plan = self._pack_plans[key] = HF.PackPlan(B, T, rows, x2.device, self.pack_granule)
        return plan.fill(lens)

    def _norm2d(self, norm, x2, lens, T, masked):
        return HF.rmsnorm(x2, norm.scale, norm.eps, lengths=lens if masked else None, T=T)

    def run(self, tgt: TensorMask, memory: Optional[TensorMask] = None,
            past_kv: Optional[List] = None, return_attn: bool = False,
            return_kv: bool = False) -> Mapping[str, Any]:
        B, T = tgt.value.shape[:2]
        D = self.hp.layer.dim
        dt = hipvg.compute_dtype()
        mask, lens = tgt.mask, tgt.lengths32
        outputs = {}
        if return_attn:
            outputs["self_attn"] = []
        if return_kv:
            outputs["kv"] = []
        x2 = tgt.value.reshape(B * T, -1).to(dt).contiguous()
        if self.linear is not None:
            x2 = HF.linear(x2, self.linear.weight, self.linear.bias, lengths=lens, T=T)
        if self.first_norm is not None:
            x2 = self._norm2d(self.first_norm, x2, lens, T, masked=True)
        fast = past_kv is None and not return_attn and not return_kv
        layer_outs = []
        # rows that arrive packed (LVTR.forward's packed step: a pseudo batch of one-frame sequences, the time structure
        # in the plan that rides on the mask): no gather here, no scatter at the end
        given = getattr(mask, "_vg_plan", None) if fast and self.first_norm is None else None
        plan = given if given is not None else (self._pack_plan(B, T, lens, x2) if fast else None)
        Bl, Tl = (plan.B, plan.T) if given is not None else (B, T)
        if fast:
            slopes = _slopes_from((self.rpe_id, self.rpe), None, x2.device).slopes
            cut = getattr(self, "grad_cut_layer", None)
            if plan is not None and given is None:
                # the stack runs on the valid frames only (the padded ones are zero rows that every layer re-masks in
                # the reference, utils/tensormask.py:63-67): gather once here, scatter back after the final norm
                x2 = HF.pack_rows(x2, plan)
            for l, layer in enumerate(self.layers):
                if cut is not None and l in cut and torch.is_grad_enabled() and x2.requires_grad:
                    # backward cut (trainers.speech.lvtr: segmented hipGraph replay): the tape stops at this leaf;
                    # the trainer later feeds its gradient into the tape of the layers below
                    leaf = x2.detach().requires_grad_(True)
                    self.grad_cuts.append(((x2,), (leaf,)))
                    x2 = leaf
                x2 = layer.forward_2d(x2, Bl, Tl, lens, slopes, plan)
                if plan is None:
                    layer_outs.append(TensorMask(x2.view(B, T, D), mask))
        else:
            past = past_kv if past_kv is not None else [None] * len(self.layers)
            rpe_pair, rpe_bias = (self.rpe_id, self.rpe), None
            cur = TensorMask(x2.view(B, T, D), mask)
            for layer, pkv in zip(self.layers, past):
                res = layer(cur, memory, rpe_pair=rpe_pair, rpe_bias=rpe_bias, past_kv=pkv,
                            return_attn=return_attn, return_kv=return_kv)
                if "rpe_bias" in res:
                    rpe_pair, rpe_bias = None, res["rpe_bias"]
                if return_attn:
                    outputs["self_attn"].append(res["self_attn"].detach())
                if return_kv:
                    outputs["kv"].append(res["kv"])
                cur = res["output"]
                layer_outs.append(cur)
            x2 = cur.value.reshape(B * T, D)
        if self.final_norm is not None:
            # the reference does not re-mask here (:186-189); zero rows stay zero under RMSNorm
            x2 = self._norm2d(self.final_norm, x2, lens, T, masked=plan is None)
            if plan is not None and given is None:
                x2 = HF.unpack_rows(x2, plan)
            layer_outs.append(TensorMask(x2.view(B, T, D), mask))
        elif plan is not None and given is None:
            x2 = HF.unpack_rows(x2, plan)
        if self.out is not None:
            x2 = HF.linear(x2, self.out.weight, self.out.bias, lengths=lens, T=T)
        outputs["output"] = TensorMask(x2.view(B, T, -1), mask)
        outputs["layers"] = layer_outs
        return outputs

    def forward(self, tgt: TensorMask, memory: Optional[TensorMask] = None) -> TensorMask:
        return self.run(tgt, memory=memory)["output"]

    def custom_weight_init(self, init_std: float):
        return None     # only T5RPE tables were initialised here in the reference (:201-204)
