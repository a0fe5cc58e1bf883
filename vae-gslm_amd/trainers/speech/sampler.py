"""Prompted autoregressive sampling (reference trainers/speech/sampler.py:17-72).

Same call contract as the reference ``ARTRSampler``: encode the prompt with the posterior, run
``length`` autoregressive frames, decode prompt + continuation with the diffusion decoder.  On the GPU
the frames come from :class:`inference.speech.session.DecodeSession` (pre-allocated KV cache, hipGraph
replay); the reference's ``model.step`` loop remains as the generic path (attention maps, truncated
normal sampling, more than 16 sequences)."""
from __future__ import annotations

from typing import Mapping, Optional, Tuple

import torch
import torch.nn as nn

from utils.tensormask import TensorMask


class ARTRSampler(object):
    def __init__(self, model: nn.Module, use_graph: bool = True):
        self.model = model
        self.use_graph = use_graph
        self.has_utterance = getattr(model, "utterance_encoder", None) is not None
        self.model_use_tokens = bool(getattr(model, "use_tokens", False))
        self.last_session = None

    def _fast_ok(self, prior: torch.Tensor, truncated_norm, return_attn) -> bool:
        return (prior.is_cuda and self.model_use_tokens and truncated_norm is None and not return_attn
                and prior.shape[0] <= 16 and getattr(self.model, "transformer_flow", None) is not None)

    @torch.no_grad()
    def __call__(self, length: int, prior: torch.Tensor, temperature: float = 1.0,
                 token_temperature: float = 1.0, truncated_norm: Optional[Tuple[float, float]] = None,
                 return_attn: bool = False, encoder_temperature: float = 1.0) -> Mapping:
        model = self.model
        u_c = model.encode_utterance(TensorMask(prior)) if self.has_utterance else None
        prior = model.encode(TensorMask(prior), temperature=encoder_temperature).value
        outputs = {"output": [prior]}
        if return_attn:
            outputs["attn"] = []
        if self._fast_ok(prior, truncated_norm, return_attn):
            from inference.speech.session import DecodeSession
            sess = DecodeSession(model, prior.shape[0], prior.shape[1] + 1 + length, temperature=temperature,
                                 token_temperature=token_temperature, use_graph=self.use_graph)
            outputs["output"].append(sess.prefill(prior.float()))
            if length > 1:
                outputs["output"].append(sess.generate(length - 1))
            self.last_session = sess
        else:
            if self.model_use_tokens:
                state = prior
            else:
                state = torch.cat([model.initial_state(prior.shape[0], device=prior.device), prior], 1)
            iters = {"output": state, "kv": None}
            for i in range(length):
                iters = model.step(iters["output"], temperature=temperature, token_temperature=token_temperature,
                                   truncated_norm=truncated_norm, past_kv=iters["kv"], return_attn=return_attn,
                                   push_init_state=(i == 0 and self.model_use_tokens))
                if i == 0:
                    iters["output"] = iters["output"][:, -1:]
                outputs["output"].append(iters["output"])
                if return_attn:
                    outputs["attn"].append(iters["self_attn"])
        frames = TensorMask(torch.cat([o.float() for o in outputs["output"]], 1))
        outputs["frames"] = frames.value
        outputs["output"] = model.decode(frames, u_c=u_c) if self.has_utterance else model.decode(frames)
        return outputs
