"""Autoregressive decode session for ``LVTR`` on the MI355X.

``LVTR.step`` (reference models/speech/lvtr.py:227-286) keeps the reference's call contract: it returns
the grown key/value tensors and the caller (``ARTRSampler``, trainers/speech/sampler.py:17-72) feeds them
back, which costs a ``torch.cat`` of every layer's cache and ~150 small launches from Python per frame.
This session produces the same frames from a different machine model:

* the key/value caches are allocated once, ``[layers][B, Tmax, d]`` in the compute dtype; the attention
  kernel of the step writes the new row at ``pos[b]`` itself (``vg_attn_decode_append``);
* the prompt is consumed by ONE pass of the training-path kernels (flash attention over the whole prompt),
  whose per-layer key/value projections are copied into the caches;
* every Linear of the step is ``vg_gemm_rows`` (HBM-bound: the step streams the 403 MB of bf16 weights
  once: up to 16 sequences as groups of 8 rows on the exact-fp32 dot-product kernel, 17 to 64 -- the reference's
  inference batch, configs/infer/speech/vae-gslm.yaml:27 -- on the matrix-core rows kernel), norms / flow reverse
  are the row kernels of the training path;
  from 4 to 16 sequences the attention sub-layer of a layer is ONE launch on an fp32 residual stream
  (``vg_attn_layer_decode``: RMSNorm, QKV rows of a head, cache append, attention, out-projection band);
* the step's random draws (Gaussian latent noise, the uniform number of the token draw) come from one launch of a
  counter-based generator keyed by (session seed, sequence, device-side frame counter) (``vg_decode_noise``);
* the whole step -- embedding of the previous output, 16 layers, prior head, reverse flow, token
  soft-max and draw, write-back of the new frame into the step's own input buffer, ``pos += 1`` -- is
  captured once into a hipGraph and replayed per frame (58 graph nodes at the full config).

Only the token + flow model of ``vae-gslm.yaml`` is covered (the same scope as ``LVTR.forward``).
"""
from __future__ import annotations

import os
from typing import Optional

import torch

import hipvg
from hipvg import functional as HF
from utils.tensormask import TensorMask


class DecodeSession:
    def __init__(self, model, batch_size: int, max_frames: int, *, temperature: float = 1.0,
                 token_temperature: float = 1.0, use_graph: bool = True, device=None, keep_latent: bool = False):
        if not model.use_tokens or model.transformer_flow is None:
            raise NotImplementedError("DecodeSession implements the token + flow model of vae-gslm.yaml")
        self.model = model
        self.B, self.Tmax = int(batch_size), int(max_frames)
        self.temperature, self.token_temperature = float(temperature), float(token_temperature)
        self.use_graph = bool(use_graph)
        self.keep_latent = bool(keep_latent)
        stack = model.transformer[0]
        self.stack = stack
        self.L = len(stack.layers)
        self.D = stack.hp.layer.dim
        self.H = stack.layers[0].self_attn.nheads
        dev = device if device is not None else next(model.parameters()).device
        self.dev = torch.device(dev)
        self.dt = hipvg.compute_dtype()
        if self.B > 64:
            raise NotImplementedError("vg_gemm_rows handles up to 64 sequences per session")
        head = model.transformer[1]
        if not head.plain:
            raise NotImplementedError("DecodeSession needs the plain (mean, logstd) prior head")
        if stack.first_norm is not None or stack.out is not None or stack.final_norm is None:
            raise NotImplementedError("DecodeSession: stack layout differs from vae-gslm.yaml")
        self.kc = [torch.zeros(self.B, self.Tmax, self.D, dtype=self.dt, device=self.dev) for _ in range(self.L)]
        self.vc = [torch.zeros(self.B, self.Tmax, self.D, dtype=self.dt, device=self.dev) for _ in range(self.L)]
        self.pos = torch.zeros(self.B, dtype=torch.int32, device=self.dev)
        self.frame = torch.zeros(self.B, 1, 1 + model.hp.latent_dim, dtype=torch.float32, device=self.dev)
        from modules.attention.attention import _slopes_from
        self.slopes = _slopes_from((stack.rpe_id, stack.rpe), None, self.dev).slopes
        # per-session constants of the heads (concatenated projections, packed flow parameters)
        flow = model.transformer_flow
        if not flow._hip_ok(self.frame, c=True):
            raise NotImplementedError("DecodeSession needs the vae-gslm.yaml coupling stack (vg_flow_reverse)")
        with torch.no_grad():
            # prior head (mean | logstd) and the FiLM projections of the 4 coupling layers as ONE product
            self._heads_w = torch.cat([head.mean.weight, head.logstd.weight] +
                                      [l.film.linear.weight for l in flow.layers], 0).to(self.dt).contiguous()
            self._heads_b = torch.cat([head.mean.bias, head.logstd.bias] +
                                      [l.film.linear.bias for l in flow.layers], 0).float().contiguous()
            self._flow_params = []
            for l in flow.layers:
                self._flow_params += [l.linear1.weight, l.linear1.bias, l.norm.weight, l.norm.bias,
                                      l.linear2.weight, l.linear2.bias]
            self._flow_packed = HF.pack_flow_params(self._flow_params)
            hi, lo = flow.layers[0].scale_range
            self._flow_kw = dict(eps=flow.layers[0].norm.eps, hi=float(hi), lo=float(lo))
        # fused layer path (round 3): fp32 residual stream, the attention sub-layer as ONE launch per layer
        # (vg_attn_layer_decode: three graph nodes become one) that accumulates into one of two zero-initialised buffers.
        # A (head, sequence) block then pulls 0.5 MB of weights through one CU's 134 GB/s instead of 384 blocks sharing
        # them, so the node it saves is worth it from 4 sequences up (measured per frame, fused / five launches per layer:
        # B = 1 0.675 / 0.663 ms, 4 0.663 / 0.673, 8 0.663 / 0.684, 16 0.93 / 1.16).  VG_DECODE_FUSED=1 / 0 forces it.
        mode = os.environ.get("VG_DECODE_FUSED", "auto")
        fits = self.D % 256 == 0 and self.D <= 1024 and self.D == 64 * self.H
        # ... and up to 16: the fused launch re-reads a head's weights once per SEQUENCE (one block per (head, sequence)),
        # which is what the batch of the reference's inference config (64) cannot afford; from 17 sequences on every
        # Linear is one launch of the matrix-core rows kernel (weights streamed once) around the cache attention
        # (round 5, later: in bf16 the five-launch form with the split products below beats the fused launch at EVERY batch
        # -- B = 1: 0.643 -> 0.582 ms per frame, 8: 0.646 -> 0.605, 16: 0.668 -> 0.648 -- so `auto` fuses fp32 sessions only)
        self._fused = fits and (mode == "1" or (mode != "0" and 4 <= self.B <= 16 and self.dt != torch.bfloat16))
        # bf16 sessions (round 5; first from 17 sequences up, then at every batch): an fp32 residual stream with the two N = d_model products of a layer split
        # over K across blocks (vg_gemm_rows_acc: fp32 atomics into a buffer an earlier launch of the layer cleared);
        # VG_DECODE_ACC=<splits> (0 = off: five launches on a bf16 stream, every Linear one block per 16 columns)
        acc_min_b = int(os.environ.get("VG_DECODE_ACC_MINB", "1"))        # 17: rounds 1-5a (the split form from 17 sequences up only)
        self._acc = 0 if (self._fused or self.B < acc_min_b or self.dt != torch.bfloat16) else int(os.environ.get("VG_DECODE_ACC", "4"))
        # seed of this session's draws (vg_decode_noise), taken from torch's CPU generator: torch.manual_seed reproduces the
        # DRAWS of a run.  The frames are bitwise repeatable only in the reproducible mode VG_DECODE_ACC=0 (the
        # split products add their K slices with fp32 atomics in a run-dependent order; tests/test_parity_round6_gpu.py
        # checks both statements)
        self._seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        # draw epoch: a device word the captured graph READS (the seed itself is baked into the graph by value); every
        # prefill() bumps it, so a second generation on this session -- the frame counter starts over -- does not replay
        # the first one's Gaussian / uniform stream
        self._epoch = torch.zeros(1, dtype=torch.int32, device=self.dev)
        self._x1 = [torch.zeros(self.B, self.D, dtype=torch.float32, device=self.dev) for _ in range(2)]
        # VG_DECODE_PREFETCH=<blocks> (lab): while layer l computes, `blocks` narrow workgroups on a
        # second stream read layer l + 1's 26 MB of weights, so that the latency-bound kernels of the small-batch step
        # find them in the Infinity Cache.  Measured and not kept as a default: profiles/r05/README.md.
        self._prefetch = int(os.environ.get("VG_DECODE_PREFETCH", "0"))
        self._side = torch.cuda.Stream(device=self.dev) if self._prefetch > 0 else None
        self._graph = None
        self._last = {}

    # ------------------------------------------------------------------ weights in the compute dtype
    def _w(self, weight):
        return HF.shadow(weight, self.dt)

    # ------------------------------------------------------------------ prompt
    @torch.no_grad()
    def prefill(self, prior: torch.Tensor, init_state: Optional[torch.Tensor] = None,
                noise: Optional[torch.Tensor] = None) -> torch.Tensor:
        """prior: (B, Tp, 1 + latent) -- token ids (as float) and posterior latents of the prompt.
        Runs the first ``LVTR.step`` of the reference sampler (``push_init_state=True``), fills the
        caches with its Tp + 1 frames and returns the first sampled frame (B, 1, 1 + latent)."""
        B, Tp = prior.shape[:2]
        assert B == self.B and Tp + 1 <= self.Tmax
        out = self.model.step(prior, push_init_state=True, temperature=self.temperature,
                              token_temperature=self.token_temperature, init_state=init_state, noise=noise)
        for l, kv in enumerate(out["kv"]):
            self.kc[l][:, :Tp + 1].copy_(kv["key"])
            self.vc[l][:, :Tp + 1].copy_(kv["value"])
        self.pos.fill_(Tp + 1)
        self._epoch += 1
        first = out["output"][:, -1:].float()
        self.frame.copy_(first)
        self._last = {"transformer_latent": out["transformer_latent"].value[:, -1:], "logits": out["logits"][:, -1:],
                      "prior": out["prior"]}
        return first.clone()

    # ------------------------------------------------------------------ one frame
    def _step_body(self, noise: Optional[torch.Tensor] = None, uniform: Optional[torch.Tensor] = None) -> None:
        """~90 launches: frame embedding, 16 x (qkv, attention+cache append, out-proj, FFN-in, FFN-out) with the
        RMSNorms folded into the following projection, heads, Gaussian draw + reverse flow, token draw."""
        m, dt, B = self.model, self.dt, self.B
        lat_dim = m.hp.latent_dim
        fuser = m.token_fuser.linear
        frame2d = self.frame.view(B, -1)
        x = HF.embed_fuse(frame2d, m.token_embedding.weight.detach(), fuser.weight.detach(),
                          None if fuser.bias is None else fuser.bias.detach(), dt)
        st = self.stack
        fused = self._fused
        acc = self._acc
        if st.linear is not None:
            x = HF.rows_linear(x, self._w(st.linear.weight), st.linear.bias, out_f32=fused or acc > 0)
        elif fused or acc > 0:
            x = x.float()
        if fused and self.L % 2 == 1:          # layer l accumulates into _x1[l % 2] and clears the other one
            self._x1[0].zero_()
        main = torch.cuda.current_stream()
        for l, layer in enumerate(st.layers):
            att = layer.self_attn
            if self._prefetch > 0 and l + 1 < self.L:
                nxt = st.layers[l + 1]
                self._side.wait_stream(main)             # fork: the branch starts when layer l does
                with torch.cuda.stream(self._side):
                    for wgt in (nxt.self_attn.in_proj.weight, nxt.self_attn.out_proj.weight, nxt.linear1.weight,
                                nxt.linear2.weight):
                        sh = self._w(wgt)
                        hipvg.check(hipvg.lib().vg_touch(hipvg.ptr(sh), sh.numel() * sh.element_size(), self._prefetch,
                                                         torch.cuda.current_stream().cuda_stream), "vg_touch")
            if fused:
                # fp32 residual stream; the attention sub-layer is ONE launch that adds into x1 (zero on entry)
                x1 = HF.attention_layer_decode(x, layer.norm1.scale.detach(), layer.norm1.eps,
                                               self._w(att.in_proj.weight), att.in_proj.bias,
                                               self._w(att.out_proj.weight), att.out_proj.bias, self.kc[l], self.vc[l],
                                               self.slopes, self.pos, self.H, self._x1[l % 2], zero=self._x1[(l + 1) % 2])
                mid = HF.rows_linear_mixed(x1, self._w(layer.linear1.weight), layer.linear1.bias, act=hipvg.ACT_GELU,
                                           norm_scale=layer.norm3.scale.detach(), norm_eps=layer.norm3.eps, out_f32=True)
                x = HF.rows_linear_mixed(mid, self._w(layer.linear2.weight), layer.linear2.bias, residual=x1, out_f32=True)
                continue
            if acc > 0:
                # fp32 residual stream; buffers: xa receives x1 (cleared by the QKV launch), xb the layer's output
                # (cleared by the FFN-in launch); the next layer reads xb while its QKV launch clears xa again
                xa, xb = self._x1
                qkv = HF.rows_linear_mixed(x, self._w(att.in_proj.weight), att.in_proj.bias,
                                           norm_scale=layer.norm1.scale.detach(), norm_eps=layer.norm1.eps, zero=xa)
                ctx = HF.attention_decode_append(qkv, self.kc[l], self.vc[l], self.slopes, self.pos, self.H)
                x1 = HF.rows_linear_acc(ctx, self._w(att.out_proj.weight), att.out_proj.bias, x, xa, splits=acc)
                # (from the second layer on x IS xb: the QKV and out-projection launches above have read it before the
                # FFN-in launch below clears it -- launches of one stream run in order -- and the FFN-out launch then
                # accumulates the layer's output into it)
                mid = HF.rows_linear_mixed(x1, self._w(layer.linear1.weight), layer.linear1.bias, act=hipvg.ACT_GELU,
                                           norm_scale=layer.norm3.scale.detach(), norm_eps=layer.norm3.eps, zero=xb)
                x = HF.rows_linear_acc(mid, self._w(layer.linear2.weight), layer.linear2.bias, x1, xb, splits=acc)
                continue
            qkv = HF.rows_linear(x, self._w(att.in_proj.weight), att.in_proj.bias,
                                 norm_scale=layer.norm1.scale.detach(), norm_eps=layer.norm1.eps)
            ctx = HF.attention_decode_append(qkv, self.kc[l], self.vc[l], self.slopes, self.pos, self.H)
            x1 = HF.rows_linear(ctx, self._w(att.out_proj.weight), att.out_proj.bias, residual=x)
            mid = HF.rows_linear(x1, self._w(layer.linear1.weight), layer.linear1.bias, act=hipvg.ACT_GELU,
                                 norm_scale=layer.norm3.scale.detach(), norm_eps=layer.norm3.eps)
            x = HF.rows_linear(mid, self._w(layer.linear2.weight), layer.linear2.bias, residual=x1)
        if self._prefetch > 0:
            main.wait_stream(self._side)                 # join (also inside a captured graph)
        fn = dict(norm_scale=st.final_norm.scale.detach(), norm_eps=st.final_norm.eps)
        qs, ts, tp = m.q_spliter.linear, m.token_spliter.linear, m.token_predictor.linear
        head_linear = HF.rows_linear_mixed if (fused or acc > 0) else HF.rows_linear
        cond = head_linear(x, self._w(qs.weight), qs.bias, act=hipvg.ACT_RELU, **fn)
        heads = HF.rows_linear(cond, self._heads_w, self._heads_b, out_f32=True)    # (B, 2 latent + L*128): prior | FiLM
        mu_ls, wb = heads[:, :2 * lat_dim], heads[:, 2 * lat_dim:]
        # the step's random draws: one launch keyed by (session seed, sequence, frame counter) instead of torch.randn +
        # torch.rand (under graph replay those cost two fill launches for the generator state and one launch each)
        drawn = None
        if noise is None or uniform is None:
            drawn = HF.decode_noise(self._seed, self.pos, lat_dim, self._epoch)
        eps = noise if noise is not None else drawn[0]
        HF.coupling_flow_reverse(eps.reshape(B, lat_dim), wb, self._flow_params, packed=self._flow_packed,
                                 mu_ls=mu_ls, temperature=self.temperature, out=frame2d[:, 1:], **self._flow_kw)
        hid = head_linear(x, self._w(ts.weight), ts.bias, act=hipvg.ACT_RELU, **fn)
        logits = HF.rows_linear(hid, self._w(tp.weight), tp.bias, out_f32=True)      # (B, vocab)
        u01 = uniform if uniform is not None else drawn[1]
        HF.sample_token(logits, self.token_temperature, u01, frame2d, self.pos)      # also pos += 1
        self._last = {"logits": logits.view(B, 1, -1), "mu_ls": heads[:, None, :2 * lat_dim], "hidden": x}
        if self.keep_latent:      # the normalised state itself is only needed by tests / callers that ask for it
            lat, _ = HF.rmsnorm_fwd_raw(x, st.final_norm.scale.detach().float(), st.final_norm.eps, None, 0)
            self._last["transformer_latent"] = lat.view(B, 1, -1)

    @torch.no_grad()
    def step(self, noise: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Consumes the frame produced by the previous call (or by :meth:`prefill`) and returns the next
        one, (B, 1, 1 + latent) -- a copy, or ``out`` (same shape) filled in place (``generate`` hands in the slice of its
        result: one copy per frame instead of two).  With ``noise`` (teacher-forced tests) the step runs eagerly."""
        def result():
            if out is None:
                return self.frame.clone()
            out.copy_(self.frame)
            return out
        if noise is not None or not self.use_graph:
            self._step_body(noise)
            return result()
        if self._graph is None:
            # lazy initialisations (weight casts, allocator) happen on the first, eager, frame; the
            # second frame is captured and every later one replays it
            self._step_body()
            first, eager_last = result(), self._last
            side = torch.cuda.Stream(device=self.dev)
            side.wait_stream(torch.cuda.current_stream())
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.stream(side):
                with torch.cuda.graph(graph, stream=side):
                    self._step_body()
            torch.cuda.current_stream().wait_stream(side)
            self._graph, self._graph_last = graph, self._last     # the capture's output buffers
            self._last = eager_last
            return first
        # The captured step is one chain of launches (no parallel branch: hip::Graph::UpdateStreams has nothing to assign) unless
        # the lab prefetch branch is on; then the launch-stream rule of hipvg.functional.graph_launch_stream applies.
        launch, cur = (HF.graph_launch_stream(self.dev) if self._side is not None else None), torch.cuda.current_stream(self.dev)
        if launch is None or launch == cur:
            self._graph.replay()
        else:
            launch.wait_stream(cur)
            with torch.cuda.stream(launch):
                self._graph.replay()
            cur.wait_stream(launch)
        self._last = self._graph_last
        return result()

    def force_frame(self, frame: torch.Tensor) -> None:
        """Overwrite the frame the next :meth:`step` consumes (teacher forcing)."""
        self.frame.copy_(frame.reshape(self.frame.shape))

    @torch.no_grad()
    def generate(self, length: int) -> torch.Tensor:
        """``length`` further frames after :meth:`prefill`'s first one -> (B, length, 1 + latent)."""
        assert int(self.pos.max()) + length <= self.Tmax, "cache too small for the requested continuation"
        out = torch.empty(self.B, length, self.frame.shape[-1], dtype=torch.float32, device=self.dev)
        for t in range(length):
            self.step(out=out[:, t:t + 1])
        return out
