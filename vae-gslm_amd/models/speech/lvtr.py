"""LVTR -- the VAE-GSLM model -- on the MI355X HIP hot path.

Drop-in for the reference ``models/speech/lvtr.py`` (``LVTR`` :18-395): same
constructor ``LVTR(hp, input_dim, memory_dim)``, sub-module names (hence the
same ``state_dict`` keys), and the same ``forward`` / ``step`` /
``decode`` / ``encode`` / ``encode_utterance`` / ``initial_state`` /
``likelihood`` / ``fuse_inputs`` / ``split_inputs`` surface.

Where the work goes in ``forward`` (one training micro-step):
  * Transformer stack, the q/token splitter + predictor heads, the prior
    Gaussian head, the flow's FiLM projections: MFMA GEMM / flash-attention /
    RMSNorm HIP kernels (bf16 storage, fp32 accumulation; or exact fp32);
  * posterior reparameterisation + entropy, prior log-density + KL sum, token
    cross-entropy (+arg-max): fused HIP row kernels, fp32;
  * conv posterior encoder, utterance encoder, diffusion UNet, the flow's tiny
    coupling nets: stock PyTorch-ROCm ops (autocast to bf16 in the fast mode).

Additions over the reference API (all optional): ``noise=`` lets a caller
inject the five random draws of a step (parity tests); the result dict also
carries the fused ``kld`` sum, the token ``logits`` arg-max and ``lengths``.
"""
from __future__ import annotations

import math
import os
from typing import List, Mapping, Optional, Tuple

import torch
import torch.nn as nn

import hipvg
from hipvg import functional as HF
from hparams.hp import Hparams
from modules.conv.layers import BottleNeckResNet, CNNStack
from modules.diffusion.ddpm import GaussianDiffusion1D
from modules.diffusion.unet import ConditionalBottleNeckUNet
from modules.flow.layers import CouplingStack
from modules.flow.utils import TensorLogdet
from modules.linear.layers import (Embedding, GaussianParameterize, Linear, TimeAggregation,
                                   dense_2d)
from modules.transformer.layers import TransformerLayerStack
from training_lib.losses import masked_ce_loss
from utils.tensormask import TensorMask

HALF_LOG_2PI = 0.5 * math.log(2 * math.pi)


def _side_autocast():
    """bf16 autocast for the stock-PyTorch (non-HIP) sub-networks in the fast mode."""
    fast = hipvg.compute_dtype() == torch.bfloat16
    return torch.autocast("cuda", dtype=torch.bfloat16, enabled=fast)


class LVTR(nn.Module):
    def __init__(self, hp: Hparams, input_dim: Optional[int] = None,
                 memory_dim: Optional[int] = None) -> None:
        super().__init__()
        hp.check_arg_in_hparams("encoder", "decoder", "transformer", "latent_dim")
        self.input_dim, self.hp = input_dim, hp
        enc_kind = hp.encoder.get("identifier", "ResNet")
        if enc_kind == "BottleNeckResNet":
            enc_cls = BottleNeckResNet
        elif enc_kind == "CNNStack":
            enc_cls = CNNStack
        elif enc_kind == "ResNet":
            raise NotImplementedError("the plain ResNet encoder is not used by vae-gslm.yaml")
        else:
            raise ValueError(f"{enc_kind} not recoginized.")
        latent = hp.latent_dim
        self.encoder = nn.Sequential(
            enc_cls(hp.encoder, input_dim=input_dim, output_dim=latent),
            GaussianParameterize(latent, latent,
                                 std=hp.encoder.get("fix_std", None),
                                 std_range=hp.encoder.get("std_range", None),
                                 truncated_norm=hp.encoder.get("truncated_norm", None),
                                 total_std=hp.encoder.get("total_std", None),
                                 use_tanh=False,
                                 normalization=hp.encoder.get("normalization", False)))
        width = hp.transformer.layer.dim
        self.tokens = hp.get("tokens", None)
        if self.tokens is not None:
            self.tokens.check_arg_in_hparams("embedding_dim", "vocab_size")
            self.token_embedding_dim = self.tokens.embedding_dim
            self.token_embedding = Embedding(self.tokens.vocab_size, self.tokens.embedding_dim)
            self.token_predictor = Linear(width, self.tokens.vocab_size)
            self.token_fuser = Linear(latent, self.tokens.embedding_dim, activation=nn.ReLU())
            self.token_spliter = Linear(width, width, activation=nn.ReLU())
            self.q_spliter = Linear(width, width, activation=nn.ReLU())
        else:
            self.q_spliter = nn.Identity()
        self.use_tokens = self.tokens is not None
        frame_dim = self.tokens.embedding_dim if self.use_tokens else latent
        cond_dim = frame_dim + (hp.utterance_encoder.embedding_dim if hp.has("utterance_encoder") else 0)
        dec_kind = hp.decoder.diffusion.get("identifier", "ConditionalUNet")
        if dec_kind != "ConditionalBottleNeckUNet":
            if dec_kind == "ConditionalUNet":
                raise NotImplementedError("ConditionalUNet decoder is not used by vae-gslm.yaml")
            raise ValueError(f"{dec_kind} not recoginized.")
        hp.decoder.check_arg_in_hparams("cond_unet")
        self.decoder = GaussianDiffusion1D(
            ConditionalBottleNeckUNet(cond_dim, input_dim, hp.decoder.cond_unet), hp.decoder.diffusion)
        self.diff_scaling = hp.decoder.diffusion.get("input_scale", 1.0)
        self.transformer_flow = None
        if hp.transformer.has("flow"):
            conditional = hp.transformer.flow.get("conditional", False)
            self.transformer_flow = CouplingStack(latent, hp.transformer.flow,
                                                  condition_dim=width if conditional else None)
        self.transformer = nn.Sequential(
            TransformerLayerStack(hp.transformer, input_dim=frame_dim, memory_dim=memory_dim),
            GaussianParameterize(width, latent, std=hp.transformer.get("fix_std", None),
                                 std_range=hp.transformer.get("std_range", None), use_tanh=False,
                                 mean=hp.transformer.get("fix_mean", None)))
        # Packed step (hip.packed_step; set per batch by the trainer): None = padded rows; "auto" = pack whenever the batch
        # has enough padding (reads the lengths on the host: eager launches only); an int = pack into exactly that many
        # rows (the trainer's hipGraph path picks the bucket before it picks the graph).  See _forward_packed.
        self.pack_rows = None
        self.pack_granule = 1024
        self.pack_fill = 0.94              # packed rows / padded rows above which a batch is not worth packing
        self.pack_return = "all"           # "scalars": the packed step returns its scalar losses / monitors only (the trainer)
        self._pack_plans = {}
        self.utterance_encoder = None
        if hp.has("utterance_encoder"):
            self.utterance_encoder = nn.Sequential(
                CNNStack(hp.utterance_encoder, input_dim=input_dim,
                         output_dim=hp.utterance_encoder.embedding_dim),
                TimeAggregation())

    @property
    def sample_ratio(self) -> float:
        return self.encoder[0].sample_ratio

    # ------------------------------------------------------------------ helpers
    def split_inputs(self, x: TensorMask):
        return x.split(1)

    def fuse_inputs(self, x: TensorMask, tokens: TensorMask) -> TensorMask:
        return tokens + self.token_fuser(x).value.float()

    def initial_state(self, bsize: int, device=None, nfeat=None):
        if nfeat is None:
            nfeat = self.token_embedding_dim if self.tokens is not None else self.hp.latent_dim
        return torch.rand(bsize, 1, nfeat, device=device) * 2.0 - 1.0

    def _embed(self, x: TensorMask):
        ids, feats = self.split_inputs(x)
        ids = TensorMask(ids.value.long().squeeze(-1), ids.mask)
        return ids, feats, self.token_embedding(ids)

    def _prior_stats(self, latent: TensorMask):
        """q_spliter -> fused (mean | logstd) fp32 projection of the prior head."""
        cond = self.q_spliter(latent)
        head = self.transformer[1]
        if head.plain:
            return cond, head.project(cond.value)
        g = head(cond)
        return cond, torch.cat([g.mean.value.float(), g.logstd.value.float()], -1)

    def _logits(self, latent: TensorMask) -> torch.Tensor:
        hid = self.token_spliter(latent)
        lin = self.token_predictor.linear
        return dense_2d(hid.value, lin.weight, lin.bias, out_f32=True)

    # ------------------------------------------------------------------ packed step
    def pack_halo(self) -> int:
        """Padded frames after a sequence's end that its valid frames depend on: the look-ahead taps of the diffusion
        UNet's upward blocks (the posterior encoder and the downward blocks are causal)."""
        nets = [self.encoder[0], self.decoder.model.unet]
        return max(n.lookahead_frames() if hasattr(n, "lookahead_frames") else 0 for n in nets)

    def packable(self) -> bool:
        nets = [self.encoder[0], self.decoder.model.unet]
        # (the row gathers move whole 16-byte pieces: fp32 mel frames and latent noise rows of a multiple of four values)
        return (self.use_tokens and self.transformer_flow is not None and hipvg.compute_dtype() == torch.bfloat16
                and (self.input_dim or 0) % 4 == 0 and self.hp.latent_dim % 4 == 0
                and all(hasattr(n, "packable") and n.packable() for n in nets)
                and self.transformer[0].first_norm is None and os.environ.get("VG_CONV_STOCK", "0") != "1")

    def _step_plan(self, B: int, T: int, lens, device):
        rows = self.pack_rows
        if rows is None or lens is None or not self.packable():
            return None
        halo = self.pack_halo()
        M = B * T
        if rows == "auto":
            rows = HF.pack_rows_bucket(int(torch.clamp(lens + halo, max=T).sum().item()), self.pack_granule)
        else:
            rows = int(rows)
            if not torch.cuda.is_current_stream_capturing():        # the caller's promise is cheap to check outside a capture
                total = int(torch.clamp(lens + halo, max=T).sum().item())
                npseudo = -(-min(rows, self.pack_granule or rows) // T)
                if total > rows or rows - total > npseudo * T:
                    return None
        if rows > int(self.pack_fill * M):
            return None
        key = (B, T, rows, device, halo)
        plan = self._pack_plans.get(key)
        if plan is None:
            plan = HF.PackPlan(B, T, rows, device, self.pack_granule, halo=halo)
            if plan.nseq > 64:
                return None            # the packed conv kernels hold one sequence per lane
            self._pack_plans[key] = plan
        return plan.fill(lens)

    def _forward_packed(self, plan, x: TensorMask, c, spkr, utterance, noise) -> Mapping:
        """The whole training forward on the VALID frames of a ragged batch (the reference pads, utils/helpers.py:80-135,
        and computes on the padding): the inputs are gathered into ``plan.rows`` rows -- every sequence's frames followed by
        an 18-row halo of its padding (hipvg.functional.PackPlan) -- and handed to ``forward`` as a pseudo batch of
        ``rows`` one-frame sequences whose mask says which rows hold a frame.  Row-local modules need nothing else
        (``lengths = valid, T = 1`` is their row predicate); the four places with time structure read the plan from the
        mask: the conv blocks (per-sequence zero padding), the Transformer stack (varlen attention), the one-frame shift
        and the per-utterance vectors.  The outputs are scattered back to (B, T, .)."""
        B, T = x.mask.shape
        rows = plan.rows
        noise = dict(noise or {})
        v = x.value
        ids_f = v[..., 0].reshape(-1)
        sel = plan.idx.clamp(min=0).long()
        ids_p = torch.where(plan.idx >= 0, ids_f[sel], torch.zeros((), dtype=v.dtype, device=v.device))
        mel_p = HF.pack_rows(v[..., 1:].reshape(B * T, -1).contiguous(), plan)
        xp = torch.cat([ids_p[:, None], mel_p], -1).view(rows, 1, -1)
        mask_p = (plan.valid > 0)[:, None]
        mask_p._vg_plan = plan
        mask_p._vg_len32 = plan.valid
        for k in ("eps_q", "eps_diff"):
            if noise.get(k) is not None:
                n2 = noise[k].reshape(B * T, -1).float().contiguous()
                noise[k] = HF.pack_rows(n2, plan).view(rows, 1, -1)
        out = self.forward(TensorMask(xp, mask_p), c, spkr, utterance, None, noise)
        mask = x.mask

        def back(t):
            if isinstance(t, TensorMask):
                val = t.value
                flat = val.reshape(rows, -1) if val.dim() > 2 else val.reshape(rows, 1)
                if (flat.shape[1] * flat.element_size()) % 16 == 0 and flat.is_floating_point():
                    full = HF.unpack_rows(flat.contiguous(), plan)
                else:
                    inv = plan.inv.long()
                    full = torch.where((inv >= 0)[:, None], flat[inv.clamp(min=0)], torch.zeros((), dtype=flat.dtype, device=flat.device))
                return TensorMask(full.view(B, T, *val.shape[2:]), mask)
            if isinstance(t, dict):
                return {k: back(u) for k, u in t.items()}
            return t

        if self.pack_return == "scalars" and out.get("log_p_mean") is not None and out.get("log_q_mean") is not None:
            # the trainer's step reads the scalar losses and monitors only (scattering every per-frame tensor
            # back to (B, T, .) was tens of MB of gather traffic and a dozen autograd nodes per training step for nothing)
            res = {k: u for k, u in out.items() if u is None or (torch.is_tensor(u) and u.numel() == 1)}
            res["valid_frames"] = plan.valid.sum()
            return res
        res = {k: back(u) for k, u in out.items() if k not in ("token_argmax", "logits", "lengths")}
        res["token_argmax"] = back(TensorMask(out["token_argmax"].view(rows, 1, 1), mask_p)).value.view(B, T)
        res["logits"] = back(TensorMask(out["logits"].view(rows, 1, -1), mask_p)).value.reshape(B * T, -1)   # [B * T, V] like the padded path
        return res

    # ------------------------------------------------------------------ training forward
    def forward(self, x: TensorMask, c: Optional[TensorMask] = None, spkr=None,
                utterance: Optional[TensorMask] = None, diff_input: Optional[TensorMask] = None,
                noise: Optional[Mapping[str, torch.Tensor]] = None) -> Mapping[str, TensorMask]:
        if not self.use_tokens or self.transformer_flow is None:
            raise NotImplementedError("HIP LVTR.forward implements the token + flow model of vae-gslm.yaml")
        noise = noise or {}
        mask, lens = x.mask, x.lengths32
        B, T = mask.shape
        D = self.hp.latent_dim
        plan = getattr(mask, "_vg_plan", None)      # set: this IS the packed pseudo batch (see _forward_packed)
        if plan is None and diff_input is None and x.value.is_cuda and self.pack_rows is not None:
            step_plan = self._step_plan(B, T, lens, x.value.device)
            if step_plan is not None:
                return self._forward_packed(step_plan, x, c, spkr, utterance, noise)
        nseq = plan.B if plan is not None else B      # real sequences (per-utterance vectors, start frames, diffusion steps)
        # ---- side branch (hipvg.functional.fork_side): the utterance encoder and the decoder's time-embedding MLPs depend on
        # nothing computed below until the UNet reads them; on a second stream they (and, through autograd, their
        # backward) run under the Transformer stack
        side = None
        side_ok = (os.environ.get("VG_SIDE_STREAM", "1") != "0" and x.value.is_cuda and diff_input is None and utterance is not None
                   and self.utterance_encoder is not None and hasattr(self.decoder.model, "time_embedding"))
        side_late = os.environ.get("VG_SIDE_FORK", "early") == "late"

        def run_side():
            nonlocal side, main, t_diff, u_c_side, temb_side, tes_side
            t_diff = noise.get("t_diff")
            if t_diff is None:
                t_diff = torch.randint(0, self.decoder.num_timesteps, (nseq,), device=x.value.device).long()
            side, main = HF.fork_side(x.value.device)
            with torch.cuda.stream(side):
                with _side_autocast():
                    u_c_side = self.utterance_encoder(utterance).float()
                # the time-embedding MLPs ([B, 256] activations) in fp32: under autocast every call re-casts their
                # weights and biases (about twenty cast launches per step for three Linears and their backward)
                with torch.autocast("cuda", enabled=False):
                    temb_side = self.decoder.model.time_embedding(t_diff).float()
                    tes_side = self.decoder.model.unet.time_projections(temb_side)

        main = t_diff = u_c_side = temb_side = tes_side = None
        if side_ok and not side_late:
            run_side()
        # token embedding + token_fuser + add as one row kernel when the configuration is the yaml's (ReLU fuser,
        # fp32 embedding table); the module path below is the general one
        fuser = self.token_fuser
        fast_fuse = (diff_input is None and x.value.is_cuda and isinstance(getattr(fuser, "activation", None), nn.ReLU)
                     and self.hp.latent_dim <= 8 and self.token_embedding.weight.dtype == torch.float32
                     and os.environ.get("VG_EMBED_FUSE", "1") != "0")
        if fast_fuse:
            ids, mel = self.split_inputs(x)
            ids = TensorMask(ids.value.long().squeeze(-1), ids.mask)
            tokens = None
        else:
            ids, mel, tokens = self._embed(x)
        # ---- posterior q(z | mel): conv encoder (stock ops) -> fused head + reparameterisation
        with _side_autocast():
            enc = self.encoder[0](mel)
        # optional backward cuts (set by the trainer; see LVTRTrainer._graphed_micro_step): below Transformer layer
        # `grad_cut_layer`, below the fused (token, z) input and below the reparameterised sample.  Each entry is
        # (tensors with tape, detached leaves used downstream); the backward of the part above a cut leaves its
        # gradient in the leaves, and every piece of tape is walked exactly once.
        self.grad_cuts = []
        stack = self.transformer[0]
        stack.grad_cut_layer, stack.grad_cuts = None, self.grad_cuts
        cutting = getattr(self, "grad_cut_layer", None) is not None and torch.is_grad_enabled()
        q_head = self.encoder[1]
        if q_head.plain:
            mu_ls_q = q_head.project(enc.value.float())
            mu_q, ls_q = mu_ls_q[..., :D], mu_ls_q[..., D:]
        else:
            g = q_head(enc)
            mu_q, ls_q = g.mean.value.float(), g.logstd.value.float()
        eps_q = noise.get("eps_q")
        if eps_q is None:
            eps_q = torch.randn(B, T, D, device=mel.device)
        z2, lq2 = HF.reparameterize(mu_q.reshape(-1, D), ls_q.reshape(-1, D), eps_q.reshape(-1, D),
                                    1.0, lens, T)
        cutting = cutting and z2.requires_grad
        if cutting:
            leaves = (z2.detach().requires_grad_(True), lq2.detach().requires_grad_(True))
            self.grad_cuts.append(((z2, lq2), leaves))
            z2, lq2 = leaves
            cl = self.grad_cut_layer
            stack.grad_cut_layer = tuple(int(c) for c in cl) if isinstance(cl, (list, tuple)) else (int(cl),)
        sample_q = TensorMask(z2.view(B, T, D), mask)
        log_q = TensorMask(lq2.view(B, T, D), mask)
        # ---- shift right by one frame, prior network
        if fast_fuse:
            fused = TensorMask(HF.embed_fuse_train(ids.value.reshape(-1), z2, self.token_embedding.weight,
                                                   fuser.linear.weight, fuser.linear.bias, lens, T).view(B, T, -1), mask)
        else:
            fused = self.fuse_inputs(sample_q, tokens)
        if cutting:
            leaf = fused.value.detach().requires_grad_(True)
            self.grad_cuts.append(((fused.value,), (leaf,)))
            fused = TensorMask(leaf, fused.mask)
        init = noise.get("init_state")
        if init is None:
            init = self.initial_state(nseq, mel.device)
        if plan is not None:
            shifted = TensorMask(HF.shift_rows(fused.value.reshape(B * T, -1), init, plan).view(B, T, -1), mask)
        elif getattr(mask, "_vg_full", False):
            # a batch without padding stays without padding under the shift: no mask arithmetic, no re-mask, and the
            # stack's input projection (forward epilogue and backward) runs unmasked
            shifted = TensorMask(torch.cat([init.to(fused.value.dtype), fused.value[:, :-1]], 1), mask)
        else:
            shifted = fused.push(init.to(fused.value.dtype)).pop(1).apply_mask()
        if side_ok and side_late:
            run_side()
        rec_side = None
        unet_side = os.environ.get("VG_SIDE_UNET")
        unet_side = getattr(self, "side_unet", False) if unet_side is None else unet_side == "1"
        if side is not None and unet_side:
            # hip.side_unet (off by default): the WHOLE diffusion decoder on the side branch -- it reads the fused
            # (token, z) frames, not the Transformer's output -- so that its forward (and, through autograd, its
            # backward) runs beside the stack: its half-empty N = 512 launches and ~150 small ones fill in around the
            # stack's tiles.  Step 28.42 -> 27.68 ms (+2.7 % tokens/s, alternating runs of one call); NOT the default
            # because the co-running launches stretch every kernel they share the chip with -- the layer GEMMs and the
            # attention kernels measure 6 % longer each (attn_ffn_path_frac 0.418 -> 0.392) although nothing about them
            # changed, and the per-kernel roofline record of this build is taken kernel by kernel.
            HF.side_defer(True)
            side.wait_stream(main)
            fused.value.record_stream(side)
            with torch.cuda.stream(side), _side_autocast():
                if plan is not None:
                    d_in = fused.cat(HF.seq_rows(u_c_side, plan)[:, None])
                else:
                    d_in = fused.cat(u_c_side[:, None].expand(-1, T, -1))
                rec_side = self.decoder(mel / self.diff_scaling, d_in, t=t_diff, noise=noise.get("eps_diff"),
                                        temb=temb_side, tes=tes_side)
        latent = self.transformer[0](shifted, c)
        cond, mu_ls_p = self._prior_stats(latent)
        # ---- flow + prior log-density + KL (fused row kernel)
        p_z = self.transformer_flow(TensorLogdet(sample_q, 0.0), c=cond)
        u, logdet = p_z.tensor, p_z.logdet
        log_p2, kld = HF.prior_logp_kl(mu_ls_p.reshape(-1, 2 * D), u.value.reshape(-1, D),
                                       logdet.sum(-1).reshape(-1), lq2, lens, T)
        log_p = TensorMask(log_p2.view(B, T, D), mask)
        # ---- token cross-entropy (fused log-softmax + NLL + arg-max)
        logits = self._logits(latent)
        ce_loss, argmax = HF.cross_entropy_sum(logits.reshape(B * T, -1), ids.value.reshape(-1), lens, T)
        # ---- diffusion decoder loss (stock ops)
        if diff_input is None:
            diffusion_input = fused
        else:
            with _side_autocast():
                diffusion_input = self.fuse_inputs(self.encoder(diff_input).sample, tokens)
        u_c = None
        dec_kw = {}
        if side is not None:
            HF.join_side(side, main, [u_c_side, temb_side, *tes_side.values()])
            dec_kw = {"temb": temb_side, "tes": tes_side}
            noise = dict(noise)
            noise["t_diff"] = t_diff
        with _side_autocast():
            if self.utterance_encoder is not None:
                u_c = u_c_side if side is not None else self.utterance_encoder(utterance).float()
                if plan is not None:
                    diffusion_input = diffusion_input.cat(HF.seq_rows(u_c, plan)[:, None])
                else:
                    diffusion_input = diffusion_input.cat(u_c[:, None].expand(-1, T, -1))
            target = mel if diff_input is None else diff_input
            t_diff = noise.get("t_diff")
            if t_diff is None and plan is not None:        # one diffusion step per SEQUENCE (the pseudo batch has `rows`)
                t_diff = torch.randint(0, self.decoder.num_timesteps, (nseq,), device=mel.device).long()
            if rec_side is not None:
                main.wait_stream(side)
                rec_side.record_stream(main)
                rec = rec_side
            else:
                rec = self.decoder(target / self.diff_scaling, diffusion_input,
                                   t=t_diff, noise=noise.get("eps_diff"), **dec_kw)
        mu_p, ls_p = mu_ls_p[..., :D], mu_ls_p[..., D:]
        # the monitors (TensorMask.mean() of the prior / posterior statistics, |posterior mean|, log p, log q): one launch
        # for all seven instead of five stock launches each; the stock expressions remain the fallback
        two_d = lambda t: t.reshape(-1, t.shape[-1]) if t.is_contiguous() else t.reshape(B * T, -1)
        flat = [two_d(mu_ls_p)[:, D:], two_d(mu_ls_p)[:, :D]]
        flat += [ls_q.reshape(B * T, -1) if not ls_q.is_contiguous() else ls_q.view(B * T, -1),
                 mu_q.reshape(B * T, -1) if not mu_q.is_contiguous() else mu_q.view(B * T, -1)]
        stats = None
        if os.environ.get("VG_FUSED_MONITORS", "1") != "0" and all(t.dtype == torch.float32 and t.stride(-1) == 1 for t in flat):
            stats = HF.masked_means([(flat[0], False), (flat[1], False), (flat[2], False), (flat[3], False), (flat[3], True),
                                     (log_p2.view(B * T, -1), False), (lq2.view(B * T, -1), False)], lens, T)
        if stats is not None:
            m_lsp, m_mup, m_lsq, m_muq, m_absq, m_logp, m_logq = stats.unbind(0)
        else:
            m_lsp, m_mup = TensorMask(ls_p, mask).mean(), TensorMask(mu_p, mask).mean()
            m_lsq, m_muq = TensorMask(ls_q, mask).mean(), TensorMask(mu_q, mask).mean()
            m_absq = TensorMask(mu_q, mask).abs().mean()
            m_logp = m_logq = None
        return {
            "log_p": log_p,
            "log_q": log_q,
            "decoder_output": rec,
            "sample_q": sample_q,
            "transformer_latent": latent,
            "logstd": m_lsp,
            "mean": m_mup,
            "q_logstd": m_lsq,
            "q_mean": m_muq,
            "q_z": dict(mean=TensorMask(mu_q, mask), logstd=TensorMask(ls_q, mask), sample=sample_q),
            "u_c": u_c,
            "q_mean_abs": m_absq,
            "log_p_mean": m_logp,
            "log_q_mean": m_logq,
            "ce_loss": ce_loss,
            # build-specific extras
            "kld": kld,
            "token_argmax": argmax.view(B, T),
            "logits": logits,
        }

    # ------------------------------------------------------------------ autoregressive step
    def step(self, x: torch.Tensor, c: Optional[TensorMask] = None, spkr=None,
             past_kv: Optional[List] = None, temperature: float = 1.0, token_temperature: float = 1.0,
             truncated_norm: Optional[Tuple[float, float]] = None, return_attn: bool = False,
             return_distrbution: bool = False, push_init_state: bool = False,
             noise: Optional[torch.Tensor] = None, **kwargs) -> Mapping:
        """x: (B, Tq, 1 + latent): channel 0 = token id (as float), rest = z."""
        xm = TensorMask(x)
        if self.use_tokens:
            _, feats, tokens = self._embed(xm)
            xm = self.fuse_inputs(feats, tokens)
        if push_init_state:
            init = kwargs.get("init_state")
            if init is None:
                init = self.initial_state(xm.value.shape[0], xm.value.device)
            xm = xm.push(init.to(xm.value.dtype)).apply_mask()
        run = self.transformer[0].run(xm, memory=c, past_kv=past_kv, return_attn=return_attn,
                                      return_kv=True)
        outputs = {"transformer_latent": run["output"], "kv": run["kv"]}
        if return_distrbution:
            outputs["z_given"] = run
        if return_attn:
            outputs["self_attn"] = run["self_attn"]
        cond = self.q_spliter(run["output"])
        g = self.transformer[1](cond, temperature=temperature, truncated_norm=truncated_norm, noise=noise)
        outputs["prior"] = g
        sample_z = g.sample
        if self.transformer_flow is not None:
            sample_z = self.transformer_flow.reverse(sample_z, c=cond)
        outputs["output"] = sample_z.value
        if self.use_tokens:
            logits = self._logits(run["output"])
            outputs["logits"] = logits
            b, t, v = logits.shape
            probs = torch.softmax(logits / token_temperature, dim=-1).reshape(b * t, v)
            picked = torch.multinomial(probs, 1).reshape(b, t, 1).float()
            outputs["output"] = torch.cat([picked, outputs["output"]], -1)
        return outputs

    # ------------------------------------------------------------------ encode / decode / likelihood
    def decode(self, x: TensorMask, c: Optional[TensorMask] = None,
               u_c: Optional[torch.Tensor] = None) -> TensorMask:
        ratio = 1.0 / self.sample_ratio
        shape = [x.value.size(0), int(x.value.size(1) * ratio), self.input_dim]
        start = TensorMask.fromlength(torch.randn(shape, device=x.device),
                                      TensorMask.resize_length(x.length, ratio)).apply_mask()
        if self.use_tokens:
            _, feats, tokens = self._embed(x)
            x = self.fuse_inputs(feats, tokens)
        if u_c is not None:
            x = x.cat(u_c[:, None].expand(-1, x.value.size(1), -1))
        with _side_autocast():
            return self.decoder.sample(start, x.apply_mask()) * self.diff_scaling

    def encode(self, x: TensorMask, temperature: float = 1.0, beta=None,
               utterance: Optional[TensorMask] = None) -> TensorMask:
        ids = None
        if self.use_tokens:
            ids, x = self.split_inputs(x)
        with _side_autocast():
            enc = self.encoder[0](x)
        z = self.encoder[1](TensorMask(enc.value.float(), enc.mask), temperature).sample.apply_mask()
        return ids.cat(z) if self.use_tokens else z

    def encode_utterance(self, utterance: TensorMask) -> torch.Tensor:
        if self.use_tokens:
            _, utterance = self.split_inputs(utterance)
        with _side_autocast():
            return self.utterance_encoder(utterance).float()

    def likelihood(self, x: TensorMask, temperature: float = 0.0, gamma: Optional[float] = 1.0,
                   **kwargs) -> torch.Tensor:
        """Per-sequence mean token log-likelihood, reference :337-388 (for a token model the reference returns
        ``sum_t log softmax(logits)[token_t] / length`` and discards the latent density it also evaluates).
        ``init_state=`` (keyword) injects the start frame, as ``noise['init_state']`` does in ``forward``."""
        if not self.use_tokens:
            raise NotImplementedError("HIP LVTR.likelihood implements the token model of vae-gslm.yaml")
        mask = x.mask
        B = mask.shape[0]
        ids, mel, tokens = self._embed(x)
        with _side_autocast():
            enc = self.encoder[0](mel)
        q = self.encoder[1](TensorMask(enc.value.float(), enc.mask), temperature).sample
        init = kwargs.get("init_state")
        if init is None:
            init = self.initial_state(B, mel.device)
        fused = self.fuse_inputs(q, tokens)
        shifted = fused.push(init.to(fused.value.dtype)).pop().apply_mask()
        latent = self.transformer[0](shifted)
        logits = self._logits(latent)
        logp = torch.log_softmax(logits.float(), -1).gather(-1, ids.value.unsqueeze(-1)).squeeze(-1)
        return TensorMask.use_mask(logp, mask).sum(-1) / x.length
