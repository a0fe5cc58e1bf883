"""VAE-GSLM trainer: loss assembly and the optimisation step.

Mirrors the reference ``trainers/speech/lvtr.py`` (``LVTRTrainer`` :14-180):
same hparams, same KL-weight schedule, same total loss
``rec * rec_scale + kld * w + ce * token_kld_weight * w``, manual optimisation
with gradient accumulation, same logged scalar names.  Differences:

* no Lightning: ``training_step`` is called by ``scripts/train.py``; data
  parallelism is the bucketed RCCL reducer (``training_lib/dp.py``), which
  all-reduces once per optimizer step instead of once per micro-batch (the
  mean over ranks is linear, so the update is the same);
* the HiFi-GAN vocoder is only consulted for ``n_mels`` in the reference's
  training path (:32-34); when no vocoder checkpoint directory exists a stub
  with ``n_mels = 80`` is used so training does not depend on that file;
* the KL sum comes fused from the model (``out['kld']``).
"""
from __future__ import annotations

import os
from typing import Mapping, Optional

import torch
import yaml

from hparams.hp import Hparams
from models.speech.lvtr import LVTR
from training_lib.dp import GradReducer
from training_lib.optimizer import create_optimizer
from training_lib.trainer import BaseTrainer


def _vocoder_hparams(path: str) -> Hparams:
    cfg = os.path.join(path, "hp.yaml")
    if os.path.exists(cfg):
        return Hparams.from_yamlfile(cfg)
    return Hparams(n_mels=80, sample_rate=16000)


class _WeightedSum(torch.autograd.Function):
    """sum_i w[i] * term_i for scalar terms: a stack and a dot product; the backward hands every term g * w[i]."""

    @staticmethod
    def forward(ctx, w, *terms):
        ctx.save_for_backward(w)
        ctx.shapes = [t.shape for t in terms]
        return torch.dot(torch.stack([t.reshape(()) for t in terms]), w)

    @staticmethod
    def backward(ctx, g):
        (w,) = ctx.saved_tensors
        return (None, *(gi.reshape(sh) for gi, sh in zip((g * w).unbind(0), ctx.shapes)))


class LVTRTrainer(BaseTrainer):
    def __init__(self, hp: Hparams) -> None:
        super().__init__(hp)
        hp.check_arg_in_hparams("vocoder")
        hp.vocoder.check_arg_in_hparams("path")
        tr = hp.training
        self.rec_loss_scale = tr.get("rec_loss_scale", 1.0)
        self.kld_scale = tr.get("kld_scale", 1.0)
        self.fixed_beta = tr.get("fixed_beta", None)
        if self.fixed_beta is not None:
            if tr.get("scale_rec_beta", True):
                self.rec_loss_scale *= 1 - self.fixed_beta
            self.kld_scale *= self.fixed_beta
        self.mel_rescale = tr.get("mel_rescale", None)
        self.vocoder_hp = _vocoder_hparams(hp.vocoder.path)
        self.model = LVTR(hp.model, input_dim=self.vocoder_hp.n_mels)
        self.apply(self.init_weights)
        self.zero_kld = tr.scheduler.get("zero_kld", 0)
        self.warmup_kld = tr.scheduler.get("warmup_kld", 0)
        self.entropy_weight = tr.get("entropy_weight", 1.0)
        self.use_tokens = self.model.use_tokens
        self.token_kld_weight = tr.get("token_kld_weight", 1.0)
        self.optimizer = self.scheduler = self.reducer = None
        hip = hp.get("hip", None)
        self.use_graph = bool(hip.get("graph", False)) if hip is not None else False
        self._graphs = {}
        self._kw_dev = None
        self._compute_stream = None
        self._owned = None             # parameters whose gradients arrive through autograd's AccumulateGrad
        self._fold_accum = os.environ.get("VG_FOLD_ACCUM", "1") != "0"
        # optional: the micro-batches of an accumulation window as one batch (same gradient, taller GEMMs)
        self.coalesce = bool(hip.get("coalesce_accumulation", False)) if hip is not None else False
        # hip.packed_rows: the Transformer stack of a ragged batch runs on its valid frames only
        self.packed_rows = bool(hip.get("packed_rows", False)) if hip is not None else False
        self.packed_granule = int(hip.get("packed_rows_granule", 1024)) if hip is not None else 1024
        # hip.packed_step: the WHOLE step of a ragged batch on its valid frames (conv stacks, heads and losses too; every
        # sequence carries an 18-frame halo of its padding for the UNet's look-ahead blocks, models.speech.lvtr.
        # LVTR._forward_packed); falls back to packed_rows / padded rows when a batch or the build cannot take it
        ps = hip.get("packed_step", False) if hip is not None else False
        if os.environ.get("VG_PACKED_STEP") is not None:
            ps = os.environ["VG_PACKED_STEP"]
        if isinstance(ps, str):              # ("false" / "off" used to switch the packed step ON)
            key = ps.strip().lower()
            if key not in ("0", "1", "true", "false", "auto"):
                raise ValueError(f"hip.packed_step / VG_PACKED_STEP: expected 0, 1, true, false or auto, got {ps!r}")
            ps = "auto" if key == "auto" else key in ("1", "true")
        # "auto": only where it pays -- the conv stacks' 256-row GEMM tiles need as many rounds at 13,312 rows as at 16,384,
        # so the packed step is level with packed_rows at 77 % fill and ahead from about half-full batches down
        # (bench.py --ragged --ragged-range: +3.7 % at U{0.2 T .. 0.7 T}, +7.5 % at U{0.1 T .. 0.5 T}, +10.6 % at U{0.05 T .. 0.3 T})
        self.packed_step = bool(ps)
        # the fill threshold (packed rows / padded rows) below which a batch is packed: hip.packed_step_fill when given
        # (forced on or auto), else 0.62 for auto and 0.94 when forced on
        fill = hip.get("packed_step_fill", None) if hip is not None else None
        self.packed_step_fill = float(fill) if fill is not None else (0.62 if ps == "auto" else 0.94)
        # hip.side_unet: the diffusion decoder runs beside the Transformer stack on the step's side branch (LVTR.forward)
        if hasattr(self.model, "pack_rows"):
            self.model.side_unet = bool(hip.get("side_unet", False)) if hip is not None else False
        self._held = []
        self._clean_epoch = None       # hipvg.functional.write_epoch() at the moment the gradients were last cleared
        self._watched = False

    def _watch_autograd_writes(self) -> None:
        """A parameter the gradient sink writes may ALSO receive a dense autograd gradient in the same pass (a module
        of the library and a stock op sharing a weight): the sink has to know, or a deferred grouped launch would store
        over it.  One post-accumulate hook per fp32 parameter; it runs for the ~50 parameters autograd owns."""
        if self._watched:
            return
        self._watched = True
        from hipvg import functional as HF
        fresh_start = all(p.grad is None or not bool(p.grad.any()) for p in self.model.parameters()) \
            if next(self.model.parameters()).is_cuda else False
        for p in self.model.parameters():
            if p.requires_grad and p.dtype == torch.float32:
                p.register_post_accumulate_grad_hook(HF.note_autograd_write)
        if fresh_start:
            self._clean_epoch = HF.write_epoch()

    # ------------------------------------------------------------ optimisation plumbing
    def configure_optimizers(self):
        opt, sch = create_optimizer(self.hp.training, self.model.parameters(), self.hp.trainer.total_steps)
        self.optimizer, self.scheduler = opt, sch["scheduler"]
        return [opt], [sch]

    def attach_reducer(self, group=None) -> GradReducer:
        hip = self.hp.get("hip", None)
        bucket = hip.get("bucket_mb", 50) if hip is not None else 50
        if self.use_graph and hip is not None:
            # hipGraph mode reduces the buckets back-to-back after the replay (nothing to overlap with), so fewer,
            # larger messages are strictly better for the ring all-reduce; four buckets still let each bucket's
            # AdamW launch run under the next bucket's all-reduce
            bucket = hip.get("graph_bucket_mb", 256)
        overlap = hip.get("overlap", True) if hip is not None else True
        comm = hip.get("comm", "torch") if hip is not None else "torch"
        # Segmented replay (hipGraph mode, more than one rank): the micro-step is captured as several graphs cut
        # below the Transformer layers `graph_cut_layer` (with two more cuts near the input so that every piece of
        # the autograd tape is walked once): after each graph the gradients of everything above its cut are final and
        # their buckets go on the wire while the next graph runs more of backward.
        import torch.distributed as dist
        world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        cut = hip.get("graph_cut_layer", "auto") if hip is not None else None
        stack = self.model.transformer[0]
        nl = len(stack.layers)
        if os.environ.get("VG_GRAPH_CUT"):           # lab override: one layer or a comma-separated list
            cut = [int(v) for v in os.environ["VG_GRAPH_CUT"].split(",")]
        if cut == "auto":
            # Forced on one rank (VG_GRAPH_SEGMENTS=force, full model, 16 layers, 42.05 ms per step with one graph):
            # further graphs are free as long as the FIRST one stays short -- cuts at 12 / 12,8 / 13,10,7 /
            # 14,12,10,8: 42.05 / 41.95 / 41.94 / 42.04 ms -- while a first graph that reaches down to layer 10 or
            # lower costs 1.0-1.2 ms (cut at 10, 8, 4, 1 alone).  Quarter points, top one first.
            cut = [(3 * nl) // 4, nl // 2, nl // 4]
        cuts = [] if cut is None else sorted({int(c) for c in (cut if isinstance(cut, (list, tuple)) else [cut])},
                                             reverse=True)
        cuts = [c for c in cuts if 0 < c < nl]
        seg_env = os.environ.get("VG_GRAPH_SEGMENTS", "2")       # "1": one graph; "force": segmented on one rank too
        single_rank = os.environ.get("VG_DP_SINGLE_RANK", "0") == "1"      # one-rank communicator, whole DP step (dp.py)
        self._segmented = bool(self.use_graph and (world > 1 or single_rank or seg_env == "force") and cuts and seg_env != "1")
        boundaries = ()
        params = list(self.model.parameters())
        m = self.model
        above = [p for mod in (m.decoder, m.utterance_encoder, m.transformer_flow, m.transformer[1],
                               stack.final_norm, getattr(stack, "out", None), m.q_spliter, m.token_spliter,
                               m.token_predictor) if mod is not None for p in mod.parameters()]
        if self._segmented:
            # Buckets are filled walking the parameter list backwards, so the list is put in the order in which backward
            # FINISHES gradients, reversed: [input side: encoder, embeddings, the stack's own input projection]
            # [layer 0] .. [layer L-1] [everything above the stack].  Registration order alone put the stack's input
            # projection (registered after its layers, final only when backward has walked the whole stack) into the
            # bucket of the top layers -- no bucket was final after the first graph and the segmented replay switched
            # itself off at the full configuration (round 4: every collective ran after the last backward kernel).
            # A bucket ends at every cut, where the layers begin and where they end.
            seen = set()
            above = [p for p in above if not (id(p) in seen or seen.add(id(p)))]
            layer_params = [p for layer in stack.layers for p in layer.parameters()]
            taken = {id(p) for p in above} | {id(p) for p in layer_params}
            late = [p for p in params if id(p) not in taken]
            params = late + layer_params + above
            boundaries = tuple(list(stack.layers[c - 1].parameters())[-1] for c in cuts)
            boundaries += (list(stack.layers[nl - 1].parameters())[-1],)
            if late:
                boundaries += (late[-1],)
        wire = str(hip.get("comm_dtype", "fp32")) if hip is not None else "fp32"
        wire = os.environ.get("VG_COMM_DTYPE", wire)
        self.reducer = GradReducer(params, bucket_mb=bucket, overlap=overlap, group=group, comm=comm,
                                   boundaries=boundaries, wire_dtype=wire)
        self._cut_layers, self._early_buckets = [], []
        if self._segmented:
            top = nl
            for c in cuts:          # after graph k the parameters above cut k are final
                for layer in stack.layers[c:top]:
                    above += list(layer.parameters())
                top = c
                self._early_buckets.append(self.reducer.buckets_within(above))
            # one more piece: the cuts near the input (fused token / z input, reparameterised sample) are a graph of
            # their own, so that the bottom layers' bucket goes on the wire under the posterior encoder's backward
            for layer in stack.layers[0:top]:
                above += list(layer.parameters())
            self._early_buckets.append(self.reducer.buckets_within(above))
            self._cut_layers = cuts
            if self._early_buckets[0]:
                self.model.grad_cut_layer = tuple(cuts)
            else:
                import warnings
                warnings.warn("segmented hipGraph replay is off: no gradient bucket is final after the first graph")
                self._segmented = False
        self.graph_cuts = list(cuts) if self._segmented else []      # reported by bench.py's comm block
        bind = getattr(self.optimizer, "bind", None)
        if callable(bind) and next(self.model.parameters()).is_cuda:
            bind(self.reducer)                 # AdamW + bf16 weight refresh + gradient clear: one launch per bucket
            # weights loaded into the bound model later must reach the bf16 copies the GEMMs read
            self.model.register_load_state_dict_post_hook(lambda module, incompatible: self.optimizer.sync_shadows())
        return self.reducer

    def current_kld_weight(self) -> float:
        w = self.kld_scale
        step = self.global_step
        if self.warmup_kld > 0 and self.zero_kld < step + 1 <= self.warmup_kld:
            w = self.kld_scale * ((step - self.zero_kld) / self.warmup_kld)
        if self.zero_kld > 0 and step <= self.zero_kld:
            w = 0.0
        return w

    # ------------------------------------------------------------ one micro-batch
    def _backward_tail(self, segment: Optional[int] = None) -> None:
        """Backward of the parts below the model's backward cuts (none unless ``model.grad_cut_layer`` is set): the
        gradient left in each cut's leaves is fed into the tape below it, deepest cut last.  ``segment`` k runs only
        the piece that belongs to graph k + 2 of a segmented replay: the k-th Transformer cut from the top; the piece
        after the last of them holds the cuts near the input."""
        cuts = list(reversed(getattr(self.model, "grad_cuts", [])))      # top Transformer cut first ... z last
        if segment is not None:
            n = len(self._cut_layers)
            cuts = cuts[segment:segment + 1] if segment < n else cuts[n:]
        from hipvg import functional as HF
        HF.defer_vec_grads(self._defer_colsums())
        try:
            for below, leaves in cuts:
                pairs = [(t, l.grad) for t, l in zip(below, leaves) if l.grad is not None]
                if pairs:
                    torch.autograd.backward([t for t, _ in pairs], [g for _, g in pairs])
        except BaseException:
            HF.reset_vec_grads()
            raise
        HF.defer_vec_grads(False)          # flushes: every bucket of this piece is complete when it returns

    def _defer_colsums(self) -> bool:
        """The finishing launches of the bias / scale column sums may wait for the end of a backward piece wherever
        nothing acts on "gradient ready" inside it: one rank, a non-final micro-step, or a captured segment (its
        buckets are reduced after the replay)."""
        return (self.reducer is None or not self.reducer.exchange or not self.reducer.sync_now
                or getattr(self, "_segmented", False))

    def _training_loop(self, batch: Mapping, batch_idx: int, noise: Optional[Mapping] = None,
                       kld_weight=None, backward_tail: bool = True):
        if kld_weight is None:
            kld_weight = self.current_kld_weight()
        kwargs = {}
        if self.model.utterance_encoder is not None:
            kwargs["utterance"] = batch["cropped_mel_utt"]
        if "cropped_mel" in batch:
            kwargs["diff_input"] = batch["cropped_mel"]
        model_input = batch["mel"]
        if self.use_tokens:
            model_input = batch["tokens"].expand().cat(batch["mel"])
        if hasattr(self.model, "pack_return"):      # a packed step hands back what this function reads, nothing per frame
            self.model.pack_return = "scalars" if (self.entropy_weight == 1.0 and self.model.training) else "all"
        try:
            out = self.model(model_input, noise=noise, **kwargs)
        finally:
            if hasattr(self.model, "pack_return"):
                self.model.pack_return = "all"
        kld = out["kld"] if self.entropy_weight == 1.0 else None
        if kld is None:   # non-default entropy weighting: generic path
            from training_lib.losses import masked_loss
            kld = masked_loss(out["log_q"] * self.entropy_weight, out["log_p"], fn=lambda a, b: a - b)
        rec = out["decoder_output"]
        # loss = rec * rec_scale + kld * w + ce * token_kld_weight * w as ONE weighted sum (a stack + a dot forward, one
        # product backward) instead of seven scalar launches forward and as many backward; the KL weight may be the
        # device scalar of a captured step (it changes from replay to replay)
        terms = [rec, kld] + ([out["ce_loss"]] if self.use_tokens else [])
        base = [float(self.rec_loss_scale), 1.0] + ([float(self.token_kld_weight)] if self.use_tokens else [])
        if (os.environ.get("VG_LOSS_DOT", "1") != "0" and all(torch.is_tensor(t) and t.is_cuda and t.numel() == 1 and t.dtype == torch.float32
                                                                for t in terms)):
            dev = rec.device
            key = (dev, tuple(base))
            cache = self.__dict__.setdefault("_loss_w", {})
            if key not in cache:      # [rec_scale, 0, 0] + w * [0, 1, token_kld_weight]
                cache[key] = (torch.tensor([base[0]] + [0.0] * (len(base) - 1), device=dev),
                              torch.tensor([0.0] + base[1:], device=dev))
            fixed, scaled = cache[key]
            kw = kld_weight if torch.is_tensor(kld_weight) else float(kld_weight)
            loss = _WeightedSum.apply(torch.addcmul(fixed, scaled, kw) if torch.is_tensor(kw) else fixed + scaled * kw, *terms)
        else:
            loss = rec * self.rec_loss_scale + kld * kld_weight
            if self.use_tokens:
                loss = loss + out["ce_loss"] * (self.token_kld_weight * kld_weight)
        if self.reducer is not None:
            # the set of inputs decides which sub-networks run how often (``cropped_mel`` sends the posterior
            # encoder and the token fuser through a second time), hence how often each parameter reports
            self.reducer.new_backward(signature=tuple(sorted(batch.keys())))
        # Parameters autograd still owns (stock sub-networks) get their gradient through AccumulateGrad, which ADDS
        # into the bucket view (one tiny launch per parameter, ~50 per step).  With the views taken away autograd
        # keeps the fresh gradient tensors instead, and one multi-tensor launch adds them afterwards.  Only where
        # nothing acts on "gradient ready" before the pass ends (one rank, or a captured / non-final micro-step).
        # (Not in a segmented pass: folding per piece -- five multi-tensor launches instead of one -- measured 0.2-0.3 ms
        # SLOWER than the per-parameter adds on the one-rank RCCL step, 31.63-31.73 against 31.42 ms, round 4.)
        fold = None
        if (self._owned is not None and self.reducer is not None and backward_tail and not getattr(self, "_segmented", False)
                and (not self.reducer.exchange or not self.reducer.sync_now) and self._fold_accum):
            fold = [(p, p.grad) for p in self._owned if p.grad is not None]
            for p, _ in fold:
                p.grad = None
        from hipvg import functional as HF
        # first backward pass since the gradients were cleared: grouped weight-gradient launches may store their whole
        # tiles instead of adding to the zeros underneath (hipvg.functional.begin_backward_pass)
        HF.begin_backward_pass(getattr(self, "_wgrad_fresh", False))
        HF.defer_vec_grads(self._defer_colsums())
        try:
            loss.backward()
        except BaseException:
            HF.reset_vec_grads()
            HF.end_backward_pass()
            raise
        if backward_tail:
            # the pieces below the model's own cuts (the posterior encoder sits below the reparameterised sample) belong
            # to the same bracket: their weight gradients leave with this pass's instead of as a small launch of their own
            self._backward_tail()
        else:
            HF.defer_vec_grads(False)
        if fold is not None:
            pairs = [(v, p.grad) for p, v in fold if p.grad is not None]
            if pairs:
                torch._foreach_add_([v for v, _ in pairs], [g for _, g in pairs])
            for p, v in fold:
                p.grad = v
        elif self._owned is None and self.reducer is not None:
            # after the first backward: who was never written by the library?
            self._owned = [p for p in self.model.parameters()
                           if p.requires_grad and p.grad is not None and not getattr(p, "_vg_sunk", False)]
        # (log_p_mean / log_q_mean: the same masked means from the model's fused monitor launch, when it ran)
        lp_mean = out.get("log_p_mean")
        lq_mean = out.get("log_q_mean")
        result = {"kld": kld.detach(), "rec_loss": rec.detach(),
                  "log_p": -(lp_mean if lp_mean is not None else out["log_p"].mean()).detach(),
                  "length": out["log_p"].length.sum() if "log_p" in out else out["valid_frames"], "kld_weight": kld_weight,
                  "logstd": out["logstd"].detach(), "q_logstd": out["q_logstd"].detach(),
                  "log_q": -(lq_mean if lq_mean is not None else out["log_q"].mean()).detach(),
                  "q_mean_abs": out["q_mean_abs"].detach(),
                  "loss": loss.detach()}
        if self.use_tokens:
            result["token_kld"] = out["ce_loss"].detach()
        return result

    @staticmethod
    def _concat_batches(batches) -> Mapping:
        """Micro-batches of one accumulation window as ONE batch (rows = all their sequences): the loss terms are
        sums over frames, so one forward/backward over the concatenation yields the accumulated gradient."""
        from utils.tensormask import TensorMask
        out = {}
        for k in batches[0]:
            vals = [b[k].value for b in batches]
            T = max(v.shape[1] for v in vals)
            full = all(getattr(b[k].mask, "_vg_full", False) and b[k].value.shape[1] == T for b in batches)
            pad = lambda v, t: v if t == T else torch.nn.functional.pad(v, (0, 0) * (v.dim() - 2) + (0, T - t))
            value = torch.cat([pad(v, v.shape[1]) for v in vals], 0)
            if full:
                out[k] = TensorMask(value)
            else:
                out[k] = TensorMask(value, torch.cat([torch.nn.functional.pad(b[k].mask, (0, T - b[k].mask.shape[1]))
                                                      for b in batches], 0))
                counts = [b[k].value.shape[0] * b[k].value.shape[1] if getattr(b[k].mask, "_vg_full", False)
                          else getattr(b[k].mask, "_vg_valid", None) for b in batches]
                if all(c is not None for c in counts):
                    out[k].mask._vg_valid = int(sum(counts))
        return out

    def training_step(self, batch: Mapping, batch_idx: int, noise: Optional[Mapping] = None):
        if self.use_graph and noise is None and os.environ.get("VG_GRAPH_OWN_STREAM", "1") != "0":
            self.enter_compute_stream(batch["mel"].value.device)
        return self._training_step(batch, batch_idx, noise)

    def enter_compute_stream(self, dev) -> None:
        """(Called by ``training_step``; callers that stage inputs on other streams may call it first so that the
        staging already synchronises with the stream the step will run on.)
        hipGraph mode never launches on the null stream.  On ROCm 7.2 a graph launched on the null stream after
        a cross-stream hipStreamWaitEvent (what the reducer issues once world > 1) replays with corrupt kernel
        arguments -- reproduced in one process, DESIGN.md "hipGraph on the null stream"; on a created stream the
        same sequence is clean.  The calling thread is moved onto a stream the trainer owns, once (switching per
        step costs 1 ms of cross-stream waits per step), so the caller's later work is ordered after the step as
        before."""
        from hipvg import functional as HF
        cur = torch.cuda.current_stream(dev)
        # The stream is hipvg.functional.graph_launch_stream's: one per process and device, shared by every trainer, of the
        # kind VG_LAUNCH_STREAM names (ROCm 7.0's hipGraphLaunch can walk off the exec's internal stream list when the
        # launch stream shares a pooled hardware queue with two of them -- the hazard, the safe kinds and their price are
        # described there).
        if HF.is_launch_stream(cur):
            return
        self._compute_stream = HF.graph_launch_stream(dev)
        self._compute_stream.wait_stream(cur)
        torch.cuda.set_stream(self._compute_stream)

    def _training_step(self, batch: Mapping, batch_idx: int, noise: Optional[Mapping] = None):
        last = (batch_idx + 1) % self.gradient_update_step == 0
        if self.coalesce and noise is None and self.gradient_update_step > 1:
            # hip.coalesce_accumulation: hold the window's micro-batches and run them as one launch sequence
            self._held.append(batch)
            if not last:
                return {"coalesced": True}
            batch, self._held = self._concat_batches(self._held), []
        if self.reducer is not None:
            self.reducer.sync_now = last
        # "Fresh" = the gradient buffers hold zeros when this pass starts.  Taken from their real state, not from
        # batch_idx (per-epoch batch indices, or a backward outside training_step, would leave accumulated
        # gradients under a grouped launch that STORES): whoever cleared them recorded the library's write epoch, and
        # nothing has written a gradient since (hipvg.functional.write_epoch).
        from hipvg import functional as _HF
        self._watch_autograd_writes()
        self._wgrad_fresh = self._clean_epoch is not None and self._clean_epoch == _HF.write_epoch()
        try:
            if self.use_graph and noise is None:
                try:
                    out = self._graphed_micro_step(batch, batch_idx, last)
                finally:
                    # the row count chosen for this capture / replay must not outlive it: a later direct call of the model
                    # (likelihood(), a user's forward) would pack a different batch into it
                    self._clear_pack_rows()
            else:
                try:
                    self._choose_pack_rows(batch, eager=True)
                    out = self._training_loop(batch, batch_idx, noise)
                finally:
                    self._clear_pack_rows()        # (the eager branch left "auto" on the model for later direct calls)
        finally:
            self._wgrad_fresh = False
            self._clean_epoch = None               # a pass ran (eagerly, captured or replayed): the buffers hold gradients
            _HF.end_backward_pass()                # a backward outside the trainer never inherits "gradients are zero"
        if last:
            clip = self.hp.training.get("gradient_clip_val", None)
            pipelined = (clip is None and self.reducer is not None and self.reducer.exchange
                         and getattr(self.optimizer, "clears_gradients", False))
            if self.reducer is not None and not pipelined:
                self.reducer.finish()
            if clip is not None:
                torch.nn.utils.clip_grad_norm_(self.model.parameters(), clip)
            if pipelined:      # each bucket's AdamW launch waits for that bucket's all-reduce only
                self.reducer.flush()                   # buckets that nothing launched yet go on the wire first
                self.optimizer.step(bucket_wait=self.reducer.wait_bucket)
                self.reducer.finish()
            else:
                self.optimizer.step()
            if getattr(self.optimizer, "clears_gradients", False):
                pass                                   # vg_adamw zeroed the buckets in the same pass
            elif self.reducer is not None:
                self.reducer.zero_grad()
            else:
                self.optimizer.zero_grad(set_to_none=True)
            self._clean_epoch = _HF.write_epoch()      # zeros underneath until the library's write epoch moves
            if self.use_graph and not getattr(self.optimizer, "clears_gradients", False):
                from hipvg import functional as HF
                HF.refresh_shadows(self.model.parameters())   # captured GEMMs read the bf16 copies by address
            n = out["length"]
            self.log("train/kld", out["kld"] / n)
            self.log("train/rec_loss", out["rec_loss"] / n)
            self.log("train/kld_weight", out["kld_weight"])
            self.log("train/z_given_logstd", out["logstd"])
            self.log("train/q_logstd", out["q_logstd"])
            self.log("train/q_entropy", out["log_q"])
            self.log("train/q_mean_abs", out["q_mean_abs"])
            self.log("train/cross_entropy", out["log_p"])
            self.log("train/lr", self.scheduler.get_last_lr()[0])
            if self.use_tokens:
                self.log("train/token_kld", out["token_kld"] / n)
            self.scheduler.step()
            self.global_step += 1
        return out

    # ------------------------------------------------------------ packed rows
    def _choose_pack_rows(self, batch: Optional[Mapping], eager: bool = False):
        """Sets ``LVTR.pack_rows`` (packed step) or ``TransformerLayerStack.pack_rows`` for the coming forward and returns
        the choice (None: padded rows) -- it is part of a hipGraph's key."""
        stack = self.model.transformer[0] if hasattr(self.model, "transformer") else None
        if stack is None or not hasattr(stack, "pack_rows"):
            return None
        rows = None
        stack.pack_granule = self.packed_granule
        step_ok = self.packed_step and hasattr(self.model, "packable") and self.model.packable() and "cropped_mel" not in (batch or {})
        if hasattr(self.model, "pack_rows"):
            self.model.pack_rows, self.model.pack_granule, self.model.pack_fill = None, self.packed_granule, self.packed_step_fill
        if (self.packed_rows or step_ok) and batch is not None:
            tm = batch.get("tokens", batch.get("mel"))
            if tm is not None and not getattr(tm.mask, "_vg_full", False):
                if eager:
                    rows = "auto"
                    if step_ok:
                        self.model.pack_rows = "auto"      # (a batch it declines falls through to the stack's own packing)
                else:
                    from hipvg import functional as HF
                    B, T = tm.mask.shape[:2]
                    # the valid-frame count rides with the mask when the loader knew it on the host
                    # (training_lib.prefetch: `_vg_valid`); a device -> host read is the fallback
                    total = getattr(tm.mask, "_vg_valid", None)
                    if total is None:
                        total = int(tm.mask.sum().item())
                    # need >= total, so a batch whose valid frames alone are past the threshold can never qualify: decline
                    # without reading the per-sequence lengths back (the common 77 %-fill case paid a blocking
                    # device -> host read and two small launches per step for nothing)
                    if step_ok and HF.pack_rows_bucket(int(total), self.packed_granule) > int(self.packed_step_fill * B * T):
                        step_ok = False
                    if step_ok:
                        # + the halo rows: min(len + halo, T) per sequence needs the lengths; the bound len + halo is
                        # within a granule of it and a bucket only has to be large enough
                        halo = self.model.pack_halo()
                        lens = tm.mask.sum(-1)
                        need = int(torch.clamp(lens + halo, max=T).sum().item())
                        cand = HF.pack_rows_bucket(need, self.packed_granule)
                        nseq = B + -(-min(cand, self.packed_granule) // T)
                        if os.environ.get("VG_DEBUG_PACK"):
                            print(f"[packed step] B={B} T={T} need={need} bucket={cand} nseq={nseq} limit={int(0.94 * B * T)}", flush=True)
                        if cand <= int(self.packed_step_fill * B * T) and nseq <= 64:
                            self.model.pack_rows = cand
                            stack.pack_rows = None
                            return ("step", cand)
                    if self.packed_rows:
                        cand = HF.pack_rows_bucket(int(total), self.packed_granule)
                        rows = cand if cand <= int(0.94 * B * T) else None
        stack.pack_rows = rows if self.packed_rows else None
        return rows

    def _clear_pack_rows(self) -> None:
        stack = self.model.transformer[0] if hasattr(self.model, "transformer") else None
        if stack is not None and hasattr(stack, "pack_rows"):
            stack.pack_rows = None
        if hasattr(self.model, "pack_rows"):
            self.model.pack_rows = None

    # ------------------------------------------------------------ hipGraph replay of a micro-step
    def _pad_for_graph(self, batch: Mapping) -> Mapping:
        """Real batches have a different number of frames every step; one graph per length would mean a capture
        per step.  Ragged batches are therefore right-padded (masked frames, which every kernel already skips)
        to the next multiple of ``hip.graph_pad_frames`` (default 64), so at most maxT/64 graphs exist; they share
        one memory pool because only one replays at a time."""
        hip = self.hp.get("hip", None)
        mult = int(hip.get("graph_pad_frames", 64)) if hip is not None else 64
        from utils.tensormask import TensorMask
        out = dict(batch)
        for k in ("tokens", "mel"):
            v = batch.get(k)
            if v is None or getattr(v.mask, "_vg_full", False) or mult <= 1:
                continue
            T = v.value.shape[1]
            pad = (-T) % mult
            if pad:
                val = torch.nn.functional.pad(v.value, (0, 0) * (v.value.dim() - 2) + (0, pad))
                out[k] = TensorMask(val, torch.nn.functional.pad(v.mask, (0, pad)))
                if hasattr(v.mask, "_vg_valid"):
                    out[k].mask._vg_valid = v.mask._vg_valid
        return out

    def _graphed_micro_step(self, batch: Mapping, batch_idx: int, last: bool):
        """Forward + backward of one micro-batch captured once per input shape into a hipGraph
        and replayed afterwards: ~2,700 kernel launches per micro-batch become one graph launch,
        so the GPU is never starved by the host.  The KL weight is a device scalar (it changes
        during warm-up), inputs are copied into static buffers, gradients accumulate into the
        reducer's static buckets.  With N > 1 ranks the bucket all-reduces are issued after the
        last replay of the window (bulk, not overlapped with backward)."""
        self.enter_compute_stream(batch["mel"].value.device)     # (callers that come here directly: the launch-stream rule)
        batch = self._pad_for_graph(batch)
        key = tuple((k, tuple(v.value.shape), getattr(v.mask, "_vg_full", False)) for k, v in sorted(batch.items()))
        # packed rows: the number of packed rows is part of the graph's shape.  It is chosen here, on the host, from
        # the batch's valid-frame count (one small device -> host read per micro-step) and rounded up to the granule,
        # so a ragged data stream needs one graph per (padded length, bucket) and not one per batch.
        key = key + (self._choose_pack_rows(batch), bool(getattr(self, "_wgrad_fresh", False)))
        dev = batch["mel"].value.device
        if self._kw_dev is None:
            self._kw_dev = torch.zeros((), dtype=torch.float32, device=dev)
        self._kw_dev.fill_(self.current_kld_weight())
        ent = self._graphs.get(key)
        if ent is None:
            from utils.tensormask import TensorMask
            static = {k: TensorMask(v.value.clone(), v.mask if getattr(v.mask, "_vg_full", False) else v.mask.clone())
                      for k, v in batch.items()}
            # A new shape may first appear in the middle of an accumulation window: the warm-up pass and the capture
            # below must not disturb the gradients already accumulated, so the buckets are set aside and put back.
            saved = [b["flat"].clone() for b in self.reducer.buckets] if self.reducer is not None else None

            def restore_buckets():
                if saved is not None:
                    for b, keep in zip(self.reducer.buckets, saved):
                        b["flat"].copy_(keep)

            def forget_lengths():
                # The sequence lengths are cached on the mask tensors (TensorMask.lengths32 / .length).  The graph must
                # RECOMPUTE them from the static masks on every replay, so the reductions have to be recorded inside
                # the capture: drop what the warm-up pass cached (a graph that kept reading the first batch's lengths
                # would mask every later batch of this shape with them).
                for tm in static.values():
                    if not getattr(tm.mask, "_vg_full", False):
                        tm.mask._vg_len32 = None
                        tm.mask._vg_len64 = None

            if self.reducer is not None:
                self.reducer.sync_now = False          # collectives stay outside the graph
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):              # eager pass on the capture side-stream (lazy inits)
                self._training_loop(static, batch_idx, kld_weight=self._kw_dev)
            torch.cuda.current_stream().wait_stream(side)
            forget_lengths()
            if getattr(self, "profile_in_graph", False):
                import hipvg
                hipvg.prof_enable(True)                # event-record nodes become part of the graph
            graph = torch.cuda.CUDAGraph()
            if getattr(self, "_graph_pool", None) is None:
                self._graph_pool = torch.cuda.graph_pool_handle()
            graph2 = None
            try:
                if getattr(self, "_segmented", False):
                    with torch.cuda.graph(graph, pool=self._graph_pool):
                        out = self._training_loop(static, batch_idx, kld_weight=self._kw_dev, backward_tail=False)
                    graph2 = []
                    n_cut = len(self._cut_layers)
                    # the Transformer cuts, then (if the model placed them) the cuts near the input as a last piece
                    for k in range(n_cut + (1 if len(getattr(self.model, "grad_cuts", [])) > n_cut else 0)):
                        g = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(g, pool=self._graph_pool):
                            self._backward_tail(k)
                        graph2.append(g)
                else:
                    with torch.cuda.graph(graph, pool=self._graph_pool):
                        out = self._training_loop(static, batch_idx, kld_weight=self._kw_dev)
            except Exception as exc:      # e.g. another library touching the device mid-capture: run eagerly instead
                import warnings
                warnings.warn(f"hipGraph capture of the micro-step failed ({exc!r}); continuing with eager launches")
                self.use_graph = False
                self._segmented = False
                self.model.grad_cut_layer = None     # eager launches report gradients as they complete: no cuts
                torch.cuda.synchronize()
                restore_buckets()
                if self.reducer is not None:
                    self.reducer.sync_now = last
                return self._training_loop(batch, batch_idx, None)
            restore_buckets()                          # capture does not execute; the warm-up pass is undone
            ent = self._graphs[key] = (graph, static, out, graph2)
        graph, static, out, graph2 = ent
        for k, v in batch.items():
            static[k].value.copy_(v.value, non_blocking=True)
            if static[k].mask is not v.mask and not getattr(static[k].mask, "_vg_full", False):
                static[k].mask.copy_(v.mask, non_blocking=True)     # lengths are recomputed inside the graph
        if self.reducer is not None:
            self.reducer.phase = 0
        graph.replay()
        reduce_now = last and self.reducer is not None and self.reducer.exchange
        if graph2 is not None:
            for k, g in enumerate(graph2):
                if reduce_now:     # final already: on the wire under the next graph
                    self.reducer.reduce_buckets(self._early_buckets[k])
                self.reducer.phase = k + 1         # launches logged from here on came after piece k + 1 was queued
                g.replay()
        if reduce_now:
            self.reducer.phase = (len(graph2) if graph2 is not None else 0) + 1      # after the last backward kernel
            self.reducer.reduce_all()
        res = dict(out)
        res["kld_weight"] = self.current_kld_weight()
        return res

    # ------------------------------------------------------------ validation
    def on_validation_start(self) -> None:
        self.sampled = 0
        self._val_acc = None

    @torch.no_grad()
    def validation_step(self, batch: Mapping, batch_idx: int, noise: Optional[Mapping] = None):
        """Reference ``validation_step`` (trainers/speech/lvtr.py:182-286) without its audio logging (vocoder,
        sampler and TensorBoard are outside this build): the same forward, ``val/kld``, ``val/rec_loss`` and
        ``val/token_kld`` per valid frame, averaged over the epoch with weight ``len(batch)`` and, at
        ``on_validation_end``, over ranks -- Lightning's ``log(on_epoch=True, sync_dist=True, batch_size=...)``."""
        kwargs = {}
        if self.model.utterance_encoder is not None:
            kwargs["utterance"] = batch["cropped_mel_utt"]
        if "cropped_mel" in batch:
            kwargs["diff_input"] = batch["cropped_mel"]
        model_input = batch["mel"]
        if self.use_tokens:
            model_input = batch["tokens"].expand().cat(batch["mel"])
        self._choose_pack_rows(batch, eager=True)     # never a stale row count from a training micro-step
        was_training = self.model.training
        self.model.eval()
        try:
            out = self.model(model_input, noise=noise, **kwargs)
        finally:
            self.model.train(was_training)
        n = out["log_p"].length.sum()
        vals = [out["kld"] / n, out["decoder_output"] / n]
        if self.use_tokens:
            vals.append(out["ce_loss"] / n)
        bs = float(len(batch["mel"]))
        step = torch.stack([v.detach().float().reshape(()) for v in vals] + [torch.ones((), device=n.device)]) * bs
        self._val_acc = step if getattr(self, "_val_acc", None) is None else self._val_acc + step
        res = {"kld": vals[0], "rec_loss": vals[1], "length": n}
        if self.use_tokens:
            res["token_kld"] = vals[2]
        return res

    def on_validation_end(self, group=None) -> Mapping[str, float]:
        """One all-reduce of the epoch sums (three scalars and their weight) over the ranks; logs and returns the
        means under the reference's names."""
        import torch.distributed as dist
        self.sampled = 0
        acc = getattr(self, "_val_acc", None)
        world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        nvals = 3 if self.use_tokens else 2
        if world > 1:
            # Lightning reduces each rank's epoch MEAN with equal weights (sync_dist=True).  EVERY rank takes part in
            # the collective: a rank whose shard was empty (uneven validation shards, limit_val_batches) contributes
            # zeros and a zero "has data" flag, and the mean is taken over the ranks that saw data.
            dev = acc.device if acc is not None else next(self.model.parameters()).device
            msg = torch.zeros(nvals + 1, dtype=torch.float32, device=dev)
            if acc is not None and float(acc[-1]) > 0:
                msg[:-1] = acc[:-1] / acc[-1]
                msg[-1] = 1.0
            dist.all_reduce(msg, op=dist.ReduceOp.SUM, group=group)
            if float(msg[-1]) == 0:
                self._val_acc = None
                return {}
            mean = msg[:-1] / msg[-1]
        else:
            if acc is None:
                return {}
            mean = acc[:-1] / acc[-1]
        names = ["val/kld", "val/rec_loss"] + (["val/token_kld"] if self.use_tokens else [])
        out = {k: float(v) for k, v in zip(names, mean.cpu())}
        for k, v in out.items():
            self.log(k, v)
        self._val_acc = None
        return out

    # ------------------------------------------------------------ checkpoints
    def save_checkpoint(self, filepath: str) -> None:
        """Compact checkpoint = ``LVTR.state_dict()`` (reference :294-296)."""
        torch.save(self.model.state_dict(), filepath)

    def save_full_checkpoint(self, filepath: str) -> None:
        """Everything a resumed run needs (what Lightning's ``ModelCheckpoint`` keeps for the reference,
        scripts/train.py:59-66): weights, optimizer moments and step, LR schedule, ``global_step``."""
        torch.save({"state_dict": self.model.state_dict(),
                    "optimizer": self.optimizer.state_dict() if self.optimizer is not None else None,
                    "scheduler": self.scheduler.state_dict() if self.scheduler is not None else None,
                    "global_step": int(self.global_step)}, filepath)

    def load_checkpoint(self, filepath: str, map_location=None, ckpt=None) -> bool:
        """Load a full or a compact checkpoint.  Call after ``configure_optimizers`` / ``attach_reducer`` so the
        optimizer state lands in the bound flat buffers.  Returns True when training state was restored too
        (``fit(ckpt_path=...)`` of the reference, scripts/train.py:104)."""
        if ckpt is None:                 # ``ckpt``: the already loaded file (scripts.train reads it once)
            ckpt = torch.load(filepath, map_location=map_location)
        if not (isinstance(ckpt, dict) and "state_dict" in ckpt and "global_step" in ckpt):
            self.model.load_state_dict(ckpt)
            return False
        self.model.load_state_dict(ckpt["state_dict"])
        if ckpt.get("optimizer") is not None and self.optimizer is not None:
            self.optimizer.load_state_dict(ckpt["optimizer"])
        if ckpt.get("scheduler") is not None and self.scheduler is not None:
            self.scheduler.load_state_dict(ckpt["scheduler"])
        self.global_step = int(ckpt["global_step"])
        return True

    def save_hparams(self, directory: str) -> None:
        with open(os.path.join(directory, "hp.yaml"), "w") as f:
            yaml.dump(self.hp.to_dict(), f)
