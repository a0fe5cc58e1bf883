"""Data-parallel gradient exchange: one process per GPU, RCCL over xGMI.

The reference gets this from Lightning's ``strategy="ddp"`` (scripts/train.py:93-95),
i.e. torch DDP hooks that all-reduce 25 MB buckets on EVERY micro-batch
(manual optimisation without ``no_sync``, trainers/speech/lvtr.py:46,131,150).
This reducer is written for the MI355X node instead:

* gradients live in a few large flat fp32 buckets (default 50 MiB = one
  Transformer layer); ``param.grad`` is a view into its bucket,
  so autograd accumulates in place and no copy-in / copy-out happens;
* buckets are filled in reverse registration order (the order backward
  finishes them); as soon as the last gradient of a bucket has been
  accumulated -- and only on the LAST micro-batch of an accumulation window --
  its all-reduce is launched on a dedicated communication stream, overlapping
  the rest of backward;
* the mean over ranks is taken by the collective (``ReduceOp.AVG``) -- xGMI is
  point-to-point, so fewer, larger messages are preferred over DDP's 25 MB;
* ``finish()`` makes the compute stream wait for the communication stream
  before the optimizer step.

Works with any ``torch.distributed`` backend (``nccl`` = RCCL on ROCm; ``gloo``
in the CPU tests).  With ``world_size == 1`` it degenerates to the bucket views.
"""
from __future__ import annotations

import os

from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mb: float = 50.0,
                 overlap: bool = True, group: Optional[dist.ProcessGroup] = None, comm: str = "torch",
                 boundaries=(), wire_dtype: str = "fp32"):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        # "torch": torch.distributed all_reduce (nccl = RCCL, or gloo in the CPU / one-device tests);
        # "abi": RCCL through the library's own C entry points (vg_comm_init / vg_allreduce_bucket)
        if comm not in ("torch", "abi"):
            raise ValueError(f"hip.comm must be 'torch' or 'abi', got {comm!r}")
        self.comm = comm
        # hip.comm_dtype (round 6): "fp32" (default) sends the flat fp32 buckets as they are -- 908 MB per optimizer step
        # at the full configuration; "bf16" sends a bf16 copy (454 MB): on the communication stream the bucket is cast
        # into a bf16 staging buffer, that buffer is all-reduced (the collective adds and averages in bf16), and the
        # result is written back over the fp32 bucket, which the optimizer reads as before.  xGMI is point-to-point
        # (7 links x ~153 GB/s per GPU): a ring all-reduce moves 2 (N - 1) / N of the bytes over ONE link direction per
        # step, so at N = 8 the fp32 wire is >= 10 ms of link time per step if RCCL ends up on one ring and the bf16
        # wire half of it.  Cost: every rank's gradient is rounded to 8 significant bits before the sum and the partial
        # sums again (tests/test_dp_gloo.py bounds the difference from the fp32 wire); two more elementwise launches
        # per bucket on the communication stream.  Opt-in.
        if wire_dtype not in ("fp32", "bf16"):
            raise ValueError(f"hip.comm_dtype must be 'fp32' or 'bf16', got {wire_dtype!r}")
        self.wire_dtype = wire_dtype
        # Collectives run when there is more than one rank -- or, VG_DP_SINGLE_RANK=1, on a ONE-rank communicator: the
        # whole data-parallel step (segmented hipGraph replay, bucket launches on the communication stream, per-bucket
        # optimizer waits) through real RCCL on a one-GPU box.  The average over one rank is the identity, so the run
        # must end where the plain single-GPU step ends (tests/test_dp_gpu.py), and its timing is the step's cost of
        # the machinery without the wire (bench.py --single-rank-rccl).
        self.exchange = self.world > 1 or os.environ.get("VG_DP_SINGLE_RANK", "0") == "1"
        if self.exchange and self.world == 1 and comm == "torch" and not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("VG_DP_SINGLE_RANK=1 with hip.comm=torch needs an initialised one-rank process group")
        if comm == "abi" and self.exchange:
            from hipvg import comm as vg_comm
            rank = dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0
            vg_comm.init(rank, self.world, group)
        self.overlap = overlap
        self.sync_now = True           # set False on non-final micro-batches
        self._epoch = 0                # backward passes announced through new_backward()
        self._expected = {}            # signature -> {id(param): reports per backward pass}
        self._calibrating, self._in_pass, self._surprise, self._sig = False, False, None, None
        self._counts, self._remaining = {}, {}
        self._next = 0                 # buckets [0, _next) are on the wire: launches go out in bucket-index order
        # measurement hooks (bench.py): event pairs around every collective on the stream it runs on, and a switch
        # that skips the collectives (the same step without its exchange: what is left is the exposed part)
        self.time_collectives = False
        self.stub_collectives = False
        self._comm_events = []
        # launch log (tests, bench): one (bucket index, phase) per collective put on the wire since the last finish();
        # `phase` is whatever the caller set before it launched more work -- the trainer's segmented hipGraph replay
        # sets it to the index of the graph piece about to be replayed, so the log shows WHICH buckets were already
        # travelling when the last piece of backward started (the property a bucket-order bug silently breaks)
        self.phase = 0
        self.launch_log = []
        self.last_launch_log = []
        dev = self.params[0].device
        self.comm_stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        cu_mask = int(os.environ.get("VG_COMM_CU_MASK", "0"))
        if cu_mask > 0 and dev.type == "cuda" and self.exchange:
            # lab: the communication stream confined to `cu_mask` CUs; measured on a one-rank RCCL
            # communicator in profiles/r06/labs/ -- not the default
            from hipvg.comm import masked_stream
            self.comm_stream = masked_stream(dev, cu_mask)
        limit = int(bucket_mb * (1 << 20) / 4)
        # reverse order: the last-registered parameters receive gradients first.  Every parameter starts
        # on a 256-element boundary of its bucket (1 KiB): the flat optimizer kernel (vg_adamw) works on
        # 256-element chunks that must not straddle two parameters (per-group lr / weight decay).
        self.buckets: List[dict] = []
        cur, cur_n = [], 0
        breaks = {id(p) for p in boundaries}     # a bucket ends before each of these (walking in reverse order)
        for p in reversed(self.params):
            n = self._padded(p.numel())
            if cur and (cur_n + n > limit or id(p) in breaks):
                self._close(cur, cur_n, dev)
                cur, cur_n = [], 0
            cur.append(p)
            cur_n += n
        if cur:
            self._close(cur, cur_n, dev)
        self._handles = []
        for bi, b in enumerate(self.buckets):
            for p in b["params"]:
                hook = self._make_hook(bi)
                p.register_post_accumulate_grad_hook(hook)
                p._vg_grad_hooks = [hook]      # fired by hipvg's gradient sink (bypasses AccumulateGrad)

    ALIGN = 256

    @classmethod
    def _padded(cls, n: int) -> int:
        return (n + cls.ALIGN - 1) // cls.ALIGN * cls.ALIGN

    def _close(self, plist, n, dev):
        flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off, offsets = 0, []
        for p in plist:
            p.grad = flat[off: off + p.numel()].view_as(p)
            offsets.append(off)
            off += self._padded(p.numel())
        wire = torch.empty(n, dtype=torch.bfloat16, device=dev) if (self.wire_dtype == "bf16" and self.exchange) else None
        self.buckets.append(dict(params=plist, offsets=offsets, flat=flat, wire=wire, pending=len(plist), need=len(plist)))

    def new_backward(self, signature=None) -> None:
        """Call before every backward pass.  A parameter can report "gradient ready" several times in one pass: once
        per use from hipvg's gradient sink (a sunk parameter used twice in one forward reports after EACH
        contribution) and once from autograd, whose post-accumulate hooks also run when a custom backward returned
        None for that parameter (torch >= 2.x).  The reducer therefore learns how many reports each parameter
        makes: the first pass of every ``signature`` (anything hashable that identifies the structure of the step,
        e.g. the batch's keys) only counts -- nothing is launched before ``flush()`` -- and later passes launch a
        bucket when every parameter has made its LAST expected report.  Parameters that made none (no gradient in
        this structure) are not waited for; their buckets go on the wire in ``flush()``.  Without any call to this
        method every report counts once (plain autograd modules)."""
        self._end_pass()
        self._epoch += 1
        self._sig = signature
        exp = self._expected.get(signature)
        self._calibrating = exp is None
        self._in_pass = True
        if self._calibrating:
            self._counts = {}
        else:
            self._remaining = dict(exp)
            for b in self.buckets:
                b["pending"] = sum(1 for p in b["params"] if exp.get(id(p), 0) > 0)
                b["ready"] = self.sync_now and b["pending"] == 0 and not b.get("launched", False)

    def _end_pass(self) -> None:
        if getattr(self, "_in_pass", False):
            self._in_pass = False
            if self._calibrating:
                self._expected[self._sig] = self._counts
            if self._surprise is not None:
                name, self._surprise = self._surprise, None
                self._expected.pop(self._sig, None)
                raise RuntimeError(
                    "GradReducer: a parameter reported 'gradient ready' more often than in the first pass with this "
                    f"signature ({self._sig!r}; parameter of shape {name}); its bucket may have been all-reduced before "
                    "the last contribution.  Pass a signature that distinguishes the two step structures.")

    def _make_hook(self, bi: int):
        def hook(param):
            # a report arrives on the stream that wrote the gradient -- the main stream, or the side branch of the step
            # (hipvg.functional.fork_side: autograd runs a node's backward on the stream of its forward): the bucket's
            # collective has to wait for every stream that wrote into it, not only for the one that completes it
            if param.is_cuda:
                st = torch.cuda.current_stream(param.device)
                self.buckets[bi].setdefault("streams", {})[st.cuda_stream] = st
            if not self._epoch:            # plain autograd use: every report counts
                b = self.buckets[bi]
                b["pending"] -= 1
                if b["pending"] == 0:
                    b["pending"] = b["need"]
                    if self.sync_now:          # a non-final micro-batch completes the bucket again later
                        b["ready"] = True
                        self._drain()
                return
            key = id(param)
            if self._calibrating:
                self._counts[key] = self._counts.get(key, 0) + 1
                return
            rem = self._remaining.get(key, 0)
            if rem <= 0:
                if self.sync_now and self.exchange:
                    self._surprise = tuple(param.shape)
                return
            self._remaining[key] = rem - 1
            if rem == 1:
                b = self.buckets[bi]
                b["pending"] -= 1
                if b["pending"] == 0 and self.sync_now:
                    b["ready"] = True
                    self._drain()
        return hook

    def _drain(self) -> None:
        """Launch every complete bucket up to the first incomplete one.  Collectives leave in bucket-INDEX order on
        every rank whatever order backward completed them in and whether or not this rank is still counting reports
        for the step's signature (a counting rank launches nothing before ``flush()``, which also walks the buckets
        in index order) -- so ranks that disagree about the calibration state still issue the same sequence of
        collectives."""
        if not (self.sync_now and self.exchange):
            return
        while self._next < len(self.buckets):
            b = self.buckets[self._next]
            if b.get("launched", False):
                self._next += 1
            elif b.get("ready", False):
                self._launch(b)
                self._next += 1
            else:
                break

    def _timed_allreduce(self, flat, wire=None):
        if self.stub_collectives:
            return None
        if wire is not None:
            # bf16 wire: cast, exchange, write back -- all on the stream this runs on (the communication stream); the
            # handle of an asynchronous collective is waited for HERE (a stream-level wait) so that the write-back is
            # ordered behind it, and the bucket then has no handle: wait_bucket() waits for the stream
            wire.copy_(flat)
            a = b = None
            if self.time_collectives and flat.is_cuda:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
            h = self._allreduce(wire)
            if h is not None:
                h.wait()
            if a is not None:
                b.record()
                self._comm_events.append((a, b, wire.numel() * wire.element_size()))
            flat.copy_(wire)
            return None
        if not (self.time_collectives and flat.is_cuda):
            return self._allreduce(flat)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        h = self._allreduce(flat)
        b.record()
        self._comm_events.append((a, b, flat.numel() * flat.element_size()))
        return h

    def comm_stats(self, reset: bool = True) -> dict:
        """Summed duration (ms, events on the communication stream) and bytes of the collectives recorded since the
        last call; the duration of overlapping collectives on one stream adds up, so this is wire time, not
        exposed time."""
        ms = sum(a.elapsed_time(b) for a, b, _ in self._comm_events)
        out = {"allreduce_ms": ms, "allreduce_bytes": sum(n for _, _, n in self._comm_events),
               "collectives": len(self._comm_events)}
        if reset:
            self._comm_events = []
        return out

    def communicator_ranks(self) -> int:
        """Size of the communicator the collectives actually run on (not the launcher's environment)."""
        if not self.exchange:
            return 1
        if self.comm == "abi":
            import hipvg
            return int(hipvg.lib().vg_comm_world())
        return int(dist.get_world_size(self.group))

    def _launch(self, b):
        writers = list(b.pop("streams", {}).values()) if b["flat"].is_cuda else []
        if self.comm_stream is not None and self.overlap:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            for st in writers:
                self.comm_stream.wait_stream(st)
            with torch.cuda.stream(self.comm_stream):
                h = self._timed_allreduce(b["flat"], b.get("wire"))
        else:
            cur = torch.cuda.current_stream() if b["flat"].is_cuda else None
            for st in writers:
                if st.cuda_stream != cur.cuda_stream:
                    cur.wait_stream(st)
            h = self._timed_allreduce(b["flat"], b.get("wire"))
        b["handle"], b["launched"] = h, True
        self.launch_log.append((next(i for i, x in enumerate(self.buckets) if x is b), self.phase))
        self._handles.append(h)

    def wait_bucket(self, i: int) -> None:
        """Make the current stream wait for bucket ``i``'s all-reduce only (the optimizer can then update that
        bucket while later buckets are still on the wire)."""
        if not self.exchange:
            return
        b = self.buckets[i]
        if not b.get("launched", False):
            return
        h = b.get("handle")
        if h is not None:
            h.wait()                       # stream-level wait for an asynchronous collective
        elif self.comm_stream is not None and self.overlap:
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    def _allreduce(self, flat):
        if self.comm == "abi":          # RCCL through the C ABI: stream-ordered on the current (comm) stream
            from hipvg import comm as vg_comm
            vg_comm.all_reduce_(flat, average=True)
            return None
        backend = dist.get_backend(self.group)
        if backend == "nccl":
            return dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group, async_op=True)
        h = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        h.wait()
        flat.div_(self.world)          # (a bf16 wire divides in bf16 too: exact for a power-of-two world)
        return None

    def reduce_all(self) -> None:
        """Launch the all-reduce of every bucket not yet on the wire (used when backward ran inside a hipGraph)."""
        if self.exchange:
            for b in self.buckets:
                if not b.get("launched", False):
                    self._launch(b)

    def reduce_buckets(self, indices) -> None:
        """Launch the all-reduce of the given buckets only (their gradients are final although backward is not:
        segmented hipGraph replay)."""
        if self.exchange:
            for i in indices:
                if not self.buckets[i].get("launched", False):
                    self._launch(self.buckets[i])

    def buckets_within(self, params) -> List[int]:
        """Indices of the buckets all of whose parameters are in ``params``."""
        ids = {id(p) for p in params}
        return [i for i, b in enumerate(self.buckets) if all(id(p) in ids for p in b["params"])]

    def flush(self) -> None:
        """End of the window's last backward: every bucket not yet on the wire is launched now (the counting pass of
        a new signature, buckets whose parameters received no gradient, hipGraph replays)."""
        self._end_pass()
        if self.sync_now and self.exchange:
            self.reduce_all()

    def finish(self) -> None:
        """Call after the last backward of the window, before ``optimizer.step()``."""
        self.flush()
        for h in self._handles:
            if h is not None:
                h.wait()
        self._handles.clear()
        self.last_launch_log, self.launch_log, self.phase = self.launch_log, [], 0
        for b in self.buckets:
            b["handle"], b["launched"], b["ready"] = None, False, False
            b["pending"] = b["need"]
        self._next = 0
        if self.comm_stream is not None and self.overlap and self.exchange:
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    def zero_grad(self) -> None:
        """Zero the buckets (keeps the views alive -- never set grads to None)."""
        for b in self.buckets:
            b["flat"].zero_()
