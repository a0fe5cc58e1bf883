"""Autograd-aware wrappers over the C ABI (include/vaegslm_hip.h).

Each ``torch.autograd.Function`` here owns a hand-written backward that calls
the matching HIP kernels; none of them falls back to ATen math.  Conventions:

* activations are 2-D ``[M, C]`` row-major with rows = frames in (b, t) order
  (``M = B * T``), in the compute dtype (``hipvg.compute_dtype()``);
* ``lengths`` is an int32 device tensor ``[B]`` (right-padded prefix masks,
  utils/tensormask.py:45-54 of the reference) or ``None``;
* parameters stay fp32 (master copy); bf16 shadows are cached per parameter
  version, so they are re-cast once per optimizer step, not per micro-batch.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional, Tuple

import torch

from . import (ACT_DERIV_U8, ACT_GELU, ACT_IDS, ACT_NONE, ACT_RELU, ACT_SAVE_DERIV, ACT_SILU, ACT_STORED, GemmDesc, check,
               dtype_id, lib, ptr, stream)

Tensor = torch.Tensor

# ---------------------------------------------------------------- switches (A/B measurements; defaults = fastest measured)
def _flag(name: str, default: str) -> bool:
    return os.environ.get(name, default) != "0"


# bias gradient inside the wgrad launch (vg_gemm colsum_out): measured 2 % slower end-to-end than separate
# column-sum launches on MI355X (the conditional MFMA inside the K loop costs the wgrad ~8 us)
_FUSE_BIAS_GRAD = _flag("VG_FUSE_BIAS_GRAD", "0")
_STORED_DERIV = _flag("VG_STORED_DERIV", "1")   # forward stores act'(u): the backward epilogue is one multiply
# round 6: in bf16 the FFN's stored GELU' is ONE BYTE per element (VG_ACT_DERIV_U8, include/vaegslm_hip.h: 256 codes of
# step 0.005 over [-0.13, 1.145], absolute error <= 0.0025 -- what bf16 keeps near 1): half the bytes of the second
# M x 4096 stream of both FFN-in launches.  VG_DERIV_U8=0 keeps the bf16 derivative (A/B); fp32 keeps fp32.
_DERIV_U8 = _flag("VG_DERIV_U8", "1")


def set_deriv_u8(on: bool) -> bool:
    """Switch the 8-bit stored GELU derivative of the bf16 FFN (tests, A/B runs); returns the previous setting."""
    global _DERIV_U8
    prev, _DERIV_U8 = _DERIV_U8, bool(on)
    return prev


def _gelu_deriv_buffer(M: int, F_: int, dt, device):
    """(buffer for the stored GELU derivative, act flags of the forward launch, dact flags of the dgrad launch)"""
    if _DERIV_U8 and dt == torch.bfloat16 and F_ % 8 == 0:
        return (torch.empty((M, F_), dtype=torch.uint8, device=device), ACT_GELU | ACT_SAVE_DERIV | ACT_DERIV_U8,
                ACT_STORED | ACT_DERIV_U8)
    return torch.empty((M, F_), dtype=dt, device=device), ACT_GELU | ACT_SAVE_DERIV, ACT_STORED
_COLSUM_MULTI = _flag("VG_COLSUM_MULTI", "1")   # the small column sums of a backward node in one launch
_COLSUM_BIG = _flag("VG_COLSUM_BIG", "1")     # a layer's large bias column sums (first stage) in one launch
_COLPART = _flag("VG_COLPART", "1")             # dgrad launches also reduce their result per row tile (bias gradients)
# in-launch slab reduction of split-K weight gradients instead of fp32 atomics: measured SLOWER on these tiles
# (64 KiB of slab per slice; write-through slabs + ticket: 80 vs 67 us per wgrad launch, 289k vs 305k tokens/s;
# with an agent-scope release fence per block instead of write-through stores: 90 us)
_SPLIT_SLABS = _flag("VG_SPLIT_SLABS", "0")
_PH_WGRAD = _flag("VG_PH_WGRAD", "1")           # weight gradients on 256x256 ring tiles (split sized for one block per CU)
_GROUP_MIN_TILES = int(os.environ.get("VG_GROUP_MIN_TILES", "24"))   # fewer 256x256 tiles than this: one launch per product
_TRACE_TN = _flag("VG_TRACE_TN", "0")
_PH_GROUP = _flag("VG_PH_GROUP", "1")           # a layer's four weight gradients as one grouped launch
_GRAD_SINK = _flag("VG_GRAD_SINK", "1")         # wgrad / column sums write straight into param.grad
# sunk weight gradients of a layer on a second stream (parallel graph branch): measured slower, 55.4 vs 53.2 ms
# per step -- both branches are machine-filling GEMMs and only thrash each other's L2 / LDS residency
_WGRAD_STREAM = _flag("VG_WGRAD_STREAM", "0")

# ---------------------------------------------------------------- weight shadows

def shadow(weight: Tensor, dtype: torch.dtype) -> Tensor:
    """fp32 master parameter -> tensor in the compute dtype (cached by version)."""
    if weight.dtype == dtype:
        return weight.detach()
    flat = getattr(weight, "_vg_flat_shadow", None)      # kept current by the flat optimizer step (vg_adamw)
    if flat is not None and flat.dtype == dtype:
        return flat
    ent = getattr(weight, "_vg_shadow", None)
    ver = weight._version
    if ent is not None and ent[0] == ver and ent[1].dtype == dtype:
        return ent[1]
    assert weight.dtype == torch.float32 and dtype == torch.bfloat16
    w = weight.detach()
    if not w.is_contiguous():
        w = w.contiguous()
    # re-cast into the SAME tensor when the parameter changed: launches captured in a hipGraph keep
    # reading a valid, current copy (see refresh_shadows)
    out = ent[1] if ent is not None and ent[1].shape == w.shape and ent[1].dtype == dtype \
        else torch.empty(w.shape, dtype=dtype, device=w.device)
    check(lib().vg_cast_f32_to_bf16(ptr(w), ptr(out), w.numel(), stream()), "vg_cast_f32_to_bf16")
    weight._vg_shadow = (ver, out)
    return out


def refresh_shadows(params) -> None:
    """Re-cast, in place, every cached bf16 weight copy whose master parameter has changed.  Call after an
    optimizer step that is not the flat one when micro-steps are replayed from a hipGraph: the captured
    GEMMs read the cached copies by address and never run the lazy cast of :func:`shadow`."""
    for p in params:
        ent = getattr(p, "_vg_shadow", None)
        if ent is not None and ent[0] != p._version:
            shadow(p, ent[1].dtype)


# ---------------------------------------------------------------- raw ops
_SPLIT_WS = {}


def _split_workspace(device):
    """Scratch of the in-launch split-K reduction (fp32 slabs + tile counters), per device and per SIDE stream.  Launches
    on one stream serialise, so they share it; 64 MiB covers every weight gradient of the full config (the side branch
    of the step -- side_stream() -- has its own, smaller one: its launches overlap the main stream's)."""
    side = on_side_stream()
    key = (device, stream()) if side else device
    ent = _SPLIT_WS.get(key)
    if ent is None:
        ent = (torch.empty((2 << 20) if side else (16 << 20), dtype=torch.float32, device=device),
               torch.zeros(4096, dtype=torch.int32, device=device))
        _SPLIT_WS[key] = ent
    return ent


# ---------------------------------------------------------------- the side branch of the training step
# The utterance encoder and the diffusion decoder's time-embedding MLPs are ~100 launches of a few microseconds each,
# forward and backward, that depend on nothing the main chain computes until the UNet reads their results: on a second
# HIP stream (forked and joined with stream waits, so a hipGraph capture records two parallel branches) they run under
# the Transformer stack instead of in front of the UNet.  Autograd runs a node's backward on the stream of its forward,
# so their backward runs under the Transformer stack's backward as well.  What is shared with the main stream is kept
# apart: the split-K scratch (above), the deferral queues (a product issued from the side stream launches at once), and
# "gradient ready" reports (delivered on the main stream after it has waited for the side stream).
_SIDE = {"streams": {}, "ids": {}}
_SIDE_DEFER = [os.environ.get("VG_SIDE_UNET", "0") == "1"]    # weight-gradient products issued on the side branch are queued too


def side_defer(on: bool) -> None:
    """With the diffusion decoder on the side branch (hip.side_unet) its weight-gradient products are queued like the main
    chain's, in queues of their own: a queue holds products of ONE stream, a mid-pass flush launches on the stream that
    enqueues, and the flush at the end of the backward piece comes after autograd has joined the streams."""
    _SIDE_DEFER[0] = bool(on)


def side_stream(device) -> "torch.cuda.Stream":
    dev = torch.device(device)
    st = _SIDE["streams"].get(dev)
    if st is None:
        st = torch.cuda.Stream(device=dev)
        _SIDE["streams"][dev] = st
        _SIDE["ids"][st.cuda_stream] = st
    return st


def on_side_stream() -> bool:
    return bool(_SIDE["ids"]) and torch.cuda.is_available() and torch.cuda.current_stream().cuda_stream in _SIDE["ids"]


_MAIN_OF_SIDE = {}

# ---- the stream a hipGraph with parallel branches is launched on.  ROCm 7.0's hip::Graph::UpdateStreams (every hipGraphLaunch
# of an exec with more than one branch) hands branch i the next of the exec's max_streams internal streams whose virtual device
# differs from the launch stream's, and walks past the end of that vector when two of them sit where the launch stream sits:
# SIGSEGV inside hipGraphLaunch.  Streams share the device's few pooled hardware queues (least-loaded first), so it takes a
# long-lived process with an uneven stream population -- the GPU test suite hit it at its 306th test, deterministically for a
# given tree, and tools/lab/hipgraph_queue_collision.py reproduces it in 40 lines; a fresh process (bench.py, a training run:
# torch's 32 pooled streams spread evenly, then the graph's) was never seen to.  The internal streams are ordinary streams on
# the pooled normal-priority queues; a launch stream on a queue OUTSIDE that pool can never match one, so the walk takes the
# first max_streams - 1 streams and stops.  Two kinds of stream have such a queue: a high-priority stream (its own pool) and a
# stream created with a full CU mask (a queue of its own).  Both are safe and both have a price: every cross-stream wait that
# involves them blocks the HOST until the awaited work is done (the traces under rocprofv3 show the same GPU time; the host no
# longer runs ahead).  A step that is one graph launch does not notice (27.71 - 27.89 ms either way); steps with eager launches
# and stream traffic between graph pieces do: two micro-steps per optimizer step 34.2 -> 43.6 ms, one-rank RCCL 28.8 -> 40.3 ms
# (profiles/r06/labs/hipgraph_launch_stream.txt).  VG_LAUNCH_STREAM = normal (default: an ordinary created stream, what every
# bench line of this repository was measured on) | prio | mask.  tests/conftest.py selects `prio` for the long-lived pytest
# process, which is exactly the population the hazard needs.
_LAUNCH_STREAMS = {}
_LAUNCH_IDS = set()


def launch_stream_kind() -> str:
    kind = os.environ.get("VG_LAUNCH_STREAM", "normal").lower()
    if kind not in ("mask", "prio", "normal"):
        raise ValueError(f"VG_LAUNCH_STREAM={kind!r}: expected mask, prio or normal")
    return kind


def is_launch_stream(st) -> bool:
    """Is ``st`` one of this process's launch streams (of whatever kind VG_LAUNCH_STREAM selected)?"""
    return st.cuda_stream in _LAUNCH_IDS


def graph_launch_stream(device) -> "torch.cuda.Stream":
    """The calling thread's current stream if it is one of this process's launch streams, else the launch stream of ``device``
    (created once per process)."""
    dev = torch.device(device)
    cur = torch.cuda.current_stream(dev)
    if cur.cuda_stream in _LAUNCH_IDS:
        return cur
    st = _LAUNCH_STREAMS.get(dev)
    if st is None:
        kind = launch_stream_kind()
        if kind == "mask":
            from hipvg.comm import masked_stream
            total = torch.cuda.get_device_properties(dev).multi_processor_count
            st = masked_stream(dev, total, total)
        else:
            st = torch.cuda.Stream(device=dev, priority=-1 if kind == "prio" else 0)
        _LAUNCH_STREAMS[dev] = st
        _LAUNCH_IDS.add(st.cuda_stream)
    return st


def fork_side(device):
    """-> (side, main): the side stream now waits for everything queued on the current stream."""
    main = torch.cuda.current_stream(device)
    side = side_stream(device)
    side.wait_stream(main)
    _MAIN_OF_SIDE[side.cuda_stream] = main
    return side, main


def join_side(side, main, tensors=()) -> None:
    """The current (main) stream waits for the side stream; ``tensors`` were allocated there and are read here."""
    main.wait_stream(side)
    for t in tensors:
        if torch.is_tensor(t) and t.is_cuda:
            t.record_stream(main)



def gemm(A: Tensor, B: Tensor, M: int, N: int, K: int, *, a_tr=False, b_tr=False,
         out: Optional[Tensor] = None, out_f32=False, bias: Optional[Tensor] = None,
         residual: Optional[Tensor] = None, aux_in: Optional[Tensor] = None,
         aux_out: Optional[Tensor] = None, lengths: Optional[Tensor] = None, T: int = 0,
         act: int = ACT_NONE, dact: int = ACT_NONE, accumulate=False, split_k: int = 1,
         alpha: float = 1.0, tile_cfg: int = 0, pre_add: Optional[Tensor] = None,
         colsum_out: Optional[Tensor] = None, colpart: Optional[list] = None) -> Tensor:
    assert A.dim() == 2 and B.dim() == 2 and A.stride(1) == 1 and B.stride(1) == 1
    assert A.dtype == B.dtype
    if _TRACE_TN and a_tr and b_tr:      # who still launches a weight gradient of its own (VG_TRACE_TN=1, eager runs)
        import sys
        import traceback
        fr = [f"{f.name}:{f.lineno}" for f in traceback.extract_stack(limit=7)[:-1]]
        print(f"[vg_tn] M={M} N={N} K={K} split={split_k} dtype={A.dtype} <- {' < '.join(reversed(fr))}", file=sys.stderr)
    if A.data_ptr() % 16:      # e.g. a channel slice of a single frame: the kernels need 16-byte aligned operands
        A = A.clone()
    if B.data_ptr() % 16:
        B = B.clone()
    if out is None:
        odt = torch.float32 if out_f32 else A.dtype
        out = (torch.zeros if split_k > 1 else torch.empty)((M, N), dtype=odt, device=A.device)
    d = GemmDesc()
    d.A, d.B, d.C = ptr(A), ptr(B), ptr(out)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc = A.stride(0), B.stride(0), out.stride(0)
    d.a_tr, d.b_tr, d.dtype = int(a_tr), int(b_tr), dtype_id(A.dtype)
    d.bias, d.residual = ptr(bias), ptr(residual)
    d.aux_in, d.aux_out = ptr(aux_in), ptr(aux_out)
    d.lengths, d.T = ptr(lengths), int(T)
    d.act, d.dact = act, dact
    d.out_f32 = int(out.dtype == torch.float32)
    d.accumulate, d.split_k, d.alpha = int(accumulate), int(split_k), float(alpha)
    d.tile_cfg = int(tile_cfg)
    d.pre_add = ptr(pre_add)
    d.colsum_out = ptr(colsum_out)     # fp32 [M], += row sums of A (bias gradient of a wgrad launch)
    if split_k > 1 and _SPLIT_SLABS and A.is_cuda:
        ws, cnt = _split_workspace(A.device)
        d.split_ws, d.split_cnt, d.split_ws_floats = ptr(ws), ptr(cnt), ws.numel()
    part = None
    if colpart is not None and split_k <= 1:
        # per-row-tile column sums of the result from the same launch (the caller reduces the few rows): only
        # on the LDS-DMA path, whose tile height the library reports for this descriptor
        # (the library says how many row-tiles the call writes: a product it splits over two launches -- whole rounds of
        # 256-row tiles + the remaining row band on another tile shape -- has more than ceil(M / tile rows))
        nrows = lib().vg_gemm_colpart_rows(C.byref(d))
        if nrows > 0:
            part = torch.empty((nrows, N), dtype=torch.float32, device=A.device)
            d.colpart = ptr(part)
    if colpart is not None:
        colpart.append(part)
    check(lib().vg_gemm(C.byref(d), stream()), "vg_gemm")
    return out


def wgrad_splits(rows_out: int, cols_out: int, k_red: int, dtype: torch.dtype) -> int:
    """Split the reduction (frame) dimension of a weight-gradient GEMM so that the grid fills the 256 CUs (partial
    sums are combined with fp32 atomics).  The exact-f32 kernel is only split for the tiny latent-side weights (a
    single tile would otherwise walk all 8000 frames serially)."""
    tiles = ((rows_out + 127) // 128) * ((cols_out + 127) // 128)
    if dtype == torch.float32:
        return 1 if tiles > 4 else max(1, min(32, k_red // 256))
    tiles256 = ((rows_out + 255) // 256) * ((cols_out + 255) // 256)
    if _PH_WGRAD and k_red % 64 == 0 and tiles256 >= 12 and k_red >= 2048:
        # 256x256 ring tiles (vg_gemm_ph.hip): one block per CU; every slice keeps >= 16 whole K tiles
        s = max(1, min(10, round(256 / tiles256), k_red // 1024))
        while s > 1 and (-(-k_red // s) + 63) // 64 * 64 * (s - 1) >= k_red:      # the library's rounding must keep s slices
            s -= 1
        return s
    # 128x128 tiles, two blocks per CU; measured on MI355X (tools/wgrad_split_sweep.py): 2 for the big layer
    # weights, ~6 for 64-tile outputs, up to 12 for the small ones
    if tiles >= 128:
        s = 2
    elif tiles >= 48:
        s = 6 if k_red < 12000 else 8      # 16,000 frames (coalesced accumulation window): 67.7 vs 74.5 us
    else:
        s = min(12, max(1, 768 // max(tiles, 1)))
    return max(1, min(s, k_red // 256))


GROUP_MAX = 48        # VG_GROUP_MAX of include/vaegslm_hip.h: products per vg_gemm_grouped launch

# "Fresh" backward passes (round 4).  The trainer opens every backward pass with begin_backward_pass(fresh): fresh = the
# gradient buffers hold zeros (first pass since the optimizer / zero_grad cleared them).  In such a pass the FIRST
# product written into a gradient region by a grouped launch may STORE its whole-K tiles instead of read-modify-write
# (accumulate = 0 of vg_gemm_grouped: 4 bytes per parameter less traffic).  Every path that writes a weight gradient
# records its region here, so that a second contribution to the same region in the same pass (a module applied twice, a
# product that took another route first) accumulates.
_WPASS = {"fresh": False, "written": set(), "auto": set(), "epoch": 0}
_WGRAD_STORE = _flag("VG_WGRAD_STORE", "1")


def begin_backward_pass(fresh: bool) -> None:
    _WPASS["fresh"] = bool(fresh) and _WGRAD_STORE
    _WPASS["written"] = set()
    _WPASS["auto"] = set()
    _WPASS["epoch"] += 1


def end_backward_pass() -> None:
    _WPASS["fresh"] = False
    _WPASS["written"] = set()
    _WPASS["auto"] = set()


def write_epoch() -> int:
    """Counts everything that may have written a gradient through this library (every backward pass opened, every
    "gradient ready" report of the sink, every dense autograd gradient that reached a watched parameter).  Whoever
    clears the gradient buffers remembers the value; the buffers still hold zeros only while it has not moved.  This is
    how the trainer decides `fresh` -- from the state of the buffers, not from a batch index."""
    return _WPASS["epoch"]


def note_autograd_write(p) -> None:
    """post-accumulate-grad hook of a parameter the sink may also write: a dense autograd gradient has been ADDED to
    ``p.grad``, so a grouped launch that lands later in this pass must accumulate, not store."""
    _WPASS["auto"].add(id(p))
    _WPASS["epoch"] += 1


def _wgrad_region(g: Tensor):
    return (g.data_ptr(), g.shape[0], g.shape[1])


def _wgrad_grad_view(w, x, col0):
    N = w.shape[0]
    return _grad_buffer(w).view(N, -1)[:, col0:col0 + x.shape[1]]


def _wgrad_item_ok(w, dy, x, col0) -> bool:
    """Can ``w.grad[:, col0:col0 + K] += dy^T x`` be one product of a vg_gemm_grouped launch?"""
    N, K, M = w.shape[0], x.shape[1], x.shape[0]
    return (dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and M % 64 == 0 and M >= 1024
            and N % 8 == 0 and K % 8 == 0 and col0 % 4 == 0 and dy.stride(1) == 1 and x.stride(1) == 1 and w.is_contiguous()
            and dy.data_ptr() % 16 == 0 and x.data_ptr() % 16 == 0 and dy.shape[0] == M and dy.shape[1] == N)


def _wgrad_tiles(w, x) -> int:
    return ((w.shape[0] + 255) // 256) * ((x.shape[1] + 255) // 256)


def _launch_wgrad_items(items) -> None:
    """One vg_gemm_grouped launch for ``items`` (all qualified, same reduction length, at most GROUP_MAX), or one
    split-K launch each when they are too few tiles to be worth a persistent grid."""
    total = sum(_wgrad_tiles(w, x) for w, _, x, _ in items)
    if _TRACE_TN:
        import sys
        print(f"[vg_tn] group of {len(items)} products, {total} tiles, K={items[0][2].shape[0]}: "
              + " ".join(f"{w.shape[0]}x{x.shape[1]}" for w, _, x, _ in items), file=sys.stderr)
    if not _PH_GROUP or total < _GROUP_MIN_TILES:
        for w, dy, x, col0 in items:
            N, K, M = w.shape[0], x.shape[1], x.shape[0]
            sp = wgrad_splits(N, K, M, x.dtype)
            g = _wgrad_grad_view(w, x, col0)
            _WPASS["written"].add(_wgrad_region(g))
            gemm(dy, x, N, K, M, a_tr=True, b_tr=True, out=g, split_k=sp, accumulate=(sp == 1))
        return
    descs = (GemmDesc * len(items))()
    for d, (w, dy, x, col0) in zip(descs, items):
        g = _wgrad_grad_view(w, x, col0)
        region = _wgrad_region(g)
        # zeros underneath (nothing of this pass -- library or autograd -- has written there): whole-K tiles may store
        store = _WPASS["fresh"] and region not in _WPASS["written"] and id(w) not in _WPASS["auto"]
        _WPASS["written"].add(region)
        d.A, d.B, d.C = ptr(dy), ptr(x), ptr(g)
        d.M, d.N, d.K = w.shape[0], x.shape[1], x.shape[0]
        d.lda, d.ldb, d.ldc = dy.stride(0), x.stride(0), g.stride(0)
        d.a_tr, d.b_tr, d.dtype = 1, 1, dtype_id(torch.bfloat16)
        d.out_f32, d.accumulate, d.split_k, d.alpha = 1, 0 if store else 1, 1, 1.0
    check(lib().vg_gemm_grouped(descs, len(items), stream()), "vg_gemm_grouped")


# Deferred weight gradients (round 4).  Nothing reads a weight gradient before the optimizer (or the bucket exchange),
# so between defer_vec_grads(True) and the flush at the end of the backward piece the qualified dW += dY^T X products
# are queued -- their operands kept alive -- and leave as FEW, LARGE grouped launches: the four products of a Transformer
# layer are 192 tiles of 256x256 (three quarters of a round of 256 CUs: a head / tail split with fp32 atomics), those
# of four layers are 768 = three whole rounds of full-K tiles that add with plain 16-byte accesses; the conv blocks'
# two or three products each (32-40 tiles, eight-way split with 67 MB of atomics per launch) and the ~20 small split-12
# launches of the heads become one launch of ~240 whole tiles.  A queue (one per tag, so that the layer stack keeps
# its own rhythm) is flushed when it holds a whole number of rounds (>= 2), when a weight comes back a second time
# (two products adding into one gradient inside one launch would race), and at the end of the piece.
_WDEFER = {"on": False, "q": {}, "keys": {}, "fire": []}
_WDEFER_MIN_ROUNDS = int(os.environ.get("VG_WDEFER_MIN_ROUNDS", "2"))      # whole rounds of CUs a queue must hold before it leaves


_CUS = []


def _cu_count() -> int:
    if not _CUS:
        _CUS.append(torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
                    if torch.cuda.is_available() else 256)
    return _CUS[0]


def _wgrad_enqueue(items, fire: bool, tag: str) -> None:
    q = _WDEFER["q"].setdefault(tag, [])
    keys = _WDEFER["keys"].setdefault(tag, set())
    new_keys = [(_wgrad_grad_view(w, x, col0).data_ptr(), w.shape[0], x.shape[1]) for w, _, x, col0 in items]
    if any(k in keys for k in new_keys):
        flush_wgrads(tag)
        q = _WDEFER["q"].setdefault(tag, [])
        keys = _WDEFER["keys"].setdefault(tag, set())
    q.extend(items)
    keys.update(new_keys)
    if fire:                                      # one "gradient ready" report per weight and node, as without the queue
        seen = set()
        for w, _, _, _ in items:
            if id(w) not in seen:
                seen.add(id(w))
                _WDEFER["fire"].append(w)
    tiles = sum(_wgrad_tiles(w, x) for w, _, x, _ in q)
    cus = _cu_count()
    # whole rounds (>= 2) leave at once; a queue whose tile count never lands on a round (e.g. 108 tiles per layer at
    # d = 768) leaves at six rounds anyway, so the queued operands of every layer do not stay alive for the whole
    # backward piece
    if (tiles >= _WDEFER_MIN_ROUNDS * cus and tiles % cus == 0) or tiles >= max(6, _WDEFER_MIN_ROUNDS) * cus:
        flush_wgrads(tag)


def flush_wgrads(tag: Optional[str] = None) -> None:
    tags = [tag] if tag is not None else list(_WDEFER["q"].keys())
    for t in tags:
        q = _WDEFER["q"].pop(t, [])
        _WDEFER["keys"].pop(t, None)
        by_m = {}
        for it in q:                              # one launch shares its reduction length (lockstep plan)
            by_m.setdefault(it[2].shape[0], []).append(it)
        for same in by_m.values():
            for i in range(0, len(same), GROUP_MAX):
                _launch_wgrad_items(same[i:i + GROUP_MAX])
    if not _WDEFER["q"]:
        fire, _WDEFER["fire"] = _WDEFER["fire"], []
        for w in fire:
            _fire(w)


def sink_wgrad_group(items, fire: bool = True, tag: str = "misc") -> None:
    """``items``: (weight, dy[M, N], x[M, K]) or (weight, dy, x, col0) of one backward node.  The weight gradients
    ``weight.grad[N, col0:col0 + K] += dy^T x`` of all of them in ONE launch (``vg_gemm_grouped``: persistent grid of
    256x256 tiles with equal K ranges per CU) when every product qualifies and together they are at least
    ``VG_GROUP_MIN_TILES`` tiles; one split-K launch each otherwise.  ``col0`` addresses a column slice of a weight
    whose input is a concatenation (the conv block's [u | cond] Linear).  Inside a deferral bracket
    (``defer_vec_grads``) qualified products are queued instead and leave with the products of other nodes."""
    items = [it if len(it) == 4 else (*it, 0) for it in items if it is not None]
    if not items:
        return
    ok = _PH_GROUP and len(items) <= GROUP_MAX and all(_wgrad_item_ok(*it) for it in items)
    if ok and _WDEFER["on"] and (not on_side_stream() or _SIDE_DEFER[0]):
        _wgrad_enqueue(items, fire, tag + "@side" if on_side_stream() else tag)       # (see side_defer)
        return
    if ok and len({it[2].shape[0] for it in items}) == 1:
        _launch_wgrad_items(items)
    else:
        for w, dy, x, col0 in items:
            N, K, M = w.shape[0], x.shape[1], x.shape[0]
            sp = wgrad_splits(N, K, M, x.dtype)
            g = _wgrad_grad_view(w, x, col0)
            _WPASS["written"].add(_wgrad_region(g))
            gemm(dy, x, N, K, M, a_tr=True, b_tr=True, out=g, split_k=sp, accumulate=(sp == 1))
    if fire:
        seen = set()
        for w, _, _, _ in items:
            if id(w) not in seen:
                seen.add(id(w))
                _fire(w)


def colsum(x: Tensor, into: Optional[Tensor] = None) -> Tensor:
    """Column sums of [M, N]; ``into`` (fp32, N elements, contiguous) receives ``+=`` the result."""
    M, N = x.shape
    nb = lib().vg_colsum_blocks(M)
    ws = torch.empty((nb, N), dtype=torch.float32, device=x.device) if nb > 1 else None
    out = into if into is not None else torch.empty((N,), dtype=torch.float32, device=x.device)
    check(lib().vg_colsum(ptr(x), M, N, x.stride(0), ptr(ws), ptr(out), dtype_id(x.dtype),
                          int(into is not None), stream()), "vg_colsum")
    return out


# Deferred second stage of the small column sums (bias / norm-scale gradients).  A backward pass produces ~100 of them;
# each finishing launch is ~4.5 us of graph-node floor for a few KB of work.  Between defer_vec_grads(True) and
# flush_vec_grads() the (partial-sum array -> parameter gradient) tasks of sunk parameters are queued and leave in
# launches of up to COLSUM_MAX_TASKS; the trainer brackets every backward piece that nothing reads gradients inside of.
_DEFER = {"on": False, "tasks": [], "keep": [], "fire": []}


def defer_vec_grads(on: bool) -> None:
    """Bracket of a backward piece nothing reads gradients inside of: the finishing launches of the small column sums
    and (round 4) the qualified weight-gradient products wait for its end."""
    if not on:
        flush_vec_grads()
    _DEFER["on"] = bool(on) and _COLSUM_MULTI and _flag("VG_DEFER_COLSUM", "1")
    _WDEFER["on"] = bool(on) and _PH_GROUP and _flag("VG_DEFER_WGRAD", "1")


def reset_vec_grads() -> None:
    """Drop the queue without launching (a backward piece that raised, e.g. inside a failed graph capture: the queued
    addresses belong to tensors of that attempt)."""
    _DEFER["on"] = False
    _DEFER["tasks"], _DEFER["keep"], _DEFER["fire"] = [], [], []
    _WDEFER["on"] = False
    _WDEFER["q"], _WDEFER["keys"], _WDEFER["fire"] = {}, {}, []


def flush_vec_grads() -> None:
    """End of a deferral bracket: the queued weight gradients, then the queued column-sum tasks."""
    flush_wgrads()
    _flush_vec_tasks()


def _flush_vec_tasks() -> None:
    import hipvg
    # the queue is emptied BEFORE the launches: if one of them raises, nothing stale (task pointers into tensors of
    # this attempt) is left for the next flush to relaunch; `keep` holds the tensors alive until the launches are queued
    tasks, fire, keep = _DEFER["tasks"], _DEFER["fire"], _DEFER["keep"]
    _DEFER["tasks"], _DEFER["keep"], _DEFER["fire"] = [], [], []
    for i in range(0, len(tasks), hipvg.COLSUM_MAX_TASKS):
        chunk = tasks[i:i + hipvg.COLSUM_MAX_TASKS]
        arr = (hipvg.ColsumTask * len(chunk))(*chunk)
        check(lib().vg_colsum_multi(arr, len(chunk), stream()), "vg_colsum_multi")
    del keep
    for p in fire:
        _fire(p)


def _small_f32(src: Tensor, n: int) -> bool:
    return (src.dtype == torch.float32 and src.dim() == 2 and src.stride(1) == 1 and src.shape[0] <= 2048
            and src.shape[1] % 4 == 0 and src.stride(0) % 4 == 0 and src.data_ptr() % 16 == 0 and src.shape[1] == n)


def _defer_task(p: Tensor, src: Tensor) -> None:
    import hipvg
    dst = _grad_buffer(p).view(-1)
    _DEFER["tasks"].append(hipvg.ColsumTask(src.data_ptr(), src.shape[0], src.shape[1], src.stride(0), dst.data_ptr(), 1))
    _DEFER["keep"].append((src, dst))
    _DEFER["fire"].append(p)
    if len(_DEFER["tasks"]) >= 4 * hipvg.COLSUM_MAX_TASKS:
        _flush_vec_tasks()


def vec_grad(p, src2d: Tensor):
    """Gradient of a vector-like parameter = column sums of ``src2d``: accumulated straight into
    ``p.grad`` when the gradient sink applies (returns None), else returned as a tensor."""
    if p is None:
        return None
    if _sinkable(p):
        if _DEFER["on"] and src2d.dim() == 2 and not on_side_stream():
            if _small_f32(src2d, p.numel()):
                _defer_task(p, src2d)
                return None
            parts = colsum_partials([src2d]) if src2d.shape[1] == p.numel() else None
            if parts is not None and _small_f32(parts[0], p.numel()):
                _defer_task(p, parts[0])        # first stage now (the matrix is alive), second stage with the others
                return None
        sink_colsum(p, src2d)
        return None
    return colsum(src2d).view_as(p)


def vec_grads(pairs):
    """``[vec_grad(p, src) for p, src in pairs]`` with all the small fp32 partial-sum arrays reduced by ONE launch
    (vg_colsum_multi); anything that does not qualify takes the single path."""
    import hipvg
    out = [None] * len(pairs)
    tasks, fire, dense = [], [], []
    for i, (p, src) in enumerate(pairs):
        if p is None or src is None:
            continue
        ok = (_COLSUM_MULTI and src.dtype == torch.float32 and src.dim() == 2 and src.stride(1) == 1
              and src.shape[0] <= 2048 and src.shape[1] % 4 == 0 and src.stride(0) % 4 == 0
              and src.data_ptr() % 16 == 0 and src.shape[1] == p.numel() and len(tasks) < hipvg.COLSUM_MAX_TASKS)
        if not ok:
            out[i] = vec_grad(p, src)
            continue
        if _DEFER["on"] and _sinkable(p) and not on_side_stream():
            _defer_task(p, src)
            continue
        if _sinkable(p):
            dst, acc = _grad_buffer(p).view(-1), 1
            fire.append(p)
        else:
            dst, acc = torch.empty(p.numel(), dtype=torch.float32, device=src.device), 0
            out[i] = dst.view_as(p)
        tasks.append(hipvg.ColsumTask(src.data_ptr(), src.shape[0], src.shape[1], src.stride(0), dst.data_ptr(), acc))
        dense.append((src, dst))                     # keep both alive until the launch is enqueued
    if tasks:
        arr = (hipvg.ColsumTask * len(tasks))(*tasks)
        check(lib().vg_colsum_multi(arr, len(tasks), stream()), "vg_colsum_multi")
        for p in fire:
            _fire(p)
    return out


def colsum_partials(mats):
    """First stage of the column sums of several large [M, N_i] matrices of one dtype in ONE launch
    (``vg_colsum_partials_multi``): returns fp32 partial-sum arrays [nb, N_i] to be folded by ``vec_grads``; None
    for the whole list when some matrix does not qualify."""
    import hipvg
    if not mats or len(mats) > hipvg.COLSUM_MAX_TASKS:
        return None
    dt = mats[0].dtype
    vec = 8 if dt == torch.bfloat16 else 4
    M = mats[0].shape[0]
    nb = lib().vg_colsum_blocks(M)
    for x in mats:
        if (x.dtype != dt or dt not in (torch.bfloat16, torch.float32) or x.dim() != 2 or x.stride(1) != 1
                or x.shape[0] != M or x.shape[1] % vec or x.stride(0) % vec or x.data_ptr() % 16 or nb <= 1):
            return None
    parts = [torch.empty((nb, x.shape[1]), dtype=torch.float32, device=x.device) for x in mats]
    arr = (hipvg.ColsumTask * len(mats))(*[hipvg.ColsumTask(x.data_ptr(), M, x.shape[1], x.stride(0), q.data_ptr(), 0)
                                           for x, q in zip(mats, parts)])
    check(lib().vg_colsum_partials_multi(arr, len(mats), nb, dtype_id(dt), stream()), "vg_colsum_partials_multi")
    return parts


def masked_means(items, lengths: Optional[Tensor], T: int) -> Optional[Tensor]:
    """``items``: [(x [M, cols] fp32 with unit column stride, absolute: bool)] -> fp32 [len(items)] of
    ``TensorMask(x, mask).mean()`` (or of |x|) in ONE launch (``vg_masked_means``); None when an item does not qualify
    (the caller falls back to the stock expression).  Detached: these are monitors."""
    import hipvg
    if not items or len(items) > hipvg.MEAN_MAX_TASKS:
        return None
    M = items[0][0].shape[0]
    for x, _ in items:
        if (x.dtype != torch.float32 or x.dim() != 2 or x.shape[0] != M or x.stride(1) != 1 or x.shape[1] > 64
                or not x.is_cuda):
            return None
    keep = [x.detach() for x, _ in items]
    arr = (hipvg.MeanTask * len(items))(*[hipvg.MeanTask(k.data_ptr(), k.stride(0), k.shape[1], int(bool(a)))
                                          for k, (_, a) in zip(keep, items)])
    out = torch.empty(len(items), dtype=torch.float32, device=keep[0].device)
    part = torch.empty((lib().vg_masked_means_blocks(M), hipvg.MEAN_MAX_TASKS + 1), dtype=torch.float32, device=out.device)
    check(lib().vg_masked_means(arr, len(items), M, ptr(lengths), int(T), ptr(part), ptr(out), stream()), "vg_masked_means")
    return out


def segment_colsum(x: Tensor, nseg) -> Tensor:
    """``x.view(nseg, -1, C).float().sum(1)`` for [M, C] rows (M = nseg * T) in two small launches
    (``vg_colsum_segments``): fp32 [nseg, C].  ``nseg`` may be a ``PackPlan``: the segments are its (ragged) sequences."""
    M, Cc = x.shape
    vec = 8 if x.dtype == torch.bfloat16 else 4
    if isinstance(nseg, PackPlan):
        plan = nseg
        if x.dtype not in (torch.bfloat16, torch.float32) or Cc % vec or x.stride(1) != 1 or x.stride(0) % vec or x.data_ptr() % 16:
            x = x.float().contiguous()
            vec = 4
            if Cc % 4:
                raise RuntimeError("segment_colsum over packed rows needs a multiple of 4 columns")
        nb = max(1, min(64, plan.T // 64))
        part = torch.empty((nb, plan.nseq, Cc), dtype=torch.float32, device=x.device)
        out = torch.empty((plan.nseq, Cc), dtype=torch.float32, device=x.device)
        check(lib().vg_colsum_segments_cu(ptr(x), ptr(plan.cu), plan.nseq, Cc, x.stride(0), ptr(part), nb, ptr(out),
                                          dtype_id(x.dtype), stream()), "vg_colsum_segments_cu")
        return out
    rows = M // nseg
    if (x.dtype not in (torch.bfloat16, torch.float32) or M % nseg or Cc % vec or x.stride(1) != 1 or x.stride(0) % vec
            or x.data_ptr() % 16 or nseg > 65535):
        return x.view(nseg, rows, Cc).float().sum(1)
    nb = max(1, min(64, rows // 64))
    part = torch.empty((nb, nseg, Cc), dtype=torch.float32, device=x.device)
    out = torch.empty((nseg, Cc), dtype=torch.float32, device=x.device)
    check(lib().vg_colsum_segments(ptr(x), nseg, rows, Cc, x.stride(0), ptr(part), nb, ptr(out), dtype_id(x.dtype), stream()),
          "vg_colsum_segments")
    return out


def sink_colsum(p: Tensor, x: Tensor) -> None:
    """p.grad (any shape with N elements) += column sums of x[M, N]."""
    colsum(x, into=_grad_buffer(p).view(-1))
    _fire(p)


def act_bwd(dy: Tensor, aux: Tensor, act: int) -> Tensor:
    out = torch.empty_like(dy)
    check(lib().vg_act_bwd(ptr(dy), ptr(aux), ptr(out), dy.numel(), act, dtype_id(dy.dtype), stream()),
          "vg_act_bwd")
    return out


def mask_rows(x: Tensor, lengths: Tensor, T: int) -> Tensor:
    out = torch.empty_like(x)
    check(lib().vg_mask_rows(ptr(x), ptr(out), x.shape[0], x.shape[1], ptr(lengths), int(T), dtype_id(x.dtype),
                             stream()), "vg_mask_rows")
    return out


def sum_f32(x: Tensor) -> Tensor:
    out = torch.empty((), dtype=torch.float32, device=x.device)
    check(lib().vg_sum_f32(ptr(x), x.numel(), ptr(out), stream()), "vg_sum_f32")
    return out


def _as(x: Tensor, dtype: torch.dtype) -> Tensor:
    x = x if x.dtype == dtype else x.to(dtype)
    return x if x.is_contiguous() else x.contiguous()


# ---------------------------------------------------------------- Linear (+bias, act, residual, mask)
class LinearFn(torch.autograd.Function):
    """y = mask( act(x W^T + b) + residual )   (modules/linear/layers.py:192-193,
    modules/attention/attention.py:52,79-80, modules/transformer/layers.py:151-154)."""

    @staticmethod
    def forward(ctx, x, weight, bias, residual, lengths, T, act, out_f32):
        M, K = x.shape
        N = weight.shape[0]
        w = shadow(weight, x.dtype)
        b = None if bias is None else bias.detach().float()
        aux = None
        y_dtype_f32 = bool(out_f32)
        if act == ACT_GELU:
            aux = torch.empty((M, N), dtype=x.dtype, device=x.device)
        y = gemm(x, w, M, N, K, bias=b, residual=residual, lengths=lengths, T=T, act=act,
                 aux_out=aux, out_f32=y_dtype_f32)
        if act == ACT_RELU:
            aux = y
        ctx.save_for_backward(x, w, aux, lengths)
        ctx.meta = (T, act, bias is not None, residual is not None, weight.shape)
        ctx.params = (weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, aux, lengths = ctx.saved_tensors
        T, act, has_bias, has_res, wshape = ctx.meta
        M, K = x.shape
        N = wshape[0]
        dy = _as(dy, x.dtype)
        if lengths is not None:      # forward masked the output rows: so is the gradient
            dy = mask_rows(dy, lengths, T)
        du = dy
        if act != ACT_NONE:
            du = act_bwd(dy, _as(aux, x.dtype), act)
        dx = dW = db = dres = None
        if ctx.needs_input_grad[0]:
            dx = gemm(du, w, M, K, N, b_tr=True, lengths=lengths, T=T)
        weight, bias = ctx.params
        want_b = has_bias and ctx.needs_input_grad[2]
        if ctx.needs_input_grad[1]:
            dW, db = wgrad_pair(weight, bias if want_b else None, du, x)
        elif want_b:
            db = vec_grad(bias, du)
        if has_res and ctx.needs_input_grad[3]:
            dres = dy
        return dx, dW, db, dres, None, None, None, None


def linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, *, act=None,
           residual: Optional[Tensor] = None, lengths: Optional[Tensor] = None, T: int = 0,
           out_f32: bool = False) -> Tensor:
    return LinearFn.apply(x, weight, bias, residual, lengths, T, ACT_IDS[act] if not isinstance(act, int) else act,
                          out_f32)


class SmallLinearFn(torch.autograd.Function):
    """y = a W^T + b in fp32 for a FEW rows (the diffusion-step embedding, one row per sequence: the time-embedding MLP
    and the blocks' batched projection of it, reference modules/diffusion/unet.py:20-29 and modules/conv/layers.py:93-96).
    The forward is the stock addmm (6 - 14 µs).  The stock BACKWARD is not usable: `mm([16, 3072], [3072, 256])`, the
    input gradient of the batched projection, lands on a vendor kernel (MT32x16x256, 16 workgroups walking K = 3072
    with 16-wide fp32 MFMAs) that takes 0.46 ms inside the replayed graph and 1.07 ms in an eager step for 25 MFLOP
    (tools/lab/mm_shapes.py) -- the longest single launch of the step after the grouped weight gradient.  Here the
    input gradient is the split-K fp32 HIP product and the weight gradient the thin kernel (K = rows <= 16) or the
    same split-K product."""

    @staticmethod
    def forward(ctx, a, weight, bias):
        a = a.float().contiguous()
        w = weight.detach().float()
        y = torch.addmm(bias.detach().float(), a, w.t()) if bias is not None else a @ w.t()
        ctx.save_for_backward(a, w)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        a, w = ctx.saved_tensors
        dy = dy.float().contiguous()
        R, N, K = a.shape[0], w.shape[0], w.shape[1]
        da = dW = db = None
        if ctx.needs_input_grad[0]:
            da = gemm(dy, w, R, K, N, b_tr=True, out_f32=True, split_k=max(1, min(16, N // 256)))
        if ctx.needs_input_grad[1]:
            dW = gemm(dy, a, N, K, R, a_tr=True, b_tr=True, out_f32=True)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum(0)
        return da, dW, db


def small_linear(a: Tensor, weight: Tensor, bias: Optional[Tensor]) -> Tensor:
    return SmallLinearFn.apply(a, weight, bias)


class SliceLinearFn(torch.autograd.Function):
    """y = x W[:, col0:col0 + K]^T (+ b) (+ residual) for a column slice of a PARAMETER (a k = 1 Conv1d whose input is
    a concatenation computed piece by piece: the UNet's skip convolutions, modules/conv/layers.py of the reference).
    The weight-gradient slice goes straight into ``W.grad[:, col0:col0 + K]`` (deferred with the other products of the
    backward piece); passing the slice itself to ``linear`` would hand autograd a dense dW per piece plus the
    zero-fill / copy / add launches that assemble them."""

    @staticmethod
    def forward(ctx, x, wparam, col0, bias, residual):
        M, K = x.shape
        N = wparam.shape[0]
        ws = shadow(wparam, x.dtype).view(N, -1)[:, col0:col0 + K]
        b = None if bias is None else bias.detach().float()
        y = gemm(x, ws, M, N, K, bias=b, residual=residual)
        ctx.save_for_backward(x, ws)
        ctx.params = (wparam, bias)
        ctx.meta = (col0, residual is not None)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, ws = ctx.saved_tensors
        wparam, bias = ctx.params
        col0, has_res = ctx.meta
        M, K = x.shape
        N = wparam.shape[0]
        dy = _as(dy, x.dtype)
        dx = gemm(dy, ws, M, K, N, b_tr=True) if ctx.needs_input_grad[0] else None
        dW = None
        if ctx.needs_input_grad[1]:
            if _sinkable(wparam) and wparam.is_contiguous():
                sink_wgrad_group([(wparam, dy, x, col0)])
            else:
                dW = torch.zeros((N, wparam.numel() // N), dtype=torch.float32, device=x.device)
                gemm(dy, x, N, K, M, a_tr=True, b_tr=True, out=dW[:, col0:col0 + K], split_k=1, accumulate=True)
                dW = dW.view_as(wparam)
        db = vec_grad(bias, dy) if (bias is not None and ctx.needs_input_grad[3]) else None
        dres = dy if (has_res and ctx.needs_input_grad[4]) else None
        return dx, dW, None, db, dres


def slice_linear(x: Tensor, wparam: Tensor, col0: int, bias: Optional[Tensor] = None, residual: Optional[Tensor] = None):
    return SliceLinearFn.apply(x, wparam, int(col0), bias, residual)


class StackedLinearFn(torch.autograd.Function):
    """y = x [W_0; W_1; ...]^T + [b_0; b_1; ...]: several Linears of one input as ONE product (the coupling stack's four
    FiLM projections, modules/flow/layers.py:15-40 of the reference, computed from the same conditioning rows).
    ``torch.cat`` of the parameters in front of ``linear`` would hand autograd a dense gradient of the concatenation
    (one split-K launch) plus the slice / accumulate launches that take it apart; here each W_i's gradient is a
    column block of dY and goes into ``W_i.grad`` with the other weight gradients of the backward piece."""

    @staticmethod
    def forward(ctx, x, out_f32, n, *params):
        weights, biases = params[:n], params[n:]
        M, K = x.shape
        ws = torch.cat([shadow(w, x.dtype).view(w.shape[0], -1) for w in weights], 0)
        N = ws.shape[0]
        b = torch.cat([bb.detach().float() for bb in biases], 0) if biases else None
        y = gemm(x, ws, M, N, K, bias=b, out_f32=bool(out_f32))
        ctx.save_for_backward(x, ws)
        ctx.params = (weights, biases)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, ws = ctx.saved_tensors
        weights, biases = ctx.params
        n = len(weights)
        M, K = x.shape
        N = ws.shape[0]
        dy = _as(dy, x.dtype)
        dx = gemm(dy, ws, M, K, N, b_tr=True) if ctx.needs_input_grad[0] else None
        offs = [0]
        for w in weights:
            offs.append(offs[-1] + w.shape[0])
        gw = [None] * n
        if all(_sinkable(w) and w.is_contiguous() for w in weights):
            sink_wgrad_group([(w, dy[:, offs[i]:offs[i + 1]], x, 0) for i, w in enumerate(weights)])
        else:
            dW = gemm(dy, x, N, K, M, a_tr=True, b_tr=True, out_f32=True, split_k=wgrad_splits(N, K, M, x.dtype))
            gw = [dW[offs[i]:offs[i + 1]].view_as(w) for i, w in enumerate(weights)]
        gb = []
        if biases:
            db = colsum(dy)
            gb = [db[offs[i]:offs[i + 1]] for i in range(n)]
        return (dx, None, None, *gw, *gb)


def stacked_linear(x: Tensor, weights, biases=None, out_f32: bool = False) -> Tensor:
    biases = list(biases) if biases is not None and all(b is not None for b in biases) else []
    return StackedLinearFn.apply(x, bool(out_f32), len(weights), *weights, *biases)


# ---------------------------------------------------------------- fused FFN
class FFNFn(torch.autograd.Function):
    """y = mask( residual + W2 gelu(W1 x + b1) + b2 )  (modules/transformer/layers.py:78-86)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, residual, lengths, T):
        M, K = x.shape
        F_ = w1.shape[0]
        s1, s2 = shadow(w1, x.dtype), shadow(w2, x.dtype)
        u, act_f, dact_b = _gelu_deriv_buffer(M, F_, x.dtype, x.device)
        # u receives GELU'(pre-activation): the backward epilogue is then a single multiply
        h = gemm(x, s1, M, F_, K, bias=None if b1 is None else b1.detach(), act=act_f, aux_out=u)
        y = gemm(h, s2, M, K, F_, bias=None if b2 is None else b2.detach(), residual=residual,
                 lengths=lengths, T=T)
        ctx.save_for_backward(x, s1, s2, u, h, lengths)
        ctx.meta = (T, b1 is not None, b2 is not None, residual is not None, dact_b)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, s1, s2, u, h, lengths = ctx.saved_tensors
        T, hb1, hb2, has_res, dact_b = ctx.meta
        M, K = x.shape
        F_ = s1.shape[0]
        dy = _as(dy, x.dtype)
        du = gemm(dy, s2, M, F_, K, b_tr=True, dact=dact_b, aux_in=u)
        dW2 = gemm(dy, h, K, F_, M, a_tr=True, b_tr=True, out_f32=True,
                   split_k=wgrad_splits(K, F_, M, x.dtype)) if ctx.needs_input_grad[3] else None
        db2 = colsum(dy) if (hb2 and ctx.needs_input_grad[4]) else None
        dx = gemm(du, s1, M, K, F_, b_tr=True, lengths=lengths, T=T) if ctx.needs_input_grad[0] else None
        dW1 = gemm(du, x, F_, K, M, a_tr=True, b_tr=True, out_f32=True,
                   split_k=wgrad_splits(F_, K, M, x.dtype)) if ctx.needs_input_grad[1] else None
        db1 = colsum(du) if (hb1 and ctx.needs_input_grad[2]) else None
        dres = dy if (has_res and ctx.needs_input_grad[5]) else None
        return dx, dW1, db1, dW2, db2, dres, None, None


def ffn(x, w1, b1, w2, b2, *, residual=None, lengths=None, T=0):
    return FFNFn.apply(x, w1, b1, w2, b2, residual, lengths, T)


# ---------------------------------------------------------------- RMSNorm
class RMSNormFn(torch.autograd.Function):
    """modules/norm.py:28-32 (+ re-mask, modules/transformer/layers.py:52-54)."""

    @staticmethod
    def forward(ctx, x, scale, eps, lengths, T):
        M, Cc = x.shape
        y = torch.empty_like(x)
        rstd = torch.empty((M,), dtype=torch.float32, device=x.device)
        sc = scale.detach().float().contiguous()
        check(lib().vg_rmsnorm_fwd(ptr(x), ptr(sc), ptr(y), ptr(rstd), M, Cc, float(eps), ptr(lengths),
                                   int(T), dtype_id(x.dtype), stream()), "vg_rmsnorm_fwd")
        ctx.save_for_backward(x, sc, rstd, lengths)
        ctx.T = T
        return y

    @staticmethod
    def backward(ctx, dy):
        x, sc, rstd, lengths = ctx.saved_tensors
        M, Cc = x.shape
        dy = _as(dy, x.dtype)
        # (the stack's final norm: its dx is the incoming gradient of the top Transformer layer)
        dx, part = rmsnorm_bwd_raw(dy, x, sc, rstd, None, lengths, int(ctx.T), dx_colsum=True)
        dscale = colsum(part) if ctx.needs_input_grad[1] else None
        return dx, dscale, None, None, None


def rmsnorm(x, scale, eps, *, lengths=None, T=0):
    return RMSNormFn.apply(x, scale, eps, lengths, T)


# ---------------------------------------------------------------- attention
def alibi_slopes(nheads: int):
    """modules/position/alibi.py:19-30 (positive slopes; the bias is -slope*|i-j|)."""
    def pow2(n):
        start = 2 ** (-2 ** -(math.log2(n) - 3))
        return [start * start ** i for i in range(n)]
    if math.log2(nheads).is_integer():
        return pow2(nheads)
    c = 2 ** math.floor(math.log2(nheads))
    return pow2(c) + alibi_slopes(2 * c)[0::2][:nheads - c]


def attn_workspace(B: int, T: int, H: int, rows: int, device) -> Tensor:
    """fp32 [H * rows + vg_attn_stats_floats]: the log-sum-exp rows the backward needs, followed by the per-(batch, head)
    statistics of its ALiBi window (include/vaegslm_hip.h, vg_attn_fwd_stats) -- one allocation, one saved tensor."""
    return torch.empty(H * rows + lib().vg_attn_stats_floats(B, T, H), dtype=torch.float32, device=device)


def attn_fwd_raw(qkv, out, ws, slopes, B, T, H, lengths, cu=None, rows=0) -> None:
    M = rows if cu is not None else B * T
    check(lib().vg_attn_fwd_stats(ptr(qkv), ptr(out), ptr(ws), ptr(slopes), B, T, H, ptr(lengths), ptr(cu), int(rows),
                                  ws.data_ptr() + 4 * H * M, dtype_id(qkv.dtype), stream()), "vg_attn_fwd")


def attn_bwd_raw(qkv, out, dout, ws, slopes, dqkv, delta, B, T, H, lengths, cu=None, rows=0) -> None:
    M = rows if cu is not None else B * T
    check(lib().vg_attn_bwd_stats(ptr(qkv), ptr(out), ptr(dout), ptr(ws), ptr(slopes), ptr(dqkv), ptr(delta), B, T, H,
                                  ptr(lengths), ptr(cu), int(rows), ws.data_ptr() + 4 * H * M, dtype_id(qkv.dtype),
                                  stream()), "vg_attn_bwd")


class AttentionFn(torch.autograd.Function):
    """Causal ALiBi attention over the packed in_proj output
    (modules/attention/attention.py:52-78)."""

    @staticmethod
    def forward(ctx, qkv, slopes, B, T, H, lengths):
        D = H * 64
        assert qkv.shape == (B * T, 3 * D) and qkv.is_contiguous()
        out = torch.empty((B * T, D), dtype=qkv.dtype, device=qkv.device)
        lse = attn_workspace(B, T, H, B * T, qkv.device)
        attn_fwd_raw(qkv, out, lse, slopes, B, T, H, lengths)
        ctx.save_for_backward(qkv, out, lse, slopes, lengths)
        ctx.dims = (B, T, H)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse, slopes, lengths = ctx.saved_tensors
        B, T, H = ctx.dims
        dout = _as(dout, qkv.dtype)
        dqkv = torch.empty_like(qkv)
        delta = torch.empty((H, B * T), dtype=torch.float32, device=qkv.device)
        attn_bwd_raw(qkv, out, dout, lse, slopes, dqkv, delta, B, T, H, lengths)
        return dqkv, None, None, None, None, None


def attention(qkv, slopes, B, T, H, lengths=None):
    return AttentionFn.apply(qkv, slopes, B, T, H, lengths)


# ---------------------------------------------------------------- packed rows (valid frames of a right-padded batch)
class PackPlan:
    """Row maps of one right-padded batch (``utils/tensormask.py:45-54``: prefix masks) for running on its VALID frames
    only: ``rows`` packed rows (the sum of the lengths rounded up to ``granule``, so that a hipGraph is captured per
    bucket and not per batch), every tensor of static shape and computed on the device from ``lengths`` -- the plan can
    be (re)filled inside a captured graph.

    ``idx[i]``  frame (b * T + t) of packed row i, -1 for the rows the rounding added (and for halo rows);
    ``inv[m]``  packed row of frame m, -1 for padded frames;
    ``cu`` / ``lengths``  row ranges of the B real sequences followed by zero-length pseudo sequences (each at most T
    rows) that cover the added rows: the attention kernels zero-fill those, every other kernel of the stack is
    row-local and sees them as all-zero frames (their gradients are exactly zero).

    ``halo`` > 0 (round 5: the whole step on packed rows, not only the Transformer stack): sequence b owns
    min(len_b + halo, T) rows, its valid frames followed by the first padded ones.  The reference convolves the padding
    (modules/conv/layers.py:70-135 runs on the padded (B, C, T) tensor): under the diffusion UNet's three look-ahead
    blocks (configs/train/speech/vae-gslm.yaml:156-157, 7 taps each) a valid frame depends on up to 18 padded frames
    after its sequence's end, and those padded frames hold computed values (biases, time embedding, left context), not
    zeros.  With an 18-row halo per sequence the valid frames -- values and gradients -- are those of the padded run:
    after the k-th look-ahead block the rows from len + 18 - 6 k on differ from the padded run's, the gradient reaches
    exactly the rows below that, and the end of a sequence's range is padded with zeros as the end of the batch is
    (min(.., T): a sequence that fills the batch has nothing after it there either).
    ``valid[i]`` 1 for a real frame (the row predicate of every row-local kernel: ``lengths = valid, T = 1``), ``seq[i]``
    the sequence of row i, ``shift_src`` / ``shift_dst`` the row maps of the one-frame shift inside each sequence."""

    def __init__(self, B: int, T: int, rows: int, device, granule: Optional[int] = None, halo: int = 0):
        self.B, self.T, self.M, self.rows, self.halo = B, T, B * T, rows, int(halo)
        # the rows the rounding adds (at most one granule: rows = max(granule, total rounded up); all of them when the
        # granule is not given) are covered by zero-length pseudo sequences of at most T rows each
        self.npseudo = -(-min(rows, granule or rows) // T)
        self.nseq = B + self.npseudo
        self.idx = torch.empty(rows, dtype=torch.int32, device=device)
        self.inv = torch.empty(B * T, dtype=torch.int32, device=device)
        self.cu = torch.empty(self.nseq + 1, dtype=torch.int32, device=device)
        self.lengths = torch.empty(self.nseq, dtype=torch.int32, device=device)
        self.valid = torch.empty(rows, dtype=torch.int32, device=device)
        self.seq = torch.empty(rows, dtype=torch.int32, device=device)
        self.shift_src = torch.empty(rows, dtype=torch.int32, device=device)
        self.shift_dst = torch.empty(rows + B, dtype=torch.int32, device=device)
        self._t = torch.arange(T, device=device, dtype=torch.int32)[None]
        self._m = torch.arange(B * T, device=device, dtype=torch.int32)
        self._j = torch.arange(self.npseudo + 1, device=device, dtype=torch.int32)
        self._r = torch.arange(rows, device=device, dtype=torch.int32)
        self._b = torch.arange(B, device=device, dtype=torch.int32)

    def fill(self, lengths32: Tensor) -> "PackPlan":
        """Device-only (no host synchronisation): safe to record into a hipGraph whose ``lengths32`` is a static input."""
        B, T, rows = self.B, self.T, self.rows
        lens = lengths32.to(torch.int32).clamp(0, T)
        ext = torch.clamp(lens + self.halo, max=T) if self.halo else lens
        ends = torch.cumsum(ext, 0, dtype=torch.int32)
        starts = ends - ext
        valid = self._t < lens[:, None]
        inv = torch.where(valid, starts[:, None] + self._t, torch.full_like(self._t, -1)).reshape(-1)
        self.inv.copy_(inv)
        buf = torch.full((rows + 1,), -1, dtype=torch.int32, device=lens.device)
        buf.scatter_(0, torch.where(inv >= 0, inv, torch.full_like(inv, rows)).long(), self._m)
        self.idx.copy_(buf[:rows])
        total = ends[-1:]
        tail = torch.minimum(total + self._j * T, torch.full_like(self._j, rows))       # pseudo-sequence boundaries
        self.cu.copy_(torch.cat([torch.zeros(1, dtype=torch.int32, device=lens.device), ends[:-1], tail]))
        self.lengths.copy_(torch.cat([lens, torch.zeros(self.npseudo, dtype=torch.int32, device=lens.device)]))
        self.valid.copy_((self.idx >= 0).to(torch.int32))
        seq = torch.searchsorted(self.cu[1:].contiguous(), self._r, right=True).clamp(max=self.nseq - 1).to(torch.int32)
        self.seq.copy_(seq)
        # one-frame shift inside every sequence: row i takes row i - 1, a sequence's first row takes start row b of the
        # source's B extra rows; shift_dst is the inverse map (source row -> the row it moved to)
        first = self._r == self.cu[seq.long()]
        ok = self.valid > 0
        src = torch.where(ok, torch.where(first, rows + seq, self._r - 1), torch.full_like(self._r, -1))
        self.shift_src.copy_(src)
        nxt = torch.cat([src[1:], torch.full((1,), -1, dtype=torch.int32, device=lens.device)])
        fwd = torch.where(nxt == self._r, self._r + 1, torch.full_like(self._r, -1))
        self.shift_dst.copy_(torch.cat([fwd, torch.where(lens > 0, starts, torch.full_like(starts, -1))]))
        return self


def pack_rows_bucket(total: int, granule: int = 256) -> int:
    return max(granule, -(-total // granule) * granule)


class GatherRowsFn(torch.autograd.Function):
    """dst[i] = src[map[i]] (0 where map[i] < 0); the backward is the gather with the inverse map."""

    @staticmethod
    def forward(ctx, src, fwd_map, bwd_map):
        assert src.dim() == 2 and src.is_contiguous() and (src.shape[1] * src.element_size()) % 16 == 0
        n = fwd_map.numel()
        dst = torch.empty((n, src.shape[1]), dtype=src.dtype, device=src.device)
        check(lib().vg_gather_rows(ptr(src), ptr(fwd_map), ptr(dst), n, src.shape[1] * src.element_size(), stream()),
              "vg_gather_rows")
        ctx.save_for_backward(bwd_map)
        return dst

    @staticmethod
    def backward(ctx, ddst):
        (bwd_map,) = ctx.saved_tensors
        ddst = ddst.contiguous()
        n = bwd_map.numel()
        dsrc = torch.empty((n, ddst.shape[1]), dtype=ddst.dtype, device=ddst.device)
        check(lib().vg_gather_rows(ptr(ddst), ptr(bwd_map), ptr(dsrc), n, ddst.shape[1] * ddst.element_size(), stream()),
              "vg_gather_rows")
        return dsrc, None, None


def pack_rows(x2: Tensor, plan: PackPlan) -> Tensor:
    """[B * T, C] padded rows -> [plan.rows, C] packed rows."""
    return GatherRowsFn.apply(x2, plan.idx, plan.inv)


def unpack_rows(xp: Tensor, plan: PackPlan) -> Tensor:
    """[plan.rows, C] packed rows -> [B * T, C] padded rows (zeros on padded frames)."""
    return GatherRowsFn.apply(xp, plan.inv, plan.idx)


def shift_rows(xp: Tensor, start: Tensor, plan: PackPlan) -> Tensor:
    """The one-frame right shift of ``TensorMask.push(start).pop(1).apply_mask()`` (reference models/speech/lvtr.py:
    shifted prior input) on packed rows: row i of a sequence takes row i - 1, its first row takes ``start[b]``; rows
    that hold no frame are zero."""
    src = torch.cat([xp, start.to(xp.dtype).reshape(plan.B, -1)], 0).contiguous()
    return GatherRowsFn.apply(src, plan.shift_src, plan.shift_dst)[:plan.rows]


class SeqRowsFn(torch.autograd.Function):
    """rows[i] = per_seq[seq[i]]: a per-sequence vector on every packed row of its sequence (``u_c[:, None].expand(-1, T,
    -1)`` of the padded layout, rows of pseudo sequences take the last real one); the backward is the per-sequence
    column sum, in a fixed order (an index_add would not be)."""

    @staticmethod
    def forward(ctx, per_seq, plan):
        src = per_seq.contiguous()
        assert src.dim() == 2 and (src.shape[1] * src.element_size()) % 16 == 0
        fmap = plan.seq.clamp(max=plan.B - 1)
        dst = torch.empty((plan.rows, src.shape[1]), dtype=src.dtype, device=src.device)
        check(lib().vg_gather_rows(ptr(src), ptr(fmap), ptr(dst), plan.rows, src.shape[1] * src.element_size(), stream()),
              "vg_gather_rows")
        ctx.plan = plan
        return dst

    @staticmethod
    def backward(ctx, d):
        plan = ctx.plan
        return segment_colsum(d.contiguous(), plan)[:plan.B].to(d.dtype), None


def seq_rows(per_seq: Tensor, plan: PackPlan) -> Tensor:
    return SeqRowsFn.apply(per_seq, plan)


def attention_decode(q, kcache, vcache, slopes, pos, H):
    B, D = q.shape
    out = torch.empty_like(q)
    check(lib().vg_attn_decode(ptr(q), ptr(kcache), ptr(vcache), ptr(out), ptr(slopes), ptr(pos), B,
                               kcache.shape[1], H, dtype_id(q.dtype), stream()), "vg_attn_decode")
    return out


# ---------------------------------------------------------------- token cross-entropy
class CrossEntropyFn(torch.autograd.Function):
    """training_lib/losses.py:30-41: sum over valid frames of -log softmax(logits)[target]."""

    @staticmethod
    def forward(ctx, logits, targets, lengths, T):
        M, V = logits.shape
        assert logits.is_contiguous()
        rows = torch.empty((M,), dtype=torch.float32, device=logits.device)
        lse = torch.empty((M,), dtype=torch.float32, device=logits.device)
        amax = torch.empty((M,), dtype=torch.int32, device=logits.device)
        tg = targets.reshape(-1).contiguous()
        check(lib().vg_ce_fwd(ptr(logits), ptr(tg), ptr(rows), ptr(lse), ptr(amax), M, V, logits.stride(0),
                              ptr(lengths), int(T), dtype_id(logits.dtype), stream()), "vg_ce_fwd")
        ctx.save_for_backward(logits, tg, lse, lengths)
        ctx.T = T
        ctx.mark_non_differentiable(amax)
        return sum_f32(rows), amax

    @staticmethod
    def backward(ctx, g, _):
        logits, tg, lse, lengths = ctx.saved_tensors
        M, V = logits.shape
        gs = g.detach().float().reshape(1).contiguous()
        dl = torch.empty_like(logits)
        check(lib().vg_ce_bwd(ptr(logits), ptr(tg), ptr(lse), ptr(gs), ptr(dl), M, V, logits.stride(0),
                              ptr(lengths), int(ctx.T), dtype_id(logits.dtype), stream()), "vg_ce_bwd")
        return dl, None, None, None


def cross_entropy_sum(logits, targets, lengths=None, T=0):
    return CrossEntropyFn.apply(logits, targets, lengths, T)


# ---------------------------------------------------------------- VAE terms
class ReparamFn(torch.autograd.Function):
    """z = mask(mu + eps * exp(logstd) * temperature), log_q = mask(-logstd - 0.5 - 0.5 ln 2pi)
    (modules/linear/layers.py:110-128; models/speech/lvtr.py:157-160)."""

    @staticmethod
    def forward(ctx, mu, logstd, eps, temperature, lengths, T):
        mu, logstd, eps = (_as(t, torch.float32) for t in (mu, logstd, eps))
        M, D = mu.shape
        z, lq = torch.empty_like(mu), torch.empty_like(mu)
        check(lib().vg_reparam_fwd(ptr(mu), ptr(logstd), ptr(eps), ptr(z), ptr(lq), M, D, float(temperature),
                                   ptr(lengths), int(T), stream()), "vg_reparam_fwd")
        ctx.save_for_backward(logstd, eps, lengths)
        ctx.meta = (float(temperature), T)
        return z, lq

    @staticmethod
    def backward(ctx, dz, dlq):
        logstd, eps, lengths = ctx.saved_tensors
        temperature, T = ctx.meta
        M, D = logstd.shape
        dz = None if dz is None else _as(dz, torch.float32)
        dlq = None if dlq is None else _as(dlq, torch.float32)
        dmu, dls = torch.empty_like(logstd), torch.empty_like(logstd)
        check(lib().vg_reparam_bwd(ptr(dz), ptr(dlq), ptr(logstd), ptr(eps), ptr(dmu), ptr(dls), M, D,
                                   temperature, ptr(lengths), int(T), stream()), "vg_reparam_bwd")
        return dmu, dls, None, None, None, None


def reparameterize(mu, logstd, eps, temperature=1.0, lengths=None, T=0):
    return ReparamFn.apply(mu, logstd, eps, temperature, lengths, T)


class PriorKLFn(torch.autograd.Function):
    """log_p under the conditional flow prior and the summed KL
    (models/speech/lvtr.py:182-191; training_lib/losses.py:16-18,27)."""

    @staticmethod
    def forward(ctx, mu_ls, u, logdet_sum, log_q, lengths, T):
        mu_ls, u, logdet_sum, log_q = (_as(t, torch.float32) for t in (mu_ls, u, logdet_sum, log_q))
        M, D = u.shape
        log_p = torch.empty_like(u)
        rows = torch.empty((M,), dtype=torch.float32, device=u.device)
        check(lib().vg_prior_logp_fwd(ptr(mu_ls), mu_ls.stride(0), ptr(u), ptr(logdet_sum), ptr(log_q),
                                      ptr(log_p), ptr(rows), M, D, ptr(lengths), int(T), stream()),
              "vg_prior_logp_fwd")
        ctx.save_for_backward(mu_ls, u, lengths)
        ctx.T = T
        return log_p, sum_f32(rows)

    @staticmethod
    def backward(ctx, dlog_p, dkl):
        mu_ls, u, lengths = ctx.saved_tensors
        M, D = u.shape
        dlp = None if dlog_p is None else _as(dlog_p, torch.float32)
        dkr = None if dkl is None else dkl.detach().float().reshape(1).expand(M).contiguous()
        dmu_ls = torch.empty((M, 2 * D), dtype=torch.float32, device=u.device)
        du = torch.empty_like(u)
        dld = torch.empty((M,), dtype=torch.float32, device=u.device)
        dlq = torch.empty_like(u)
        check(lib().vg_prior_logp_bwd(ptr(dlp), ptr(dkr), ptr(mu_ls), mu_ls.stride(0), ptr(u), ptr(dmu_ls),
                                      ptr(du), ptr(dld), ptr(dlq), M, D, ptr(lengths), int(ctx.T), stream()),
              "vg_prior_logp_bwd")
        return dmu_ls, du, dld, dlq, None, None


def prior_logp_kl(mu_ls, u, logdet_sum, log_q, lengths=None, T=0):
    return PriorKLFn.apply(mu_ls, u, logdet_sum, log_q, lengths, T)


def qsample(x0: Tensor, noise: Tensor, coef_x0: Tensor, coef_noise: Tensor, t: Tensor, lengths, T: int) -> Tensor:
    """x_t = mask(coef_x0[t_b] x0 + coef_noise[t_b] noise) on [M = B*T, C] fp32 rows (no gradient: x0 is data)."""
    M, C = x0.shape
    out = torch.empty(M, C, dtype=torch.float32, device=x0.device)
    check(lib().vg_qsample(ptr(_as(x0, torch.float32)), ptr(_as(noise, torch.float32)), ptr(coef_x0), ptr(coef_noise),
                           ptr(t.contiguous()), ptr(lengths), int(T), ptr(out), M, C, stream()), "vg_qsample")
    return out


class MaskedL1SumFn(torch.autograd.Function):
    """sum over valid frames of mean_c |pred - target| (training_lib/losses.py:9-27 with l1, default reductions)."""

    @staticmethod
    def forward(ctx, pred, target, lengths, T):
        M, C = pred.shape
        pred = pred.contiguous()
        target = _as(target, torch.float32)
        rows = torch.empty(M, dtype=torch.float32, device=pred.device)
        check(lib().vg_l1_rows_fwd(ptr(pred), ptr(target), ptr(lengths), int(T), ptr(rows), M, C, dtype_id(pred.dtype),
                                   stream()), "vg_l1_rows_fwd")
        ctx.save_for_backward(pred, target, lengths)
        ctx.T = int(T)
        return sum_f32(rows)

    @staticmethod
    def backward(ctx, g):
        pred, target, lengths = ctx.saved_tensors
        M, C = pred.shape
        dpred = torch.empty_like(pred)
        g = _as(g, torch.float32).reshape(1)
        check(lib().vg_l1_rows_bwd(ptr(pred), ptr(target), ptr(g), ptr(lengths), ctx.T, ptr(dpred), M, C,
                                   dtype_id(pred.dtype), stream()), "vg_l1_rows_bwd")
        return dpred, None, None, None


def masked_l1_sum(pred: Tensor, target: Tensor, lengths=None, T: int = 0) -> Tensor:
    return MaskedL1SumFn.apply(pred, target, lengths, T)


class EmbedFuseFn(torch.autograd.Function):
    """out[m] = mask(E[ids[m]]) + relu(Wf z[m] + bf) in fp32 (Embedding.forward
    modules/linear/layers.py:150-152 + token_fuser :184-193 + LVTR.fuse_inputs models/speech/lvtr.py:390-392)."""

    @staticmethod
    def forward(ctx, ids, z, emb, wf, bf, lengths, T):
        M, D = z.shape
        E = emb.shape[1]
        ids = ids.contiguous()
        z = _as(z, torch.float32)
        embd, wfd = emb.detach().float().contiguous(), wf.detach().float().contiguous()
        bfd = None if bf is None else bf.detach().float().contiguous()
        out = torch.empty(M, E, dtype=torch.float32, device=z.device)
        check(lib().vg_embed_fuse_fwd(ptr(ids), ptr(z), z.stride(0), ptr(embd), emb.shape[0], E, ptr(wfd), ptr(bfd), D,
                                      ptr(lengths), int(T), ptr(out), M, stream()), "vg_embed_fuse_fwd")
        ctx.save_for_backward(ids, z, wfd, bfd, lengths)
        ctx.params = (emb, wf, bf)
        ctx.meta = (int(T), emb.shape[0], E)
        return out

    @staticmethod
    def backward(ctx, dout):
        ids, z, wfd, bfd, lengths = ctx.saved_tensors
        emb, wf, bf = ctx.params
        T, vocab, E = ctx.meta
        M, D = z.shape
        dout = _as(dout, torch.float32)
        need_z = ctx.needs_input_grad[1]
        dz = torch.empty(M, D, dtype=torch.float32, device=z.device) if need_z else None
        g_emb = None
        demb = None
        if ctx.needs_input_grad[2]:
            if _sinkable(emb):
                demb = _grad_buffer(emb)
            else:
                demb = g_emb = torch.zeros(vocab, E, dtype=torch.float32, device=z.device)
        nb = lib().vg_embed_fuse_blocks(M)
        part = torch.empty(nb, E * (D + 1), dtype=torch.float32, device=z.device)
        check(lib().vg_embed_fuse_bwd(ptr(dout), ptr(ids), ptr(z), z.stride(0), vocab, E, ptr(wfd), ptr(bfd), D,
                                      ptr(lengths), T, ptr(demb), ptr(dz), D, ptr(part), M, stream()),
              "vg_embed_fuse_bwd")
        if demb is not None and g_emb is None:
            _fire(emb)
        folded = colsum(part).view(E, D + 1)
        g_wf = folded[:, :D].contiguous().view_as(wf) if ctx.needs_input_grad[3] else None
        g_bf = folded[:, D].contiguous().view_as(bf) if (bf is not None and ctx.needs_input_grad[4]) else None
        return None, dz, g_emb, g_wf, g_bf, None, None


def embed_fuse_train(ids, z, emb, wf, bf, lengths=None, T=0):
    """ids int64 [M], z [M, D<=8] -> fp32 [M, E]."""
    return EmbedFuseFn.apply(ids, z, emb, wf, bf, lengths, T)


# ---------------------------------------------------------------- gradient sink
# Weight gradients of the fused Transformer layer are accumulated by the wgrad GEMM
# DIRECTLY into ``param.grad`` (fp32; split-K atomics or a read-modify-write epilogue),
# instead of materialising dW and letting autograd add it: no zero-fill, no extra add
# pass, and under data parallelism ``param.grad`` is already a view into the flat
# all-reduce bucket.  Hooks registered in ``param._vg_grad_hooks`` (the DP reducer) are
# fired by hand because autograd's AccumulateGrad node never runs for these tensors.


def set_grad_sink(on: bool) -> None:
    global _GRAD_SINK
    _GRAD_SINK = bool(on)


def _sinkable(p) -> bool:
    return _GRAD_SINK and isinstance(p, torch.nn.Parameter) and p.requires_grad and p.dtype == torch.float32


def _grad_buffer(p: Tensor) -> Tensor:
    if p.grad is None:
        p.grad = torch.zeros_like(p, memory_format=torch.contiguous_format)
    return p.grad


def _fire(p: Tensor) -> None:
    p._vg_sunk = True            # this parameter's gradient is written by the library, not by AccumulateGrad
    _WPASS["epoch"] += 1         # (a library write outside any trainer pass also ends "the buffers hold zeros")
    hooks = getattr(p, "_vg_grad_hooks", ())
    if hooks and on_side_stream():
        # written from the side branch: whoever acts on the report (the reducer puts buckets on the wire behind the
        # CURRENT stream) must see this write -- report on the main stream, after it has waited for the side stream
        side = torch.cuda.current_stream()
        main = _MAIN_OF_SIDE.get(side.cuda_stream)
        if main is not None:
            main.wait_stream(side)
            with torch.cuda.stream(main):
                for h in hooks:
                    h(p)
            return
    for h in hooks:
        h(p)


def sink_wgrad(p: Tensor, dy: Tensor, x: Tensor, bias: Optional[Tensor] = None, fire: bool = True) -> None:
    """p.grad viewed as [N,K] += dy[M,N]^T x[M,K] (conv-shaped [N,K,1] weights share that memory); with
    ``bias`` also bias.grad[N] += column sums of dy, computed by the same launch from the dy tiles it
    already staged in LDS."""
    N = p.shape[0]
    K = p.numel() // N
    M = x.shape[0]
    if (bias is None and _WDEFER["on"] and not on_side_stream() and x.dim() == 2 and x.shape[1] == K and p.is_contiguous()
            and _wgrad_item_ok(p, dy, x, 0)):
        _wgrad_enqueue([(p, dy, x, 0)], fire, "misc")      # leaves with the other products of this backward piece
        return
    g = _grad_buffer(p).view(N, K)
    _WPASS["written"].add(_wgrad_region(g))
    s = wgrad_splits(N, K, M, x.dtype)
    bg = None if bias is None else _grad_buffer(bias).view(-1)
    gemm(dy, x, N, K, M, a_tr=True, b_tr=True, out=g, split_k=s, accumulate=(s == 1), colsum_out=bg)
    if fire:
        _fire(p)
        if bias is not None:
            _fire(bias)


_WGRAD_SIDE = {}


class WgradStream:
    """Weight gradients are off the critical path of backward (nothing consumes them before the optimizer), so a
    layer's backward can issue them on a second HIP stream: the dgrad chain on the main stream and the wgrad
    launches then fill each other's tails (last partial wave of blocks, the atomic-add phase of a split-K
    epilogue, the gap between dependent launches).  Fork/join are stream waits, so the pattern is captured into a
    hipGraph as two parallel branches.  Join before the tensors the side launches read can die."""

    def __init__(self, device, enabled: bool):
        self.enabled = bool(enabled) and device.type == "cuda"
        self.pending = []
        if self.enabled:
            self.main = torch.cuda.current_stream(device)
            side = _WGRAD_SIDE.get(device)
            if side is None:
                side = _WGRAD_SIDE[device] = torch.cuda.Stream(device=device)
            self.side = side

    def wgrad(self, weight, bias, g_out: Tensor, inp: Tensor):
        """Same contract as :func:`wgrad_pair`; the weight-gradient GEMM goes to the side stream when it is sunk."""
        # inside a deferral bracket the product is only QUEUED: the queue owns the "gradient ready" report (it fires
        # when the launch exists), so the side stream and its early join-time report are bypassed
        if not (self.enabled and _sinkable(weight) and weight.is_contiguous()) or _WDEFER["on"]:
            return wgrad_pair(weight, bias, g_out, inp)
        self.side.wait_stream(self.main)             # g_out / inp were produced on the main stream
        with torch.cuda.stream(self.side):
            sink_wgrad(weight, g_out, inp, None, fire=False)
        self.pending.append(weight)
        return None, vec_grad(bias, g_out)           # column sums stay on the main stream

    def join(self) -> None:
        if self.enabled and self.pending:
            self.main.wait_stream(self.side)
        for p in self.pending:                       # gradient-ready reports only after the join
            _fire(p)
        self.pending = []


def wgrad_pair(weight, bias, g_out: Tensor, inp: Tensor):
    """(dW, db) of y = inp W^T + b for the incoming gradient ``g_out``; an entry is None where the
    gradient went straight into ``.grad`` (sink) or the parameter is absent."""
    if _sinkable(weight) and weight.is_contiguous():
        fused = _FUSE_BIAS_GRAD and bias is not None and _sinkable(bias)
        sink_wgrad(weight, g_out, inp, bias if fused else None)
        return None, (None if fused else vec_grad(bias, g_out))
    N = weight.shape[0]
    K = weight.numel() // N
    M = inp.shape[0]
    dW = gemm(g_out, inp, N, K, M, a_tr=True, b_tr=True, out_f32=True,
              split_k=wgrad_splits(N, K, M, inp.dtype)).view_as(weight)
    return dW, vec_grad(bias, g_out)


def sink_vector(p: Tensor, value: Tensor) -> None:
    _grad_buffer(p).add_(value.view_as(p))
    _fire(p)


_DX_COLSUM = _flag("VG_DX_COLSUM", "1")         # rmsnorm_bwd leaves the column sums of its dx for the layer below


def rmsnorm_bwd_raw(dy, x, sc, rstd, dx_add, lengths, T, dx_colsum: bool = False):
    """(dx, per-block partial sums of the scale gradient).  ``dx_colsum``: the launch also leaves per-block column sums of
    the dx it stores, attached to the returned tensor as ``dx._vg_colparts`` ([nblocks, C] fp32): dx is the incoming
    gradient of the node below, which can take its bias gradient from there instead of re-reading the tensor."""
    M, Cc = x.shape
    dx = torch.empty_like(x)
    nb = lib().vg_rmsnorm_bwd_blocks(M)
    part = torch.empty((nb, Cc), dtype=torch.float32, device=x.device)
    vec = 8 if x.dtype == torch.bfloat16 else 4
    if dx_colsum and _DX_COLSUM and Cc // vec <= 128:
        cpart = torch.empty((nb, Cc), dtype=torch.float32, device=x.device)
        check(lib().vg_rmsnorm_bwd_colsum(ptr(dy), ptr(x), ptr(sc), ptr(rstd), ptr(dx_add), ptr(dx), ptr(part), ptr(cpart), M,
                                          Cc, ptr(lengths), int(T), dtype_id(x.dtype), stream()), "vg_rmsnorm_bwd_colsum")
        dx._vg_colparts = cpart
        # the sums describe THESE bytes: if autograd later adds a second consumer's gradient into the tensor in place
        # (InputBuffer / AccumulateGrad keep the Python object), version or address no longer match and the consumer
        # falls back to its own column sums
        dx._vg_colparts_of = (dx.data_ptr(), dx._version)
        return dx, part
    check(lib().vg_rmsnorm_bwd(ptr(dy), ptr(x), ptr(sc), ptr(rstd), ptr(dx_add), ptr(dx), ptr(part), M, Cc,
                               ptr(lengths), int(T), dtype_id(x.dtype), stream()), "vg_rmsnorm_bwd")
    return dx, part


def rmsnorm_fwd_raw(x, sc, eps, lengths, T):
    M, Cc = x.shape
    y = torch.empty_like(x)
    rstd = torch.empty((M,), dtype=torch.float32, device=x.device)
    check(lib().vg_rmsnorm_fwd(ptr(x), ptr(sc), ptr(y), ptr(rstd), M, Cc, float(eps), ptr(lengths), int(T),
                               dtype_id(x.dtype), stream()), "vg_rmsnorm_fwd")
    return y, rstd


# ---------------------------------------------------------------- fused pre-LN Transformer layer
class TransformerLayerFn(torch.autograd.Function):
    """One reference ``TransformerLayer.forward`` (modules/transformer/layers.py:41-93) as a single
    autograd node: 6 kernels forward, 12 backward, residual-gradient adds fused into the RMSNorm
    backward, weight gradients sunk into ``param.grad``.

    Incoming gradients are zero on padded frames by construction (every loss term is masked and
    all ops are row-local or causal), which is what lets the un-masked wgrad /
    bias-grad reductions match the reference's masked ones."""

    # every launch of the node carries the measurement scope "Transformer layer" (vg_prof_tag: bench.py reports the
    # attention + FFN path of the north_star apart from the conv stacks and heads)
    @staticmethod
    def forward(ctx, *args):
        import hipvg
        prev = hipvg.prof_tag(hipvg.PROF_TAG_LAYER)
        try:
            return TransformerLayerFn._forward(ctx, *args)
        finally:
            hipvg.prof_tag(prev)

    @staticmethod
    def backward(ctx, dy):
        import hipvg
        prev = hipvg.prof_tag(hipvg.PROF_TAG_LAYER)
        try:
            return TransformerLayerFn._backward(ctx, dy)
        finally:
            hipvg.prof_tag(prev)

    @staticmethod
    def _forward(ctx, x, n1s, wqkv, bqkv, wo, bo, n3s, w1, b1, w2, b2, slopes, lengths, B, T, H, eps, pack=None):
        M, D = x.shape
        F_ = w1.shape[0]
        dt = x.dtype
        if pack is not None:         # packed rows: every row is a valid frame (or an all-zero row): no row masks
            assert M == pack.rows and pack.T == T
            lengths = None
        sq, so, s1, s2 = shadow(wqkv, dt), shadow(wo, dt), shadow(w1, dt), shadow(w2, dt)
        sc1, sc3 = n1s.detach().float().contiguous(), n3s.detach().float().contiguous()
        f32 = lambda b: None if b is None else b.detach().float()
        n1, rstd1 = rmsnorm_fwd_raw(x, sc1, eps, lengths, T)
        qkv = gemm(n1, sq, M, 3 * D, D, bias=f32(bqkv))
        att = torch.empty((M, D), dtype=dt, device=x.device)
        if pack is None:
            lse = attn_workspace(B, T, H, M, x.device)
            attn_fwd_raw(qkv, att, lse, slopes, B, T, H, lengths)
        else:
            lse = attn_workspace(pack.nseq, T, H, M, x.device)
            attn_fwd_raw(qkv, att, lse, slopes, pack.nseq, T, H, pack.lengths, pack.cu, M)
        x1 = gemm(att, so, M, D, D, bias=f32(bo), residual=x, lengths=lengths, T=T)
        n3, rstd3 = rmsnorm_fwd_raw(x1, sc3, eps, lengths, T)
        if _STORED_DERIV:
            u, act_f, dact_b = _gelu_deriv_buffer(M, F_, dt, x.device)     # bf16: one byte per element (round 6)
        else:
            u, act_f, dact_b = torch.empty((M, F_), dtype=dt, device=x.device), ACT_GELU, ACT_GELU
        h = gemm(n3, s1, M, F_, D, bias=f32(b1), act=act_f, aux_out=u)     # u = GELU'(pre-activation) (or the pre-activation itself)
        y = gemm(h, s2, M, D, F_, bias=f32(b2), residual=x1, lengths=lengths, T=T)
        ctx.save_for_backward(x, n1, rstd1, qkv, att, lse, x1, n3, rstd3, u, h, sq, so, s1, s2, sc1, sc3, slopes,
                              lengths)
        ctx.params = (n1s, wqkv, bqkv, wo, bo, n3s, w1, b1, w2, b2)
        ctx.dims = (B, T, H)
        ctx.dact_b = dact_b
        ctx.pack = pack
        return y

    @staticmethod
    def _backward(ctx, dy):
        (x, n1, rstd1, qkv, att, lse, x1, n3, rstd3, u, h, sq, so, s1, s2, sc1, sc3, slopes,
         lengths) = ctx.saved_tensors
        n1s, wqkv, bqkv, wo, bo, n3s, w1, b1, w2, b2 = ctx.params
        B, T, H = ctx.dims
        pack = ctx.pack
        M, D = x.shape
        F_ = s1.shape[0]
        dt = x.dtype
        dy = _as(dy, dt)
        grads = {}

        def vgrad(p, value_fn):
            if p is None:
                return None
            v = value_fn()
            if _sinkable(p):
                sink_vector(p, v)
                return None
            return v

        # ---- FFN
        want_part = b1 is not None and ctx.needs_input_grad[8] and _COLPART
        parts = [] if want_part else None
        du = gemm(dy, s2, M, F_, D, b_tr=True, dact=ctx.dact_b, aux_in=u,
                  colpart=parts)                 # + column sums of du per row tile (b1's gradient) for free
        ws = WgradStream(x.device, _WGRAD_STREAM)
        # the layer's four weight gradients run as ONE grouped launch at the end of this node when they can be sunk
        group = [] if (_PH_GROUP and not ws.enabled and dt == torch.bfloat16 and
                       all(_sinkable(w) and w.is_contiguous() for w in (w1, w2, wo, wqkv))) else None

        big = []                                 # (bias, gradient rows): first-stage column sums in ONE launch below

        def wgrad(weight, bias, g_out, inp):
            if group is None:
                return ws.wgrad(weight, bias, g_out, inp)
            group.append((weight, g_out, inp))
            if bias is not None and _COLSUM_MULTI and _COLSUM_BIG:
                big.append((bias, g_out))
                return None, None
            return None, vec_grad(bias, g_out)
        # b2's gradient = column sums of dy.  When dy is the dx of the rmsnorm_bwd launch of the node above (the layer
        # above's first norm, or the stack's final norm) that launch has left them (`_vg_colparts`): no pass over dy
        dy_parts = getattr(dy, "_vg_colparts", None)
        if dy_parts is not None and (dy_parts.shape[1] != D or not _sinkable(b2)
                                     or getattr(dy, "_vg_colparts_of", None) != (dy.data_ptr(), dy._version)):
            dy_parts = None
        g_w2, g_b2 = wgrad(w2, None if dy_parts is not None else b2, dy, h)
        dn3 = gemm(du, s1, M, D, F_, b_tr=True)
        small = []                               # (parameter, fp32 partial sums): folded by one launch at the end
        g_b1 = None
        if want_part and parts[0] is not None:
            g_w1, _ = wgrad(w1, None, du, n3)
            small.append((b1, parts[0]))
        else:
            g_w1, g_b1 = wgrad(w1, b1, du, n3)
        dx1, ds3 = rmsnorm_bwd_raw(dn3, x1, sc3, rstd3, dy, lengths, T)
        small.append((n3s, ds3))
        # ---- attention
        datt = gemm(dx1, so, M, D, D, b_tr=True)
        g_wo, g_bo = wgrad(wo, bo, dx1, att)
        dqkv = torch.empty_like(qkv)
        delta = torch.empty((H, M), dtype=torch.float32, device=x.device)
        if pack is None:
            attn_bwd_raw(qkv, att, datt, lse, slopes, dqkv, delta, B, T, H, lengths)
        else:
            attn_bwd_raw(qkv, att, datt, lse, slopes, dqkv, delta, pack.nseq, T, H, pack.lengths, pack.cu, M)
        dn1 = gemm(dqkv, sq, M, D, 3 * D, b_tr=True)
        g_wq, g_bq = wgrad(wqkv, bqkv, dqkv, n1)
        dx, ds1 = rmsnorm_bwd_raw(dn1, x, sc1, rstd1, dx1, lengths, T, dx_colsum=True)
        small.append((n1s, ds1))
        if dy_parts is not None:
            small.append((b2, dy_parts))
        if big:                                  # b2 <- dy, bo <- dx1, bqkv <- dqkv (and b1 <- du without colpart)
            parts_big = colsum_partials([g for _, g in big])
            if parts_big is None:
                parts_big = [None] * len(big)
            for (bp, g), q in zip(big, parts_big):
                if q is None:
                    folded_one = vec_grad(bp, g)
                    if bp is b2: g_b2 = folded_one
                    elif bp is bo: g_bo = folded_one
                    elif bp is bqkv: g_bq = folded_one
                    elif bp is b1: g_b1 = folded_one
                else:
                    small.append((bp, q))
        folded = dict(zip((id(p) for p, _ in small), vec_grads(small)))
        g_n1, g_n3 = folded[id(n1s)], folded[id(n3s)]
        if b1 is not None and id(b1) in folded:
            g_b1 = folded[id(b1)]
        if b2 is not None and id(b2) in folded:
            g_b2 = folded[id(b2)]
        if bo is not None and id(bo) in folded:
            g_bo = folded[id(bo)]
        if bqkv is not None and id(bqkv) in folded:
            g_bq = folded[id(bqkv)]
        if group is not None:
            sink_wgrad_group(group, tag="layer")
        ws.join()                                # before qkv / att / du ... can be released
        return (dx, g_n1, g_wq, g_bq, g_wo, g_bo, g_n3, g_w1, g_b1, g_w2, g_b2,
                None, None, None, None, None, None, None)


def transformer_layer(x, n1s, wqkv, bqkv, wo, bo, n3s, w1, b1, w2, b2, slopes, lengths, B, T, H, eps, pack=None):
    return TransformerLayerFn.apply(x, n1s, wqkv, bqkv, wo, bo, n3s, w1, b1, w2, b2, slopes, lengths, B, T, H, eps, pack)


# ---------------------------------------------------------------- conv bottleneck block (channels-last)
def dwnorm_fwd_raw(x, w, cb, te, gamma, beta, T, taps, shift, eps):
    """``T``: frames per sequence, or a ``PackPlan`` (packed rows: ragged sequences laid end to end)."""
    M, Cc = x.shape
    y = torch.empty_like(x)
    mean = torch.empty((M,), dtype=torch.float32, device=x.device)
    rstd = torch.empty((M,), dtype=torch.float32, device=x.device)
    if isinstance(T, PackPlan):
        if taps == 0:        # the norm alone is row-local
            T = 1
        else:
            check(lib().vg_dwnorm_fwd_seg(ptr(x), ptr(w), ptr(cb), ptr(te), ptr(gamma), ptr(beta), ptr(y), ptr(mean),
                                          ptr(rstd), M, Cc, ptr(T.cu), T.nseq, T.B, int(taps), int(shift), float(eps),
                                          dtype_id(x.dtype), stream()), "vg_dwnorm_fwd_seg")
            return y, mean, rstd
    check(lib().vg_dwnorm_fwd(ptr(x), ptr(w), ptr(cb), ptr(te), ptr(gamma), ptr(beta), ptr(y), ptr(mean),
                              ptr(rstd), M, Cc, int(T), int(taps), int(shift), float(eps), dtype_id(x.dtype),
                              stream()), "vg_dwnorm_fwd")
    return y, mean, rstd


# round 6: the backward of a conv block's (depthwise conv -> norm) pair as ONE launch that keeps du in LDS
# (include/vaegslm_hip.h: vg_dwnorm_bwd_fused).  VG_DW_FUSED=0: the two run kernels, for A/B runs.
_DW_FUSED = _flag("VG_DW_FUSED", "1")


def _dwnorm_bwd_fused(dy, x, w, cb, te, gamma, mean, rstd, dx_add, T, taps, shift, want_du=True, du_sums=False):
    """-> (du or None, dx, gamma partials, beta partials, tap partials[, per-sequence column sums of du: fp32 [nseq, C]])."""
    M, Cc = x.shape
    plan = T if isinstance(T, PackPlan) else None
    nseq, max_len = (plan.nseq, plan.T) if plan else (M // int(T), int(T))
    nb = lib().vg_dwnorm_bwd_fused_blocks(nseq, max_len)
    du = torch.empty_like(x) if want_du else None
    dx = torch.empty_like(x)
    npart = torch.empty((nb, 2 * Cc), dtype=torch.float32, device=x.device)
    wpart = torch.empty((nb, Cc * taps), dtype=torch.float32, device=x.device)
    dupart = torch.empty((nb, Cc), dtype=torch.float32, device=x.device) if du_sums else None
    check(lib().vg_dwnorm_bwd_fused(ptr(dy), dy.stride(0), ptr(x), ptr(w), ptr(cb), ptr(te), ptr(gamma), ptr(mean), ptr(rstd),
                                    ptr(dx_add), ptr(du), ptr(dx), ptr(npart), ptr(wpart), ptr(dupart), M, Cc, 0 if plan else int(T),
                                    ptr(plan.cu) if plan else None, plan.nseq if plan else 0, plan.B if plan else 0, max_len,
                                    int(taps), int(shift), dtype_id(x.dtype), stream()), "vg_dwnorm_bwd_fused")
    if du_sums == "partials":      # fp32 [nseq * bps, C]: blocks [s * bps, (s + 1) * bps) are sequence s
        return du, dx, npart[:, :Cc], npart[:, Cc:], wpart, dupart
    if du_sums:
        return du, dx, npart[:, :Cc], npart[:, Cc:], wpart, segment_colsum(dupart, nseq)
    return du, dx, npart[:, :Cc], npart[:, Cc:], wpart


def dwnorm_bwd_block(dy, x, w, cb, te, gamma, mean, rstd, dx_add, T, taps, shift, need_dte=True):
    """The (depthwise conv -> norm) backward as a conv block needs it: (dx, per-sequence column sums of du [B, C] or None,
    gamma / beta / tap partial sums, rows whose column sum over ALL rows is the conv-bias gradient).  One launch without a du
    tensor where vg_dwnorm_bwd_fused runs -- the per-sequence sums are then segments of its per-block partial rows, folded
    only for a block that has a time embedding; else the two run kernels, du through HBM and a pass over it."""
    nb_seq = T.B if isinstance(T, PackPlan) else x.shape[0] // int(T)
    if _dw_fused_ok(dy, x, T, taps, shift):
        _, dx, pg, pb, pw, dupart = _dwnorm_bwd_fused(dy, x, w, cb, te, gamma, mean, rstd, dx_add, T, taps, shift, want_du=False,
                                                      du_sums="partials")
        nseq = T.nseq if isinstance(T, PackPlan) else nb_seq
        dte = segment_colsum(dupart, nseq)[:nb_seq] if need_dte else None
        return dx, dte, pg, pb, pw, dupart
    wide = dy.stride(0) != x.shape[1]
    dv, dx, pg, pb, pw = (dwnorm_bwd_ld_raw if wide else dwnorm_bwd_raw)(dy, x, w, cb, te, gamma, mean, rstd, dx_add, T, taps, shift)
    dte = segment_colsum(dv, T)[:nb_seq] if isinstance(T, PackPlan) else segment_colsum(dv, nb_seq)
    return dx, dte, pg, pb, pw, dte


def _dw_fused_ok(dy, x, T, taps, shift) -> bool:
    return (_DW_FUSED and x.dtype == torch.bfloat16 and dy.dtype == torch.bfloat16 and x.shape[1] == 512 and taps == 7
            and 0 <= shift <= 6 and dy.stride(1) == 1 and dy.stride(0) % 8 == 0 and dy.data_ptr() % 16 == 0
            and (isinstance(T, PackPlan) or (int(T) > 0 and x.shape[0] % int(T) == 0)))


def dwnorm_bwd_raw(dy, x, w, cb, te, gamma, mean, rstd, dx_add, T, taps, shift):
    M, Cc = x.shape
    if _dw_fused_ok(dy, x, T, taps, shift):
        return _dwnorm_bwd_fused(dy, x, w, cb, te, gamma, mean, rstd, dx_add, T, taps, shift)
    nb = lib().vg_dwnorm_blocks(M)
    du = torch.empty_like(x)
    dx = torch.empty_like(x) if taps > 0 else du
    npart = torch.empty((nb, 2 * Cc), dtype=torch.float32, device=x.device)
    wpart = torch.empty((nb, Cc * max(taps, 1)), dtype=torch.float32, device=x.device)
    if isinstance(T, PackPlan) and taps > 0:
        check(lib().vg_dwnorm_bwd_seg(ptr(dy), ptr(x), ptr(w), ptr(cb), ptr(te), ptr(gamma), ptr(mean), ptr(rstd),
                                      ptr(dx_add), ptr(du), ptr(dx), ptr(npart), ptr(wpart), M, Cc, ptr(T.cu), T.nseq, T.B,
                                      int(taps), int(shift), dtype_id(x.dtype), stream()), "vg_dwnorm_bwd_seg")
    else:
        if isinstance(T, PackPlan):
            T = 1
        check(lib().vg_dwnorm_bwd(ptr(dy), ptr(x), ptr(w), ptr(cb), ptr(te), ptr(gamma), ptr(mean), ptr(rstd),
                                  ptr(dx_add), ptr(du), ptr(dx), ptr(npart), ptr(wpart), M, Cc, int(T), int(taps),
                                  int(shift), dtype_id(x.dtype), stream()), "vg_dwnorm_bwd")
    return du, dx, npart[:, :Cc], npart[:, Cc:], (wpart if taps > 0 else None)


# round 6: the conditioning of a conv block rides in its first 1x1 convolution (one K = C + 64 product instead of a K = C
# product that adds what a K = 32 product wrote first; include/vaegslm_hip.h: vg_dwnorm_fwd_cat).  VG_COND_MERGE=0: the
# round-5 form (pre-activation operand), for A/B runs.
_COND_MERGE = _flag("VG_COND_MERGE", "1")


def _cond_merge_ok(x: Tensor, cond, taps: int) -> bool:
    return (_COND_MERGE and cond is not None and x.dtype == torch.bfloat16 and x.shape[1] == 512 and taps == 7
            and cond.dtype == torch.bfloat16 and cond.dim() == 2 and cond.shape[1] % 8 == 0 and 0 < cond.shape[1] <= 64
            and cond.stride(1) == 1 and cond.stride(0) % 8 == 0 and cond.data_ptr() % 16 == 0)


def _padded_weight(owner: Tensor, s2: Tensor, Kx: int) -> Tensor:
    """The bf16 [Hd, C + cond] weight of the 1x1 convolution over [activations ; condition] as [Hd, Kx] with zero columns
    up to Kx = C + 64 (whole 64-deep K tiles for the phase-pipelined GEMM kernels).  One buffer per weight, kept on the
    PARAMETER (`owner`: the bf16 view handed in is a fresh tensor object on every call), its zero tail written once; the
    live columns are re-copied on every call (one 2.4 MB launch: the optimizer rewrites the source, and inside a hipGraph
    the copy is a node like any other)."""
    Hd, Kw = s2.shape
    buf = getattr(owner, "_vg_kpad", None)
    if buf is None or buf.shape != (Hd, Kx) or buf.device != s2.device or buf.dtype != s2.dtype:
        buf = torch.zeros((Hd, Kx), dtype=s2.dtype, device=s2.device)
        owner._vg_kpad = buf
    buf[:, :Kw].copy_(s2)
    return buf


def dwnorm_fwd_cat_raw(x, w, cb, te, gamma, beta, T, taps, shift, eps, cond, Kx):
    """dwnorm_fwd_raw whose output rows are Kx wide: [norm(dwconv(x) + te) ; cond ; 0] (vg_dwnorm_fwd_cat)."""
    M, Cc = x.shape
    y = torch.empty((M, Kx), dtype=x.dtype, device=x.device)
    mean = torch.empty((M,), dtype=torch.float32, device=x.device)
    rstd = torch.empty((M,), dtype=torch.float32, device=x.device)
    plan = T if isinstance(T, PackPlan) else None
    check(lib().vg_dwnorm_fwd_cat(ptr(x), ptr(w), ptr(cb), ptr(te), ptr(gamma), ptr(beta), ptr(y), Kx, ptr(cond),
                                  cond.stride(0), cond.shape[1], ptr(mean), ptr(rstd), M, Cc, 0 if plan else int(T),
                                  ptr(plan.cu) if plan else None, plan.nseq if plan else 0, plan.B if plan else 0,
                                  int(taps), int(shift), float(eps), dtype_id(x.dtype), stream()), "vg_dwnorm_fwd_cat")
    return y, mean, rstd


def dwnorm_bwd_ld_raw(dy, x, w, cb, te, gamma, mean, rstd, dx_add, T, taps, shift):
    """dwnorm_bwd_raw for an incoming gradient that is the first C columns of wider rows (vg_dwnorm_bwd_ld)."""
    M, Cc = x.shape
    assert dy.shape[0] == M and dy.stride(1) == 1 and dy.stride(0) >= Cc
    if _dw_fused_ok(dy, x, T, taps, shift):
        return _dwnorm_bwd_fused(dy, x, w, cb, te, gamma, mean, rstd, dx_add, T, taps, shift)
    nb = lib().vg_dwnorm_blocks(M)
    du = torch.empty_like(x)
    dx = torch.empty_like(x)
    npart = torch.empty((nb, 2 * Cc), dtype=torch.float32, device=x.device)
    wpart = torch.empty((nb, Cc * taps), dtype=torch.float32, device=x.device)
    plan = T if isinstance(T, PackPlan) else None
    check(lib().vg_dwnorm_bwd_ld(ptr(dy), dy.stride(0), ptr(x), ptr(w), ptr(cb), ptr(te), ptr(gamma), ptr(mean), ptr(rstd),
                                 ptr(dx_add), ptr(du), ptr(dx), ptr(npart), ptr(wpart), M, Cc, 0 if plan else int(T),
                                 ptr(plan.cu) if plan else None, plan.nseq if plan else 0, plan.B if plan else 0,
                                 int(taps), int(shift), dtype_id(x.dtype), stream()), "vg_dwnorm_bwd_ld")
    return du, dx, npart[:, :Cc], npart[:, Cc:], wpart


def conv_rows_packable(x: Tensor, taps: int) -> bool:
    """Can a conv block of this shape run on packed rows (``vg_dwnorm_*_seg``: bf16, 512 channels, 7 taps)?"""
    return x.dtype == torch.bfloat16 and x.shape[1] == 512 and taps == 7


def _sink_or_return(p, value):
    """value: fp32 tensor shaped like p (or broadcastable view)."""
    if p is None:
        return None
    if _sinkable(p):
        sink_vector(p, value)
        return None
    return value.view_as(p)


class ConvBlockFn(torch.autograd.Function):
    """x + conv3(act(conv2([norm(dwconv(x) + t_emb) ; cond])))  -- one bottleneck block of the posterior
    encoder / diffusion UNet (reference modules/conv/layers.py:70-135,231-295) on [B*T, C] rows:
    fused depthwise-conv+norm row kernel, the two 1x1 convolutions as MFMA GEMMs (conditioning enters
    as a pre-activation add, the residual in the second GEMM's epilogue)."""

    @staticmethod
    def forward(ctx, x, te, cond, c1w, c1b, nw, nb_, c2w, c2b, c3w, c3b, T, taps, shift, eps, act):
        M, Cc = x.shape
        dt = x.dtype
        Hd = c2w.shape[0]
        w1 = c1w.detach().float().reshape(Cc, taps).contiguous()
        cb = c1b.detach().float().contiguous()
        gamma, beta = nw.detach().float().contiguous(), nb_.detach().float().contiguous()
        te32 = None if te is None else te.detach().float().contiguous()
        s2 = shadow(c2w, dt).view(Hd, -1)
        s3 = shadow(c3w, dt).view(Cc, Hd)
        # pre receives act'(pre-activation) (ReLU: the output itself carries it)
        pre = torch.empty((M, Hd), dtype=dt, device=x.device) if act != ACT_RELU else None
        merged = _cond_merge_ok(x, cond, taps)
        if merged:
            # one product over K = C + 64: the norm kernel appends the condition channels (and zeros) to its output rows,
            # the weight is zero-padded to the same width (round 6; was: K = C product + a pre-activation operand written
            # by a K = 32 product)
            Kx = Cc + 64
            u, mean, rstd = dwnorm_fwd_cat_raw(x, w1, cb, te32, gamma, beta, T, taps, shift, eps, cond, Kx)
            # (kept on ctx, not among the saved tensors: the buffer is rewritten -- with the same values -- by the next
            # forward of this block, which autograd's version check would take for a hazard)
            ctx.wpad = _padded_weight(c2w, s2, Kx)
            h = gemm(u, ctx.wpad, M, Hd, Kx, bias=c2b.detach().float(),
                     act=(act | ACT_SAVE_DERIV) if pre is not None else act, aux_out=pre)
        else:
            u, mean, rstd = dwnorm_fwd_raw(x, w1, cb, te32, gamma, beta, T, taps, shift, eps)
            Wa = s2[:, :Cc]
            pre_add = None
            if cond is not None:
                Wc = s2[:, Cc:]
                pre_add = gemm(cond, Wc, M, Hd, cond.shape[1])
            h = gemm(u, Wa, M, Hd, Cc, bias=c2b.detach().float(), pre_add=pre_add,
                     act=(act | ACT_SAVE_DERIV) if pre is not None else act, aux_out=pre)
        y = gemm(h, s3, M, Cc, Hd, bias=c3b.detach().float(), residual=x)
        ctx.save_for_backward(x, u, mean, rstd, h, pre, cond, s2, s3, w1, cb, te32, gamma)
        ctx.params = (c1w, c1b, nw, nb_, c2w, c2b, c3w, c3b)
        ctx.meta = (T, taps, shift, act, te is not None, merged)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, u, mean, rstd, h, pre, cond, s2, s3, w1, cb, te32, gamma = ctx.saved_tensors
        c1w, c1b, nw, nb_, c2w, c2b, c3w, c3b = ctx.params
        T, taps, shift, act, has_te, merged = ctx.meta
        M, Cc = x.shape
        Hd = s2.shape[0]
        dt = x.dtype
        dy = _as(dy, dt)
        Wa = s2[:, :Cc]
        parts = [] if _COLPART else None
        dpre = gemm(dy, s3, M, Hd, Cc, b_tr=True, dact=(ACT_RELU if act == ACT_RELU else ACT_STORED),
                    aux_in=(h if act == ACT_RELU else pre), colpart=parts)   # + column sums of dpre (c2b's gradient)

        fused_bias = set()

        def wgrad_into(p, rows, col0, cols, g_out, inp, bias=None):
            """p.grad[:, col0:col0+cols] (+)= g_out^T inp (+ the bias gradient from the same launch), or
            return the dense gradient."""
            if _sinkable(p):
                g = _grad_buffer(p).view(rows, -1)[:, col0:col0 + cols]
                _WPASS["written"].add(_wgrad_region(g))
                s = wgrad_splits(rows, cols, M, dt)
                bg = None
                if _FUSE_BIAS_GRAD and bias is not None and _sinkable(bias):
                    bg = _grad_buffer(bias).view(-1)
                    fused_bias.add(id(bias))
                gemm(g_out, inp, rows, cols, M, a_tr=True, b_tr=True, out=g, split_k=s, accumulate=(s == 1),
                     colsum_out=bg)
                if bg is not None:
                    _fire(bias)
                return None
            return gemm(g_out, inp, rows, cols, M, a_tr=True, b_tr=True, out_f32=True,
                        split_k=wgrad_splits(rows, cols, M, dt))

        # both (or, with a conditioning input, all three) weight gradients as ONE grouped launch when they can be sunk
        grouped = (_PH_GROUP and not _FUSE_BIAS_GRAD and dt == torch.bfloat16 and _sinkable(c2w) and _sinkable(c3w)
                   and c2w.is_contiguous() and c3w.is_contiguous())
        group = []
        if grouped:
            g_c3 = None
            group.append((c3w, dy, h, 0))
        else:
            g_c3 = wgrad_into(c3w, Cc, 0, Hd, dy, h, c3b)
        g_c3b = None if id(c3b) in fused_bias else vec_grad(c3b, dy)
        dcond = gc = None
        if merged:
            # (ctx.wpad is the zero-padded [Hd, C + 64] weight, u the [M, C + 64] rows [norm output ; cond ; 0]): one dgrad gives
            # d(norm output) and d(cond) side by side, one weight gradient covers both column ranges of c2w
            Kc = cond.shape[1]
            du_x = gemm(dpre, ctx.wpad, M, Cc + 64, Hd, b_tr=True)
            du, dcond = du_x[:, :Cc], du_x[:, Cc:Cc + Kc]
            if grouped:
                ga = None
                group.append((c2w, dpre, u[:, :Cc + Kc], 0))
            else:
                ga = wgrad_into(c2w, Hd, 0, Cc + Kc, dpre, u[:, :Cc + Kc], c2b)
        else:
            du = gemm(dpre, Wa, M, Cc, Hd, b_tr=True)
            if grouped:
                ga = None
                group.append((c2w, dpre, u, 0))
            else:
                ga = wgrad_into(c2w, Hd, 0, Cc, dpre, u, c2b)
        if cond is not None and not merged:
            Wc = s2[:, Cc:]
            Kc = cond.shape[1]
            dcond = gemm(dpre, Wc, M, Kc, Hd, b_tr=True)
            if grouped:
                group.append((c2w, dpre, cond, Cc))
            else:
                gc = wgrad_into(c2w, Hd, Cc, Kc, dpre, cond)
        if group:
            # one "gradient ready" report per weight, issued by sink_wgrad_group itself: at once when it launches, from
            # the deferral queue (after the launch exists) when it only queues -- never before the dW launch
            sink_wgrad_group(group, fire=True)
        if _sinkable(c2w):
            if not group:
                _fire(c2w)
            g_c2 = None
        else:
            g_c2 = (ga if gc is None else torch.cat([ga, gc], 1)).view_as(c2w)
        if _sinkable(c3w):
            if not group:
                _fire(c3w)
        elif g_c3 is not None:
            g_c3 = g_c3.view_as(c3w)
        # dte: per-sequence column sums of d loss / d v (the pseudo sequences' rows of a packed layout carry no gradient)
        dx, dte, pg, pb, pw, du_rows = dwnorm_bwd_block(du, x, w1, cb, te32, gamma, mean, rstd, dy, T, taps, shift,
                                                         need_dte=has_te)
        # the five small reductions of this block (conv weight / bias, norm weight / bias, c2's bias) in one launch
        g_c2b, g_c1w, g_c1b, g_nw, g_nb = vec_grads([
            (None if id(c2b) in fused_bias else c2b, parts[0] if parts and parts[0] is not None else dpre),
            (c1w, pw), (c1b, du_rows), (nw, pg), (nb_, pb)])
        return (dx, dte if has_te else None, dcond, g_c1w, g_c1b, g_nw, g_nb, g_c2, g_c2b, g_c3, g_c3b,
                None, None, None, None, None)


def conv_block(x, te, cond, c1w, c1b, nw, nb_, c2w, c2b, c3w, c3b, *, T, taps, shift, eps, act):
    return ConvBlockFn.apply(x, te, cond, c1w, c1b, nw, nb_, c2w, c2b, c3w, c3b, T, taps, shift, eps,
                             ACT_IDS[act] if not isinstance(act, int) else act)


class ChannelNormFn(torch.autograd.Function):
    """Per-frame channel norm with unbiased variance (reference modules/norm.py:43-47) on [M, C] rows."""

    @staticmethod
    def forward(ctx, x, weight, bias, T, eps):
        gamma, beta = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        y, mean, rstd = dwnorm_fwd_raw(x, None, None, None, gamma, beta, T, 0, 0, eps)
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.T = T
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        du, _, pg, pb, _ = dwnorm_bwd_raw(_as(dy, x.dtype), x, None, None, None, gamma, mean, rstd, None,
                                          ctx.T, 0, 0)
        return du, colsum(pg), colsum(pb), None, None


def channel_norm(x, weight, bias, *, T, eps):
    return ChannelNormFn.apply(x, weight, bias, T, eps)


class ConvGatherFn(torch.autograd.Function):
    """rows [B*t_out, k*C] of a strided 1-D convolution's windows from x [B, T, C] (tap-major, channel-minor)."""

    @staticmethod
    def forward(ctx, x, k, stride, pad_left, pad_right):
        x = x.contiguous()
        B, T, C = x.shape
        t_out = (T + pad_left + pad_right - k) // stride + 1
        rows = torch.empty(B * t_out, k * C, dtype=x.dtype, device=x.device)
        check(lib().vg_conv_gather(ptr(x), ptr(rows), B, T, C, t_out, int(k), int(stride), int(pad_left),
                                   dtype_id(x.dtype), stream()), "vg_conv_gather")
        ctx.meta = (B, T, C, t_out, int(k), int(stride), int(pad_left))
        return rows

    @staticmethod
    def backward(ctx, drows):
        B, T, C, t_out, k, stride, pl = ctx.meta
        drows = drows.contiguous()
        dx = torch.empty(B, T, C, dtype=drows.dtype, device=drows.device)
        check(lib().vg_conv_scatter(ptr(drows), ptr(dx), B, T, C, t_out, k, stride, pl, dtype_id(drows.dtype), stream()),
              "vg_conv_scatter")
        return dx, None, None, None, None


def conv_gather(x, k, stride, pad_left, pad_right):
    return ConvGatherFn.apply(x, k, stride, pad_left, pad_right)


class NarrowChannelNormFn(torch.autograd.Function):
    """Per-frame channel norm (unbiased variance, reference modules/norm.py:35-47) with an optional fused ReLU for
    row widths the fused conv+norm kernels do not take (vg_chnorm_*)."""

    @staticmethod
    def forward(ctx, x, weight, bias, eps, relu):
        x = x.contiguous()
        M, C = x.shape
        gamma, beta = weight.detach().float().contiguous(), bias.detach().float().contiguous()
        y = torch.empty_like(x)
        mean = torch.empty(M, dtype=torch.float32, device=x.device)
        rstd = torch.empty(M, dtype=torch.float32, device=x.device)
        check(lib().vg_chnorm_fwd(ptr(x), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), M, C, float(eps),
                                  int(bool(relu)), dtype_id(x.dtype), stream()), "vg_chnorm_fwd")
        ctx.save_for_backward(x, y if relu else None, gamma, mean, rstd)
        ctx.params = (weight, bias)
        ctx.relu = bool(relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, mean, rstd = ctx.saved_tensors
        weight, bias = ctx.params
        M, C = x.shape
        dy = _as(dy, x.dtype)
        dx = torch.empty_like(x)
        nb = lib().vg_chnorm_blocks(M)
        part = torch.empty(nb, 2 * C, dtype=torch.float32, device=x.device)
        check(lib().vg_chnorm_bwd(ptr(dy), ptr(x), ptr(y), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), ptr(part), M, C,
                                  int(ctx.relu), dtype_id(x.dtype), stream()), "vg_chnorm_bwd")
        g_w, g_b = vec_grads([(weight, part[:, :C]), (bias, part[:, C:])])
        return dx, g_w, g_b, None, None


def narrow_channel_norm(x, weight, bias, *, eps, relu=False):
    return NarrowChannelNormFn.apply(x, weight, bias, eps, relu)


# ---------------------------------------------------------------- coupling flow on the latent (row kernel)
FLOW_PARAMS_PER_LAYER = 580     # W1[64][2] b1[64] ln_w[64] ln_b[64] W2[4][64] b2[4]


class FlowFn(torch.autograd.Function):
    """The conditional coupling stack (reference modules/flow/layers.py:15-98,199-245) as one forward and
    one backward row kernel; ``params`` = per layer (linear1.weight, linear1.bias, norm.weight, norm.bias,
    linear2.weight, linear2.bias).  Returns (u [M,4], logdet_sum [M])."""

    @staticmethod
    def forward(ctx, z, wb, lengths, T, eps, hi, lo, *params):
        M = z.shape[0]
        L = len(params) // 6
        assert z.dtype == torch.float32 and wb.dtype == torch.float32 and z.shape[1] == 4 and wb.stride(1) == 1
        z = z.contiguous()
        packed = torch.cat([p.detach().reshape(-1).float() for p in params])
        assert packed.numel() == L * FLOW_PARAMS_PER_LAYER
        u = torch.empty_like(z)
        logdet = torch.empty((M,), dtype=torch.float32, device=z.device)
        states = torch.empty((M, L, 4), dtype=torch.float32, device=z.device)
        check(lib().vg_flow_fwd(ptr(z), ptr(wb), wb.stride(0), ptr(packed), L, ptr(u), ptr(logdet), ptr(states), M,
                                float(eps), float(hi), float(lo), ptr(lengths), int(T), stream()), "vg_flow_fwd")
        ctx.save_for_backward(states, wb, packed, lengths)
        ctx.meta = (int(T), float(eps), float(hi), float(lo), L)
        ctx.params = params
        return u, logdet

    @staticmethod
    def backward(ctx, du, dlogdet):
        states, wb, packed, lengths = ctx.saved_tensors
        T, eps, hi, lo, L = ctx.meta
        M = states.shape[0]
        du = du.contiguous().float()
        dlogdet = dlogdet.contiguous().float()
        dz = torch.empty((M, 4), dtype=torch.float32, device=du.device)
        dwb = torch.empty_like(wb) if wb.shape[1] == 128 * L else torch.zeros_like(wb)
        nb = lib().vg_flow_blocks(M)
        part = torch.empty((nb, L * FLOW_PARAMS_PER_LAYER), dtype=torch.float32, device=du.device)
        check(lib().vg_flow_bwd(ptr(states), ptr(wb), wb.stride(0), ptr(packed), L, ptr(du), ptr(dlogdet), ptr(dz),
                                ptr(dwb), ptr(part), M, eps, hi, lo, ptr(lengths), T, stream()), "vg_flow_bwd")
        g = colsum(part)
        grads, sunk_g, sunk_v, off = [], [], [], 0
        for i, p in enumerate(ctx.params):
            n = p.numel()
            piece = g[off: off + n].view_as(p)
            off += n
            if not ctx.needs_input_grad[7 + i]:
                grads.append(None)
            elif _sinkable(p):
                sunk_g.append(_grad_buffer(p))
                sunk_v.append(piece)
                grads.append(None)
            else:
                grads.append(piece)
        if sunk_g:
            torch._foreach_add_(sunk_g, sunk_v)      # one multi-tensor launch for the 6 L small parameters
            for p in ctx.params:
                if _sinkable(p):
                    _fire(p)
        return (dz, dwb, None, None, None, None, None, *grads)


def coupling_flow(z, wb, params, *, eps, hi, lo, lengths=None, T=0):
    return FlowFn.apply(z, wb, lengths, T, eps, hi, lo, *params)


def pack_flow_params(params) -> Tensor:
    return torch.cat([p.detach().reshape(-1).float() for p in params])


def coupling_flow_reverse(u, wb, params, *, eps, hi, lo, packed: Optional[Tensor] = None,
                          mu_ls: Optional[Tensor] = None, temperature: float = 1.0, out: Optional[Tensor] = None):
    """Reverse pass of the coupling stack.  With ``mu_ls`` ([M, 8] rows = mean | logstd of the prior head) ``u``
    is unit noise and the kernel draws the Gaussian sample first; ``out`` may be a strided [M, 4] view (e.g. the
    latent columns of the decode session's frame buffer)."""
    M = u.shape[0]
    L = len(params) // 6
    u = u.contiguous().float()
    if packed is None:
        packed = pack_flow_params(params)
    z = out if out is not None else torch.empty_like(u)
    assert z.shape == (M, 4) and z.dtype == torch.float32 and z.stride(1) == 1
    check(lib().vg_flow_reverse(ptr(u), ptr(wb), wb.stride(0), ptr(packed), L, ptr(z), z.stride(0), M, float(eps),
                                float(hi), float(lo), ptr(mu_ls), 0 if mu_ls is None else mu_ls.stride(0),
                                float(temperature), stream()), "vg_flow_reverse")
    return z


# ---------------------------------------------------------------- decode step (no autograd: inference only)
def rows_linear(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, *, act: int = ACT_NONE,
                residual: Optional[Tensor] = None, out_f32: bool = False, norm_scale: Optional[Tensor] = None,
                norm_eps: float = 0.0) -> Tensor:
    """y = act(x W^T + b) + residual for a handful of rows (M <= 64, groups of 8): the HBM-bound Linear of the
    autoregressive step (vg_gemm_rows).  ``weight`` must already be in x's dtype (see :func:`shadow`)."""
    M, K = x.shape
    N = weight.shape[0]
    assert x.dtype == weight.dtype and x.stride(1) == 1 and weight.stride(1) == 1 and weight.shape[1] == K
    y = torch.empty((M, N), dtype=torch.float32 if out_f32 else x.dtype, device=x.device)
    b = None if bias is None else bias.detach().float()
    check(lib().vg_gemm_rows(ptr(x), x.stride(0), ptr(weight), weight.stride(0), ptr(b), ptr(residual),
                             0 if residual is None else residual.stride(0), ptr(y), y.stride(0), M, N, K, int(act),
                             int(out_f32), ptr(norm_scale), float(norm_eps), dtype_id(x.dtype), stream()),
          "vg_gemm_rows")
    return y


def rows_linear_mixed(x: Tensor, weight: Tensor, bias: Optional[Tensor] = None, *, act: int = ACT_NONE,
                      residual: Optional[Tensor] = None, out_f32: bool = False, norm_scale: Optional[Tensor] = None,
                      norm_eps: float = 0.0, zero: Optional[Tensor] = None, out: Optional[Tensor] = None) -> Tensor:
    """:func:`rows_linear` for the fused decode path: fp32 input rows (and fp32 residual) against ``weight`` in the
    compute dtype; ``zero``: an fp32 buffer the launch clears (the next layer's accumulation target)."""
    M, K = x.shape
    N = weight.shape[0]
    assert x.dtype == torch.float32 and x.stride(1) == 1 and weight.stride(1) == 1 and weight.shape[1] == K
    assert residual is None or residual.dtype == torch.float32
    assert zero is None or (zero.dtype == torch.float32 and zero.is_contiguous())
    y = out if out is not None else torch.empty((M, N), dtype=torch.float32 if out_f32 else weight.dtype, device=x.device)
    assert y.dtype == (torch.float32 if out_f32 else weight.dtype)
    b = None if bias is None else bias.detach().float()
    check(lib().vg_gemm_rows_mixed(ptr(x), x.stride(0), ptr(weight), weight.stride(0), ptr(b), ptr(residual),
                                   0 if residual is None else residual.stride(0), ptr(y), y.stride(0), M, N, K, int(act),
                                   int(out_f32), ptr(norm_scale), float(norm_eps), ptr(zero),
                                   0 if zero is None else zero.numel(), dtype_id(weight.dtype), stream()),
          "vg_gemm_rows_mixed")
    return y


def rows_linear_acc(x: Tensor, weight: Tensor, bias: Optional[Tensor], residual: Tensor, out: Tensor, *, splits: int = 4,
                    zero: Optional[Tensor] = None) -> Tensor:
    """``out`` (fp32, ZERO on entry) += x W^T + b + residual (vg_gemm_rows_acc): bf16 rows against bf16 weights, the
    reduction split over ``splits`` groups of blocks that meet in ``out`` through fp32 atomics; ``zero``: an fp32 buffer
    the launch clears (the accumulator of a later launch)."""
    M, K = x.shape
    N = weight.shape[0]
    assert x.dtype == torch.bfloat16 and weight.dtype == torch.bfloat16 and x.stride(1) == 1 and weight.stride(1) == 1
    assert residual.dtype == torch.float32 and out.dtype == torch.float32 and out.shape == (M, N) and out.stride(1) == 1
    assert zero is None or (zero.dtype == torch.float32 and zero.is_contiguous() and zero.data_ptr() != out.data_ptr())
    b = None if bias is None else bias.detach().float()
    check(lib().vg_gemm_rows_acc(ptr(x), x.stride(0), ptr(weight), weight.stride(0), ptr(b), ptr(residual), residual.stride(0),
                                 ptr(out), out.stride(0), M, N, K, int(splits), ptr(zero), 0 if zero is None else zero.numel(),
                                 stream()), "vg_gemm_rows_acc")
    return out


def decode_noise(seed: int, pos: Tensor, n_normal: int, epoch: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """(normal [B, n_normal], uniform [B]) for the frame at pos[b] of every sequence (vg_decode_noise): a function of
    (seed, epoch[0], b, pos[b]) only, so hipGraph replays draw fresh numbers as the device-side counter advances;
    ``epoch`` (one int32 on the device) is bumped by the caller whenever ``pos`` is rewound."""
    B = pos.numel()
    normal = torch.empty((B, n_normal), dtype=torch.float32, device=pos.device)
    uniform = torch.empty((B,), dtype=torch.float32, device=pos.device)
    check(lib().vg_decode_noise(int(seed) & 0xFFFFFFFFFFFFFFFF, ptr(pos), ptr(epoch) if epoch is not None else None,
                                ptr(normal), n_normal, ptr(uniform), B, stream()), "vg_decode_noise")
    return normal, uniform


def attention_layer_decode(x: Tensor, norm_scale: Tensor, norm_eps: float, wqkv: Tensor, bqkv: Optional[Tensor], wo: Tensor,
                           bo: Optional[Tensor], kcache: Tensor, vcache: Tensor, slopes: Tensor, pos: Tensor, H: int,
                           x1: Tensor, zero: Optional[Tensor] = None) -> Tensor:
    """x1 += attention sub-layer of one new frame (vg_attn_layer_decode); x, x1 fp32 [B, 64 H], x1 zero on entry."""
    B, D = x.shape
    assert x.dtype == torch.float32 and x1.dtype == torch.float32 and x.is_contiguous() and x1.is_contiguous()
    assert wqkv.is_contiguous() and wo.is_contiguous() and wqkv.shape == (3 * D, D) and wo.shape == (D, D) and D == 64 * H
    assert zero is None or (zero.dtype == torch.float32 and zero.is_contiguous() and zero.shape == x.shape)
    bq = None if bqkv is None else bqkv.detach().float()
    b_o = None if bo is None else bo.detach().float()
    check(lib().vg_attn_layer_decode(ptr(x), ptr(norm_scale), float(norm_eps), ptr(wqkv), ptr(bq), ptr(wo), ptr(b_o),
                                     ptr(kcache), ptr(vcache), ptr(slopes), ptr(pos), ptr(x1), ptr(zero), B,
                                     kcache.shape[1], H, dtype_id(wqkv.dtype), stream()), "vg_attn_layer_decode")
    return x1


def embed_fuse(frame: Tensor, emb: Tensor, wf: Tensor, bf: Optional[Tensor], dtype: torch.dtype) -> Tensor:
    """frame [B, 1 + latent] fp32 (token id, z) -> E[id] + relu(Wf z + bf) as [B, E] in ``dtype``."""
    B, E = frame.shape[0], emb.shape[1]
    assert frame.dtype == torch.float32 and frame.stride(-1) == 1 and emb.is_contiguous() and wf.is_contiguous()
    out = torch.empty((B, E), dtype=dtype, device=frame.device)
    check(lib().vg_embed_fuse(ptr(frame), frame.stride(0), ptr(emb), emb.shape[0], E, ptr(wf), ptr(bf), wf.shape[1],
                              ptr(out), B, dtype_id(dtype), stream()), "vg_embed_fuse")
    return out


def sample_token(logits: Tensor, temperature: float, uniform: Tensor, frame: Tensor, pos: Optional[Tensor]) -> None:
    """frame[b, 0] <- categorical draw from softmax(logits[b] / temperature) (inverse CDF with uniform[b]);
    pos[b] += 1 when ``pos`` is given."""
    B, V = logits.shape
    assert logits.dtype == torch.float32 and logits.is_contiguous() and frame.dtype == torch.float32
    check(lib().vg_sample_token(ptr(logits), V, float(temperature), ptr(uniform), ptr(frame), frame.stride(0), ptr(pos),
                                B, stream()), "vg_sample_token")


def attention_decode_append(qkv: Tensor, kcache: Tensor, vcache: Tensor, slopes: Tensor, pos: Tensor, H: int) -> Tensor:
    """qkv [B, 3*H*64] of the new frame; caches [B, Tmax, H*64] (updated in place at pos[b])."""
    B = qkv.shape[0]
    Tmax = kcache.shape[1]
    out = torch.empty((B, H * 64), dtype=qkv.dtype, device=qkv.device)
    check(lib().vg_attn_decode_append(ptr(qkv), ptr(kcache), ptr(vcache), ptr(out), ptr(slopes), ptr(pos), B, Tmax, H,
                                      dtype_id(qkv.dtype), stream()), "vg_attn_decode_append")
    return out


def advance(pos: Tensor, by: int = 1) -> None:
    check(lib().vg_advance(ptr(pos), pos.numel(), int(by), stream()), "vg_advance")
