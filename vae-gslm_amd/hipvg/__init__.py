"""ctypes binding of ``libvaegslm_hip.so`` (C ABI: include/vaegslm_hip.h).

PyTorch is used only as plumbing here: it owns device memory and streams;
every compute call goes through the C ABI with raw device pointers.  There is
NO CPU fallback: loading fails loudly when the library is missing, and every
wrapper refuses non-GPU tensors.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# VG_LIB: load another build of the same C ABI (A/B runs of two kernel versions in one session)
LIB_PATH = os.environ.get("VG_LIB") or os.path.join(os.path.dirname(_HERE), "csrc", "libvaegslm_hip.so")

VG_F32, VG_BF16 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU, ACT_SILU = 0, 1, 2, 3
ACT_STORED, ACT_SAVE_DERIV = 4, 16     # dact: multiply by a stored derivative; act flag: store act'(pre) in aux_out
ACT_DERIV_U8 = 32                      # flag (act with GELU | SAVE_DERIV, dact with STORED): the stored derivative is uint8 codes
DERIV_U8_STEP, DERIV_U8_LO = 0.005, -0.13   # value = code * step + lo (VG_DERIV_U8_STEP / _LO of include/vaegslm_hip.h)
ACT_IDS = {None: ACT_NONE, "none": ACT_NONE, "relu": ACT_RELU, "gelu": ACT_GELU, "silu": ACT_SILU}

_vp, _i, _i64, _f = C.c_void_p, C.c_int, C.c_int64, C.c_float


class GemmDesc(C.Structure):
    _fields_ = [("A", _vp), ("B", _vp), ("C", _vp),
                ("M", _i), ("N", _i), ("K", _i),
                ("lda", _i64), ("ldb", _i64), ("ldc", _i64),
                ("a_tr", _i), ("b_tr", _i), ("dtype", _i),
                ("bias", _vp), ("residual", _vp), ("aux_in", _vp), ("aux_out", _vp),
                ("lengths", _vp), ("T", _i),
                ("act", _i), ("dact", _i), ("out_f32", _i), ("accumulate", _i),
                ("split_k", _i), ("alpha", _f), ("pre_add", _vp), ("tile_cfg", _i), ("colsum_out", _vp), ("colpart", _vp),
                ("split_ws", _vp), ("split_cnt", _vp), ("split_ws_floats", _i64)]


class ColsumTask(C.Structure):
    _fields_ = [("src", _vp), ("rows", _i), ("cols", _i), ("ld", _i64), ("dst", _vp), ("accumulate", _i)]


COLSUM_MAX_TASKS = 32


class MeanTask(C.Structure):
    _fields_ = [("src", C.c_void_p), ("ld", C.c_int64), ("cols", C.c_int32), ("absolute", C.c_int32)]


MEAN_MAX_TASKS = 8

# name -> argtypes (restype is always int); must list EVERY symbol of the header
SIGNATURES = {
    "vg_version": [],
    "vg_last_error": [C.c_char_p, _i],
    "vg_gemm": [C.POINTER(GemmDesc), _vp],
    "vg_gemm_tile_rows": [C.POINTER(GemmDesc)],
    "vg_gemm_colpart_rows": [C.POINTER(GemmDesc)],
    "vg_gemm_grouped": [C.POINTER(GemmDesc), _i, _vp],
    "vg_rmsnorm_fwd": [_vp, _vp, _vp, _vp, _i, _i, _f, _vp, _i, _i, _vp],
    "vg_rmsnorm_bwd_blocks": [_i],
    "vg_rmsnorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp],
    "vg_rmsnorm_bwd_colsum": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _vp],
    "vg_attn_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp],
    "vg_attn_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _vp],
    "vg_attn_fwd_varlen": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _vp],
    "vg_attn_bwd_varlen": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _i, _vp],
    "vg_attn_stats_floats": [_i, _i, _i],
    "vg_attn_fwd_stats": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp],
    "vg_attn_bwd_stats": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp],
    "vg_gather_rows": [_vp, _vp, _vp, _i, _i, _vp],
    "vg_attn_decode": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "vg_ce_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _vp, _i, _i, _vp],
    "vg_ce_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i64, _vp, _i, _i, _vp],
    "vg_reparam_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _i, _vp],
    "vg_reparam_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _i, _vp],
    "vg_prior_logp_fwd": [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp],
    "vg_prior_logp_bwd": [_vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp],
    "vg_sum_f32": [_vp, _i64, _vp, _vp],
    "vg_colsum_blocks": [_i],
    "vg_colsum": [_vp, _i, _i, _i64, _vp, _vp, _i, _i, _vp],
    "vg_colsum_multi": [C.POINTER(ColsumTask), _i, _vp],
    "vg_masked_means": [C.POINTER(MeanTask), _i, _i, _vp, _i, _vp, _vp, _vp],
    "vg_masked_means_blocks": [_i],
    "vg_colsum_partials_multi": [C.POINTER(ColsumTask), _i, _i, _i, _vp],
    "vg_colsum_segments": [_vp, _i, _i, _i, _i64, _vp, _i, _vp, _i, _vp],
    "vg_colsum_segments_cu": [_vp, _vp, _i, _i, _i64, _vp, _i, _vp, _i, _vp],
    "vg_act_bwd": [_vp, _vp, _vp, _i64, _i, _i, _vp],
    "vg_cast_f32_to_bf16": [_vp, _vp, _i64, _vp],
    "vg_mask_rows": [_vp, _vp, _i, _i, _vp, _i, _i, _vp],
    "vg_adamw": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, C.POINTER(C.c_float), C.POINTER(C.c_float), _i, _f, _f, _f, _i,
                 _vp, _i, _vp],
    "vg_gemm_rows": [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _vp, _f, _i, _vp],
    "vg_embed_fuse": [_vp, _i, _vp, _i, _i, _vp, _vp, _i, _vp, _i, _i, _vp],
    "vg_sample_token": [_vp, _i, _f, _vp, _vp, _i, _vp, _i, _vp],
    "vg_attn_decode_append": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "vg_decode_noise": [C.c_uint64, _vp, _vp, _vp, _i, _vp, _i, _vp],
    "vg_attn_layer_decode": [_vp, _vp, _f, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "vg_gemm_rows_mixed": [_vp, _i64, _vp, _i64, _vp, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _i, _vp, _f, _vp, _i, _i, _vp],
    "vg_gemm_rows_acc": [_vp, C.c_int64, _vp, C.c_int64, _vp, _vp, C.c_int64, _vp, C.c_int64, _i, _i, _i, _i, _vp, _i, _vp],
    "vg_advance": [_vp, _i, _i, _vp],
    "vg_touch": [_vp, C.c_int64, _i, _vp],
    "vg_flow_blocks": [_i],
    "vg_flow_fwd": [_vp, _vp, _i64, _vp, _i, _vp, _vp, _vp, _i, _f, _f, _f, _vp, _i, _vp],
    "vg_flow_reverse": [_vp, _vp, _i64, _vp, _i, _vp, _i64, _i, _f, _f, _f, _vp, _i64, _f, _vp],
    "vg_flow_bwd": [_vp, _vp, _i64, _vp, _i, _vp, _vp, _vp, _vp, _vp, _i, _f, _f, _f, _vp, _i, _vp],
    "vg_dwnorm_blocks": [_i],
    "vg_dwnorm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp],
    "vg_dwnorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp],
    "vg_dwnorm_fwd_seg": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _f, _i, _vp],
    "vg_dwnorm_bwd_seg": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i,
                          _vp],
    "vg_dwnorm_fwd_cat": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _i, _vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _i, _f,
                          _i, _vp],
    "vg_dwnorm_bwd_ld": [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _i,
                         _i, _vp],
    "vg_dwnorm_bwd_fused_blocks": [_i, _i],
    "vg_dwnorm_bwd_fused": [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, _i,
                            _i, _i, _i, _vp],
    "vg_embed_fuse_blocks": [_i],
    "vg_embed_fuse_fwd": [_vp, _vp, _i64, _vp, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _vp],
    "vg_embed_fuse_bwd": [_vp, _vp, _vp, _i64, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _vp, _i64, _vp, _i, _vp],
    "vg_qsample": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _i, _i, _vp],
    "vg_l1_rows_fwd": [_vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp],
    "vg_l1_rows_bwd": [_vp, _vp, _vp, _vp, _i, _vp, _i, _i, _i, _vp],
    "vg_chnorm_blocks": [_i],
    "vg_chnorm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _i, _i, _vp],
    "vg_chnorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "vg_conv_gather": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "vg_conv_scatter": [_vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _vp],
    "vg_comm_unique_id": [_vp, _i],
    "vg_comm_init": [_i, _i, _vp, _i],
    "vg_comm_world": [],
    "vg_allreduce_bucket": [_vp, _i64, _i, _i, _vp],
    "vg_comm_destroy": [],
    "vg_prof_enable": [_i],
    "vg_prof_read": [_i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i)],
    "vg_prof_read_bytes": [_i, C.POINTER(C.c_double)],
    "vg_prof_tag": [_i],
    "vg_prof_read_tag": [_i, _i, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(_i)],
    "vg_probe_mfma": [_vp, _i, _i, _vp],
    "vg_probe_copy": [_vp, _vp, _i64, _i, _vp],
}

PROF_KINDS = {"gemm_bf16_nt": 0, "gemm_bf16_nn": 1, "gemm_bf16_tn": 2, "gemm_f32": 3,
              "attn_fwd": 4, "attn_bwd": 5, "rmsnorm_fwd": 6, "rmsnorm_bwd": 7, "adamw": 8,
              "dwnorm_fwd": 9, "dwnorm_bwd": 10}


def probe_peaks(device=None) -> dict:
    """Measured peaks of the GPU this process runs on: dense bf16 MFMA (register-fed independent chains) and a
    streaming HBM copy.  A few hundred milliseconds; used by bench.py for ``roofline.peak_measured``."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else device
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    blocks, iters = cus * 8, 4000
    out = torch.empty(blocks * 256, dtype=torch.float32, device=dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    check(lib().vg_probe_mfma(ptr(out), blocks, 100, stream()), "vg_probe_mfma")
    torch.cuda.synchronize(dev)
    a.record()
    check(lib().vg_probe_mfma(ptr(out), blocks, iters, stream()), "vg_probe_mfma")
    b.record()
    torch.cuda.synchronize(dev)
    mfma = blocks * 4 * iters * 4 * 32768.0 / (a.elapsed_time(b) * 1e-3) / 1e12
    nbytes = 1 << 30
    src = torch.ones(nbytes // 4, dtype=torch.float32, device=dev)
    dst = torch.empty_like(src)
    check(lib().vg_probe_copy(ptr(src), ptr(dst), nbytes, cus * 16, stream()), "vg_probe_copy")
    torch.cuda.synchronize(dev)
    a.record()
    for _ in range(4):
        check(lib().vg_probe_copy(ptr(src), ptr(dst), nbytes, cus * 16, stream()), "vg_probe_copy")
    b.record()
    torch.cuda.synchronize(dev)
    copy = 4 * 2.0 * nbytes / (a.elapsed_time(b) * 1e-3) / 1e12
    return {"mfma_bf16_dense_tflops": mfma, "hbm_copy_tb_per_s": copy, "compute_units": cus}


def prof_enable(on: bool) -> None:
    lib().vg_prof_enable(int(on))


def prof_read(kind: str):
    ms, work, n = C.c_double(), C.c_double(), _i()
    lib().vg_prof_read(PROF_KINDS[kind], C.byref(ms), C.byref(work), C.byref(n))
    return ms.value, work.value, n.value


PROF_TAG_NONE, PROF_TAG_LAYER = 0, 1


def prof_tag(tag: int) -> int:
    """Scope tag of the launches recorded from now on; returns the previous one."""
    return lib().vg_prof_tag(int(tag))


def prof_read_tag(kind: str, tag: int):
    ms, work, n = C.c_double(), C.c_double(), _i()
    lib().vg_prof_read_tag(PROF_KINDS[kind], int(tag), C.byref(ms), C.byref(work), C.byref(n))
    return ms.value, work.value, n.value


def prof_read_bytes(kind: str) -> float:
    """Summed algorithmic bytes (operands and results once each) of the recorded launches of one GEMM kind."""
    b = C.c_double()
    lib().vg_prof_read_bytes(PROF_KINDS[kind], C.byref(b))
    return b.value

_lib = None
_lock = threading.Lock()


def lib() -> C.CDLL:
    """Load the HIP library (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        with _lock:
            if _lib is None:
                if not os.path.exists(LIB_PATH):
                    raise RuntimeError(
                        f"{LIB_PATH} is missing: the MI355X HIP hot path has no fallback. "
                        "Build it with `python vae-gslm_amd/hipvg/build.py` "
                        "(or `__graft_entry__.build()`).")
                handle = C.CDLL(LIB_PATH)
                for name, argtypes in SIGNATURES.items():
                    fn = getattr(handle, name)     # AttributeError if the ABI lost a symbol
                    fn.argtypes = argtypes
                    fn.restype = C.c_int
                if os.environ.get("VG_ROCTX", "0") == "1":
                    handle = _with_roctx_ranges(handle)
                _lib = handle
    return _lib


class _RoctxLib:
    """The library with every launch entry point bracketed by a roctx range named after the symbol, so that
    `rocprofv3 --marker-trace` shows which C-ABI call a kernel belongs to (VG_ROCTX=1; tracing only)."""

    def __init__(self, handle, roctx):
        self._h, self._push, self._pop = handle, roctx.roctxRangePushA, roctx.roctxRangePop
        self._push.argtypes, self._push.restype = [C.c_char_p], C.c_int
        self._pop.argtypes, self._pop.restype = [], C.c_int
        self._cache = {}

    def __getattr__(self, name):
        fn = self._cache.get(name)
        if fn is None:
            raw = getattr(self._h, name)
            if name.endswith("_blocks") or name in ("vg_version", "vg_last_error", "vg_comm_world", "vg_gemm_tile_rows", "vg_gemm_colpart_rows"):
                fn = raw
            else:
                tag, push, pop = name.encode(), self._push, self._pop

                def fn(*args, _raw=raw, _tag=tag):
                    push(_tag)
                    try:
                        return _raw(*args)
                    finally:
                        pop()
            self._cache[name] = fn
        return fn


def _with_roctx_ranges(handle):
    for cand in ("libroctx64.so", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so"):
        try:
            return _RoctxLib(handle, C.CDLL(cand))
        except OSError:
            continue
    raise RuntimeError("VG_ROCTX=1 but libroctx64.so was not found")


def last_error() -> str:
    buf = C.create_string_buffer(512)
    lib().vg_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed (code {rc}): {last_error()}")


def dtype_id(t: torch.dtype) -> int:
    if t == torch.float32:
        return VG_F32
    if t == torch.bfloat16:
        return VG_BF16
    raise TypeError(f"HIP hot path supports float32 / bfloat16 activations, got {t}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("vae-gslm_amd HIP ops need tensors on an MI355X device "
                           "(there is no CPU fallback); got a CPU tensor")
    return t.data_ptr()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


# ---------------------------------------------------------------- precision policy
class _Policy(threading.local):
    def __init__(self):
        self.dtype = torch.bfloat16


_policy = _Policy()


def set_precision(p) -> None:
    """'bf16' (fast path) or 'fp32' (exact-f32 MFMA parity path)."""
    if isinstance(p, str):
        p = {"bf16": torch.bfloat16, "bfloat16": torch.bfloat16,
             "fp32": torch.float32, "float32": torch.float32}[p]
    assert p in (torch.bfloat16, torch.float32)
    _policy.dtype = p


def compute_dtype() -> torch.dtype:
    return _policy.dtype
