"""RCCL gradient exchange through the C ABI (``vg_comm_*`` / ``vg_allreduce_bucket``, include/vaegslm_hip.h).

The communicator id is drawn on rank 0 and shipped to the other ranks over whatever ``torch.distributed`` group
already exists (any backend: it is 128 bytes, once).  After that the data path does not touch ``torch.distributed``:
``all_reduce_`` is one stream-ordered RCCL launch on the caller's stream.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import VG_BF16, VG_F32, check, lib, ptr

ID_BYTES = 128


def world() -> int:
    """Ranks of the live communicator (0: not initialised)."""
    return int(lib().vg_comm_world())


def init(rank: int, world_size: int, group=None) -> None:
    """Create the process's communicator on the current device.  With ``world_size > 1`` a ``torch.distributed``
    group (default group if ``None``) carries the id from rank 0 to the others."""
    if world() == world_size:
        return
    buf = (C.c_uint8 * ID_BYTES)()
    if rank == 0:
        check(lib().vg_comm_unique_id(C.cast(buf, C.c_void_p), ID_BYTES), "vg_comm_unique_id")
    if world_size > 1:
        import torch.distributed as dist
        t = torch.tensor(list(buf), dtype=torch.uint8)
        dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else t.device
        t = t.to(dev)
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        buf = (C.c_uint8 * ID_BYTES)(*t.cpu().tolist())
    check(lib().vg_comm_init(int(rank), int(world_size), C.cast(buf, C.c_void_p), ID_BYTES), "vg_comm_init")


def all_reduce_(flat: torch.Tensor, average: bool = True, stream=None) -> None:
    """In-place sum (or mean) of a contiguous fp32 / bf16 device tensor over the ranks, on ``stream`` (default: the
    current stream).  Returns as soon as the collective is enqueued."""
    if not (flat.is_cuda and flat.is_contiguous()):
        raise RuntimeError("hipvg.comm.all_reduce_: contiguous device tensor required")
    dt = {torch.float32: VG_F32, torch.bfloat16: VG_BF16}[flat.dtype]
    st = (stream or torch.cuda.current_stream(flat.device)).cuda_stream
    check(lib().vg_allreduce_bucket(ptr(flat), flat.numel(), dt, 1 if average else 0, st), "vg_allreduce_bucket")


def destroy() -> None:
    check(lib().vg_comm_destroy(), "vg_comm_destroy")


def masked_stream(device, n_cus: int, total_cus: int = 256):
    """A HIP stream whose kernels may only run on ``n_cus`` of the GPU's compute units (hipExtStreamCreateWithCUMask), spread
    evenly over the XCDs, as a ``torch.cuda.ExternalStream``.  Lab / tuning knob for the communication stream
    (training_lib/dp.py, ``VG_COMM_CU_MASK``): collective kernels confined to a few CUs cannot take a GEMM tile's CU in the
    middle of a round.  The stream is never destroyed (one per process)."""
    import ctypes
    n_cus = max(1, min(int(n_cus), total_cus))
    words = (total_cus + 31) // 32
    mask = (ctypes.c_uint32 * words)()
    step = total_cus / n_cus
    for i in range(n_cus):
        cu = int(i * step)
        mask[cu // 32] |= 1 << (cu % 32)
    hip = ctypes.CDLL("libamdhip64.so")
    st = ctypes.c_void_p()
    with torch.cuda.device(device):
        rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), ctypes.c_uint32(words), mask)
    if rc != 0 or not st.value:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask failed ({rc})")
    return torch.cuda.ExternalStream(st.value, device=device)
