"""Build ``csrc/libvaegslm_hip.so`` for gfx950 with hipcc (cross-compiles
without a GPU).  In-tree output so the library travels with the source
snapshot; no torch / pybind dependency -- the ABI is plain C (include/vaegslm_hip.h).
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
SOURCES = ("vg_rows.hip", "vg_gemm.hip", "vg_gemm_dma.hip", "vg_gemm_ph.hip", "vg_attention.hip", "vg_conv.hip", "vg_flow.hip", "vg_optim.hip", "vg_decode.hip", "vg_prof.hip", "vg_comm.hip")
LIB = os.path.join(CSRC, "libvaegslm_hip.so")
ARCH = "gfx950"


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    deps.append(os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "vaegslm_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not _stale():
        return LIB
    # one builder at a time: the ranks of a multi-GPU launch import the package together, and a snapshot copy may
    # not keep the library newer than its sources
    import fcntl
    with open(os.path.join(CSRC, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():
                return LIB
            return _build_locked(verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(verbose: bool) -> str:
    cc = _hipcc()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]

    def compile_one(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [cc, f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-c",
               os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-4000:]}")
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, srcs))
    tmp = LIB + f".tmp{os.getpid()}"          # link beside the target, then rename: a reader never sees a partial file
    cmd = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", tmp] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stderr[-4000:]}")
    os.replace(tmp, LIB)
    if verbose:
        print(f"[hipvg] built {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
