"""The C-ABI library loads and exports every symbol include/vaegslm_hip.h
declares; the ctypes table in hipvg covers exactly that set.  No compute
calls (runs without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "vaegslm_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\bint\s+(vg_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built_lib():
    from hipvg.build import build
    return build(force=False, verbose=False)


def test_header_symbols_exported(built_lib):
    handle = ctypes.CDLL(built_lib)
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(handle, n), f"{n} declared in the header but not exported"


def test_ctypes_table_matches_header(built_lib):
    import hipvg
    assert sorted(hipvg.SIGNATURES) == declared_symbols()
    lib = hipvg.lib()
    assert lib.vg_version() >= 100
    assert lib.vg_rmsnorm_bwd_blocks(8000) == 512
    assert lib.vg_colsum_blocks(100) == 1 and lib.vg_colsum_blocks(8000) == 128
    assert hipvg.last_error() == ""


def test_gemm_descriptor_layout_matches_header():
    """Field order/types of the ctypes mirror == the C struct (parsed from the header)."""
    import hipvg
    text = open(HEADER).read()
    body = re.search(r"typedef struct vg_gemm_desc \{(.*?)\} vg_gemm_desc;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = [n.strip().lstrip("*") for n in decl.split(",")]
        names[0] = names[0].split()[-1].lstrip("*")
        fields += names
    assert fields == [f[0] for f in hipvg.GemmDesc._fields_]


def test_product_refuses_cpu_tensors(built_lib):
    """No CPU fallback: HIP wrappers raise on host tensors instead of computing."""
    import torch
    from hipvg import functional as HF
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        HF.rmsnorm(torch.randn(4, 256), torch.ones(256), 1e-6)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import hipvg
    monkeypatch.setattr(hipvg, "_lib", None)
    monkeypatch.setattr(hipvg, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no fallback"):
        hipvg.lib()


def test_comm_entry_points_fail_cleanly_without_a_communicator(built_lib):
    """vg_allreduce_bucket before vg_comm_init is an error code + message, not a crash (no GPU needed: the
    check precedes any use of the pointer)."""
    import hipvg
    lib = hipvg.lib()
    assert lib.vg_comm_world() == 0
    assert lib.vg_allreduce_bucket(ctypes.c_void_p(4096), 256, hipvg.VG_F32, 1, None) != 0
    assert "vg_comm_init" in hipvg.last_error()
    assert lib.vg_comm_init(3, 2, ctypes.c_void_p(4096), 128) != 0          # rank outside the world
    assert lib.vg_comm_destroy() == 0                                         # nothing to destroy is fine


def test_colsum_task_layout_matches_header():
    """Field order of the ctypes mirror of vg_colsum_task == the C struct (parsed from the header), and the task
    limit agrees."""
    import hipvg
    text = open(HEADER).read()
    body = re.search(r"typedef struct vg_colsum_task \{(.*?)\} vg_colsum_task;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = [n.strip().lstrip("*") for n in decl.split(",")]
        names[0] = names[0].split()[-1].lstrip("*")
        fields += names
    assert fields == [f[0] for f in hipvg.ColsumTask._fields_]
    assert int(re.search(r"VG_COLSUM_MAX_TASKS\s*=\s*(\d+)", text).group(1)) == hipvg.COLSUM_MAX_TASKS
    assert ctypes.sizeof(hipvg.ColsumTask) == 40


def test_roctx_ranges_wrap_the_entry_points(monkeypatch):
    """VG_ROCTX=1: launch entry points are bracketed by roctx ranges (tracing aid); query functions stay direct.
    Checked in a fresh interpreter so the cached library handle of this process is left alone."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r); import hipvg; L = hipvg.lib(); "
            "assert type(L).__name__ == '_RoctxLib'; assert L.vg_version() >= 100; "
            "assert L.vg_comm_destroy() == 0; assert L.vg_rmsnorm_bwd_blocks(8000) == 512; print('ok')"
            % os.path.join(ROOT, "vae-gslm_amd"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, VG_ROCTX="1"), capture_output=True, text=True)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.skipif(os.environ.get("VG_RUN_ASAN", "0") != "1",
                    reason="host ASan + UBSan build of the library (about a minute of hipcc): VG_RUN_ASAN=1 python -m pytest tests/test_cabi.py")
def test_host_code_is_clean_under_asan_and_ubsan():
    """tools/asan_host_check.py: every source built with -fsanitize=address,undefined for the HOST side (-fno-gpu-sanitize;
    GPU sanitizers do not exist on this pool), then the tile-choice cost model over 55 k descriptors, the block-count
    helpers and the argument validation of the launchers run under the sanitizer runtime without a report."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asan_host_check.py")], capture_output=True, text=True)
    assert r.returncode == 0 and "no report" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
