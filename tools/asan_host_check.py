#!/usr/bin/env python3
"""Host-side AddressSanitizer + UBSan pass over the C-ABI library (GPU sanitizers are not available on this pool: the
device code is built as usual, `-fno-gpu-sanitize`).  Builds every source of vae-gslm_amd/csrc with
`-fsanitize=address,undefined` into /tmp, loads it in a child process under the ASan runtime and drives every entry
point that runs host code without a device: tile-configuration choice over a sweep of GEMM shapes, block-count helpers,
argument validation of the launchers (bad shapes / dtypes / alignment: an error code and a message, never a launch),
communicator error paths.  Exit code 0 and no sanitizer report = clean.      python tools/asan_host_check.py [--keep]"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vae-gslm_amd", "csrc")
OUT = os.environ.get("VG_ASAN_DIR", "/tmp/vg_asan")
HIPCC = "/opt/rocm/bin/hipcc"
FLAGS = ["--offload-arch=gfx950", "-O1", "-g", "-std=c++17", "-fPIC", "-fsanitize=address,undefined", "-fno-gpu-sanitize",
         "-fno-omit-frame-pointer", "-w"]

DRIVER = r'''
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(%(root)r, "vae-gslm_amd"))
os.environ["VG_LIB"] = %(lib)r
import hipvg
L = hipvg.lib()
assert L.vg_version() >= 100
n = 0
# tile choice / tile rows over a sweep of descriptors (pure host code: the cost model of pick_cfg)
for M in (1, 7, 200, 512, 1024, 5120, 8000, 10240, 12288, 16000, 32000):
    for N in (8, 32, 64, 80, 200, 512, 1024, 2048, 3072, 4096):
        for K in (8, 32, 64, 80, 512, 1024, 4096, 16000):
            for a_tr, b_tr in ((0, 0), (0, 1), (1, 1)):
                for split in (1, 4, 12):
                    for cfg in (0, 1, 3, 9, 13, 14, 15):
                        d = hipvg.GemmDesc()
                        d.M, d.N, d.K = M, N, K
                        d.lda = M if a_tr else K
                        d.ldb = N if b_tr else K
                        d.ldc = N
                        d.a_tr, d.b_tr, d.dtype, d.split_k, d.tile_cfg = a_tr, b_tr, 1, split, cfg
                        L.vg_gemm_tile_rows(C.byref(d))
                        n += 1
for m in (0, 1, 100, 8000, 16000, 1 << 20):
    L.vg_rmsnorm_bwd_blocks(m); L.vg_colsum_blocks(m); L.vg_dwnorm_blocks(m); L.vg_chnorm_blocks(m); L.vg_embed_fuse_blocks(m)
# argument validation: every call below must come back with an error code and a message, without touching a device
bad = 0
def expect_error(rc):
    global bad
    assert rc != 0, "a malformed call was accepted"
    assert hipvg.last_error() != ""
    bad += 1
d = hipvg.GemmDesc(); d.M = d.N = d.K = 0
expect_error(L.vg_gemm(C.byref(d), None))
d.M, d.N, d.K, d.dtype = 16, 16, 16, 7
expect_error(L.vg_gemm(C.byref(d), None))
expect_error(L.vg_gemm_rows(None, 8, None, 8, None, None, 0, None, 8, 0, 8, 8, 0, 0, None, 0.0, 1, None))
expect_error(L.vg_gemm_rows(None, 8, None, 8, None, None, 0, None, 8, 65, 8, 8, 0, 0, None, 0.0, 1, None))
expect_error(L.vg_gemm_rows_mixed(None, 8, None, 8, None, None, 0, None, 8, 4, 8, 12, 0, 0, None, 0.0, None, 0, 1, None))
expect_error(L.vg_attn_decode_append(None, None, None, None, None, None, 0, 0, 0, 1, None))
expect_error(L.vg_attn_layer_decode(None, None, 0.0, None, None, None, None, None, None, None, None, None, None, 2, 16, 3, 1, None))
expect_error(L.vg_decode_noise(1, None, None, 4, None, 2, None))
expect_error(L.vg_sample_token(None, 0, 1.0, None, None, 0, None, 1, None))
expect_error(L.vg_embed_fuse(None, 0, None, 0, 0, None, None, 0, None, 0, 1, None))
expect_error(L.vg_dwnorm_fwd(None, None, None, None, None, None, None, None, None, 10, 512, 3, 7, 3, 1e-5, 1, None))
expect_error(L.vg_allreduce_bucket(None, 16, 0, 1, None))
print("asan host check: %%d descriptors, %%d rejected calls, no report" %% (n, bad))
'''


def main():
    os.makedirs(OUT, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))

    def compile_one(src):
        obj = os.path.join(OUT, src.replace(".hip", ".o"))
        r = subprocess.run([HIPCC, *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise SystemExit(f"sanitized build of {src} failed:\n{r.stderr[-3000:]}")
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, srcs))
    lib = os.path.join(OUT, "libvaegslm_hip_asan.so")
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address,undefined", "-fno-gpu-sanitize",
                        "-o", lib, *objs], capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit("link failed:\n" + r.stderr[-3000:])
    rt = subprocess.run([HIPCC, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.exists(rt):
        cand = subprocess.run("ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so", shell=True,
                              capture_output=True, text=True).stdout.split()
        rt = cand[0] if cand else rt
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", DRIVER % {"root": ROOT, "lib": lib}], env=env, capture_output=True, text=True)
    sys.stdout.write(r.stdout)
    report = "ERROR: AddressSanitizer" in r.stderr or "runtime error:" in r.stderr
    if r.returncode != 0 or report:
        sys.stderr.write(r.stderr[-6000:])
        raise SystemExit(f"sanitizer check FAILED (rc {r.returncode})")
    if "--keep" not in sys.argv:
        for f in os.listdir(OUT):
            os.remove(os.path.join(OUT, f))


if __name__ == "__main__":
    main()
