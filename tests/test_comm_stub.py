"""The library's RCCL binding (csrc/vg_comm.hip, hipvg/comm.py) against a test double of librccl (tests/stubs/
fake_rccl.c, selected through VG_RCCL_LIB): order and arguments of the RCCL calls, error propagation, teardown and
re-initialisation, and the two-rank id hand-off of ``hipvg.comm.init`` over a gloo group -- everything about
``hip.comm=abi`` that can be checked without two GPUs.  CPU only."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vae-gslm_amd")


@pytest.fixture(scope="module")
def fake_rccl(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("fake_rccl") / "libfake_rccl.so")
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-o", out, os.path.join(ROOT, "tests", "stubs", "fake_rccl.c"), "-ldl"],
                   check=True)
    from hipvg.build import build
    build()
    return out


def run(code, env_extra, nproc=1, tmp=None):
    env = dict(os.environ, PYTHONPATH=PKG + os.pathsep + ROOT, **env_extra)
    if nproc == 1:
        return subprocess.run([sys.executable, "-c", textwrap.dedent(code)], env=env, capture_output=True, text=True, timeout=180)
    script = os.path.join(tmp, "two_rank.py")
    with open(script, "w") as f:
        f.write(textwrap.dedent(code))
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
                           "--master-addr", "127.0.0.1", "--master-port", "29731", script],
                          env=env, capture_output=True, text=True, timeout=300)


SINGLE = """
import ctypes, hipvg
L = hipvg.lib()
fake = ctypes.CDLL(__import__('os').environ['VG_RCCL_LIB'])
fake.fake_rccl_log.restype = ctypes.c_char_p
log = lambda: fake.fake_rccl_log().decode()
buf = (ctypes.c_uint8 * 128)()
assert L.vg_allreduce_bucket(ctypes.c_void_p(4096), 16, 0, 1, None) != 0 and 'vg_comm_init' in hipvg.last_error()
assert log() == '', log()                                  # nothing reached RCCL before a communicator exists
assert L.vg_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p), 128) == 0
assert bytes(buf)[:4] == bytes([3, 10, 17, 24])
assert L.vg_comm_init(1, 2, ctypes.cast(buf, ctypes.c_void_p), 128) == 0 and L.vg_comm_world() == 2
assert L.vg_comm_init(1, 2, ctypes.cast(buf, ctypes.c_void_p), 128) != 0 and 'already initialised' in hipvg.last_error()
x = (ctypes.c_float * 16)()
assert L.vg_allreduce_bucket(ctypes.cast(x, ctypes.c_void_p), 16, 0, 1, None) == 0      # fp32, mean, in place, live comm
assert L.vg_allreduce_bucket(ctypes.cast(x, ctypes.c_void_p), 16, 1, 0, None) == 0      # bf16, sum
assert L.vg_allreduce_bucket(ctypes.cast(x, ctypes.c_void_p), 16, 5, 0, None) != 0      # bad dtype: refused before RCCL
assert L.vg_comm_destroy() == 0 and L.vg_comm_world() == 0
assert L.vg_comm_destroy() == 0                                                           # idempotent
assert L.vg_allreduce_bucket(ctypes.cast(x, ctypes.c_void_p), 16, 0, 1, None) != 0      # after teardown: refused again
assert L.vg_comm_init(0, 1, ctypes.cast(buf, ctypes.c_void_p), 128) == 0 and L.vg_comm_world() == 1   # re-initialisation
assert L.vg_comm_destroy() == 0
assert log() == 'GetUniqueId;CommInitRank:1/2;AllReduce:7,4,I,L;AllReduce:9,0,I,L;CommDestroy;CommInitRank:0/1;CommDestroy;', log()
print('ok')
"""


def test_rccl_call_sequence_and_teardown(fake_rccl):
    r = run(SINGLE, {"VG_RCCL_LIB": fake_rccl})
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


FAIL = """
import ctypes, hipvg
L = hipvg.lib()
buf = (ctypes.c_uint8 * 128)()
assert L.vg_comm_unique_id(ctypes.cast(buf, ctypes.c_void_p), 128) == 0
assert L.vg_comm_init(0, 2, ctypes.cast(buf, ctypes.c_void_p), 128) == 0
x = (ctypes.c_float * 16)()
assert L.vg_allreduce_bucket(ctypes.cast(x, ctypes.c_void_p), 16, 0, 1, None) != 0
assert 'ncclAllReduce' in hipvg.last_error() and 'all-reduce refused' in hipvg.last_error(), hipvg.last_error()
assert L.vg_comm_world() == 2                 # a failed collective does not tear the communicator down
assert L.vg_comm_destroy() == 0
print('ok')
"""


def test_rccl_errors_carry_the_library_message(fake_rccl):
    r = run(FAIL, {"VG_RCCL_LIB": fake_rccl, "FAKE_RCCL_FAIL": "allreduce"})
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
    r = run("import ctypes, hipvg\nL = hipvg.lib()\nb = (ctypes.c_uint8 * 128)()\n"
            "assert L.vg_comm_init(0, 2, ctypes.cast(b, ctypes.c_void_p), 128) != 0\n"
            "assert 'ncclCommInitRank' in hipvg.last_error() and L.vg_comm_world() == 0\nprint('ok')\n",
            {"VG_RCCL_LIB": fake_rccl, "FAKE_RCCL_FAIL": "init"})
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


TWO_RANK = """
import ctypes, os
import torch.distributed as dist
import hipvg
from hipvg import comm
dist.init_process_group('gloo')
rank = dist.get_rank()
fake = ctypes.CDLL(os.environ['VG_RCCL_LIB'])
fake.fake_rccl_log.restype = ctypes.c_char_p
fake.fake_rccl_last_id.restype = ctypes.POINTER(ctypes.c_uint8)
comm.init(rank, 2)                      # rank 0 draws the id, the gloo group carries it, both ranks create their communicator
assert comm.world() == 2
log = fake.fake_rccl_log().decode()
assert log == ('GetUniqueId;' if rank == 0 else '') + f'CommInitRank:{rank}/2;', log
ident = bytes(fake.fake_rccl_last_id()[:128])
assert ident[:4] == bytes([3, 10, 17, 24]) and ident[127] == (127 * 7 + 3) % 256      # rank 1 received rank 0's id
comm.init(rank, 2)                      # same world: no second communicator
assert fake.fake_rccl_log().decode() == log
comm.destroy()
assert comm.world() == 0
dist.barrier()
dist.destroy_process_group()
# one write per rank into its own file: two ranks printing into one pipe interleave ("okok  01")
with open(os.path.join(os.environ['VG_TEST_OUT'], f'rank{rank}.ok'), 'w') as f:
    f.write(f'ok {rank}')
"""


def test_two_rank_id_handoff_over_gloo(fake_rccl, tmp_path):
    r = run(TWO_RANK, {"VG_RCCL_LIB": fake_rccl, "MASTER_ADDR": "127.0.0.1", "VG_TEST_OUT": str(tmp_path)}, nproc=2,
            tmp=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    for rank in (0, 1):
        assert (tmp_path / f"rank{rank}.ok").read_text() == f"ok {rank}", r.stdout[-2000:] + r.stderr[-3000:]


TWO_RANK_VALUES = """
import ctypes, os, struct
import torch.distributed as dist
import hipvg
from hipvg import comm
dist.init_process_group('gloo')
rank = dist.get_rank()
L = hipvg.lib()
comm.init(rank, 2)
n = 1000
# fp32, mean: rank r holds (r + 1) * (i - 300) / 7
x = (ctypes.c_float * n)(*[(rank + 1) * (i - 300) / 7.0 for i in range(n)])
assert L.vg_allreduce_bucket(ctypes.cast(x, ctypes.c_void_p), n, 0, 1, None) == 0, hipvg.last_error()
for i in (0, 1, 299, 300, 999):
    a, b = ctypes.c_float((i - 300) / 7.0).value, ctypes.c_float(2 * (i - 300) / 7.0).value
    want = ctypes.c_float(ctypes.c_float(a + b).value * 0.5).value
    assert x[i] == want, (i, x[i], want)
# fp32, sum, a second collective on the same communicator (sequence numbers advance together)
y = (ctypes.c_float * 8)(*[float(rank * 10 + i) for i in range(8)])
assert L.vg_allreduce_bucket(ctypes.cast(y, ctypes.c_void_p), 8, 0, 0, None) == 0
assert list(y) == [10.0 + 2 * i for i in range(8)], list(y)
# bf16, sum: small integers are exact in bf16
def bf(v):
    return struct.unpack('<I', struct.pack('<f', float(v)))[0] >> 16
z = (ctypes.c_uint16 * 16)(*[bf(rank + i) for i in range(16)])
assert L.vg_allreduce_bucket(ctypes.cast(z, ctypes.c_void_p), 16, 1, 0, None) == 0
assert list(z) == [bf(2 * i + 1) for i in range(16)], list(z)
comm.destroy()
dist.barrier()
dist.destroy_process_group()
with open(os.path.join(os.environ['VG_TEST_OUT'], f'rank{rank}.ok'), 'w') as f:
    f.write(f'ok {rank}')
"""


def test_two_rank_allreduce_values_through_the_abi(fake_rccl, tmp_path):
    """hip.comm=abi with two ranks, checked for VALUES: the test double reduces for real (FAKE_RCCL_DIR: the ranks
    exchange their buffers through files), so vg_allreduce_bucket's dtype / average arguments, the in-place contract
    and the order of successive collectives are exercised end to end on host buffers."""
    ex = tmp_path / "exchange"
    ex.mkdir()
    r = run(TWO_RANK_VALUES, {"VG_RCCL_LIB": fake_rccl, "MASTER_ADDR": "127.0.0.1", "VG_TEST_OUT": str(tmp_path),
                              "FAKE_RCCL_DIR": str(ex)}, nproc=2, tmp=str(tmp_path))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    for rank in (0, 1):
        assert (tmp_path / f"rank{rank}.ok").read_text() == f"ok {rank}", r.stdout[-2000:] + r.stderr[-3000:]
