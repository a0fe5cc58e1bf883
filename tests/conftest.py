import os
import sys

import numpy as np
import pytest
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "vae-gslm_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


# The GPU suite is one long-lived process that creates and destroys hundreds of graphs and streams: the stream population
# ROCm 7.0's hipGraphLaunch needs to walk off an exec's internal stream list (hipvg.functional.graph_launch_stream; the suite
# died of it at its 306th test).  Graphs are launched from a high-priority stream here; the product default stays an ordinary
# stream (tests/test_parity_round6_gpu.py::test_graph_launch_survives_an_uneven_stream_population covers the kinds).
os.environ.setdefault("VG_LAUNCH_STREAM", "prio")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def full_cfg():
    with open(os.path.join(PKG, "configs/train/speech/vae-gslm.yaml")) as f:
        return yaml.safe_load(f)


@pytest.fixture(scope="session")
def golden():
    def _load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    return _load


@pytest.fixture(autouse=True)
def _resource_trace(request):
    """VG_DEBUG_RESOURCES=1: native threads, open files and device memory after every test (stderr; run with -s)."""
    yield
    if os.environ.get("VG_DEBUG_RESOURCES", "0") != "1":
        return
    try:
        with open("/proc/self/status") as f:
            st = {l.split(":")[0]: l.split(":")[1].strip() for l in f if ":" in l}
        nfd = len(os.listdir("/proc/self/fd"))
        import gc
        import torch
        graphs = sum(1 for o in gc.get_objects() if type(o).__name__ == "CUDAGraph")
        mem = torch.cuda.memory_reserved() >> 20 if torch.cuda.is_available() else 0
        print(f"[vg_res] {request.node.nodeid[-70:]:70s} threads={st.get('Threads')} fds={nfd} rss={st.get('VmRSS')} graphs={graphs} reserved={mem}MiB",
              file=sys.stderr, flush=True)
    except Exception as exc:      # noqa: BLE001
        print(f"[vg_res] {exc!r}", file=sys.stderr)
