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
