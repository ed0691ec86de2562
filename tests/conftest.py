import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built library (artefacts are git-ignored): build it once (hipcc cross-compiles without a GPU) so the ABI tests
    and everything behind pulse_amd._lib can load it.  A present library is left alone -- the GPU box runs the one that travelled with the tree."""
    from pulse_amd import _lib
    if os.path.exists(_lib.LIB_PATH) or os.environ.get("PULSE_HIP_LIB"):
        return
    try:
        from pulse_amd.csrc import build
        build.build()
    except Exception as exc:                                     # the tests that need the library will say so themselves
        print(f"[conftest] could not build libpulse_hip.so: {exc}")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


class Golden:
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name))

    def np(self, k):
        return self.z[k]

    def t(self, k, device="cpu"):
        return torch.from_numpy(self.z[k]).to(device)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]
    return get


@pytest.fixture(scope="session")
def dev():
    return torch.device("cuda:0")
