import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "experiment-yolo_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import ultralytics.hip  # noqa: E402,F401  (sets DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 before any test initialises HIP)

GOLDEN = os.path.join(ROOT, "tests", "golden")
CFG_DIR = os.path.join(PKG, "ultralytics", "cfg", "models")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


class Golden:
    """Lazy view of one tests/golden/*.npz fixture with '/'-separated keys."""

    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)

    def __getitem__(self, k):
        return self.z[k]

    def t(self, k):
        import torch

        a = self.z[k]
        return torch.from_numpy(a.astype(np.float32) if a.dtype == np.float16 else a)

    def keys(self, prefix=""):
        return [k for k in self.z.files if k.startswith(prefix)]

    def __contains__(self, k):
        return k in self.z.files


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return get
