import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "particle-simulator_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


class Golden:
    """One committed fixture: inputs (meta) + outputs recorded from the reference."""

    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        self.meta = json.loads(bytes(self.z["meta_json"]).decode())

    def __getitem__(self, k):
        return self.z[k]

    def __contains__(self, k):
        return k in self.z.files

    @property
    def n(self):
        return self.meta["n"]

    @property
    def sim(self):
        return self.meta["sim"]

    @property
    def bounds(self):
        return tuple(self.meta["bounds"])

    @property
    def steps(self):
        return self.meta["steps"]

    def rng(self):
        return np.random.default_rng(self.meta["seed"])

    def init_velocities(self):
        s = self.meta.get("init_velocity_scale", 0.0)
        if not s:
            return None
        return np.random.default_rng(self.meta["seed"] + 1000).standard_normal((self.n, 2)) * s


_cache = {}


def load_golden(name):
    if name not in _cache:
        _cache[name] = Golden(name)
    return _cache[name]


@pytest.fixture
def golden():
    return load_golden


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
