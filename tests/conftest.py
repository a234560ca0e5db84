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


class Golden:
    """Lazy accessor for tests/golden/<name>.npz with 'a/b/c' keys."""

    def __init__(self, name):
        self._z = np.load(os.path.join(GOLDEN, name + ".npz"))

    def __call__(self, key, dtype=None):
        a = self._z[key]
        if a.dtype.kind in "US":
            return [str(s) for s in a]
        t = torch.from_numpy(a.copy())
        return t if dtype is None else t.to(dtype)

    def has(self, key):
        return key in self._z.files

    def keys(self):
        return list(self._z.files)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = Golden(name)
        return cache[name]

    return get


LAYER_SHAPES = [(8, 4, 128), (16, 6, 64), (32, 12, 8), (32, 10, 16), (8, 39, 64)]
MODEL_SHAPES = [(32, 10, 16), (16, 39, 64), (16, 6, 64)]
CIN_CASES = ["a", "b", "c", "d", "e", "f"]
PAIR_SHAPES = [(8, 4, 128), (16, 6, 64), (32, 12, 8), (32, 10, 16), (8, 12, 64), (4, 39, 64)]   # pairs.npz (8f N3 layers)


def pair_heavy_ok(N, E):
    """make_golden_pairs.py skips the per-pair E x E parameter variants ('mat', 'each') above 300 k elements"""
    return N * (N - 1) // 2 * E * E <= 300_000


def rel_err(a, b):
    a = a.double()
    b = b.double()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
