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


AFM_DROP_SHAPES = [(8, 4, 128), (16, 6, 64), (32, 12, 8), (8, 12, 64), (4, 39, 64)]     # afm_drop.npz (training-mode AFM)


def pair_heavy_ok(N, E):
    """make_golden_pairs.py skips the per-pair E x E parameter variants ('mat', 'each') above 300 k elements"""
    return N * (N - 1) // 2 * E * E <= 300_000


def _report(kind, val):
    """TRS_TOL_REPORT=<file>: append (norm, value, calling test line) -- how the tolerances in tests/ were calibrated"""
    path = os.environ.get("TRS_TOL_REPORT")
    if path:
        import inspect
        fr = inspect.stack()[2]
        with open(path, "a") as f:
            f.write(f"{os.path.basename(fr.filename)}:{fr.lineno}\t{kind}\t{val:.3e}\n")
    return val


def rel_err(a, b):
    a = a.detach().double()
    b = b.detach().double()
    if os.environ.get("TRS_ROWS_REPORT") and a.dim() >= 2 and a.shape == b.shape and a.shape[0] > 1:
        # calibration aid: what the per-row norm would say where only the global one is asserted
        a2, b2 = a.reshape(a.shape[0], -1), b.reshape(b.shape[0], -1)
        scale = b2.abs().amax(dim=1).clamp_min(1e-2 * float(b2.abs().max().clamp_min(1e-30)))
        import inspect
        fr = inspect.stack()[1]
        with open(os.environ["TRS_ROWS_REPORT"], "a") as f:
            f.write(f"{os.path.basename(fr.filename)}:{fr.lineno}\trows\t{float(((a2 - b2).abs().amax(dim=1) / scale).max()):.3e}\t"
                    f"global\t{float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)):.3e}\n")
    return _report("rel_err", float((a - b).abs().max() / b.abs().max().clamp_min(1e-30)))


def rel_err_both(a, b, rows_slack=1.5):
    """max(global norm, per-row norm / rows_slack): the assert `rel_err_both(a, b) <= tol` holds the tensor to ``tol`` in
    the max norm AND every leading-index row to ``rows_slack * tol`` relative to the row's OWN largest reference value
    (floored at 1 % of the tensor's), so a wrong row of small values cannot hide behind a large value elsewhere.  Slack:
    a row's own maximum is typically 2-3 x below the tensor's while its absolute rounding error is not; calibrated over
    every assert site of tests/test_gpu_layers.py and tests/test_gpu_pairx.py (round 6, TRS_ROWS_REPORT): per-row errors
    1.7-2.4 x the global ones, never above 0.98e-2 for bf16 and 2.7e-6 for fp32."""
    g = rel_err(a, b)
    a2, b2 = a.detach().double(), b.detach().double()
    if a2.dim() < 2 or a2.shape != b2.shape or a2.shape[0] < 2:
        return g
    return max(g, rel_err_rows(a2, b2) / rows_slack)


def rel_err_rows(a, b, floor_frac=1e-2):
    """Worst PER-ROW relative error: every leading-index row is normalised by its OWN largest reference magnitude
    (floored at ``floor_frac`` of the global one, so an all-zero row does not divide by zero), so a row of small
    values that is wrong cannot hide behind a large value elsewhere in the tensor, as it can in ``rel_err``."""
    a = a.detach().double().reshape(a.shape[0], -1)
    b = b.detach().double().reshape(b.shape[0], -1)
    scale = b.abs().amax(dim=1).clamp_min(floor_frac * float(b.abs().max().clamp_min(1e-30)))
    return _report("rel_err_rows", float(((a - b).abs().amax(dim=1) / scale).max()))


def sum_err(a, b, terms_abs):
    """Error of a computed sum relative to the magnitude of what was summed: max |a-b| / terms_abs, where
    ``terms_abs`` = sum_i |term_i| of the reference (same shape as b).  This is the norm in which a floating-point
    sum is backward stable; |b| itself can be arbitrarily small through cancellation."""
    return float(((a.double() - b.double()).abs() / terms_abs.double().clamp_min(1e-30)).max())


BF16_U = 2.0 ** -8        # relative size of one bf16 rounding (round to nearest: 2**-9), doubled for two roundings


def batch_sum_err(a, b, terms_sq_sum, tol):
    """For a gradient that is a SUM OVER THE BATCH of signed terms (BatchNorm gamma / beta, biases), computed by a
    pipeline that stores its activations in bf16: every term carries an independent rounding error of relative
    size <= BF16_U per bf16 store it went through (contraction output, hidden state, their gradients: 3-5 stores), so
    the sum is off by a few BF16_U * sqrt(sum_i t_i^2) however much the terms cancel in the sum itself; the worst of
    several hundred channels sits ~4 standard deviations out.  Returns max over elements of
    |a-b| / (tol * max|b| + 8 * BF16_U * sqrt(sum t^2)); parity means <= 1."""
    a, b = a.double(), b.double()
    allowed = tol * b.abs().max() + 8.0 * BF16_U * terms_sq_sum.double().sqrt()
    return float(((a - b).abs() / allowed.clamp_min(1e-30)).max())
