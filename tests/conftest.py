import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


def rel_err(a, b):
    """SURVEY §8d parity metric: |a-b| / max(|b|, 1e-6*max|b|) per cell."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    scale = np.maximum(np.abs(b), 1e-6 * max(np.max(np.abs(b)), 1e-300))
    return np.abs(a - b) / scale


def parity_report(got, ref, keys, tol):
    """max rel err, 99.99 percentile and number of cells above tol for each output."""
    rep = {}
    for k in keys:
        e = rel_err(got[k], ref[k])
        rep[k] = dict(max=float(e.max()), p9999=float(np.quantile(e, 0.9999)), n_bad=int((e > tol).sum()))
    return rep


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle
    if not os.path.exists(pyoracle.ORACLE_SO):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), os.path.join(ROOT, "oracle", "liboracle.so")])
    return pyoracle
