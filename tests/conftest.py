import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "fa-vae_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


def fa_inputs(n=512, rank=5, seed=1):
    """Same generator as oracle/make_goldens.py:fa_inputs (inputs are regenerated, outputs are stored)."""
    rs = np.random.RandomState(seed)
    Zt = rs.randn(n, rank)
    X0 = Zt @ rs.randn(rank, 20) + 0.1 * rs.randn(n, 20)
    lab = (Zt @ rs.randn(rank, 40) > 0).astype(int)
    return X0, lab


def ss_inputs():
    """fa_inputs with a semi-supervised 'reg' view: the last 128 rows and every 7th entry of column 3 unobserved (NaN)."""
    X0, lab = fa_inputs()
    X0 = X0.copy()
    X0[384:, :] = np.nan
    X0[::7, 3] = np.nan
    return X0, lab


def ss2_inputs():
    """ss_inputs plus unobserved labels in the multilabel view (rows 300-419, every 5th entry of label 7)."""
    X0, lab = ss_inputs()
    lab = lab.astype(float)
    lab[300:420, :] = np.nan
    lab[::5, 7] = np.nan
    return X0, lab


def img_inputs(n, seed=1, zt=None):
    """Same generator as oracle/make_goldens.py:img_inputs."""
    rs = np.random.RandomState(seed)
    Zt = rs.randn(n, 5) if zt is None else zt
    Pm = rs.randn(5, 3 * 64 * 64) * 0.7
    img = (1 / (1 + np.exp(-(Zt @ Pm)))).reshape(n, 3, 64, 64).astype(np.float32)
    lab = (Zt @ rs.randn(5, 40) > 0).astype(int)
    return img, lab


def checksum(t):
    a = np.asarray(t, dtype=np.float64).ravel()
    idx = np.linspace(0, a.size - 1, 64).astype(np.int64)
    return np.concatenate(([a.sum(), np.sqrt((a * a).sum())], a[idx]))
