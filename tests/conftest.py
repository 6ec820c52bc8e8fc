import os
import sys

import pytest

ROOT = os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
for p in (os.path.join(ROOT, "spart-python_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import spart_oracle
    return spart_oracle


@pytest.fixture(scope="session")
def tables(oracle):
    return oracle.load_tables()


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    d = os.path.join(ROOT, "tests", "golden")
    return {n: np.load(os.path.join(d, n + ".npz")) for n in ("prospect", "bsm", "sailh", "smac", "e2e", "rdry", "edge", "jpl", "grids", "s2_f64", "canopy_state", "config2")}


def rel_err(a, b, floor=1e-6):
    """max |a-b| / max(|b|, floor) over entries where the reference b is finite (NaN-aware)."""
    import numpy as np
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    m = np.isfinite(b)
    assert np.isfinite(a[m]).all(), "non-finite result where the reference is finite"
    if not m.any():
        return 0.0
    return float(np.max(np.abs(a[m] - b[m]) / np.maximum(np.abs(b[m]), floor)))
