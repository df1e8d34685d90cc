import os
import sys

# The oracle (OpenMP C sweep, numpy/LAPACK stages) checks small problems; on a 256-core GPU host the default "one thread
# per core" makes every tiny parallel region and LAPACK call cost milliseconds to seconds (observed: a 5-test file taking
# 14 minutes on a busy box).  Must be set before numpy / libgomp load.
for _var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_var, "8")
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def rel_fro(a, b):
    """Relative Frobenius error ||a-b||_F / ||b||_F (the parity metric of BASELINE.json)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.linalg.norm(b)
    return float(np.linalg.norm(a - b) / (den if den > 0 else 1.0))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def _fdx_switches_follow_the_environment(monkeypatch):
    """libfdx reads its FDX_* switches once and caches them (csrc/fdx_env.cpp): tests that set one through monkeypatch get the
    cache re-read at once, and every test starts from the environment as the previous test's undo left it."""
    import sys

    def reload():
        mod = sys.modules.get("flashdeconv_amd._lib")
        if mod is not None:
            mod.env_reload()

    reload()
    set0, del0 = monkeypatch.setenv, monkeypatch.delenv

    def setenv(name, value, *a, **k):
        set0(name, value, *a, **k)
        if str(name).startswith("FDX_"):
            reload()

    def delenv(name, *a, **k):
        del0(name, *a, **k)
        if str(name).startswith("FDX_"):
            reload()

    monkeypatch.setenv, monkeypatch.delenv = setenv, delenv
    yield
    monkeypatch.undo()
    reload()
