import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "m3f.pytorch_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


_BLAS_LIMIT = []


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the numpy oracle's fp64 GEMMs on at most 4 BLAS threads: with every core in the pool, a second busy process on the machine (a build, another
    # suite) turns OpenBLAS' spinning workers against each other -- the full-size C3 oracle test went from 10 s to 700 s that way (round 6)
    try:
        from threadpoolctl import threadpool_limits
        _BLAS_LIMIT.append(threadpool_limits(limits=min(4, os.cpu_count() or 1), user_api="blas"))
    except Exception:      # noqa: BLE001  (threadpoolctl missing: the default pool)
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
