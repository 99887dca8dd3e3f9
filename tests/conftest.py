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
    # the numpy oracle's fp64 GEMMs on ONE BLAS thread: the full-size C3 oracle test takes 13 s that way against 10 s on every core -- and 700-900 s
    # on every core when the machine's vCPUs are contended (a build beside the suite, a noisy neighbour of the VM: OpenBLAS' spinning workers
    # wait for a descheduled sibling at every call; seen twice in round 6, once with nothing else running in the container)
    # (the CPU suite only: the GPU suite's float64 references run on the GPU box's own cores)
    if "not gpu" in (config.getoption("markexpr", "") or ""):
        try:
            from threadpoolctl import threadpool_limits
            _BLAS_LIMIT.append(threadpool_limits(limits=1, user_api="blas"))
        except Exception:      # noqa: BLE001  (threadpoolctl missing: the default pool)
            pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def load_golden(name):
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False))
