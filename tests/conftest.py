import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# the test process starts concurrent fits long after its first GPU call: configure the HIP
# runtime's hardware queues while that still works (sparsepoly_amd/_capi.py ensure_hw_queues)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def golden_csr(z):
    shape = tuple(int(v) for v in z["X_shape"])
    return sp.csr_matrix((z["X_data"], z["X_indices"], z["X_indptr"]), shape=shape)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with gcc."""
    from oracle import oracle as orc

    orc.build()
    orc.lib()
    return orc
