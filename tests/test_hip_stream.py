"""The entry stream of the 64-column persistent passes built on the device (csrc/spfm_ingest.hip,
device_rowblock_stream) against the host builder (csrc/spfm_schedule.cpp, build_rowblock_stream):
the two must produce the same slot boundaries, the same entry order and the same long-slot masks,
so every epoch on top is bit-identical -- for several workgroup counts (empty row blocks), skewed
columns (long slots), both schedules, cd_linear and pcd, degree 2 and 3."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _matrix(n, d, per_row, seed, zipf=False):
    rng = np.random.RandomState(seed)
    rows = np.repeat(np.arange(n), per_row)
    if zipf:
        p = 1.0 / np.arange(1, d + 1) ** 1.1
        cols = rng.choice(d, size=n * per_row, p=p / p.sum())
    else:
        cols = rng.randint(0, d, size=n * per_row)
    X = sp.csr_matrix((rng.randn(n * per_row), (rows, cols)), shape=(n, d))
    X.sum_duplicates()
    X.sort_indices()
    return X, rng.randn(n)


def _run(X, y, device, schedule, groups, degree, precision="f32"):
    from sparsepoly_amd.engine import HipEngine

    eng = HipEngine(0, precision)
    eng.set_option("stream_device", device)
    eng.set_option("wide", 0)
    if groups:
        eng.set_option("prb_groups", groups)
    eng.set_data(X, y)
    d, k = X.shape[1], 3
    eng.set_params(0.01 * np.random.RandomState(0).randn(degree - 1, k, d), np.zeros(d), np.ones(k))
    eng.configure("pcd", "squared", "squaredl12" if degree == 2 else "omegati", degree)
    eng.init_pred(degree, True, degree > 2)
    eng.set_schedule(schedule, np.arange(d, dtype=np.int32))
    v = []
    for _ in range(2):
        a = eng.cd_linear_epoch(1.0)
        for deg in list(range(2, degree)) + [degree]:
            o = degree - deg if deg != degree else 0
            a += eng.pcd_epoch(o, deg, 5.0, 1e-3, 1.0, np.arange(k, dtype=np.int32))
        v.append(a)
    used = eng.get_option("stream_device_used")
    assert eng.get_option("persistent_active") == 1
    P, w = eng.get_params()
    out = (np.array(v), P, w, eng.get_y_pred())
    eng.close()
    return used, out


CASES = [
    # n, d, per_row, zipf, schedule, workgroups, degree
    (40000, 3000, 30, False, "colored", 0, 2),
    (40000, 3000, 30, False, "colored", 5, 2),       # few, big row blocks
    (40000, 3000, 30, False, "colored", 200, 2),     # 200 rows per block
    (2500, 1500, 700, False, "colored", 256, 2),     # ten rows per block, dense rows
    (80000, 2500, 24, True, "colored", 0, 2),        # very frequent columns: long slots
    (40000, 3000, 30, False, "exact", 0, 2),         # narrow steps: relaxed runs on top
    (40000, 3000, 30, False, "colored", 0, 3),
]


@pytest.mark.parametrize("n,d,per_row,zipf,schedule,groups,degree", CASES)
def test_device_stream_equals_host_stream(n, d, per_row, zipf, schedule, groups, degree):
    X, y = _matrix(n, d, per_row, seed=n % 89 + d % 17, zipf=zipf)
    assert X.nnz >= (1 << 20)
    dev = _run(X, y, 1, schedule, groups, degree)
    host = _run(X, y, 0, schedule, groups, degree)
    assert dev[0] == 1 and host[0] == 0
    for a, b in zip(dev[1], host[1]):
        assert np.array_equal(a, b)


def test_device_stream_f64_and_small_problem_on_the_host():
    X, y = _matrix(40000, 3000, 30, seed=5)
    dev = _run(X, y, 1, "colored", 0, 2, "f64")
    host = _run(X, y, 0, "colored", 0, 2, "f64")
    assert dev[0] == 1
    for a, b in zip(dev[1], host[1]):
        assert np.array_equal(a, b)
    Xs, ys = _matrix(3000, 300, 8, seed=6)            # under a million entries: host builder
    assert _run(Xs, ys, 1, "colored", 0, 2)[0] == 0
