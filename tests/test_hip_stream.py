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


# ---- round 4: the pbcd stream (balanced slot groups) and the wide stream on the device
def _run_pbcd(X, y, device, groups, balance, k=6, degree=2, precision="f32"):
    from sparsepoly_amd.engine import HipEngine

    eng = HipEngine(0, precision)
    eng.set_option("stream_device", device)
    eng.set_option("pbprb_balance", balance)
    if groups:
        eng.set_option("pbprb_groups", groups)
    eng.set_data(X, y)
    d = X.shape[1]
    eng.set_params(0.01 * np.random.RandomState(0).randn(degree - 1, k, d), np.zeros(d), np.ones(k))
    eng.configure("pbcd", "squared", "omegacs", degree)
    eng.init_pred(degree, True, degree > 2)
    eng.set_schedule("colored", np.arange(d, dtype=np.int32))
    v = [eng.pbcd_epoch(0, degree, 1.0, 1e-3, 1.0) for _ in range(2)]
    used = eng.get_option("pb_stream_device_used")
    assert eng.get_option("pbprb_active") == 1
    P, w = eng.get_params()
    out = (np.array(v), P, w, eng.get_y_pred())
    eng.close()
    return used, out


@pytest.mark.parametrize("n,d,per_row,zipf,groups,balance,k,degree", [
    (40000, 3000, 30, False, 0, 1, 6, 2),      # 256 row blocks, balanced groups
    (40000, 3000, 30, False, 0, 0, 6, 2),      # the fixed map slot -> group
    (40000, 3000, 30, False, 37, 1, 6, 2),     # fewer workgroups than slots: owner rounds
    (80000, 2500, 24, True, 100, 1, 6, 2),     # very frequent columns: entries beyond the LDS rows
    (40000, 3000, 30, False, 0, 1, 40, 2),     # 64 lanes per group: 8 groups x 8 slots
    (40000, 3000, 30, False, 0, 1, 5, 3),
])
def test_device_pbcd_stream_equals_host_stream(n, d, per_row, zipf, groups, balance, k, degree):
    X, y = _matrix(n, d, per_row, seed=n % 89 + d % 17 + 1, zipf=zipf)
    assert X.nnz >= (1 << 20)
    dev = _run_pbcd(X, y, 1, groups, balance, k, degree)
    host = _run_pbcd(X, y, 0, groups, balance, k, degree)
    assert dev[0] == 1 and host[0] == 0
    for a, b in zip(dev[1], host[1]):
        assert np.array_equal(a, b)


def test_balanced_groups_change_the_order_of_the_sums_only(oracle):
    """Balanced slot groups deal a step's columns to other groups than the fixed map: the same
    arithmetic in another grouping -- equal to rounding, and both equal the oracle."""
    X, y = _matrix(40000, 3000, 30, seed=11)
    a = _run_pbcd(X, y, 0, 0, 1, precision="f64")[1]
    b = _run_pbcd(X, y, 0, 0, 0, precision="f64")[1]
    np.testing.assert_allclose(a[0], b[0], rtol=1e-10)
    np.testing.assert_allclose(a[1], b[1], rtol=0, atol=1e-10)


def _run_wide(X, y, device, groups):
    from sparsepoly_amd.engine import HipEngine

    eng = HipEngine(0, "f32")
    eng.set_option("stream_device", device)
    eng.set_option("wide_min_cols", 1)  # take the wide pass whatever the class width
    if groups:
        eng.set_option("pcdw_groups", groups)
    eng.set_data(X, y)
    d, k = X.shape[1], 3
    eng.set_params(0.01 * np.random.RandomState(0).randn(1, k, d), np.zeros(d), np.ones(k))
    eng.configure("pcd", "squared", "squaredl12", 2)
    eng.init_pred(2, True, False)
    eng.set_schedule("colored", np.arange(d, dtype=np.int32))
    assert eng.get_option("wide_active") == 1
    v = []
    for _ in range(2):
        v.append(eng.cd_linear_epoch(1.0)
                 + eng.pcd_epoch(0, 2, 5.0, 1e-3, 1.0, np.arange(k, dtype=np.int32)))
    used = eng.get_option("wide_stream_device_used")
    P, w = eng.get_params()
    out = (np.array(v), P, w, eng.get_y_pred())
    eng.close()
    return used, out


@pytest.mark.parametrize("n,d,per_row,groups", [(200000, 30000, 8, 0), (200000, 30000, 8, 48)])
def test_device_wide_stream_equals_host_stream(n, d, per_row, groups):
    X, y = _matrix(n, d, per_row, seed=3)
    assert X.nnz >= (1 << 20)
    dev = _run_wide(X, y, 1, groups)
    host = _run_wide(X, y, 0, groups)
    assert dev[0] == 1 and host[0] == 0
    for a, b in zip(dev[1], host[1]):
        assert np.array_equal(a, b)
