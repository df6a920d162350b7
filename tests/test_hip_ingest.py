"""Device-side CSR -> CSC ingest (csrc/spfm_ingest.hip; SURVEY.md 8f N4, the replacement of the
reference's X.tocsc(), dataset.py:119-123) against the host-thread transposition and the CSC entry
point: the three must install bit-identical images, so schedules, epochs and predictions agree
exactly; malformed CSR input is refused with the same error on both ingest paths."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _epochs(X, y, dtype, how, degree=2):
    from sparsepoly_amd.engine import HipEngine

    eng = HipEngine(0, dtype)
    if how == "host":
        eng.set_option("ingest_device", 0)
    eng.set_data(X.tocsc() if how == "csc" else X.tocsr(), y)
    used = eng.get_option("ingest_device_used")
    d = X.shape[1]
    k = 4
    n_orders = degree - 1
    eng.set_params(0.01 * np.random.RandomState(0).randn(n_orders, k, d), np.zeros(d), np.ones(k))
    eng.configure("pcd", "squared", "l1", degree)
    eng.init_pred(degree, True, degree > 2)
    order = eng.set_schedule("colored", np.arange(d, dtype=np.int32))
    v = eng.cd_linear_epoch(1.0)
    for o in range(n_orders):
        v += eng.pcd_epoch(o, degree - o, 10.0, 1e-3, 1.0, np.arange(k, dtype=np.int32))
    P, w = eng.get_params()
    out = (order, v, P, w, eng.get_y_pred(), eng.predict(X.tocsr()[: min(64, X.shape[0])], degree,
                                                         True, degree > 2))
    eng.close()
    return used, out


def _matrix(n, d, per_row, seed, empty_rows=True, empty_cols=True):
    rng = np.random.RandomState(seed)
    X = sp.random(n, d, density=per_row / d, format="lil", random_state=rng, data_rvs=rng.randn)
    X = X.tocsr()
    if empty_rows and n > 4:
        keep = np.ones(n)
        keep[[0, n // 2, n - 1]] = 0
        X = sp.diags(keep) @ X
    if empty_cols and d > 4:
        keep = np.ones(d)
        keep[[0, d // 3, d - 1]] = 0
        X = X @ sp.diags(keep)
    X = sp.csr_matrix(X)
    X.eliminate_zeros()
    X.sort_indices()
    return X, rng.randn(n)


@pytest.mark.parametrize("dtype", ["f32", "f64"])
@pytest.mark.parametrize("shape", [(300, 40, 6), (5000, 700, 12), (40000, 3000, 30), (7, 3, 2),
                                   (1, 5, 3), (2000, 1, 1)])
def test_device_ingest_equals_host_ingest_and_csc(dtype, shape):
    n, d, per_row = shape
    X, y = _matrix(n, d, per_row, seed=n + d)
    res = {}
    for how in ("device", "host", "csc"):
        used, res[how] = _epochs(X, y, dtype, how)
        assert used == (1 if how == "device" else 0)
    for how in ("host", "csc"):
        for a, b in zip(res["device"], res[how]):
            assert np.array_equal(a, b)


def test_device_ingest_degree3_and_all_zero_matrix():
    X, y = _matrix(3000, 200, 8, seed=3)
    used, a = _epochs(X, y, "f32", "device", degree=3)
    _, b = _epochs(X, y, "f32", "host", degree=3)
    assert used == 1
    for u, v in zip(a, b):
        assert np.array_equal(u, v)
    Z = sp.csr_matrix((50, 9))
    used, a = _epochs(Z, np.ones(50), "f64", "device")
    _, b = _epochs(Z, np.ones(50), "f64", "host")
    assert used == 1
    for u, v in zip(a, b):
        assert np.array_equal(u, v)


@pytest.mark.parametrize("device", [1, 0])
@pytest.mark.parametrize("defect", ["unsorted", "duplicate", "negative", "too_big"])
def test_malformed_csr_is_refused_on_both_paths(device, defect):
    from sparsepoly_amd import _capi
    from sparsepoly_amd.engine import HipEngine

    X, y = _matrix(400, 30, 6, seed=11, empty_rows=False, empty_cols=False)
    idx = X.indices.copy()
    r = int(np.argmax(np.diff(X.indptr) >= 3))
    lo = X.indptr[r]
    if defect == "unsorted":
        idx[lo], idx[lo + 1] = idx[lo + 1], idx[lo]
    elif defect == "duplicate":
        idx[lo + 1] = idx[lo]
    elif defect == "negative":
        idx[lo] = -1
    else:
        idx[X.indptr[r + 1] - 1] = X.shape[1]
    eng = HipEngine(0, "f32")
    eng.set_option("ingest_device", device)
    ip, ii, dd, yy = _capi.i64(X.indptr), _capi.i32(idx), _capi.f64(X.data), _capi.f64(y)
    rc = eng._lib.spfm_set_data_csr(eng._h, X.shape[0], X.shape[1], ip[1], ii[1], dd[1], yy[1])
    assert rc == _capi.SPFM_ERR_INVALID
    # the handle stays usable: a well-formed matrix goes in afterwards
    eng.set_data(X, y)
    assert eng.get_option("ingest_device_used") == device
    eng.close()


def test_estimator_fit_on_csr_uses_the_device_ingest():
    from sparsepoly_amd import SparseFactorizationMachineRegressor

    X, y = _matrix(2000, 120, 10, seed=5)
    a = SparseFactorizationMachineRegressor(n_components=3, max_iter=3, tol=0, random_state=0,
                                            beta=1.0, gamma=1e-3).fit(X, y)
    b = SparseFactorizationMachineRegressor(n_components=3, max_iter=3, tol=0, random_state=0,
                                            beta=1.0, gamma=1e-3).fit(X.tocsc(), y)
    assert np.array_equal(a.P_, b.P_) and np.array_equal(a.w_, b.w_)
    assert np.array_equal(a.predict(X), b.predict(X))
