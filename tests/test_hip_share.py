"""Several handles on ONE device image of the training matrix (spfm_share_data): the fits of a
regularisation path / a grid / one-vs-rest targets -- what the reference runs one after the
other through warm_start (sparse_factorization_machines.py:380-391, base.py:130-136).  A tenant
that attaches to the image must train exactly as if it had uploaded the matrix itself."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _problem(n=3000, d=300, seed=0):
    from sparsepoly_amd.synth import make_problem

    X, y = make_problem(n, d, 12, seed=seed)
    return X.tocsr(), y


def _train(eng, solver, reg, k, d, gamma, iters=2, degree=2):
    eng.set_params(0.01 * np.random.RandomState(1).randn(degree - 1, k, d), np.zeros(d), np.ones(k))
    eng.configure(solver, "squared", reg, degree)
    eng.init_pred(degree, True, degree == 3)
    eng.set_schedule("colored", np.arange(d, dtype=np.int32))
    ic = np.arange(k, dtype=np.int32)
    v = []
    for _ in range(iters):
        t = eng.cd_linear_epoch(1.0)
        t += (eng.pcd_epoch(0, degree, 10.0, gamma, 1.0, ic) if solver == "pcd"
              else eng.pbcd_epoch(0, degree, 1.0, gamma, 1.0))
        v.append(t)
    P, w = eng.get_params()
    return np.array(v), P, w, eng.get_y_pred()


@pytest.mark.parametrize("precision", ["f32", "f64"])
@pytest.mark.parametrize("solver,reg", [("pcd", "squaredl12"), ("pbcd", "omegacs")])
def test_tenant_on_a_shared_image_equals_its_own_upload(solver, reg, precision):
    from sparsepoly_amd.engine import HipEngine

    X, y = _problem()
    d = X.shape[1]
    y2 = np.random.RandomState(5).randn(X.shape[0])  # the tenant's OWN targets
    solo = HipEngine(0, precision)
    solo.set_data(X, y2)
    ref = _train(solo, solver, reg, 6, d, 1e-3)
    solo.close()
    owner = HipEngine(0, precision)
    owner.set_data(X, y)
    ten = HipEngine(0, precision)
    ten.share_data(owner, y2)
    got_owner = _train(owner, solver, reg, 6, d, 1e-3)   # builds the entry stream
    got = _train(ten, solver, reg, 6, d, 1e-3)           # takes it from the cache
    if solver == "pcd":
        assert ten.get_option("stream_device_used") in (0, 2)  # 2 = a co-tenant's stream
    for a, b in zip(ref, got):
        np.testing.assert_array_equal(a, b)
    # the owner was not disturbed, and it trained on ITS targets
    assert not np.array_equal(got_owner[1], got[1])
    # targets default to the source's
    ten2 = HipEngine(0, precision)
    ten2.share_data(owner)
    again = _train(ten2, solver, reg, 6, d, 1e-3)
    for a, b in zip(got_owner, again):
        np.testing.assert_array_equal(a, b)
    # the image outlives the handle that uploaded it
    owner.close()
    once_more = _train(ten, solver, reg, 6, d, 1e-3)
    for a, b in zip(ref, once_more):
        np.testing.assert_array_equal(a, b)
    ten.close()
    ten2.close()


def test_share_data_argument_errors_and_detach():
    from sparsepoly_amd.engine import HipEngine

    X, y = _problem(500, 40)
    a = HipEngine(0, "f32")
    b = HipEngine(0, "f64")
    with pytest.raises(ValueError, match="no data"):
        b.share_data(a)
    a.set_data(X, y)
    with pytest.raises(ValueError, match="same device and storage type"):
        b.share_data(a)
    c = HipEngine(0, "f32")
    with pytest.raises(ValueError, match="entries"):
        c.share_data(a, y[:-1])
    c.share_data(a)
    # a new matrix on the tenant detaches it; the owner keeps training on the old one
    X2, y2 = _problem(400, 40, seed=3)
    c.set_data(X2, y2)
    assert (c.n, a.n) == (400, 500)
    ra = _train(a, "pcd", "l1", 3, 40, 1e-3)
    rc = _train(c, "pcd", "l1", 3, 40, 1e-3)
    solo = HipEngine(0, "f32")
    solo.set_data(X2, y2)
    rs = _train(solo, "pcd", "l1", 3, 40, 1e-3)
    for u, v in zip(rc, rs):
        np.testing.assert_array_equal(u, v)
    assert ra[3].shape == (500,)
    for e in (a, b, c, solo):
        e.close()


def test_concurrent_fits_share_one_image_and_equal_their_solo_runs():
    """fit_path over gamma: one upload for all clones (shared_image_ on the followers), results
    bit-identical to solo fits."""
    from sparsepoly_amd import SparseFactorizationMachineRegressor
    from sparsepoly_amd.concurrent import fit_path

    X, y = _problem(20000, 2000)
    base = SparseFactorizationMachineRegressor(
        degree=2, n_components=8, solver="pcd", regularizer="squaredl12", beta=10.0, max_iter=3,
        tol=0, random_state=0, schedule="colored", device=0)
    gammas = [1e-2, 1e-3, 1e-4, 1e-5]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        fitted = fit_path(base, X, y, gamma=gammas)
        assert sum(bool(e.shared_image_) for e in fitted) == len(gammas) - 1
        for g, est in zip(gammas, fitted):
            solo = SparseFactorizationMachineRegressor(**{**base.get_params(), "gamma": g}).fit(X, y)
            assert not solo.shared_image_
            np.testing.assert_array_equal(est.P_, solo.P_)
            np.testing.assert_array_equal(est.w_, solo.w_)
