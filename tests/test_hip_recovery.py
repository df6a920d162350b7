"""A persistent pass that cannot finish (a workgroup that is not resident, a peer that never
answers) must not leave the model half-updated: the reference's epochs are all-or-nothing
(pcd.py:71-137, pbcd.py:82-148, cd_linear.py:8-33).  The library snapshots the epoch's parameters,
and after a time-out restores them, recomputes y_pred and redoes the epoch on the multi-kernel
engine.  Forced here by launching a pass WITHOUT its last workgroup (option
``debug_drop_group``) with a short poll bound (``debug_spin_max``): the others wait for it, give
up, and the epoch's result must still equal the oracle's.  Needs a real MI355X."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _problem(n, d, per_row, seed):
    rng = np.random.RandomState(seed)
    rows = np.repeat(np.arange(n), per_row)
    cols = rng.randint(0, d, size=n * per_row)
    vals = rng.randn(n * per_row).astype(np.float32).astype(np.float64)
    X = sp.csr_matrix((vals, (rows, cols)), shape=(n, d))
    X.sum_duplicates()
    X.sort_indices()
    y = rng.randn(n).astype(np.float32).astype(np.float64)
    return X, y


CASES = {
    # name: (solver, regularizer, degree, problem, expected engine of the first try)
    "pcd_64col": ("pcd", "squaredl12", 2, (3000, 400, 12, 3), "persistent_active"),
    "pcd_degree3": ("pcd", "omegati", 3, (3000, 400, 12, 4), "persistent_active"),
    "pcd_wide": ("pcd", "squaredl12", 2, (6000, 3000, 4, 11), "wide_active"),
    "pbcd": ("pbcd", "omegacs", 2, (3000, 400, 12, 5), "pbprb_active"),
}


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("case", sorted(CASES))
@pytest.mark.parametrize("fail_at", ["cd_linear", "factor"])
def test_epoch_survives_a_persistent_pass_time_out(oracle, case, precision, fail_at):
    from sparsepoly_amd.engine import HipEngine

    solver, reg, degree, prob, active_key = CASES[case]
    X, y = _problem(*prob)
    d, k = X.shape[1], 4
    beta = 10.0 if solver == "pcd" else 1.0
    n_orders = degree - 1
    eng = HipEngine(0, precision)
    eng.set_option("debug_spin_max", 4096)
    eng.set_data(X, y)
    P0 = 0.05 * np.random.RandomState(1).randn(n_orders, k, d)
    eng.set_params(P0, np.zeros(d), np.ones(k))
    eng.configure(solver, "squared", reg, degree)
    eng.init_pred(degree, True, degree == 3)
    order = eng.set_schedule("colored", np.arange(d, dtype=np.int32))
    if active_key != "pbprb_active":
        assert eng.get_option(active_key) == 1
    ic = np.arange(k, dtype=np.int32)

    def factor_epochs():
        v = 0.0
        for deg in list(range(2, degree)) + [degree]:
            o = degree - deg if deg != degree else 0
            if solver == "pcd":
                v += eng.pcd_epoch(o, deg, beta, 1e-3, 1.0, ic)
            else:
                v += eng.pbcd_epoch(o, deg, beta, 1e-3, 1.0)
        return v

    viol = [eng.cd_linear_epoch(0.5) + factor_epochs()]   # a clean iteration, persistent passes
    assert eng.get_option("persistent_fallbacks") == 0 and eng.get_option("persistent_failed") == 0
    if active_key == "pbprb_active":
        assert eng.get_option("pbprb_active") == 1
    # second iteration: the first persistent launch of the chosen epoch lacks a workgroup; that
    # epoch times out, is rolled back and redone on the multi-kernel engine
    if fail_at == "cd_linear":
        eng.set_option("debug_drop_group", 1)
        v = eng.cd_linear_epoch(0.5)
        assert eng.get_option("persistent_fallbacks") == 1
        v += factor_epochs()
    else:
        v = eng.cd_linear_epoch(0.5)
        assert eng.get_option("persistent_fallbacks") == 0
        eng.set_option("debug_drop_group", 1)
        v += factor_epochs()
        assert eng.get_option("persistent_fallbacks") == 1
    viol.append(v)
    assert eng.get_option("persistent_failed") == 1
    viol.append(eng.cd_linear_epoch(0.5) + factor_epochs())  # the handle keeps working
    assert eng.get_option("persistent_fallbacks") == 1
    P, w = eng.get_params()
    yp = eng.get_y_pred()
    eng.close()
    fm = oracle.OracleFM(degree=degree, loss="squared", n_components=k, solver=solver,
                         regularizer=reg, alpha=0.5, beta=beta, gamma=1e-3, tol=0, max_iter=3,
                         fit_linear=True, feature_order=order)
    fm.fit(X, y, P_init=P0, lams_init=np.ones(k))
    ref = [h[0] for h in fm.history]
    if precision == "f64":
        np.testing.assert_allclose(viol, ref, rtol=1e-9)
        np.testing.assert_allclose(P, fm.P_, rtol=0, atol=1e-8)
        np.testing.assert_allclose(w, fm.w_, rtol=0, atol=1e-8)
        np.testing.assert_allclose(yp, fm.y_pred_, rtol=0, atol=1e-7)
    else:
        np.testing.assert_allclose(viol, ref, rtol=5e-5)
        np.testing.assert_allclose(P, fm.P_, rtol=0, atol=1e-4)
        np.testing.assert_allclose(w, fm.w_, rtol=0, atol=1e-4)
