"""All-subsets model on the device vs the reference-generated
goldens (g8) and the oracle.  Needs a real MI355X: ``pytest -m gpu``."""
import json
import warnings

import numpy as np
import pytest
import scipy.sparse as sp
from conftest import load_golden

pytestmark = pytest.mark.gpu

P_ATOL = {"f64": 1e-8, "f32": 1e-4}
TRAJ_RTOL = {"f64": 1e-9, "f32": 1e-5}


def _cells():
    return [str(c) for c in load_golden("g8_all_subsets.npz")["cells"]]


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("cell", _cells())
def test_reference_test_cells(cell, precision):
    """reference tests/test_pcd.py:354-432, tests/test_pbcd.py:345-425 through the estimators."""
    from sparsepoly_amd import SparseAllSubsetsClassifier, SparseAllSubsetsRegressor

    z = load_golden("g8_all_subsets.npz")
    solver, regname, mean, loss = cell.split("|")
    kw = dict(n_components=5, beta=1, gamma=1e-3, regularizer=regname, warm_start=False, tol=1e-3,
              max_iter=5, random_state=0, mean=bool(int(mean[4:])), shuffle=False, solver=solver,
              precision=precision)
    if loss == "squared":
        est, y = SparseAllSubsetsRegressor(**kw), z["y"]
    else:
        est, y = SparseAllSubsetsClassifier(loss=loss, **kw), np.sign(z["y"])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        est.fit(z["X"], y)
    np.testing.assert_allclose(est.P_, z["P|" + cell], rtol=0, atol=P_ATOL[precision])
    assert est.n_iter_ == int(z["n_iter|" + cell])
    got = est.decision_function(z["X"]) if loss != "squared" else est.predict(z["X"])
    np.testing.assert_allclose(got, z["pred|" + cell], rtol=0,
                               atol=1e-7 if precision == "f64" else 2e-3)


def _scases():
    return [str(c) for c in load_golden("g8_all_subsets.npz")["scases"]]


def _run(X, y, solver, regname, loss, P0, lams, meta, precision, schedule, options=None):
    from sparsepoly_amd.engine import HipEngine

    d = X.shape[1]
    k = P0.shape[0]
    eng = HipEngine(0, precision)
    for key, val in (options or {}).items():
        eng.set_option(key, val)
    eng.set_data(X, y)
    eng.set_params(P0[None], np.zeros(d), lams)
    eng.configure(solver, loss, regname, -1)
    eng.init_pred(-1, False, False)
    order = eng.set_schedule(schedule, np.arange(d, dtype=np.int32))
    viol, lo = [], []
    for it in range(3):
        if solver == "pcd":
            v = eng.pcd_epoch(0, -1, meta["beta"], meta["gamma"], meta["eta0"],
                              np.arange(k, dtype=np.int32))
        else:
            v = eng.pbcd_epoch(0, -1, meta["beta"], meta["gamma"], meta["eta0"])
        viol.append(v)
        lo.append(eng.loss_sum())
    P, _ = eng.get_params()
    yp = eng.get_y_pred()
    eng.close()
    return order, viol, lo, P[0], yp


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("case", _scases())
def test_sparse_trajectories(case, precision):
    z = load_golden("g8_all_subsets.npz")
    X = sp.csr_matrix((z["Xs_data"], z["Xs_indices"], z["Xs_indptr"]),
                      shape=tuple(int(v) for v in z["Xs_shape"]))
    solver, regname, loss = case.split("|")
    meta = json.loads(str(z["smeta"]))
    ys = z["ys"]
    y = ys if loss == "squared" else np.where(ys > np.median(ys), 1.0, -1.0)
    _, viol, lo, P, yp = _run(X, y, solver, regname, loss, z["sP0|" + case], z["slams|" + case],
                              meta, precision, "exact")
    np.testing.assert_allclose(viol, z["sviol|" + case], rtol=TRAJ_RTOL[precision])
    np.testing.assert_allclose(lo, z["sloss|" + case], rtol=TRAJ_RTOL[precision])
    np.testing.assert_allclose(P, z["sP|" + case], rtol=0, atol=P_ATOL[precision])
    np.testing.assert_allclose(yp, z["sy_pred|" + case], rtol=0,
                               atol=1e-7 if precision == "f64" else 1e-3)


@pytest.mark.parametrize("options", [None, {"persistent": 0}, {"persistent": 0, "fuse_chain": 0},
                                     {"prb_groups": 2, "prb_long": 16}])
@pytest.mark.parametrize("case", ["pcd|omegati|squared", "pcd|l1|logistic",
                                  "pbcd|omegacs|squared", "pbcd|l21|logistic"])
def test_colored_schedule_and_engines_vs_oracle(oracle, case, options):
    """Coloured order through both engines (persistent / multi-kernel, long slots) equals the
    oracle's sequential sweep in the reported order."""
    z = load_golden("g8_all_subsets.npz")
    X = sp.csr_matrix((z["Xs_data"], z["Xs_indices"], z["Xs_indptr"]),
                      shape=tuple(int(v) for v in z["Xs_shape"]))
    solver, regname, loss = case.split("|")
    meta = json.loads(str(z["smeta"]))
    ys = z["ys"]
    y = ys if loss == "squared" else np.where(ys > np.median(ys), 1.0, -1.0)
    order, viol, lo, P, yp = _run(X, y, solver, regname, loss, z["sP0|" + case],
                                  z["slams|" + case], meta, "f64", "colored", options)
    fm = oracle.OracleAllSubsets(loss=loss, n_components=P.shape[0], solver=solver,
                                 beta=meta["beta"], gamma=meta["gamma"], eta0=meta["eta0"],
                                 regularizer=regname, tol=0, max_iter=3, feature_order=order)
    fm.fit(X, y, P_init=z["sP0|" + case], lams_init=z["slams|" + case])
    np.testing.assert_allclose(viol, [h[0] for h in fm.history], rtol=1e-9)
    np.testing.assert_allclose(lo, [h[1] for h in fm.history], rtol=1e-9)
    np.testing.assert_allclose(P, fm.P_, rtol=0, atol=1e-8)
    np.testing.assert_allclose(yp, fm.y_pred_, rtol=0, atol=1e-7)


def test_api_and_errors():
    from sklearn.base import clone
    from sklearn.exceptions import NotFittedError

    from sparsepoly_amd import SparseAllSubsetsClassifier, SparseAllSubsetsRegressor

    z = load_golden("g8_all_subsets.npz")
    X, y = z["X"], z["y"]
    est = SparseAllSubsetsRegressor(n_components=3, max_iter=2, tol=0, random_state=0)
    assert clone(est).get_params() == est.get_params() and est.eta0 == 0.1
    assert est.regularizer == "omegati"
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        est.fit(X, y)
    assert est.n_iter_ == 1 and est.P_.shape == (3, 4)
    assert str(w[-1].message) == "Objective did not converge. Increase max_iter."
    with pytest.raises(ValueError, match="Regularizer squaredl12 not supported"):
        SparseAllSubsetsRegressor(regularizer="squaredl12").fit(X, y)
    with pytest.raises(ValueError):  # l21 has no pcd protocol
        SparseAllSubsetsRegressor(regularizer="l21", solver="pcd", max_iter=1).fit(X, y)
    with pytest.raises(ValueError, match="Solver nope is not supported."):
        SparseAllSubsetsRegressor(solver="nope").fit(X, y)
    with pytest.raises(NotFittedError):
        SparseAllSubsetsRegressor().predict(X)
    clf = SparseAllSubsetsClassifier(loss="logistic", max_iter=2, random_state=0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        clf.fit(X, np.where(y > 0, "p", "n"))
    pr = clf.predict_proba(X)
    assert pr.shape == (20,) and np.all((pr > 0) & (pr < 1))
    assert set(clf.predict(X)) <= {"p", "n"}
