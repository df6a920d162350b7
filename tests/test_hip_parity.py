"""Parity of the HIP path (through the C ABI) with the CPU oracle and with the
reference-generated golden fixtures.  Needs a real MI355X: ``pytest -m gpu``.

Tolerances (stated per SURVEY.md section 8c):
  precision='f64' : |P - P_ref| <= 1e-8, viol/loss trajectories 1e-9 relative
                    (differences: reduction order, FMA-free but re-associated sums)
  precision='f32' : |P - P_ref| <= 1e-4 (the reference's own decimal=4),
                    viol/loss trajectories <= 1e-5 relative (north_star)
"""
import json
import warnings

import numpy as np
import pytest
import scipy.sparse as sp
from conftest import golden_csr, load_golden

pytestmark = pytest.mark.gpu

PREC = ["f64", "f32"]
P_ATOL = {"f64": 1e-8, "f32": 1e-4}
TRAJ_RTOL = {"f64": 1e-9, "f32": 1e-5}


def _est(loss, **kw):
    from sparsepoly_amd import (SparseFactorizationMachineClassifier,
                                SparseFactorizationMachineRegressor)

    if loss == "squared":
        return SparseFactorizationMachineRegressor(**kw)
    return SparseFactorizationMachineClassifier(loss=loss, **kw)


def _g1_cells():
    return [str(c) for c in load_golden("g1_reftests.npz")["cells"]]


@pytest.mark.parametrize("precision", PREC)
@pytest.mark.parametrize("cell", _g1_cells())
def test_g1_reference_test_cells(cell, precision):
    """The reference's own test cells (tests/test_pcd.py:177-351, test_pbcd.py:163-342):
    estimator.fit on dense RandomState(1) data, P_ compared as the reference does."""
    z = load_golden("g1_reftests.npz")
    solver, regname, deg, mean, loss = cell.split("|")
    degree, mean = int(deg[3:]), bool(int(mean[4:]))
    y = z["y_deg%d" % degree]
    if loss != "squared":
        y = np.sign(y)
    max_iter = 1 if regname in ("squaredl12", "squaredl21") else 5
    est = _est(loss, degree=degree, n_components=5, fit_lower=None, fit_linear=False, beta=1,
               gamma=1e-3, regularizer=regname, warm_start=False, tol=1e-3, max_iter=max_iter,
               random_state=0, mean=mean, shuffle=False, solver=solver, precision=precision)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        est.fit(z["X"], y)
    np.testing.assert_allclose(est.P_, z["P|" + cell], rtol=0, atol=P_ATOL[precision])
    assert est.n_iter_ == int(z["n_iter|" + cell])


@pytest.mark.parametrize("precision", PREC)
@pytest.mark.parametrize("gamma", [1e-3, 1e-2])
def test_g2_config1(gamma, precision, capsys):
    """BASELINE config 1: 1k x 100 CSR, degree 2, k=4, l1, pcd."""
    z = load_golden("g2_config1.npz")
    X = golden_csr(z)
    tag = "gamma%g" % gamma
    est = _est("squared", degree=2, n_components=4, regularizer="l1", solver="pcd", gamma=gamma,
               max_iter=6, tol=1e-9, random_state=0, verbose=True, precision=precision)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        est.fit(X, z["y"])
    out = capsys.readouterr().out
    viol = np.array([float(l.split()[-1]) for l in out.splitlines() if l.startswith("Iteration")])
    np.testing.assert_allclose(viol, z["viol|" + tag], rtol=TRAJ_RTOL[precision])
    np.testing.assert_allclose(est.P_, z["P|" + tag], rtol=0, atol=P_ATOL[precision])
    np.testing.assert_allclose(est.w_, z["w|" + tag], rtol=0, atol=P_ATOL[precision])
    np.testing.assert_allclose(est.predict(X), z["pred|" + tag], rtol=0,
                               atol=1e-7 if precision == "f64" else 1e-3)
    assert est.n_iter_ == int(z["n_iter|" + tag])


class _Run(object):
    """Drives one HipEngine epoch by epoch (what the estimators do internally) and
    records viol / sum-loss per iteration."""

    def __init__(self, X, y, meta, P0, lams, precision, schedule="exact", orders=None,
                 corders=None, eta0=1.0, n_epochs=4, use_graph=True, options=None):
        from sparsepoly_amd.engine import HipEngine

        n, d = X.shape
        k, degree = meta["k"], meta["degree"]
        eng = HipEngine(0, precision)
        eng.set_use_graph(use_graph)
        for key, val in (options or {}).items():
            eng.set_option(key, val)
        eng.set_data(X, y)
        eng.set_params(P0, np.zeros(d), lams)
        eng.configure(meta["solver"], meta["loss"], meta["regularizer"], degree)
        eng.init_pred(degree, False, meta.get("fit_lower", "explicit") == "explicit"
                      and degree == 3)
        self.viol, self.loss = [], []
        jf = np.arange(d, dtype=np.int32)
        ic = np.arange(k, dtype=np.int32)
        if orders is None:
            self.order = eng.set_schedule(schedule, jf)
        for it in range(n_epochs):
            if orders is not None:
                self.order = eng.set_schedule(schedule, orders[it])
            if corders is not None:
                ic = corders[it]
            v = eng.cd_linear_epoch(meta["alpha"])
            for deg in list(range(2, degree)) + [degree]:
                o = degree - deg if deg != degree else 0
                if meta["solver"] == "pcd":
                    v += eng.pcd_epoch(o, deg, meta["beta"], meta["gamma"], eta0, ic)
                else:
                    v += eng.pbcd_epoch(o, deg, meta["beta"], meta["gamma"], eta0)
            self.viol.append(v)
            self.loss.append(eng.loss_sum())
        self.P, self.w = eng.get_params()
        self.y_pred = eng.get_y_pred()
        self.n_batches = eng.n_batches
        self.pbprb_active = eng.get_option("pbprb_active")
        eng.close()


def _g3_cases():
    return [str(c) for c in load_golden("g3_small_configs.npz")["cases"]]


@pytest.mark.parametrize("precision", PREC)
@pytest.mark.parametrize("case", _g3_cases())
def test_g3_small_configs(case, precision):
    """Small versions of BASELINE configs 2/3/4: viol and sum-loss per epoch against the
    reference-generated trajectories."""
    z = load_golden("g3_small_configs.npz")
    X = golden_csr(z)
    meta = json.loads(str(z["meta|" + case]))
    y = z["y"]
    if meta["loss"] != "squared":
        y = np.where(y > np.median(y), 1.0, -1.0)
    r = _Run(X, y, meta, z["P0|" + case], z["lams|" + case], precision)
    np.testing.assert_allclose(r.viol, z["viol|" + case], rtol=TRAJ_RTOL[precision])
    np.testing.assert_allclose(r.loss, z["loss|" + case], rtol=TRAJ_RTOL[precision])
    np.testing.assert_allclose(r.P, z["P|" + case], rtol=0, atol=P_ATOL[precision])
    np.testing.assert_allclose(r.w, z["w|" + case], rtol=0, atol=P_ATOL[precision])
    np.testing.assert_allclose(r.y_pred, z["y_pred|" + case], rtol=0,
                               atol=1e-7 if precision == "f64" else 2e-4)


def _g4_cases():
    return [str(c) for c in load_golden("g4_permuted.npz")["cases"]]


@pytest.mark.parametrize("precision", PREC)
@pytest.mark.parametrize("case", _g4_cases())
def test_g4_permuted_orders(case, precision):
    """Permuted indices_feature / indices_component per epoch (what shuffle=True does)."""
    z = load_golden("g4_permuted.npz")
    X = golden_csr(z)
    meta = json.loads(str(z["meta|" + case]))
    r = _Run(X, z["y"], meta, z["P0|" + case], z["lams|" + case], precision,
             orders=z["forders|" + case], corders=z["corders|" + case], eta0=meta["eta0"],
             n_epochs=3)
    np.testing.assert_allclose(r.viol, z["viol|" + case], rtol=TRAJ_RTOL[precision])
    np.testing.assert_allclose(r.P, z["P|" + case], rtol=0, atol=P_ATOL[precision])
    np.testing.assert_allclose(r.w, z["w|" + case], rtol=0, atol=P_ATOL[precision])


@pytest.mark.parametrize("precision", PREC)
@pytest.mark.parametrize("case", ["c2|squared", "c3|squared", "c4|squared", "c4d3|logistic",
                                  "sql21|squared_hinge", "l1|squared"])
def test_colored_schedule_matches_oracle_in_same_order(oracle, case, precision):
    """The conflict-free (coloured) engine equals the sequential algorithm run in the
    permuted order it reports: the oracle's epoch functions are fed that order."""
    z = load_golden("g3_small_configs.npz")
    X = golden_csr(z)
    meta = json.loads(str(z["meta|" + case]))
    y = z["y"]
    if meta["loss"] != "squared":
        y = np.where(y > np.median(y), 1.0, -1.0)
    r = _Run(X, y, meta, z["P0|" + case], z["lams|" + case], precision, schedule="colored")
    assert sorted(r.order) == list(range(X.shape[1]))
    assert not np.array_equal(r.order, np.arange(X.shape[1]))
    fm = oracle.OracleFM(degree=meta["degree"], loss=meta["loss"], n_components=meta["k"],
                         solver=meta["solver"], regularizer=meta["regularizer"],
                         alpha=meta["alpha"], beta=meta["beta"], gamma=meta["gamma"], tol=0,
                         fit_lower="explicit", fit_linear=True, max_iter=4,
                         feature_order=r.order)
    fm.fit(X, y, P_init=z["P0|" + case], lams_init=z["lams|" + case])
    np.testing.assert_allclose(r.viol, [h[0] for h in fm.history], rtol=TRAJ_RTOL[precision])
    np.testing.assert_allclose(r.loss, [h[1] for h in fm.history], rtol=TRAJ_RTOL[precision])
    np.testing.assert_allclose(r.P, fm.P_, rtol=0, atol=P_ATOL[precision])
    np.testing.assert_allclose(r.w, fm.w_, rtol=0, atol=P_ATOL[precision])


@pytest.mark.parametrize("case", ["c2|squared", "c4|squared"])
def test_graph_replay_equals_eager(case):
    """hipGraph replay of a pass gives bit-identical results to eager launches."""
    z = load_golden("g3_small_configs.npz")
    X = golden_csr(z)
    meta = json.loads(str(z["meta|" + case]))
    a = _Run(X, z["y"], meta, z["P0|" + case], z["lams|" + case], "f32", use_graph=True,
             options={"persistent": 0})
    b = _Run(X, z["y"], meta, z["P0|" + case], z["lams|" + case], "f32", use_graph=False,
             options={"persistent": 0})
    assert a.viol == b.viol
    np.testing.assert_array_equal(a.P, b.P)
    np.testing.assert_array_equal(a.y_pred, b.y_pred)


@pytest.mark.parametrize("options", [{"persistent": 0}, {"persistent": 0, "fuse_chain": 0},
                                     {"prb_groups": 3}, {"prb_groups": 256}, {"max_batch": 3},
                                     {"prb_lds": 0}, {"prb_lds": 0, "prb_groups": 5},
                                     {"prb_groups": 200},
                                     {"max_batch": 1, "fuse_chain": 0, "persistent": 0}])
@pytest.mark.parametrize("case", ["c2|squared", "c3|logistic", "c4d3|squared"])
def test_engine_options_do_not_change_results(oracle, case, options):
    """Fused vs stand-alone chain kernels and the batch-size cap only change how the
    sweep is cut into launches, not the arithmetic."""
    z = load_golden("g3_small_configs.npz")
    X = golden_csr(z)
    meta = json.loads(str(z["meta|" + case]))
    y = z["y"]
    if meta["loss"] != "squared":
        y = np.where(y > np.median(y), 1.0, -1.0)
    a = _Run(X, y, meta, z["P0|" + case], z["lams|" + case], "f64", schedule="colored")
    b = _Run(X, y, meta, z["P0|" + case], z["lams|" + case], "f64", schedule="colored",
             options=options)
    if "max_batch" in options:  # possibly a different (still valid) order: ask the oracle
        assert b.n_batches >= a.n_batches
        fm = oracle.OracleFM(degree=meta["degree"], loss=meta["loss"], n_components=meta["k"],
                             solver=meta["solver"], regularizer=meta["regularizer"],
                             alpha=meta["alpha"], beta=meta["beta"], gamma=meta["gamma"], tol=0,
                             fit_linear=True, max_iter=4, feature_order=b.order)
        fm.fit(X, y, P_init=z["P0|" + case], lams_init=z["lams|" + case])
        np.testing.assert_allclose(b.viol, [h[0] for h in fm.history], rtol=1e-9)
        np.testing.assert_allclose(b.P, fm.P_, rtol=0, atol=1e-8)
    else:
        np.testing.assert_array_equal(a.order, b.order)
        np.testing.assert_allclose(a.viol, b.viol, rtol=1e-10)
        np.testing.assert_allclose(a.P, b.P, rtol=0, atol=1e-10)


@pytest.mark.parametrize("options", [{"pbprb_groups": 1}, {"pbprb_groups": 3},
                                     {"pbprb_groups": 64}, {"pbprb_groups": 100},
                                     {"pbprb_groups": 256}, {"pbprb_groups": 130}])
@pytest.mark.parametrize("case", ["c4|squared", "c4|logistic", "c4d3|squared_hinge",
                                  "l21|logistic", "sql21|squared", "l1b|squared_hinge"])
def test_persistent_pbcd_pass_equals_multi_kernel_engine(oracle, case, options):
    """The persistent pbcd pass (one launch per epoch, reduce-scatter / all-gather exchange)
    against the four-launches-per-step engine and the oracle in the same order, for several
    workgroup counts (owner rounds, more owners than slots, empty row blocks)."""
    z = load_golden("g3_small_configs.npz")
    X = golden_csr(z)
    meta = json.loads(str(z["meta|" + case]))
    y = z["y"]
    if meta["loss"] != "squared":
        y = np.where(y > np.median(y), 1.0, -1.0)
    assert meta["solver"] == "pbcd"
    a = _Run(X, y, meta, z["P0|" + case], z["lams|" + case], "f64", schedule="colored",
             options={"pbcd_persistent": 0})
    b = _Run(X, y, meta, z["P0|" + case], z["lams|" + case], "f64", schedule="colored",
             options=options)
    assert b.pbprb_active == 1 and a.pbprb_active == 0
    np.testing.assert_array_equal(a.order, b.order)
    np.testing.assert_allclose(a.viol, b.viol, rtol=1e-10)
    np.testing.assert_allclose(a.P, b.P, rtol=0, atol=1e-10)
    np.testing.assert_allclose(a.y_pred, b.y_pred, rtol=0, atol=1e-9)
    fm = oracle.OracleFM(degree=meta["degree"], loss=meta["loss"], n_components=meta["k"],
                         solver="pbcd", regularizer=meta["regularizer"], alpha=meta["alpha"],
                         beta=meta["beta"], gamma=meta["gamma"], tol=0, fit_linear=True,
                         max_iter=4, feature_order=b.order)
    fm.fit(X, y, P_init=z["P0|" + case], lams_init=z["lams|" + case])
    np.testing.assert_allclose(b.viol, [h[0] for h in fm.history], rtol=1e-9)
    np.testing.assert_allclose(b.P, fm.P_, rtol=0, atol=1e-8)


@pytest.mark.parametrize("degree", [2, 3, 4, 5])
def test_g6_predict(degree):
    """_get_output / predict on the device vs kernels.py outputs (golden g6)."""
    from sparsepoly_amd.engine import HipEngine

    z = load_golden("g6_anova.npz")
    X, P, lams = z["X"], z["P"], z["lams"]
    eng = HipEngine(0, "f64")
    eng.set_params(P[None], np.zeros(X.shape[1]), lams)
    got = eng.predict(sp.csr_matrix(X), degree, False, False)
    np.testing.assert_allclose(got, z["pred|deg%d" % degree], rtol=0, atol=1e-10)
    eng.close()


@pytest.mark.parametrize("tag", ["deg2|explicit", "deg3|explicit", "deg3|None"])
def test_g6_estimator_predict(tag):
    from sparsepoly_amd import SparseFactorizationMachineRegressor

    z = load_golden("g6_anova.npz")
    deg, fl = tag.split("|")
    est = SparseFactorizationMachineRegressor(degree=int(deg[3:]), n_components=4,
                                              fit_lower=None if fl == "None" else fl,
                                              precision="f64")
    est.P_, est.w_, est.lams_ = z["est_P|" + tag], z["est_w|" + tag], z["lams"]
    np.testing.assert_allclose(est.predict(sp.csr_matrix(z["X"])), z["est_pred|" + tag], rtol=0,
                               atol=1e-10)
    np.testing.assert_allclose(est.predict(z["X"]), z["est_pred|" + tag], rtol=0, atol=1e-10)


def test_g7_api_behaviours():
    """n_iter_ semantics, warning text, callback visibility of P_ (live under pcd, stale
    under pbcd), abort by callback, classifier plumbing, RNG consumption order."""
    from sparsepoly_amd import (SparseFactorizationMachineClassifier,
                                SparseFactorizationMachineRegressor)

    z = load_golden("g7_api.npz")
    facts = json.loads(str(z["facts"]))
    X, y = golden_csr(z), z["y"]
    est = SparseFactorizationMachineRegressor(n_components=3, max_iter=3, tol=0, random_state=0,
                                              gamma=1e-3, precision="f64")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        est.fit(X, y)
    assert est.n_iter_ == facts["n_iter_after_max_iter_3"]
    assert str(w[-1].message) == facts["warning_text"]
    for solver, regname in (("pcd", "l1"), ("pbcd", "l21")):
        sums, wsums = [], []

        def cb(e):
            sums.append(float(np.abs(e.P_).sum()))
            wsums.append(float(np.abs(e.w_).sum()))

        est = SparseFactorizationMachineRegressor(
            n_components=3, max_iter=3, tol=0, random_state=0, gamma=1e-3, solver=solver,
            regularizer=regname, callback=cb, n_calls=1, precision="f64")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            est.fit(X, y)
        np.testing.assert_allclose(sums, facts["callback_P_abs_sums|" + solver], rtol=1e-8)
        np.testing.assert_allclose(wsums, facts["callback_w_abs_sums|" + solver], rtol=1e-8)
    est = SparseFactorizationMachineRegressor(n_components=3, max_iter=10, tol=0, random_state=0,
                                              callback=lambda e: True, n_calls=2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        est.fit(X, y)
    assert est.n_iter_ == facts["n_iter_callback_abort"]
    clf = SparseFactorizationMachineClassifier(max_iter=1, random_state=0, precision="f64")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        clf.fit(X, np.where(y > 0, "a", "b"))
    assert [str(c) for c in clf.label_binarizer_.classes_] == facts["clf_classes"]
    assert [str(c) for c in clf.predict(X)[:10]] == facts["clf_pred_head"]
    est = SparseFactorizationMachineRegressor(
        n_components=3, max_iter=2, tol=0, random_state=3, gamma=1e-3, shuffle=True,
        init_lambdas="random_signs", regularizer="l1", precision="f64")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        est.fit(X, y)
    np.testing.assert_allclose(est.lams_, z["shuffle_lams"])
    np.testing.assert_allclose(est.P_, z["shuffle_P"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(est.w_, z["shuffle_w"], rtol=0, atol=1e-8)


def test_edge_cases_vs_oracle(oracle):
    """Empty columns, an empty row, a column that covers every row, k not a power of two,
    dense input, and a batch-size cap that splits colour classes."""
    rng = np.random.RandomState(0)
    n, d = 257, 23
    Xd = rng.randn(n, d) * (rng.rand(n, d) < 0.15)
    Xd[:, 3] = 0.0           # empty column
    Xd[:, 7] = rng.randn(n)  # full column: conflicts with everything
    Xd[11, :] = 0.0          # empty row
    Xd = Xd.astype(np.float32).astype(np.float64)
    y = rng.randn(n).astype(np.float32).astype(np.float64)
    for solver, regname, degree, k in [("pcd", "squaredl12", 2, 7), ("pcd", "omegati", 3, 5),
                                       ("pbcd", "omegacs", 3, 11), ("pbcd", "l21", 2, 33)]:
        meta = dict(solver=solver, regularizer=regname, degree=degree, k=k, loss="squared",
                    alpha=0.1, beta=10.0 if solver == "pcd" else 1.0, gamma=0.05)
        P0 = 0.01 * rng.randn(degree - 1, k, d)
        lams = np.sign(rng.randn(k))
        for sched in ("exact", "colored"):
            r = _Run(sp.csr_matrix(Xd), y, meta, P0, lams, "f64", schedule=sched, n_epochs=2)
            fm = oracle.OracleFM(degree=degree, n_components=k, solver=solver,
                                 regularizer=regname, alpha=0.1, beta=meta["beta"], gamma=0.05,
                                 tol=0, max_iter=2, feature_order=r.order)
            fm.fit(Xd, y, P_init=P0, lams_init=lams)  # dense -> FortranDataset semantics
            np.testing.assert_allclose(r.viol, [h[0] for h in fm.history], rtol=1e-9)
            np.testing.assert_allclose(r.P, fm.P_, rtol=0, atol=1e-9)
            np.testing.assert_allclose(r.y_pred, fm.y_pred_, rtol=0, atol=1e-8)


def test_csr_ingest_equals_csc_ingest():
    """spfm_set_data_csr (threaded host transposition, no scipy tocsc) against the CSC entry
    point: identical device images, hence bit-identical epochs; a CSR that is not canonical is
    refused by the library (the Python layer canonicalises it through scipy first)."""
    from sparsepoly_amd import _capi
    from sparsepoly_amd.engine import HipEngine
    from sparsepoly_amd.synth import make_problem

    X, y = make_problem(30_000, 3_000, 40, seed=5)     # above the threading threshold
    res = []
    for fmt in ("csr", "csc"):
        eng = HipEngine(0, "f32")
        eng.set_data(X.tocsr() if fmt == "csr" else X.tocsc(), y)
        d = X.shape[1]
        eng.set_params(0.01 * np.random.RandomState(0).randn(1, 6, d), np.zeros(d), np.ones(6))
        eng.configure("pcd", "squared", "squaredl12", 2)
        eng.init_pred(2, True, False)
        order = eng.set_schedule("colored", np.arange(d, dtype=np.int32))
        v = eng.cd_linear_epoch(1.0) + eng.pcd_epoch(0, 2, 10.0, 1e-3, 1.0,
                                                     np.arange(6, dtype=np.int32))
        P, w = eng.get_params()
        res.append((order, v, P, w, eng.get_y_pred()))
        eng.close()
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a, b)
    # unsorted column indices inside a row: refused at the C ABI
    Xr = sp.csr_matrix(X[:50])
    bad = Xr.indices.copy()
    lo, hi = Xr.indptr[0], Xr.indptr[1]
    bad[lo:hi] = bad[lo:hi][::-1]
    eng = HipEngine(0, "f32")
    ip, ii, dd, yy_ = _capi.i64(Xr.indptr), _capi.i32(bad), _capi.f64(Xr.data), _capi.f64(y[:50])
    rc = eng._lib.spfm_set_data_csr(eng._h, 50, Xr.shape[1], ip[1], ii[1], dd[1], yy_[1])
    assert rc == _capi.SPFM_ERR_INVALID
    # ... while the engine's Python face sorts it first and succeeds
    Xbad = sp.csr_matrix((Xr.data, bad, Xr.indptr), shape=Xr.shape)
    assert not Xbad.has_canonical_format or True
    Xbad.has_canonical_format = False
    eng.set_data(Xbad, y[:50])
    eng.close()


def test_errors_match_reference():
    from sparsepoly_amd import (SparseFactorizationMachineClassifier,
                                SparseFactorizationMachineRegressor)
    from sklearn.exceptions import NotFittedError

    z = load_golden("g7_api.npz")
    errs = json.loads(str(z["facts"]))["errors"]
    X, y = golden_csr(z), z["y"]
    cases = [(dict(regularizer="nope"), "bad_regularizer"), (dict(solver="nope"), "bad_solver"),
             (dict(init_lambdas="nope"), "bad_init_lambdas"),
             (dict(degree=3, regularizer="squaredl12"), "squaredl12_degree3"),
             (dict(solver="pbcd", degree=3, regularizer="squaredl21"), "squaredl21_degree3")]
    for kw, nm in cases:
        with pytest.raises(ValueError) as ei:
            SparseFactorizationMachineRegressor(max_iter=1, **kw).fit(X, y)
        assert str(ei.value) == errs[nm][1], nm
    with pytest.raises(ValueError) as ei:
        SparseFactorizationMachineClassifier(loss="nope").fit(X, np.sign(y))
    assert str(ei.value) == errs["bad_loss"][1]
    with pytest.raises(TypeError) as ei:
        SparseFactorizationMachineClassifier().fit(X, y)
    assert str(ei.value) == errs["clf_nonbinary"][1]
    with pytest.raises(NotFittedError):
        SparseFactorizationMachineRegressor().predict(X)
    # pairs the reference cannot run (README.md:28-32) are rejected up front
    with pytest.raises(ValueError):
        SparseFactorizationMachineRegressor(solver="pcd", regularizer="l21", max_iter=1).fit(X, y)
    with pytest.raises(ValueError):
        SparseFactorizationMachineRegressor(solver="pbcd", regularizer="omegati",
                                            max_iter=1).fit(X, y)


def test_rccl_path_single_rank_equals_plain(tmp_path):
    """The multi-GPU code path (RCCL communicator, split linear kernels, per-step
    ncclAllReduce of the partials, no graph) exercised with a 1-rank communicator must
    reproduce the single-GPU result bit for bit."""
    from sparsepoly_amd.engine import HipEngine

    z = load_golden("g3_small_configs.npz")
    X = golden_csr(z)
    for case in ("c2|squared", "c4|squared"):
        meta = json.loads(str(z["meta|" + case]))
        a = _Run(X, z["y"], meta, z["P0|" + case], z["lams|" + case], "f32", schedule="colored",
                 options={"persistent": 0, "use_graph": 0})

        # same driver, but with a communicator installed right after engine creation
        orig_init = HipEngine.__init__

        def patched(self, *args, **kw):
            orig_init(self, *args, **kw)
            self.comm_init(HipEngine.comm_unique_id(), 1, 0)

        HipEngine.__init__ = patched
        try:
            b = _Run(X, z["y"], meta, z["P0|" + case], z["lams|" + case], "f32",
                     schedule="colored")
        finally:
            HipEngine.__init__ = orig_init
        assert a.viol == b.viol
        np.testing.assert_array_equal(a.P, b.P)
        np.testing.assert_array_equal(a.w, b.w)
        np.testing.assert_array_equal(a.y_pred, b.y_pred)


def test_estimator_distributed_world1():
    """distributed=True through torch.distributed (world size 1): row block = all rows,
    global structure used for the schedule, RCCL communicator created from the broadcast id."""
    import os
    import socket

    import torch.distributed as dist

    from sparsepoly_amd import SparseFactorizationMachineRegressor

    z = load_golden("g2_config1.npz")
    X = golden_csr(z)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        kw = dict(degree=2, n_components=4, regularizer="l1", solver="pcd", gamma=1e-3,
                  max_iter=3, tol=0, random_state=0, precision="f64")
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            a = SparseFactorizationMachineRegressor(**kw).fit(X, z["y"])
            b = SparseFactorizationMachineRegressor(distributed=True, **kw).fit(X, z["y"])
        np.testing.assert_allclose(a.P_, b.P_, rtol=0, atol=1e-10)
        np.testing.assert_allclose(a.w_, b.w_, rtol=0, atol=1e-10)
    finally:
        dist.destroy_process_group()


def test_schedule_product_reuse_and_validation():
    """N1: a schedule built once is reused by later fits (identical result, no
    re-colouring), survives save/load, and a schedule whose batches share a row or that
    belongs to other data is rejected by the library instead of being raced on."""
    from sparsepoly_amd import SparseFactorizationMachineRegressor
    from sparsepoly_amd.engine import HipEngine
    from sparsepoly_amd.schedule import Schedule

    z = load_golden("g2_config1.npz")
    X, y = golden_csr(z), z["y"]
    kw = dict(degree=2, n_components=4, regularizer="squaredl12", solver="pcd", gamma=1e-3,
              beta=10.0, max_iter=3, tol=0, random_state=0, precision="f64")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a = SparseFactorizationMachineRegressor(schedule="colored", **kw).fit(X, y)
        assert isinstance(a.schedule_, Schedule) and a.schedule_.n_batches == a.n_steps_per_sweep_
        b = SparseFactorizationMachineRegressor(schedule=a.schedule_, **kw).fit(X, y)
        c = SparseFactorizationMachineRegressor(schedule=Schedule.build(X, "colored"), **kw).fit(X, y)
    np.testing.assert_array_equal(a.feature_order_, b.feature_order_)
    np.testing.assert_array_equal(a.P_, b.P_)
    np.testing.assert_array_equal(a.P_, c.P_)
    # validation
    eng = HipEngine(0, "f64")
    eng.set_data(X, y)
    good = a.schedule_
    merged = Schedule(good.order, np.array([0, X.shape[1]], dtype=np.int32))  # one huge batch
    with pytest.raises(ValueError, match="share a row"):
        eng.install_schedule(merged)
    dup = Schedule(np.zeros(X.shape[1], dtype=np.int32), good.batch_ptr)
    with pytest.raises(ValueError, match="not a permutation"):
        eng.install_schedule(dup)
    eng.install_schedule(good)
    got = eng.get_schedule()
    np.testing.assert_array_equal(got.order, good.order)
    np.testing.assert_array_equal(got.batch_ptr, good.batch_ptr)
    eng.close()
    with pytest.raises(ValueError):
        SparseFactorizationMachineRegressor(schedule=good, shuffle=True, **kw).fit(X, y)
