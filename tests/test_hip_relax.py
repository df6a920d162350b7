"""Relaxed runs: a schedule of tiny steps -- the reference's own visiting order,
`schedule='exact'`, the estimators' default: 2.6 row-disjoint columns per step on BASELINE config 2
-- is run by cd_linear and the degree-2 / degree-3 pcd passes as merged steps of ~20 consecutive columns.  The few rows two
columns of a merged step share are taken out of the row blocks' parallel sums and replayed, in
column order and with their intermediate state, by every workgroup's chain.  The result is the
sequential sweep's (pcd.py:97-135): checked against the oracle in the SAME order and against the
strict engine (`relax=0`), for the three pcd regularizers, all losses, both storage precisions,
rows in LDS and in global memory, several workgroup counts, a shuffled order, very frequent
features.  Needs a real MI355X."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _run(X, y, loss, reg, precision, options, order, k=4, epochs=2, gamma=1e-3):
    from sparsepoly_amd.engine import HipEngine

    d = X.shape[1]
    eng = HipEngine(0, precision)
    for key, val in options.items():
        eng.set_option(key, val)
    eng.set_data(X, y)
    P0 = 0.05 * np.random.RandomState(1).randn(1, k, d)
    lams = np.where(np.arange(k) % 2 == 0, 1.0, -1.0)
    eng.set_params(P0, np.zeros(d), lams)
    eng.configure("pcd", loss, reg, 2)
    eng.init_pred(2, True, False)
    got = eng.set_schedule("exact", order)
    np.testing.assert_array_equal(got, order)
    ic = np.arange(k, dtype=np.int32)
    viol = [eng.cd_linear_epoch(0.5) + eng.pcd_epoch(0, 2, 10.0, gamma, 1.0, ic)
            for _ in range(epochs)]
    P, w = eng.get_params()
    out = dict(viol=np.array(viol), P=P, w=w, y_pred=eng.get_y_pred(), strict=eng.n_batches,
               relaxed=eng.get_option("relax_steps"), lds=eng.get_option("prb_lds_active"),
               fallbacks=eng.get_option("persistent_fallbacks"), P0=P0, lams=lams)
    eng.close()
    return out


def _oracle(oracle, X, y, loss, reg, order, r, k=4, epochs=2, gamma=1e-3):
    fm = oracle.OracleFM(degree=2, loss=loss, n_components=k, solver="pcd", regularizer=reg,
                         alpha=0.5, beta=10.0, gamma=gamma, tol=0, max_iter=epochs, fit_linear=True,
                         feature_order=order)
    fm.fit(X, y, P_init=r["P0"], lams_init=r["lams"])
    return fm


def _problem(loss, n=10_000, d=2_000, per_row=10, seed=2):
    """BASELINE config 2's conflict density at 1/100 of its size: 50-entry columns over 10 000
    rows -- two columns share a row with probability 0.22, 0.25 shared rows per pair."""
    from sparsepoly_amd.synth import make_problem

    X, y = make_problem(n, d, per_row, seed=seed)
    if loss != "squared":
        y = np.where(y > np.median(y), 1.0, -1.0)
    return sp.csr_matrix(X), y


@pytest.mark.parametrize("precision,options", [("f64", {}), ("f32", {}), ("f32", {"prb_lds": 0}),
                                               ("f64", {"prb_groups": 5}),
                                               ("f32", {"prb_groups": 200})])
@pytest.mark.parametrize("loss,reg", [("squared", "squaredl12"), ("logistic", "omegati"),
                                      ("squared_hinge", "l1"), ("squared", "omegati")])
def test_relaxed_runs_equal_the_sequential_sweep(oracle, loss, reg, precision, options):
    X, y = _problem(loss)
    order = np.arange(X.shape[1], dtype=np.int32)
    r = _run(X, y, loss, reg, precision, options, order)
    assert r["fallbacks"] == 0
    # the strict schedule has ~2.6 columns per step; the relaxed runs are several times longer
    assert r["strict"] > 500 and 0 < r["relaxed"] < 0.4 * r["strict"], (r["strict"], r["relaxed"])
    if precision == "f32" and options.get("prb_lds", 1):
        assert r["lds"] in (1, 2)
    fm = _oracle(oracle, X, y, loss, reg, order, r)
    ref = [h[0] for h in fm.history]
    assert 0.05 < (fm.P_ != 0).mean()
    if precision == "f64":
        np.testing.assert_allclose(r["viol"], ref, rtol=1e-9)
        np.testing.assert_allclose(r["P"], fm.P_, rtol=0, atol=1e-9)
        np.testing.assert_allclose(r["w"], fm.w_, rtol=0, atol=1e-9)
        np.testing.assert_allclose(r["y_pred"], fm.y_pred_, rtol=0, atol=1e-8)
    else:
        np.testing.assert_allclose(r["viol"], ref, rtol=2e-5)
        np.testing.assert_allclose(r["P"], fm.P_, rtol=0, atol=1e-4)
        np.testing.assert_allclose(r["w"], fm.w_, rtol=0, atol=1e-4)


def test_relaxed_runs_equal_the_strict_engine_and_can_be_switched_off():
    X, y = _problem("squared")
    order = np.arange(X.shape[1], dtype=np.int32)
    a = _run(X, y, "squared", "squaredl12", "f64", {}, order)
    b = _run(X, y, "squared", "squaredl12", "f64", {"relax": 0}, order)
    assert a["relaxed"] > 0 and b["relaxed"] == 0 and a["strict"] == b["strict"]
    np.testing.assert_allclose(a["viol"], b["viol"], rtol=1e-11)
    np.testing.assert_allclose(a["P"], b["P"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(a["y_pred"], b["y_pred"], rtol=0, atol=1e-10)
    # float storage: the conflict rows are rounded to float after each of their two updates, as
    # the strict engine's scatter does, so the two stay within a few float ulps of each other
    a = _run(X, y, "squared", "squaredl12", "f32", {}, order)
    b = _run(X, y, "squared", "squaredl12", "f32", {"relax": 0}, order)
    np.testing.assert_allclose(a["viol"], b["viol"], rtol=1e-6)
    np.testing.assert_allclose(a["P"], b["P"], rtol=0, atol=2e-6)


def test_relaxed_runs_shuffled_order_and_frequent_features(oracle):
    """shuffle=True hands a permuted order; a few very frequent features (shared rows with almost
    every other column: the run must close early, rows touched by three columns are never
    admitted) and an empty column."""
    rng = np.random.RandomState(9)
    X, y = _problem("squared", n=6000, d=900, per_row=8, seed=4)
    X = sp.lil_matrix(X)
    X[:, 17] = rng.randn(6000, 1)                       # a dense column
    X[rng.rand(6000) < 0.3, 400] = 1.5                  # a frequent one
    X[:, 5] = 0                                         # an empty one
    X = sp.csr_matrix(X)
    X.eliminate_zeros()
    X.data = X.data.astype(np.float32).astype(np.float64)
    order = rng.permutation(900).astype(np.int32)
    for precision in ("f64", "f32"):
        r = _run(X, y, "squared", "squaredl12", precision, {}, order)
        assert r["relaxed"] > 0 and r["fallbacks"] == 0
        fm = _oracle(oracle, X, y, "squared", "squaredl12", order, r)
        tol = dict(rtol=1e-9) if precision == "f64" else dict(rtol=2e-5)
        np.testing.assert_allclose(r["viol"], [h[0] for h in fm.history], **tol)
        np.testing.assert_allclose(r["P"], fm.P_, rtol=0, atol=1e-9 if precision == "f64" else 1e-4)


def test_estimator_default_schedule_uses_relaxed_runs(oracle):
    """fit() with the defaults (schedule='exact') = the reference's trajectory (g-level parity at
    the estimator boundary), now through merged steps."""
    import warnings

    from sparsepoly_amd import SparseFactorizationMachineRegressor

    X, y = _problem("squared", n=8000, d=800, per_row=6, seed=6)
    kw = dict(degree=2, n_components=3, solver="pcd", regularizer="squaredl12", alpha=0.5,
              beta=10.0, gamma=1e-3, max_iter=3, tol=0, random_state=0)
    est = SparseFactorizationMachineRegressor(precision="f64", device=0, **kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        est.fit(X, y)
    assert est.n_steps_per_sweep_ > 100
    fm = oracle.OracleFM(**kw)
    fm.fit(X, y)
    np.testing.assert_allclose(est.P_, fm.P_, rtol=0, atol=1e-9)
    np.testing.assert_allclose(est.w_, fm.w_, rtol=0, atol=1e-9)


@pytest.mark.parametrize("precision,options", [("f64", {}), ("f32", {}), ("f32", {"prb_lds": 0}),
                                               ("f32", {"prb_lds": 0, "prb_pack": 0}),
                                               ("f64", {"prb_groups": 7})])
@pytest.mark.parametrize("loss", ["squared", "logistic"])
def test_relaxed_runs_degree3_explicit_lower_order(oracle, loss, precision, options):
    """Degree 3 with fit_lower='explicit' (BASELINE configs[2]'s shape): the degree-2 epoch on
    P_[1] and the degree-3 epoch on P_[0] both run the reference order as merged steps -- the
    conflict rows carry two cache values, rows in LDS, in packed 16-byte records or in plain
    arrays -- against the oracle in natural order."""
    from sparsepoly_amd.engine import HipEngine

    X, y = _problem(loss, n=8_000, d=1_200, per_row=8, seed=12)
    d, k = X.shape[1], 3
    order = np.arange(d, dtype=np.int32)
    eng = HipEngine(0, precision)
    for key, val in options.items():
        eng.set_option(key, val)
    eng.set_data(X, y)
    P0 = 0.05 * np.random.RandomState(1).randn(2, k, d)
    lams = np.ones(k)
    eng.set_params(P0, np.zeros(d), lams)
    eng.configure("pcd", loss, "omegati", 3)
    eng.init_pred(3, True, True)
    eng.set_schedule("exact", order)
    ic = np.arange(k, dtype=np.int32)
    viol = []
    for _ in range(2):
        v = eng.cd_linear_epoch(0.5)
        v += eng.pcd_epoch(1, 2, 50.0, 1e-3, 1.0, ic)
        v += eng.pcd_epoch(0, 3, 50.0, 1e-3, 1.0, ic)
        viol.append(v)
    P, w = eng.get_params()
    strict, merged = eng.n_batches, eng.get_option("relax_steps")
    packed = eng.get_option("prb_pack_active")
    assert eng.get_option("persistent_fallbacks") == 0
    eng.close()
    assert 0 < merged < 0.4 * strict, (strict, merged)
    if precision == "f32" and options.get("prb_lds", 1) == 0:
        assert packed == (0 if options.get("prb_pack", 1) == 0 else 1)
    fm = oracle.OracleFM(degree=3, loss=loss, n_components=k, solver="pcd", regularizer="omegati",
                         alpha=0.5, beta=50.0, gamma=1e-3, tol=0, max_iter=2, fit_linear=True,
                         fit_lower="explicit", feature_order=order)
    fm.fit(X, y, P_init=P0, lams_init=lams)
    ref = [h[0] for h in fm.history]
    if precision == "f64":
        np.testing.assert_allclose(viol, ref, rtol=1e-9)
        np.testing.assert_allclose(P, fm.P_, rtol=0, atol=1e-9)
        np.testing.assert_allclose(w, fm.w_, rtol=0, atol=1e-9)
    else:
        np.testing.assert_allclose(viol, ref, rtol=5e-5)
        np.testing.assert_allclose(P, fm.P_, rtol=0, atol=2e-4)
        np.testing.assert_allclose(w, fm.w_, rtol=0, atol=2e-4)


# ---- round 4: relaxed runs for pbcd (pbcd_prb_kernel CR; degree 2, k <= 30, one GPU)
def _run_pbcd(X, y, loss, reg, precision, options, order, k=5, epochs=2, gamma=1e-3):
    from sparsepoly_amd.engine import HipEngine

    d = X.shape[1]
    eng = HipEngine(0, precision)
    for key, val in options.items():
        eng.set_option(key, val)
    eng.set_data(X, y)
    P0 = 0.05 * np.random.RandomState(1).randn(1, k, d)
    lams = np.where(np.arange(k) % 2 == 0, 1.0, -1.0)
    eng.set_params(P0, np.zeros(d), lams)
    eng.configure("pbcd", loss, reg, 2)
    eng.init_pred(2, True, False)
    got = eng.set_schedule("exact", order)
    np.testing.assert_array_equal(got, order)
    viol = [eng.cd_linear_epoch(0.5) + eng.pbcd_epoch(0, 2, 1.0, gamma, 1.0) for _ in range(epochs)]
    P, w = eng.get_params()
    out = dict(viol=np.array(viol), P=P, w=w, y_pred=eng.get_y_pred(), strict=eng.n_batches,
               relaxed=eng.get_option("relax_steps"), active=eng.get_option("pb_relax_active"),
               pbprb=eng.get_option("pbprb_active"),
               fallbacks=eng.get_option("persistent_fallbacks"), P0=P0, lams=lams)
    eng.close()
    return out


def _oracle_pbcd(oracle, X, y, loss, reg, order, r, k=5, epochs=2, gamma=1e-3):
    fm = oracle.OracleFM(degree=2, loss=loss, n_components=k, solver="pbcd", regularizer=reg,
                         alpha=0.5, beta=1.0, gamma=gamma, tol=0, max_iter=epochs, fit_linear=True,
                         feature_order=order)
    fm.fit(X, y, P_init=r["P0"], lams_init=r["lams"])
    return fm


@pytest.mark.parametrize("precision,options", [("f64", {}), ("f32", {}),
                                               ("f64", {"pbprb_groups": 5}),
                                               ("f32", {"pbprb_groups": 64}),
                                               ("f32", {"pbprb_groups": 256})])
@pytest.mark.parametrize("loss,reg", [("squared", "omegacs"), ("logistic", "squaredl21"),
                                      ("squared_hinge", "l21"), ("squared", "l1")])
def test_relaxed_pbcd_runs_equal_the_sequential_sweep(oracle, loss, reg, precision, options):
    """pbcd in the reference's own order (pbcd.py:99,110-146 with indices_feature = arange):
    merged steps whose conflict rows every workgroup replays, against the oracle in that order."""
    X, y = _problem(loss)
    order = np.arange(X.shape[1], dtype=np.int32)
    r = _run_pbcd(X, y, loss, reg, precision, options, order)
    assert r["fallbacks"] == 0 and r["pbprb"] == 1 and r["active"] == 1
    assert r["strict"] > 500 and 0 < r["relaxed"] < 0.4 * r["strict"], (r["strict"], r["relaxed"])
    fm = _oracle_pbcd(oracle, X, y, loss, reg, order, r)
    ref = [h[0] for h in fm.history]
    assert 0.05 < (fm.P_ != 0).mean()
    if precision == "f64":
        np.testing.assert_allclose(r["viol"], ref, rtol=1e-9)
        np.testing.assert_allclose(r["P"], fm.P_, rtol=0, atol=1e-9)
        np.testing.assert_allclose(r["w"], fm.w_, rtol=0, atol=1e-9)
        np.testing.assert_allclose(r["y_pred"], fm.y_pred_, rtol=0, atol=1e-8)
    else:
        np.testing.assert_allclose(r["viol"], ref, rtol=2e-5)
        np.testing.assert_allclose(r["P"], fm.P_, rtol=0, atol=1e-4)
        np.testing.assert_allclose(r["w"], fm.w_, rtol=0, atol=1e-4)


def test_relaxed_pbcd_runs_equal_the_strict_engine_and_can_be_switched_off():
    X, y = _problem("squared")
    order = np.arange(X.shape[1], dtype=np.int32)
    a = _run_pbcd(X, y, "squared", "omegacs", "f64", {}, order, k=30)
    b = _run_pbcd(X, y, "squared", "omegacs", "f64", {"relax": 0}, order, k=30)
    assert a["relaxed"] > 0 and a["active"] == 1 and b["relaxed"] == 0 and b["active"] == 0
    assert a["strict"] == b["strict"]
    np.testing.assert_allclose(a["viol"], b["viol"], rtol=1e-11)
    np.testing.assert_allclose(a["P"], b["P"], rtol=0, atol=1e-11)
    np.testing.assert_allclose(a["y_pred"], b["y_pred"], rtol=0, atol=1e-10)
    a = _run_pbcd(X, y, "squared", "omegacs", "f32", {}, order, k=30)
    b = _run_pbcd(X, y, "squared", "omegacs", "f32", {"relax": 0}, order, k=30)
    np.testing.assert_allclose(a["viol"], b["viol"], rtol=1e-6)
    np.testing.assert_allclose(a["P"], b["P"], rtol=0, atol=2e-6)
    # more than 30 components / degree 3: strict steps (the relaxed pass is built for the
    # 32-lane groups and one cache value per row)
    c = _run_pbcd(X, y, "squared", "omegacs", "f64", {}, order, k=33)
    assert c["active"] == 0 and c["pbprb"] == 1


def test_relaxed_pbcd_shuffled_order_and_frequent_features(oracle):
    rng = np.random.RandomState(7)
    X, y = _problem("squared", n=6000, d=900, per_row=9, seed=5)
    X = sp.lil_matrix(X)
    for j, dens in ((3, 0.5), (400, 0.2)):  # very frequent features
        rows = np.flatnonzero(rng.rand(X.shape[0]) < dens)
        X[rows, j] = rng.randn(rows.size)
    X[:, 17] = 0  # an empty column
    X = sp.csr_matrix(X)
    X.eliminate_zeros()
    order = rng.permutation(X.shape[1]).astype(np.int32)
    r = _run_pbcd(X, y, "squared", "omegacs", "f64", {}, order)
    assert r["fallbacks"] == 0
    fm = _oracle_pbcd(oracle, X, y, "squared", "omegacs", order, r)
    np.testing.assert_allclose(r["viol"], [h[0] for h in fm.history], rtol=1e-9)
    np.testing.assert_allclose(r["P"], fm.P_, rtol=0, atol=1e-9)
    np.testing.assert_allclose(r["y_pred"], fm.y_pred_, rtol=0, atol=1e-8)
