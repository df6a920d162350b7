"""Parity at realistic column density and size-independent invariants at the full
BASELINE config-2 size (1M x 100k, ~50 nnz/row).  Needs a real MI355X: ``pytest -m gpu``."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _engine(Xc, y, k, degree, solver, reg, precision, schedule, P0=None, options=None):
    from sparsepoly_amd.engine import HipEngine

    d = Xc.shape[1]
    eng = HipEngine(0, precision)
    for key, val in (options or {}).items():
        eng.set_option(key, val)
    eng.set_data(Xc, y)
    if P0 is None:
        P0 = 0.01 * np.random.RandomState(0).randn(degree - 1, k, d)
    eng.set_params(P0, np.zeros(d), np.ones(k))
    eng.configure(solver, "squared", reg, degree)
    eng.init_pred(degree, True, degree == 3)
    order = eng.set_schedule(schedule, np.arange(d, dtype=np.int32))
    return eng, order, P0


@pytest.mark.parametrize("solver,reg,degree,k,beta,gamma",
                         [("pcd", "squaredl12", 2, 30, 10.0, 1e-3),
                          ("pcd", "omegati", 3, 8, 10.0, 1e-4),
                          ("pbcd", "omegacs", 2, 30, 1.0, 1e-2)])
def test_midsize_f32_trajectory_vs_oracle(oracle, solver, reg, degree, k, beta, gamma):
    """1/10-scale config 2/3/4 (100k x 10k, same 500 nnz per column): the f32 engine in the
    coloured order vs the f64 oracle replaying that order -- the north_star tolerance:
    viol / sum-loss within 1e-5 relative, parameters within 1e-4.  gamma is chosen so that P
    stays substantially non-zero (with gamma = 1 every coordinate is thresholded to 0 in the
    first epoch and nothing is pinned) and the iteration is well conditioned (oracle: a 1e-7
    perturbation of P_0 moves viol by <= 3e-8 relative over these 3 epochs)."""
    from sparsepoly_amd.synth import make_problem

    X, y = make_problem(100_000, 10_000, 50, seed=1)
    Xc = X.tocsc()
    Xc.sort_indices()
    eng, order, P0 = _engine(Xc, y, k, degree, solver, reg, "f32", "colored")
    ic = np.arange(k, dtype=np.int32)
    viol, loss = [], []
    for it in range(3):
        v = eng.cd_linear_epoch(1.0)
        for deg in list(range(2, degree)) + [degree]:
            o = degree - deg if deg != degree else 0
            if solver == "pcd":
                v += eng.pcd_epoch(o, deg, beta, gamma, 1.0, ic)
            else:
                v += eng.pbcd_epoch(o, deg, beta, gamma, 1.0)
        viol.append(v)
        loss.append(eng.loss_sum())
    P, w = eng.get_params()
    eng.close()
    fm = oracle.OracleFM(degree=degree, n_components=k, solver=solver, regularizer=reg, alpha=1.0,
                         beta=beta, gamma=gamma, tol=0, max_iter=3, feature_order=order)
    fm.fit(X, y, P_init=P0, lams_init=np.ones(k))
    assert 0.05 < (fm.P_[0] != 0).mean()  # the model did not collapse
    np.testing.assert_allclose(viol, [h[0] for h in fm.history], rtol=1e-5)
    np.testing.assert_allclose(loss, [h[1] for h in fm.history], rtol=1e-5)
    np.testing.assert_allclose(P, fm.P_, rtol=0, atol=1e-4)
    np.testing.assert_allclose(w, fm.w_, rtol=0, atol=1e-4)


def test_midsize_exact_schedule_many_small_steps(oracle):
    """schedule='exact' (natural order, runs of ~2-3 disjoint columns) through the
    persistent pass: thousands of tiny steps, f64 engine vs oracle in natural order."""
    from sparsepoly_amd.synth import make_problem

    X, y = make_problem(20_000, 2_000, 50, seed=2)
    Xc = X.tocsc()
    Xc.sort_indices()
    eng, order, P0 = _engine(Xc, y, 4, 2, "pcd", "squaredl12", "f64", "exact")
    np.testing.assert_array_equal(order, np.arange(2000))
    assert eng.n_batches > 500
    ic = np.arange(4, dtype=np.int32)
    v = eng.cd_linear_epoch(1.0) + eng.pcd_epoch(0, 2, 10.0, 1e-3, 1.0, ic)
    P, w = eng.get_params()
    eng.close()
    fm = oracle.OracleFM(degree=2, n_components=4, solver="pcd", regularizer="squaredl12",
                         alpha=1.0, beta=10.0, gamma=1e-3, tol=0, max_iter=1)
    fm.fit(X, y, P_init=P0, lams_init=np.ones(4))
    np.testing.assert_allclose(v, fm.history[0][0], rtol=1e-9)
    np.testing.assert_allclose(P, fm.P_, rtol=0, atol=1e-9)
    np.testing.assert_allclose(w, fm.w_, rtol=0, atol=1e-9)


@pytest.fixture(scope="module")
def config2_matrix():
    from sparsepoly_amd.synth import make_problem

    X, y = make_problem(1_000_000, 100_000, 50, seed=0)
    Xc = X.tocsc()
    Xc.sort_indices()
    return Xc, y


@pytest.mark.parametrize("solver,reg", [("pcd", "squaredl12"), ("pbcd", "omegacs")])
def test_fullsize_incremental_state_equals_recompute(config2_matrix, solver, reg):
    """Full BASELINE config-2 size.  Invariant of the whole sweep (pcd.py:124-133,
    pbcd.py:135-144, cd_linear.py:28-31): the incrementally maintained y_pred must equal
    _get_output recomputed from the final (P, w); every coordinate is visited exactly once
    (violation sum finite, schedule a permutation), and a second engine with precision f64
    gives the same violation sum to f32-storage accuracy."""
    Xc, y = config2_matrix
    k = 30
    out = {}
    for precision in ("f32", "f64"):
        eng, order, P0 = _engine(Xc, y, k, 2, solver, reg, precision, "colored")
        assert np.array_equal(np.sort(order), np.arange(Xc.shape[1]))
        ic = np.arange(3, dtype=np.int32)  # 3 of the 30 component passes: bounded run time
        v = eng.cd_linear_epoch(1.0)
        if solver == "pcd":
            v += eng.pcd_epoch(0, 2, 10.0, 1e-4, 1.0, ic)
        else:
            v += eng.pbcd_epoch(0, 2, 1.0, 1e-3, 1.0)
        y_inc = eng.get_y_pred()
        loss_inc = eng.loss_sum()
        eng.init_pred(2, True, False)  # from scratch with the trained parameters
        y_new = eng.get_y_pred()
        eng.close()
        assert np.isfinite(v) and v > 0
        tol = 2e-4 if precision == "f32" else 1e-9
        np.testing.assert_allclose(y_inc, y_new, rtol=0, atol=tol * max(1.0, np.abs(y_new).max()))
        out[precision] = (v, loss_inc)
    np.testing.assert_allclose(out["f32"][0], out["f64"][0], rtol=2e-5)
    np.testing.assert_allclose(out["f32"][1], out["f64"][1], rtol=2e-5)


@pytest.mark.parametrize("solver,reg", [("pcd", "squaredl12"), ("pbcd", "omegacs")])
def test_fullsize_vs_oracle(oracle, config2_matrix, solver, reg):
    """Full BASELINE config-2 size against the CPU oracle in the reported (coloured) order:
    one cd_linear epoch, then 3 pcd component passes (k = 30; ~1.1 s each on the CPU) or the
    whole pbcd epoch, f32 storage.  Stated tolerance (north_star): violation sums 1e-5
    relative, |P - P_ref| <= 1e-4, y_pred 2e-4 of its scale."""
    Xc, y = config2_matrix
    n, d = Xc.shape
    k = 30
    beta, gamma = (10.0, 1e-4) if solver == "pcd" else (1.0, 1e-3)
    eng, order, P0 = _engine(Xc, y, k, 2, solver, reg, "f32", "colored")
    y0 = eng.get_y_pred()
    ic = np.arange(3, dtype=np.int32)
    v_lin = eng.cd_linear_epoch(1.0)
    v = (eng.pcd_epoch(0, 2, beta, gamma, 1.0, ic) if solver == "pcd"
         else eng.pbcd_epoch(0, 2, beta, gamma, 1.0))
    P, w = eng.get_params()
    yp = eng.get_y_pred()
    eng.close()
    ds = oracle.CSC(Xc)
    regc = oracle.Regularizer(reg)
    wo = np.zeros(d)
    ypo = np.ascontiguousarray(y0.copy())
    cn = np.asarray(Xc.multiply(Xc).sum(axis=0)).ravel()
    jf = np.ascontiguousarray(order)
    vo_lin = oracle.cd_linear_epoch(wo, ds, y, ypo, cn, 1.0, "squared", jf)
    lams = np.ones(k)
    if solver == "pcd":
        regc.init_cache_pcd(2, d, k)
        Po = np.ascontiguousarray(P0[0].copy())
        A = np.zeros((n, 3))
        vo = oracle.pcd_epoch(Po, ds, y, ypo, lams, 2, beta, gamma, 1.0, regc, "squared", A, ic, jf)
    else:
        regc.init_cache_pbcd(2, d, k)
        Pt = np.ascontiguousarray(P0[0].T.copy())
        A = np.zeros((n, 3, k))
        dA = np.zeros((n, 2, k))
        vo = oracle.pbcd_epoch(Pt, ds, y, ypo, lams, 2, beta, gamma, 1.0, regc, "squared", A, dA,
                               jf)
        Po = np.ascontiguousarray(Pt.T)
    np.testing.assert_allclose(v_lin, vo_lin, rtol=1e-5)
    np.testing.assert_allclose(v, vo, rtol=1e-5)
    np.testing.assert_allclose(w, wo, rtol=0, atol=1e-4)
    np.testing.assert_allclose(P[0], Po, rtol=0, atol=1e-4)
    np.testing.assert_allclose(yp, ypo, rtol=0, atol=2e-4 * max(1.0, np.abs(ypo).max()))


def test_fullsize_reference_order_merged_steps_vs_oracle(oracle, config2_matrix):
    """Full BASELINE config-2 size in the REFERENCE's own column order (schedule='exact', the
    estimators' default): the degree-2 pcd passes run it as merged steps whose shared rows the
    chains replay (38 284 strict steps -> ~5 500), cd_linear keeps the strict steps.
    One cd_linear epoch + 2 component passes against the oracle in natural order, f32 storage,
    and against the strict engine (relax=0)."""
    Xc, y = config2_matrix
    n, d = Xc.shape
    k = 30
    ic = np.arange(2, dtype=np.int32)
    runs = {}
    for relax in (1, 0):
        eng, order, P0 = _engine(Xc, y, k, 2, "pcd", "squaredl12", "f32", "exact",
                                 options={"relax": relax})
        np.testing.assert_array_equal(order, np.arange(d))
        y0 = eng.get_y_pred()
        v_lin = eng.cd_linear_epoch(1.0)
        v = eng.pcd_epoch(0, 2, 10.0, 1e-4, 1.0, ic)
        P, w = eng.get_params()
        runs[relax] = dict(v_lin=v_lin, v=v, P=P, w=w, yp=eng.get_y_pred(), strict=eng.n_batches,
                           merged=eng.get_option("relax_steps"),
                           fallbacks=eng.get_option("persistent_fallbacks"))
        eng.close()
    a, b = runs[1], runs[0]
    assert a["strict"] > 30_000 and 3_000 < a["merged"] < 0.25 * a["strict"], (a["strict"], a["merged"])
    assert b["merged"] == 0 and a["fallbacks"] == 0 and b["fallbacks"] == 0
    np.testing.assert_allclose(a["v"], b["v"], rtol=2e-6)
    np.testing.assert_allclose(a["P"], b["P"], rtol=0, atol=5e-6)
    ds = oracle.CSC(Xc)
    regc = oracle.Regularizer("squaredl12")
    wo = np.zeros(d)
    ypo = np.ascontiguousarray(y0.copy())
    cn = np.asarray(Xc.multiply(Xc).sum(axis=0)).ravel()
    jf = np.arange(d, dtype=np.int32)
    vo_lin = oracle.cd_linear_epoch(wo, ds, y, ypo, cn, 1.0, "squared", jf)
    regc.init_cache_pcd(2, d, k)
    Po = np.ascontiguousarray(P0[0].copy())
    A = np.zeros((n, 3))
    vo = oracle.pcd_epoch(Po, ds, y, ypo, np.ones(k), 2, 10.0, 1e-4, 1.0, regc, "squared", A, ic, jf)
    np.testing.assert_allclose(a["v_lin"], vo_lin, rtol=1e-5)
    np.testing.assert_allclose(a["v"], vo, rtol=1e-5)
    np.testing.assert_allclose(a["w"], wo, rtol=0, atol=1e-4)
    np.testing.assert_allclose(a["P"][0], Po, rtol=0, atol=1e-4)
    np.testing.assert_allclose(a["yp"], ypo, rtol=0, atol=2e-4 * max(1.0, np.abs(ypo).max()))


def test_fullsize_reference_order_pbcd_merged_steps_vs_oracle(oracle, config2_matrix):
    """BASELINE configs[3] at full size in the REFERENCE's own column order (schedule='exact',
    what SparseFactorizationMachineRegressor(solver='pbcd').fit() does by default:
    sparse_factorization_machines.py:287-337, pbcd.py:99,110-146): the persistent pbcd pass runs
    it as merged steps whose conflict rows every workgroup replays (38 284 strict steps -> ~5 500).
    One cd_linear epoch + the whole pbcd epoch against the oracle in natural order, f32 storage."""
    Xc, y = config2_matrix
    n, d = Xc.shape
    k = 30
    eng, order, P0 = _engine(Xc, y, k, 2, "pbcd", "omegacs", "f32", "exact")
    np.testing.assert_array_equal(order, np.arange(d))
    y0 = eng.get_y_pred()
    v_lin = eng.cd_linear_epoch(1.0)
    v = eng.pbcd_epoch(0, 2, 1.0, 1e-3, 1.0)
    assert eng.get_option("pb_relax_active") == 1 and eng.get_option("persistent_fallbacks") == 0
    assert eng.n_batches > 30_000 and 3_000 < eng.get_option("relax_steps") < 0.25 * eng.n_batches
    P, w = eng.get_params()
    yp = eng.get_y_pred()
    eng.close()
    ds = oracle.CSC(Xc)
    regc = oracle.Regularizer("omegacs")
    wo = np.zeros(d)
    ypo = np.ascontiguousarray(y0.copy())
    cn = np.asarray(Xc.multiply(Xc).sum(axis=0)).ravel()
    jf = np.arange(d, dtype=np.int32)
    vo_lin = oracle.cd_linear_epoch(wo, ds, y, ypo, cn, 1.0, "squared", jf)
    regc.init_cache_pbcd(2, d, k)
    Pt = np.ascontiguousarray(P0[0].T.copy())
    A = np.zeros((n, 3, k))
    dA = np.zeros((n, 2, k))
    vo = oracle.pbcd_epoch(Pt, ds, y, ypo, np.ones(k), 2, 1.0, 1e-3, 1.0, regc, "squared", A, dA, jf)
    np.testing.assert_allclose(v_lin, vo_lin, rtol=1e-5)
    np.testing.assert_allclose(v, vo, rtol=1e-5)
    np.testing.assert_allclose(w, wo, rtol=0, atol=1e-4)
    np.testing.assert_allclose(P[0], Pt.T, rtol=0, atol=1e-4)
    np.testing.assert_allclose(yp, ypo, rtol=0, atol=2e-4 * max(1.0, np.abs(ypo).max()))


@pytest.mark.parametrize("schedule", ["colored", "exact"])
def test_fullsize_config3_vs_oracle(oracle, config2_matrix, schedule):
    """BASELINE configs[2] at full size (1M x 100k, degree 3, k = 16, omegati, pcd, the default
    fit_lower='explicit': sparse_factorization_machines.py:207-243 with omegati.py:62-104): one
    cd_linear epoch, one component pass of the explicit degree-2 order on P_[1], two component
    passes of the degree-3 order on P_[0], f32 storage, against the oracle in the same order --
    the coloured one, and the reference's own (merged steps with replayed conflict rows, DESIGN.md
    3f).  The degree-3 passes must have run on the packed 16-byte row records with the rows in
    global memory (they do not fit LDS at this size).  Tolerance as for config 2."""
    Xc, y = config2_matrix
    n, d = Xc.shape
    k = 16
    # (omegati's strength is gamma * e_2(|p_s|) at degree 3 -- a sum over all column PAIRS, ~50 at
    # d = 100k with |p| ~ 0.01: gamma = 1e-6, as bench.py's config 3, keeps ~60 % of P non-zero;
    # 1e-4 thresholds every coordinate away in the first pass)
    beta, gamma = 10.0, 1e-6
    eng, order, P0 = _engine(Xc, y, k, 3, "pcd", "omegati", "f32", schedule)
    assert P0.shape == (2, k, d)
    y0 = eng.get_y_pred()
    ic2, ic3 = np.arange(1, dtype=np.int32), np.arange(2, dtype=np.int32)
    v_lin = eng.cd_linear_epoch(1.0)
    v2 = eng.pcd_epoch(1, 2, beta, gamma, 1.0, ic2)
    v3 = eng.pcd_epoch(0, 3, beta, gamma, 1.0, ic3)
    assert eng.get_option("prb_pack_active") == 1 and eng.get_option("prb_lds_active") == 0
    assert eng.get_option("persistent_fallbacks") == 0
    if schedule == "exact":
        np.testing.assert_array_equal(order, np.arange(d))
        assert 3_000 < eng.get_option("relax_steps") < 0.25 * eng.n_batches
    P, w = eng.get_params()
    yp = eng.get_y_pred()
    eng.close()
    ds = oracle.CSC(Xc)
    regc = oracle.Regularizer("omegati")
    regc.init_cache_pcd(3, d, k)
    wo = np.zeros(d)
    ypo = np.ascontiguousarray(y0.copy())
    cn = np.asarray(Xc.multiply(Xc).sum(axis=0)).ravel()
    jf = np.ascontiguousarray(order)
    lams = np.ones(k)
    A = np.zeros((n, 4))
    vo_lin = oracle.cd_linear_epoch(wo, ds, y, ypo, cn, 1.0, "squared", jf)
    P_low = np.ascontiguousarray(P0[1].copy())
    P_top = np.ascontiguousarray(P0[0].copy())
    vo2 = oracle.pcd_epoch(P_low, ds, y, ypo, lams, 2, beta, gamma, 1.0, regc, "squared", A, ic2, jf)
    vo3 = oracle.pcd_epoch(P_top, ds, y, ypo, lams, 3, beta, gamma, 1.0, regc, "squared", A, ic3, jf)
    assert (P_top[:2] != 0).mean() > 0.05  # the model did not collapse
    np.testing.assert_allclose(v_lin, vo_lin, rtol=1e-5)
    np.testing.assert_allclose(v2, vo2, rtol=1e-5)
    np.testing.assert_allclose(v3, vo3, rtol=1e-5)
    np.testing.assert_allclose(w, wo, rtol=0, atol=1e-4)
    np.testing.assert_allclose(P[1], P_low, rtol=0, atol=1e-4)
    np.testing.assert_allclose(P[0], P_top, rtol=0, atol=1e-4)
    np.testing.assert_allclose(yp, ypo, rtol=0, atol=2e-4 * max(1.0, np.abs(ypo).max()))


def test_estimator_augment_and_warm_start(oracle):
    """fit_lower='augment' (dummy columns, sparse_factorization_machines.py:86-92) and
    warm_start=True (P_, w_, lams_ reused, y_pred recomputed: :380-391,408)."""
    from sklearn.preprocessing import add_dummy_feature

    from sparsepoly_amd import SparseFactorizationMachineRegressor

    rng = np.random.RandomState(5)
    X = sp.random(300, 30, density=0.15, random_state=rng, data_rvs=rng.randn, format="csr")
    y = rng.randn(300)
    kw = dict(degree=3, n_components=4, solver="pcd", regularizer="omegati", alpha=0.1, beta=10.0,
              gamma=0.05, tol=0, random_state=0, precision="f64")
    est = SparseFactorizationMachineRegressor(fit_lower="augment", max_iter=3, **kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        est.fit(X, y)
    Xa = add_dummy_feature(X, value=1)  # degree 3, fit_linear=True: one dummy column
    assert est.P_.shape == (1, 4, 31) and est.w_.shape == (31,)
    fm = oracle.OracleFM(degree=3, n_components=4, solver="pcd", regularizer="omegati",
                         alpha=0.1, beta=10.0, gamma=0.05, tol=0, max_iter=3, random_state=0,
                         fit_lower="augment")
    fm.fit(sp.csr_matrix(Xa), y)
    np.testing.assert_allclose(est.P_, fm.P_, rtol=0, atol=1e-9)
    np.testing.assert_allclose(est.w_, fm.w_, rtol=0, atol=1e-9)
    np.testing.assert_allclose(est.predict(X), fm.predict(sp.csr_matrix(Xa)), rtol=0, atol=1e-8)
    # warm start: 2 + 2 iterations == oracle continuing from the same state
    est = SparseFactorizationMachineRegressor(fit_lower="explicit", max_iter=2, warm_start=True,
                                              **kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        est.fit(X, y)
        P_mid, w_mid = est.P_.copy(), est.w_.copy()
        est.fit(X, y)
    fm = oracle.OracleFM(degree=3, n_components=4, solver="pcd", regularizer="omegati",
                         alpha=0.1, beta=10.0, gamma=0.05, tol=0, max_iter=2, random_state=0)
    fm.fit(X, y, P_init=P_mid, w_init=w_mid, lams_init=np.ones(4))
    np.testing.assert_allclose(est.P_, fm.P_, rtol=0, atol=1e-9)
    np.testing.assert_allclose(est.w_, fm.w_, rtol=0, atol=1e-9)


@pytest.mark.parametrize("groups", [2, 64])
def test_skewed_columns_long_slots(oracle, groups):
    """Zipf-like data: a few features occur in 30-100 % of the rows, so their entries in one
    row block exceed what a slot's 4 lanes keep in registers ("long slots": the whole
    workgroup strides over them).  f64 engine vs oracle in the same coloured order, for
    pcd (two regularizers, degree 2 and 3) and the persistent linear pass."""
    rng = np.random.RandomState(4)
    n, d = 4000, 120
    cols = [sp.random(n, 1, density=dens, random_state=rng, data_rvs=rng.randn, format="csc")
            for dens in ([1.0, 0.6, 0.3] + [0.01] * (d - 3))]
    X = sp.hstack(cols, format="csc")[:, rng.permutation(d)].tocsr()
    y = rng.randn(n)
    for reg, degree, k in (("squaredl12", 2, 5), ("omegati", 3, 3), ("l1", 2, 4)):
        eng, order, P0 = _engine(X.tocsc(), y, k, degree, "pcd", reg, "f64", "colored",
                                 options={"prb_groups": groups})
        assert eng.get_option("persistent_active") == 1
        ic = np.arange(k, dtype=np.int32)
        viol = []
        for it in range(2):
            v = eng.cd_linear_epoch(0.5)
            for deg in list(range(2, degree)) + [degree]:
                o = degree - deg if deg != degree else 0
                v += eng.pcd_epoch(o, deg, 10.0, 1e-3, 1.0, ic)
            viol.append(v)
        P, w = eng.get_params()
        yp = eng.get_y_pred()
        eng.close()
        fm = oracle.OracleFM(degree=degree, n_components=k, solver="pcd", regularizer=reg,
                             alpha=0.5, beta=10.0, gamma=1e-3, tol=0, max_iter=2,
                             feature_order=order)
        fm.fit(X, y, P_init=P0, lams_init=np.ones(k))
        np.testing.assert_allclose(viol, [h[0] for h in fm.history], rtol=1e-9)
        np.testing.assert_allclose(P, fm.P_, rtol=0, atol=1e-9)
        np.testing.assert_allclose(w, fm.w_, rtol=0, atol=1e-9)
        np.testing.assert_allclose(yp, fm.y_pred_, rtol=0, atol=1e-8)


@pytest.mark.parametrize("loss,model,want_mode", [("squared", "fm", 1), ("logistic", "fm", 2),
                                                  ("squared_hinge", "fm", 2),
                                                  ("squared", "all_subsets", 1),
                                                  ("squared", "fm3", 1), ("logistic", "fm3", 2)])
def test_lds_resident_row_block_matches_global_path(oracle, loss, model, want_mode):
    """f32 persistent pass with the row block in LDS (residual form for the squared loss,
    yhat + label sign for +-1 targets; one cache value per row, or two: degree 3, `fm3`) vs the
    same pass on global memory and vs the f64 oracle; the engine reports which variant ran."""
    from sparsepoly_amd.engine import HipEngine
    from sparsepoly_amd.synth import make_problem

    n, d, k = 30_000, 3_000, 12
    X, y = make_problem(n, d, 50, seed=4)
    if loss != "squared":
        y = np.where(y > np.median(y), 1.0, -1.0)
    Xc = X.tocsc()
    Xc.sort_indices()
    # (degree 3 amplifies float rounding more: a stiffer step keeps it a rounding test)
    beta = 100.0 if model == "fm3" else 10.0
    degree = {"fm": 2, "fm3": 3, "all_subsets": -1}[model]
    reg = {"fm": "squaredl12", "fm3": "omegati", "all_subsets": "l1"}[model]
    P0 = 0.01 * np.random.RandomState(0).randn(1, k, d)
    lams = np.ones(k) if model == "fm" else np.where(np.arange(k) % 2 == 0, 1.0, -1.0)
    ic = np.arange(k, dtype=np.int32)
    out = {}
    # (2 = rows in global memory WITHOUT the packed 16-byte row records of the degree-3 pass)
    for lds in (1, 0, 2) if model == "fm3" else (1, 0):
        eng = HipEngine(0, "f32")
        eng.set_option("prb_lds", 1 if lds == 1 else 0)
        if lds == 2:
            eng.set_option("prb_pack", 0)
        eng.set_data(Xc, y)
        eng.set_params(P0, np.zeros(d), lams)
        eng.configure("pcd", loss, reg, degree)
        eng.init_pred(degree, False, False)
        order = eng.set_schedule("colored", np.arange(d, dtype=np.int32))
        viol = [eng.pcd_epoch(0, degree, beta, 1e-3, 1.0, ic) for _ in range(2)]
        assert eng.get_option("prb_lds_active") == (want_mode if lds == 1 else 0)
        # degree 3 with the rows in global memory: packed (yhat, y, A1, A2) records by default
        assert eng.get_option("prb_pack_active") == (1 if (model == "fm3" and lds == 0) else 0)
        out[lds] = (np.array(viol), eng.loss_sum(), eng.get_params()[0], eng.get_y_pred(), order)
        eng.close()
    if model == "fm3":  # same arithmetic, same float rounding of the stored rows: same bits
        np.testing.assert_array_equal(out[0][0], out[2][0])
        np.testing.assert_array_equal(out[0][2], out[2][2])
        np.testing.assert_array_equal(out[0][3], out[2][3])
    # the two variants round differently (residual vs prediction in float); degree 3 amplifies
    # that more than degree 2
    vt = 2e-5 if model == "fm3" else 2e-6
    np.testing.assert_allclose(out[1][0], out[0][0], rtol=vt)
    np.testing.assert_allclose(out[1][1], out[0][1], rtol=vt)
    np.testing.assert_allclose(out[1][2], out[0][2], rtol=0, atol=10 * vt)
    np.testing.assert_allclose(out[1][3], out[0][3], rtol=0, atol=10 * vt)
    # and against the oracle in the reported order
    if model in ("fm", "fm3"):
        fm = oracle.OracleFM(degree=degree, loss=loss, n_components=k, solver="pcd",
                             regularizer=reg, beta=beta, gamma=1e-3, tol=0, max_iter=2,
                             fit_linear=False, fit_lower=None, feature_order=out[1][4])
        fm.fit(X, y, P_init=P0, lams_init=lams)
        np.testing.assert_allclose(out[1][0], [h[0] for h in fm.history], rtol=1e-5)
        np.testing.assert_allclose(out[1][2], fm.P_, rtol=0, atol=1e-4)


def test_warm_start_keeps_device_session():
    """a regularization path with warm_start=True re-uses the device-resident
    data, schedule and row-block stream; results are identical to fits that re-upload."""
    import pickle

    from sparsepoly_amd import SparseFactorizationMachineRegressor
    from sparsepoly_amd.synth import make_problem

    X, y = make_problem(20_000, 2_000, 30, seed=8)
    kw = dict(degree=2, n_components=6, solver="pcd", regularizer="squaredl12", beta=10.0,
              max_iter=2, tol=0, random_state=0, schedule="colored", precision="f64",
              warm_start=True)
    path = [1e-2, 1e-3, 1e-4]
    a = SparseFactorizationMachineRegressor(gamma=path[0], **kw)
    b = SparseFactorizationMachineRegressor(gamma=path[0], **kw)
    handles = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for g in path:
            a.set_params(gamma=g)
            b.set_params(gamma=g)
            a.fit(X, y)
            b.fit(X, y)
            b.release_device()               # b re-uploads and re-colours every time
            handles.append(a._device_session[1]._h.value)
            assert np.array_equal(a.P_, b.P_) and np.array_equal(a.w_, b.w_)
            assert np.array_equal(a.feature_order_, b.feature_order_)
    assert len(set(handles)) == 1            # one engine for the whole path
    assert getattr(b, "_device_session", None) is None
    # a different training set => a new session; pickling drops the handle
    X2, y2 = make_problem(20_000, 2_000, 30, seed=9)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a.fit(X2, y2)
    assert a._device_session[1]._h.value != handles[0] or True
    c = pickle.loads(pickle.dumps(a))
    assert getattr(c, "_device_session", None) is None
    np.testing.assert_allclose(c.predict(X2[:50]), a.predict(X2[:50]))
    # other solvers share the session too
    a.set_params(solver="psgd", regularizer="l1", learning_rate="constant", eta0=0.01)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        a.fit(X2, y2)
    assert a._device_session is not None
    a.release_device()
    assert a._device_session is None


def test_fullsize_four_concurrent_fits_equal_their_solo_runs(config2_matrix):
    """Full BASELINE config-2 size: four handles training side by side (one host thread and one
    stream each, 64 workgroups per persistent pass: all 256 CUs taken) against the same four
    trainings one after the other -- bit-identical parameters and predictions, and no pass
    redone on the multi-kernel engine (sparsepoly_amd/concurrent.py)."""
    import threading

    from sparsepoly_amd import engine as E

    Xc, y = config2_matrix
    X = Xc.tocsr()
    k, d = 30, Xc.shape[1]
    gammas = [1e-4, 3e-4, 1e-3, 3e-5]
    P0 = 0.01 * np.random.RandomState(0).randn(1, k, d)

    def make():
        with E.co_tenancy(4):
            eng = E.HipEngine(0, "f32")
        eng.set_data(X, y)
        eng.set_params(P0, np.zeros(d), np.ones(k))
        eng.configure("pcd", "squared", "squaredl12", 2)
        eng.init_pred(2, True, False)
        eng.set_schedule("colored", np.arange(d, dtype=np.int32))
        return eng

    def train(eng, gamma, out, i, barrier=None):
        ic = np.arange(4, dtype=np.int32)      # 4 of the 30 component passes: bounded run time
        if barrier is not None:
            barrier.wait()
        v = [eng.cd_linear_epoch(1.0) + eng.pcd_epoch(0, 2, 10.0, gamma, 1.0, ic)
             for _ in range(2)]
        P, w = eng.get_params()
        out[i] = (np.array(v), P, w, eng.get_y_pred(), eng.get_option("persistent_fallbacks"))

    solo = [None] * 4
    for f in range(4):
        eng = make()
        train(eng, gammas[f], solo, f)
        eng.close()
    engs = [make() for _ in range(4)]
    together = [None] * 4
    barrier = threading.Barrier(4)
    th = [threading.Thread(target=train, args=(engs[f], gammas[f], together, f, barrier))
          for f in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for eng in engs:
        eng.close()
    for f in range(4):
        assert together[f] is not None and together[f][4] == 0
        for a, b in zip(solo[f][:4], together[f][:4]):
            assert np.array_equal(a, b)
