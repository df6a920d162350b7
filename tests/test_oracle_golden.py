"""Pins the CPU oracle (oracle/spfm_oracle.c + oracle/oracle.py) against fixtures
that were produced by running the reference's own source (oracle/gen_golden.py).
Tolerance: 1e-10 absolute -- the only intended difference
is NumPy's pairwise np.sum vs a plain loop (~1e-16 relative)."""
import json

import numpy as np
import pytest
from conftest import golden_csr, load_golden

TOL = 1e-10


def _cells():
    z = load_golden("g1_reftests.npz")
    return [str(c) for c in z["cells"]]


@pytest.mark.parametrize("cell", _cells())
def test_g1_reference_test_cells(oracle, cell):
    """Replicas of reference tests/test_pcd.py:177-351 and tests/test_pbcd.py:163-342."""
    z = load_golden("g1_reftests.npz")
    solver, regname, deg, mean, loss = cell.split("|")
    degree = int(deg[3:])
    mean = bool(int(mean[4:]))
    X = z["X"]
    y = z["y_deg%d" % degree]
    if loss != "squared":
        y = np.sign(y)
    max_iter = 1 if regname in ("squaredl12", "squaredl21") else 5
    fm = oracle.OracleFM(degree=degree, loss=loss, n_components=5, solver=solver,
                         regularizer=regname, beta=1, gamma=1e-3, mean=mean, tol=1e-3,
                         fit_lower=None, fit_linear=False, max_iter=max_iter, random_state=0)
    fm.fit(X, y)
    np.testing.assert_allclose(fm.P_, z["P|" + cell], rtol=0, atol=TOL)
    assert fm.n_iter_ == int(z["n_iter|" + cell])


@pytest.mark.parametrize("gamma", [1e-3, 1e-2])
def test_g2_config1(oracle, gamma):
    """BASELINE config 1: 1k x 100 CSR, degree 2, k=4, l1, pcd (defaults otherwise)."""
    z = load_golden("g2_config1.npz")
    X = golden_csr(z)
    tag = "gamma%g" % gamma
    fm = oracle.OracleFM(degree=2, n_components=4, regularizer="l1", solver="pcd", gamma=gamma,
                         max_iter=6, tol=1e-9, random_state=0)
    fm.fit(X, z["y"])
    viol = np.array([h[0] for h in fm.history])
    np.testing.assert_allclose(viol, z["viol|" + tag], rtol=1e-11, atol=TOL)
    np.testing.assert_allclose(fm.P_, z["P|" + tag], rtol=0, atol=TOL)
    np.testing.assert_allclose(fm.w_, z["w|" + tag], rtol=0, atol=TOL)
    np.testing.assert_allclose(fm.predict(X), z["pred|" + tag], rtol=0, atol=1e-9)
    assert fm.n_iter_ == int(z["n_iter|" + tag])


def _g3_cases():
    return [str(c) for c in load_golden("g3_small_configs.npz")["cases"]]


@pytest.mark.parametrize("case", _g3_cases())
def test_g3_small_configs(oracle, case):
    """Small versions of BASELINE configs 2/3/4: viol and sum-loss per epoch."""
    z = load_golden("g3_small_configs.npz")
    X = golden_csr(z)
    meta = json.loads(str(z["meta|" + case]))
    y = z["y"]
    if meta["loss"] != "squared":
        y = np.where(y > np.median(y), 1.0, -1.0)
    fm = oracle.OracleFM(degree=meta["degree"], loss=meta["loss"], n_components=meta["k"],
                         solver=meta["solver"], regularizer=meta["regularizer"],
                         alpha=meta["alpha"], beta=meta["beta"], gamma=meta["gamma"], tol=0,
                         fit_lower=meta["fit_lower"], fit_linear=True, max_iter=4)
    fm.fit(X, y, P_init=z["P0|" + case], lams_init=z["lams|" + case])
    viol = np.array([h[0] for h in fm.history])
    loss = np.array([h[1] for h in fm.history])
    np.testing.assert_allclose(viol, z["viol|" + case], rtol=1e-10, atol=TOL)
    np.testing.assert_allclose(loss, z["loss|" + case], rtol=1e-10, atol=TOL)
    np.testing.assert_allclose(fm.P_, z["P|" + case], rtol=0, atol=TOL)
    np.testing.assert_allclose(fm.w_, z["w|" + case], rtol=0, atol=TOL)
    np.testing.assert_allclose(fm.y_pred_, z["y_pred|" + case], rtol=0, atol=1e-9)


def _g4_cases():
    return [str(c) for c in load_golden("g4_permuted.npz")["cases"]]


@pytest.mark.parametrize("case", _g4_cases())
def test_g4_permuted_orders(oracle, case):
    """Direct epoch calls with permuted indices_feature / indices_component
    (reference pcd.py:86-87,97; pbcd.py:99,110) -- pins order injection."""
    z = load_golden("g4_permuted.npz")
    X = golden_csr(z)
    meta = json.loads(str(z["meta|" + case]))
    y = z["y"]
    n, d = X.shape
    k, degree = meta["k"], meta["degree"]
    ds = oracle.CSC(X)
    reg = oracle.Regularizer(meta["regularizer"])
    P_ = np.array(z["P0|" + case])
    lams = np.ascontiguousarray(z["lams|" + case])
    w = np.zeros(d)
    fm = oracle.OracleFM(degree=degree, n_components=k)
    fm.P_, fm.w_, fm.lams_ = P_, w, lams
    fm.fit_linear = False
    y_pred = np.ascontiguousarray(fm._get_output(X))
    col_norm_sq = np.asarray(X.multiply(X).sum(axis=0)).ravel()
    viols = []
    if meta["solver"] == "pcd":
        A = np.zeros((n, degree + 1))
        reg.init_cache_pcd(degree, d, k)
        P = P_
    else:
        A = np.zeros((n, degree + 1, k))
        dA = np.zeros((n, degree, k))
        reg.init_cache_pbcd(degree, d, k)
        P = np.ascontiguousarray(P_.swapaxes(1, 2))
    for it in range(z["forders|" + case].shape[0]):
        jf, ic = z["forders|" + case][it], z["corders|" + case][it]
        v = oracle.cd_linear_epoch(w, ds, y, y_pred, col_norm_sq, meta["alpha"], "squared", jf)
        for deg in list(range(2, degree)) + [degree]:
            o = degree - deg if deg != degree else 0
            if meta["solver"] == "pcd":
                v += oracle.pcd_epoch(P[o], ds, y, y_pred, lams, deg, meta["beta"],
                                      meta["gamma"], meta["eta0"], reg, "squared", A, ic, jf)
            else:
                v += oracle.pbcd_epoch(P[o], ds, y, y_pred, lams, deg, meta["beta"],
                                       meta["gamma"], meta["eta0"], reg, "squared", A, dA, jf)
        viols.append(v)
    if meta["solver"] == "pbcd":
        P_ = np.array(P.swapaxes(1, 2))
    np.testing.assert_allclose(viols, z["viol|" + case], rtol=1e-10, atol=TOL)
    np.testing.assert_allclose(P_, z["P|" + case], rtol=0, atol=TOL)
    np.testing.assert_allclose(w, z["w|" + case], rtol=0, atol=TOL)
    np.testing.assert_allclose(y_pred, z["y_pred|" + case], rtol=0, atol=1e-9)


def _g5_tags(prefix):
    z = load_golden("g5_reg_traces.npz")
    return sorted(k[len("P0|"):] for k in z.files if k.startswith("P0|" + prefix))


@pytest.mark.parametrize("tag", _g5_tags("pcd"))
def test_g5_prox_cd_traces(oracle, tag):
    """regularizer/{l1,squaredl12,omegati}.py prox_cd + cache recurrences, call by call."""
    z = load_golden("g5_reg_traces.npz")
    _, regname, deg = tag.split("|")
    degree = int(deg[3:])
    P = np.array(z["P0|" + tag])
    k, d = P.shape
    reg = oracle.Regularizer(regname)
    reg.init_cache_pcd(degree, d, k)
    q = 0
    for s in range(k):
        reg.compute_cache_pcd(P, degree, s)
        for j in range(d):
            new = reg.prox_cd(float(z["p_in|" + tag][q]), float(z["strength|" + tag][q]),
                              degree, j)
            assert abs(new - z["p_out|" + tag][q]) <= TOL
            if new == 0.0 and regname == "omegati":
                # omegati.py:92,104: sign = 1 if p > 0 else -1; sign * np.maximum(.., 0) -> -0.0
                # (squaredl12.py:57 uses the builtin max(.., 0): int 0 under CPython, so the
                # sign of its zero is interpreter-dependent and not pinned)
                assert np.signbit(new) == np.signbit(z["p_out|" + tag][q])
            P[s, j] = new
            reg.update_cache_pcd(P, degree, s, j)
            if regname != "l1":
                st = reg.state()
                nc = 1 if regname == "squaredl12" else degree + 1
                np.testing.assert_allclose(st["cache"][:nc], z["cache|" + tag][q][:nc],
                                           rtol=1e-12, atol=TOL)
                if regname == "omegati":
                    np.testing.assert_allclose(st["dcache"], z["dcache|" + tag][q],
                                               rtol=1e-12, atol=TOL)
            q += 1
    np.testing.assert_allclose(P, z["P_end|" + tag], rtol=0, atol=TOL)


@pytest.mark.parametrize("tag", _g5_tags("pbcd"))
def test_g5_prox_bcd_traces(oracle, tag):
    """regularizer/{l1,l21,squaredl21,omegacs}.py prox_bcd incl. the 'numerical error'
    fallback branches (omegacs.py:75-76,90-96; squaredl21.py:48-49) via poisoned caches."""
    z = load_golden("g5_reg_traces.npz")
    _, regname, deg, mode = tag.split("|")
    degree = int(deg[3:])
    P = np.array(z["P0|" + tag])
    d, k = P.shape
    reg = oracle.Regularizer(regname)
    reg.init_cache_pbcd(degree, d, k)
    reg.compute_cache_pbcd(P, degree)
    q = 0
    for sweep in range(2):
        for j in range(d):
            if mode == "poison" and sweep == 1 and j % 4 == 1:
                st = reg.state()
                if regname == "omegacs":
                    if j % 8 == 1:
                        st["cache"][degree - 1] = -abs(st["cache"][degree - 1]) - 1e-3
                    st["norms"][j] = st["norms"][j] + 50.0
                else:
                    st["cache"][0] = st["norms"][j] - 1e-3
                oracle.lib().spo_reg_set_state(
                    reg._h, oracle._d(st["cache"]), oracle._d(st["dcache"]),
                    oracle._d(st["abs_p"]), oracle._d(st["norms"]))
            pj = np.array(z["v_in|" + tag][q])
            reg.prox_bcd(pj, float(z["strength|" + tag][q]), degree, j)
            np.testing.assert_allclose(pj, z["v_out|" + tag][q], rtol=0, atol=TOL)
            P[j] = pj
            reg.update_cache_pbcd(P, degree, j)
            if regname in ("squaredl21", "omegacs"):
                st = reg.state()
                nc = 1 if regname == "squaredl21" else degree + 1
                np.testing.assert_allclose(st["cache"][:nc], z["cache|" + tag][q][:nc],
                                           rtol=1e-12, atol=TOL)
                np.testing.assert_allclose(st["norms"], z["norms|" + tag][q], rtol=1e-12,
                                           atol=TOL)
                if regname == "omegacs":
                    np.testing.assert_allclose(st["dcache"], z["dcache|" + tag][q],
                                               rtol=1e-12, atol=TOL)
            q += 1
    np.testing.assert_allclose(P, z["P_end|" + tag], rtol=0, atol=TOL)


@pytest.mark.parametrize("loss", ["squared", "squared_hinge", "logistic"])
def test_g5_losses(oracle, loss):
    """loss.py:13-71 incl. the |z|>18 clamps of Logistic."""
    z = load_golden("g5_reg_traces.npz")
    p = z["loss_p"]
    L = oracle.lib()
    assert L.spo_loss_mu(oracle.LOSSES[loss]) == float(z["mu|" + loss])
    for yv in (-1.0, 1.0, 0.37):
        got = oracle.dloss(loss, p, np.full_like(p, yv))
        np.testing.assert_allclose(got, z["dloss|%s|y%g" % (loss, yv)], rtol=1e-14, atol=0)
        gl = np.array([L.spo_loss(oracle.LOSSES[loss], pi, yv) for pi in p])
        np.testing.assert_allclose(gl, z["loss|%s|y%g" % (loss, yv)], rtol=1e-14, atol=0)


@pytest.mark.parametrize("degree", [2, 3, 4, 5])
def test_g6_anova_kernel(oracle, degree):
    """kernels.py:71-115,140-153 on dense and sparse input + the row-DP evaluator."""
    import scipy.sparse as sp

    z = load_golden("g6_anova.npz")
    X, P, lams = z["X"], z["P"], z["lams"]
    np.testing.assert_allclose(oracle.anova_kernel(X, P, degree), z["K_dense|deg%d" % degree],
                               rtol=0, atol=1e-12)
    np.testing.assert_allclose(oracle.anova_kernel(sp.csr_matrix(X), P, degree),
                               z["K_sparse|deg%d" % degree], rtol=0, atol=1e-12)
    np.testing.assert_allclose(oracle.poly_predict(X, P, lams, degree),
                               z["pred|deg%d" % degree], rtol=0, atol=1e-12)
    out = np.zeros(X.shape[0])
    oracle.anova_predict_dp(sp.csr_matrix(X), P, lams, degree, out)
    np.testing.assert_allclose(out, z["pred|deg%d" % degree], rtol=0, atol=1e-10)


@pytest.mark.parametrize("tag", ["deg2|explicit", "deg3|explicit", "deg3|None"])
def test_g6_get_output(oracle, tag):
    """sparse_factorization_machines.py:437-451 (_get_output)."""
    import scipy.sparse as sp

    z = load_golden("g6_anova.npz")
    deg, fl = tag.split("|")
    fm = oracle.OracleFM(degree=int(deg[3:]), n_components=4,
                         fit_lower=None if fl == "None" else fl)
    fm.P_, fm.w_, fm.lams_ = z["est_P|" + tag], z["est_w|" + tag], z["lams"]
    np.testing.assert_allclose(fm.predict(sp.csr_matrix(z["X"])), z["est_pred|" + tag], rtol=0,
                               atol=1e-12)


def test_g7_api_facts(oracle):
    """n_iter_ is the 0-based index of the last iteration; pbcd callbacks see stale P_."""
    z = load_golden("g7_api.npz")
    facts = json.loads(str(z["facts"]))
    X = golden_csr(z)
    y = z["y"]
    fm = oracle.OracleFM(n_components=3, max_iter=3, tol=0, random_state=0, gamma=1e-3)
    fm.fit(X, y)
    assert fm.n_iter_ == facts["n_iter_after_max_iter_3"] == 2
    for solver, regname in (("pcd", "l1"), ("pbcd", "l21")):
        sums, wsums = [], []

        def cb(e):
            sums.append(float(np.abs(e.P_).sum()))
            wsums.append(float(np.abs(e.w_).sum()))

        fm = oracle.OracleFM(n_components=3, max_iter=3, tol=0, random_state=0, gamma=1e-3,
                             solver=solver, regularizer=regname, callback=cb, n_calls=1)
        fm.fit(X, y)
        np.testing.assert_allclose(sums, facts["callback_P_abs_sums|" + solver], rtol=1e-10)
        np.testing.assert_allclose(wsums, facts["callback_w_abs_sums|" + solver], rtol=1e-10)
    fm = oracle.OracleFM(n_components=3, max_iter=2, tol=0, random_state=3, gamma=1e-3,
                         shuffle=True, init_lambdas="random_signs", regularizer="l1")
    fm.fit(X, y)
    np.testing.assert_allclose(fm.lams_, z["shuffle_lams"])
    np.testing.assert_allclose(fm.P_, z["shuffle_P"], rtol=0, atol=TOL)
    np.testing.assert_allclose(fm.w_, z["shuffle_w"], rtol=0, atol=TOL)


# ---------------------------------------------------------------- all-subsets (N2)
def _g8_cells():
    return [str(c) for c in load_golden("g8_all_subsets.npz")["cells"]]


@pytest.mark.parametrize("cell", _g8_cells())
def test_g8_all_subsets_reference_cells(oracle, cell):
    """Replicas of reference tests/test_pcd.py:354-432 and tests/test_pbcd.py:345-425."""
    z = load_golden("g8_all_subsets.npz")
    solver, regname, mean, loss = cell.split("|")
    y = z["y"] if loss == "squared" else np.sign(z["y"])
    fm = oracle.OracleAllSubsets(loss=loss, n_components=5, solver=solver, beta=1, gamma=1e-3,
                                 regularizer=regname, tol=1e-3, max_iter=5, random_state=0,
                                 mean=bool(int(mean[4:])))
    fm.fit(z["X"], y)
    np.testing.assert_allclose(fm.P_, z["P|" + cell], rtol=0, atol=TOL)
    assert fm.n_iter_ == int(z["n_iter|" + cell])
    np.testing.assert_allclose(fm.predict(z["X"]), z["pred|" + cell], rtol=0, atol=1e-9)


def _g8_scases():
    return [str(c) for c in load_golden("g8_all_subsets.npz")["scases"]]


@pytest.mark.parametrize("case", _g8_scases())
def test_g8_all_subsets_sparse_trajectories(oracle, case):
    import scipy.sparse as sp

    z = load_golden("g8_all_subsets.npz")
    X = sp.csr_matrix((z["Xs_data"], z["Xs_indices"], z["Xs_indptr"]),
                      shape=tuple(int(v) for v in z["Xs_shape"]))
    solver, regname, loss = case.split("|")
    meta = json.loads(str(z["smeta"]))
    ys = z["ys"]
    y = ys if loss == "squared" else np.where(ys > np.median(ys), 1.0, -1.0)
    fm = oracle.OracleAllSubsets(loss=loss, n_components=z["sP0|" + case].shape[0],
                                 solver=solver, beta=meta["beta"], gamma=meta["gamma"],
                                 eta0=meta["eta0"], regularizer=regname, tol=0, max_iter=3)
    fm.fit(X, y, P_init=z["sP0|" + case], lams_init=z["slams|" + case])
    np.testing.assert_allclose([h[0] for h in fm.history], z["sviol|" + case], rtol=1e-10)
    np.testing.assert_allclose([h[1] for h in fm.history], z["sloss|" + case], rtol=1e-10)
    np.testing.assert_allclose(fm.P_, z["sP|" + case], rtol=0, atol=TOL)
    np.testing.assert_allclose(fm.y_pred_, z["sy_pred|" + case], rtol=0, atol=1e-9)


def test_g8_all_subsets_kernel(oracle):
    z = load_golden("g8_all_subsets.npz")
    got = oracle.all_subsets_predict(z["X"], z["P_true"], z["lams_true"])
    np.testing.assert_allclose(got, z["K"] @ z["lams_true"], rtol=0, atol=1e-12)


# ------------------------------------------------------------------ g9: psgd
def _g9_groups():
    """The 720 reference test cells, grouped by (degree, loss, regularizer) so the suite
    stays small: each test checks its 20 batch_size x learning_rate x fit_linear cells."""
    cells = [str(c) for c in load_golden("g9_psgd.npz")["cells"]]
    groups = {}
    for c in cells:
        deg, bs, lr, fl, loss, reg = c.split("|")
        groups.setdefault("%s|%s|%s" % (deg, loss, reg), []).append(c)
    return groups


@pytest.mark.parametrize("group", sorted(_g9_groups()))
def test_g9_psgd_reference_test_cells(oracle, group):
    """Replicas of reference tests/test_psgd.py:240-366 (fit on RandomState(1) data)."""
    z = load_golden("g9_psgd.npz")
    X = z["X"]
    for cell in _g9_groups()[group]:
        deg, bs, lr, fl, loss, regname = cell.split("|")
        degree = int(deg[3:])
        y = z["y|deg%d" % degree]
        if loss != "squared":
            y = np.sign(y)
        fm = oracle.OracleFM(degree=degree, loss=loss, n_components=5, solver="psgd",
                             regularizer=regname, alpha=1e-3, beta=1e-3, gamma=0.0, tol=1e-3,
                             fit_lower=None, fit_linear=bool(int(fl)), max_iter=10,
                             random_state=0, learning_rate=lr, eta0=0.01,
                             batch_size=bs if bs == "auto" else int(bs))
        fm.fit(X, y)
        np.testing.assert_allclose(fm.P_, z["P|" + cell], rtol=0, atol=TOL, err_msg=cell)
        np.testing.assert_allclose(fm.w_, z["w|" + cell], rtol=0, atol=TOL, err_msg=cell)
        assert [fm.n_iter_, fm.it_] == [int(v) for v in z["n_iter|" + cell]], cell


@pytest.mark.parametrize("regname", ["l1", "l21", "squaredl12", "squaredl21"])
def test_g9_prox_operators(oracle, regname):
    """regularizer.prox (l1.py:50, l21.py:43-48, squaredl12.py:66-75, squaredl21.py:63-74)
    at the strengths of the reference's tests/test_prox.py:54,69 (+ 0)."""
    z = load_golden("g9_psgd.npz")
    for si, st in enumerate(z["prox_strengths"]):
        P = np.array(z["prox_in"])
        oracle.reg_prox(regname, P, float(st))
        want = z["prox|%s|%d" % (regname, si)]
        np.testing.assert_allclose(P, want, rtol=1e-13, atol=1e-13, err_msg=str(st))
        assert np.array_equal(P == 0, want == 0), st  # identical support


def _g9_tcases():
    return [str(c) for c in load_golden("g9_psgd.npz")["tcases"]]


@pytest.mark.parametrize("case", _g9_tcases())
def test_g9_psgd_sparse_trajectories(oracle, case):
    z = load_golden("g9_psgd.npz")
    m = json.loads(str(z["tmeta|" + case]))
    import scipy.sparse as sp
    X = sp.csr_matrix((z["X_data"], z["X_indices"], z["X_indptr"]), shape=tuple(z["X_shape"]))
    ys = z["ys"]
    y = ys if m["loss"] == "squared" else np.where(ys > np.median(ys), 1.0, -1.0)
    fm = oracle.OracleFM(degree=m["degree"], loss=m["loss"], n_components=m["k"],
                         solver="psgd", regularizer=m["regularizer"], alpha=m["alpha"],
                         beta=m["beta"], gamma=m["gamma"], tol=-1.0, fit_lower=m["fit_lower"],
                         fit_linear=True, max_iter=m["max_iter"], shuffle=m["shuffle"],
                         random_state=m["random_state"], learning_rate=m["learning_rate"],
                         eta0=m["eta0"], power_t=m["power_t"], batch_size=m["batch_size"],
                         n_iter_no_change=1000)
    fm.fit(X, y, P_init=z["tP0|" + case], lams_init=z["tlams|" + case])
    np.testing.assert_allclose([h[0] for h in fm.history], z["tloss|" + case], rtol=1e-11)
    np.testing.assert_allclose(fm.P_, z["tP|" + case], rtol=0, atol=TOL)
    np.testing.assert_allclose(fm.w_, z["tw|" + case], rtol=0, atol=TOL)
    assert [fm.n_iter_, fm.it_] == [int(v) for v in z["tit|" + case]]
