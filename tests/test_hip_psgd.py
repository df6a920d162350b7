"""psgd solver on the device vs the reference-generated goldens
(g9) and the oracle.  Needs a real MI355X: ``pytest -m gpu``.

Tolerances: the gradient scatter uses hardware f64 atomics (summation order inside a
minibatch is not fixed), the dot products are tree reductions -- 1e-16-relative effects, so
'f64' is held to 1e-9.  'f32' stores X and y in float32; the reference's own bar for this
solver is 4 decimals (tests/test_psgd.py:299-300)."""
import json
import warnings

import numpy as np
import pytest
import scipy.sparse as sp
from conftest import load_golden

pytestmark = pytest.mark.gpu

P_ATOL = {"f64": 1e-9, "f32": 5e-5}


def _groups():
    cells = [str(c) for c in load_golden("g9_psgd.npz")["cells"]]
    groups = {}
    for c in cells:
        deg, bs, lr, fl, loss, reg = c.split("|")
        groups.setdefault("%s|%s|%s" % (deg, loss, reg), []).append(c)
    return groups


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("group", sorted(_groups()))
def test_reference_test_cells(group, precision):
    """reference tests/test_psgd.py:240-366 through the estimators (all 720 cells)."""
    from sparsepoly_amd import (SparseFactorizationMachineClassifier,
                                SparseFactorizationMachineRegressor)

    z = load_golden("g9_psgd.npz")
    X = z["X"]
    for cell in _groups()[group]:
        deg, bs, lr, fl, loss, regname = cell.split("|")
        degree = int(deg[3:])
        kw = dict(degree=degree, n_components=5, fit_lower=None, fit_linear=bool(int(fl)),
                  alpha=1e-3, beta=1e-3, gamma=0.0, regularizer=regname, learning_rate=lr,
                  eta0=0.01, warm_start=False, tol=1e-3, max_iter=10, random_state=0,
                  shuffle=False, solver="psgd", batch_size=bs if bs == "auto" else int(bs),
                  precision=precision)
        y = z["y|deg%d" % degree]
        if loss == "squared":
            est = SparseFactorizationMachineRegressor(**kw)
        else:
            est, y = SparseFactorizationMachineClassifier(loss=loss, **kw), np.sign(y)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            est.fit(X, y)
        np.testing.assert_allclose(est.P_, z["P|" + cell], rtol=0, atol=P_ATOL[precision],
                                   err_msg=cell)
        np.testing.assert_allclose(est.w_, z["w|" + cell], rtol=0, atol=P_ATOL[precision],
                                   err_msg=cell)
        assert [est.n_iter_, est.it_] == [int(v) for v in z["n_iter|" + cell]], cell


def _tcases():
    return [str(c) for c in load_golden("g9_psgd.npz")["tcases"]]


def _golden_problem(z, loss):
    X = sp.csr_matrix((z["X_data"], z["X_indices"], z["X_indptr"]), shape=tuple(z["X_shape"]))
    ys = z["ys"]
    return X, (ys if loss == "squared" else np.where(ys > np.median(ys), 1.0, -1.0))


@pytest.mark.parametrize("precision", ["f64", "f32"])
@pytest.mark.parametrize("case", _tcases())
def test_sparse_trajectories(case, precision, capsys):
    """gamma > 0 on a sparse problem: per-epoch losses (verbose output, as the reference
    prints them), it_, final P_/w_ against the reference-generated fixture."""
    from sparsepoly_amd import (SparseFactorizationMachineClassifier,
                                SparseFactorizationMachineRegressor)

    z = load_golden("g9_psgd.npz")
    m = json.loads(str(z["tmeta|" + case]))
    X, y = _golden_problem(z, m["loss"])
    kw = dict(degree=m["degree"], n_components=m["k"], fit_lower=m["fit_lower"], fit_linear=True,
              alpha=m["alpha"], beta=m["beta"], gamma=m["gamma"], regularizer=m["regularizer"],
              learning_rate=m["learning_rate"], eta0=m["eta0"], power_t=m["power_t"],
              warm_start=True, tol=-1.0, n_iter_no_change=1000, max_iter=m["max_iter"],
              random_state=m["random_state"], shuffle=m["shuffle"], solver="psgd",
              batch_size=m["batch_size"], verbose=True, precision=precision)
    if m["loss"] == "squared":
        est = SparseFactorizationMachineRegressor(**kw)
    else:
        est = SparseFactorizationMachineClassifier(loss=m["loss"], **kw)
    est.P_ = np.array(z["tP0|" + case])
    est.w_ = np.zeros(X.shape[1])
    est.lams_ = np.array(z["tlams|" + case])
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        est.fit(X, y)
    out = capsys.readouterr().out
    losses = [float(l.split()[-1]) for l in out.splitlines() if l.startswith("Epoch")]
    f64 = precision == "f64"
    np.testing.assert_allclose(losses, z["tloss|" + case], rtol=1e-10 if f64 else 1e-5)
    np.testing.assert_allclose(est.P_, z["tP|" + case], rtol=0, atol=1e-9 if f64 else 2e-5)
    np.testing.assert_allclose(est.w_, z["tw|" + case], rtol=0, atol=1e-9 if f64 else 2e-5)
    assert [est.n_iter_, est.it_] == [int(v) for v in z["tit|" + case]]
    if m["regularizer"] != "l21":
        assert np.array_equal(est.P_ == 0, z["tP|" + case] == 0) or not f64


@pytest.mark.parametrize("regname,k,degree,n_orders,batch", [
    ("l1", 30, 2, 1, 64), ("l21", 30, 2, 1, 100), ("squaredl12", 30, 2, 1, 128),
    ("squaredl21", 30, 2, 1, 128), ("squaredl12", 70, 3, 2, 500), ("l1", 130, 4, 3, 4000),
    ("squaredl21", 12, 3, 2, 333), ("squaredl12", 16, 2, 1, 1),
])
def test_midsize_against_oracle(oracle, regname, k, degree, n_orders, batch):
    """Every lane width (16/32/64) and component chunking (k > 64), several orders, ragged
    last batch, empty rows; f64 engine vs oracle after two epochs of a shuffled order."""
    from sparsepoly_amd.engine import HipEngine
    from sparsepoly_amd.synth import make_problem

    n, d = (4000, 600) if batch > 1 else (300, 600)
    X, y = make_problem(n, d, 12, seed=5)
    X = sp.csr_matrix(X)
    X.data[:] = np.round(X.data * 64) / 64
    lil = X.tolil()
    lil[7, :] = 0        # an empty row
    X = sp.csr_matrix(lil)
    X.eliminate_zeros()
    rng = np.random.RandomState(1)
    P0 = 0.05 * rng.randn(n_orders, k, d)
    lams = np.sign(rng.randn(k))
    w0 = 0.01 * rng.randn(d)
    gamma = {"l1": 0.003, "l21": 0.5, "squaredl12": 1e-3, "squaredl21": 2e-2}[regname]
    alpha, beta, eta0, lr, power_t = 1e-2, 0.5, 0.05, "optimal", 0.8
    eng = HipEngine(0, "f64")
    eng.set_data(X, y)
    eng.set_params(P0, w0, lams)
    eng.configure("psgd", "squared", regname, degree)
    Po = np.ascontiguousarray(P0.swapaxes(1, 2))
    wo = w0.copy()
    Xr = oracle.CSR(X)
    it_d = it_o = 1
    for ep in range(2):
        idx = rng.permutation(n).astype(np.int32)
        sl_d, it_d = eng.psgd_epoch(degree, alpha, beta, gamma, eta0, lr, power_t, batch, idx,
                                    True, it_d)
        sl_o, it_o = oracle.psgd_epoch(Po, wo, Xr, y, lams, degree, alpha, beta, gamma, regname,
                                       "squared", idx, True, eta0, lr, power_t, batch, it_o)
        assert it_d == it_o
        np.testing.assert_allclose(sl_d, sl_o, rtol=1e-11)
    Pd = np.empty_like(P0)
    wd = np.empty(d)
    eng.get_params(Pd, wd)
    eng.close()
    np.testing.assert_allclose(Pd, Po.swapaxes(1, 2), rtol=0, atol=1e-10)
    np.testing.assert_allclose(wd, wo, rtol=0, atol=1e-10)
    if regname in ("l1", "squaredl12"):
        nz = np.mean(Pd != 0)
        if 1 < batch <= 128:                   # enough updates for the prox to bite
            assert 0.02 < nz < 0.98, nz       # ... it really prunes and really keeps
        assert np.array_equal(Pd == 0, Po.swapaxes(1, 2) == 0)


def test_classifier_losses_midsize(oracle):
    from sparsepoly_amd.engine import HipEngine
    from sparsepoly_amd.synth import make_problem

    n, d, k = 3000, 400, 20
    X, y = make_problem(n, d, 10, seed=9)
    yb = np.where(y > np.median(y), 1.0, -1.0)
    rng = np.random.RandomState(2)
    P0 = 0.05 * rng.randn(1, k, d)
    lams = np.ones(k)
    for loss in ("squared_hinge", "logistic"):
        for prec, tol in (("f64", 1e-10), ("f32", 2e-5)):
            eng = HipEngine(0, prec)
            eng.set_data(X, yb)
            eng.set_params(P0, np.zeros(d), lams)
            eng.configure("psgd", loss, "l1", 2)
            Po = np.ascontiguousarray(P0.swapaxes(1, 2))
            wo = np.zeros(d)
            idx = np.arange(n, dtype=np.int32)
            sl_d, it_d = eng.psgd_epoch(2, 1e-3, 0.1, 0.05, 0.1, "invscaling", 0.5, 50, idx,
                                        True, 1)
            sl_o, it_o = oracle.psgd_epoch(Po, wo, oracle.CSR(X), yb, lams, 2, 1e-3, 0.1, 0.05,
                                           "l1", loss, idx, True, 0.1, "invscaling", 0.5, 50, 1)
            Pd = np.empty_like(P0)
            wd = np.empty(d)
            eng.get_params(Pd, wd)
            eng.close()
            assert it_d == it_o == 61
            np.testing.assert_allclose(sl_d, sl_o, rtol=1e-10 if prec == "f64" else 1e-5)
            np.testing.assert_allclose(Pd, Po.swapaxes(1, 2), rtol=0, atol=tol)
            np.testing.assert_allclose(wd, wo, rtol=0, atol=tol)


def test_psgd_errors_and_api():
    from sparsepoly_amd import SparseFactorizationMachineRegressor
    from sparsepoly_amd.engine import HipEngine

    rng = np.random.RandomState(0)
    X = sp.random(60, 12, density=0.3, random_state=rng, format="csr")
    y = rng.randn(60)
    for regname in ("omegati", "omegacs"):
        with pytest.raises(ValueError):
            SparseFactorizationMachineRegressor(solver="psgd", regularizer=regname,
                                                max_iter=1).fit(X, y)
    with pytest.raises(ValueError, match="learning_rate"):
        SparseFactorizationMachineRegressor(solver="psgd", regularizer="l1", max_iter=1,
                                            learning_rate="nope").fit(X, y)
    # callbacks: initial P_, live w_ (reference trains a copy of P_)
    seen = []
    est = SparseFactorizationMachineRegressor(
        solver="psgd", regularizer="l1", max_iter=3, tol=-1, n_iter_no_change=100, gamma=1e-3,
        random_state=0, n_calls=1,
        callback=lambda e: seen.append((float(np.abs(e.P_).sum()), float(np.abs(e.w_).sum()))))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        est.fit(X, y)
    assert len(seen) == 3 and seen[0][0] == seen[1][0] == seen[2][0]
    assert len({s[1] for s in seen}) == 3
    assert abs(np.abs(est.P_).sum() - seen[0][0]) > 0
    # warm start keeps it_
    it1 = est.it_
    est.set_params(warm_start=True, callback=None)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        est.fit(X, y)
    assert est.it_ == it1 + (it1 - 1)
    # ABI argument checks
    eng = HipEngine(0, "f64")
    eng.set_data(X, y)
    eng.set_params(np.zeros((1, 2, 12)), np.zeros(12), np.ones(2))
    eng.configure("psgd", "squared", "l1", 2)
    idx = np.arange(60, dtype=np.int32)
    with pytest.raises(ValueError):
        eng.psgd_epoch(2, 1, 1, 1, 0.1, "optimal", 1.0, 0, idx, True, 1)       # batch 0
    with pytest.raises(ValueError):
        eng.psgd_epoch(2, 1, 1, 1, 0.1, "optimal", 1.0, 5, idx[:-1], True, 1)  # short order
    bad = idx.copy()
    bad[3] = bad[4]
    with pytest.raises(ValueError):
        eng.psgd_epoch(2, 1, 1, 1, 0.1, "optimal", 1.0, 5, bad, True, 1)       # not a perm
    with pytest.raises(ValueError):
        eng.psgd_epoch(3, 1, 1, 1, 0.1, "optimal", 1.0, 5, idx, True, 1)       # degree
    with pytest.raises(ValueError):
        eng.pcd_epoch(0, 2, 1, 1, 1, np.arange(2, dtype=np.int32))             # wrong solver
    eng.close()


@pytest.mark.parametrize("regname", ["squaredl12", "squaredl21", "l1"])
def test_graph_replay_and_eager_redo_give_the_same_result(oracle, regname):
    """l1 / l21 and (after the first, cold epoch) the squared-norm regularizers replay runs of
    32 minibatches from a hipGraph.  For the squared norms the graph records a fixed number of
    support-search sweeps; with too few (0) every replayed epoch fails its check and is redone
    eagerly from the snapshot.  All three ways must agree with the oracle."""
    from sparsepoly_amd.engine import HipEngine
    from sparsepoly_amd.synth import make_problem

    n, d, k = 6400, 500, 12
    X, y = make_problem(n, d, 12, seed=11)
    rng = np.random.RandomState(3)
    P0 = 0.05 * rng.randn(1, k, d)
    lams = np.ones(k)
    gamma = 1e-3 if regname != "l1" else 3e-3
    args = (2, 1e-2, 0.5, gamma, 0.05, "optimal", 0.8, 50)   # 128 minibatches per epoch
    res = {}
    for mode, opts in (("graph", {}), ("eager", {"psgd_eager": 1}),
                       ("redo", {"psgd_graph_sweeps": 0})):
        eng = HipEngine(0, "f64")
        for key, val in opts.items():
            eng.set_option(key, val)
        eng.set_data(X, y)
        eng.set_params(P0, np.zeros(d), lams)
        eng.configure("psgd", "squared", regname, 2)
        it, sls = 1, []
        r2 = np.random.RandomState(4)
        for _ in range(3):
            idx = r2.permutation(n).astype(np.int32)
            sl, it = eng.psgd_epoch(*args, idx, True, it)
            sls.append(sl)
        res[mode] = (eng.get_params(), np.array(sls), it, eng.get_option("psgd_redone"))
        eng.close()
    Po = np.ascontiguousarray(P0.swapaxes(1, 2))
    wo = np.zeros(d)
    it_o, r2, slo = 1, np.random.RandomState(4), []
    for _ in range(3):
        idx = r2.permutation(n).astype(np.int32)
        sl, it_o = oracle.psgd_epoch(Po, wo, oracle.CSR(X), y, lams, 2, 1e-2, 0.5, gamma, regname,
                                     "squared", idx, True, 0.05, "optimal", 0.8, 50, it_o)
        slo.append(sl)
    for mode in res:
        (P, w), sls, it, redone = res[mode]
        assert it == it_o == 1 + 3 * 128
        np.testing.assert_allclose(sls, slo, rtol=1e-10, err_msg=mode)
        np.testing.assert_allclose(P, Po.swapaxes(1, 2), rtol=0, atol=1e-10, err_msg=mode)
        np.testing.assert_allclose(w, wo, rtol=0, atol=1e-10, err_msg=mode)
    assert res["eager"][3] == 0
    if regname == "l1":
        assert res["graph"][3] == 0 and res["redo"][3] == 0
    else:
        assert res["redo"][3] == 2      # epochs 2 and 3 (epoch 1 runs eagerly: cold start)
        assert res["graph"][3] == 0     # 4 recorded sweeps were enough
