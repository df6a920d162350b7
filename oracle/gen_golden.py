#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ from the REFERENCE itself.

TEST INFRASTRUCTURE.  Runs only in the build container, where /root/reference is
mounted:

    PYTHONDONTWRITEBYTECODE=1 \
    PYTHONPATH=oracle/numba_stub:/root/reference python3 oracle/gen_golden.py

The reference (neonnnnn/sparsepoly) is pure Python + Numba; Numba is not
installable here, so ``oracle/numba_stub`` turns ``@njit``/``@jitclass`` into
identity decorators and the reference's own source runs under CPython (same
float64 operations in the same order; the reference uses no fastmath/parallel).
Nothing from the reference is copied: only inputs and outputs are stored.

Fixture families:
  g1  replicas of the reference's own test cells (tests/test_pcd.py:177-351,
      tests/test_pbcd.py:163-342): estimator.fit on the RandomState(1) data
  g2  BASELINE config 1 (1k x 100 CSR, k=4, l1, pcd) with gamma in {1e-3, 1e-2}
  g3  small (300 x 60) versions of configs 2/3/4 (+ omegacs degree 3), three
      losses, driven epoch by epoch -> viol and sum-loss trajectories.  beta is
      chosen so the iteration is well conditioned: with beta <= 0.1 the pcd
      trajectory on this tiny problem amplifies a 1e-16 perturbation to O(1)
      within one epoch (measured with the oracle), which would pin nothing.
  g4  direct pcd_epoch / pbcd_epoch calls with permuted coordinate orders
  g5  per-call regularizer traces (prox_cd / prox_bcd inputs -> outputs, caches)
  g6  anova_kernel / poly_predict, degree 2,3,4, sparse and dense
  g7  API behaviours (n_iter_ semantics, stale P_ in pbcd callbacks, messages)
  g8  all-subsets model: the reference's own test cells + sparse trajectories
  g9  psgd solver: the reference's own test grid, prox operators, sparse trajectories
  g10 regularizer eval() of all six plug-ins (2-D and stacked 3-D inputs, both `transpose`
      settings, degrees 2-4 and -1) -- never called by a solver, but part of the plug-in
      surface a user of the reference can call
"""
import contextlib
import io
import json
import os
import sys
import warnings

import numpy as np
import scipy.sparse as sp

import sparsepoly  # the reference (via PYTHONPATH)
from sparsepoly import (
    SparseFactorizationMachineClassifier,
    SparseFactorizationMachineRegressor,
)
from sparsepoly.dataset import get_dataset
from sparsepoly.kernels import anova_kernel, poly_predict
from sparsepoly.loss import CLASSIFICATION_LOSSES
from sparsepoly.optimizer import cd_linear, pbcd, pcd
from sparsepoly.regularizer import REGULARIZATION

assert "/root/reference" in os.path.abspath(sparsepoly.__file__), sparsepoly.__file__

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
OUT = os.path.abspath(OUT)
os.makedirs(OUT, exist_ok=True)


def save(name, **arrays):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **arrays)
    print("wrote %s (%d arrays, %d bytes)" % (name, len(arrays), os.path.getsize(path)))


def fit_verbose(est, X, y):
    """fit() with verbose=True; returns the per-iteration 'violation sum' values."""
    est.set_params(verbose=True)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        est.fit(X, y)
    viols = []
    for line in buf.getvalue().splitlines():
        if line.startswith("Iteration"):
            viols.append(float(line.split()[-1]))
    return np.array(viols)


# --------------------------------------------------------------------- g1
def gen_g1():
    rng = np.random.RandomState(1)
    X = rng.randn(20, 4)
    P = rng.randn(5, 4)
    lams = rng.randn(5)
    out = {"X": X, "P_true": P, "lams_true": lams}
    cells = []
    # (solver, regularizer, degrees, max_iter)  -- tests/test_pcd.py, tests/test_pbcd.py
    plan = [
        ("pcd", "l1", [2, 3, 4], 5),
        ("pcd", "omegati", [2, 3, 4], 5),
        ("pcd", "squaredl12", [2], 1),
        ("pbcd", "l1", [2, 3, 4], 5),
        ("pbcd", "l21", [2, 3, 4], 5),
        ("pbcd", "omegacs", [2, 3, 4], 5),
        ("pbcd", "squaredl21", [2], 1),
    ]
    for solver, regname, degrees, max_iter in plan:
        for degree in degrees:
            y_reg = poly_predict(X, P, lams, kernel="anova", degree=degree)
            out["y_deg%d" % degree] = y_reg
            for mean in (True, False):
                for loss in ("squared", "squared_hinge", "logistic"):
                    common = dict(
                        degree=degree, n_components=5, fit_lower=None, fit_linear=False,
                        beta=1, gamma=1e-3, regularizer=regularizer_name(regname),
                        warm_start=False, tol=1e-3, max_iter=max_iter, random_state=0,
                        mean=mean, shuffle=False, solver=solver,
                    )
                    if loss == "squared":
                        est = SparseFactorizationMachineRegressor(**common)
                        y = y_reg
                    else:
                        est = SparseFactorizationMachineClassifier(loss=loss, **common)
                        y = np.sign(y_reg)
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        est.fit(X, y)
                    key = "%s|%s|deg%d|mean%d|%s" % (solver, regname, degree, int(mean), loss)
                    cells.append(key)
                    out["P|" + key] = est.P_.copy()
                    out["n_iter|" + key] = np.array(est.n_iter_)
    out["cells"] = np.array(cells)
    save("g1_reftests.npz", **out)


def regularizer_name(n):
    return n


# --------------------------------------------------------------------- g2
def gen_g2():
    X = sp.random(1000, 100, density=0.05, random_state=np.random.RandomState(0),
                  data_rvs=np.random.RandomState(0).randn, format="csr")
    y = np.random.RandomState(42).randn(1000)
    out = {"X_data": X.data, "X_indices": X.indices, "X_indptr": X.indptr,
           "X_shape": np.array(X.shape), "y": y}
    for gamma in (1e-3, 1e-2):
        est = SparseFactorizationMachineRegressor(
            degree=2, n_components=4, regularizer="l1", solver="pcd", gamma=gamma,
            max_iter=6, tol=1e-9, random_state=0)
        viols = fit_verbose(est, X, y)
        tag = "gamma%g" % gamma
        out["viol|" + tag] = viols
        out["P|" + tag] = est.P_.copy()
        out["w|" + tag] = est.w_.copy()
        out["pred|" + tag] = est.predict(X)
        out["n_iter|" + tag] = np.array(est.n_iter_)
    save("g2_config1.npz", **out)


# ----------------------------------------------------------- epoch driving
def drive(X, y, solver, regname, loss, degree, k, alpha, beta, gamma, eta0, n_epochs,
          fit_linear, fit_lower, P0, lams, feature_orders=None, component_orders=None):
    """Restates the loop of _fit_pcd/_fit_pbcd (sparse_factorization_machines.py:196-256,
    :287-352) around the REFERENCE's epoch functions so per-epoch viol, sum-loss and
    y_pred can be recorded, and so coordinate orders can be injected."""
    n, d = X.shape
    ds = get_dataset(X, order="fortran")
    loss_obj = CLASSIFICATION_LOSSES[loss]
    reg = REGULARIZATION[regname]()
    n_orders = P0.shape[0]
    P_ = P0.copy()
    w = np.zeros(d)
    est_like_output = poly_predict(X, P_[0], lams, kernel="anova", degree=degree)
    y_pred = np.array(est_like_output, dtype=np.float64)
    if fit_lower == "explicit" and degree == 3:
        y_pred += poly_predict(X, P_[1], lams, kernel="anova", degree=2)
    Xc = sp.csc_matrix(X) if sp.issparse(X) else np.asarray(X)
    if sp.issparse(X):
        col_norm_sq = np.asarray(Xc.multiply(Xc).sum(axis=0)).ravel()
    else:
        col_norm_sq = (Xc ** 2).sum(axis=0)
    viols, losses = [], []
    if solver == "pcd":
        A = np.zeros((n, degree + 1))
        dA = np.zeros(degree)
        A[:, 0] = 1.0
        reg.init_cache_pcd(degree, d, k)
        P = P_
    else:
        A = np.zeros((n, degree + 1, k))
        dA = np.zeros((n, degree, k))
        grad, inv_ss, p_old = np.zeros(k), np.zeros(k), np.zeros(k)
        A[:, 0] = 1.0
        reg.init_cache_pbcd(degree, d, k)
        P = np.array(P_.swapaxes(1, 2))
    for it in range(n_epochs):
        jf = (np.arange(d, dtype=np.int32) if feature_orders is None
              else np.asarray(feature_orders[it], dtype=np.int32))
        ic = (np.arange(k, dtype=np.int32) if component_orders is None
              else np.asarray(component_orders[it], dtype=np.int32))
        viol = 0
        if fit_linear:
            viol += cd_linear._cd_linear_epoch(w, ds, y, y_pred, col_norm_sq, alpha, loss_obj, jf)
        degs = list(range(2, degree)) if fit_lower == "explicit" else []
        for deg in degs + [degree]:
            order = degree - deg if deg != degree else 0
            if solver == "pcd":
                viol += pcd.pcd_epoch(P[order], ds, y, y_pred, lams, deg, beta, gamma, eta0,
                                      reg, loss_obj, A, dA, ic, jf)
            else:
                viol += pbcd.pbcd_epoch(P[order], ds, y, y_pred, lams, deg, beta, gamma, eta0,
                                        reg, loss_obj, A, dA, grad, inv_ss, p_old, jf)
        viols.append(viol)
        losses.append(sum(loss_obj.loss(y_pred[i], y[i]) for i in range(n)))
    if solver == "pbcd":
        P_ = np.array(P.swapaxes(1, 2))
    return dict(viol=np.array(viols), loss=np.array(losses), P=P_, w=w, y_pred=y_pred)


def small_problem(n=300, d=60, density=0.1, seed=7):
    X = sp.random(n, d, density=density, random_state=np.random.RandomState(seed),
                  data_rvs=np.random.RandomState(seed + 1).randn, format="csr")
    # values exactly representable in float32, so fp32-storage engines read the same numbers
    X.data = X.data.astype(np.float32).astype(np.float64)
    rng = np.random.RandomState(seed + 2)
    Pt = rng.randn(3, d) * (rng.rand(3, d) < 0.3)
    y = poly_predict(X, Pt, np.ones(3), kernel="anova", degree=2) + 0.1 * rng.randn(n)
    y = y.astype(np.float32).astype(np.float64)
    return X, y


def gen_g3():
    X, y = small_problem()
    n, d = X.shape
    out = {"X_data": X.data, "X_indices": X.indices, "X_indptr": X.indptr,
           "X_shape": np.array(X.shape), "y": y}
    cases = [
        # tag, solver, reg, degree, k, fit_lower
        ("c2", "pcd", "squaredl12", 2, 30, "explicit"),
        ("c3", "pcd", "omegati", 3, 16, "explicit"),
        ("c4", "pbcd", "omegacs", 2, 30, "explicit"),
        ("c4d3", "pbcd", "omegacs", 3, 16, "explicit"),
        ("l1", "pcd", "l1", 2, 8, "explicit"),
        ("l21", "pbcd", "l21", 2, 8, "explicit"),
        ("sql21", "pbcd", "squaredl21", 2, 8, "explicit"),
        ("l1b", "pbcd", "l1", 3, 8, "explicit"),
        ("ti4", "pcd", "omegati", 4, 6, "explicit"),
    ]
    names = []
    for tag, solver, regname, degree, k, fit_lower in cases:
        n_orders = degree - 1 if fit_lower == "explicit" else 1
        P0 = 0.01 * np.random.RandomState(0).randn(n_orders, k, d)
        lams = np.sign(np.random.RandomState(5).randn(k))
        for loss in ("squared", "squared_hinge", "logistic"):
            yy = y if loss == "squared" else np.where(y > np.median(y), 1.0, -1.0)
            beta = 10.0 if solver == "pcd" else 1.0
            gamma = 0.1
            r = drive(X, yy, solver, regname, loss, degree, k, alpha=1e-2, beta=beta,
                      gamma=gamma, eta0=1.0,
                      n_epochs=4, fit_linear=True, fit_lower=fit_lower, P0=P0, lams=lams)
            key = "%s|%s" % (tag, loss)
            names.append(key)
            out["meta|" + key] = np.array(
                json.dumps(dict(solver=solver, regularizer=regname, degree=degree, k=k,
                                fit_lower=fit_lower, loss=loss, alpha=1e-2, beta=beta,
                                gamma=gamma)))
            out["P0|" + key] = P0
            out["lams|" + key] = lams
            for kk, v in r.items():
                out["%s|%s" % (kk, key)] = v
    out["cases"] = np.array(names)
    save("g3_small_configs.npz", **out)


def gen_g4():
    X, y = small_problem(n=200, d=40, density=0.15, seed=11)
    n, d = X.shape
    out = {"X_data": X.data, "X_indices": X.indices, "X_indptr": X.indptr,
           "X_shape": np.array(X.shape), "y": y}
    names = []
    prm = np.random.RandomState(3)
    for tag, solver, regname, degree, k in [
        ("pcd_l12", "pcd", "squaredl12", 2, 6),
        ("pcd_ti3", "pcd", "omegati", 3, 5),
        ("pcd_l1", "pcd", "l1", 2, 6),
        ("pbcd_cs", "pbcd", "omegacs", 2, 6),
        ("pbcd_l21", "pbcd", "squaredl21", 2, 6),
    ]:
        n_epochs = 3
        forders = np.stack([prm.permutation(d) for _ in range(n_epochs)]).astype(np.int32)
        corders = np.stack([prm.permutation(k) for _ in range(n_epochs)]).astype(np.int32)
        n_orders = degree - 1
        P0 = 0.01 * np.random.RandomState(1).randn(n_orders, k, d)
        lams = np.sign(np.random.RandomState(2).randn(k))
        beta = 10.0 if solver == "pcd" else 1.0
        r = drive(X, y, solver, regname, "squared", degree, k, alpha=1e-2, beta=beta,
                  gamma=0.05, eta0=0.7, n_epochs=n_epochs, fit_linear=True,
                  fit_lower="explicit", P0=P0, lams=lams, feature_orders=forders,
                  component_orders=corders)
        names.append(tag)
        out["meta|" + tag] = np.array(json.dumps(dict(
            solver=solver, regularizer=regname, degree=degree, k=k, loss="squared",
            alpha=1e-2, beta=beta, gamma=0.05, eta0=0.7)))
        out["forders|" + tag] = forders
        out["corders|" + tag] = corders
        out["P0|" + tag] = P0
        out["lams|" + tag] = lams
        for kk, v in r.items():
            out["%s|%s" % (kk, tag)] = v
    out["cases"] = np.array(names)
    save("g4_permuted.npz", **out)


# --------------------------------------------------------------------- g5
def gen_g5():
    """Per-call traces of the regularizer protocol, including inputs that reach the
    'numerical error' branches (omegati.py:97-98, omegacs.py:75-76,90-96,
    squaredl21.py:48-49)."""
    out = {}
    rng = np.random.RandomState(9)
    d, k = 12, 5
    # ---- pcd-side: prox_cd / update_cache_pcd sweeps
    for regname, degree in [("l1", 2), ("squaredl12", 2), ("omegati", 2), ("omegati", 3),
                            ("omegati", 4)]:
        reg = REGULARIZATION[regname]()
        reg.init_cache_pcd(degree, d, k)
        P = rng.randn(k, d) * (rng.rand(k, d) < 0.7)
        P0 = P.copy()
        p_in, strengths, p_out, cache_tr, dcache_tr = [], [], [], [], []
        for s in range(k):
            reg.compute_cache_pcd(P, degree, s)
            for j in range(d):
                cand = P[s, j] + 0.3 * rng.randn()
                st = abs(0.2 * rng.randn())
                new = reg.prox_cd(cand, st, degree, j)
                P[s, j] = new
                reg.update_cache_pcd(P, degree, s, j)
                p_in.append(cand)
                strengths.append(st)
                p_out.append(new)
                if regname != "l1":
                    c = np.zeros(degree + 1)
                    c[: len(reg._cache)] = reg._cache
                    cache_tr.append(c)
                    dc = np.zeros(degree + 1)
                    if hasattr(reg, "_dcache") and regname == "omegati":
                        dc[:] = reg._dcache
                    dcache_tr.append(dc)
        tag = "pcd|%s|deg%d" % (regname, degree)
        out["P0|" + tag] = P0
        out["p_in|" + tag] = np.array(p_in)
        out["strength|" + tag] = np.array(strengths)
        out["p_out|" + tag] = np.array(p_out)
        out["P_end|" + tag] = P
        if cache_tr:
            out["cache|" + tag] = np.array(cache_tr)
            out["dcache|" + tag] = np.array(dcache_tr)
    # ---- pbcd-side
    for regname, degree, mode in [("l1", 2, "plain"), ("l21", 2, "plain"),
                                  ("squaredl21", 2, "plain"), ("omegacs", 2, "plain"),
                                  ("omegacs", 3, "plain"), ("omegacs", 4, "plain"),
                                  ("omegacs", 3, "poison"), ("squaredl21", 2, "poison")]:
        reg = REGULARIZATION[regname]()
        reg.init_cache_pbcd(degree, d, k)
        P = rng.randn(d, k) * (rng.rand(d, 1) < 0.8)
        P0 = P.copy()
        reg.compute_cache_pbcd(P, degree)
        v_in, strengths, v_out, cache_tr, dcache_tr, norms_tr = [], [], [], [], [], []
        for sweep in range(2):
            for j in range(d):
                if mode == "poison" and sweep == 1 and j % 4 == 1:
                    # push the running caches into the fallback branches
                    if regname == "omegacs":
                        reg._cache[degree - 1] = -abs(reg._cache[degree - 1]) - 1e-3 \
                            if j % 8 == 1 else reg._cache[degree - 1]
                        reg._norms[j] = reg._norms[j] + 50.0
                    else:
                        reg._cache = reg._norms[j] - 1e-3
                cand = P[j] + 0.3 * rng.randn(k)
                st = abs(0.2 * rng.randn())
                v_in.append(cand.copy())
                strengths.append(st)
                pj = cand.copy()
                reg.prox_bcd(pj, st, degree, j)
                P[j] = pj
                reg.update_cache_pbcd(P, degree, j)
                v_out.append(pj.copy())
                if regname in ("squaredl21", "omegacs"):
                    c = np.zeros(degree + 1)
                    if regname == "squaredl21":
                        c[0] = reg._cache
                    else:
                        c[:] = reg._cache
                        dc = np.array(reg._dcache)
                        dcache_tr.append(dc)
                    cache_tr.append(c)
                    norms_tr.append(np.array(reg._norms))
        tag = "pbcd|%s|deg%d|%s" % (regname, degree, mode)
        out["P0|" + tag] = P0
        out["v_in|" + tag] = np.array(v_in)
        out["strength|" + tag] = np.array(strengths)
        out["v_out|" + tag] = np.array(v_out)
        out["P_end|" + tag] = P
        if cache_tr:
            out["cache|" + tag] = np.array(cache_tr)
            out["norms|" + tag] = np.array(norms_tr)
        if dcache_tr:
            out["dcache|" + tag] = np.array(dcache_tr)
    # ---- losses
    p = np.concatenate([np.linspace(-30, 30, 61), [0.0, 1e-9, -1e-9]])
    for loss in ("squared", "squared_hinge", "logistic"):
        lo = CLASSIFICATION_LOSSES[loss]
        for yv in (-1.0, 1.0, 0.37):
            out["dloss|%s|y%g" % (loss, yv)] = np.array([lo.dloss(pi, yv) for pi in p])
            out["loss|%s|y%g" % (loss, yv)] = np.array([lo.loss(pi, yv) for pi in p])
        out["mu|" + loss] = np.array(float(lo.mu))
    out["loss_p"] = p
    save("g5_reg_traces.npz", **out)


# --------------------------------------------------------------------- g6
def gen_g6():
    rng = np.random.RandomState(21)
    Xd = rng.randn(30, 9) * (rng.rand(30, 9) < 0.5)
    Xs = sp.csr_matrix(Xd)
    P = rng.randn(4, 9)
    lams = np.sign(rng.randn(4))
    out = {"X": Xd, "P": P, "lams": lams}
    for degree in (2, 3, 4, 5):
        out["K_dense|deg%d" % degree] = anova_kernel(Xd, P, degree)
        out["K_sparse|deg%d" % degree] = np.asarray(anova_kernel(Xs, P, degree))
        out["pred|deg%d" % degree] = poly_predict(Xd, P, lams, kernel="anova", degree=degree)
    # estimator-level _get_output incl. linear and explicit degree-2 term
    for degree, fit_lower in [(2, "explicit"), (3, "explicit"), (3, None)]:
        est = SparseFactorizationMachineRegressor(degree=degree, n_components=4,
                                                  fit_lower=fit_lower)
        n_orders = degree - 1 if fit_lower == "explicit" else 1
        est.P_ = rng.randn(n_orders, 4, 9)
        est.w_ = rng.randn(9)
        est.lams_ = lams
        tag = "deg%d|%s" % (degree, fit_lower)
        out["est_P|" + tag] = est.P_
        out["est_w|" + tag] = est.w_
        out["est_pred|" + tag] = est.predict(Xs)
    save("g6_anova.npz", **out)


# --------------------------------------------------------------------- g7
def gen_g7():
    facts = {}
    X, y = small_problem(n=80, d=15, density=0.3, seed=4)
    est = SparseFactorizationMachineRegressor(n_components=3, max_iter=3, tol=0,
                                              random_state=0, gamma=1e-3)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        est.fit(X, y)
    facts["n_iter_after_max_iter_3"] = int(est.n_iter_)
    facts["warning_text"] = str(w[-1].message)
    # callback sees live P_ under pcd, stale under pbcd
    for solver, regname in (("pcd", "l1"), ("pbcd", "l21")):
        sums, wsums = [], []

        def cb(e):
            sums.append(float(np.abs(e.P_).sum()))
            wsums.append(float(np.abs(e.w_).sum()))

        est = SparseFactorizationMachineRegressor(n_components=3, max_iter=3, tol=0,
                                                  random_state=0, gamma=1e-3, solver=solver,
                                                  regularizer=regname, callback=cb, n_calls=1)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            est.fit(X, y)
        facts["callback_P_abs_sums|" + solver] = sums
        facts["callback_w_abs_sums|" + solver] = wsums
    # callback returning non-None aborts
    est = SparseFactorizationMachineRegressor(n_components=3, max_iter=10, tol=0,
                                              random_state=0, callback=lambda e: True,
                                              n_calls=2)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        est.fit(X, y)
    facts["n_iter_callback_abort"] = int(est.n_iter_)
    # error types
    errs = {}
    for kw, nm in [(dict(regularizer="nope"), "bad_regularizer"),
                   (dict(solver="nope"), "bad_solver"),
                   (dict(init_lambdas="nope"), "bad_init_lambdas"),
                   (dict(degree=3, regularizer="squaredl12"), "squaredl12_degree3"),
                   (dict(solver="pbcd", degree=3, regularizer="squaredl21"),
                    "squaredl21_degree3")]:
        try:
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                SparseFactorizationMachineRegressor(max_iter=1, **kw).fit(X, y)
            errs[nm] = None
        except Exception as e:  # noqa
            errs[nm] = [type(e).__name__, str(e)]
    try:
        SparseFactorizationMachineClassifier(loss="nope").fit(X, np.sign(y))
    except Exception as e:
        errs["bad_loss"] = [type(e).__name__, str(e)]
    try:
        SparseFactorizationMachineClassifier().fit(X, y)
    except Exception as e:
        errs["clf_nonbinary"] = [type(e).__name__, str(e)]
    try:
        SparseFactorizationMachineRegressor().predict(X)
    except Exception as e:
        errs["not_fitted"] = [type(e).__name__, str(e)]
    clf = SparseFactorizationMachineClassifier(max_iter=1, random_state=0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        clf.fit(X, np.where(y > 0, "a", "b"))
    try:
        clf.predict_proba(X)
    except Exception as e:
        errs["predict_proba_nonlogistic"] = [type(e).__name__, str(e)]
    facts["clf_classes"] = [str(c) for c in clf.label_binarizer_.classes_]
    facts["clf_pred_head"] = [str(c) for c in clf.predict(X)[:10]]
    facts["errors"] = errs
    # random_signs + shuffle consume the RNG in a fixed order (P, lams, then shuffles)
    est = SparseFactorizationMachineRegressor(n_components=3, max_iter=2, tol=0,
                                              random_state=3, gamma=1e-3, shuffle=True,
                                              init_lambdas="random_signs", regularizer="l1")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        est.fit(X, y)
    arr = {"X_data": X.data, "X_indices": X.indices, "X_indptr": X.indptr,
           "X_shape": np.array(X.shape), "y": y, "shuffle_P": est.P_, "shuffle_w": est.w_,
           "shuffle_lams": est.lams_, "facts": np.array(json.dumps(facts))}
    save("g7_api.npz", **arr)


# --------------------------------------------------------------------- g8
def gen_g8():
    """All-subsets model: replicas of the reference's own test cells
    (tests/test_pcd.py:354-432, tests/test_pbcd.py:345-425) + a sparse case with
    per-epoch trajectories driven through pcd_all / pbcd_all directly."""
    from sparsepoly import SparseAllSubsetsClassifier, SparseAllSubsetsRegressor
    from sparsepoly.kernels import all_subsets_kernel
    from sparsepoly.optimizer import pbcd_all, pcd_all

    rng = np.random.RandomState(1)
    X = rng.randn(20, 4)
    P = rng.randn(5, 4)
    lams = rng.randn(5)
    y_reg = poly_predict(X, P, lams, kernel="all-subsets")
    out = {"X": X, "y": y_reg, "K": all_subsets_kernel(X, P), "P_true": P, "lams_true": lams}
    cells = []
    for solver, regs in (("pcd", ["l1", "omegati"]), ("pbcd", ["l1", "l21", "omegacs"])):
        for regname in regs:
            for mean in (True, False):
                for loss in ("squared", "squared_hinge", "logistic"):
                    common = dict(n_components=5, beta=1, gamma=1e-3, regularizer=regname,
                                  warm_start=False, tol=1e-3, max_iter=5, random_state=0,
                                  mean=mean, shuffle=False, solver=solver)
                    if loss == "squared":
                        est = SparseAllSubsetsRegressor(**common)
                        yy = y_reg
                    else:
                        est = SparseAllSubsetsClassifier(loss=loss, **common)
                        yy = np.sign(y_reg)
                    with warnings.catch_warnings():
                        warnings.simplefilter("ignore")
                        est.fit(X, yy)
                    key = "%s|%s|mean%d|%s" % (solver, regname, int(mean), loss)
                    cells.append(key)
                    out["P|" + key] = est.P_.copy()
                    out["n_iter|" + key] = np.array(est.n_iter_)
                    out["pred|" + key] = est._get_output(X)
    out["cells"] = np.array(cells)
    # sparse case, epoch by epoch
    Xs, ys = small_problem(n=250, d=50, density=0.12, seed=23)
    Xs.data *= 0.5
    n, d = Xs.shape
    out["Xs_data"], out["Xs_indices"], out["Xs_indptr"] = Xs.data, Xs.indices, Xs.indptr
    out["Xs_shape"] = np.array(Xs.shape)
    out["ys"] = ys
    names = []
    for solver, regname, k in (("pcd", "omegati", 6), ("pcd", "l1", 6), ("pbcd", "omegacs", 6),
                               ("pbcd", "l21", 6), ("pbcd", "l1", 5)):
        for loss in ("squared", "logistic"):
            yy = ys if loss == "squared" else np.where(ys > np.median(ys), 1.0, -1.0)
            ds = get_dataset(Xs, order="fortran")
            loss_obj = CLASSIFICATION_LOSSES[loss]
            reg = REGULARIZATION[regname]()
            P0 = 0.05 * np.random.RandomState(3).randn(k, d)
            lam = np.sign(np.random.RandomState(4).randn(k))
            y_pred = poly_predict(Xs, P0, lam, kernel="all-subsets")
            beta, gamma, eta = 5.0, 0.01, 0.5
            viols, losses = [], []
            jf = np.arange(d, dtype=np.int32)
            ic = np.arange(k, dtype=np.int32)
            if solver == "pcd":
                Pw = P0.copy()
                A = np.ones(n)
                reg.init_cache_pcd(-1, d, k)
            else:
                Pw = np.array(P0.T)
                A = np.ones((n, k))
                reg.init_cache_pbcd(-1, d, k)
                grad, inv_ss, p_old = np.zeros(k), np.zeros(k), np.zeros(k)
            for it in range(3):
                if solver == "pcd":
                    v = pcd_all.pcd_epoch(Pw, ds, yy, y_pred, lam, beta, gamma, eta, reg,
                                          loss_obj, A, ic, jf)
                else:
                    v = pbcd_all.pbcd_epoch(Pw, ds, yy, y_pred, lam, beta, gamma, eta, reg,
                                            loss_obj, A, grad, inv_ss, p_old, jf)
                viols.append(v)
                losses.append(sum(loss_obj.loss(y_pred[i], yy[i]) for i in range(n)))
            key = "%s|%s|%s" % (solver, regname, loss)
            names.append(key)
            out["sP0|" + key] = P0
            out["slams|" + key] = lam
            out["sviol|" + key] = np.array(viols)
            out["sloss|" + key] = np.array(losses)
            out["sP|" + key] = Pw if solver == "pcd" else np.array(Pw.T)
            out["sy_pred|" + key] = y_pred
    out["scases"] = np.array(names)
    out["smeta"] = np.array(json.dumps(dict(beta=5.0, gamma=0.01, eta0=0.5)))
    save("g8_all_subsets.npz", **out)


# --------------------------------------------------------------------- g9
def fit_psgd_verbose(est, X, y):
    """fit() with verbose=True; returns the per-epoch mean losses printed by _fit_psgd."""
    est.set_params(verbose=True)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf), warnings.catch_warnings():
        warnings.simplefilter("ignore")
        est.fit(X, y)
    vals = []
    for line in buf.getvalue().splitlines():
        if line.startswith("Epoch"):
            vals.append(float(line.split()[-1]))
    return np.array(vals)


def gen_g9():
    """psgd solver.
      cells  the reference's own test grid (tests/test_psgd.py:240-366): estimator.fit on
             the RandomState(1) data, all degree x batch_size x learning_rate x
             fit_linear x loss x regularizer combinations (gamma = 0 as in the reference)
      prox   regularizer.prox on seeded arrays for the strengths of tests/test_prox.py:54,69
      traj   sparse 300 x 60 problem with gamma > 0: per-epoch mean loss, it_, final P_/w_
    """
    from itertools import product
    from sparsepoly.regularizer import L1, L21, SquaredL12, SquaredL21

    out = {}
    # ---- cells
    rng = np.random.RandomState(1)
    n_components, n_features, n_samples = 5, 4, 50
    X = rng.randn(n_samples, n_features)
    P = rng.randn(n_components, n_features)
    lams = rng.randn(n_components)
    out["X"] = X
    names = []
    for degree in (2, 3, 4):
        y_reg = poly_predict(X, P, lams, kernel="anova", degree=degree)
        out["y|deg%d" % degree] = y_reg
        for batch_size, lr, fit_linear, loss, regname in product(
                [1, 5, 8, n_samples, "auto"], ["constant", "optimal"], [True, False],
                ["squared", "squared_hinge", "logistic"], ["l1", "l21", "squaredl12",
                                                           "squaredl21"]):
            kw = dict(degree=degree, n_components=n_components, fit_lower=None,
                      fit_linear=fit_linear, alpha=1e-3, beta=1e-3, gamma=0.0,
                      regularizer=regname, learning_rate=lr, eta0=0.01, warm_start=False,
                      tol=1e-3, max_iter=10, random_state=0, shuffle=False, solver="psgd",
                      batch_size=batch_size)
            if loss == "squared":
                est = SparseFactorizationMachineRegressor(**kw)
                y = y_reg
            else:
                est = SparseFactorizationMachineClassifier(loss=loss, **kw)
                y = np.sign(y_reg)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                est.fit(X, y)
            key = "deg%d|%s|%s|%d|%s|%s" % (degree, batch_size, lr, fit_linear, loss, regname)
            names.append(key)
            out["P|" + key] = est.P_
            out["w|" + key] = est.w_
            out["n_iter|" + key] = np.array([est.n_iter_, est.it_])
    out["cells"] = np.array(names)

    # ---- prox
    prng = np.random.RandomState(33)
    arr = 10 * (prng.rand(150, 12) * 2 - 1)       # (n_features, n_components)
    arr[7] = 0.0                                  # an all-zero row
    arr[9, 3] = 0.0
    arr[20:23] *= 1e-3                            # small-norm rows (l21 leaves them unchanged)
    out["prox_in"] = arr
    strengths = [0.0, 0.0001, 0.001, 0.01, 0.1, 1, 10]
    out["prox_strengths"] = np.array(strengths)
    for regname, cls in (("l1", L1), ("l21", L21), ("squaredl12", SquaredL12),
                         ("squaredl21", SquaredL21)):
        for si, st in enumerate(strengths):
            reg = cls()
            reg.init_cache_psgd(2, arr.shape[0], arr.shape[1])
            Pq = np.array(arr)
            with warnings.catch_warnings(), np.errstate(all="ignore"):
                warnings.simplefilter("ignore")
                reg.prox(Pq, st, 2)
            out["prox|%s|%d" % (regname, si)] = Pq

    # ---- trajectories on a sparse problem
    Xs, ys = small_problem()
    n, d = Xs.shape
    out.update({"X_data": Xs.data, "X_indices": Xs.indices, "X_indptr": Xs.indptr,
                "X_shape": np.array(Xs.shape), "ys": ys})
    tnames = []
    tcases = [
        # tag, reg, degree, k, fit_lower, lr, batch, shuffle, gamma, power_t
        ("l1", "l1", 2, 8, "explicit", "optimal", "auto", False, 1e-2, 1.0),
        ("l21", "l21", 2, 8, "explicit", "constant", 16, False, 3e-2, 1.0),
        ("sq12", "squaredl12", 2, 8, "explicit", "optimal", 32, False, 1e-2, 1.0),
        ("sq21", "squaredl21", 2, 8, "explicit", "invscaling", 25, False, 1e-2, 0.5),
        ("l1d3", "l1", 3, 6, "explicit", "pegasos", 20, False, 1e-3, 1.0),
        ("sq12d3", "squaredl12", 3, 6, None, "optimal", 300, False, 1e-2, 1.0),
        ("l21sh", "l21", 2, 8, "explicit", "optimal", 7, True, 3e-2, 0.75),
        ("sq21sh", "squaredl21", 3, 5, "explicit", "constant", 1, True, 1e-3, 1.0),
    ]
    for tag, regname, degree, k, fit_lower, lr, batch, shuffle, gamma, power_t in tcases:
        n_orders = degree - 1 if fit_lower == "explicit" else 1
        P0 = 0.1 * np.random.RandomState(0).randn(n_orders, k, d)
        lam = np.sign(np.random.RandomState(5).randn(k))
        # pegasos uses eta = 1 / (beta * it): needs a large beta to be stable
        alpha, beta = (20.0, 20.0) if lr == "pegasos" else (1e-2, 0.1)
        for loss in ("squared", "squared_hinge", "logistic"):
            yy = ys if loss == "squared" else np.where(ys > np.median(ys), 1.0, -1.0)
            kw = dict(degree=degree, n_components=k, fit_lower=fit_lower, fit_linear=True,
                      alpha=alpha, beta=beta, gamma=gamma, regularizer=regname,
                      learning_rate=lr, eta0=0.05, power_t=power_t, warm_start=True, tol=-1.0,
                      n_iter_no_change=1000, max_iter=4, random_state=3, shuffle=shuffle,
                      solver="psgd", batch_size=batch)
            if loss == "squared":
                est = SparseFactorizationMachineRegressor(**kw)
            else:
                est = SparseFactorizationMachineClassifier(loss=loss, **kw)
            est.P_ = np.array(P0)
            est.w_ = np.zeros(d)
            est.lams_ = np.array(lam)
            losses = fit_psgd_verbose(est, Xs, yy)
            key = "%s|%s" % (tag, loss)
            tnames.append(key)
            out["tmeta|" + key] = np.array(json.dumps(dict(
                regularizer=regname, degree=degree, k=k, fit_lower=fit_lower,
                learning_rate=lr, batch_size=batch, shuffle=shuffle, gamma=gamma,
                power_t=power_t, loss=loss, alpha=alpha, beta=beta, eta0=0.05, max_iter=4,
                random_state=3)))
            out["tP0|" + key] = P0
            out["tlams|" + key] = lam
            out["tloss|" + key] = losses
            out["tP|" + key] = est.P_
            out["tw|" + key] = est.w_
            out["tit|" + key] = np.array([est.n_iter_, est.it_])
            assert np.isfinite(est.P_).all() and np.isfinite(losses).all(), key
            print(key, "loss", losses, "nnz(P)=%.2f" % np.mean(est.P_ != 0),
                  "|P|max=%.3g" % np.abs(est.P_).max())
    out["tcases"] = np.array(tnames)
    save("g9_psgd.npz", **out)


# -------------------------------------------------------------------- g10
def gen_g10():
    """eval() of the six regularizers (regularizer/*.py: l1.py:17-18, l21.py:19-21,
    squaredl12.py:20-22, squaredl21.py:23-25, omegati.py:19-47, omegacs.py:22-39)."""
    rng = np.random.RandomState(10)
    out = {}
    P2 = rng.randn(7, 4)          # (n_features, n_components) in the psgd / pbcd convention
    P3 = rng.randn(3, 7, 4)       # a stack (e.g. the orders of P_)
    out["P2"], out["P3"] = P2, P3
    for name, P in (("P2", P2), ("P3", P3)):
        out["l1|%s" % name] = np.asarray(REGULARIZATION["l1"]().eval(P, 2))
        for tr in (False, True):
            out["l21|%s|t%d" % (name, tr)] = np.asarray(REGULARIZATION["l21"](tr).eval(P))
            out["squaredl12|%s|t%d" % (name, tr)] = np.asarray(
                REGULARIZATION["squaredl12"](tr).eval(P))
            out["squaredl21|%s|t%d" % (name, tr)] = np.asarray(
                REGULARIZATION["squaredl21"](tr).eval(P))
        for deg in (2, 3, 4, -1):
            out["omegati|%s|deg%d" % (name, deg)] = np.asarray(
                REGULARIZATION["omegati"]().eval(P, deg))
        for deg in (2, 3, 4):
            out["omegacs|%s|deg%d" % (name, deg)] = np.asarray(
                REGULARIZATION["omegacs"]().eval(P, deg))
    save("g10_reg_eval.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g10"]
    for g in which:
        globals()["gen_" + g]()
