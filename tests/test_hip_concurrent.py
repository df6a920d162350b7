"""Independent fits side by side on one GPU (sparsepoly_amd/concurrent.py, engine option
`co_tenants`): every fit must equal its solo run bit for bit -- the concurrency is between whole
fits (one handle, stream and host thread each), never inside one."""
import threading

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _problem(n=6000, d=400, per_row=8, seed=0):
    from sparsepoly_amd.synth import make_problem

    return make_problem(n, d, per_row, seed=seed)


def _engine_run(X, y, solver, reg, degree, gamma, tenants, iters=2, k=5):
    from sparsepoly_amd import engine as E

    with E.co_tenancy(tenants):
        eng = E.HipEngine(0, "f32")
    assert eng.get_option("co_tenants") == tenants
    d = X.shape[1]
    eng.set_data(X, y)
    eng.set_params(0.01 * np.random.RandomState(1).randn(1, k, d), np.zeros(d), np.ones(k))
    eng.configure(solver, "squared", reg, degree)
    eng.init_pred(degree, True, False)
    eng.set_schedule("colored", np.arange(d, dtype=np.int32))
    return eng


def _iterate(eng, solver, degree, gamma, iters, out, i, barrier=None):
    k = eng.k
    if barrier is not None:
        barrier.wait()
    v = []
    for _ in range(iters):
        a = eng.cd_linear_epoch(1.0)
        if solver == "pcd":
            a += eng.pcd_epoch(0, degree, 5.0, gamma, 1.0, np.arange(k, dtype=np.int32))
        else:
            a += eng.pbcd_epoch(0, degree, 5.0, gamma, 1.0)
        v.append(a)
    P, w = eng.get_params()
    out[i] = (np.array(v), P, w, eng.get_y_pred(), eng.get_option("persistent_fallbacks"))


@pytest.mark.parametrize("solver,reg,degree", [("pcd", "squaredl12", 2), ("pcd", "omegati", 3),
                                              ("pbcd", "omegacs", 2), ("pcd", "l1", 2)])
def test_four_engines_at_once_equal_their_solo_runs(solver, reg, degree):
    X, y = _problem()
    gammas = [1e-3, 3e-3, 1e-2, 3e-4]
    solo = [None] * 4
    for f in range(4):   # alone on the GPU, with the same share of the CUs (same sum order)
        eng = _engine_run(X, y, solver, reg, degree, gammas[f], 4)
        _iterate(eng, solver, degree, gammas[f], 3, solo, f)
        eng.close()
    whole = [None]           # ... and with the whole GPU: equal up to the order of the sums
    eng = _engine_run(X, y, solver, reg, degree, gammas[0], 1)
    _iterate(eng, solver, degree, gammas[0], 3, whole, 0)
    eng.close()
    for a, b in zip(solo[0][:4], whole[0][:4]):
        np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-9)
    engs = [_engine_run(X, y, solver, reg, degree, gammas[f], 4) for f in range(4)]
    together = [None] * 4
    barrier = threading.Barrier(4)
    th = [threading.Thread(target=_iterate,
                           args=(engs[f], solver, degree, gammas[f], 3, together, f, barrier))
          for f in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    for eng in engs:
        eng.close()
    for f in range(4):
        assert together[f] is not None
        for a, b in zip(solo[f][:4], together[f][:4]):
            assert np.array_equal(a, b)
        assert together[f][4] == 0     # no pass had to be redone on the multi-kernel engine


def test_co_tenants_caps_the_workgroup_counts():
    from sparsepoly_amd import engine as E

    eng = E.HipEngine(0, "f32")
    full = {k: eng.get_option(k) for k in ("prb_groups", "pbprb_groups")}
    eng.set_option("co_tenants", 4)
    assert eng.get_option("prb_groups") == min(full["prb_groups"], 64)
    assert eng.get_option("pbprb_groups") == min(full["pbprb_groups"], 64)
    with pytest.raises(ValueError):
        eng.set_option("co_tenants", 0)
    eng.close()


def test_fit_path_equals_solo_fits():
    from sklearn.base import clone

    from sparsepoly_amd import SparseFactorizationMachineRegressor

    X, y = _problem(4000, 300, 8, seed=2)
    base = SparseFactorizationMachineRegressor(n_components=4, max_iter=4, tol=0, beta=1.0,
                                               random_state=0, schedule="colored")
    gammas = [1e-2, 1e-3, 1e-4, 1e-5, 3e-3, 3e-4]      # six fits through four threads
    import warnings

    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        path = base.fit_path(X, y, gamma=gammas)
        assert not hasattr(base, "P_")
        assert [e.gamma for e in path] == gammas
        for e in path:
            s = clone(base).set_params(gamma=e.gamma).fit(X, y)   # default pass: 64 row blocks
            assert np.array_equal(s.P_, e.P_) and np.array_equal(s.w_, e.w_)
            assert s.n_iter_ == e.n_iter_
            assert np.array_equal(s.predict(X), e.predict(X))
    # sparser models towards the strong end of the path
    nz = [np.count_nonzero(e.P_) for e in path[:4]]
    assert nz[0] <= nz[3]


def test_fit_concurrently_with_one_data_set_per_estimator_and_classifier():
    import warnings

    from sparsepoly_amd import (SparseFactorizationMachineClassifier,
                                SparseFactorizationMachineRegressor)
    from sparsepoly_amd.concurrent import fit_concurrently

    X, y = _problem(3000, 200, 8, seed=3)
    folds = [(X[:2000], y[:2000]), (X[1000:], y[1000:]), (X[500:2500], y[500:2500])]
    kw = dict(n_components=3, max_iter=3, tol=0, random_state=1, schedule="colored")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        ests = [SparseFactorizationMachineRegressor(solver="pbcd", regularizer="omegacs", **kw),
                SparseFactorizationMachineClassifier(loss="squared_hinge", **kw),
                SparseFactorizationMachineRegressor(degree=3, regularizer="omegati", **kw)]
        Xs = [f[0] for f in folds]
        ys = [folds[0][1], np.where(folds[1][1] > 0, 1, -1), folds[2][1]]
        got = fit_concurrently(ests, Xs, ys)
        from sklearn.base import clone

        from sparsepoly_amd.engine import co_tenancy

        for e, Xf, yf in zip(got, Xs, ys):
            with co_tenancy(3):          # alone, with the same share of the CUs
                s = clone(e).fit(Xf, yf)
            assert np.array_equal(s.P_, e.P_) and np.array_equal(s.w_, e.w_)
            s = clone(e).fit(Xf, yf)     # the whole GPU: another order of the partial sums
            np.testing.assert_allclose(s.P_, e.P_, rtol=1e-6, atol=1e-9)


def test_errors_of_a_fit_reach_the_caller():
    from sparsepoly_amd import SparseFactorizationMachineRegressor
    from sparsepoly_amd.concurrent import fit_concurrently, fit_path

    X, y = _problem(500, 40, 5, seed=4)
    good = SparseFactorizationMachineRegressor(max_iter=1, n_components=2)
    bad = SparseFactorizationMachineRegressor(max_iter=1, regularizer="nope")
    with pytest.raises(ValueError):
        fit_concurrently([good, bad], X, y)
    with pytest.raises(ValueError):
        fit_path(good, X, y, gamma=[1.0, 2.0], beta=[1.0])
    with pytest.raises(ValueError):
        fit_path(good, X, y, no_such_parameter=[1.0])
    with pytest.raises(ValueError):
        fit_path(good, X, y)
    with pytest.raises(ValueError):
        fit_concurrently([good], [X, X], [y, y])
    assert fit_concurrently([], X, y) == []


def test_all_subsets_path_equals_solo_fits():
    import warnings

    from sklearn.base import clone

    from sparsepoly_amd import SparseAllSubsetsRegressor

    X, y = _problem(2500, 150, 6, seed=6)
    base = SparseAllSubsetsRegressor(n_components=3, max_iter=3, tol=0, beta=1.0, random_state=0,
                                     schedule="colored")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        path = base.fit_path(X, y, gamma=[1e-2, 1e-3, 1e-4])
        for e in path:
            s = clone(base).set_params(gamma=e.gamma).fit(X, y)
            assert np.array_equal(s.P_, e.P_)
            assert np.array_equal(s.predict(X), e.predict(X))


def test_a_second_fit_on_the_same_matrix_reuses_the_colouring():
    """The process-wide memory of coloured schedules (engine.shared_schedule): same structure,
    same visiting order -> installed instead of computed; another matrix, another order or
    shuffle=True -> computed.  The fits equal those of a process that never remembered anything."""
    import warnings

    from sparsepoly_amd import SparseFactorizationMachineRegressor
    from sparsepoly_amd import engine as E

    X, y = _problem(3000, 250, 8, seed=9)
    kw = dict(n_components=3, max_iter=3, tol=0, beta=1.0, random_state=0, schedule="colored")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        E._SCHEDULE_LRU.clear()
        E._SCHEDULE_STATS.update(hits=0, misses=0)
        a = SparseFactorizationMachineRegressor(gamma=1e-3, **kw).fit(X, y)
        assert E._SCHEDULE_STATS == {"hits": 0, "misses": 1}
        b = SparseFactorizationMachineRegressor(gamma=1e-4, **kw).fit(X, y)        # same matrix
        assert E._SCHEDULE_STATS == {"hits": 1, "misses": 1}
        Xv = X.copy()
        Xv.data = Xv.data * 2.0                                    # same structure, other values
        SparseFactorizationMachineRegressor(gamma=1e-4, **kw).fit(Xv, y)
        assert E._SCHEDULE_STATS == {"hits": 2, "misses": 1}
        SparseFactorizationMachineRegressor(gamma=1e-4, **kw).fit(X[:2000], y[:2000])  # another one
        assert E._SCHEDULE_STATS == {"hits": 2, "misses": 2}
        SparseFactorizationMachineRegressor(gamma=1e-4, shuffle=True, **kw).fit(X, y)
        assert E._SCHEDULE_STATS == {"hits": 2, "misses": 2}       # shuffled orders are not kept
        E._SCHEDULE_LRU.clear()
        b2 = SparseFactorizationMachineRegressor(gamma=1e-4, **kw).fit(X, y)       # computed afresh
    assert np.array_equal(b.P_, b2.P_) and np.array_equal(b.w_, b2.w_)
    assert np.array_equal(a.feature_order_, b.feature_order_)
    assert b.n_steps_per_sweep_ == b2.n_steps_per_sweep_
