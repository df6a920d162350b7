"""The regularizer plug-in surface (reference: regularizer/__init__.py:8-15, base.py:27-34 -- any
object with the duck-typed protocol can be registered under a name).  The six built-ins run inside
the device chains; anything else is honoured through HOST-STEPPED epochs: per dependent step the
device forms the column sums and scatter-updates, the object's prox_cd / prox_bcd and cache hooks
run on the host in visiting order (include/spfm.h).  Checked (a) with subclasses of the built-ins
-- same mathematics through the other path, including the stateful caches -- against the device
path, and (b) with a regularizer the library does not know (elastic net) against a NumPy
restatement of pcd.py.  Needs a real MI355X."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _data(n=400, d=40, density=0.15, seed=3):
    rng = np.random.RandomState(seed)
    X = sp.random(n, d, density=density, random_state=rng, data_rvs=rng.randn, format="csr")
    y = rng.randn(n)
    return X, y


def _fit(cls, X, y, regname, registry=None, **kw):
    class Est(cls):
        pass

    if registry is not None:
        Est._REGULARIZERS = dict(cls._REGULARIZERS, **registry)
    Est.__name__ = cls.__name__
    est = Est(regularizer=regname, precision="f64", device=0, tol=0, random_state=0, **kw)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        est.fit(X, y)
    return est


@pytest.mark.parametrize("schedule", ["exact", "colored"])
@pytest.mark.parametrize("solver,base,degree", [("pcd", "l1", 2), ("pcd", "squaredl12", 2),
                                                ("pcd", "omegati", 3), ("pbcd", "l21", 2),
                                                ("pbcd", "squaredl21", 2), ("pbcd", "omegacs", 3)])
def test_subclassed_builtin_runs_host_stepped_and_equals_the_device_path(solver, base, degree,
                                                                         schedule):
    from sparsepoly_amd import SparseFactorizationMachineRegressor as Reg
    from sparsepoly_amd.regularizer import REGULARIZATION

    class Mine(REGULARIZATION[base]):  # same protocol, not a built-in class: host-stepped
        calls = 0

        def prox_cd(self, *a):
            type(self).calls += 1
            return super().prox_cd(*a)

        def prox_bcd(self, *a):
            type(self).calls += 1
            return super().prox_bcd(*a)

    X, y = _data()
    kw = dict(degree=degree, n_components=3, solver=solver, alpha=0.1,
              beta=10.0 if solver == "pcd" else 1.0, gamma=0.02, max_iter=3, schedule=schedule)
    a = _fit(Reg, X, y, base, **kw)
    b = _fit(Reg, X, y, "mine", registry={"mine": Mine}, **kw)
    assert Mine.calls > 0 and b._plugin_reg is not None and a._plugin_reg is None
    np.testing.assert_array_equal(a.feature_order_, b.feature_order_)
    np.testing.assert_allclose(b.P_, a.P_, rtol=0, atol=1e-10)
    np.testing.assert_allclose(b.w_, a.w_, rtol=0, atol=1e-10)
    np.testing.assert_allclose(b.predict(X), a.predict(X), rtol=0, atol=1e-9)


class ElasticNet(object):
    """gamma (|p| + rho/2 p^2): prox = soft-threshold, then shrink.  Stateless; pcd protocol."""
    rho = 3.0

    def init_cache_pcd(self, degree, n_features, n_components):
        self.seen = 0

    def compute_cache_pcd_all(self, P, degree):
        pass

    def compute_cache_pcd(self, P, degree, s):
        pass

    def prox_cd(self, p, strength, degree, j):
        self.seen += 1
        return np.sign(p) * max(abs(p) - strength, 0.0) / (1.0 + self.rho * strength)

    def update_cache_pcd(self, P, degree, s, j):
        pass


def _slow_pcd(X, y, P, lams, beta, gamma, eta, prox, epochs):
    """pcd.py:71-137 at degree 2, squared loss, no linear term, natural order (dense NumPy)."""
    k, d = P.shape
    XP = X @ P.T
    y_pred = 0.5 * ((XP ** 2) - (X ** 2) @ (P ** 2).T) @ lams
    for _ in range(epochs):
        for s in range(k):
            A1 = X @ P[s]
            for j in range(d):
                x = X[:, j]
                dA = x * (A1 - P[s, j] * x)
                g = float(((y_pred - y) * dA).sum())
                h = float((dA * dA).sum())
                inv = h + beta
                upd = (lams[s] * g + beta * P[s, j]) / inv
                pn = prox(P[s, j] - eta * upd, eta * gamma / inv)
                delta = P[s, j] - pn
                P[s, j] = pn
                A1 -= delta * x
                y_pred -= lams[s] * delta * dA
    return P


def test_unknown_regularizer_object_equals_numpy_restatement():
    from sparsepoly_amd import SparseFactorizationMachineRegressor as Reg

    X, y = _data(n=300, d=25, density=0.3, seed=5)
    kw = dict(degree=2, n_components=3, solver="pcd", beta=5.0, gamma=0.05, max_iter=3,
              fit_linear=False, fit_lower=None, schedule="exact")
    est = _fit(Reg, X, y, "enet", registry={"enet": ElasticNet}, **kw)
    assert est._plugin_reg.seen == 3 * 3 * 25
    P0 = 0.01 * np.random.RandomState(0).randn(1, 3, 25)
    en = ElasticNet()
    en.init_cache_pcd(2, 25, 3)
    ref = _slow_pcd(X.toarray(), y, P0[0].copy(), np.ones(3), 5.0, 0.05, 1.0,
                    lambda p, st: en.prox_cd(p, st, 2, 0), 3)
    np.testing.assert_allclose(est.P_[0], ref, rtol=0, atol=1e-10)
    assert 0.1 < (est.P_ != 0).mean() < 1.0  # the threshold did something, not everything


def test_plugin_regularizer_limits():
    from sparsepoly_amd import SparseFactorizationMachineRegressor as Reg

    X, y = _data()
    with pytest.raises(ValueError):
        _fit(Reg, X, y, "enet", registry={"enet": ElasticNet}, solver="psgd", max_iter=1)
    with pytest.raises(ValueError):  # unknown names still fail as in the reference (base.py:28-33)
        _fit(Reg, X, y, "nope", max_iter=1)
