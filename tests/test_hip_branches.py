"""The reference's "numerical error" branches on the device, FORCED (not met by luck):

  omegati.py:97-98     _dcache[t] = max(_cache[t-1] - _dcache[t-1] |p_j|, 0)   (clip active)
  omegacs.py:90-96     a negative _dcache entry -> recompute with degree - 1
  omegacs.py:75-76     a negative _cache entry after update_cache_pbcd -> recompute
  squaredl21.py:48-49  _cache < _norms[j] -> re-sum the norms

Construction: two EMPTY columns a, b (their gradient is zero, so with beta = eta = 1 the update
drives them to exactly 0) whose norms n_a, n_b are exactly representable (vectors (3t, 4t):
norm 5t) and chosen so that fl(fl(n_a + n_b) - n_a) is BELOW n_b ("neg": the running cache minus
n_b goes negative at column b) or ABOVE it ("pos": the second-order cache goes negative after
column a).  A third column carries the data and starts at 0, so the initial sums do not depend on
the summation order.  The device's branch counters (spfm_debug_branch_counts) must tick, and the
result must equal the oracle's, for the persistent and the multi-kernel engine.
Needs a real MI355X: ``pytest -m gpu``."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

# t values found by search: n = 5 t; (n_a + n_b) - n_a - n_b = -8.1e-10 ("neg") / +2.4e-9 ("pos")
PAIRS = {"neg": (6206275.0, 0.0013605052372440696), "pos": (9374380.0, 0.001622503506951034)}


def _problem(kind, k):
    ta, tb = PAIRS[kind]
    na, nb = 5 * ta, 5 * tb
    r = (na + nb) - na
    assert (r < nb) if kind == "neg" else (r > nb)
    rng = np.random.RandomState(3)
    n, d = 6, 3
    X = np.zeros((n, d))
    X[:, 2] = rng.randn(n)
    y = rng.randn(n)
    P0 = np.zeros((1, k, d))
    if k == 1:  # pcd: |p| itself is the quantity that is summed
        P0[0, 0, 0], P0[0, 0, 1] = na, -nb
    else:
        P0[0, :2, 0] = (3 * ta, 4 * ta)
        P0[0, :2, 1] = (-3 * tb, 4 * tb)
    return sp.csc_matrix(X), y, P0


def _run_engine(X, y, P0, solver, reg, degree, options):
    from sparsepoly_amd.engine import HipEngine

    d, k = X.shape[1], P0.shape[1]
    eng = HipEngine(0, "f64")
    for key, val in options.items():
        eng.set_option(key, val)
    eng.set_data(X, y)
    eng.set_params(P0, np.zeros(d), np.ones(k))
    eng.configure(solver, "squared", reg, degree)
    eng.init_pred(degree, False, False)
    y0 = eng.get_y_pred()
    eng.set_schedule("exact", np.arange(d, dtype=np.int32))
    eng.debug_branch_counts(reset=True)
    if solver == "pcd":
        v = eng.pcd_epoch(0, degree, 1.0, 0.05, 1.0, np.arange(k, dtype=np.int32))
    else:
        v = eng.pbcd_epoch(0, degree, 1.0, 0.05, 1.0)
    counts = eng.debug_branch_counts(reset=True)
    P, _ = eng.get_params()
    yp = eng.get_y_pred()
    eng.close()
    return v, P, yp, y0, counts


def _run_oracle(oracle, X, y, P0, y0, solver, reg, degree):
    n, d = X.shape
    k = P0.shape[1]
    ds = oracle.CSC(X)
    regc = oracle.Regularizer(reg)
    yp = np.ascontiguousarray(y0.copy())
    jf = np.arange(d, dtype=np.int32)
    if solver == "pcd":
        regc.init_cache_pcd(degree, d, k)
        P = np.ascontiguousarray(P0[0].copy())
        A = np.zeros((n, degree + 1))
        v = oracle.pcd_epoch(P, ds, y, yp, np.ones(k), degree, 1.0, 0.05, 1.0, regc, "squared", A,
                             np.arange(k, dtype=np.int32), jf)
        return v, P, yp
    regc.init_cache_pbcd(degree, d, k)
    Pt = np.ascontiguousarray(P0[0].T.copy())
    A = np.zeros((n, degree + 1, k))
    dA = np.zeros((n, degree, k))
    v = oracle.pbcd_epoch(Pt, ds, y, yp, np.ones(k), degree, 1.0, 0.05, 1.0, regc, "squared", A, dA,
                          jf)
    return v, np.ascontiguousarray(Pt.T), yp


CASES = [
    # solver, regularizer, degree, k, pair, counter that must tick
    ("pbcd", "omegacs", 2, 3, "neg", "omegacs_dcache"),
    ("pbcd", "omegacs", 2, 3, "pos", "omegacs_cache"),
    # degree 3: the third-order cache goes negative first, in the update after column a
    ("pbcd", "omegacs", 3, 3, "neg", "omegacs_cache"),
    ("pbcd", "squaredl21", 2, 3, "neg", "squaredl21_resum"),
    ("pcd", "omegati", 2, 1, "neg", "omegati_clip"),
    ("pcd", "omegati", 3, 1, "neg", "omegati_clip"),
]


@pytest.mark.parametrize("engine", ["persistent", "multi_kernel"])
@pytest.mark.parametrize("solver,reg,degree,k,pair,counter", CASES)
def test_forced_numerical_error_branch(oracle, solver, reg, degree, k, pair, counter, engine):
    X, y, P0 = _problem(pair, k)
    if degree == 3:  # P_ has one order per degree (fit_lower='explicit'); the top order is used
        P0 = np.concatenate([P0, np.zeros_like(P0)], axis=0)
    options = {} if engine == "persistent" else {"persistent": 0, "pbcd_persistent": 0}
    v, P, yp, y0, counts = _run_engine(X, y, P0, solver, reg, degree, options)
    assert counts[counter] > 0, (counts, "the forced branch did not run on the device")
    vo, Po, ypo = _run_oracle(oracle, X, y, P0, y0, solver, reg, degree)
    np.testing.assert_allclose(v, vo, rtol=1e-9)
    np.testing.assert_allclose(P[0], Po, rtol=0, atol=1e-8 * max(1.0, np.abs(Po).max()))
    np.testing.assert_allclose(yp, ypo, rtol=0, atol=1e-7)
    # the two emptied columns are exactly zero, as in the reference
    assert np.all(P[0][:, :2] == 0.0)
