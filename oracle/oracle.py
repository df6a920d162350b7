"""CPU oracle for the sparse-FM proximal coordinate-descent hot path.

TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Importable only from ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg.  The product
package ``sparsepoly_amd`` never imports this module.

Parity status: PINNED against fixtures produced by the reference's own source
(``oracle/gen_golden.py`` -> ``tests/golden/*.npz``; checked by
``tests/test_oracle_golden.py``).

What is here
------------
* ctypes bindings to ``libspfm_oracle.so`` (``spfm_oracle.c``): float64,
  single-thread restatements of ``_cd_linear_epoch`` (reference
  ``sparsepoly/optimizer/cd_linear.py:8-33``), ``pcd_epoch``
  (``optimizer/pcd.py:71-137``), ``pbcd_epoch`` (``optimizer/pbcd.py:82-148``),
  the six regularizers' CD/BCD protocol (``regularizer/*.py``) and the three
  losses (``loss.py:13-71``).
* ``anova_kernel`` / ``poly_predict``: NumPy restatement of
  ``sparsepoly/kernels.py:43-48,71-115,140-153``.
* ``OracleFM``: the epoch drivers ``_fit_pcd`` / ``_fit_pbcd`` / ``fit`` /
  ``_get_output`` (``sparse_factorization_machines.py:175-451``) with per-epoch
  recording and optional explicit coordinate orders (needed to replay the
  conflict-free batched order of the HIP engine).
"""
import ctypes as C
import os
import subprocess

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libspfm_oracle.so")

LOSSES = {"squared": 0, "squared_hinge": 1, "logistic": 2}
REGULARIZERS = {
    "l1": 0,
    "l21": 1,
    "squaredl12": 2,
    "squaredl21": 3,
    "omegati": 4,
    "omegacs": 5,
}

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_lp = C.POINTER(C.c_int64)


def build(force=False):
    """Compile spfm_oracle.c with the committed Makefile (gcc, no fast-math)."""
    src = os.path.join(_HERE, "spfm_oracle.c")
    if (
        force
        or not os.path.exists(_LIB_PATH)
        or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)
    ):
        subprocess.check_call(["make", "-C", _HERE, "-s", "libspfm_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    L.spo_loss_mu.restype = C.c_double
    L.spo_loss_mu.argtypes = [C.c_int]
    L.spo_dloss.restype = C.c_double
    L.spo_dloss.argtypes = [C.c_int, C.c_double, C.c_double]
    L.spo_loss.restype = C.c_double
    L.spo_loss.argtypes = [C.c_int, C.c_double, C.c_double]
    L.spo_dloss_vec.restype = None
    L.spo_dloss_vec.argtypes = [C.c_int, C.c_int64, _dp, _dp, _dp]
    L.spo_loss_sum.restype = C.c_double
    L.spo_loss_sum.argtypes = [C.c_int, C.c_int64, _dp, _dp]
    L.spo_reg_create.restype = C.c_void_p
    L.spo_reg_create.argtypes = [C.c_int]
    L.spo_reg_destroy.restype = None
    L.spo_reg_destroy.argtypes = [C.c_void_p]
    for nm in ("spo_reg_init_cache_pcd", "spo_reg_init_cache_pbcd"):
        f = getattr(L, nm)
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.spo_reg_compute_cache_pcd.restype = None
    L.spo_reg_compute_cache_pcd.argtypes = [C.c_void_p, _dp, C.c_int, C.c_int]
    L.spo_reg_update_cache_pcd.restype = None
    L.spo_reg_update_cache_pcd.argtypes = [C.c_void_p, _dp, C.c_int, C.c_int, C.c_int]
    L.spo_reg_prox_cd.restype = C.c_double
    L.spo_reg_prox_cd.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_int, C.c_int]
    L.spo_reg_compute_cache_pbcd.restype = None
    L.spo_reg_compute_cache_pbcd.argtypes = [C.c_void_p, _dp, C.c_int]
    L.spo_reg_update_cache_pbcd.restype = None
    L.spo_reg_update_cache_pbcd.argtypes = [C.c_void_p, _dp, C.c_int, C.c_int]
    L.spo_reg_prox_bcd.restype = None
    L.spo_reg_prox_bcd.argtypes = [C.c_void_p, _dp, C.c_double, C.c_int, C.c_int]
    L.spo_reg_get_state.restype = None
    L.spo_reg_get_state.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
    L.spo_reg_set_state.restype = None
    L.spo_reg_set_state.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
    L.spo_cd_linear_epoch.restype = C.c_double
    L.spo_cd_linear_epoch.argtypes = [
        _dp, C.c_int64, C.c_int, _lp, _ip, _dp, _dp, _dp, _dp, C.c_double, C.c_int, _ip,
        C.c_int,
    ]
    L.spo_pcd_precompute_A.restype = None
    L.spo_pcd_precompute_A.argtypes = [
        C.c_int64, C.c_int, _lp, _ip, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int,
    ]
    L.spo_pcd_epoch.restype = C.c_double
    L.spo_pcd_epoch.argtypes = [
        _dp, C.c_int, C.c_int64, C.c_int, _lp, _ip, _dp, _dp, _dp, _dp, C.c_int, C.c_double,
        C.c_double, C.c_double, C.c_void_p, C.c_int, _dp, C.c_int, _ip, C.c_int, _ip, C.c_int,
    ]
    L.spo_pbcd_precompute_A.restype = None
    L.spo_pbcd_precompute_A.argtypes = [
        C.c_int64, C.c_int, _lp, _ip, _dp, _dp, C.c_int, _dp, C.c_int, C.c_int,
    ]
    L.spo_pbcd_epoch.restype = C.c_double
    L.spo_pbcd_epoch.argtypes = [
        _dp, C.c_int, C.c_int64, C.c_int, _lp, _ip, _dp, _dp, _dp, _dp, C.c_int, C.c_double,
        C.c_double, C.c_double, C.c_void_p, C.c_int, _dp, _dp, C.c_int, _ip, C.c_int,
    ]
    L.spo_pcd_all_epoch.restype = C.c_double
    L.spo_pcd_all_epoch.argtypes = [
        _dp, C.c_int, C.c_int64, C.c_int, _lp, _ip, _dp, _dp, _dp, _dp, C.c_double, C.c_double,
        C.c_double, C.c_void_p, C.c_int, _dp, _ip, C.c_int, _ip, C.c_int,
    ]
    L.spo_pbcd_all_epoch.restype = C.c_double
    L.spo_pbcd_all_epoch.argtypes = [
        _dp, C.c_int, C.c_int64, C.c_int, _lp, _ip, _dp, _dp, _dp, _dp, C.c_double, C.c_double,
        C.c_double, C.c_void_p, C.c_int, _dp, _ip, C.c_int,
    ]
    L.spo_all_subsets_predict_csr.restype = None
    L.spo_all_subsets_predict_csr.argtypes = [
        C.c_int64, _lp, _ip, _dp, _dp, C.c_int, C.c_int, _dp, _dp,
    ]
    L.spo_anova_predict_csr.restype = None
    L.spo_anova_predict_csr.argtypes = [
        C.c_int64, _lp, _ip, _dp, _dp, C.c_int, C.c_int, _dp, C.c_int, _dp,
    ]
    L.spo_prox_squaredl12.restype = None
    L.spo_prox_squaredl12.argtypes = [_dp, C.c_int64, C.c_int64, C.c_double]
    L.spo_reg_prox.restype = C.c_int
    L.spo_reg_prox.argtypes = [C.c_int, _dp, C.c_int, C.c_int, C.c_double]
    L.spo_psgd_get_eta.restype = None
    L.spo_psgd_get_eta.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                   C.c_int64, _dp, _dp]
    L.spo_psgd_epoch.restype = C.c_double
    L.spo_psgd_epoch.argtypes = [
        _dp, _dp, _dp, C.c_int, C.c_int, C.c_int64, C.c_int, _lp, _ip, _dp, _dp, C.c_int,
        C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, _ip, C.c_int, C.c_double,
        C.c_int, C.c_double, C.c_int64, C.POINTER(C.c_int64),
    ]
    _lib = L
    return L


def _d(a):
    assert a.dtype == np.float64 and a.flags.c_contiguous
    return a.ctypes.data_as(_dp)


def _i(a):
    assert a.dtype == np.int32 and a.flags.c_contiguous
    return a.ctypes.data_as(_ip)


def _l(a):
    assert a.dtype == np.int64 and a.flags.c_contiguous
    return a.ctypes.data_as(_lp)


class CSC(object):
    """Column view of X as the reference's CSCDataset (dataset.py:94-116).

    Dense input follows FortranDataset (dataset.py:39-57): every column lists
    all n rows, explicit zeros included.
    """

    def __init__(self, X):
        if sp.issparse(X):
            Xc = sp.csc_matrix(X, dtype=np.float64)
            Xc.sort_indices()
            self.indptr = np.ascontiguousarray(Xc.indptr, dtype=np.int64)
            self.indices = np.ascontiguousarray(Xc.indices, dtype=np.int32)
            self.data = np.ascontiguousarray(Xc.data, dtype=np.float64)
        else:
            Xf = np.asarray(X, dtype=np.float64)
            n, d = Xf.shape
            self.indptr = np.arange(0, (d + 1) * n, n, dtype=np.int64)
            self.indices = np.ascontiguousarray(np.tile(np.arange(n, dtype=np.int32), d))
            self.data = np.ascontiguousarray(Xf.T.reshape(-1))
        self.n, self.d = X.shape


class Regularizer(object):
    """Handle on a C-side regularizer (regularizer/*.py jitclass instances)."""

    def __init__(self, name):
        if name not in REGULARIZERS:
            raise ValueError("Regularizer %s not supported." % name)
        self.name = name
        self.kind = REGULARIZERS[name]
        self._h = lib().spo_reg_create(self.kind)
        self.top_degree = None
        self.d = None

    def __del__(self):
        try:
            lib().spo_reg_destroy(self._h)
        except Exception:
            pass

    @staticmethod
    def _check(rc, name, solver):
        if rc == -1:
            raise ValueError("%s: unsupported degree" % name)
        if rc == -2:
            raise ValueError("regularizer %s cannot be used with solver %s" % (name, solver))
        if rc != 0:
            raise ValueError("regularizer init failed (%d)" % rc)

    def init_cache_pcd(self, degree, d, k):
        self._check(lib().spo_reg_init_cache_pcd(self._h, degree, d, k), self.name, "pcd")
        self.top_degree, self.d, self.k = degree, d, k

    def init_cache_pbcd(self, degree, d, k):
        self._check(lib().spo_reg_init_cache_pbcd(self._h, degree, d, k), self.name, "pbcd")
        self.top_degree, self.d, self.k = degree, d, k

    def compute_cache_pcd(self, P, degree, s):
        lib().spo_reg_compute_cache_pcd(self._h, _d(P), degree, s)

    def update_cache_pcd(self, P, degree, s, j):
        lib().spo_reg_update_cache_pcd(self._h, _d(P), degree, s, j)

    def prox_cd(self, p_sj, strength, degree, j):
        return lib().spo_reg_prox_cd(self._h, p_sj, strength, degree, j)

    def compute_cache_pbcd(self, P, degree):
        lib().spo_reg_compute_cache_pbcd(self._h, _d(P), degree)

    def update_cache_pbcd(self, P, degree, j):
        lib().spo_reg_update_cache_pbcd(self._h, _d(P), degree, j)

    def prox_bcd(self, p_j, strength, degree, j):
        lib().spo_reg_prox_bcd(self._h, _d(p_j), strength, degree, j)

    def state(self):
        nc = self.top_degree + 1
        cache, dcache = np.zeros(nc), np.zeros(nc)
        abs_p, norms = np.zeros(self.d), np.zeros(self.d)
        lib().spo_reg_get_state(self._h, _d(cache), _d(dcache), _d(abs_p), _d(norms))
        return {"cache": cache, "dcache": dcache, "abs_p": abs_p, "norms": norms}


def dloss(loss, p, y):
    p = np.ascontiguousarray(p, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    out = np.empty_like(p)
    lib().spo_dloss_vec(LOSSES[loss], p.size, _d(p), _d(y), _d(out))
    return out


def loss_sum(loss, p, y):
    p = np.ascontiguousarray(p, dtype=np.float64)
    y = np.ascontiguousarray(y, dtype=np.float64)
    return lib().spo_loss_sum(LOSSES[loss], p.size, _d(p), _d(y))


def cd_linear_epoch(w, X, y, y_pred, col_norm_sq, alpha, loss, indices_feature):
    """optimizer/cd_linear.py:8-33.  X is a CSC wrapper; w, y_pred updated in place."""
    idx = np.ascontiguousarray(indices_feature, dtype=np.int32)
    return lib().spo_cd_linear_epoch(
        _d(w), X.n, X.d, _l(X.indptr), _i(X.indices), _d(X.data), _d(y), _d(y_pred),
        _d(col_norm_sq), float(alpha), LOSSES[loss], _i(idx), idx.size,
    )


def pcd_epoch(P, X, y, y_pred, lams, degree, beta, gamma, eta, regularizer, loss, A,
              indices_component, indices_feature):
    """optimizer/pcd.py:71-137.  P (k, d) and y_pred, A updated in place."""
    ic = np.ascontiguousarray(indices_component, dtype=np.int32)
    jf = np.ascontiguousarray(indices_feature, dtype=np.int32)
    assert P.shape[1] == X.d and A.shape[0] == X.n
    return lib().spo_pcd_epoch(
        _d(P), P.shape[0], X.n, X.d, _l(X.indptr), _i(X.indices), _d(X.data), _d(y),
        _d(y_pred), _d(lams), degree, float(beta), float(gamma), float(eta), regularizer._h,
        LOSSES[loss], _d(A), A.shape[1], _i(ic), ic.size, _i(jf), jf.size,
    )


def pbcd_epoch(P, X, y, y_pred, lams, degree, beta, gamma, eta, regularizer, loss, A, dA,
               indices_feature):
    """optimizer/pbcd.py:82-148.  P is (d, k)."""
    jf = np.ascontiguousarray(indices_feature, dtype=np.int32)
    assert P.shape[0] == X.d and A.shape[0] == X.n and A.shape[2] == P.shape[1]
    return lib().spo_pbcd_epoch(
        _d(P), P.shape[1], X.n, X.d, _l(X.indptr), _i(X.indices), _d(X.data), _d(y),
        _d(y_pred), _d(lams), degree, float(beta), float(gamma), float(eta), regularizer._h,
        LOSSES[loss], _d(A), _d(dA), A.shape[1], _i(jf), jf.size,
    )


def pcd_all_epoch(P, X, y, y_pred, lams, beta, gamma, eta, regularizer, loss, A,
                  indices_component, indices_feature):
    """optimizer/pcd_all.py:44-102.  P (k, d); A (n)."""
    ic = np.ascontiguousarray(indices_component, dtype=np.int32)
    jf = np.ascontiguousarray(indices_feature, dtype=np.int32)
    return lib().spo_pcd_all_epoch(
        _d(P), P.shape[0], X.n, X.d, _l(X.indptr), _i(X.indices), _d(X.data), _d(y),
        _d(y_pred), _d(lams), float(beta), float(gamma), float(eta), regularizer._h,
        LOSSES[loss], _d(A), _i(ic), ic.size, _i(jf), jf.size,
    )


def pbcd_all_epoch(P, X, y, y_pred, lams, beta, gamma, eta, regularizer, loss, A,
                   indices_feature):
    """optimizer/pbcd_all.py:68-132.  P (d, k); A (n, k)."""
    jf = np.ascontiguousarray(indices_feature, dtype=np.int32)
    return lib().spo_pbcd_all_epoch(
        _d(P), P.shape[1], X.n, X.d, _l(X.indptr), _i(X.indices), _d(X.data), _d(y),
        _d(y_pred), _d(lams), float(beta), float(gamma), float(eta), regularizer._h,
        LOSSES[loss], _d(A), _i(jf), jf.size,
    )


LEARNING_RATE = {"constant": 0, "optimal": 1, "pegasos": 2, "invscaling": 3}


class CSR(object):
    """Row-major view used by psgd (reference: dataset.py get_dataset(X, "c"))."""

    def __init__(self, X):
        Xr = sp.csr_matrix(X, dtype=np.float64)
        Xr.sum_duplicates()
        Xr.sort_indices()
        self.n, self.d = Xr.shape
        self.indptr = np.ascontiguousarray(Xr.indptr, dtype=np.int64)
        self.indices = np.ascontiguousarray(Xr.indices, dtype=np.int32)
        self.data = np.ascontiguousarray(Xr.data, dtype=np.float64)
        self.nnz = int(self.indptr[-1])


def reg_prox(name, P, strength):
    """regularizer.prox(P, strength, degree) on P (d, k) in place (l1.py:50, l21.py:43,
    squaredl12.py:66, squaredl21.py:63)."""
    assert P.flags.c_contiguous and P.dtype == np.float64
    rc = lib().spo_reg_prox(REGULARIZERS[name], _d(P), P.shape[0], P.shape[1], float(strength))
    if rc:
        raise AttributeError("%s has no prox" % name)


def prox_squaredl12(p, strength):
    """regularizer/utils.py:27-70 on a contiguous vector, in place."""
    assert p.flags.c_contiguous and p.dtype == np.float64
    lib().spo_prox_squaredl12(_d(p), p.size, 1, float(strength))


def psgd_epoch(P, w, X, y, lams, degree, alpha, beta, gamma, regularizer, loss,
               indices_samples, fit_linear, eta0, learning_rate, power_t, batch_size, it):
    """optimizer/psgd.py:125-199.  P (n_orders, d, k); X a CSR view.  Returns (sum_loss, it)."""
    idx = np.ascontiguousarray(indices_samples, dtype=np.int32)
    itc = C.c_int64(int(it))
    sl = lib().spo_psgd_epoch(
        _d(P), _d(w), _d(lams), P.shape[0], P.shape[2], X.n, X.d, _l(X.indptr), _i(X.indices),
        _d(X.data), _d(y), LOSSES[loss], REGULARIZERS[regularizer], int(degree), float(alpha),
        float(beta), float(gamma), _i(idx), int(bool(fit_linear)), float(eta0),
        LEARNING_RATE[learning_rate] if isinstance(learning_rate, str) else int(learning_rate),
        float(power_t), int(batch_size), C.byref(itc))
    return sl, itc.value


def all_subsets_predict(X, P, lams):
    """kernels.py:117-137,140-153 (kernel='all-subsets'): sum_s lams[s] prod_j (1 + x_ij p_sj)."""
    Xr = sp.csr_matrix(X, dtype=np.float64)
    indptr = np.ascontiguousarray(Xr.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(Xr.indices, dtype=np.int32)
    data = np.ascontiguousarray(Xr.data, dtype=np.float64)
    P = np.ascontiguousarray(P, dtype=np.float64)
    lams = np.ascontiguousarray(lams, dtype=np.float64)
    out = np.zeros(Xr.shape[0])
    lib().spo_all_subsets_predict_csr(Xr.shape[0], _l(indptr), _i(indices), _d(data), _d(P),
                                      P.shape[0], P.shape[1], _d(lams), _d(out))
    return out


# --------------------------------------------------------------- kernels.py


def _safe_power(X, degree):
    """kernels.py:14-40"""
    if sp.issparse(X):
        return X.power(degree)
    return X ** degree


def _dot(X, B):
    """sklearn safe_sparse_dot(dense_output irrelevant for dense B)."""
    out = X @ B
    return np.asarray(out)


def _D(X, P, degree):
    """kernels.py:43-48"""
    return _dot(_safe_power(X, degree), P.T ** degree)


def homogeneous_kernel(X, P, degree):
    """kernels.py:51-68: sklearn polynomial_kernel(gamma=1, coef0=0) = (X P^T)^degree."""
    K = _dot(X, P.T)
    return K ** degree


def anova_kernel(X, P, degree=2):
    """kernels.py:71-115 -- same closed forms / recursion, same operation order."""
    if degree == 2:
        K = homogeneous_kernel(X, P, 2)
        K -= _D(X, P, 2)
        K /= 2
    elif degree == 3:
        K = homogeneous_kernel(X, P, 3)
        K -= 3 * _D(X, P, 2) * _D(X, P, 1)
        K += 2 * _D(X, P, 3)
        K /= 6
    else:
        n1, n2 = X.shape[0], P.shape[0]
        Ds = [_dot(X, P.T)]
        Ds += [_D(X, P, t) for t in range(2, degree + 1)]
        anovas = [1.0, Ds[0]]
        for m in range(2, degree + 1):
            anova = np.zeros((n1, n2))
            sign = 1.0
            for t in range(1, m + 1):
                anova += sign * anovas[m - t] * Ds[t - 1]
                sign *= -1.0
            anova /= 1.0 * m
            anovas.append(anova)
        K = anovas[-1]
    return K


def poly_predict(X, P, lams, degree=2):
    """kernels.py:140-153, kernel='anova' branch."""
    return np.dot(anova_kernel(X, P, degree), lams)


def anova_predict_dp(Xcsr, P, lams, degree, out):
    """Row DP evaluation (large sizes; cpu_baseline leg). Accumulates into out."""
    Xr = sp.csr_matrix(Xcsr, dtype=np.float64)
    indptr = np.ascontiguousarray(Xr.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(Xr.indices, dtype=np.int32)
    data = np.ascontiguousarray(Xr.data, dtype=np.float64)
    P = np.ascontiguousarray(P, dtype=np.float64)
    lams = np.ascontiguousarray(lams, dtype=np.float64)
    lib().spo_anova_predict_csr(
        Xr.shape[0], _l(indptr), _i(indices), _d(data), _d(P), P.shape[0], P.shape[1],
        _d(lams), degree, _d(out),
    )
    return out


# ------------------------------------------------- epoch drivers (L3 / L4)


class OracleFM(object):
    """Restatement of _BaseSparseFactorizationMachine.fit for solver pcd/pbcd.

    sparse_factorization_machines.py:355-435 (fit), :175-258 (_fit_pcd),
    :260-353 (_fit_pbcd), :437-451 (_get_output).  ``y`` must already be the
    regression target or the +-1 label vector (base.py:40-50,126-142).

    Extra, oracle-only knobs: ``feature_order`` / ``component_order`` replace the
    ``np.arange`` coordinate orders (the reference's epoch functions accept any
    order: pcd.py:86-87,97; pbcd.py:99,110); ``history`` records per-iteration
    (viol, sum loss) so trajectories can be compared.
    """

    def __init__(self, degree=2, loss="squared", n_components=2, solver="pcd",
                 regularizer="squaredl12", alpha=1, beta=1, gamma=1, mean=False, tol=1e-6,
                 fit_lower="explicit", fit_linear=True, init_lambdas="ones", max_iter=100,
                 shuffle=False, eta0=1.0, random_state=None, feature_order=None,
                 component_order=None, callback=None, n_calls=10, batch_size="auto",
                 learning_rate="optimal", power_t=1.0, n_iter_no_change=5):
        self.__dict__.update(locals())
        del self.__dict__["self"]

    def _get_output(self, X):
        """sparse_factorization_machines.py:437-451"""
        y_pred = poly_predict(X, self.P_[0], self.lams_, self.degree)
        if self.fit_linear:
            y_pred += np.asarray(X @ self.w_).ravel()
        if self.fit_lower == "explicit" and self.degree == 3:
            y_pred += poly_predict(X, self.P_[1], self.lams_, 2)
        return y_pred

    def predict(self, X):
        return self._get_output(X)

    def fit(self, X, y, P_init=None, w_init=None, lams_init=None):
        from sklearn.utils import check_random_state

        y = np.ascontiguousarray(y, dtype=np.float64)
        n, d = X.shape
        rng = check_random_state(self.random_state)
        reg = Regularizer(self.regularizer)
        self.w_ = np.zeros(d) if w_init is None else np.array(w_init, dtype=np.float64)
        n_orders = self.degree - 1 if self.fit_lower == "explicit" else 1
        if P_init is None:
            self.P_ = 0.01 * rng.randn(n_orders, self.n_components, d)
        else:
            self.P_ = np.array(P_init, dtype=np.float64)
        if lams_init is not None:
            self.lams_ = np.array(lams_init, dtype=np.float64)
        elif self.init_lambdas == "ones":
            self.lams_ = np.ones(self.n_components)
        elif self.init_lambdas == "random_signs":
            self.lams_ = np.sign(rng.randn(self.n_components))
        else:
            raise ValueError("bad init_lambdas")
        ds = CSC(X)
        y_pred = np.ascontiguousarray(self._get_output(X), dtype=np.float64)
        if sp.issparse(X):
            col_norm_sq = np.asarray(X.multiply(X).sum(axis=0)).ravel().astype(np.float64)
        else:
            col_norm_sq = np.einsum("ij,ij->j", np.asarray(X, float), np.asarray(X, float))
        self.history = []
        self.y_pred_ = y_pred
        if self.solver == "psgd":
            self.it_ = 1
            conv, self.n_iter_ = self._fit_psgd(CSR(X), y, rng)
        elif self.solver == "pcd":
            conv, self.n_iter_ = self._fit_pcd(ds, y, y_pred, col_norm_sq, reg, rng)
        elif self.solver == "pbcd":
            conv, self.n_iter_ = self._fit_pbcd(ds, y, y_pred, col_norm_sq, reg, rng)
        else:
            raise ValueError("Solver %s is not supported." % self.solver)
        self.converged_ = conv
        return self

    def _scaled(self, n):
        if self.mean:
            return self.alpha * n, self.beta * n, self.gamma * n
        return self.alpha, self.beta, self.gamma

    def _orders(self, d):
        jf = (np.arange(d, dtype=np.int32) if self.feature_order is None
              else np.array(self.feature_order, dtype=np.int32))
        ic = (np.arange(self.n_components, dtype=np.int32) if self.component_order is None
              else np.array(self.component_order, dtype=np.int32))
        return ic, jf

    def _fit_pcd(self, X, y, y_pred, col_norm_sq, reg, rng):
        n, d = X.n, X.d
        ic, jf = self._orders(d)
        alpha, beta, gamma = self._scaled(n)
        A = np.zeros((n, self.degree + 1))
        A[:, 0] = 1.0
        reg.init_cache_pcd(self.degree, d, self.n_components)
        converged = False
        it = 0
        for it in range(self.max_iter):
            viol = 0
            if self.shuffle:
                rng.shuffle(ic)
                rng.shuffle(jf)
            if self.fit_linear:
                viol += cd_linear_epoch(self.w_, X, y, y_pred, col_norm_sq, alpha, self.loss, jf)
            if self.fit_lower == "explicit":
                for deg in range(2, self.degree):
                    viol += pcd_epoch(self.P_[self.degree - deg], X, y, y_pred, self.lams_, deg,
                                      beta, gamma, self.eta0, reg, self.loss, A, ic, jf)
            viol += pcd_epoch(self.P_[0], X, y, y_pred, self.lams_, self.degree, beta, gamma,
                              self.eta0, reg, self.loss, A, ic, jf)
            self.history.append((viol, loss_sum(self.loss, y_pred, y)))
            if (self.callback is not None) and it % self.n_calls == 0:
                if self.callback(self) is not None:
                    break
            if viol < self.tol:
                converged = True
                break
        return converged, it

    def _fit_psgd(self, X, y, rng):
        """sparse_factorization_machines.py:94-173"""
        n, d = X.n, X.d
        idx = np.arange(n, dtype=np.int32)
        bs = int(n * d / X.nnz) if self.batch_size == "auto" else int(self.batch_size)
        if self.learning_rate not in LEARNING_RATE:
            raise ValueError("learning_rate %s is not supported." % self.learning_rate)
        P = np.ascontiguousarray(self.P_.swapaxes(1, 2))
        best, noimp, converged, epoch = np.inf, 0, False, 0
        for epoch in range(self.max_iter):
            if self.shuffle:
                rng.shuffle(idx)
            sl, self.it_ = psgd_epoch(P, self.w_, X, y, self.lams_, self.degree, self.alpha,
                                      self.beta, self.gamma, self.regularizer, self.loss, idx,
                                      self.fit_linear, self.eta0, self.learning_rate,
                                      self.power_t, bs, self.it_)
            if (self.callback is not None) and epoch % self.n_calls == 0:
                if self.callback(self) is not None:
                    break
            sl /= n
            self.history.append((sl, self.it_))
            if sl > (best - self.tol):
                noimp += 1
            else:
                noimp = 0
            if sl < best:
                best = sl
            if noimp >= self.n_iter_no_change:
                converged = True
                break
        self.P_[:, :, :] = np.array(P.swapaxes(1, 2))
        return converged, epoch

    def _fit_pbcd(self, X, y, y_pred, col_norm_sq, reg, rng):
        n, d = X.n, X.d
        _, jf = self._orders(d)
        alpha, beta, gamma = self._scaled(n)
        k = self.n_components
        A = np.zeros((n, self.degree + 1, k))
        dA = np.zeros((n, self.degree, k))
        A[:, 0] = 1.0
        reg.init_cache_pbcd(self.degree, d, k)
        P = np.ascontiguousarray(self.P_.swapaxes(1, 2))  # (n_orders, d, k) copy
        converged = False
        it = 0
        for it in range(self.max_iter):
            viol = 0
            if self.shuffle:
                rng.shuffle(jf)
            if self.fit_linear:
                viol += cd_linear_epoch(self.w_, X, y, y_pred, col_norm_sq, alpha, self.loss, jf)
            if self.fit_lower == "explicit":
                for deg in range(2, self.degree):
                    viol += pbcd_epoch(P[self.degree - deg], X, y, y_pred, self.lams_, deg, beta,
                                       gamma, self.eta0, reg, self.loss, A, dA, jf)
            viol += pbcd_epoch(P[0], X, y, y_pred, self.lams_, self.degree, beta, gamma,
                               self.eta0, reg, self.loss, A, dA, jf)
            self.history.append((viol, loss_sum(self.loss, y_pred, y)))
            if (self.callback is not None) and it % self.n_calls == 0:
                if self.callback(self) is not None:
                    break
            if viol < self.tol:
                converged = True
                break
        self.P_[:, :, :] = np.array(P.swapaxes(1, 2))
        return converged, it


class OracleAllSubsets(object):
    """Restatement of _BaseSparseAllSubsets.fit for solver pcd/pbcd
    (sparse_all_subsets.py:80-272): P_ (k, d), no linear term, eta0 default 0.1."""

    def __init__(self, loss="squared", n_components=2, solver="pcd", beta=1, gamma=1, eta0=0.1,
                 mean=False, tol=1e-6, regularizer="omegati", init_lambdas="ones", max_iter=100,
                 shuffle=False, random_state=None, feature_order=None):
        self.__dict__.update(locals())
        del self.__dict__["self"]

    def predict(self, X):
        return all_subsets_predict(X, self.P_, self.lams_)

    def fit(self, X, y, P_init=None, lams_init=None):
        from sklearn.utils import check_random_state

        y = np.ascontiguousarray(y, dtype=np.float64)
        n, d = X.shape
        k = self.n_components
        rng = check_random_state(self.random_state)
        reg = Regularizer(self.regularizer)
        self.P_ = (0.01 * rng.randn(k, d) if P_init is None
                   else np.array(P_init, dtype=np.float64))
        if lams_init is not None:
            self.lams_ = np.array(lams_init, dtype=np.float64)
        elif self.init_lambdas == "ones":
            self.lams_ = np.ones(k)
        else:
            self.lams_ = np.sign(rng.randn(k))
        ds = CSC(X)
        y_pred = np.ascontiguousarray(self.predict(X))
        beta = self.beta * n if self.mean else self.beta
        gamma = self.gamma * n if self.mean else self.gamma
        jf = (np.arange(d, dtype=np.int32) if self.feature_order is None
              else np.array(self.feature_order, dtype=np.int32))
        ic = np.arange(k, dtype=np.int32)
        self.history = []
        converged = False
        it = 0
        if self.solver == "pcd":
            A = np.ones(n)
            reg.init_cache_pcd(-1, d, k)
            P = self.P_
        else:
            A = np.ones((n, k))
            reg.init_cache_pbcd(-1, d, k)
            P = np.ascontiguousarray(self.P_.T)
        for it in range(self.max_iter):
            if self.shuffle:
                if self.solver == "pcd":
                    rng.shuffle(ic)
                rng.shuffle(jf)
            if self.solver == "pcd":
                viol = pcd_all_epoch(P, ds, y, y_pred, self.lams_, beta, gamma, self.eta0, reg,
                                     self.loss, A, ic, jf)
            else:
                viol = pbcd_all_epoch(P, ds, y, y_pred, self.lams_, beta, gamma, self.eta0, reg,
                                      self.loss, A, jf)
            self.history.append((viol, loss_sum(self.loss, y_pred, y)))
            if viol < self.tol:
                converged = True
                break
        if self.solver == "pbcd":
            self.P_[:, :] = P.T
        self.y_pred_ = y_pred
        self.n_iter_ = it
        self.converged_ = converged
        return self
