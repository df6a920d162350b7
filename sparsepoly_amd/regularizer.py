"""Regularizer plug-in objects (reference: ``sparsepoly/regularizer/*.py``, registry
``regularizer/__init__.py:8-15``).

In the reference these are Numba jitclass objects whose ``prox_cd`` / ``prox_bcd`` and cache
hooks are called from inside the epoch kernels.  In this library the six built-ins run as
enum-dispatched device functions inside the HIP chain kernels (``csrc/spfm_pcd.hip.h``,
``spfm_pbcd.hip.h``, ``spfm_psgd.hip.h``); ``fit`` never calls the Python objects below.  They
are the HOST face of the same protocol -- what ``from sparsepoly import L1; L1().prox_cd(...)``
gives a user of the reference: ``eval``, the pcd protocol (``init_cache_pcd``,
``compute_cache_pcd``, ``prox_cd``, ``update_cache_pcd``), the pbcd protocol
(``init_cache_pbcd``, ``compute_cache_pbcd``, ``prox_bcd``, ``update_cache_pbcd``) and the
full-matrix ``prox`` of the psgd solver, with the reference's state attributes (``_cache``,
``_dcache``, ``_abs_p``, ``_norms``) readable and writable.  They are checked call by call against
traces recorded from the reference (``tests/golden/g5_reg_traces.npz``).  A user-defined Python
regularizer cannot run on the device; the estimators reject it with ``ValueError``.

Conventions as in the reference: pcd works on ``P`` of shape (n_components, n_features), pbcd on
(n_features, n_components); ``degree = -1`` selects the all-subsets form of OmegaTI / OmegaCS.
"""
import numpy as np


def _soft(x, t):
    """sign(x) max(|x| - t, 0) with numpy's sign convention (sign(0) = 0)"""
    return np.sign(x) * np.maximum(np.abs(x) - t, 0.0)


def _sign_pm(x):
    """the reference's ``1 if x > 0 else -1`` (squaredl12.py:56, omegati.py:92)"""
    return 1.0 if x > 0 else -1.0


def _prox_sq_l1(v, strength):
    """argmin_u 0.5 ||u - v||^2 + strength ||u||_1^2, exact: with the magnitudes sorted
    descending the support is the longest prefix whose last element exceeds the threshold
    tau = 2 c S / (1 + 2 c m) of that prefix (S = prefix sum, m = its length).  The reference
    finds the same support by randomised pivoting (regularizer/utils.py:27-70)."""
    a = np.abs(v)
    if a.size == 0:
        return v
    srt = np.sort(a)[::-1]
    csum = np.cumsum(srt)
    m = np.arange(1, a.size + 1)
    tau_all = 2.0 * strength * csum / (1.0 + 2.0 * strength * m)
    ok = np.nonzero(srt > tau_all)[0]
    if ok.size == 0:
        return np.zeros_like(v)
    tau = tau_all[ok[-1]]
    return np.sign(v) * np.maximum(a - tau, 0.0)


def _esp_table(values, degree):
    """e_0..e_degree of `values` by the in-place DP the reference uses (descending t)"""
    c = np.zeros(degree + 1)
    c[0] = 1.0
    for v in values:
        for t in range(degree, 0, -1):
            c[t] += c[t - 1] * v
    return c


class _Regularizer(object):
    name = None
    solvers = ()

    def __repr__(self):
        return "%s()" % type(self).__name__

    # stateless defaults (regularizer/l1.py:20-30,35-42)
    def init_cache_pcd(self, degree, n_features, n_components):
        self._require("pcd")

    def compute_cache_pcd_all(self, P, degree):
        pass

    def compute_cache_pcd(self, P, degree, s):
        pass

    def update_cache_pcd(self, P, degree, s, j):
        pass

    def init_cache_pbcd(self, degree, n_features, n_components):
        self._require("pbcd")

    def compute_cache_pbcd(self, P, degree):
        pass

    def update_cache_pbcd(self, P, degree, j):
        pass

    def _require(self, solver):
        if solver not in self.solvers:
            raise ValueError("%s cannot be used with solver='%s'" % (type(self).__name__, solver))


class L1(_Regularizer):
    """regularizer/l1.py:12-51 -- pcd, pbcd, psgd; stateless"""
    name = "l1"
    solvers = ("pcd", "pbcd", "psgd")

    def eval(self, P, degree=None):
        return float(np.abs(P).sum())

    def prox_cd(self, p_sj, strength, degree, j):
        return float(_soft(p_sj, strength))

    def prox_bcd(self, p_j, strength, degree, j):
        p_j[:] = _soft(p_j, strength)

    def prox(self, P, strength, degree):
        P[...] = _soft(P, strength)


class L21(_Regularizer):
    """regularizer/l21.py:14-48 -- pbcd, psgd; rows of P (n_features, n_components) are groups"""
    name = "l21"
    solvers = ("pbcd", "psgd")

    def eval(self, P, degree=None):
        return float(np.linalg.norm(P, axis=-1).sum())

    def prox_bcd(self, p_j, strength, degree, j):
        nrm = float(np.sqrt(np.dot(p_j, p_j)))
        if nrm > strength:
            p_j *= 1.0 - strength / nrm
        else:
            p_j[:] = 0.0

    def prox(self, P, strength, degree):
        # reference quirk (l21.py:46-48): rows with norm <= strength are left UNCHANGED
        nrm = np.linalg.norm(P, axis=1)
        nrm[nrm <= strength] = np.inf
        P *= (1.0 - strength / nrm)[:, None]


class SquaredL12(_Regularizer):
    """regularizer/squaredl12.py:15-78 -- pcd (degree 2 only), psgd;  Omega = sum_s ||P[s,:]||_1^2"""
    name = "squaredl12"
    solvers = ("pcd", "psgd")

    def eval(self, P, degree=None):
        return float((np.abs(P).sum(axis=-1) ** 2).sum())

    def init_cache_pcd(self, degree, n_features, n_components):
        if degree > 2:
            raise ValueError("SquaredL12 supports only degree=2.")
        self._abs_p = np.zeros(n_features)
        self._cache = np.zeros(1)

    def compute_cache_pcd(self, P, degree, s):
        self._abs_p[:] = np.abs(P[s])
        self._cache[0] = 0.0
        for v in self._abs_p:           # sequential sum: the reference's order
            self._cache[0] += v

    def prox_cd(self, p_sj, strength, degree, j):
        others = self._cache[0] - self._abs_p[j]
        p = p_sj / (1.0 + 2.0 * strength)
        return _sign_pm(p) * max(abs(p) - 2.0 * strength * others / (1.0 + 2.0 * strength), 0.0)

    def update_cache_pcd(self, P, degree, s, j):
        # _abs_p[j] keeps the snapshot of compute_cache_pcd (each j is visited once per pass)
        self._cache[0] += abs(P[s, j]) - self._abs_p[j]

    def prox(self, P, strength, degree):
        # psgd: P is (n_features, n_components); every component (column) separately
        for s in range(P.shape[1]):
            P[:, s] = _prox_sq_l1(P[:, s].copy(), strength)


class SquaredL21(_Regularizer):
    """regularizer/squaredl21.py:18-74 -- pbcd (degree 2 only), psgd;
    Omega = (sum_j ||P[j,:]||_2)^2"""
    name = "squaredl21"
    solvers = ("pbcd", "psgd")

    def eval(self, P, degree=None):
        return float(np.linalg.norm(P, axis=-1).sum() ** 2)

    def init_cache_pbcd(self, degree, n_features, n_components):
        if degree != 2:
            raise ValueError("SquaredL21 supports only degree=2.")
        self._norms = np.zeros(n_features)
        self._cache = 0.0

    def compute_cache_pbcd(self, P, degree):
        self._norms[:] = np.sqrt((P * P).sum(axis=1))
        self._cache = float(self._norms.sum())

    def prox_bcd(self, p_j, strength, degree, j):
        p_j /= 1.0 + 2.0 * strength
        if self._cache < self._norms[j]:          # "numerical error" branch (:48-49)
            self._cache = float(self._norms.sum())
        others = self._cache - self._norms[j]
        lam = 2.0 * strength * others / (1.0 + 2.0 * strength)
        nrm = float(np.sqrt(np.dot(p_j, p_j)))
        if nrm > lam:
            p_j *= 1.0 - lam / nrm
        else:
            p_j[:] = 0.0

    def update_cache_pbcd(self, P, degree, j):
        new = float(np.sqrt(np.dot(P[j], P[j])))
        self._cache -= self._norms[j]
        self._cache += new
        self._norms[j] = new

    def prox(self, P, strength, degree):
        nrm = np.linalg.norm(P, axis=1)
        nz = nrm > 0
        P[nz] /= nrm[nz][:, None]
        P *= _prox_sq_l1(nrm, strength)[:, None]


class OmegaTI(_Regularizer):
    """regularizer/omegati.py:14-104 -- pcd, any degree (degree = -1: all-subsets);
    Omega = sum_s e_m(|P[s,:]|) (elementary symmetric polynomial)"""
    name = "omegati"
    solvers = ("pcd",)

    def eval(self, P, degree):
        P2 = np.atleast_2d(P)
        if degree == -1:
            return float(np.prod(np.abs(P2) + 1.0, axis=-1).sum())
        if degree <= 0:
            raise ValueError("degree must be a positive int or -1 (all).")
        return float(sum(_esp_table(np.abs(row), degree)[degree] for row in P2))

    def init_cache_pcd(self, degree, n_features, n_components):
        if degree == -1:
            self._cache = np.ones(1)
            self._dcache = np.ones(1)
        elif degree > 0:
            self._cache = np.zeros(degree + 1)
            self._dcache = np.zeros(degree + 1)
        else:
            raise ValueError("degree must be a positive int or -1 (all).")
        self._abs_p = np.zeros(n_features)

    def compute_cache_pcd(self, P, degree, s):
        self._abs_p[:] = np.abs(P[s])
        if degree == -1:
            self._cache[0] = np.prod(1.0 + self._abs_p)
            return
        self._cache[:] = _esp_table(self._abs_p, degree)
        self._dcache[:] = 0.0
        self._dcache[1] = 1.0

    def prox_cd(self, p_sj, strength, degree, j):
        sgn = _sign_pm(p_sj)
        if degree == -1:
            self._cache[0] /= 1.0 + self._abs_p[j]
            return sgn * max(abs(p_sj) - strength * self._cache[0], 0.0)
        for t in range(2, degree + 1):   # e_{t-1} of the other coordinates, clipped at 0 (:97-98)
            self._dcache[t] = max(self._cache[t - 1] - self._dcache[t - 1] * self._abs_p[j], 0.0)
        return sgn * max(abs(p_sj) - strength * self._dcache[degree], 0.0)

    def update_cache_pcd(self, P, degree, s, j):
        new = abs(P[s, j])
        if degree == -1:
            self._cache[0] *= 1.0 + new
        else:
            for t in range(1, degree):
                self._cache[t] = self._dcache[t + 1] + self._dcache[t] * new
        self._abs_p[j] = new


class OmegaCS(_Regularizer):
    """regularizer/omegacs.py:17-106 -- pbcd, any degree (degree = -1: all-subsets): OmegaTI
    on the block norms ||P[j,:]||_2"""
    name = "omegacs"
    solvers = ("pbcd",)

    def eval(self, P, degree):
        nrm = np.linalg.norm(P, axis=-1)
        if degree == -1:
            return float(np.prod(1.0 + nrm))
        if degree <= 0:
            raise ValueError("degree must be a positive int or -1.")
        return float(_esp_table(nrm, degree)[degree])

    def init_cache_pbcd(self, degree, n_features, n_components):
        if degree == -1:
            self._cache = np.ones(1)
            self._dcache = np.ones(1)
        elif degree > 0:
            self._cache = np.zeros(degree + 1)
            self._dcache = np.zeros(degree + 1)
            self._dcache[1] = 1.0
        else:
            raise ValueError("degree must be a positive int or -1.")
        self._norms = np.zeros(n_features)

    def _recompute(self, degree):
        if degree == -1:
            self._cache[0] = np.prod(1.0 + self._norms)
            return
        self._cache[:] = 0.0
        self._cache[:degree + 1] = _esp_table(self._norms, degree)

    def compute_cache_pbcd(self, P, degree):
        self._norms[:] = np.sqrt((P * P).sum(axis=1))
        self._recompute(degree)

    def prox_bcd(self, p_j, strength, degree, j):
        if degree == -1:
            self._cache[0] /= 1.0 + self._norms[j]
            lam = strength * self._cache[0]
        else:
            for t in range(2, degree + 1):
                self._dcache[t] = self._cache[t - 1] - self._dcache[t - 1] * self._norms[j]
            if np.min(self._dcache) < 0:      # "numerical error" branch (:90-96)
                self._norms[j] = 0.0
                self._recompute(degree - 1)
                for t in range(2, degree + 1):
                    self._dcache[t] = self._cache[degree - 1]
            lam = strength * self._dcache[degree]
        nrm = float(np.sqrt(np.dot(p_j, p_j)))
        if nrm > lam:
            p_j *= 1.0 - lam / nrm
        else:
            p_j[:] = 0.0

    def update_cache_pbcd(self, P, degree, j):
        new = float(np.sqrt(np.dot(P[j], P[j])))
        if degree == -1:
            self._cache[0] *= 1.0 + new
        else:
            for t in range(1, degree + 1):
                self._cache[t] += self._dcache[t] * new
                self._cache[t] -= self._dcache[t] * self._norms[j]
            if np.min(self._cache) < 0:       # "numerical error" branch (:75-76)
                self._norms[j] = new
                self._recompute(degree)
        self._norms[j] = new


# same key order as the reference registry (it shows in the error message)
REGULARIZATION = {
    "squaredl12": SquaredL12,
    "squaredl21": SquaredL21,
    "l1": L1,
    "l21": L21,
    "omegati": OmegaTI,
    "omegacs": OmegaCS,
}
