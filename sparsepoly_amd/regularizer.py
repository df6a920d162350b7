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
traces recorded from the reference (``tests/golden/g5_reg_traces.npz``).  A user-defined
regularizer -- any object with this protocol registered under a name in an estimator's
``_REGULARIZERS``, as in the reference (``base.py:27-34``) -- cannot run on the device; the
estimators honour it through host-stepped epochs (``HipEngine.pcd_epoch_host`` /
``pbcd_epoch_host``: the device forms every step's column sums and scatter-updates, the object's
prox and cache hooks run on the host; ``include/spfm.h``).

Conventions as in the reference: pcd works on ``P`` of shape (n_components, n_features), pbcd and
psgd on (n_features, n_components); ``eval`` takes (..., n_features, n_components) (stacks allowed)
and the ``transpose`` constructor argument of L21 / SquaredL12 / SquaredL21 swaps the roles of the
two axes exactly as in the reference; ``degree = -1`` selects the all-subsets form of OmegaTI /
OmegaCS, whose running product lives in ``_cache_all_subsets``.  ``eval`` is pinned by values
recorded from the reference (``tests/golden/g10_reg_eval.npz``).
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


def _unbox(v):
    """0-d results as Python floats, stacks as arrays (what the reference returns)"""
    v = np.asarray(v)
    return float(v) if v.ndim == 0 else v


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

    def init_cache_psgd(self, degree, n_features, n_components):
        self._require("psgd")

    def _require(self, solver):
        if solver not in self.solvers:
            raise ValueError("%s cannot be used with solver='%s'" % (type(self).__name__, solver))


class L1(_Regularizer):
    """regularizer/l1.py:12-51 -- pcd, pbcd, psgd; stateless"""
    name = "l1"
    solvers = ("pcd", "pbcd", "psgd")

    def eval(self, P, degree=None):  # l1.py:17-18
        return _unbox(np.abs(np.asarray(P)).sum(axis=(-2, -1)))

    def prox_cd(self, p_sj, strength, degree, j):
        return float(_soft(p_sj, strength))

    def prox_bcd(self, p_j, strength, degree, j):
        p_j[:] = _soft(p_j, strength)

    def prox(self, P, strength, degree):
        P[...] = _soft(P, strength)


class L21(_Regularizer):
    """regularizer/l21.py:14-48 -- pbcd, psgd; rows of P (n_features, n_components) are groups
    (``transpose=True``: columns; psgd only)"""
    name = "l21"
    solvers = ("pbcd", "psgd")

    def __init__(self, transpose=False):
        self.transpose = bool(transpose)

    def eval(self, P, degree=None):  # l21.py:19-21
        axis = -2 if self.transpose else -1
        return _unbox(np.linalg.norm(np.asarray(P), axis=axis).sum(axis=-1))

    def init_cache_pbcd(self, degree, n_features, n_components):
        if self.transpose:
            raise ValueError("self.transpose is True.")

    def prox_bcd(self, p_j, strength, degree, j):
        nrm = float(np.sqrt(np.dot(p_j, p_j)))
        if nrm > strength:
            p_j *= 1.0 - strength / nrm
        else:
            p_j[:] = 0.0

    def prox(self, P, strength, degree):
        # reference quirk (l21.py:46-48): groups with norm <= strength are left UNCHANGED
        axis = 0 if self.transpose else 1
        nrm = np.linalg.norm(P, axis=axis)
        nrm[nrm <= strength] = np.inf
        P *= 1.0 - strength / np.expand_dims(nrm, axis=axis)


class SquaredL12(_Regularizer):
    """regularizer/squaredl12.py:15-78 -- pcd (degree 2 only), psgd.  ``transpose=True`` (the
    default, what the estimators use): Omega = sum_s ||P[s,:]||_1^2, one running sum per component
    pass; ``transpose=False``: one sum per feature over the components"""
    name = "squaredl12"
    solvers = ("pcd", "psgd")

    def __init__(self, transpose=True):
        self.transpose = bool(transpose)

    def eval(self, P, degree=None):  # squaredl12.py:20-22
        axis = -2 if self.transpose else -1
        return _unbox((np.abs(np.asarray(P)).sum(axis=axis) ** 2).sum(axis=-1))

    def init_cache_pcd(self, degree, n_features, n_components):
        if degree > 2:
            raise ValueError("SquaredL12 supports only degree=2.")
        self._abs_p = np.zeros(n_features)
        self._cache = np.zeros(1 if self.transpose else n_features)

    def compute_cache_pcd_all(self, P, degree):  # squaredl12.py:33-40 (indexing as written there)
        if not self.transpose:
            n_components, n_features = P.shape[0], P.shape[1]
            for j in range(n_features):
                self._cache[j] = 0.0
                for s in range(n_components):
                    self._cache[j] += abs(P[j, s])

    def compute_cache_pcd(self, P, degree, s):
        self._abs_p[:] = np.abs(P[s])
        if self.transpose:
            self._cache[0] = 0.0
            for v in self._abs_p:       # sequential sum: the reference's order
                self._cache[0] += v

    def prox_cd(self, p_sj, strength, degree, j):
        i = 0 if self.transpose else j
        others = self._cache[i] - self._abs_p[j]
        p = p_sj / (1.0 + 2.0 * strength)
        return _sign_pm(p) * max(abs(p) - 2.0 * strength * others / (1.0 + 2.0 * strength), 0.0)

    def update_cache_pcd(self, P, degree, s, j):
        # _abs_p[j] keeps the snapshot of compute_cache_pcd (each j is visited once per pass);
        # two statements, as in the reference: (c - a) + b
        i = 0 if self.transpose else j
        self._cache[i] -= self._abs_p[j]
        self._cache[i] += abs(P[s, j])

    def init_cache_psgd(self, degree, n_features, n_components):
        if self.transpose:
            self._cache = np.zeros(n_features)
            self._candidates = np.arange(n_features, dtype=np.int32)
        else:
            self._candidates = np.arange(n_components, dtype=np.int32)

    def prox(self, P, strength, degree):
        # psgd: P is (n_features, n_components); transpose: every component (column) separately,
        # else every feature (row)
        if self.transpose:
            for s in range(P.shape[1]):
                P[:, s] = _prox_sq_l1(P[:, s].copy(), strength)
        else:
            for j in range(P.shape[0]):
                P[j] = _prox_sq_l1(P[j].copy(), strength)


class SquaredL21(_Regularizer):
    """regularizer/squaredl21.py:18-74 -- pbcd (degree 2 only), psgd;
    Omega = (sum_j ||P[j,:]||_2)^2 (``transpose=True``: groups are the columns; psgd only)"""
    name = "squaredl21"
    solvers = ("pbcd", "psgd")

    def __init__(self, transpose=False):
        self.transpose = bool(transpose)

    def eval(self, P, degree=None):  # squaredl21.py:23-25
        axis = -2 if self.transpose else -1
        return _unbox(np.linalg.norm(np.asarray(P), axis=axis).sum(axis=-1) ** 2)

    def init_cache_pbcd(self, degree, n_features, n_components):
        if degree != 2:
            raise ValueError("SquaredL21 supports only degree=2.")
        if self.transpose:
            raise ValueError("transpose != False.")
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

    def init_cache_psgd(self, degree, n_features, n_components):
        n = n_components if self.transpose else n_features
        self._candidates = np.arange(n, dtype=np.int32)

    def prox(self, P, strength, degree):
        axis = 0 if self.transpose else 1
        nrm = np.linalg.norm(P, axis=axis)
        nz = nrm > 0
        new = _prox_sq_l1(nrm, strength)
        if self.transpose:
            P[:, nz] /= nrm[nz][None, :]
        else:
            P[nz] /= nrm[nz][:, None]
        P *= np.expand_dims(new, axis=axis)


class OmegaTI(_Regularizer):
    """regularizer/omegati.py:14-104 -- pcd, any degree (degree = -1: all-subsets);
    Omega = sum_s e_m(|P[s,:]|) (elementary symmetric polynomial)"""
    name = "omegati"
    solvers = ("pcd",)

    def eval(self, P, degree):  # omegati.py:19-47: P is (..., n_features, n_components)
        P = np.asarray(P)
        Ps = np.abs(P.reshape(-1, P.shape[-2], P.shape[-1]))
        if degree == -1:
            return float(np.prod(Ps + 1.0, axis=1).sum())
        if degree <= 0:
            raise ValueError("degree must be a positive int or -1 (all).")
        res = np.empty(len(Ps))
        for q, M in enumerate(Ps):
            cache = np.zeros((degree + 1, M.shape[1]))
            cache[0] = 1.0
            for j in range(M.shape[0]):
                for t in range(degree, 0, -1):
                    cache[t] += cache[t - 1] * M[j]
            res[q] = cache[degree].sum()
        return float(res[0]) if P.ndim == 2 else res.reshape(P.shape[:-2])

    def init_cache_pcd(self, degree, n_features, n_components):
        self._abs_p = np.zeros(n_features)
        if degree > 0:
            self._cache = np.zeros(degree + 1)
            self._dcache = np.zeros(degree + 1)
        elif degree == -1:
            self._cache_all_subsets = 1.0
        else:
            raise ValueError("degree must be a positive int or -1 (all).")

    def compute_cache_pcd(self, P, degree, s):
        self._abs_p[:] = np.abs(P[s])
        if degree == -1:
            self._cache_all_subsets = 1.0
            for v in self._abs_p:
                self._cache_all_subsets *= 1.0 + v
            return
        # the caches are sized for the top degree; a lower-order epoch (fit_lower='explicit')
        # fills entries 0..degree and leaves the rest at 0 (omegati.py:62-74)
        self._cache[:] = 0.0
        self._cache[:degree + 1] = _esp_table(self._abs_p, degree)
        self._dcache[:] = 0.0
        self._dcache[1] = 1.0

    def prox_cd(self, p_sj, strength, degree, j):
        sgn = _sign_pm(p_sj)
        if degree == -1:
            self._cache_all_subsets /= 1.0 + self._abs_p[j]
            return sgn * max(abs(p_sj) - strength * self._cache_all_subsets, 0.0)
        for t in range(2, degree + 1):   # e_{t-1} of the other coordinates, clipped at 0 (:97-98)
            self._dcache[t] = max(self._cache[t - 1] - self._dcache[t - 1] * self._abs_p[j], 0.0)
        return sgn * max(abs(p_sj) - strength * self._dcache[degree], 0.0)

    def update_cache_pcd(self, P, degree, s, j):
        new = abs(P[s, j])
        if degree == -1:
            self._cache_all_subsets *= 1.0 + new
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
        # omegacs.py:22-39 as written: the stack is RESHAPED to (-1, shape[-1], shape[-2]) (not
        # transposed), the 2-norms run over the new last axis and the polynomial over the middle
        # one.  There is no degree = -1 form.
        P = np.asarray(P)
        if degree <= 0:
            raise ValueError("degree must be a positive int.")
        Ps = P.reshape(-1, P.shape[-1], P.shape[-2])
        nrm = np.linalg.norm(Ps, axis=-1)
        res = np.array([_esp_table(row, degree)[degree] for row in nrm])
        return float(res[0]) if P.ndim == 2 else res.reshape(P.shape[:-2])

    def init_cache_pbcd(self, degree, n_features, n_components):
        self._norms = np.zeros(n_features)
        if degree > 0:
            self._cache = np.zeros(degree + 1)
            self._dcache = np.zeros(degree + 1)
            self._dcache[1] = 1.0
        elif degree == -1:
            self._cache_all_subsets = 1.0
        else:
            raise ValueError("degree must be a positive int or -1.")

    def _recompute(self, degree):
        if degree == -1:
            self._cache_all_subsets = 1.0
            for v in self._norms:
                self._cache_all_subsets *= 1.0 + v
            return
        self._cache[:] = 0.0
        self._cache[:degree + 1] = _esp_table(self._norms, degree)

    def compute_cache_pbcd(self, P, degree):
        self._norms[:] = np.sqrt((P * P).sum(axis=1))
        self._recompute(degree)

    def prox_bcd(self, p_j, strength, degree, j):
        if degree == -1:
            self._cache_all_subsets /= 1.0 + self._norms[j]
            lam = strength * self._cache_all_subsets
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
            self._cache_all_subsets *= 1.0 + new
            if self._cache_all_subsets < 0:   # omegacs.py:80-81
                self._norms[j] = new
                self._recompute(degree)
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
