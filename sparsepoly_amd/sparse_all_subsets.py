"""Sparse all-subsets models on MI355X -- sklearn-style estimators.

Drop-in for ``sparsepoly.SparseAllSubsets{Regressor,Classifier}`` (reference
``sparsepoly/sparse_all_subsets.py``): model ``y(x) = sum_s lam_s prod_j (1 + p_sj x_j)``
trained by pcd (``optimizer/pcd_all.py:44-102``) or pbcd (``optimizer/pbcd_all.py:68-132``)
with the ``l1`` / ``omegati`` (pcd) and ``l1`` / ``l21`` / ``omegacs`` (pbcd) regularizers
called with ``degree = -1``.  Same constructor keywords, order and defaults
(``sparse_all_subsets.py:42-78``), same fitted attributes (``P_ (k, d)``, ``lams_``,
``n_iter_``), same warnings; extra trailing keywords ``schedule, precision, device`` as for
the factorization-machine estimators.  The device engine is the same one: the all-subsets
kind is the ``M = 0`` instantiation of the kernels (``csrc/spfm_kernels.hip.h``).
"""
import os
import warnings
from abc import ABCMeta, abstractmethod

import numpy as np
import scipy.sparse as sp
from sklearn.utils import check_random_state
from sklearn.utils.validation import NotFittedError, check_array

from .base import BaseSparsePoly, SparsePolyClassifierMixin, SparsePolyRegressorMixin
from .engine import HipEngine, canonical_csc
from .loss import CLASSIFICATION_LOSSES, REGRESSION_LOSSES
from .regularizer import L1, L21, OmegaCS, OmegaTI
from .schedule import Schedule


class _BaseSparseAllSubsets(BaseSparsePoly, metaclass=ABCMeta):
    # sparse_all_subsets.py:33-38
    _REGULARIZERS = {"l1": L1, "l21": L21, "omegacs": OmegaCS, "omegati": OmegaTI}

    @abstractmethod
    def __init__(self, loss="squared", n_components=2, solver="pcd", beta=1, gamma=1, eta0=0.1,
                 mean=False, tol=1e-6, regularizer="omegati", warm_start=False,
                 init_lambdas="ones", max_iter=100, shuffle=False, verbose=False, callback=None,
                 n_calls=10, random_state=None, schedule="exact", precision="f32", device=None):
        self.loss = loss
        self.n_components = n_components
        self.solver = solver
        self.beta = beta
        self.gamma = gamma
        self.eta0 = eta0
        self.mean = mean
        self.tol = tol
        self.regularizer = regularizer
        self.warm_start = warm_start
        self.init_lambdas = init_lambdas
        self.max_iter = max_iter
        self.shuffle = shuffle
        self.verbose = verbose
        self.callback = callback
        self.n_calls = n_calls
        self.random_state = random_state
        self.schedule = schedule
        self.precision = precision
        self.device = device

    def _new_engine(self):
        from .sparse_factorization_machines import _fit_device

        return HipEngine(device=_fit_device(self), precision=self.precision)

    def _set_schedule(self, engine, indices_feature):
        if isinstance(self.schedule, Schedule):
            self.feature_order_ = engine.install_schedule(self.schedule)
        else:
            self.feature_order_ = engine.set_schedule(self.schedule, indices_feature)

    def fit(self, X, y):
        """sparse_all_subsets.py:203-258 with _fit_pcd (:80-138) / _fit_pbcd (:140-201)."""
        X, y = self._check_X_y(X, y)
        n_samples, n_features = X.shape
        rng = check_random_state(self.random_state)
        self._get_loss(self.loss)
        self._get_regularizer(self.regularizer)
        if not (self.warm_start and hasattr(self, "P_")):
            self.P_ = 0.01 * rng.randn(self.n_components, n_features)
        if not (self.warm_start and hasattr(self, "lams_")):
            if self.init_lambdas == "ones":
                self.lams_ = np.ones(self.n_components)
            elif self.init_lambdas == "random_signs":
                self.lams_ = np.sign(rng.randn(self.n_components))
            else:
                raise ValueError(
                    "Lambdas must be initialized as ones "
                    "(init_lambdas='ones') or as random "
                    "+/- 1 (init_lambdas='random_signs')."
                )
        if self.solver not in ("pcd", "pbcd"):
            msg = f"Solver {self.solver} is not supported."
            raise ValueError(msg)
        if not isinstance(self.schedule, Schedule) and self.schedule not in ("exact", "colored"):
            raise ValueError("schedule must be 'exact', 'colored' or a Schedule object.")
        if isinstance(self.schedule, Schedule) and self.shuffle:
            raise ValueError("a fixed Schedule cannot be combined with shuffle=True.")
        beta = self.beta * n_samples if self.mean else self.beta
        gamma = self.gamma * n_samples if self.mean else self.gamma

        self.P_ = np.ascontiguousarray(self.P_, dtype=np.double)
        engine = self._new_engine()
        try:
            # canonical CSR goes to the library as it is (transposed on the device)
            csr_direct = sp.isspmatrix_csr(X) and X.has_canonical_format
            from . import engine as _engine_mod

            _engine_mod.shared_set_data(engine, X if csr_direct else canonical_csc(X), y)
            engine.set_params(self.P_[None], np.zeros(n_features), self.lams_)
            engine.configure(self.solver, self.loss, self.regularizer, -1)
            engine.init_pred(-1, False, False)  # y_pred = self._get_output(X) (:241)
            indices_feature = np.arange(n_features, dtype=np.int32)
            indices_component = np.arange(self.n_components, dtype=np.int32)
            if not self.shuffle:
                self._set_schedule(engine, indices_feature)
            converged = False
            it = 0
            for it in range(self.max_iter):
                viol = 0
                if self.shuffle:
                    if self.solver == "pcd":
                        rng.shuffle(indices_component)
                    rng.shuffle(indices_feature)
                    self._set_schedule(engine, indices_feature)
                if self.solver == "pcd":
                    viol += engine.pcd_epoch(0, -1, beta, gamma, self.eta0, indices_component)
                else:
                    viol += engine.pbcd_epoch(0, -1, beta, gamma, self.eta0)
                if (self.callback is not None) and it % self.n_calls == 0:
                    if self.solver == "pcd":  # pbcd trains a transposed copy (:167,199)
                        engine.get_params(self.P_[None], None)
                    if self.callback(self) is not None:
                        break
                if self.verbose:
                    print(f"Iteration {it+1} violation sum {viol}")
                if viol < self.tol:
                    if self.verbose:
                        print(f"Converged at iteration {it+1}")
                    converged = True
                    break
            engine.get_params(self.P_[None], None)
            self.n_iter_ = it
            self.n_steps_per_sweep_ = engine.n_batches
        finally:
            engine.close()
        if not converged:
            warnings.warn("Objective did not converge. Increase max_iter.")
        return self

    def fit_path(self, X, y, max_concurrent=None, **grid):
        """Clones of this estimator over a parameter grid (``gamma=[...]``, ...), fitted side by
        side on the GPU (sparsepoly_amd/concurrent.py); each equals its solo ``fit``."""
        from .concurrent import fit_path

        return fit_path(self, X, y, max_concurrent=max_concurrent, **grid)

    def _get_output(self, X):
        """sparse_all_subsets.py:260-263 (poly_predict(..., 'all-subsets')), on the device."""
        engine = self._new_engine()
        try:
            P = np.ascontiguousarray(self.P_, dtype=np.double)
            engine.set_params(P[None], np.zeros(P.shape[1]), self.lams_)
            return engine.predict(X, -1, False, False)
        finally:
            engine.close()

    def _predict(self, X):
        """sparse_all_subsets.py:265-269"""
        if not hasattr(self, "P_"):
            raise NotFittedError("Estimator not fitted.")
        X = check_array(X, accept_sparse=("csr", "csc"), dtype=np.double)  # row-major on the device
        return self._get_output(X)


class SparseAllSubsetsRegressor(_BaseSparseAllSubsets, SparsePolyRegressorMixin):
    """Sparse all-subsets model for regression (squared loss).
    Reference: sparse_all_subsets.py:272-390."""

    _LOSSES = REGRESSION_LOSSES

    def __init__(self, n_components=2, solver="pcd", beta=1, gamma=1, eta0=0.1, mean=False,
                 tol=1e-6, regularizer="omegati", warm_start=False, init_lambdas="ones",
                 max_iter=100, shuffle=False, verbose=False, callback=None, n_calls=10,
                 random_state=None, schedule="exact", precision="f32", device=None):
        super(SparseAllSubsetsRegressor, self).__init__(
            "squared", n_components, solver, beta, gamma, eta0, mean, tol, regularizer,
            warm_start, init_lambdas, max_iter, shuffle, verbose, callback, n_calls,
            random_state, schedule, precision, device)


class SparseAllSubsetsClassifier(_BaseSparseAllSubsets, SparsePolyClassifierMixin):
    """Sparse all-subsets model for binary classification.
    Reference: sparse_all_subsets.py:393-519."""

    _LOSSES = CLASSIFICATION_LOSSES

    def __init__(self, loss="squared_hinge", n_components=2, solver="pcd", beta=1, gamma=1,
                 eta0=0.1, mean=False, regularizer="omegati", tol=1e-6, warm_start=False,
                 init_lambdas="ones", max_iter=100, shuffle=False, verbose=False, callback=None,
                 n_calls=10, random_state=None, schedule="exact", precision="f32", device=None):
        # (regularizer before tol: the reference's keyword order, sparse_all_subsets.py:481-499)
        super(SparseAllSubsetsClassifier, self).__init__(
            loss, n_components, solver, beta, gamma, eta0, mean, tol, regularizer, warm_start,
            init_lambdas, max_iter, shuffle, verbose, callback, n_calls, random_state, schedule,
            precision, device)
