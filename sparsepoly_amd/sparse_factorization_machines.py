"""Sparse factorization machines on MI355X -- sklearn-style estimators.

Drop-in for ``sparsepoly.SparseFactorizationMachine{Regressor,Classifier}``
(reference ``sparsepoly/sparse_factorization_machines.py``) for
``solver in {'pcd', 'pbcd', 'psgd'}``: same constructor keywords in the same order with
the same defaults, same fitted attributes (``P_ (n_orders, k, d)``, ``w_``,
``lams_``, ``n_iter_`` = 0-based index of the last iteration), same warnings and
error types.  The epoch loops below restate ``_fit_pcd`` (:175-258),
``_fit_pbcd`` (:260-353) and ``_fit_psgd`` (:94-173); each reference epoch-function call is one call through
the C ABI (``include/spfm.h``) into hand-written HIP kernels.  There is no CPU
path: without the HIP library or a GPU, ``fit``/``predict`` raise.

Additional keywords (after the reference's, so positional use is unchanged):

``schedule``   a ``sparsepoly_amd.schedule.Schedule`` built earlier for this data (reused
               as is; fitted estimators expose theirs as ``schedule_``), or
               'exact' (default): coordinates are visited exactly in the
               reference's order (``np.arange`` or the shuffled order); consecutive
               columns that share no row are processed as one dependent step.
               'colored': the column conflict graph is coloured once and the
               colour classes are visited one after another -- the same algorithm
               run in the permuted order ``feature_order_`` (the reference's epoch
               functions take any order, pcd.py:86-87).
``precision``  'f32' (default) stores X, the ANOVA caches and y_pred in float32
               (reductions, prox and parameters stay float64); 'f64' stores them
               in float64.
``device``     HIP device ordinal (default: ``LOCAL_RANK`` or 0).
``warm_start=True`` additionally keeps the device session of the last fit (data, schedule,
row-block stream: SURVEY.md 8f N4): the next ``fit`` on the same ``X, y`` re-uploads only
the parameters, which is what a regularization path needs.  ``release_device()`` frees it.

``distributed`` shard the rows over the ranks of the initialised
               ``torch.distributed`` process group (one process per GPU); the
               per-step column partial sums are all-reduced with RCCL.
"""
import os
import warnings
from abc import ABCMeta, abstractmethod

import numpy as np
import scipy.sparse as sp
from sklearn.preprocessing import add_dummy_feature
from sklearn.utils import check_random_state
from sklearn.utils.validation import NotFittedError, check_array

from .base import BaseSparsePoly, SparsePolyClassifierMixin, SparsePolyRegressorMixin
from . import engine as _engine_mod
from .engine import HipEngine, canonical_csc
from .schedule import Schedule
from .loss import CLASSIFICATION_LOSSES, REGRESSION_LOSSES
from .regularizer import REGULARIZATION


def _default_device():
    return int(os.environ.get("LOCAL_RANK", "0"))


def _fit_device(est):
    """The device a fit runs on: the worker's device inside a ``fit_concurrently(devices=...)``
    call, else the estimator's ``device``, else LOCAL_RANK."""
    ten = _engine_mod.current_tenancy()
    if ten is not None and ten.device is not None:
        return int(ten.device)
    return _default_device() if est.device is None else est.device


def _fingerprint(Xc, y):
    """Cheap content hash of the training set (structure, values, targets): decides
    whether a warm-started fit may reuse the device-resident data of the previous one."""
    try:
        import xxhash

        h = xxhash.xxh3_64()
        upd, fin = h.update, h.hexdigest
    except Exception:  # pragma: no cover - xxhash is optional
        import zlib

        state = [1]

        def upd(b):
            state[0] = zlib.adler32(b, state[0])

        def fin():
            return "%08x" % state[0]

    for a in (Xc.indptr, Xc.indices, Xc.data, np.ascontiguousarray(y)):
        upd(memoryview(np.ascontiguousarray(a)).cast("B"))
    return (Xc.shape, int(Xc.nnz), fin())


class _BaseSparseFactorizationMachine(BaseSparsePoly, metaclass=ABCMeta):
    _REGULARIZERS = REGULARIZATION

    @abstractmethod
    def __init__(
        self,
        degree=2,
        loss="squared",
        n_components=2,
        solver="pcd",
        regularizer="squaredl12",
        alpha=1,
        beta=1,
        gamma=1,
        mean=False,
        tol=1e-6,
        fit_lower="explicit",
        fit_linear=True,
        warm_start=False,
        init_lambdas="ones",
        max_iter=100,
        shuffle=False,
        batch_size="auto",
        eta0=1.0,
        learning_rate="optimal",
        power_t=1.0,
        n_iter_no_change=5,
        verbose=False,
        callback=None,
        n_calls=10,
        random_state=None,
        schedule="exact",
        precision="f32",
        device=None,
        distributed=False,
    ):
        self.degree = degree
        self.loss = loss
        self.n_components = n_components
        self.solver = solver
        self.regularizer = regularizer
        self.alpha = alpha
        self.beta = beta
        self.gamma = gamma
        self.mean = mean
        self.tol = tol
        self.fit_lower = fit_lower
        self.fit_linear = fit_linear
        self.warm_start = warm_start
        self.init_lambdas = init_lambdas
        self.max_iter = max_iter
        self.shuffle = shuffle
        self.batch_size = batch_size
        self.eta0 = eta0
        self.learning_rate = learning_rate
        self.power_t = power_t
        self.n_iter_no_change = n_iter_no_change
        self.verbose = verbose
        self.callback = callback
        self.n_calls = n_calls
        self.random_state = random_state
        self.schedule = schedule
        self.precision = precision
        self.device = device
        self.distributed = distributed

    # ------------------------------------------------------------------ helpers
    def _augment(self, X):
        """sparse_factorization_machines.py:86-92"""
        if self.fit_lower == "augment":
            k = 2 if self.fit_linear else 1
            for _ in range(self.degree - k):
                X = add_dummy_feature(X, value=1)
        return X

    def _add_lower_deg2(self):
        """the order-2 term of _get_output, :445-449"""
        return self.fit_lower == "explicit" and self.degree == 3

    def _scaled(self, n_samples):
        """:181-188 / :265-272"""
        if self.mean:
            return self.alpha * n_samples, self.beta * n_samples, self.gamma * n_samples
        return self.alpha, self.beta, self.gamma

    def _new_engine(self):
        return HipEngine(device=_fit_device(self), precision=self.precision)

    def _set_schedule(self, engine, indices_feature, conflict_csc):
        """Fix the visiting order of the next epochs (the reference passes
        ``indices_feature`` to every epoch call: :198-205, :289-295)."""
        # a reused device session that already holds this very schedule (natural order, same
        # mode, or the same Schedule object) keeps it -- and its row-block stream
        tag = None
        if not self.shuffle:
            tag = ("obj", id(self.schedule)) if isinstance(self.schedule, Schedule) else \
                ("mode", self.schedule)
            if getattr(engine, "_sched_tag", None) == tag and engine.order is not None:
                self.schedule_ = self.schedule if isinstance(self.schedule, Schedule) else None
                self.feature_order_ = engine.order
                return engine.order
        engine._sched_tag = tag
        if isinstance(self.schedule, Schedule):
            order = engine.install_schedule(self.schedule, conflict_csc)
            self.schedule_ = self.schedule
        elif conflict_csc is None and self.schedule == "colored" and \
                getattr(self, "_struct_key", None) is not None:
            # a colouring is a function of the matrix structure and the visiting order (and of
            # what decides the step width): concurrent fits on one data set
            # (sparsepoly_amd/concurrent.py) compute it once, and so do later fits of this process
            mode = self.schedule
            jf = np.ascontiguousarray(indices_feature, dtype=np.int32)
            # (the step-width policy of spfm_set_schedule also looks at the CU share of a
            # co-tenant and at the device: both are part of the key, so that a colouring made for
            # a quarter of the CUs is never installed by a solo fit, or the reverse)
            key = (self._struct_key, mode, hash(jf.tobytes()), self.solver, self.loss,
                   self.precision, self.degree, self.fit_lower, self.fit_linear,
                   getattr(engine, "tenants", 1), getattr(engine, "device", 0))

            def compute():
                o = engine.set_schedule(mode, jf, None)
                return o, engine.get_schedule(mode)

            order = _engine_mod.shared_schedule(key, compute, engine.install_schedule)
            self.schedule_ = None  # filled in at the end of fit (needs the batch bounds)
        else:
            order = engine.set_schedule(self.schedule, indices_feature, conflict_csc)
            self.schedule_ = None
        self.feature_order_ = order
        return order

    def _sync_params(self, engine, with_P=True):
        """Copy the live device parameters into P_ / w_ (in place)."""
        engine.get_params(self.P_, self.w_, skip_P=not with_P)

    @staticmethod
    def _is_builtin_regularizer(obj):
        from . import regularizer as _r

        return type(obj) in (_r.L1, _r.L21, _r.SquaredL12, _r.SquaredL21, _r.OmegaTI, _r.OmegaCS)

    def _pcd_epoch(self, engine, order_idx, degree, beta, gamma, indices_component):
        """one ``pcd.pcd_epoch`` call: the device epoch, or -- plug-in regularizer -- the
        host-stepped one on the live ``self.P_[order_idx]`` (as the reference's view, :227-228)"""
        if self._plugin_reg is None:
            return engine.pcd_epoch(order_idx, degree, beta, gamma, self.eta0, indices_component)
        return engine.pcd_epoch_host(self._plugin_reg, self.P_[order_idx], self.lams_, self.loss,
                                     order_idx, degree, beta, gamma, self.eta0, indices_component)

    def _pbcd_epoch(self, engine, order_idx, degree, beta, gamma):
        if self._plugin_reg is None:
            return engine.pbcd_epoch(order_idx, degree, beta, gamma, self.eta0)
        return engine.pbcd_epoch_host(self._plugin_reg, self._Pt_host[order_idx], self.lams_,
                                      self.loss, order_idx, degree, beta, gamma, self.eta0)

    # ------------------------------------------------------------ epoch drivers
    def _fit_pcd(self, engine, n_samples, n_features, rng, conflict_csc):
        """Restates _fit_pcd, sparse_factorization_machines.py:175-258."""
        indices_feature = np.arange(n_features, dtype=np.int32)
        indices_component = np.arange(self.n_components, dtype=np.int32)
        converged = False
        alpha, beta, gamma = self._scaled(n_samples)
        if self._plugin_reg is not None:  # regularizer.init_cache_pcd(...) (:194)
            self._plugin_reg.init_cache_pcd(self.degree, n_features, self.n_components)
        if not self.shuffle:
            self._set_schedule(engine, indices_feature, conflict_csc)
        it = 0
        for it in range(self.max_iter):
            viol = 0
            if self.shuffle:
                rng.shuffle(indices_component)
                rng.shuffle(indices_feature)
                self._set_schedule(engine, indices_feature, conflict_csc)
            if self.fit_linear:
                viol += engine.cd_linear_epoch(alpha)
            if self.fit_lower == "explicit":
                for deg in range(2, self.degree):
                    viol += self._pcd_epoch(engine, self.degree - deg, deg, beta, gamma,
                                            indices_component)
            viol += self._pcd_epoch(engine, 0, self.degree, beta, gamma, indices_component)

            if (self.callback is not None) and it % self.n_calls == 0:
                self._sync_params(engine)  # pcd writes through self.P_[0] (:227-228)
                if self.callback(self) is not None:
                    break
            if self.verbose:
                print(f"Iteration {it+1} violation sum {viol}")
            if viol < self.tol:
                if self.verbose:
                    print(f"Converged at iteration {it+1}")
                converged = True
                break
        self._sync_params(engine)
        return converged, it

    def _fit_pbcd(self, engine, n_samples, n_features, rng, conflict_csc):
        """Restates _fit_pbcd, sparse_factorization_machines.py:260-353.  The reference
        trains a transposed COPY of P_ (:285) and writes it back after the loop (:352),
        so callbacks see the initial P_ but the live w_; kept."""
        indices_feature = np.arange(n_features, dtype=np.int32)
        converged = False
        alpha, beta, gamma = self._scaled(n_samples)
        if self._plugin_reg is not None:
            # regularizer.init_cache_pbcd(...) (:282) and the transposed working copy (:285)
            self._plugin_reg.init_cache_pbcd(self.degree, n_features, self.n_components)
            self._Pt_host = np.array(self.P_.swapaxes(1, 2))
        if not self.shuffle:
            self._set_schedule(engine, indices_feature, conflict_csc)
        it = 0
        for it in range(self.max_iter):
            viol = 0
            if self.shuffle:
                rng.shuffle(indices_feature)
                self._set_schedule(engine, indices_feature, conflict_csc)
            if self.fit_linear:
                viol += engine.cd_linear_epoch(alpha)
            if self.fit_lower == "explicit":
                for deg in range(2, self.degree):
                    viol += self._pbcd_epoch(engine, self.degree - deg, deg, beta, gamma)
            viol += self._pbcd_epoch(engine, 0, self.degree, beta, gamma)

            if (self.callback is not None) and it % self.n_calls == 0:
                self._sync_params(engine, with_P=False)
                if self.callback(self) is not None:
                    break
            if self.verbose:
                print(f"Iteration {it+1} violation sum {viol}")
            if viol < self.tol:
                if self.verbose:
                    print(f"Converged at iteration {it+1}")
                converged = True
                break
        self._sync_params(engine)
        return converged, it

    def _fit_psgd(self, engine, n_samples, n_features, nnz, rng):
        """Restates _fit_psgd, sparse_factorization_machines.py:94-173.  Like pbcd the
        reference trains a transposed copy of P_ (:113) and writes it back after the loop
        (:172), while w_ is updated in place: callbacks see the initial P_ and the live w_."""
        from ._capi import LEARNING_RATE

        indices_samples = np.arange(n_samples, dtype=np.int32)
        converged = False
        no_improvement_count = 0
        best_loss = np.inf
        if self.batch_size == "auto":
            batch_size = int(n_samples * n_features / nnz)
        else:
            batch_size = self.batch_size
        if self.learning_rate not in LEARNING_RATE:
            msg = f"learning_rate {self.learning_rate} is not supported."
            msg += f" Choose from {LEARNING_RATE}."
            raise ValueError(msg)
        epoch = 0
        for epoch in range(self.max_iter):
            if self.shuffle:
                rng.shuffle(indices_samples)
            # (distributed: the global order on every rank; each forms the gradient of its own
            # rows of a minibatch, the sums are all-reduced -- spfm_psgd_epoch_sharded)
            sum_loss, self.it_ = engine.psgd_epoch(
                self.degree, self.alpha, self.beta, self.gamma, self.eta0, self.learning_rate,
                self.power_t, batch_size, indices_samples, self.fit_linear, self.it_,
                row_lo=self._row_lo if self.distributed else None)
            if (self.callback is not None) and epoch % self.n_calls == 0:
                self._sync_params(engine, with_P=False)
                if self.callback(self) is not None:
                    break
            sum_loss /= n_samples
            if self.verbose:
                print(f"Epoch {epoch+1} loss {sum_loss}")
            if sum_loss > (best_loss - self.tol):
                no_improvement_count += 1
            else:
                no_improvement_count = 0
            if sum_loss < best_loss:
                best_loss = sum_loss
            if no_improvement_count >= self.n_iter_no_change:
                if self.verbose:
                    print(f"Converged at iteration {epoch+1}")
                converged = True
                break
        self._sync_params(engine)
        return converged, epoch

    # ---------------------------------------------------------------------- fit
    def fit(self, X, y):
        """Fit factorization machine to training data
        (sparse_factorization_machines.py:355-435).

        With ``distributed=True`` every rank passes the same global ``X, y``; each
        rank trains on its contiguous block of rows.
        """
        X, y = self._check_X_y(X, y)
        X = self._augment(X)
        n_samples, n_features = X.shape
        rng = check_random_state(self.random_state)
        self._get_loss(self.loss)
        reg_obj = self._get_regularizer(self.regularizer)
        # a registered regularizer that is not one of the six built-in classes cannot run in the
        # device chains: its epochs are stepped from the host, calling the object's own prox and
        # cache hooks (include/spfm.h "host-stepped epochs")
        self._plugin_reg = None if self._is_builtin_regularizer(reg_obj) else reg_obj

        if not (self.warm_start and hasattr(self, "w_")):
            self.w_ = np.zeros(n_features, dtype=np.double)

        if self.fit_lower == "explicit":
            n_orders = self.degree - 1
        else:
            n_orders = 1

        if not (self.warm_start and hasattr(self, "P_")):
            self.P_ = 0.01 * rng.randn(n_orders, self.n_components, n_features)

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

        if np.unique(np.abs(self.lams_)) != np.array([1.0]):
            raise ValueError("Lambdas must be +1 or -1.")

        if self.solver not in ("pcd", "pbcd", "psgd"):
            msg = f"Solver {self.solver} is not supported."
            raise ValueError(msg)
        if self.solver == "psgd":
            if not (self.warm_start and hasattr(self, "it_")):
                self.it_ = 1  # :411-412
        if not isinstance(self.schedule, Schedule) and self.schedule not in ("exact", "colored"):
            raise ValueError("schedule must be 'exact', 'colored' or a Schedule object.")
        if isinstance(self.schedule, Schedule) and self.shuffle:
            raise ValueError("a fixed Schedule cannot be combined with shuffle=True.")

        # canonical CSR goes to the library as it is (threaded transposition inside); everything
        # else -- and the paths that need the CSC on the host -- through scipy
        csr_direct = sp.isspmatrix_csr(X) and X.has_canonical_format and not self.distributed
        Xc = None if csr_direct else canonical_csc(X)
        self._struct_key = None
        if self.schedule == "colored" and not self.distributed and not self.shuffle:
            self._struct_key = _engine_mod.structure_key(X if csr_direct else Xc)
        conflict_csc = None
        # warm_start keeps the device session: same data => no re-upload,
        # no re-colouring, no new row-block stream
        key = None
        engine = None
        if self.warm_start and not self.distributed:
            key = (_fingerprint(X if csr_direct else Xc, y), "csr" if csr_direct else "csc",
                   self.precision, _fit_device(self))
            cached = getattr(self, "_device_session", None)
            self._device_session = None
            if cached is not None:
                if cached[0] == key:
                    engine = cached[1]
                else:
                    cached[1].close()
        fresh = engine is None
        if fresh:
            engine = self._new_engine()
        keep = False
        try:
            if not fresh:
                pass
            elif self.distributed:
                from . import distributed as _dist

                lo, hi = _dist.row_block(n_samples)
                self._row_lo = lo
                conflict_csc = Xc
                Xl = canonical_csc(Xc.tocsr()[lo:hi])
                _dist.init_engine_comm(engine)
                if self.solver != "psgd":  # (psgd: one collective per minibatch, no persistent pass)
                    _dist.connect_peers(engine)  # in-kernel exchange for the persistent passes
                engine.set_data(Xl, y[lo:hi])
            else:
                # (concurrent fits on one data set attach to ONE device image of it)
                self.shared_image_ = _engine_mod.shared_set_data(engine, X if csr_direct else Xc, y)
            self.P_ = np.ascontiguousarray(self.P_, dtype=np.double)
            self.w_ = np.ascontiguousarray(self.w_, dtype=np.double)
            engine.set_params(self.P_, self.w_, self.lams_)
            # regularizer.init_cache_pcd / init_cache_pbcd (:194, :282), incl. their errors
            if self._plugin_reg is not None:
                if self.solver == "psgd":
                    raise ValueError("solver='psgd' needs one of the built-in regularizers "
                                     "(its full-matrix prox runs on the device).")
                if self.distributed:
                    raise ValueError("a user-defined regularizer object runs on one GPU "
                                     "(distributed=False).")
                # the device-side regularizer is never called on this path; 'l1' only satisfies
                # the solver / regularizer pairing check of spfm_configure
                engine.configure(self.solver, self.loss, "l1", self.degree)
            else:
                engine.configure(self.solver, self.loss, self.regularizer, self.degree)
            if self.solver == "psgd":
                # X.count_nonzero() (dataset.py:32,84): stored entries, n*d when dense
                nnz = X.nnz if sp.issparse(X) else n_samples * n_features
                converged, self.n_iter_ = self._fit_psgd(engine, n_samples, n_features,
                                                         nnz, rng)
                keep = key is not None
                return self._finish_fit(converged)
            # y_pred = self._get_output(X) (:408)
            engine.init_pred(self.degree, self.fit_linear, self._add_lower_deg2())
            if self.solver == "pcd":
                converged, self.n_iter_ = self._fit_pcd(engine, n_samples, n_features, rng,
                                                        conflict_csc)
            else:
                converged, self.n_iter_ = self._fit_pbcd(engine, n_samples, n_features, rng,
                                                         conflict_csc)
            self.n_steps_per_sweep_ = engine.n_batches
            self.__dict__.pop("_Pt_host", None)
            if self.schedule_ is None:
                self.schedule_ = engine.get_schedule(self.schedule)
            keep = key is not None
        finally:
            if keep:
                self._device_session = (key, engine)
            else:
                engine.close()
        return self._finish_fit(converged)

    def fit_path(self, X, y, max_concurrent=None, **grid):
        """Clones of this estimator over a parameter grid (``gamma=[...]``, ``beta=[...]``, ... all
        of one length), fitted side by side on the GPU; see sparsepoly_amd/concurrent.py.  Each
        clone equals its own solo ``fit`` bit for bit."""
        from .concurrent import fit_path

        return fit_path(self, X, y, max_concurrent=max_concurrent, **grid)

    def release_device(self):
        """Free the device session kept by ``warm_start=True`` (data, schedule, stream)."""
        cached = getattr(self, "_device_session", None)
        self._device_session = None
        if cached is not None:
            cached[1].close()

    def __getstate__(self):
        state = dict(super().__getstate__())
        state.pop("_device_session", None)  # a device handle is not picklable
        state.pop("_plugin_reg", None)      # a user's regularizer object may not be either
        state.pop("_Pt_host", None)
        return state

    def _finish_fit(self, converged):
        if not converged:
            warnings.warn("Objective did not converge. Increase max_iter.")
        return self

    # ------------------------------------------------------------------ predict
    def _get_output(self, X):
        """sparse_factorization_machines.py:437-451, on the device."""
        engine = self._new_engine()
        try:
            engine.set_params(np.ascontiguousarray(self.P_, dtype=np.double), self.w_,
                              self.lams_)
            return engine.predict(X, self.degree, self.fit_linear, self._add_lower_deg2())
        finally:
            engine.close()

    def _predict(self, X):
        """sparse_factorization_machines.py:453-458"""
        if not hasattr(self, "P_"):
            raise NotFittedError("Estimator not fitted.")
        # (the reference asks for CSC here; the device predict pass is row-major, so CSR input
        # is taken as it is instead of being converted twice)
        X = check_array(X, accept_sparse=("csr", "csc"), dtype=np.double)
        X = self._augment(X)
        return self._get_output(X)


class SparseFactorizationMachineRegressor(_BaseSparseFactorizationMachine,
                                          SparsePolyRegressorMixin):
    """Sparse factorization machine for regression (squared loss).

    Reference: sparse_factorization_machines.py:461-684.
    """

    _LOSSES = REGRESSION_LOSSES

    def __init__(
        self,
        degree=2,
        n_components=2,
        solver="pcd",
        regularizer="squaredl12",
        alpha=1,
        beta=1,
        gamma=1,
        mean=False,
        tol=1e-6,
        fit_lower="explicit",
        fit_linear=True,
        warm_start=False,
        init_lambdas="ones",
        max_iter=100,
        shuffle=False,
        batch_size="auto",
        eta0=1.0,
        learning_rate="optimal",
        power_t=1.0,
        n_iter_no_change=5,
        verbose=False,
        callback=None,
        n_calls=10,
        random_state=None,
        schedule="exact",
        precision="f32",
        device=None,
        distributed=False,
    ):
        super(SparseFactorizationMachineRegressor, self).__init__(
            degree, "squared", n_components, solver, regularizer, alpha, beta, gamma, mean, tol,
            fit_lower, fit_linear, warm_start, init_lambdas, max_iter, shuffle, batch_size, eta0,
            learning_rate, power_t, n_iter_no_change, verbose, callback, n_calls, random_state,
            schedule, precision, device, distributed,
        )


class SparseFactorizationMachineClassifier(_BaseSparseFactorizationMachine,
                                           SparsePolyClassifierMixin):
    """Sparse factorization machine for binary classification.

    Reference: sparse_factorization_machines.py:687-920.
    """

    _LOSSES = CLASSIFICATION_LOSSES

    def __init__(
        self,
        degree=2,
        loss="squared_hinge",
        n_components=2,
        solver="pcd",
        regularizer="squaredl12",
        alpha=1,
        beta=1,
        gamma=1,
        mean=False,
        tol=1e-6,
        fit_lower="explicit",
        fit_linear=True,
        warm_start=False,
        init_lambdas="ones",
        max_iter=100,
        shuffle=False,
        batch_size="auto",
        eta0=1.0,
        learning_rate="optimal",
        power_t=1.0,
        n_iter_no_change=5,
        verbose=False,
        callback=None,
        n_calls=10,
        random_state=None,
        schedule="exact",
        precision="f32",
        device=None,
        distributed=False,
    ):
        super(SparseFactorizationMachineClassifier, self).__init__(
            degree, loss, n_components, solver, regularizer, alpha, beta, gamma, mean, tol,
            fit_lower, fit_linear, warm_start, init_lambdas, max_iter, shuffle, batch_size, eta0,
            learning_rate, power_t, n_iter_no_change, verbose, callback, n_calls, random_state,
            schedule, precision, device, distributed,
        )
