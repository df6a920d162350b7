"""Python face of one ``spfm_handle`` (include/spfm.h): NumPy in, NumPy out.

The methods map one-to-one onto the reference calls they replace
(``sparsepoly/sparse_factorization_machines.py`` epoch drivers):

=====================  =====================================================
``set_data``           ``get_dataset(X, 'fortran')`` + ``row_norms`` (:406-409)
``init_pred``          ``_get_output(X)`` (:408, :437-451)
``cd_linear_epoch``    ``cd_linear._cd_linear_epoch`` (optimizer/cd_linear.py:8-33)
``pcd_epoch``          ``pcd.pcd_epoch`` (optimizer/pcd.py:71-137)
``pbcd_epoch``         ``pbcd.pbcd_epoch`` (optimizer/pbcd.py:82-148)
``predict``            ``_predict`` (:453-458)
=====================  =====================================================
"""
import collections
import ctypes as C
import threading

import numpy as np
import scipy.sparse as sp

from . import _capi


class SpfmError(RuntimeError):
    pass


def canonical_csc(X):
    """CSC with sorted, duplicate-free row indices, float64.  Dense input is
    converted entry-wise (the reference's FortranDataset, dataset.py:39-57, also
    visits explicit zeros; they contribute exact zeros to every sum and update)."""
    if sp.issparse(X):
        Xc = sp.csc_matrix(X, dtype=np.float64, copy=True)
        Xc.sum_duplicates()
        Xc.sort_indices()
    else:
        Xc = sp.csc_matrix(np.asarray(X, dtype=np.float64))
        Xc.sort_indices()
    return Xc


# Engines created inside `co_tenancy(n)` -- or by the worker threads of one
# `fit_concurrently` call -- announce that n of them train side by side on one GPU
# (sparsepoly_amd/concurrent.py): every persistent pass keeps to 1/n of the CUs.  The state of
# such a call is a `Tenancy` object bound to the THREADS that take part in it (thread-local), so
# two calls that overlap in time from different threads do not see each other's shares.
_TLS = threading.local()
_SHARED_LOCK = threading.Lock()
_SCHEDULE_LRU = collections.OrderedDict()
_SCHEDULE_LRU_SIZE = 8
_SCHEDULE_STATS = {"hits": 0, "misses": 0}


class Tenancy(object):
    """What the fits of ONE concurrent call share: the number of tenants per device, the device
    a worker thread fits on (``devices=[...]`` fan-out), and -- one data set for all -- the
    coloured schedule and the device data image (first come computes / uploads, the others
    install / attach)."""

    def __init__(self, n, share_schedules=False, share_data=False, device=None):
        self.n = max(1, int(n))
        self.schedules = {} if share_schedules else None
        self.images = {} if share_data else None  # key -> {"done": Event, "keeper": HipEngine}
        self.device = device
        self.lock = threading.Lock()

    def on_device(self, device):
        """The same call's state as seen by a worker bound to `device` (shared caches are per
        device: a schedule object is portable, a device image is not)."""
        t = Tenancy.__new__(Tenancy)
        t.n, t.lock, t.device = self.n, self.lock, device
        t.schedules, t.images = self.schedules, self.images
        return t

    def close(self):
        if self.images:
            for ent in self.images.values():
                k = ent.get("keeper")
                if k is not None:
                    k.close()
            self.images.clear()


def current_tenancy():
    return getattr(_TLS, "tenancy", None)


def bind_tenancy(tenancy):
    """Bind `tenancy` (or None) to the calling thread; returns what was bound before."""
    old = getattr(_TLS, "tenancy", None)
    _TLS.tenancy = tenancy
    return old


def structure_key(X):
    """Content hash of a sparse matrix's STRUCTURE (shape, format, indptr, indices): what a
    colouring depends on.  ~0.05 s for 50 M entries."""
    try:
        import xxhash

        h = xxhash.xxh3_64()
    except Exception:  # pragma: no cover - xxhash is optional
        import hashlib

        h = hashlib.blake2b(digest_size=8)
    for a in (X.indptr, X.indices):
        h.update(memoryview(np.ascontiguousarray(a)).cast("B"))
    return (X.format, X.shape, int(X.nnz), h.hexdigest())


class co_tenancy(object):
    """``with co_tenancy(n):`` -- engines created by THIS thread inside the block share the device
    with n - 1 others (per-call state: see Tenancy)."""

    def __init__(self, n, share_schedules=False, share_data=False, device=None):
        self.tenancy = Tenancy(n, share_schedules, share_data, device)

    def __enter__(self):
        if self.tenancy.n > 1:
            _capi.ensure_hw_queues(self.tenancy.n)
        self._old = bind_tenancy(self.tenancy)
        return self.tenancy

    def __exit__(self, *exc):
        bind_tenancy(self._old)
        self.tenancy.close()
        return False


def shared_schedule(key, compute, install):
    """Inside a concurrent call that shares schedules: ``compute()`` (-> order, Schedule) runs in
    the first thread that asks for ``key``; every other thread waits for it and calls
    ``install(schedule)`` (-> order).  Outside such a call the process-wide memory of schedules
    is consulted instead."""
    ten = current_tenancy()
    cache = ten.schedules if ten is not None else None
    if cache is None:
        # one fit at a time: a small process-wide memory of coloured schedules, so that a second
        # fit on the same matrix in the same visiting order (a grid search, a restart) does not
        # colour it again (0.5 s on BASELINE config 2, 7 s on configs[4])
        with _SHARED_LOCK:
            sched = _SCHEDULE_LRU.get(key)
            if sched is not None:
                _SCHEDULE_LRU.move_to_end(key)
            _SCHEDULE_STATS["hits" if sched is not None else "misses"] += 1
        if sched is not None:
            return install(sched)
        order, sched = compute()
        with _SHARED_LOCK:
            _SCHEDULE_LRU[key] = sched
            while len(_SCHEDULE_LRU) > _SCHEDULE_LRU_SIZE:
                _SCHEDULE_LRU.popitem(last=False)
        return order
    with ten.lock:
        entry = cache.get(key)
        leader = entry is None
        if leader:
            entry = cache[key] = {"done": threading.Event(), "sched": None}
    if leader:
        try:
            order, entry["sched"] = compute()
        finally:
            entry["done"].set()     # on an error the followers compute for themselves
        return order
    entry["done"].wait()
    if entry["sched"] is None:
        return compute()[0]
    return install(entry["sched"])


def data_key(X, precision, device):
    """Content hash of a sparse matrix (structure AND values) + what else decides its device
    image."""
    try:
        import xxhash

        h = xxhash.xxh3_64()
    except Exception:  # pragma: no cover - xxhash is optional
        import hashlib

        h = hashlib.blake2b(digest_size=8)
    for a in (X.indptr, X.indices, X.data):
        h.update(memoryview(np.ascontiguousarray(a)).cast("B"))
    return (X.format, X.shape, int(X.nnz), h.hexdigest(), precision, int(device))


def shared_set_data(engine, X, y):
    """``engine.set_data(X, y)`` -- or, inside a concurrent call on ONE data set, attach to the
    device image the first fit of that call uploaded (Tenancy.images; kept alive by a keeper
    handle until the call ends)."""
    ten = current_tenancy()
    if ten is None or ten.images is None or not sp.issparse(X):
        engine.set_data(X, y)
        return False
    key = data_key(X, engine.precision, engine.device)
    with ten.lock:
        entry = ten.images.get(key)
        leader = entry is None
        if leader:
            entry = ten.images[key] = {"done": threading.Event(), "keeper": None}
    if leader:
        try:
            engine.set_data(X, y)
            keeper = HipEngine(engine.device, engine.precision)
            keeper.share_data(engine)
            entry["keeper"] = keeper
        finally:
            entry["done"].set()  # on an error the followers upload for themselves
        return False
    entry["done"].wait()
    if entry["keeper"] is None:
        engine.set_data(X, y)
        return False
    engine.share_data(entry["keeper"], y)
    return True


def hw_queue_report():
    """How many hardware queues the HIP runtime of this process maps streams onto, as far as the
    environment tells (GPU_MAX_HW_QUEUES is read when the runtime initialises; default 4).  More
    concurrent fits than queues share queues, and two persistent passes on one queue run one
    after the other (independent fits side by side)."""
    import os

    val = os.environ.get("GPU_MAX_HW_QUEUES")
    return {"GPU_MAX_HW_QUEUES": int(val) if val and val.isdigit() else None,
            "runtime_default": 4,
            "set_by_package": bool(_capi.QUEUES_SET_BY_PACKAGE),
            "hip_initialised_before_import": bool(_capi.HIP_INITIALISED_BEFORE_IMPORT)}


class HipEngine(object):
    def __init__(self, device=0, precision="f32"):
        if precision not in _capi.DTYPES:
            raise ValueError("precision must be 'f32' or 'f64'")
        self._lib = _capi.load()
        h = C.c_void_p()
        rc = self._lib.spfm_create(C.byref(h), int(device), _capi.DTYPES[precision])
        if rc != 0:
            msg = self._lib.spfm_last_error(None).decode()
            raise SpfmError("spfm_create failed: %s" % msg)
        self._h = h
        self.precision = precision
        self.n = self.d = self.k = self.n_orders = None
        self.order = None
        self.n_batches = None
        self.device = int(device)
        ten = current_tenancy()
        self.tenants = ten.n if ten is not None else 1
        if self.tenants > 1:
            self.set_option("co_tenants", self.tenants)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.spfm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc == 0:
            return
        msg = self._lib.spfm_last_error(self._h).decode()
        if rc == _capi.SPFM_ERR_INVALID:
            raise ValueError(msg)
        if rc == _capi.SPFM_ERR_UNSUPPORTED:
            raise NotImplementedError(msg)
        raise SpfmError(msg)

    @property
    def device_name(self):
        buf = C.create_string_buffer(128)
        self._check(self._lib.spfm_device_name(self._h, buf, 128))
        return buf.value.decode()

    # ------------------------------------------------------------------ data
    def set_data(self, X, y):
        if sp.isspmatrix_csr(X) and X.has_canonical_format:
            # CSR input: no scipy tocsc(); the library transposes on host threads
            n, d = X.shape
            keep = [_capi.i64(X.indptr), _capi.i32(X.indices), _capi.f64(X.data), _capi.f64(y)]
            if keep[3][0].shape[0] != n:
                raise ValueError("y has %d entries, X has %d rows" % (keep[3][0].shape[0], n))
            self._check(self._lib.spfm_set_data_csr(self._h, n, d, keep[0][1], keep[1][1],
                                                    keep[2][1], keep[3][1]))
            self.n, self.d = n, d
            return
        Xc = X if (sp.isspmatrix_csc(X) and X.has_canonical_format) else canonical_csc(X)
        n, d = Xc.shape
        self._keep = [_capi.i64(Xc.indptr), _capi.i32(Xc.indices), _capi.f64(Xc.data),
                      _capi.f64(y)]
        if self._keep[3][0].shape[0] != n:
            raise ValueError("y has %d entries, X has %d rows" % (self._keep[3][0].shape[0], n))
        self._check(self._lib.spfm_set_data_csc(
            self._h, n, d, self._keep[0][1], self._keep[1][1], self._keep[2][1],
            self._keep[3][1]))
        self._keep = None
        self.n, self.d = n, d

    def share_data(self, src, y=None):
        """Train on the device image of the matrix that engine ``src`` holds (no upload, no
        second copy in HBM); ``y``: this engine's own targets, default ``src``'s."""
        if y is None:
            yp = None
        else:
            ya, yp = _capi.f64(y)
            if ya.shape[0] != src.n:
                raise ValueError("y has %d entries, X has %d rows" % (ya.shape[0], src.n))
        self._check(self._lib.spfm_share_data(self._h, src._h, yp))
        self.n, self.d = src.n, src.d

    def set_params(self, P, w, lams):
        P = np.ascontiguousarray(P, dtype=np.float64)
        if P.ndim != 3:
            raise ValueError("P must be (n_orders, n_components, n_features)")
        n_orders, k, d = P.shape
        Pa, Pp = _capi.f64(P)
        wa, wp = _capi.f64(w)
        la, lp = _capi.f64(lams)
        if wa.shape[0] != d or la.shape[0] != k:
            raise ValueError("w / lams shapes do not match P")
        self._check(self._lib.spfm_set_params(self._h, n_orders, k, d, Pp, wp, lp))
        self.n_orders, self.k, self.d = n_orders, k, d

    def get_params(self, P=None, w=None, skip_P=False):
        """Copy the live device state into P (n_orders, k, d) and w (d), in place when
        arrays are given.  ``skip_P`` leaves P untouched (pbcd callbacks)."""
        if P is None and not skip_P:
            P = np.empty((self.n_orders, self.k, self.d))
        if w is None:
            w = np.empty(self.d)
        assert w.flags.c_contiguous and w.dtype == np.float64
        Pp = None
        if not skip_P:
            assert P.flags.c_contiguous and P.dtype == np.float64
            Pp = P.ctypes.data_as(_capi._dp)
        self._check(self._lib.spfm_get_params(self._h, Pp, w.ctypes.data_as(_capi._dp)))
        return P, w

    def configure(self, solver, loss, regularizer, degree):
        if solver not in _capi.SOLVERS:
            raise ValueError("Solver %s is not supported." % solver)
        self._check(self._lib.spfm_configure(
            self._h, _capi.SOLVERS[solver], _capi.LOSSES[loss], _capi.REGULARIZERS[regularizer],
            int(degree)))

    def init_pred(self, degree, fit_linear, add_lower_deg2):
        self._check(self._lib.spfm_init_pred(self._h, int(degree), int(bool(fit_linear)),
                                             int(bool(add_lower_deg2))))

    def get_y_pred(self):
        out = np.empty(self.n)
        self._check(self._lib.spfm_get_y_pred(self._h, out.ctypes.data_as(_capi._dp)))
        return out

    def loss_sum(self):
        out = C.c_double()
        self._check(self._lib.spfm_loss_sum(self._h, C.byref(out)))
        return out.value

    def predict(self, X, degree, fit_linear, add_lower_deg2):
        Xr = sp.csr_matrix(X, dtype=np.float64)
        Xr.sum_duplicates()
        n = Xr.shape[0]
        if Xr.shape[1] != self.d:
            raise ValueError("X has %d features, the model has %d" % (Xr.shape[1], self.d))
        ia, ip = _capi.i64(Xr.indptr)
        ja, jp = _capi.i32(Xr.indices)
        da, dp = _capi.f64(Xr.data)
        out = np.zeros(n)
        self._check(self._lib.spfm_predict_csr(
            self._h, n, ip, jp, dp, int(degree), int(bool(fit_linear)),
            int(bool(add_lower_deg2)), out.ctypes.data_as(_capi._dp)))
        return out

    # -------------------------------------------------------------- schedule
    def set_schedule(self, mode, indices_feature, conflict_csc=None):
        """Fix the coordinate order for the next epochs; returns the order used.
        ``conflict_csc``: global CSC structure (multi-GPU) for the disjointness test."""
        jf, jp = _capi.i32(indices_feature)
        order = np.empty(self.d, dtype=np.int32)
        nb = C.c_int32()
        if conflict_csc is None:
            cp, ci, rows = None, None, 0
        else:
            cpa, cp = _capi.i64(conflict_csc.indptr)
            cia, ci = _capi.i32(conflict_csc.indices)
            rows = conflict_csc.shape[0]
        self._check(self._lib.spfm_set_schedule(
            self._h, _capi.SCHEDULES[mode], jp, cp, ci, rows,
            order.ctypes.data_as(_capi._ip), C.byref(nb)))
        self.order, self.n_batches = order, nb.value
        return order

    def get_schedule(self, mode="colored"):
        """The installed schedule as a reusable ``Schedule`` object."""
        from .schedule import Schedule

        nb = C.c_int32()
        self._check(self._lib.spfm_get_schedule(self._h, None, None, C.byref(nb)))
        order = np.empty(self.d, dtype=np.int32)
        bp = np.empty(nb.value + 1, dtype=np.int32)
        self._check(self._lib.spfm_get_schedule(self._h, order.ctypes.data_as(_capi._ip),
                                                bp.ctypes.data_as(_capi._ip), C.byref(nb)))
        return Schedule(order, bp, mode, (self.n, self.d))

    def install_schedule(self, sched, conflict_csc=None):
        """Install a precomputed ``sparsepoly_amd.schedule.Schedule`` (validated by the
        library against the data / the global structure)."""
        oa, op = _capi.i32(sched.order)
        ba, bp = _capi.i32(sched.batch_ptr)
        if oa.shape[0] != self.d:
            raise ValueError("schedule has %d features, the data has %d" % (oa.shape[0], self.d))
        if conflict_csc is None:
            cp, ci, rows = None, None, 0
        else:
            cpa, cp = _capi.i64(conflict_csc.indptr)
            cia, ci = _capi.i32(conflict_csc.indices)
            rows = conflict_csc.shape[0]
        self._check(self._lib.spfm_set_schedule_raw(self._h, op, bp, ba.shape[0] - 1, cp, ci,
                                                    rows))
        self.order, self.n_batches = oa.copy(), ba.shape[0] - 1
        return self.order

    # ---------------------------------------------------------------- epochs
    def cd_linear_epoch(self, alpha):
        v = C.c_double()
        self._check(self._lib.spfm_cd_linear_epoch(self._h, float(alpha), C.byref(v)))
        return v.value

    def pcd_epoch(self, order_idx, degree, beta, gamma, eta, indices_component):
        ic, icp = _capi.i32(indices_component)
        v = C.c_double()
        self._check(self._lib.spfm_pcd_epoch(
            self._h, int(order_idx), int(degree), float(beta), float(gamma), float(eta), icp,
            ic.shape[0], C.byref(v)))
        return v.value

    def pbcd_epoch(self, order_idx, degree, beta, gamma, eta):
        v = C.c_double()
        self._check(self._lib.spfm_pbcd_epoch(
            self._h, int(order_idx), int(degree), float(beta), float(gamma), float(eta),
            C.byref(v)))
        return v.value

    # ---------------------------------------- host-stepped epochs (plug-in regularizers)
    _MU = {"squared": 1.0, "logistic": 0.25, "squared_hinge": 2.0}  # loss.py:18,32,59

    def _host_steps(self):
        sched = self.get_schedule()
        return sched.order, sched.batch_ptr

    def pcd_epoch_host(self, reg, P, lams, loss, order_idx, degree, beta, gamma, eta,
                       indices_component):
        """``pcd.pcd_epoch`` (optimizer/pcd.py:71-137) for a regularizer OBJECT that is not one
        of the device built-ins: the device forms every dependent step's column sums and
        scatter-updates, the object's ``compute_cache_pcd_all / compute_cache_pcd / prox_cd /
        update_cache_pcd`` run here, column by column in visiting order (include/spfm.h
        "host-stepped epochs").  ``P``: the (n_components, n_features) host array of this order,
        updated in place (it mirrors the device copy).  Returns sum_viol."""
        mu = self._MU[loss]
        order, bp = self._host_steps()
        self._check(self._lib.spfm_host_epoch_begin(self._h, int(order_idx), int(degree)))
        reg.compute_cache_pcd_all(P, degree)                                   # pcd.py:91
        viol = 0.0
        sums = np.empty((int(np.diff(bp).max()), 2))
        for s in indices_component:                                            # :92
            s = int(s)
            self._check(self._lib.spfm_host_pass_begin(self._h, s))
            reg.compute_cache_pcd(P, degree, s)                                # :96
            for b in range(len(bp) - 1):
                cols = order[bp[b]:bp[b + 1]]
                nc = len(cols)
                if nc == 0:
                    continue
                self._check(self._lib.spfm_host_step_sums(self._h, b,
                                                          sums.ctypes.data_as(_capi._dp)))
                p_new = np.empty(nc)
                for q in range(nc):                                            # :97, in order
                    j = int(cols[q])
                    p_old = float(P[s, j])
                    inv = mu * sums[q, 1] + beta                               # :61-62
                    upd = (lams[s] * sums[q, 0] + beta * p_old) / inv          # :64-66
                    pn = float(reg.prox_cd(p_old - eta * upd, eta * gamma / inv, degree, j))
                    viol += abs(p_old - pn)                                    # :119-121
                    P[s, j] = pn
                    p_new[q] = pn
                    reg.update_cache_pcd(P, degree, s, j)                      # :135
                self._check(self._lib.spfm_host_step_apply(
                    self._h, b, p_new.ctypes.data_as(_capi._dp), None))
        dv = C.c_double()
        self._check(self._lib.spfm_host_epoch_end(self._h, C.byref(dv)))
        return viol

    def pbcd_epoch_host(self, reg, Pt, lams, loss, order_idx, degree, beta, gamma, eta):
        """``pbcd.pbcd_epoch`` (optimizer/pbcd.py:82-148) for a regularizer object:
        ``compute_cache_pbcd / prox_bcd (in place) / update_cache_pbcd`` run here.  ``Pt``: the
        (n_features, n_components) host array of this order, updated in place."""
        mu = self._MU[loss]
        order, bp = self._host_steps()
        k = Pt.shape[1]
        lams = np.asarray(lams, dtype=np.float64)
        self._check(self._lib.spfm_host_epoch_begin(self._h, int(order_idx), int(degree)))
        reg.compute_cache_pbcd(Pt, degree)                                     # pbcd.py:109
        viol = 0.0
        sums = np.empty((int(np.diff(bp).max()), k + 1))
        for b in range(len(bp) - 1):
            cols = order[bp[b]:bp[b + 1]]
            nc = len(cols)
            if nc == 0:
                continue
            self._check(self._lib.spfm_host_step_sums(self._h, b, sums.ctypes.data_as(_capi._dp)))
            p_new = np.empty((nc, k))
            p_old = np.empty((nc, k))
            for q in range(nc):                                                # :110, in order
                j = int(cols[q])
                p_old[q] = Pt[j]
                inv = mu * sums[q, k] + beta                                   # :68-72
                grad = (sums[q, :k] * lams + beta * Pt[j]) / inv               # :74-77
                pj = Pt[j] - eta * grad                                        # :78
                reg.prox_bcd(pj, eta * gamma / inv, degree, j)                 # :79 (in place)
                viol += float(np.abs(p_old[q] - pj).sum())                     # :146
                Pt[j] = pj
                p_new[q] = pj
                reg.update_cache_pbcd(Pt, degree, j)                           # :145
            self._check(self._lib.spfm_host_step_apply(
                self._h, b, p_new.ctypes.data_as(_capi._dp), p_old.ctypes.data_as(_capi._dp)))
        dv = C.c_double()
        self._check(self._lib.spfm_host_epoch_end(self._h, C.byref(dv)))
        return viol

    def psgd_epoch(self, degree, alpha, beta, gamma, eta0, learning_rate, power_t, batch_size,
                   indices_samples, fit_linear, it, row_lo=None):
        """``psgd.psgd_epoch`` (optimizer/psgd.py:125-199).  Returns (sum_loss, it).
        ``row_lo`` (several ranks, ``spfm_psgd_epoch_sharded``): the handle holds the rows from
        ``row_lo`` on of the problem whose GLOBAL visiting order ``indices_samples`` is; the
        minibatch gradients are all-reduced, ``sum_loss`` is the global sum."""
        idx = np.ascontiguousarray(indices_samples, dtype=np.int32)
        itc = C.c_int64(int(it))
        sl = C.c_double()
        lr = (_capi.LEARNING_RATE[learning_rate] if isinstance(learning_rate, str)
              else int(learning_rate))
        if row_lo is not None:
            self._check(self._lib.spfm_psgd_epoch_sharded(
                self._h, int(degree), float(alpha), float(beta), float(gamma), float(eta0), lr,
                float(power_t), int(batch_size), idx.ctypes.data_as(_capi._ip), idx.size,
                int(row_lo), int(bool(fit_linear)), C.byref(itc), C.byref(sl)))
            return sl.value, itc.value
        self._check(self._lib.spfm_psgd_epoch(
            self._h, int(degree), float(alpha), float(beta), float(gamma), float(eta0), lr,
            float(power_t), int(batch_size), idx.ctypes.data_as(_capi._ip), idx.size,
            int(bool(fit_linear)), C.byref(itc), C.byref(sl)))
        return sl.value, itc.value

    # ------------------------------------------------------------- multi-GPU
    @staticmethod
    def comm_unique_id():
        lib = _capi.load()
        buf = C.create_string_buffer(128)
        rc = lib.spfm_comm_unique_id(buf)
        if rc != 0:
            raise SpfmError("spfm_comm_unique_id: %s" % lib.spfm_last_error(None).decode())
        return buf.raw

    def comm_init(self, uid, n_ranks, rank):
        assert len(uid) == 128
        self._check(self._lib.spfm_comm_init(self._h, uid, int(n_ranks), int(rank)))

    def comm_init_shm(self, name, n_ranks, rank):
        """Host shared-memory communicator for ranks sharing one GPU (test / bring-up)."""
        self._check(self._lib.spfm_comm_init_shm(self._h, name.encode(), int(n_ranks), int(rank)))

    def peer_alloc(self):
        """This rank's exchange slab for the in-kernel cross-GPU exchange: its 64-byte IPC
        handle (ship it to every rank, then call ``peer_connect``)."""
        buf = C.create_string_buffer(64)
        self._check(self._lib.spfm_peer_alloc(self._h, buf))
        return buf.raw

    def peer_connect(self, n_ranks, rank, handles):
        """Map the peers' exchange slabs (``handles``: list of n_ranks 64-byte handles in rank
        order).  The persistent passes then run with several ranks; set the schedule after."""
        blob = b"".join(handles)
        assert len(blob) == 64 * n_ranks
        self._check(self._lib.spfm_peer_connect(self._h, int(n_ranks), int(rank), blob))

    # -------------------------------------------------------- instrumentation
    def profile_enable(self, on=True):
        self._check(self._lib.spfm_profile_enable(self._h, int(bool(on))))

    def profile_reset(self):
        self._check(self._lib.spfm_profile_reset(self._h))

    def profile_get(self, which):
        ms, nl, nz = C.c_double(), C.c_int64(), C.c_int64()
        self._check(self._lib.spfm_profile_get(self._h, int(which), C.byref(ms), C.byref(nl),
                                               C.byref(nz)))
        return ms.value, nl.value, nz.value

    def set_use_graph(self, on):
        self._check(self._lib.spfm_set_use_graph(self._h, int(bool(on))))

    def debug_hop_latency(self, partner=1, rounds=20000):
        """(ns per hand-off, (xcc of workgroup 0, xcc of the partner)); see spfm.h."""
        ns = C.c_double()
        xcc = (C.c_int32 * 2)()
        self._check(self._lib.spfm_debug_hop_latency(self._h, int(partner), int(rounds),
                                                     C.byref(ns), xcc))
        return ns.value, (xcc[0], xcc[1])

    def debug_exchange_cost(self, groups=64, ncols=37, readers_mod=1, rounds=5000):
        """ns per bare exchange round of the persistent pass; see spfm.h."""
        ns = C.c_double()
        self._check(self._lib.spfm_debug_exchange_cost(self._h, int(groups), int(ncols),
                                                       int(readers_mod), int(rounds),
                                                       C.byref(ns)))
        return ns.value

    def debug_prb_stamps(self):
        buf = np.zeros(16 * 512, dtype=np.int64)
        nv = self._lib.spfm_debug_prb_stamps(self._h, buf.ctypes.data_as(_capi._lp), buf.size)
        if nv < 0:
            self._check(nv)
        return buf[:nv].reshape(-1, 16)

    def debug_branch_counts(self, reset=True):
        """How often the device chains took the reference's "numerical error" branches since
        the last reset: dict with keys omegati_clip (omegati.py:97-98), omegacs_dcache
        (omegacs.py:90-96), omegacs_cache (omegacs.py:75-76), squaredl21_resum
        (squaredl21.py:48-49)."""
        buf = (C.c_uint32 * 8)()
        self._check(self._lib.spfm_debug_branch_counts(self._h, buf, int(bool(reset))))
        return dict(omegati_clip=buf[0], omegacs_dcache=buf[1], omegacs_cache=buf[2],
                    squaredl21_resum=buf[3], relax_steps=buf[4], relax_rounds=buf[5])

    def debug_stream_probe(self):
        """One launch that reads the persistent pass's entry stream and nothing else; returns
        the bytes it requested (counter calibration, see spfm.h)."""
        nb = C.c_int64()
        self._check(self._lib.spfm_debug_stream_probe(self._h, C.byref(nb)))
        return nb.value

    def debug_write_probe(self, bytes_per_record):
        """One launch that stores one record of 4, 8 or 16 bytes per matrix entry at the entry's
        row and writes nothing else; returns the bytes it requested (WRITE_SIZE calibration)."""
        nb = C.c_int64()
        self._check(self._lib.spfm_debug_write_probe(self._h, int(bytes_per_record), C.byref(nb)))
        return nb.value

    def get_option(self, key):
        v = C.c_int()
        self._check(self._lib.spfm_get_option(self._h, key.encode(), C.byref(v)))
        return v.value

    def set_option(self, key, value):
        self._check(self._lib.spfm_set_option(self._h, key.encode(), int(value)))
