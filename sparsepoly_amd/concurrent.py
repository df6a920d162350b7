"""Independent fits side by side on ONE GPU.

A regularisation path, a cross-validation grid, a bag of restarts: F estimators over the same
(or different) data whose fits do not depend on each other.  One fit's coordinate-descent pass is
a chain of dependent steps on 64 of the 256 CUs; the other CUs idle.  Here every
fit gets its own engine handle, HIP stream and host thread, and the persistent passes of up to
four fits run at the same time on disjoint CUs: each fit is the same kernels on the same inputs
as its solo ``fit`` (bit-identical results), the aggregate throughput is ~3.8x at F = 4
(tools/concurrent_fits.py).

The reference's answer to the same use-case is ``warm_start`` -- one fit after the other
(sparse_factorization_machines.py:380-391 of the reference); this is the MI355X-side
complement, not a replacement: ``warm_start`` chains keep working as before.
"""
import threading

from . import engine as _engine

# HIP spreads a process's streams over four hardware queues, and a persistent pass of the
# default 64 row blocks takes a quarter of the CUs: more fits at once would only queue up
MAX_CONCURRENT = 4
# the persistent pbcd pass is bound by its entry loops, which lengthen as its share of the CUs
# shrinks: two at a time are its best (config 4: 25.4 / 35.3 / 33.1 iterations per second in
# aggregate with 1 / 2 / 4 fits, profiles/r03_concurrent_fits.jsonl)
MAX_CONCURRENT_PBCD = 2


def visible_devices():
    """Ids of the HIP devices of this process (asked of torch without initialising any)."""
    try:
        import torch

        return list(range(int(torch.cuda.device_count())))
    except Exception:
        return [0]


def fit_concurrently(estimators, X, y, max_concurrent=None, devices=None, share_data=True):
    """Fit every estimator of ``estimators`` on ``(X, y)``, up to ``max_concurrent`` at a time
    PER DEVICE (default: four, two when all of them use ``solver='pbcd'``).  ``X`` / ``y`` may be
    one data set for all of them or sequences with one entry per estimator (cross-validation
    folds).

    ``devices``: ``None`` = each estimator's own ``device`` (one GPU); a list of device ids, or
    ``"all"``, fans the fits out over those GPUs -- nothing is exchanged between fits, so the
    aggregate rate grows with the number of devices by construction; an estimator's ``device``
    attribute is ignored then (and restored afterwards).

    One data set for all: the fits of a device share ONE device image of the matrix and its entry
    streams (``share_data``; ``spfm_share_data``), and the first fit to reach the colouring
    computes it for everybody.  Returns the list of fitted estimators (the same objects); the
    first exception raised by a fit is re-raised after all fits have ended."""
    ests = list(estimators)
    if not ests:
        return ests
    if max_concurrent is None:
        all_pbcd = all(getattr(e, "solver", None) == "pbcd" for e in ests)
        max_concurrent = MAX_CONCURRENT_PBCD if all_pbcd else MAX_CONCURRENT
    if int(max_concurrent) < 1:
        raise ValueError("max_concurrent must be >= 1.")
    per_fit_data = isinstance(X, (list, tuple))
    if per_fit_data:
        if not isinstance(y, (list, tuple)) or len(X) != len(ests) or len(y) != len(ests):
            raise ValueError("X and y must hold one entry per estimator.")
    if devices == "all":
        devices = visible_devices()
    if devices is not None:
        devices = [int(dv) for dv in devices]
        if not devices or len(set(devices)) != len(devices) or min(devices) < 0:
            raise ValueError("devices must be a non-empty list of distinct device ids.")
    for e in ests:
        if getattr(e, "distributed", False):
            raise ValueError("concurrent fits are independent fits (distributed=False).")
        if getattr(e, "warm_start", False) and getattr(e, "_device_session", None) is not None:
            # a kept device session was sized for a solo fit (all CUs): start it afresh
            e.release_device()
    slots = [None] if devices is None else devices  # None: the estimator's own device
    per_dev = min(int(max_concurrent), -(-len(ests) // len(slots)))
    todo = list(enumerate(ests))[::-1]
    lock = threading.Lock()
    errors = []
    _engine._capi.load()  # once, before the threads race for it
    _engine._capi.ensure_hw_queues(per_dev)
    # one data set for all: the first fit to reach the colouring computes it, the others install
    # the result (same order, same steps -- what their own colouring would have produced); the
    # same for the device image of the matrix
    root = _engine.Tenancy(per_dev, share_schedules=not per_fit_data,
                           share_data=bool(share_data) and not per_fit_data)

    def work(dev):
        _engine.bind_tenancy(root.on_device(dev))
        try:
            while True:
                with lock:
                    if not todo or errors:
                        return
                    i, est = todo.pop()
                try:
                    est.fit(X[i] if per_fit_data else X, y[i] if per_fit_data else y)
                except BaseException as exc:  # re-raised by the caller's thread
                    with lock:
                        errors.append(exc)
                    return
        finally:
            _engine.bind_tenancy(None)

    try:
        threads = [threading.Thread(target=work, args=(dev,), name="spfm-fit-%s-%d" % (dev, t))
                   for dev in slots for t in range(per_dev)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    finally:
        root.close()
    for e in ests:
        # a device session kept by warm_start was sized for a share of the CUs: the next solo fit
        # starts a full-size one (fitted attributes stay)
        if getattr(e, "_device_session", None) is not None:
            e.release_device()
    if errors:
        raise errors[0]
    return ests


def fit_path(estimator, X, y, max_concurrent=None, devices=None, **grid):
    """Clones of ``estimator`` with the parameter values of ``grid`` (keyword -> sequence, all of
    one length; e.g. ``gamma=[1e-3, 1e-4, 1e-5]``), fitted side by side.  Returns the fitted
    clones in grid order; ``estimator`` itself is not touched."""
    from sklearn.base import clone

    if not grid:
        raise ValueError("fit_path needs at least one parameter sequence (e.g. gamma=[...]).")
    lengths = {len(v) for v in grid.values()}
    if len(lengths) != 1:
        raise ValueError("all parameter sequences must have the same length.")
    valid = estimator.get_params()
    for name in grid:
        if name not in valid:
            raise ValueError("%r is not a parameter of %s." % (name, type(estimator).__name__))
    ests = []
    for i in range(lengths.pop()):
        e = clone(estimator)
        e.set_params(**{name: values[i] for name, values in grid.items()})
        ests.append(e)
    return fit_concurrently(ests, X, y, max_concurrent=max_concurrent, devices=devices)
