"""Conflict-free coordinate batches, host side (``csrc/spfm_schedule.cpp``).

Two columns that share no row touch disjoint entries of the ANOVA caches and of
``y_pred``; their gradient reductions and scatter updates commute.  A *batch* is
a set of pairwise row-disjoint columns; the device runs a batch as one dependent
step, and a sweep over the batches equals the reference's sequential sweep
(optimizer/pcd.py:97, pbcd.py:110, cd_linear.py:10) over the concatenated order.
"""
import ctypes as C

import numpy as np

from . import _capi


def build_schedule(Xcsc, mode="colored", indices_feature=None, max_batch=0):
    """Returns (order int32[d], batch_ptr int32[n_batches+1]).  Pure host code."""
    lib = _capi.load()
    n, d = Xcsc.shape
    if indices_feature is None:
        indices_feature = np.arange(d, dtype=np.int32)
    ipa, ipp = _capi.i64(Xcsc.indptr)
    ixa, ixp = _capi.i32(Xcsc.indices)
    jfa, jfp = _capi.i32(indices_feature)
    order = np.empty(d, dtype=np.int32)
    bp = np.empty(d + 1, dtype=np.int32)
    nb = C.c_int32()
    rc = lib.spfm_schedule_build(_capi.SCHEDULES[mode], n, d, ipp, ixp, jfp, int(max_batch),
                                 order.ctypes.data_as(_capi._ip), bp.ctypes.data_as(_capi._ip),
                                 C.byref(nb))
    if rc != 0:
        raise ValueError("spfm_schedule_build failed (%d): bad arguments" % rc)
    return order, bp[: nb.value + 1].copy()


class Schedule(object):
    """A coordinate schedule as a reusable product: the visiting
    ``order`` of the features and the boundaries ``batch_ptr`` of its row-disjoint
    batches.  It depends only on the sparsity structure of X, so it can be built once per
    data set, stored next to it (``save`` / ``load``) and handed to any number of fits:

        sched = Schedule.build(X, "colored")
        est = SparseFactorizationMachineRegressor(schedule=sched).fit(X, y)

    The engine re-validates it against the data it is installed on (permutation,
    row-disjoint batches), so a stale or foreign schedule is rejected, never raced on."""

    def __init__(self, order, batch_ptr, mode="colored", shape=None, nnz=None):
        self.order = np.ascontiguousarray(order, dtype=np.int32)
        self.batch_ptr = np.ascontiguousarray(batch_ptr, dtype=np.int32)
        self.mode = mode
        self.shape = None if shape is None else tuple(int(v) for v in shape)
        self.nnz = None if nnz is None else int(nnz)

    @property
    def n_batches(self):
        return len(self.batch_ptr) - 1

    @classmethod
    def build(cls, X, mode="colored", indices_feature=None, max_batch=64):
        """max_batch=64 matches the persistent engine's step width (64 column slots)."""
        from .engine import canonical_csc

        Xc = canonical_csc(X)
        order, bp = build_schedule(Xc, mode, indices_feature, max_batch)
        return cls(order, bp, mode, Xc.shape, Xc.nnz)

    def save(self, path):
        np.savez_compressed(path, order=self.order, batch_ptr=self.batch_ptr,
                            mode=np.array(self.mode), shape=np.array(self.shape or (-1, -1)),
                            nnz=np.array(-1 if self.nnz is None else self.nnz))

    @classmethod
    def load(cls, path):
        z = np.load(path, allow_pickle=False)
        shape = tuple(int(v) for v in z["shape"])
        nnz = int(z["nnz"])
        return cls(z["order"], z["batch_ptr"], str(z["mode"]),
                   None if shape[0] < 0 else shape, None if nnz < 0 else nnz)

    def __repr__(self):
        return "Schedule(mode=%r, n_features=%d, n_batches=%d)" % (
            self.mode, len(self.order), self.n_batches)
