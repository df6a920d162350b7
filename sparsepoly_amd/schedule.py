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
