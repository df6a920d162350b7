"""Regularizer plug-in names (reference: ``sparsepoly/regularizer/__init__.py:8-15``).

In the reference these are Numba jitclass objects whose ``prox_cd`` / ``prox_bcd``
and cache hooks are called from inside the epoch kernels.  Here the six built-ins
are enum-dispatched device functions inside the HIP chain kernels
(``csrc/spfm_kernels.hip.h``); the classes below only carry the registry name
and the solver each one supports (reference ``README.md:28-32``).  A user-defined
Python regularizer cannot run on the device and is rejected with ``ValueError``.
"""


class _Regularizer(object):
    name = None
    solvers = ()

    def __repr__(self):
        return "%s()" % type(self).__name__


class L1(_Regularizer):
    """regularizer/l1.py:12-51 -- pcd and pbcd"""
    name = "l1"
    solvers = ("pcd", "pbcd")


class L21(_Regularizer):
    """regularizer/l21.py:14-48 -- pbcd"""
    name = "l21"
    solvers = ("pbcd",)


class SquaredL12(_Regularizer):
    """regularizer/squaredl12.py:15-78 -- pcd, degree 2 only"""
    name = "squaredl12"
    solvers = ("pcd",)


class SquaredL21(_Regularizer):
    """regularizer/squaredl21.py:18-74 -- pbcd, degree 2 only"""
    name = "squaredl21"
    solvers = ("pbcd",)


class OmegaTI(_Regularizer):
    """regularizer/omegati.py:14-104 -- pcd"""
    name = "omegati"
    solvers = ("pcd",)


class OmegaCS(_Regularizer):
    """regularizer/omegacs.py:17-106 -- pbcd"""
    name = "omegacs"
    solvers = ("pbcd",)


# same key order as the reference registry (it shows in the error message)
REGULARIZATION = {
    "squaredl12": SquaredL12,
    "squaredl21": SquaredL21,
    "l1": L1,
    "l21": L21,
    "omegati": OmegaTI,
    "omegacs": OmegaCS,
}
