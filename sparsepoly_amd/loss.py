"""Loss registries (reference: ``sparsepoly/loss.py:74-80``).  The losses are
device functions (``dloss_dev`` / ``loss_dev`` in ``csrc/spfm_kernels.hip.h``);
these tables give the names each estimator accepts and the curvature bound mu
(``loss.py:18,32,59``)."""

MU = {"squared": 1.0, "logistic": 0.25, "squared_hinge": 2.0}

REGRESSION_LOSSES = {"squared": "squared"}

CLASSIFICATION_LOSSES = {
    "squared": "squared",
    "squared_hinge": "squared_hinge",
    "logistic": "logistic",
}
