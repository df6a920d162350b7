#!/usr/bin/env python3
"""Wall time of estimator-level fit() calls on BASELINE config 2 (what a drop-in user sees):
cold fit, warm-started refit, and a four-point regularisation path side by side."""
import json
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsepoly_amd import SparseFactorizationMachineRegressor  # noqa: E402
from sparsepoly_amd.synth import make_problem  # noqa: E402

warnings.simplefilter("ignore")
X, y = make_problem(1_000_000, 100_000, 50, 0)
ITERS = int(os.environ.get("ITERS", "5"))
kw = dict(degree=2, n_components=30, solver="pcd", regularizer="squaredl12", beta=10.0,
          gamma=1e-4, alpha=1.0, tol=0, max_iter=ITERS, random_state=0)
out = {}
for sched in ("colored", "exact"):
    est = SparseFactorizationMachineRegressor(schedule=sched, warm_start=True, **kw)
    t0 = time.perf_counter()
    est.fit(X, y)
    out["fit_%s_cold_s" % sched] = round(time.perf_counter() - t0, 2)
    t0 = time.perf_counter()
    est.fit(X, y)
    out["fit_%s_warm_s" % sched] = round(time.perf_counter() - t0, 2)
    t0 = time.perf_counter()
    est.predict(X)
    out["predict_%s_s" % sched] = round(time.perf_counter() - t0, 2)
    est.release_device()
# a second estimator on the same matrix (no warm start): the colouring is remembered
t0 = time.perf_counter()
SparseFactorizationMachineRegressor(schedule="colored", **dict(kw, gamma=3e-4)).fit(X, y)
out["fit_colored_second_estimator_s"] = round(time.perf_counter() - t0, 2)
base = SparseFactorizationMachineRegressor(schedule="colored", **kw)
t0 = time.perf_counter()
base.fit_path(X, y, gamma=[1e-3, 3e-4, 1e-4, 3e-5])
out["fit_path_4_colored_s"] = round(time.perf_counter() - t0, 2)
out["iterations"] = ITERS
print(json.dumps(out))
