#!/usr/bin/env python3
"""Regularization path with warm_start=True on the config-2 matrix:
wall time of every fit() -- the first uploads, colours and builds the row-block stream,
the following ones reuse the device session and only run their epochs."""
import json
import os
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsepoly_amd import SparseFactorizationMachineRegressor  # noqa: E402
from sparsepoly_amd.synth import make_problem  # noqa: E402

n = int(os.environ.get("SPFM_BENCH_N", 1_000_000))
d = int(os.environ.get("SPFM_BENCH_D", 100_000))
X, y = make_problem(n, d, 50, 0)
est = SparseFactorizationMachineRegressor(degree=2, n_components=30, solver="pcd",
                                          regularizer="squaredl12", alpha=1.0, beta=10.0,
                                          max_iter=3, tol=0, random_state=0, schedule="colored",
                                          warm_start=True)
out = []
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    for reuse in (True, False):
        for g in (1e-3, 3e-4, 1e-4):
            est.set_params(gamma=g)
            t0 = time.perf_counter()
            est.fit(X, y)
            dt = time.perf_counter() - t0
            if not reuse:
                est.release_device()
            out.append(dict(gamma=g, session_reused=reuse and len(out) % 3 != 0,
                            fit_seconds=round(dt, 3), epochs=3,
                            nonzero_frac_P=round(float((est.P_ != 0).mean()), 4)))
            print(json.dumps(out[-1]), flush=True)
        est.release_device()
        for a in ("P_", "w_", "lams_"):
            if hasattr(est, a):
                delattr(est, a)
