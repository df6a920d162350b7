#!/usr/bin/env python3
"""Timing of the other BASELINE configs (3: degree-3 omegati pcd, 4: omegacs pbcd) on the
config-2 matrix; prints one JSON line per config.  Not the driver benchmark (bench.py)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsepoly_amd.engine import HipEngine  # noqa: E402
from sparsepoly_amd.synth import make_problem  # noqa: E402

n = int(os.environ.get("SPFM_BENCH_N", 1_000_000))
d = int(os.environ.get("SPFM_BENCH_D", 100_000))
which = sys.argv[1:] or ["c3", "c4"]
X, y = make_problem(n, d, 50, 0)
Xc = X.tocsc()
Xc.sort_indices()
nnz = Xc.nnz
CFG = {
    # gammas chosen so that P stays substantially non-zero (reported as nonzero_frac_P)
    "c3": dict(solver="pcd", reg="omegati", degree=3, k=16, beta=10.0, gamma=1e-6),
    "c4": dict(solver="pbcd", reg="omegacs", degree=2, k=30, beta=1.0, gamma=1e-3),
    "c2": dict(solver="pcd", reg="squaredl12", degree=2, k=30, beta=10.0, gamma=1e-4),
    # all-subsets model (degree -1): SparseAllSubsets*, pcd_all / pbcd_all
    # (l1 / l21: omegati / omegacs multiply the threshold by prod_j (1 + |p_j|) = e^800 here,
    # which overflows in the reference too)
    "as_pcd": dict(solver="pcd", reg="l1", degree=-1, k=30, beta=10.0, gamma=1e-3),
    "as_pbcd": dict(solver="pbcd", reg="l21", degree=-1, k=30, beta=1.0, gamma=1e-3),
}
for name in which:
    c = CFG[name]
    eng = HipEngine(0, "f32")
    for kv in filter(None, os.environ.get("SPFM_OPTS", "").split(",")):
        key, val = kv.split("=")
        eng.set_option(key, int(val))
    eng.set_data(Xc, y)
    m, k = c["degree"], c["k"]
    allsub = m == -1
    P0 = 0.01 * np.random.RandomState(0).randn(1 if allsub else m - 1, k, d)
    lams = np.where(np.arange(k) % 2 == 0, 1.0, -1.0) if allsub else np.ones(k)
    eng.set_params(P0, np.zeros(d), lams)
    eng.configure(c["solver"], "squared", c["reg"], m)
    eng.init_pred(m, not allsub, m == 3)
    eng.set_schedule("colored", np.arange(d, dtype=np.int32))
    ic = np.arange(k, dtype=np.int32)

    def it():
        v = 0.0 if allsub else eng.cd_linear_epoch(1.0)  # the all-subsets model has no linear term
        for deg in ([m] if allsub else list(range(2, m)) + [m]):
            o = m - deg if deg != m else 0
            if c["solver"] == "pcd":
                v += eng.pcd_epoch(o, deg, c["beta"], c["gamma"], 1.0, ic)
            else:
                v += eng.pbcd_epoch(o, deg, c["beta"], c["gamma"], 1.0)
        return v

    viol = [it()]
    t0 = time.perf_counter()
    steps = int(os.environ.get("STEPS", 2))
    for _ in range(steps):
        viol.append(it())
    dt = (time.perf_counter() - t0) / steps
    if allsub:  # one cache value per (row, component), like degree 2, no linear epoch
        b_alg = (8 * nnz + 4 * n * k + k * nnz * 28) if c["solver"] == "pcd" else \
                (8 * nnz + 4 * n * k + nnz * (20 + 8 * k))
        nsteps = (k if c["solver"] == "pcd" else 1) * eng.n_batches
    elif c["solver"] == "pcd":
        b_alg = sum(8 * nnz + 4 * (deg - 1) * n * k + k * nnz * (20 + 8 * (deg - 1))
                    for deg in range(2, m + 1)) + 20 * nnz
        nsteps = (k * (m - 1) + 1) * eng.n_batches
    else:
        b_alg = 8 * nnz + 4 * (m - 1) * n * k + nnz * (20 + 8 * (m - 1) * k) + 20 * nnz
        nsteps = 2 * eng.n_batches
    print(json.dumps(dict(config=name, **c, ms_per_iteration=round(dt * 1e3, 2),
                          epochs_per_s=round(1 / dt, 4), steps_per_sweep=eng.n_batches,
                          us_per_dependent_step=round(dt * 1e6 / nsteps, 3),
                          alg_GBs=round(b_alg / dt / 1e9, 2),
                          frac_of_8TBs=round(b_alg / dt / 8e12, 5),
                          viol=[round(float(v), 4) for v in viol],
                          nonzero_frac_P=[round(float((p != 0).mean()), 4)
                                          for p in eng.get_params()[0]],
                          loss=round(eng.loss_sum(), 4))), flush=True)
    eng.close()
