#!/usr/bin/env python3
"""Config 4 (pbcd, omegacs, k=30) in the reference's own column order (schedule='exact': tiny
strict steps): ms per pbcd epoch for several workgroup counts of the persistent pass."""
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
X, y = make_problem(n, d, 50, 0)
Xc = X.tocsc()
Xc.sort_indices()
k = 30
for G in [int(v) for v in (sys.argv[1:] or ["256", "128", "64", "32"])]:
    eng = HipEngine(0, "f32")
    eng.set_option("pbprb_groups", G)
    eng.set_data(Xc, y)
    eng.set_params(0.01 * np.random.RandomState(0).randn(1, k, d), np.zeros(d), np.ones(k))
    eng.configure("pbcd", "squared", "omegacs", 2)
    eng.init_pred(2, True, False)
    eng.set_schedule("exact", np.arange(d, dtype=np.int32))
    v = [eng.pbcd_epoch(0, 2, 1.0, 1e-3, 1.0)]
    eng.debug_branch_counts(reset=True)
    t0 = time.perf_counter()
    for _ in range(2):
        v.append(eng.pbcd_epoch(0, 2, 1.0, 1e-3, 1.0))
    dt = (time.perf_counter() - t0) / 2
    bc = eng.debug_branch_counts()
    merged = eng.get_option("relax_steps")
    print(json.dumps(dict(G=G, ms_per_pbcd_epoch=round(dt * 1e3, 2), steps=eng.n_batches,
                          merged_steps=merged, relaxed=eng.get_option("pb_relax_active"),
                          us_per_merged_step=round(dt * 1e6 / merged, 3) if merged else None,
                          steps_counted=bc["relax_steps"],
                          rounds_per_such_step=round(bc["relax_rounds"] / max(bc["relax_steps"], 1), 2),
                          us_per_step=round(dt * 1e6 / eng.n_batches, 3),
                          active=eng.get_option("pbprb_active"),
                          viol=[round(float(x), 3) for x in v])), flush=True)
    eng.close()
