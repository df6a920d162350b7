#!/usr/bin/env python3
"""Relaxed runs on BASELINE config 2 / 3 in the reference's column order: merged
steps per sweep, conflict rounds per merged step (device counters), ms per component pass.

    [CONFIG=3] python tools/relax_probe.py
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsepoly_amd.engine import HipEngine  # noqa: E402
from sparsepoly_amd.synth import make_problem  # noqa: E402

CONFIG = int(os.environ.get("CONFIG", "2"))
REG, DEGREE, K, BETA, GAMMA = {2: ("squaredl12", 2, 30, 10.0, 1e-4),
                               3: ("omegati", 3, 16, 10.0, 1e-6)}[CONFIG]
X, y = make_problem(1_000_000, 100_000, 50, 0)
d = X.shape[1]
eng = HipEngine(0, "f32")
eng.set_data(X, y)
eng.set_params(0.01 * np.random.RandomState(0).randn(DEGREE - 1, K, d), np.zeros(d), np.ones(K))
eng.configure("pcd", "squared", REG, DEGREE)
eng.init_pred(DEGREE, True, DEGREE == 3)
eng.set_schedule("exact", np.arange(d, dtype=np.int32))
ic = np.arange(3, dtype=np.int32)
eng.cd_linear_epoch(1.0)
eng.pcd_epoch(0, DEGREE, BETA, GAMMA, 1.0, ic[:1])
eng.debug_branch_counts(reset=True)
t0 = time.perf_counter()
eng.pcd_epoch(0, DEGREE, BETA, GAMMA, 1.0, ic)
ms = 1e3 * (time.perf_counter() - t0) / 3
c = eng.debug_branch_counts(reset=True)
merged = eng.get_option("relax_steps")
print(json.dumps({"config": CONFIG, "strict_steps_per_sweep": eng.n_batches,
                  "merged_steps_per_sweep": merged, "ms_per_component_pass": round(ms, 2),
                  "us_per_merged_step": round(1e3 * ms / max(merged, 1), 2),
                  "steps_with_conflicts_per_pass": c["relax_steps"] / 3,
                  "rounds_per_step_with_conflicts": round(c["relax_rounds"] / max(c["relax_steps"], 1), 2)}))
eng.close()
