#!/usr/bin/env python3
"""Counter calibration on the persistent pass's own access pattern: `prb_stream_probe_kernel`
reads the entry stream of BASELINE config 2 (slot bounds + 4-byte row ids + 4-byte values, the
worker threads' loads of pcd_prb_kernel) and nothing else -- a known byte count.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir> -o run -- python tools/fetch_calibration.py
    python tools/pmc_summary.py <dir>

prints the requested bytes; the ratio FETCH_SIZE / requested is the correction for this pattern
(the guide's x2 holds for 16-byte-per-lane streams only)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsepoly_amd.engine import HipEngine  # noqa: E402
from sparsepoly_amd.synth import make_problem  # noqa: E402

n = int(os.environ.get("SPFM_BENCH_N", 1_000_000))
d = int(os.environ.get("SPFM_BENCH_D", 100_000))
X, y = make_problem(n, d, 50, 0)
eng = HipEngine(0, "f32")
eng.set_data(X, y)
eng.set_params(0.01 * np.random.RandomState(0).randn(1, 30, d), np.zeros(d), np.ones(30))
eng.configure("pcd", "squared", "squaredl12", 2)
eng.init_pred(2, True, False)
eng.set_schedule("colored", np.arange(d, dtype=np.int32))
reps = 5
for _ in range(reps):
    req = eng.debug_stream_probe()
print(json.dumps(dict(kernel="prb_stream_probe_kernel", launches=reps, requested_bytes_per_launch=req,
                      nnz=int(X.nnz), steps=eng.n_batches)))
eng.close()
