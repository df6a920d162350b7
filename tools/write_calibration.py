#!/usr/bin/env python3
"""WRITE_SIZE calibration on the persistent passes' own scatter pattern: `prb_write_probe_kernel`
stores ONE record of 4 / 8 / 16 bytes per entry of BASELINE config 2's matrix at the entry's row
(the row pattern of pcd_prb_kernel with its rows in global memory: 64 row blocks, the coloured
schedule's steps) and writes nothing else -- a known byte count per launch.

    rocprofv3 --pmc WRITE_SIZE --output-format csv -d <dir> -o run -- python3 tools/write_calibration.py
    python tools/pmc_summary.py <dir>

prints the requested bytes per launch; WRITE_SIZE / requested is what the counter charges for
scattered stores of that width."""
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
eng.set_params(0.01 * np.random.RandomState(0).randn(1, 16, d), np.zeros(d), np.ones(16))
eng.configure("pcd", "squared", "omegati", 2)
eng.init_pred(2, True, False)
eng.set_schedule("colored", np.arange(d, dtype=np.int32))
reps = 4
out = []
for width in (4, 8, 16):
    for _ in range(reps):
        req = eng.debug_write_probe(width)
    out.append(dict(kernel="prb_write_probe_kernel<%d>" % width, launches=reps,
                    requested_bytes_per_launch=req))
print(json.dumps(dict(nnz=int(X.nnz), steps=eng.n_batches, probes=out)))
eng.close()
