#!/usr/bin/env python3
"""Entry-parallel wide pass (pcdwe_kernel) against the thread-per-column form (pcdw_kernel) on a
quarter-size copy of BASELINE configs[4] with the SAME per-workgroup shape (64 workgroups of 39 063
rows, ~710 entries per workgroup and step, classes of up to 512 columns): the two forms run the
same arithmetic in the same order, so parameters and predictions must be bit-identical; and the
incrementally maintained prediction must equal the recomputed one.

    python tools/ep_check.py [n] [d] [groups] [passes]      (default 2500000 250000 64 3)
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_c5  # noqa: E402
from sparsepoly_amd.synth import make_problem  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_500_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 250_000
G = int(sys.argv[3]) if len(sys.argv) > 3 else 64
passes = int(sys.argv[4]) if len(sys.argv) > 4 else 3
X, y = make_problem(n, d, 50, seed=0)
Xc = X.tocsc()
Xc.sort_indices()
del X
P0 = 0.01 * np.random.RandomState(0).randn(1, bench_c5.K, d)
extra = {}
for kv in os.environ.get("WIDE_OPTS", "").split(","):
    if kv:
        extra[kv.split("=")[0]] = int(kv.split("=")[1])
runs = {}
for name, opts in (("ep", {"wide_ep": 1}), ("ep again", {"wide_ep": 1}), ("columns", {"wide_ep": 0})):
    r = bench_c5.run_engine(Xc, y, P0, passes, dict(opts, pcdw_groups=G, wide_min_cols=0, **extra),
                            reps=0)
    runs[name] = r
    bad = np.abs(r["y_incremental"] - r["y_recomputed"]) > 2e-4 * max(1.0, np.abs(r["y_recomputed"]).max())
    print(json.dumps(dict(run=name, info=r["info"], v_lin=r["v_lin"], v=r["v"],
                          rows_off=int(bad.sum()), first_rows_off=np.nonzero(bad)[0][:10].tolist())),
          flush=True)
a, b, c = runs["ep"], runs["ep again"], runs["columns"]
for key in ("P", "w", "y_pred", "y_incremental"):
    print(json.dumps(dict(key=key, ep_equals_ep_again=bool(np.array_equal(a[key], b[key])),
                          ep_equals_columns=bool(np.array_equal(a[key], c[key])),
                          n_diff_vs_columns=int((a[key] != c[key]).sum()))), flush=True)
