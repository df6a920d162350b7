#!/usr/bin/env python3
"""Effect of the long-slot path on Zipf-skewed data (feature popularity ~ 1/(rank+5)):
same pcd iteration with the cooperative path (default) and with it disabled."""
import json, os, sys, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsepoly_amd.engine import HipEngine

rng = np.random.RandomState(0)
n, d, r = 400_000, 40_000, 50
pop = 1.0 / (np.arange(d) + 5.0)
pop /= pop.sum()
cols = rng.choice(d, size=n * r, p=pop).astype(np.int64)
rows = np.repeat(np.arange(n, dtype=np.int64), r)
X = sp.csr_matrix((rng.randn(n * r).astype(np.float32).astype(np.float64), (rows, cols)), shape=(n, d))
X.sum_duplicates()
Xc = X.tocsc(); Xc.sort_indices()
y = rng.randn(n)
cl = np.diff(Xc.indptr)
print("nnz %d, longest columns %s, median %d" % (Xc.nnz, np.sort(cl)[-5:], np.median(cl)), flush=True)
k = 8
for thresh in (48, 1 << 30):
    eng = HipEngine(0, "f32")
    eng.set_option("prb_long", thresh)
    eng.set_data(Xc, y)
    eng.set_params(0.01 * np.random.RandomState(0).randn(1, k, d), np.zeros(d), np.ones(k))
    eng.configure("pcd", "squared", "squaredl12", 2)
    eng.init_pred(2, True, False)
    eng.set_schedule("colored", np.arange(d, dtype=np.int32))
    ic = np.arange(k, dtype=np.int32)
    v0 = eng.cd_linear_epoch(1.0) + eng.pcd_epoch(0, 2, 10.0, 1e-4, 1.0, ic)
    t0 = time.perf_counter()
    v1 = eng.cd_linear_epoch(1.0) + eng.pcd_epoch(0, 2, 10.0, 1e-4, 1.0, ic)
    dt = time.perf_counter() - t0
    print(json.dumps(dict(prb_long=thresh, steps_per_sweep=eng.n_batches, ms_per_iteration=round(dt * 1e3, 2),
                          us_per_dependent_step=round(dt * 1e6 / ((k + 1) * eng.n_batches), 3), viol=[v0, v1])), flush=True)
    eng.close()
