import sys, os, json
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsepoly_amd.engine import HipEngine
rng = np.random.RandomState(0)
n, d, k = int(os.environ.get("N", 40)), int(os.environ.get("D", 8)), int(os.environ.get("K", 3))
X = sp.random(n, d, density=0.15, random_state=rng, data_rvs=rng.randn, format="csc")
y = rng.randn(n)
P0 = 0.1 * rng.randn(1, k, d)
for reg in sys.argv[1:] or ["l1", "l21", "omegacs"]:
    res = {}
    for pers in (0, 1):
        eng = HipEngine(0, "f64")
        eng.set_option("pbcd_persistent", pers)
        eng.set_option("pbprb_groups", int(os.environ.get("G", 128)))
        eng.set_data(X, y); eng.set_params(P0, np.zeros(d), np.ones(k))
        eng.configure("pbcd", "squared", reg, 2); eng.init_pred(2, False, False)
        order = eng.set_schedule("colored", np.arange(d, dtype=np.int32))
        v = [eng.pbcd_epoch(0, 2, 1.0, 1e-2, 1.0) for _ in range(2)]
        P, w = eng.get_params()
        res[pers] = (v, P.copy(), eng.get_y_pred(), order, eng.n_batches)
        eng.close()
    print(reg, "order", res[1][3], "nb", res[1][4])
    print("  viol", res[0][0], res[1][0])
    print("  max|dP|", np.abs(res[0][1] - res[1][1]).max(), "max|dy|", np.abs(res[0][2] - res[1][2]).max())
    bad = np.abs(res[0][1] - res[1][1]).max(axis=(0, 1))
    print("  per-column |dP|", np.round(bad, 6))
