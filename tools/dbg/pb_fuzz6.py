import sys, os
import numpy as np, scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from sparsepoly_amd.engine import HipEngine
def run(n, d, k, degree, reg, prec, G, dens=1.0):
    rng = np.random.RandomState(1)
    X = sp.random(n, d, density=dens, random_state=rng, data_rvs=rng.randn, format="csc")
    y = rng.randn(n)
    P0 = 0.05 * rng.randn(degree - 1, k, d)
    res = []
    for pers in (0, 1):
        eng = HipEngine(0, prec)
        eng.set_option("pbcd_persistent", pers); eng.set_option("pbprb_groups", G)
        eng.set_data(X, y); eng.set_params(P0, np.zeros(d), np.ones(k))
        eng.configure("pbcd", "squared", reg, degree); eng.init_pred(degree, False, degree == 3)
        eng.set_schedule("colored", np.arange(d, dtype=np.int32))
        v = []
        try:
            for deg in list(range(2, degree)) + [degree]:
                o = degree - deg if deg != degree else 0
                v.append(eng.pbcd_epoch(o, deg, 1.0, 1e-3, 1.0))
        except Exception as e:
            v.append("ERR " + str(e)[:60])
        res.append(v)
        eng.close()
    print("n=%d d=%d k=%d deg=%d %s %s G=%d:" % (n, d, k, degree, reg, prec, G), res[0], res[1], flush=True)
for args in [(64, 2, 33, 2, "omegacs", "f64", 128), (64, 2, 33, 3, "omegacs", "f64", 128), (64, 2, 33, 3, "omegacs", "f32", 128),
             (64, 2, 8, 3, "omegacs", "f64", 128), (64, 2, 33, 2, "l1", "f64", 128), (300, 40, 33, 2, "omegacs", "f64", 128, 0.1),
             (300, 40, 8, 3, "omegacs", "f64", 128, 0.1)]:
    run(*args)
