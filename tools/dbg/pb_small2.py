import sys, os, json
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsepoly_amd.engine import HipEngine
def run(n, d, k, reg, gamma, lams_pm, G=128, epochs=1, dens=0.15, scale=0.1, gold=False):
    rng = np.random.RandomState(0)
    X = sp.random(n, d, density=dens, random_state=rng, data_rvs=rng.randn, format="csc")
    y = rng.randn(n)
    P0 = scale * rng.randn(1, k, d)
    lams = np.where(rng.rand(k) > 0.5, 1.0, -1.0) if lams_pm else np.ones(k)
    if gold:
        ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from conftest import golden_csr
        z = np.load(os.path.join(ROOT, "tests/golden/g3_small_configs.npz"), allow_pickle=True)
        X = golden_csr(z).tocsc(); y = z["y"]; P0 = z["P0|c4|squared"]; lams = z["lams|c4|squared"]
        n, d = X.shape; k = 30
    res = {}
    for pers in (0, 1):
        eng = HipEngine(0, "f64")
        eng.set_option("pbcd_persistent", pers)
        eng.set_option("pbprb_groups", G)
        eng.set_data(X, y); eng.set_params(P0, np.zeros(d), lams)
        eng.configure("pbcd", "squared", reg, 2); eng.init_pred(2, False, False)
        order = eng.set_schedule("colored", np.arange(d, dtype=np.int32))
        sched = eng.get_schedule()
        v = [eng.pbcd_epoch(0, 2, 1.0, gamma, 1.0) for _ in range(epochs)]
        P, w = eng.get_params()
        res[pers] = (v, P.copy(), eng.get_y_pred(), order, sched.batch_ptr)
        eng.close()
    dP = np.abs(res[0][1] - res[1][1]).max(axis=(0, 1))
    order, bp = res[1][3], res[1][4]
    pos = np.empty(d, int); pos[order] = np.arange(d)
    step = np.searchsorted(bp, pos, side="right") - 1
    bad = np.nonzero(dP > 1e-9)[0]
    print("n=%d d=%d k=%d %s gamma=%g pm=%d G=%d: viol %s vs %s  max|dP|=%.3g  bad cols %d/%d"
          % (n, d, k, reg, gamma, lams_pm, G, np.round(res[0][0], 6), np.round(res[1][0], 6), dP.max(), len(bad), d))
    if len(bad):
        first = bad[np.argmin(pos[bad])]
        print("   first bad col j=%d pos=%d step=%d slot=%d (step size %d); zero in multi: %s pers: %s"
              % (first, pos[first], step[first], pos[first] - bp[step[first]], bp[step[first] + 1] - bp[step[first]],
                 (res[0][1][0, :, first] == 0).all(), (res[1][1][0, :, first] == 0).all()))
        zc = [(int(j), int(step[j])) for j in range(d) if (res[0][1][0, :, j] == 0).all()]
        print("   zero cols in multi (j, step):", zc[:12])
run(300, 60, 30, "omegacs", 0.1, 1, scale=0.01)
run(300, 60, 30, "omegacs", 0.1, 1, dens=0.1, scale=0.01)
run(300, 60, 30, "l21", 0.1, 1, dens=0.1, scale=0.01)
run(300, 60, 30, "l1", 0.1, 1, dens=0.1, scale=0.01)
run(300, 60, 30, "omegacs", 0.1, 1, gold=True)
run(300, 60, 30, "l21", 0.1, 1, gold=True)
run(300, 60, 30, "l1", 0.001, 1, gold=True)
run(300, 60, 30, "omegacs", 0.1, 1, gold=True, G=1)
run(300, 60, 30, "omegacs", 0.1, 1, gold=True, G=64)
