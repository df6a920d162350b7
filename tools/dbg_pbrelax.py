import sys, os, numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsepoly_amd.engine import HipEngine
from sparsepoly_amd.synth import make_problem
X, y = make_problem(10000, 2000, 10, seed=2); X = sp.csr_matrix(X)
d = X.shape[1]; k = 5
for precision, DBG, SPIN in (("f32", 0, 0), ("f32", 8, 0), ("f32", 0, 1 << 16), ("f32", 8, 1 << 16), ("f32", 0, 0)):
    for G in (256,):
        eng = HipEngine(0, precision)
        eng.set_option("pbprb_groups", G)
        if DBG: eng.set_option("pbprb_dbg", DBG)
        if SPIN: eng.set_option("debug_spin_max", SPIN)
        print("---- dbg", DBG, "spin", SPIN)
        eng.set_data(X, y)
        P0 = 0.05 * np.random.RandomState(1).randn(1, k, d)
        eng.set_params(P0, np.zeros(d), np.where(np.arange(k) % 2 == 0, 1.0, -1.0))
        eng.configure("pbcd", "squared", "omegacs", 2)
        eng.init_pred(2, True, False)
        eng.set_schedule("exact", np.arange(d, dtype=np.int32))
        v = []
        for e in range(1):
            v.append(eng.pbcd_epoch(0, 2, 1.0, 1e-3, 1.0))
            print(precision, G, "epoch", e, "viol", v[-1], "fallbacks", eng.get_option("persistent_fallbacks"),
                  "active", eng.get_option("pb_relax_active"), "merged", eng.get_option("relax_steps"), flush=True)
            if e == 0:
                print("dbg [.,.,.,first step, site1 owner, site2 xgpu, site3 collectB, site4 collectR | wg+1 per site]:", [int(x) for x in eng.debug_prb_stamps().ravel()[:16]], flush=True)
        eng.close()
