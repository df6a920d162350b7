import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsepoly_amd.engine import HipEngine
from sparsepoly_amd.synth import make_problem
n, d = int(os.environ.get("N", 1000000)), int(os.environ.get("D", 100000))
X, y = make_problem(n, d, 50, 0); Xc = X.tocsc(); Xc.sort_indices()
for G in [int(g) for g in os.environ.get("GS", "16,32").split(",")]:
    eng = HipEngine(0, "f32"); eng.set_option("prb_groups", G); eng.set_option("prb_stamps", 1)
    eng.set_data(Xc, y); eng.set_params(0.01*np.random.RandomState(0).randn(1,30,d), np.zeros(d), np.ones(30))
    eng.configure("pcd", "squared", "squaredl12", 2); eng.init_pred(2, True, False)
    eng.set_schedule("colored", np.arange(d, dtype=np.int32)); nb = eng.n_batches
    ic = np.arange(30, dtype=np.int32)
    eng.pcd_epoch(0, 2, 10.0, 1.0, 1.0, ic[:2])
    t=time.time(); eng.pcd_epoch(0, 2, 10.0, 1.0, 1.0, ic[:4]); dt=(time.time()-t)/4
    st = eng.debug_prb_stamps().astype(float)
    names = ["c:pre", "h:to-go", "h:sweep", "c:waitB3", "c:sum+chain", "c:B4", "c:ph3+B5", "h:B3",
             "w:gather+sum", "w:publish", "w:sweep", "w:prefetch", "w:B3", "w:waitB4", "w:scatter",
             "w:B5"]
    print("G=%d pass %.2f ms, %.2f us/step; cycles/step (WG0 | min | mean over WGs | max):" % (G, dt*1e3, dt*1e6/nb))
    for k in range(16):
        print("   %-10s %8.0f %8.0f %8.0f %8.0f" % (names[k], st[0,k]/nb, st[:,k].min()/nb, st[:,k].mean()/nb, st[:,k].max()/nb))
    print("   total cycles/step WG0: control %.0f worker %.0f" % (st[0,:8].sum()/nb, st[0,8:].sum()/nb))
    eng.close()
