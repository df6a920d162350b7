#!/usr/bin/env python3
"""Phase stamps of the degree-3 component pass of BASELINE config 3 (pcd_prb_kernel<float, 3>,
rows in global memory; diagnostic instantiation, options prb_lds=0 + prb_stamps=1):
profiles/r02_c3_step_stamps.txt."""
import os, sys, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsepoly_amd.engine import HipEngine
from sparsepoly_amd.synth import make_problem
n, d = 1000000, 100000
X, y = make_problem(n, d, 50, 0); Xc = X.tocsc(); Xc.sort_indices()
eng = HipEngine(0, "f32"); eng.set_option("prb_lds", 0); eng.set_option("prb_stamps", 1)
eng.set_data(Xc, y); eng.set_params(0.01*np.random.RandomState(0).randn(2,16,d), np.zeros(d), np.ones(16))
eng.configure("pcd", "squared", "omegati", 3); eng.init_pred(3, True, True)
eng.set_schedule("colored", np.arange(d, dtype=np.int32)); nb = eng.n_batches
ic = np.arange(16, dtype=np.int32)
eng.pcd_epoch(0, 3, 10.0, 1e-6, 1.0, ic[:2])
t=time.time(); eng.pcd_epoch(0, 3, 10.0, 1e-6, 1.0, ic[:4]); dt=(time.time()-t)/4
st = eng.debug_prb_stamps().astype(float)
names = ["c:pre", "c:-", "c:-", "c:waitB3", "c:sum+chain", "c:B4", "c:ph3+B5", "c:-",
         "w:gather+sum", "w:publish", "w:sweep", "w:prefetch", "w:B3", "w:waitB4", "w:scatter", "w:B5"]
print("pass %.2f ms, %.2f us/step; cycles/step (WG0 | min | mean | max):" % (dt*1e3, dt*1e6/nb))
for k in range(16):
    print("   %-12s %8.0f %8.0f %8.0f %8.0f" % (names[k], st[0,k]/nb, st[:,k].min()/nb, st[:,k].mean()/nb, st[:,k].max()/nb))
eng.close()
