#!/usr/bin/env python3
"""Where a step of the WIDE pcd pass (pcdw_kernel) spends its time: in-kernel
cycle stamps of thread 0 of every workgroup (diagnostic instantiation, option pcdw_stamps).

    python tools/pcdw_stamp_probe.py [n] [d] [groups]     (default 2000000 200000 256)
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsepoly_amd.engine import HipEngine
from sparsepoly_amd.synth import make_problem

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
G = int(sys.argv[3]) if len(sys.argv) > 3 else 256
X, y = make_problem(n, d, 50, 0)
Xc = X.tocsc(); Xc.sort_indices()
for stamps in (0, 1):
    eng = HipEngine(0, "f32")
    eng.set_option("pcdw_groups", G); eng.set_option("pcdw_stamps", stamps)
    eng.set_option("wide_min_cols", 0)  # the wide pass whatever the class width
    for kv in os.environ.get("WIDE_OPTS", "").split(","):   # e.g. WIDE_OPTS=wide_lds_rows=0,wide_ep=0
        if kv:
            eng.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    eng.set_data(Xc, y)
    eng.set_params(0.01 * np.random.RandomState(0).randn(1, 30, d), np.zeros(d), np.ones(30))
    eng.configure("pcd", "squared", "squaredl12", 2); eng.init_pred(2, True, False)
    eng.set_schedule("colored", np.arange(d, dtype=np.int32)); nb = eng.n_batches
    ic = np.arange(30, dtype=np.int32)
    eng.pcd_epoch(0, 2, 10.0, 1e-4, 1.0, ic[:2])
    t = time.time(); eng.pcd_epoch(0, 2, 10.0, 1e-4, 1.0, ic[:4]); dt = (time.time() - t) / 4
    print("n=%d d=%d G=%d stamps=%d: wide=%d lds_rows=%d  %d steps, pass %.2f ms, %.2f us/step"
          % (n, d, G, stamps, eng.get_option("wide_active"), eng.get_option("wide_lds_active"), nb,
             dt * 1e3, dt * 1e6 / nb), flush=True)
    if stamps:
        st = eng.debug_prb_stamps().astype(float)
        names = ["hazard rows", "sums+publish", "owner poll", "owner total+publish",
                 "prefetch issue", "collect poll", "barrier", "chain rounds", "scatter",
                 "rotate+end barrier", "(pf: row gathers)", "(pf: slot data, bounds)",
                 "(pf: entry stream)"]
        if st[:, 10:13].sum() == 0:
            names = names[:10]
        else:
            names[4] = "(pf: slot table)"
        print("   cycles/step:           WG0      min     mean      max")
        for k in range(len(names)):
            print("   %-20s %8.0f %8.0f %8.0f %8.0f" % (names[k], st[0, k] / nb, st[:, k].min() / nb,
                                                      st[:, k].mean() / nb, st[:, k].max() / nb))
        print("   total WG0 %.0f" % (st[0, :13].sum() / nb))
    eng.close()
