import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sparsepoly_amd.engine import HipEngine
from sparsepoly_amd.synth import make_problem
n, d, k = int(os.environ.get("N", 200000)), int(os.environ.get("D", 20000)), 30
X, y = make_problem(n, d, 50, 0); Xc = X.tocsc(); Xc.sort_indices()
eng = HipEngine(0, "f32")
eng.set_option("pbprb_groups", int(os.environ.get("G", 64))); eng.set_option("pbprb_dbg", 8)
eng.set_data(Xc, y); eng.set_params(0.01 * np.random.RandomState(0).randn(1, k, d), np.zeros(d), np.ones(k))
eng.configure("pbcd", "squared", "omegacs", 2); eng.init_pred(2, True, False)
eng.set_schedule("colored", np.arange(d, dtype=np.int32))
v = eng.pbcd_epoch(0, 2, 1.0, 1e-3, 1.0)
raw = eng.debug_prb_stamps().ravel()
print("viol", v, "counters", raw[:13])
dump = raw[16:16 + 2048].astype(np.uint32).view(np.float32).reshape(16, 64, 2)
for u in range(16):
    ref, got = dump[u, :, 0], dump[u, :, 1]
    if not ref.any() and not got.any():
        continue
    bad = np.nonzero(ref != got)[0]
    print("u=%2d bad lanes %s" % (u, bad.tolist()))
    print("     ref[0:4] %s ref[32:36] %s" % (ref[:4], ref[32:36]))
    print("     got[0:4] %s got[32:36] %s" % (got[:4], got[32:36]))
