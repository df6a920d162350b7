import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from conftest import golden_csr, load_golden
from sparsepoly_amd.engine import HipEngine
from sparsepoly_amd.schedule import build_schedule
import scipy.sparse as sp
z = load_golden("g3_small_configs.npz"); X = golden_csr(z)
case = "c2|squared"; meta = json.loads(str(z["meta|" + case]))
n, d = X.shape
def run(opts):
    eng = HipEngine(0, "f64")
    for k, v in opts.items(): eng.set_option(k, v)
    eng.set_data(X, z["y"]); eng.set_params(z["P0|"+case], np.zeros(d), z["lams|"+case])
    eng.configure("pcd", "squared", "squaredl12", 2); eng.init_pred(2, False, False)
    order = eng.set_schedule("colored", np.arange(d, dtype=np.int32))
    v = eng.pcd_epoch(0, 2, meta["beta"], meta["gamma"], 1.0, np.array([0], dtype=np.int32))
    P, w = eng.get_params(); yp = eng.get_y_pred(); eng.close()
    return order, P[0, 0], yp, v
o, Pr, ypr, vr = run({"persistent": 0})
Xc = sp.csc_matrix(X); Xc.sort_indices()
_, bp = build_schedule(Xc, "colored", max_batch=64)
print("nb", len(bp)-1, "batch sizes", np.diff(bp)[:20], "col nnz", np.diff(Xc.indptr)[o][:10])
for G in (1, 32):
    o2, P2, yp2, v2 = run({"prb_groups": G})
    assert np.array_equal(o, o2)
    bad = np.abs(P2[o] - Pr[o]) > 1e-9
    print("G", G, "viol", v2, vr, "first bad pos", np.argmax(bad) if bad.any() else None, "n bad", bad.sum(), "yp err", np.abs(yp2-ypr).max())
    if bad.any():
        pos = int(np.argmax(bad)); b = int(np.searchsorted(bp, pos, side="right") - 1)
        print("   batch", b, "slot", pos - bp[b], "batch size", bp[b+1]-bp[b], "col nnz", np.diff(Xc.indptr)[o[pos]], P2[o[pos]], Pr[o[pos]])
