import sys, os, json
import numpy as np, scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from sparsepoly_amd.engine import HipEngine
z = np.load(os.path.join(ROOT, "tests/golden/g3_small_configs.npz"), allow_pickle=True)
from conftest import golden_csr
X = golden_csr(z)
case = sys.argv[1] if len(sys.argv) > 1 else "c4|squared"
meta = json.loads(str(z["meta|" + case])); y = z["y"]
print(meta, "lams", z["lams|" + case][:8])
d = X.shape[1]; k = meta["k"]
for lin in (0, 1):
    res = {}
    for pers in (0, 1):
        eng = HipEngine(0, "f64")
        eng.set_option("pbcd_persistent", pers)
        eng.set_data(X, y); eng.set_params(z["P0|" + case], np.zeros(d), z["lams|" + case])
        eng.configure("pbcd", meta["loss"], meta["regularizer"], meta["degree"]); eng.init_pred(meta["degree"], False, False)
        order = eng.set_schedule("colored", np.arange(d, dtype=np.int32))
        v = []
        for it in range(3):
            a = eng.cd_linear_epoch(meta["alpha"]) if lin else 0.0
            b = eng.pbcd_epoch(0, meta["degree"], meta["beta"], meta["gamma"], 1.0)
            v.append((round(a, 9), round(b, 9)))
        P, w = eng.get_params()
        res[pers] = (v, P.copy(), eng.get_y_pred(), order, eng.n_batches)
        eng.close()
    print("lin", lin, "nb", res[1][4])
    print("  viol multi", res[0][0]); print("  viol pers ", res[1][0])
    print("  max|dP|", np.abs(res[0][1] - res[1][1]).max(), "max|dy|", np.abs(res[0][2] - res[1][2]).max())
