import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
from conftest import golden_csr, load_golden
from test_hip_parity import _Run
z = load_golden("g3_small_configs.npz"); X = golden_csr(z)
case = "c2|squared"; meta = json.loads(str(z["meta|" + case]))
ref = _Run(X, z["y"], meta, z["P0|" + case], z["lams|" + case], "f64", schedule="colored", options={"persistent": 0})
print("ref", ref.viol, ref.n_batches)
for G in (1, 2, 4, 8, 32, 64):
    for rep in range(2):
        r = _Run(X, z["y"], meta, z["P0|" + case], z["lams|" + case], "f64", schedule="colored", options={"prb_groups": G})
        print("G", G, "rep", rep, np.abs(np.array(r.viol) - np.array(ref.viol)).max(), np.abs(r.P - ref.P).max())
