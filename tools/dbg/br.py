import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
src = open(os.path.join(ROOT, "tests/test_hip_branches.py")).read().replace("pytestmark = pytest.mark.gpu", "")
ns = {}; exec(compile(src, "t", "exec"), ns)
for reg, pair in (("squaredl21", "neg"), ("squaredl21", "pos"), ("omegacs", "neg")):
    X, y, P0 = ns["_problem"](pair, 3)
    for opts in ({}, {"pbcd_persistent": 0}, {"pbprb_groups": 1}, {"pbprb_groups": 64}):
        v, P, yp, y0, counts = ns["_run_engine"](X, y, P0, "pbcd", reg, 2, opts)
        print(reg, pair, opts, "viol", v, "counts", counts, "P", P[0].ravel()[:6])
print("---- sequence: omegacs deg3 then squaredl21 deg2, persistent")
for rep in range(2):
    X, y, P0 = ns["_problem"]("neg", 3)
    P03 = np.concatenate([P0, np.zeros_like(P0)], axis=0)
    v, P, yp, y0, counts = ns["_run_engine"](X, y, P03, "pbcd", "omegacs", 3, {})
    print("omegacs3", v, counts)
    v, P, yp, y0, counts = ns["_run_engine"](X, y, P0, "pbcd", "squaredl21", 2, {})
    print("sql21", v, counts, P[0].ravel()[:6], yp[:3])
