#!/usr/bin/env python3
"""Where an estimator-level fit() spends its wall time at BASELINE config 2 (setup vs epochs)."""
import os, sys, time, warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsepoly_amd.engine import HipEngine, canonical_csc
from sparsepoly_amd.synth import make_problem
t = time.time(); X, y = make_problem(1_000_000, 100_000, 50, 0); print("generate %.2fs" % (time.time() - t))
def tick(label, t0): print("  %-28s %.2fs" % (label, time.time() - t0), flush=True)
X = X.tocsr(); X.sort_indices(); X.sum_duplicates()
for dev in (0, 1, 1):   # host-thread transposition, then the device one (cold, then warm)
    eng = HipEngine(0, "f32"); eng.set_option("ingest_device", dev)
    t0 = time.time(); eng.set_data(X, y); tick("set_data (CSR in, ingest_device=%d)" % dev, t0)
    assert eng.get_option("ingest_device_used") == dev
    engs = globals().setdefault("engs", []); engs.append(eng)
for e in engs[:-1]: e.close()
d = X.shape[1]
t0 = time.time(); eng.set_params(0.01*np.random.RandomState(0).randn(1,30,d), np.zeros(d), np.ones(30)); eng.configure("pcd","squared","squaredl12",2); eng.init_pred(2, True, False); tick("params+configure+init_pred", t0)
for dev in (0, 1, 1):   # the colouring by host threads, then on the device (cold, warm)
    eng.set_option("colour_device", dev)
    t0 = time.time(); eng.set_schedule("colored", np.arange(d, dtype=np.int32)); tick("set_schedule (colouring, colour_device=%d)" % dev, t0)
    assert eng.get_option("colour_device_used") == dev
    print("    steps per sweep:", eng.n_batches)
ic = np.arange(30, dtype=np.int32)
t0 = time.time(); eng.cd_linear_epoch(1.0); tick("first cd_linear (stream build)", t0)
t0 = time.time(); eng.pcd_epoch(0, 2, 10.0, 1e-4, 1.0, ic); tick("first pcd epoch", t0)
t0 = time.time(); eng.cd_linear_epoch(1.0); eng.pcd_epoch(0, 2, 10.0, 1e-4, 1.0, ic); tick("steady-state iteration", t0)
