#!/usr/bin/env python3
"""Config 4 (pbcd, omegacs, k=30) on the persistent pbcd pass: ms per epoch for several
workgroup counts and the in-kernel phase split (diagnostic instantiation).  One JSON line each."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsepoly_amd.engine import HipEngine  # noqa: E402
from sparsepoly_amd.synth import make_problem  # noqa: E402

n = int(os.environ.get("SPFM_BENCH_N", 1_000_000))
d = int(os.environ.get("SPFM_BENCH_D", 100_000))
groups = [int(v) for v in (sys.argv[1:] or ["64", "128", "256"])]
reg = os.environ.get("SPFM_REG", "omegacs")
X, y = make_problem(n, d, 50, 0)
Xc = X.tocsc()
Xc.sort_indices()
k = int(os.environ.get("PB_K", 30))
PHASES = ["p0 shared rows+wait", "p1 late sums+publish", "p2 owner poll", "p2 barrier+step+publish",
          "prefetch issue", "p3 collect poll", "p4 chain+barrier", "p5 scatter", "rotate+end barrier",
          "p3 barrier", "CR: replay set-up + earlier-column terms", "CR: rounds"]
for G in groups:
    for stamps in (0, 1):
        eng = HipEngine(0, "f32")
        eng.set_option("pbprb_groups", G)
        eng.set_option("pbprb_stamps", stamps)
        eng.set_option("pbprb_balance", int(os.environ.get("PB_BALANCE", 1)))
        eng.set_option("pbprb_dbg", int(os.environ.get("PB_DBG", 0)))
        eng.set_data(Xc, y)
        eng.set_params(0.01 * np.random.RandomState(0).randn(1, k, d), np.zeros(d), np.ones(k))
        eng.configure("pbcd", "squared", reg, 2)
        eng.init_pred(2, True, False)
        eng.set_schedule(os.environ.get("PB_SCHED", "colored"), np.arange(d, dtype=np.int32))
        v = [eng.pbcd_epoch(0, 2, 1.0, 1e-3, 1.0)]
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            v.append(eng.pbcd_epoch(0, 2, 1.0, 1e-3, 1.0))
        dt = (time.perf_counter() - t0) / reps
        out = dict(G=G, k=k, stamps=stamps, reg=reg, balance=int(os.environ.get("PB_BALANCE", 1)),
                   ms_per_pbcd_epoch=round(dt * 1e3, 2),
                   steps=eng.n_batches, us_per_step=round(dt * 1e6 / eng.n_batches, 3),
                   active=eng.get_option("pbprb_active"), viol=[round(float(x), 3) for x in v])
        if int(os.environ.get("PB_DBG", 0)) & 8:
            out["dbg"] = [int(x) for x in eng.debug_prb_stamps().ravel()[:16]]
        elif stamps:
            st = eng.debug_prb_stamps()[:, :12].astype(np.float64) / (eng.get_option("relax_steps") or eng.n_batches)
            scale = (dt * 1e9 / (eng.get_option("relax_steps") or eng.n_batches)) / st[0].sum()  # cycles -> ns via the wall time
            out["cycles_per_step_wg0"] = round(float(st[0].sum()))
            out["phase_ns_wg0"] = dict(zip(PHASES, [round(float(x * scale)) for x in st[0]]))
            out["phase_ns_wg1"] = dict(zip(PHASES, [round(float(x * scale)) for x in st[min(1, len(st) - 1)]]))
            out["phase_ns_mean"] = dict(zip(PHASES, [round(float(x * scale)) for x in st.mean(0)]))
            out["phase_ns_max"] = dict(zip(PHASES, [round(float(x * scale)) for x in st.max(0)]))
        print(json.dumps(out), flush=True)
        eng.close()
