#!/usr/bin/env python3
"""Set-up cost of the pbcd and wide entry streams: host threads vs device (round 4).
    python tools/stream_setup_profile.py [c4] [c5]
c4: BASELINE config 4's matrix (1M x 100k), persistent pbcd pass; c5: configs[4] (10M x 1M), wide
pass.  Prints the first epoch (stream build + epoch) and a steady epoch for stream_device=0 / 1."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sparsepoly_amd.engine import HipEngine  # noqa: E402
from sparsepoly_amd.synth import make_problem  # noqa: E402


def run(name, n, d, solver, reg, k):
    t = time.time()
    X, y = make_problem(n, d, 50, 0)
    X = X.tocsr()
    X.sort_indices()
    print("# %s: generated %dx%d in %.1fs" % (name, n, d, time.time() - t), flush=True)
    for dev in (0, 1, 1):
        eng = HipEngine(0, "f32")
        eng.set_option("stream_device", dev)
        t0 = time.time()
        eng.set_data(X, y)
        t_data = time.time() - t0
        eng.set_params(0.01 * np.random.RandomState(0).randn(1, k, d), np.zeros(d), np.ones(k))
        eng.configure(solver, "squared", reg, 2)
        eng.init_pred(2, True, False)
        t0 = time.time()
        eng.set_schedule("colored", np.arange(d, dtype=np.int32))
        t_sched = time.time() - t0
        ic = np.arange(1, dtype=np.int32)

        def epoch():
            return (eng.pbcd_epoch(0, 2, 1.0, 1e-3, 1.0) if solver == "pbcd"
                    else eng.pcd_epoch(0, 2, 10.0, 1e-4, 1.0, ic))

        t0 = time.time()
        epoch()
        t_first = time.time() - t0
        t0 = time.time()
        epoch()
        t_steady = time.time() - t0
        print(json.dumps(dict(
            case=name, stream_device=dev, set_data_s=round(t_data, 3), set_schedule_s=round(t_sched, 3),
            steps=eng.n_batches, first_epoch_s=round(t_first, 3), steady_epoch_s=round(t_steady, 3),
            stream_build_s=round(t_first - t_steady, 3),
            used=dict(pb=eng.get_option("pb_stream_device_used"),
                      wide=eng.get_option("wide_stream_device_used"),
                      wide_active=eng.get_option("wide_active")))), flush=True)
        eng.close()


cases = sys.argv[1:] or ["c4"]
if "c4" in cases:
    run("c4", 1_000_000, 100_000, "pbcd", "omegacs", 30)
if "c5" in cases:
    run("c5", 10_000_000, 1_000_000, "pcd", "squaredl12", 30)
