#!/usr/bin/env python3
"""F independent fits of a BASELINE configuration (CONFIG=2|3|4, default 2) on ONE GPU at the same
time (a regularisation path: same matrix and schedule, different gamma): one engine handle, stream
and host thread per fit; the persistent passes of the fits run side by side on disjoint CUs.
Prints one JSON line per F: wall time per iteration (the reference's iteration body) of every fit,
aggregate iterations per second, and whether every fit equals its solo run bitwise.

    [CONFIG=3] [ITERS=3] [NO_CHECK=1] python tools/concurrent_fits.py [F ...]      (default 1 2 3 4)

SHARE=1 (default): the tenants attach to ONE device image of the matrix and share the entry stream
(spfm_share_data); SHARE=0: every tenant uploads and transposes its own copy (round 3).  The line
carries the device memory in use while the F fits run (hipMemGetInfo through ctypes).

Diagnostics: WITH_TORCH=1 (a torch kernel on the null stream first), PRE=n PRE_SCHED=exact
(handles created, run and closed beforehand), GPU_MAX_HW_QUEUES=4 (the runtime's own default).
"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsepoly_amd.engine import HipEngine  # noqa: E402
from sparsepoly_amd.synth import make_problem  # noqa: E402

CONFIG = int(os.environ.get("CONFIG", "2"))      # BASELINE configs[1..3], as in bench.py
SOLVER, REG, DEGREE, K, BETA, G0 = {
    2: ("pcd", "squaredl12", 2, 30, 10.0, 1e-4),
    3: ("pcd", "omegati", 3, 16, 10.0, 1e-6),
    4: ("pbcd", "omegacs", 2, 30, 1.0, 1e-3)}[CONFIG]
ALPHA = 1.0
GAMMAS = [G0 * f for f in (1.0, 2.0, 0.5, 10.0, 3.0, 0.7, 20.0, 4.0)] * 4
# CO=n: co_tenants as given (default min(F, 4): more than four fits take turns on the CUs; CO=F makes
# every fit keep to 1/F of the CUs -- smaller row-block counts, rows in global memory beyond four)
CO = int(os.environ.get("CO", "0"))
ITERS = int(os.environ.get("ITERS", "3"))


SHARE = os.environ.get("SHARE", "1") == "1"


def device_mem_used_gb():
    import ctypes

    hip = ctypes.CDLL("libamdhip64.so")
    free, total = ctypes.c_size_t(), ctypes.c_size_t()
    if hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) != 0:
        return None
    return round((total.value - free.value) / 2 ** 30, 3)


def make_engine(X, y, P0, F, sched="colored", owner=None):
    d = X.shape[1]
    eng = HipEngine(0, "f32")
    if F > 1:
        eng.set_option("co_tenants", CO if CO > 0 else min(F, 4))
    if owner is not None:
        eng.share_data(owner, y)
    else:
        eng.set_data(X, y)
    eng.set_params(P0, np.zeros(d), np.ones(K))
    eng.configure(SOLVER, "squared", REG, DEGREE)
    eng.init_pred(DEGREE, True, DEGREE == 3)
    eng.set_schedule(sched, np.arange(d, dtype=np.int32))
    return eng


def iterate(eng, gamma, iters, out, idx, barrier=None):
    ic = np.arange(K, dtype=np.int32)
    if barrier is not None:
        barrier.wait()
    t0 = time.perf_counter()
    v = []
    for _ in range(iters):
        a = eng.cd_linear_epoch(ALPHA)
        for deg in list(range(2, DEGREE)) + [DEGREE]:   # the reference's iteration body
            o = DEGREE - deg if deg != DEGREE else 0
            if SOLVER == "pcd":
                a += eng.pcd_epoch(o, deg, BETA, gamma, 1.0, ic)
            else:
                a += eng.pbcd_epoch(o, deg, BETA, gamma, 1.0)
        v.append(a)
    out[idx] = (time.perf_counter() - t0, v)


def main():
    Fs = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4]
    if os.environ.get("WITH_TORCH"):   # as in bench.py: torch's HIP runtime loaded first
        import torch

        print("torch sum on its default (null) stream:",
              float(torch.ones(1000, device="cuda").sum().item()))
        torch.cuda.synchronize()
        print("torch", torch.__version__, torch.version.hip, flush=True)
    X, y = make_problem(1_000_000, 100_000, 50, 0)
    d = X.shape[1]
    nnz = X.nnz
    P0 = 0.01 * np.random.RandomState(0).randn(DEGREE - 1, K, d)
    solo = {}
    for _ in range(int(os.environ.get("PRE", "0"))):   # handles created and closed beforehand
        e = make_engine(X, y, P0, 1, os.environ.get("PRE_SCHED", "colored"))
        o = [None]
        if os.environ.get("PRE_RUN", "1") == "1":
            iterate(e, GAMMAS[0], 1, o, 0)
        e.close()
    for F in Fs:
        mem0 = device_mem_used_gb()
        engs = []
        for f in range(F):
            engs.append(make_engine(X, y, P0, F, owner=engs[0] if (SHARE and f > 0) else None))
        out = [None] * F
        for f, e in enumerate(engs):          # warm-up: builds the entry streams
            iterate(e, GAMMAS[f], 1, out, f)
        mem1 = device_mem_used_gb()
        barrier = threading.Barrier(F)
        th = [threading.Thread(target=iterate, args=(engs[f], GAMMAS[f], ITERS, out, f, barrier))
              for f in range(F)]
        t0 = time.perf_counter()
        for t in th:
            t.start()
        for t in th:
            t.join()
        wall = time.perf_counter() - t0
        res = []
        for f, e in enumerate(engs):
            P, w = e.get_params()
            res.append((P, w, e.get_y_pred()))
        fallbacks = [e.get_option("persistent_fallbacks") for e in engs]
        groups = engs[0].get_option("prb_groups" if SOLVER == "pcd" else "pbprb_groups")
        for e in engs:
            e.close()
        same = None
        if F == 1:
            solo[0] = res[0]
        if F > 1 and not os.environ.get("NO_CHECK"):
            if 0 not in solo:
                solo[0] = None
            # every fit against its own solo run (fit 0: the F = 1 run above; the others: now)
            same = []
            for f in range(F):
                if f == 0 and solo.get(0) is not None:
                    ref = solo[0]
                else:
                    e = make_engine(X, y, P0, 1)
                    o = [None]
                    iterate(e, GAMMAS[f], 1 + ITERS, o, 0)
                    Pq, wq = e.get_params()
                    ref = (Pq, wq, e.get_y_pred())
                    e.close()
                same.append(bool(all(np.array_equal(a, b) for a, b in zip(res[f], ref))))
        ms_iter = 1e3 * wall / ITERS
        agg = F * ITERS / wall
        print(json.dumps({
            "config": CONFIG, "fits": F, "iterations": ITERS, "ms_per_iteration_wall": round(ms_iter, 1),
            "ms_per_iteration_per_fit": [round(1e3 * o[0] / ITERS, 1) for o in out],
            "aggregate_epochs_per_s": round(agg, 3),
            "co_tenants": CO if CO > 0 else min(F, 4), "prb_groups": groups,
            "equals_solo_bitwise": same, "persistent_fallbacks": fallbacks, "shared_image": SHARE,
            "device_mem_gb_before": mem0, "device_mem_gb_with_fits": mem1,
            "device_mem_gb_of_the_fits": None if mem0 is None else round(mem1 - mem0, 3)}), flush=True)


if __name__ == "__main__":
    main()
