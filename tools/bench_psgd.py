#!/usr/bin/env python3
"""Timing of the psgd solver on the config-2 matrix (1M x 100k CSR,
50 nnz/row, degree 2, k = 30, batch_size 'auto' = 1/density = 2000 rows); prints one JSON
line per regularizer with the per-kernel split from the engine's HIP-event profiler.
Not the driver benchmark (bench.py).

Algorithmic HBM bytes per epoch (f32 storage of X, f64 parameters):
  gradient pass   nnz * (4 idx + 4 val) + nnz * k * (8 read P + 8 atomic add)   [P/grad rows
                  are L2/LLC-resident; counted once per touch]
  update pass     n_batches * d * k * (8+8 read, 8+8 write)   P and grad_P
  Michelot sweeps n_batches * sweeps * d * k * 8 (squaredl12) or * d * 8 (squaredl21)
"""
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
k = int(os.environ.get("SPFM_BENCH_K", 30))
regs = sys.argv[1:] or ["l1", "l21", "squaredl12", "squaredl21"]
X, y = make_problem(n, d, 50, 0)
Xc = X.tocsc()
Xc.sort_indices()
nnz = Xc.nnz
batch = int(os.environ.get("SPFM_BENCH_BATCH", int(n * d / nnz)))
nb = -(-n // batch)
GAMMA = {"l1": 1e-3, "l21": 1e-3, "squaredl12": 1e-6, "squaredl21": 1e-6}
cpu = None
if os.environ.get("SPFM_CPU", "0") == "1":
    # bounded CPU baseline: the oracle (single thread, f64) on the first `nbc` minibatches
    # of the same matrix with the full-width P, scaled to one epoch
    from oracle import oracle as orc

    orc.build()
    nbc = int(os.environ.get("SPFM_CPU_BATCHES", 8))
    Xs = X.tocsr()[: nbc * batch]
    Xo = orc.CSR(Xs)
    Po = np.ascontiguousarray((0.01 * np.random.RandomState(0).randn(1, k, d)).swapaxes(1, 2))
    wo = np.zeros(d)
    cpu = {}
    for reg in regs:
        t0 = time.perf_counter()
        orc.psgd_epoch(Po.copy(), wo.copy(), Xo, y[: nbc * batch], np.ones(k), 2, 1e-3, 1e-3,
                       GAMMA[reg], reg, "squared", np.arange(nbc * batch, dtype=np.int32), True,
                       0.01, "optimal", 1.0, batch, 1)
        cpu[reg] = (time.perf_counter() - t0) * nb / nbc
for reg in regs:
    eng = HipEngine(0, "f32")
    eng.set_data(Xc, y)
    P0 = 0.01 * np.random.RandomState(0).randn(1, k, d)
    eng.set_params(P0, np.zeros(d), np.ones(k))
    eng.configure("psgd", "squared", reg, 2)
    idx = np.arange(n, dtype=np.int32)
    args = (2, 1e-3, 1e-3, GAMMA[reg], 0.01, "optimal", 1.0, batch, idx, True)
    sl, it = eng.psgd_epoch(*args, 1)          # warm-up epoch
    losses = [sl / n]
    steps = int(os.environ.get("STEPS", 2))
    # timed epochs: profiler off (l1 / l21 then replay runs of minibatches from a hipGraph)
    t0 = time.perf_counter()
    for _ in range(steps):
        sl, it = eng.psgd_epoch(*args, it)
        losses.append(sl / n)
    dt = (time.perf_counter() - t0) / steps
    # one more epoch under the engine's HIP-event profiler (eager launches) for the split
    split = {}
    if os.environ.get("SPFM_PROF", "1") == "1":
        eng.profile_enable(True)
        eng.profile_reset()
        sl, it = eng.psgd_epoch(*args, it)
        for which, name in ((0, "grad"), (1, "update"), (2, "prox_support")):
            ms, launches, _ = eng.profile_get(which)
            if launches:
                split[name] = dict(ms_per_epoch=round(ms, 3),
                                   us_per_batch=round(ms * 1e3 / launches, 2))
        eng.profile_enable(False)
    b_grad = nnz * 8 + nnz * k * 16
    b_upd = nb * d * k * 32
    out = dict(solver="psgd", reg=reg, k=k, batch_size=batch, batches_per_epoch=nb,
               ms_per_epoch=round(dt * 1e3, 2), epochs_per_s=round(1 / dt, 3),
               rows_per_s=round(n / dt), split=split,
               update_pass_GBs=(round(d * k * 32 / (split["update"]["us_per_batch"] * 1e-6) / 1e9, 1)
                                if "update" in split else None),
               alg_GB_per_epoch=round((b_grad + b_upd) / 1e9, 2),
               mean_loss=[round(float(v), 6) for v in losses],
               nonzero_frac_P=round(float((eng.get_params()[0] != 0).mean()), 4))
    if cpu:
        out["cpu_oracle_s_per_epoch"] = round(cpu[reg], 2)
        out["cpu_sample"] = "%d of %d minibatches, 1 thread, scaled" % (nbc, nb)
        out["gpu_over_cpu"] = round(cpu[reg] / dt, 1)
    print(json.dumps(out), flush=True)
    eng.close()
