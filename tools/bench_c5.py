#!/usr/bin/env python3
"""BASELINE configs[4] on ONE GPU: degree=2, n_components=30, pcd on the 10M x 1M synthetic CSR
(~50 nnz/row, nnz ~ 5e8).  The configuration is specified for 8 GPUs (1.25M rows each); the whole
matrix fits one MI355X (CSC + CSR images 8 GB, caches 1.2 GB), so the single-GPU run exercises
the data path at its full size: colouring (steps per sweep), `passes` component passes + the
cd_linear epoch, incremental-vs-recomputed prediction, and -- with --oracle -- the CPU oracle on
the same passes in the reported order.  One JSON line per engine.

    python tools/bench_c5.py [--n 10000000] [--d 1000000] [--passes 2] [--oracle]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparsepoly_amd.engine import HipEngine  # noqa: E402
from sparsepoly_amd.synth import make_problem  # noqa: E402

K, BETA, GAMMA, ALPHA = 30, 10.0, 1e-4, 1.0


def run_engine(Xc, y, P0, passes, options, reps=1):
    n, d = Xc.shape
    eng = HipEngine(0, "f32")
    for key, val in options.items():
        eng.set_option(key, val)
    t0 = time.time()
    eng.set_data(Xc, y)
    t_data = time.time() - t0
    eng.set_params(P0, np.zeros(d), np.ones(K))
    eng.configure("pcd", "squared", "squaredl12", 2)
    eng.init_pred(2, True, False)
    y0 = eng.get_y_pred()
    t0 = time.time()
    order = eng.set_schedule("colored", np.arange(d, dtype=np.int32))
    t_sched = time.time() - t0
    ic = np.arange(passes, dtype=np.int32)
    t0 = time.time()
    v_lin = eng.cd_linear_epoch(ALPHA)   # first calls build the entry stream (persistent engine)
    v = eng.pcd_epoch(0, 2, BETA, GAMMA, 1.0, ic)
    t_first = time.time() - t0
    P, w = eng.get_params()
    yp = eng.get_y_pred()
    t_lin = t_pass = None
    if reps:
        t0 = time.perf_counter()
        eng.cd_linear_epoch(ALPHA)
        t_lin = time.perf_counter() - t0
        t0 = time.perf_counter()
        eng.pcd_epoch(0, 2, BETA, GAMMA, 1.0, ic)
        t_pass = (time.perf_counter() - t0) / passes
    y_inc = eng.get_y_pred()       # incrementally maintained through all the epochs above
    eng.init_pred(2, True, False)  # recomputed from the trained parameters
    y_new = eng.get_y_pred()
    info = dict(steps_per_sweep=eng.n_batches, persistent=eng.get_option("persistent_active"),
                wide_active=eng.get_option("wide_active"),
                wide_rows_in_lds=eng.get_option("wide_lds_active"),
                set_data_s=round(t_data, 1), schedule_s=round(t_sched, 1),
                first_calls_s=round(t_first, 1))
    if reps:
        info.update(ms_per_cd_linear_epoch=round(1e3 * t_lin, 1),
                    ms_per_component_pass=round(1e3 * t_pass, 1),
                    us_per_dependent_step=round(1e6 * t_pass / eng.n_batches, 2),
                    est_ms_per_iteration=round(1e3 * (t_lin + K * t_pass), 0))
    eng.close()
    return dict(v_lin=v_lin, v=v, P=P, w=w, y_pred=yp, y_incremental=y_inc, y_recomputed=y_new,
                y0=y0, order=order, info=info)


def run_oracle(Xc, y, P0, y0, order, passes):
    from oracle import oracle as orc

    orc.build()
    n, d = Xc.shape
    ds = orc.CSC(Xc)
    reg = orc.Regularizer("squaredl12")
    reg.init_cache_pcd(2, d, K)
    w = np.zeros(d)
    yp = np.ascontiguousarray(y0.copy())
    cn = np.asarray(Xc.multiply(Xc).sum(axis=0)).ravel()
    jf = np.ascontiguousarray(order)
    t0 = time.time()
    v_lin = orc.cd_linear_epoch(w, ds, y, yp, cn, ALPHA, "squared", jf)
    P = np.ascontiguousarray(P0[0].copy())
    A = np.zeros((n, 3))
    v = orc.pcd_epoch(P, ds, y, yp, np.ones(K), 2, BETA, GAMMA, 1.0, reg, "squared", A,
                      np.arange(passes, dtype=np.int32), jf)
    return dict(v_lin=v_lin, v=v, P=P, w=w, y_pred=yp, seconds=time.time() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--d", type=int, default=1_000_000)
    ap.add_argument("--passes", type=int, default=2)
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--engines", default="multi_kernel,persistent")
    args = ap.parse_args()
    t0 = time.time()
    X, y = make_problem(args.n, args.d, 50, seed=0)
    Xc = X.tocsc()
    Xc.sort_indices()
    del X
    print("[c5] %dx%d nnz=%d generated in %.0fs" % (Xc.shape + (Xc.nnz, time.time() - t0)),
          file=sys.stderr, flush=True)
    P0 = 0.01 * np.random.RandomState(0).randn(1, K, args.d)
    opts = {"multi_kernel": {"persistent": 0}, "persistent": {}}
    ref = None
    for name in args.engines.split(","):
        r = run_engine(Xc, y, P0, args.passes, opts[name])
        out = dict(config="BASELINE configs[4] on one GPU", engine=name, n=args.n, d=args.d,
                   nnz=int(Xc.nnz), passes=args.passes, **r["info"])
        out["incremental_vs_recomputed_max_abs"] = float(np.abs(r["y_incremental"] - r["y_recomputed"]).max())
        out["viol"] = [float(r["v_lin"]), float(r["v"])]
        if args.oracle and ref is None:
            ref = run_oracle(Xc, y, P0, r["y0"], r["order"], args.passes)
            out["oracle_seconds"] = round(ref["seconds"], 1)
        if ref is not None and np.array_equal(r["order"], ref.get("order", r["order"])):
            ref.setdefault("order", r["order"])
            out["vs_oracle"] = dict(
                viol_rel=[abs(r["v_lin"] - ref["v_lin"]) / abs(ref["v_lin"]),
                          abs(r["v"] - ref["v"]) / abs(ref["v"])],
                P_max_abs=float(np.abs(r["P"][0] - ref["P"]).max()),
                w_max_abs=float(np.abs(r["w"] - ref["w"]).max()),
                y_pred_max_abs=float(np.abs(r["y_pred"] - ref["y_pred"]).max()))
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
