#!/usr/bin/env python3
"""Headline benchmark: PCD epochs/sec on BASELINE config 2 (degree=2, n_components=30,
regularizer='squaredl12', solver='pcd', 1M x 100k synthetic CSR, ~50 nnz/row).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full training iteration of the reference's _fit_pcd loop
(sparse_factorization_machines.py:196-256): one cd_linear epoch + one pcd epoch
over all k components, on data already resident in HBM.  The timed region calls
exactly what ``SparseFactorizationMachineRegressor.fit`` calls per iteration
(``HipEngine.cd_linear_epoch`` + ``HipEngine.pcd_epoch`` through the C ABI).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline      dominant kernel (pcd_grad): algorithmic bytes per launch / average
                launch duration (HIP events on the engine's stream) vs 8 TB/s HBM
  cpu_baseline  the CPU oracle (float64, 1 thread) on a bounded sample of the same
                workload, timed on this host
N > 1: rows are sharded over the ranks (strong scaling on the same matrix); the
column partial sums of every step are all-reduced with RCCL.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# workload = BASELINE.json configs[1]
N_SAMPLES = int(os.environ.get("SPFM_BENCH_N", 1_000_000))
N_FEATURES = int(os.environ.get("SPFM_BENCH_D", 100_000))
NNZ_PER_ROW = 50
K = 30
DEGREE = 2
REG = "squaredl12"
# hyper-parameters: well conditioned and such that P does NOT collapse
# to zero (squaredl12's threshold scales with sum_j |P[s,j]| ~ 800 here: gamma = 1 would zero
# every coordinate in the first epoch, after which the scatter half of every step is a
# no-op).  The fraction of non-zero P entries after the run is reported.
ALPHA, BETA, GAMMA, ETA0 = 1.0, 10.0, 1e-4, 1.0
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec


def log(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="f32", choices=["f32", "f64"])
    ap.add_argument("--schedule", default="colored", choices=["colored", "exact"])
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsal on a one-GPU box: SPFM_DEVICE=0 SPFM_COMM=shm runs all ranks on device 0 with
    # the engine's host shared-memory exchange instead of RCCL (see spfm_comm_init_shm)
    local_rank = int(os.environ.get("SPFM_DEVICE", local_rank))
    import torch

    dist = None
    if world > 1:
        import torch.distributed as dist

        # torch.distributed only carries control messages (unique id, barrier, max of the
        # timings): gloo.  The data path's collectives are the engine's own RCCL calls.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    if args.gpus != world and rank == 0 and world == 1 and args.gpus > 1:
        log("--gpus %d requested but WORLD_SIZE=1: launch with torch.distributed.run" % args.gpus)

    from sparsepoly_amd import distributed as spdist
    from sparsepoly_amd.engine import HipEngine, canonical_csc
    from sparsepoly_amd.synth import make_problem

    t0 = time.time()
    X, y = make_problem(N_SAMPLES, N_FEATURES, NNZ_PER_ROW, seed=0)
    Xc = X.tocsc()
    Xc.sort_indices()
    n, d = Xc.shape
    nnz = Xc.nnz
    if rank == 0:
        log("data %dx%d nnz=%d generated in %.1fs" % (n, d, nnz, time.time() - t0))

    eng = HipEngine(local_rank, args.precision)
    for kv in filter(None, os.environ.get("SPFM_OPTS", "").split(",")):  # e.g. prb_groups=32
        key, val = kv.split("=")
        eng.set_option(key, int(val))
    conflict = None
    if world > 1:
        lo, hi = spdist.row_block(n, rank, world)
        Xl = canonical_csc(X[lo:hi])
        spdist.init_engine_comm(eng)  # RCCL unique id shipped over the gloo group
        eng.set_data(Xl, y[lo:hi])
        conflict = Xc
    else:
        eng.set_data(Xc, y)
    P0 = 0.01 * np.random.RandomState(0).randn(1, K, d)
    lams = np.ones(K)
    eng.set_params(P0, np.zeros(d), lams)
    eng.configure("pcd", "squared", REG, DEGREE)
    eng.init_pred(DEGREE, True, False)
    y_pred0 = eng.get_y_pred() if (world == 1 and not args.no_cpu_baseline) else None
    t0 = time.time()
    order = eng.set_schedule(args.schedule, np.arange(d, dtype=np.int32), conflict)
    n_batches = eng.n_batches
    if rank == 0:
        log("schedule '%s': %d dependent steps per sweep (%.1fs)" % (args.schedule, n_batches,
                                                                     time.time() - t0))
    ic = np.arange(K, dtype=np.int32)

    def one_step():
        v = eng.cd_linear_epoch(ALPHA)
        v += eng.pcd_epoch(0, DEGREE, BETA, GAMMA, ETA0, ic)
        return v

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    viols = []
    for _ in range(args.warmup):
        viols.append(one_step())
    fence()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        viols.append(one_step())
    fence()
    elapsed = time.perf_counter() - t_start
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / max(args.steps, 1)
    epochs_per_s = args.steps / elapsed
    loss_after = eng.loss_sum()
    P_end, _ = eng.get_params()
    nnz_frac_P = float((P_end != 0).mean())

    # ---- roofline of the dominant kernel (profiled pass, outside the timed region):
    # HIP events on the engine's stream around every launch of that kernel
    roof = None
    persistent = bool(eng.get_option("persistent_active"))
    eng.profile_reset()
    eng.profile_enable(True)
    eng.pcd_epoch(0, DEGREE, BETA, GAMMA, ETA0, ic[:4] if persistent else ic[:2])
    eng.profile_enable(False)
    g_ms, g_launch, g_nnz = eng.profile_get(0)
    s_ms, s_launch, s_nnz = eng.profile_get(1)
    tsz = 4 if args.precision == "f32" else 8
    if persistent:
        # pcd_prb_kernel = one whole component pass (gather + exchange + chain + scatter);
        # per column entry: row 4 + value T + (yhat, y) read 2T + yhat write T +
        # A[i,1..m-1] read and write 2T(m-1)  (= SURVEY 8d's k*nnz*(20 + 8(m-1)) term)
        kname, bytes_per_nnz = "pcd_prb_kernel", 4 + tsz + 3 * tsz + 2 * tsz * (DEGREE - 1)
    else:
        # pcd_grad_kernel reads per column entry: row 4 + value T + A[i,1..m-1] T(m-1) +
        # (yhat, y) 2T
        kname, bytes_per_nnz = "pcd_grad_kernel", 4 + tsz + tsz * (DEGREE - 1) + 2 * tsz
    if g_launch > 0 and g_ms > 0:
        avg_us = 1e3 * g_ms / g_launch
        bytes_per_launch = bytes_per_nnz * g_nnz / g_launch
        achieved = bytes_per_launch / (avg_us * 1e-6) / 1e9
        # measured HBM-side traffic per launch: PMC counters cannot be read from inside
        # this process; the value comes from the committed rocprofv3 --pmc passes of the
        # same command (profiles/r01_traffic.json, collected per MI355X_MICROARCH.md HBM)
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            if kname in tj and N_SAMPLES == 1_000_000 and N_FEATURES == 100_000 and world == 1:
                traffic = round(1024.0 * (tj[kname]["fetch_kb_per_launch"]
                                          + tj[kname]["write_kb_per_launch"]), 1)
        except Exception:
            traffic = None
        roof = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic, "avg_launch_us": round(avg_us, 3),
                "alg_bytes_per_launch": round(bytes_per_launch, 1),
                "launches_timed": int(g_launch)}
        if persistent:
            # 0 = row state in global memory, 1 / 2 = row block resident in LDS:
            # then `traffic` is below the algorithmic bytes, which charge every gather / scatter
            roof["row_block_in_lds"] = int(eng.get_option("prb_lds_active"))
            roof["dependent_steps_per_launch"] = n_batches
            roof["us_per_dependent_step_in_kernel"] = round(avg_us / n_batches, 3)
        else:
            roof["sync_kernel_avg_us"] = round(1e3 * s_ms / max(s_launch, 1), 3)
    # whole-iteration algorithmic traffic, f32 layout
    nnz_glob = nnz
    b_alg = 8 * nnz_glob + 4 * (DEGREE - 1) * n * K + K * nnz_glob * (20 + 8 * (DEGREE - 1)) \
        + 20 * nnz_glob
    iter_gbs = b_alg / (ms_per_step * 1e-3) / 1e9

    # ---- CPU baseline: the oracle on a bounded sample of the same workload
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc

        orc.build()
        ds = orc.CSC(Xc)
        Pc = np.ascontiguousarray(P0[0].copy())
        wc = np.zeros(d)
        yp = np.ascontiguousarray(y_pred0)
        cn = np.asarray(Xc.multiply(Xc).sum(axis=0)).ravel()
        regc = orc.Regularizer(REG)
        regc.init_cache_pcd(DEGREE, d, K)
        A = np.zeros((n, DEGREE + 1))
        jf = np.ascontiguousarray(order)
        t1 = time.perf_counter()
        orc.cd_linear_epoch(wc, ds, y, yp, cn, ALPHA, "squared", jf)
        t_lin = time.perf_counter() - t1
        n_pass = 1
        t1 = time.perf_counter()
        orc.pcd_epoch(Pc, ds, y, yp, lams, DEGREE, BETA, GAMMA, ETA0, regc, "squared", A,
                      ic[:n_pass], jf)
        t_pass = (time.perf_counter() - t1) / n_pass
        cpu_epoch_s = t_lin + K * t_pass
        cpu = {"value": round(1.0 / cpu_epoch_s, 6), "unit": "epochs/s", "cores": 1,
               "kind": "port",
               "sample": "1 cd_linear epoch (%.2fs) + %d of %d pcd component passes (%.2fs each) "
                         "of the same 1Mx100k workload in the same column order, extrapolated "
                         "to %d passes" % (t_lin, n_pass, K, t_pass, K),
               "host_cpus": os.cpu_count()}

    if rank == 0:
        out = {
            "metric": "pcd_epochs_per_sec",
            "value": round(epochs_per_s, 4),
            "unit": "epochs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32" if args.precision == "f32" else "f64",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: degree=2 n_components=30 "
                                   "regularizer=squaredl12 solver=pcd fit_linear=True on "
                                   "%dx%d CSR nnz=%d (~50/row)" % (n, d, nnz),
                       "schedule": args.schedule, "dependent_steps_per_sweep": n_batches,
                       "alpha": ALPHA, "beta": BETA, "gamma": GAMMA,
                       "parallelism": ("rows sharded x%d, per-step %s all-reduce"
                                       % (world, "host-shm (rehearsal)"
                                          if os.environ.get("SPFM_COMM") == "shm" else "RCCL"))
                       if world > 1 else "single GPU"},
            "roofline": roof,
            "cpu_baseline": cpu,
            "iteration_alg_GBs": round(iter_gbs, 2),
            "iteration_alg_frac_of_hbm_peak": round(iter_gbs / HBM_PEAK_GBS, 5),
            "us_per_dependent_step": round(1e3 * ms_per_step / ((K + 1) * n_batches), 3),
            "viol": [round(float(v), 6) for v in viols],
            "sum_loss_after": round(float(loss_after), 6),
            "nonzero_frac_P_after": round(nnz_frac_P, 4),
        }
        print(json.dumps(out), flush=True)
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
