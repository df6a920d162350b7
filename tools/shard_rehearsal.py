#!/usr/bin/env python3
"""What ONE rank of an N-GPU run of a BASELINE configuration computes per dependent step, measured
on one GPU: the rank's rows [0, n/N) of the matrix, the schedule coloured on the GLOBAL structure
(as rank 0 does for everybody), no peers -- i.e. the per-step time of a rank
without the cross-GPU hop.  Default: BASELINE configs[4] (10M x 1M over 8 GPUs).

    python tools/shard_rehearsal.py [--n 10000000] [--d 1000000] [--ranks 8] [--passes 2]
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
from sparsepoly_amd.synth import make_csr, make_problem  # noqa: E402

K, BETA, GAMMA, ALPHA = 30, 10.0, 1e-4, 1.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=10_000_000)
    ap.add_argument("--d", type=int, default=1_000_000)
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--passes", type=int, default=30,
                    help="components per timed epoch (30 = a whole iteration of the configuration)")
    ap.add_argument("--groups", default="0", help="pcdw_groups values to try (0 = default)")
    args = ap.parse_args()
    n, d, N = args.n, args.d, args.ranks
    t0 = time.time()
    S = make_csr(n, d, 50, seed=0, structure_only=True).tocsc()
    S.sort_indices()
    t_struct = time.time() - t0
    print("[shard] global structure %dx%d nnz=%d in %.0fs" % (n, d, S.nnz, t_struct),
          file=sys.stderr, flush=True)
    lo, hi = 0, n // N
    X, y = make_problem(n, d, 50, seed=0, row_range=(lo, hi))
    for groups in [int(g) for g in args.groups.split(",")]:
        run(args, S, X, y, n, d, N, lo, hi, t_struct, groups)


def run(args, S, X, y, n, d, N, lo, hi, t_struct, groups):
    eng = HipEngine(0, "f32")
    if groups:
        eng.set_option("pcdw_groups", groups)
    t0 = time.time()
    eng.set_data(X, y)
    t_data = time.time() - t0
    eng.set_params(0.01 * np.random.RandomState(0).randn(1, K, d), np.zeros(d), np.ones(K))
    eng.configure("pcd", "squared", "squaredl12", 2)
    eng.init_pred(2, True, False)
    t0 = time.time()
    eng.set_schedule("colored", np.arange(d, dtype=np.int32), S)
    t_sched = time.time() - t0
    ic = np.arange(args.passes, dtype=np.int32)
    eng.cd_linear_epoch(ALPHA)
    eng.pcd_epoch(0, 2, BETA, GAMMA, 1.0, ic[:2])      # builds the streams
    t0 = time.perf_counter()
    eng.cd_linear_epoch(ALPHA)
    t_lin = time.perf_counter() - t0
    t0 = time.perf_counter()
    eng.pcd_epoch(0, 2, BETA, GAMMA, 1.0, ic)
    t_pass = (time.perf_counter() - t0) / args.passes
    out = dict(workload="%dx%d over %d ranks: rows [%d, %d) of rank 0, schedule from the global "
               "structure" % (n, d, N, lo, hi), shard_nnz=int(X.nnz),
               pcdw_groups=eng.get_option("pcdw_groups"),
               steps_per_sweep=eng.n_batches, wide_active=eng.get_option("wide_active"),
               wide_rows_in_lds=eng.get_option("wide_lds_active"),
               persistent_fallbacks=eng.get_option("persistent_fallbacks"),
               structure_s=round(t_struct, 1), set_data_s=round(t_data, 2),
               schedule_s=round(t_sched, 1),
               ms_per_cd_linear_epoch=round(1e3 * t_lin, 2),
               ms_per_component_pass=round(1e3 * t_pass, 2),
               us_per_dependent_step=round(1e6 * t_pass / eng.n_batches, 2),
               passes_timed=args.passes,
               ms_per_iteration_without_hops=round(1e3 * (t_lin + K * t_pass), 0))
    eng.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
