#!/usr/bin/env python3
"""Headline benchmark: PCD epochs/sec on BASELINE config 2 (degree=2, n_components=30,
regularizer='squaredl12', solver='pcd', 1M x 100k synthetic CSR, ~50 nnz/row).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {2,3,4}]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

``--gpus N`` without a launcher (WORLD_SIZE unset) starts the N ranks itself: N child processes
of this script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, started BEFORE anything in
the parent touches the GPU; the parent relays rank 0's JSON line and exits with the children's
status.  On a box with fewer than N devices the ranks share the devices (a rehearsal: host
shared-memory communicator, exchange slabs mapped through IPC on one GPU) and the line says so.

A "step" is one full training iteration of the reference's fit loop
(sparse_factorization_machines.py:196-256 / :287-350): one cd_linear epoch, the lower-order
epochs (fit_lower='explicit') and the top-order epoch, on data already resident in HBM.  The
timed region calls exactly what the estimators call per iteration (``HipEngine.*_epoch`` through
the C ABI).  ``--config 3`` / ``4`` run the other BASELINE configurations on the same matrix
(degree 3 / omegati / pcd / k=16; degree 2 / omegacs / pbcd / k=30).

Prints ONE JSON line (rank 0).  Extra objects:
  roofline        dominant kernel: algorithmic bytes per launch / average launch duration (HIP
                  events on the engine's stream) vs 8 TB/s HBM.  `traffic` = HBM bytes per
                  launch from the rocprofv3 --pmc passes committed under profiles/ -- only if
                  that file was collected with this engine version, else null
  cpu_baseline    the CPU oracle (float64, 1 thread) on ONE FULL iteration of the same workload
                  in the same column order, timed on this host
  exact_schedule  the estimators' default schedule (reference order, parity at fit() level):
                  dependent steps per sweep and ms per iteration from 3 component passes
  f64             ms per iteration with float64 storage (the reference's own arithmetic)
  concurrent_fits four independent fits of the workload at once on the one GPU (a regularisation
                  path; sparsepoly_amd/concurrent.py): aggregate epochs/s and bytes/s.  `value`
                  stays the rate of ONE fit
N > 1: rows are sharded over the ranks; the persistent passes exchange their per-step totals
through peer-mapped slabs inside the kernels (no per-step collective).  Default is WEAK scaling:
N times the rows and columns (the family that ends in BASELINE configs[4]: 10M x 1M on 8 GPUs),
value = N x epochs/s; ``--scaling strong`` shards the 1M x 100k matrix itself.  Set-up is O(1/N)
per rank: a rank draws only its own rows from the counter-based generator and hands them to the
library as CSR; rank 0 alone draws the global STRUCTURE (no values), colours it and broadcasts
order / batch boundaries, which the other ranks install with spfm_set_schedule_raw.  The line
carries per-rank set-up seconds and peak host RSS.
"""
import argparse
import json
import os
import resource
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_SAMPLES = int(os.environ.get("SPFM_BENCH_N", 1_000_000))
N_FEATURES = int(os.environ.get("SPFM_BENCH_D", 100_000))
NNZ_PER_ROW = 50
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
# profiles/<round>_traffic.json records the library's build tag (hash of its sources,
# spfm_build_tag) it was collected with; `traffic` is only reported when that is the library
# running now
TRAFFIC_FILE = "r03_traffic.json"

# hyper-parameters: well conditioned and such that P does NOT collapse to
# zero (with gamma = 1 every coordinate is thresholded away in the first epoch and the scatter
# half of every later step is a no-op).  The non-zero fraction of P after the run is reported.
CONFIGS = {
    2: dict(name="BASELINE configs[1]", solver="pcd", reg="squaredl12", degree=2, k=30,
            alpha=1.0, beta=10.0, gamma=1e-4),
    3: dict(name="BASELINE configs[2]", solver="pcd", reg="omegati", degree=3, k=16,
            alpha=1.0, beta=10.0, gamma=1e-6),
    4: dict(name="BASELINE configs[3]", solver="pbcd", reg="omegacs", degree=2, k=30,
            alpha=1.0, beta=1.0, gamma=1e-3),
}
ETA0 = 1.0


def log(msg):
    print("[bench] " + msg, file=sys.stderr, flush=True)


def alg_bytes(cfg, n, nnz, tsz=4):
    """Compulsory traffic of one iteration, SURVEY.md section 8(d) (T-byte values, int32 ids)."""
    m, k = cfg["degree"], cfg["k"]
    lin = (4 + 4 * tsz) * nnz
    if cfg["solver"] == "pcd":
        return lin + sum(2 * 4 * nnz + tsz * (mm - 1) * n * k
                         + k * nnz * (4 + 4 * tsz + 2 * tsz * (mm - 1)) for mm in range(2, m + 1))
    return lin + 2 * 4 * nnz + tsz * (m - 1) * n * k + nnz * (4 + 4 * tsz + 2 * tsz * (m - 1) * k)


def device_count():
    """Devices of this box, asked of a short-lived child so that the launcher itself never
    initialises the GPU (it must stay free to start the ranks)."""
    try:
        out = subprocess.run([sys.executable, "-c",
                              "import torch; print(torch.cuda.device_count())"],
                             capture_output=True, text=True, timeout=600)
        return max(0, int(out.stdout.strip().splitlines()[-1]))
    except Exception:
        return 0


def launch_ranks(n_ranks, argv):
    """Start the N ranks as child processes, relay rank 0's JSON line, return the exit status."""
    ndev = int(os.environ.get("SPFM_BENCH_DEVICES", 0)) or device_count()
    if ndev < 1:
        log("no GPU visible: cannot start %d ranks" % n_ranks)
        return 1
    with socket.socket() as sk:  # a free port for the control-plane rendezvous
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    shared = ndev < n_ranks
    if shared:
        log("%d ranks on %d device(s): REHEARSAL -- ranks share a GPU (host-shm communicator, "
            "exchange slabs IPC-mapped on one device)" % (n_ranks, ndev))
    procs = []
    for r in range(n_ranks):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r % ndev), WORLD_SIZE=str(n_ranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   SPFM_BENCH_LAUNCHER="self", SPFM_BENCH_NDEV=str(ndev))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if shared:
            env["SPFM_DEVICE"] = str(r % ndev)
            env.setdefault("SPFM_COMM", "shm")
            # the persistent kernels of all ranks on a device must be co-resident (the library
            # checks and otherwise falls back to the multi-kernel engine): share the CUs out
            per_dev = -(-n_ranks // ndev)
            env.setdefault("SPFM_OPTS", "pcdw_groups=%d,pbprb_groups=%d,prb_groups=%d"
                           % (240 // per_dev, 240 // per_dev, min(64, 240 // per_dev)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    status = procs[0].returncode
    for pr in procs[1:]:
        try:
            pr.wait(timeout=300)
        except subprocess.TimeoutExpired:
            pr.kill()  # exactly the child we started
            pr.wait()
        status = status or pr.returncode
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    return status


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the exact-schedule and f64 measurements")
    ap.add_argument("--precision", default="f32", choices=["f32", "f64"])
    ap.add_argument("--schedule", default="colored", choices=["colored", "exact"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = N times the rows AND columns of the workload (rows sharded; "
                         "same dependent steps per sweep, fixed work per GPU and step; the family "
                         "that ends in BASELINE configs[4] = 10M x 1M on 8 GPUs); strong = the "
                         "same 1M x 100k matrix sharded over the ranks")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # no launcher: start the ranks ourselves (children, never a re-exec; nothing in this
        # process has touched the GPU)
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    # stdout carries ONE JSON line and nothing else: libraries that write to file descriptor 1
    # (gloo announces its connections there) are sent to stderr for the whole run
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    cfg = CONFIGS[args.config]
    K, DEGREE = cfg["k"], cfg["degree"]

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
        # timings): gloo.  The data path's collectives are the engine's own.
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    if args.gpus != world and rank == 0:
        log("--gpus %d but WORLD_SIZE=%d: the line reports the %d rank(s) that really ran"
            % (args.gpus, world, world))

    from sparsepoly_amd import _capi
    from sparsepoly_amd import distributed as spdist
    from sparsepoly_amd.engine import HipEngine
    from sparsepoly_amd.synth import make_problem

    ENGINE_TAG = _capi.build_tag()  # hash of the library's sources, written in at build time

    t0 = time.time()
    scale = world if (world > 1 and args.scaling == "weak") else 1
    n, d = N_SAMPLES * scale, N_FEATURES * scale
    setup = {}  # seconds of this rank's set-up phases
    if world > 1:
        # O(1/N) per rank: only the own rows, straight from the counter-based generator (CSR)
        lo, hi = spdist.row_block(n, rank, world)
        X, y = make_problem(n, d, NNZ_PER_ROW, seed=0, row_range=(lo, hi))
        Xc = None
        nnz_local = int(X.nnz)
        t = torch.tensor([nnz_local], dtype=torch.int64)
        dist.all_reduce(t)
        nnz = int(t.item())
    else:
        X, y = make_problem(n, d, NNZ_PER_ROW, seed=0)
        Xc = X.tocsc()
        Xc.sort_indices()
        nnz = Xc.nnz
    setup["data_s"] = round(time.time() - t0, 2)
    if rank == 0:
        log("data %dx%d nnz=%d: this rank's %d rows generated in %.1fs"
            % (n, d, nnz, X.shape[0], time.time() - t0))
    n_orders = DEGREE - 1
    P0 = 0.01 * np.random.RandomState(0).randn(n_orders, K, d)
    lams = np.ones(K)
    ic = np.arange(K, dtype=np.int32)
    jf0 = np.arange(d, dtype=np.int32)

    def global_structure():
        """CSC structure (indices only) of the whole matrix: what a colouring needs.  Rank 0."""
        from sparsepoly_amd.synth import make_csr

        S = make_csr(n, d, NNZ_PER_ROW, seed=0, structure_only=True).tocsc()
        S.sort_indices()
        return S

    def make_engine(precision, schedule):
        from sparsepoly_amd.schedule import Schedule

        eng = HipEngine(local_rank, precision)
        for kv in filter(None, os.environ.get("SPFM_OPTS", "").split(",")):  # e.g. prb_groups=32
            key, val = kv.split("=")
            eng.set_option(key, int(val))
        t1 = time.time()
        if world > 1:
            spdist.init_engine_comm(eng)
            spdist.connect_peers(eng)  # persistent passes with the in-kernel xGMI exchange
        eng.set_data(X if world > 1 else Xc, y)  # CSR shard: transposed inside the library
        eng.set_params(P0, np.zeros(d), lams)
        eng.configure(cfg["solver"], "squared", cfg["reg"], DEGREE)
        eng.init_pred(DEGREE, True, DEGREE == 3)
        setup["engine_s"] = round(time.time() - t1, 2)
        t1 = time.time()
        if world > 1:
            # rank 0 colours the global structure once (the library's own policy for the step
            # width, decided from global inputs); everybody else installs the result
            if rank == 0:
                S = global_structure()
                setup["structure_s"] = round(time.time() - t1, 2)
                order = eng.set_schedule(schedule, jf0, S)
                sch = eng.get_schedule(schedule)
                payload = [sch.order, sch.batch_ptr]
                del S
            else:
                payload = [None, None]
            dist.broadcast_object_list(payload, src=0)
            if rank != 0:
                order = eng.install_schedule(Schedule(payload[0], payload[1], schedule, (n, d)))
        else:
            order = eng.set_schedule(schedule, jf0)
        setup["schedule_s"] = round(time.time() - t1, 2)
        return eng, order, time.time() - t1

    def iteration(eng, comps=None):
        """One pass of the reference's iteration body; `comps` limits the pcd component loop."""
        v = eng.cd_linear_epoch(cfg["alpha"])
        for deg in list(range(2, DEGREE)) + [DEGREE]:
            o = DEGREE - deg if deg != DEGREE else 0
            if cfg["solver"] == "pcd":
                v += eng.pcd_epoch(o, deg, cfg["beta"], cfg["gamma"], ETA0,
                                   ic if comps is None else ic[:comps])
            else:
                v += eng.pbcd_epoch(o, deg, cfg["beta"], cfg["gamma"], ETA0)
        return v

    eng, order, t_sched = make_engine(args.precision, args.schedule)
    y_pred0 = eng.get_y_pred() if (world == 1 and not args.no_cpu_baseline) else None
    n_batches = eng.n_batches
    if rank == 0:
        log("schedule '%s': %d dependent steps per sweep (%.1fs)" % (args.schedule, n_batches,
                                                                     t_sched))

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    viols = []

    def warm_up(eng):
        """The untimed steps; with several ranks all of them learn whether any rank failed."""
        ok, why = 1, ""
        try:
            for _ in range(args.warmup):
                viols.append(iteration(eng))
            torch.cuda.synchronize()
        except RuntimeError as exc:
            if dist is None:
                raise
            ok, why = 0, str(exc)
        if dist is not None:
            t = torch.tensor([ok], dtype=torch.int32)
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            if int(t.item()) == 0 and ok == 1:
                why = "another rank failed"
            ok = int(t.item())
        return ok, why

    ok, why = warm_up(eng)
    if not ok:
        # the in-kernel peer exchange did not come up on this node (the warm-up aborted on some
        # rank): every rank rebuilds its engine with the per-step collective instead and the
        # line says so
        if os.environ.get("SPFM_PEER", "1") == "0":
            raise RuntimeError("warm-up failed: " + why)
        log("rank %d: warm-up failed with the in-kernel peer exchange (%s); falling back to the "
            "per-step collective" % (rank, why))
        try:
            eng.close()
        except Exception:
            pass
        os.environ["SPFM_PEER"] = "0"
        del viols[:]
        eng, order, t_sched = make_engine(args.precision, args.schedule)
        n_batches = eng.n_batches
        ok, why = warm_up(eng)
        if not ok:
            raise RuntimeError("warm-up failed: " + why)
    fence()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        viols.append(iteration(eng))
    fence()
    elapsed = time.perf_counter() - t_start
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = 1e3 * elapsed / max(args.steps, 1)
    # weak scaling: the unit is one epoch over a config-sized (1M x 100k) share of the matrix,
    # so the whole-job value is (shares = ranks) x epochs/s
    epochs_per_s = scale * args.steps / elapsed
    loss_after = eng.loss_sum()
    P_end, _ = eng.get_params()
    nnz_frac_P = float((P_end != 0).mean())

    # ---- roofline of the dominant kernel (profiled epoch, outside the timed region): HIP
    # events on the engine's stream around every launch of that kernel
    tsz = 4 if args.precision == "f32" else 8
    roof = None
    eng.profile_reset()
    eng.profile_enable(True)
    if cfg["solver"] == "pcd":
        persistent = bool(eng.get_option("persistent_active"))
        eng.pcd_epoch(0, DEGREE, cfg["beta"], cfg["gamma"], ETA0, ic[:4] if persistent else ic[:2])
        which = 0
        if persistent:
            # one launch = one component pass; per column entry: row 4 + value T + (yhat, y)
            # read 2T + yhat write T + A[i,1..m-1] read and write 2T(m-1)
            # (the wide pass -- steps of more than 64 columns, DESIGN 3d -- moves the same bytes)
            kname = "pcdw_kernel" if eng.get_option("wide_active") else "pcd_prb_kernel"
            bytes_per_nnz = 4 + 4 * tsz + 2 * tsz * (DEGREE - 1)
        else:
            kname, bytes_per_nnz = "pcd_grad_kernel", 4 + 3 * tsz + tsz * (DEGREE - 1)
    else:
        persistent = True
        eng.pbcd_epoch(0, DEGREE, cfg["beta"], cfg["gamma"], ETA0)
        persistent = bool(eng.get_option("pbprb_active"))
        which = 2
        # one launch = one pbcd epoch; per entry: row 4 + value T + (yhat, y) 2T + yhat write T
        # + A[i,1..m-1,:] read and write 2T(m-1)k
        kname = "pbcd_prb_kernel" if persistent else "pbcd_grad_kernel"
        bytes_per_nnz = (4 + 4 * tsz + 2 * tsz * (DEGREE - 1) * K) if persistent else \
            (4 + 3 * tsz + tsz * (DEGREE - 1) * K)
    eng.profile_enable(False)
    g_ms, g_launch, g_nnz = eng.profile_get(which)
    if g_launch > 0 and g_ms > 0:
        avg_us = 1e3 * g_ms / g_launch
        bytes_per_launch = bytes_per_nnz * g_nnz / g_launch
        achieved = bytes_per_launch / (avg_us * 1e-6) / 1e9
        traffic, traffic_src = None, "not collected for this build of the library (%s)" % ENGINE_TAG
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)))
            ent = tj.get("config%d" % args.config, {}).get(kname)
            if (ent and tj.get("engine_tag") == ENGINE_TAG and N_SAMPLES == 1_000_000
                    and N_FEATURES == 100_000 and world == 1 and args.precision == "f32"):
                traffic = round(1024.0 * (ent["fetch_kb_per_launch"]
                                          + ent["write_kb_per_launch"]), 1)
                traffic_src = "profiles/%s (rocprofv3 --pmc passes, library build %s)" \
                    % (TRAFFIC_FILE, ENGINE_TAG)
        except Exception:
            pass
        roof = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_us": round(avg_us, 3),
                "alg_bytes_per_launch": round(bytes_per_launch, 1),
                "launches_timed": int(g_launch)}
        if persistent:
            roof["dependent_steps_per_launch"] = n_batches
            roof["us_per_dependent_step_in_kernel"] = round(avg_us / n_batches, 3)
            if cfg["solver"] == "pcd":
                # 0 = row state in global memory, 1 / 2 = row block resident in LDS
                roof["row_block_in_lds"] = int(eng.get_option("prb_lds_active"))
    # what really ran: the ranks of the engine's communicator, the exchange it used, whether a
    # persistent pass had to be redone on the multi-kernel engine; per-rank set-up cost
    ranks_seen = int(eng.get_option("n_ranks"))
    peer_exchange = bool(eng.get_option("peer_ready"))
    fallbacks = int(eng.get_option("persistent_fallbacks"))
    setup["peak_rss_gb"] = round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576.0, 2)
    setup["rank"] = rank
    setups = [setup]
    if dist is not None:
        setups = [None] * world
        dist.all_gather_object(setups, setup)
    b_alg = alg_bytes(cfg, n, nnz)
    iter_gbs = b_alg / (ms_per_step * 1e-3) / 1e9
    steps_per_iter = (1 + (K * (DEGREE - 1) if cfg["solver"] == "pcd" else (DEGREE - 1))) * n_batches
    eng.close()

    # ---- the estimators' default schedule ('exact': the reference's own order) and f64 storage
    extras = {}
    skip = os.environ.get("SPFM_BENCH_SKIP", "").split(",")  # diagnostics: leave extras out
    if rank == 0 and world == 1 and not args.no_extras and cfg["solver"] == "pcd" \
            and "exact" not in skip:
        e2, _, ts = make_engine(args.precision, "exact")
        nb2 = e2.n_batches
        iteration(e2, comps=1)  # builds the entry stream, warms up
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        e2.cd_linear_epoch(cfg["alpha"])
        torch.cuda.synchronize()
        t_lin = time.perf_counter() - t1
        t1 = time.perf_counter()
        e2.pcd_epoch(0, DEGREE, cfg["beta"], cfg["gamma"], ETA0, ic[:3])
        torch.cuda.synchronize()
        t_pass = (time.perf_counter() - t1) / 3
        relax_steps = int(e2.get_option("relax_steps"))
        e2.close()
        extras["exact_schedule"] = {
            "dependent_steps_per_sweep": nb2, "schedule_build_s": round(ts, 2),
            # degree-2 pcd passes run the reference order as merged steps whose shared rows the
            # chains replay; cd_linear keeps the strict steps
            "merged_steps_per_pcd_sweep": relax_steps,
            "ms_per_iteration": round(1e3 * (t_lin + K * t_pass * (DEGREE - 1)), 1),
            "measured": "1 cd_linear epoch (%.0f ms) + 3 top-order component passes (%.0f ms "
                        "each), extrapolated to %d passes per order" % (1e3 * t_lin, 1e3 * t_pass, K),
            "is_estimator_default": True}
    if rank == 0 and world == 1 and not args.no_extras and args.precision == "f32" \
            and "f64" not in skip:
        e3, _, _ = make_engine("f64", args.schedule)
        iteration(e3)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        iteration(e3)
        torch.cuda.synchronize()
        extras["f64"] = {"ms_per_iteration": round(1e3 * (time.perf_counter() - t1), 2),
                         "schedule": args.schedule}
        e3.close()

    # ---- four independent fits of the same workload at once (a regularisation path): one handle,
    # stream and host thread per fit, their persistent passes side by side on disjoint CUs
    if rank == 0 and world == 1 and not args.no_extras and cfg["solver"] == "pcd":
        import threading

        from sparsepoly_amd.engine import co_tenancy

        F, its = 4, 3
        with co_tenancy(F):
            engs = [make_engine(args.precision, args.schedule)[0] for _ in range(F)]
        for e in engs:
            iteration(e)  # builds the entry streams, warms up
        torch.cuda.synchronize()
        bar = threading.Barrier(F + 1)
        spent = [0.0] * F

        def fit_loop(f):
            bar.wait()
            t = time.perf_counter()
            for _ in range(its):
                iteration(engs[f])
            spent[f] = time.perf_counter() - t

        th = [threading.Thread(target=fit_loop, args=(f,)) for f in range(F)]
        for t in th:
            t.start()
        bar.wait()
        t1 = time.perf_counter()
        for t in th:
            t.join()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t1
        fb = [int(e.get_option("persistent_fallbacks")) for e in engs]
        for e in engs:
            e.close()
        extras["concurrent_fits"] = {
            "fits": F, "iterations_each": its,
            "ms_per_iteration_per_fit": [round(1e3 * t / its, 1) for t in spent],
            "aggregate_epochs_per_s": round(F * its / wall, 3),
            "aggregate_GBps": round(F * its * b_alg / wall / 1e9, 1),
            "aggregate_frac_of_hbm_peak": round(F * its * b_alg / wall / 1e9 / HBM_PEAK_GBS, 4),
            "persistent_fallbacks": fb,
            "note": "independent fits (different models, same matrix); each equals its solo run "
                    "bit for bit (tests/test_hip_concurrent.py); `value` above is ONE fit"}

    # ---- CPU baseline: the oracle on one full iteration of the same workload
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as orc

        orc.build()
        ds = orc.CSC(Xc)
        Pc = np.ascontiguousarray(P0.copy())
        wc = np.zeros(d)
        yp = np.ascontiguousarray(y_pred0)
        cn = np.asarray(Xc.multiply(Xc).sum(axis=0)).ravel()
        regc = orc.Regularizer(cfg["reg"])
        jf = np.ascontiguousarray(order)
        t1 = time.perf_counter()
        orc.cd_linear_epoch(wc, ds, y, yp, cn, cfg["alpha"], "squared", jf)
        if cfg["solver"] == "pcd":
            regc.init_cache_pcd(DEGREE, d, K)
            A = np.zeros((n, DEGREE + 1))
            for deg in list(range(2, DEGREE)) + [DEGREE]:
                o = DEGREE - deg if deg != DEGREE else 0
                orc.pcd_epoch(Pc[o], ds, y, yp, lams, deg, cfg["beta"], cfg["gamma"], ETA0, regc,
                              "squared", A, ic, jf)
        else:
            regc.init_cache_pbcd(DEGREE, d, K)
            A = np.zeros((n, DEGREE + 1, K))
            dA = np.zeros((n, DEGREE, K))
            Pt = np.ascontiguousarray(Pc[0].T)
            orc.pbcd_epoch(Pt, ds, y, yp, lams, DEGREE, cfg["beta"], cfg["gamma"], ETA0, regc,
                           "squared", A, dA, jf)
        cpu_s = time.perf_counter() - t1
        cpu = {"value": round(1.0 / cpu_s, 6), "unit": "epochs/s", "cores": 1, "kind": "port",
               "sample": "one full iteration (cd_linear + every component pass / block epoch) of "
                         "the same %dx%d workload in the same column order: %.1f s" % (n, d, cpu_s),
               "host_cpus": os.cpu_count()}

    if rank == 0:
        out = {
            "metric": "pcd_epochs_per_sec" if cfg["solver"] == "pcd" else "pbcd_epochs_per_sec",
            "value": round(epochs_per_s, 4),
            "unit": "epochs/s",
            "n_gpus": ranks_seen,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True,
            "scaling": "weak" if (world == 1 or args.scaling == "weak") else "strong",
            "vs_baseline": None,
            "dtype": "f32" if args.precision == "f32" else "f64",
            "data": "synthetic",
            "config": {"workload": "%s: degree=%d n_components=%d regularizer=%s solver=%s "
                                   "fit_linear=True fit_lower=explicit on %dx%d CSR nnz=%d "
                                   "(~50/row)" % (cfg["name"], DEGREE, K, cfg["reg"],
                                                  cfg["solver"], n, d, nnz),
                       "scaling_note": ("weak scaling: %d x the rows and columns of the config (each "
                                        "rank owns a 1M-row shard); value = %d x epochs/s of the "
                                        "%dx%d problem" % (scale, scale, n, d)) if scale > 1 else
                       ("strong scaling: the config's matrix sharded by rows" if world > 1 else
                        "single GPU: the configuration BASELINE.json's metric is quoted on"),
                       "schedule": args.schedule, "dependent_steps_per_sweep": n_batches,
                       "alpha": cfg["alpha"], "beta": cfg["beta"], "gamma": cfg["gamma"],
                       "parallelism": ("rows sharded over %d ranks (one process per GPU), per-step "
                                       "exchange: %s; control plane gloo, communicator %s"
                                       % (ranks_seen,
                                          "in-kernel peer-mapped slabs (no collective)"
                                          if peer_exchange else "one all-reduce per dependent step",
                                          "host-shm (ranks share a device: rehearsal)"
                                          if os.environ.get("SPFM_COMM") == "shm" else "RCCL"))
                       if world > 1 else "single GPU",
                       "ranks_seen": ranks_seen,
                       "devices_used": min(world, int(os.environ.get("SPFM_BENCH_NDEV", world))),
                       "launcher": os.environ.get("SPFM_BENCH_LAUNCHER", "external"),
                       "persistent_fallbacks": fallbacks,
                       "setup_per_rank": setups},
            "roofline": roof,
            "cpu_baseline": cpu,
            "iteration_alg_GBs": round(iter_gbs, 2),
            "iteration_alg_frac_of_hbm_peak": round(iter_gbs / HBM_PEAK_GBS, 5),
            "us_per_dependent_step": round(1e3 * ms_per_step / steps_per_iter, 3),
            "viol": [round(float(v), 6) for v in viols],
            "sum_loss_after": round(float(loss_after), 6),
            "nonzero_frac_P_after": round(nnz_frac_P, 4),
            "engine_tag": ENGINE_TAG,
        }
        out.update(extras)
        print(json.dumps(out), file=json_out, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
