#!/usr/bin/env python3
"""Headline benchmark: PCD epochs/sec on BASELINE config 2 (degree=2, n_components=30,
regularizer='squaredl12', solver='pcd', 1M x 100k synthetic CSR, ~50 nnz/row).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {2,3,4}]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

``--gpus N`` without a launcher (WORLD_SIZE unset) starts the N ranks itself: N child processes
of this script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, started BEFORE anything in
the parent touches the GPU.  The parent supervises them: it relays rank 0's JSON line, and the
first rank that exits non-zero (or a deadline, or a signal to the launcher) ends the others --
exactly the children it started -- and becomes the exit status within seconds.  On a box with
fewer than N devices the ranks share the devices (a REHEARSAL: host shared-memory communicator,
exchange slabs mapped through IPC on one GPU); the line then says ``"rehearsal": true``,
``n_gpus`` = the devices really used, and ``value`` carries no N x multiplier.

A "step" is one full training iteration of the reference's fit loop
(sparse_factorization_machines.py:196-256 / :287-350): one cd_linear epoch, the lower-order
epochs (fit_lower='explicit') and the top-order epoch, on data already resident in HBM.  The
timed region calls exactly what the estimators call per iteration (``HipEngine.*_epoch`` through
the C ABI).

Prints ONE JSON line (rank 0).  Extra objects at N = 1:
  roofline        dominant kernel: algorithmic bytes per launch / average launch duration (HIP
                  events on the engine's stream) vs 8 TB/s HBM.  `traffic` = HBM bytes per
                  launch from the rocprofv3 --pmc passes committed under profiles/ -- only if
                  that file was collected with this build of the library, else null
  cpu_baseline    the CPU oracle (float64, 1 thread) on ONE FULL iteration of the same workload
                  in the same column order, timed on this host
  other_configs   BASELINE configs[2] (degree 3, omegati, pcd, k=16) and configs[3] (omegacs,
                  pbcd, k=30) on the same matrix: ms per iteration over --steps timed iterations,
                  each with its own roofline object and violation sums
  exact_schedule  the estimators' default schedule (the reference's own column order, parity at
                  fit() level): one WHOLE iteration, timed
  f64             ms per iteration with float64 storage (the reference's own arithmetic)
  concurrent_fits four independent fits of the workload at once on the one GPU (a regularisation
                  path; sparsepoly_amd/concurrent.py): aggregate epochs/s and bytes/s.  `value`
                  stays the rate of ONE fit
N > 1: rows are sharded over the ranks; the persistent passes exchange their per-step totals
through peer-mapped slabs inside the kernels (no per-step collective).  ONE line carries both
families: ``value`` = STRONG scaling (the 1M x 100k matrix of config 2 itself, sharded by rows:
the same epoch on N GPUs, so value(1) is the single-GPU figure), the extra ``weak`` = N times the
rows AND columns (the family that ends in BASELINE configs[4]: 10M x 1M on 8 GPUs), its value =
N x epochs/s of that N-times larger problem, and the extra ``independent_fits`` = one whole
config-2 fit per GPU with nothing exchanged (a parameter grid fanned out over the devices).
``--scaling strong|weak`` runs one family only.  Set-up is O(1/N) per rank: a rank draws only its
own rows from the counter-based generator and hands them to the library as CSR; rank 0 alone
draws the global STRUCTURE (no values), colours it and broadcasts order / batch boundaries, which
the other ranks install with spfm_set_schedule_raw.
"""
import argparse
import collections
import json
import os
import resource
import signal
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BASE_N, BASE_D = 1_000_000, 100_000  # what BASELINE.json's metric is quoted on
N_SAMPLES = int(os.environ.get("SPFM_BENCH_N", BASE_N))
N_FEATURES = int(os.environ.get("SPFM_BENCH_D", BASE_D))
RESIZED = (N_SAMPLES, N_FEATURES) != (BASE_N, BASE_D)
NNZ_PER_ROW = 50
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
# profiles/<round>_traffic.json records the library's build tag (hash of its sources,
# spfm_build_tag) it was collected with; `traffic` is only reported when that is the library
# running now
TRAFFIC_FILE = "r04_traffic.json"

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


# ------------------------------------------------------------------------------- launcher
def supervise(procs, deadline_s=None, grace_s=5.0, poll_s=0.1):
    """Wait for the child processes `procs` (the ones this process started -- nothing else is ever
    signalled).  Returns the exit status of the job: 0 when all ranks exit 0; otherwise the status
    of the FIRST rank seen to fail, after the others were terminated (SIGTERM, then SIGKILL after
    `grace_s`); 124 when `deadline_s` passes first.  A SIGTERM / SIGINT delivered to this process
    meanwhile is forwarded the same way (status 128 + signal)."""
    got = {"sig": 0}

    def on_signal(signum, _frame):
        got["sig"] = signum

    old = {}
    if threading.current_thread() is threading.main_thread():
        for sg in (signal.SIGTERM, signal.SIGINT):
            old[sg] = signal.signal(sg, on_signal)

    def stop_all():
        for pr in procs:
            if pr.poll() is None:
                try:
                    pr.terminate()
                except OSError:
                    pass
        t_end = time.time() + grace_s
        for pr in procs:
            try:
                pr.wait(timeout=max(0.0, t_end - time.time()))
            except subprocess.TimeoutExpired:
                pr.kill()
                pr.wait()

    t0 = time.time()
    status = 0
    try:
        while True:
            codes = [pr.poll() for pr in procs]
            bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
            if bad:
                r, c = bad[0]
                log("launcher: rank %d exited with status %d; stopping the other ranks" % (r, c))
                stop_all()
                status = c if c > 0 else 128 - c  # killed by a signal: 128 + signal
                break
            if all(c == 0 for c in codes):
                break
            if got["sig"]:
                log("launcher: signal %d; stopping the ranks" % got["sig"])
                stop_all()
                status = 128 + got["sig"]
                break
            if deadline_s is not None and time.time() - t0 > deadline_s:
                log("launcher: deadline of %.0f s passed; stopping the ranks" % deadline_s)
                stop_all()
                status = 124
                break
            time.sleep(poll_s)
    finally:
        for sg, h in old.items():
            signal.signal(sg, h)
    return status


def launch_ranks(n_ranks, argv, ndev=None, child_cmd=None, deadline_s=None):
    """Start the N ranks as child processes, relay rank 0's JSON line, return the exit status.
    `child_cmd` (tests) replaces `python bench.py argv`."""
    if ndev is None:
        ndev = int(os.environ.get("SPFM_BENCH_DEVICES", 0)) or device_count()
    if ndev < 1:
        log("no GPU visible: cannot start %d ranks" % n_ranks)
        return 1
    if deadline_s is None:
        deadline_s = float(os.environ.get("SPFM_BENCH_DEADLINE", 1700))
    with socket.socket() as sk:  # a free port for the control-plane rendezvous
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    shared = ndev < n_ranks
    if shared:
        log("%d ranks on %d device(s): REHEARSAL -- ranks share a GPU (host-shm communicator, "
            "exchange slabs IPC-mapped on one device)" % (n_ranks, ndev))
    cmd = child_cmd or ([sys.executable, os.path.abspath(__file__)] + list(argv))
    procs, tails, pumps = [], [], []
    out0 = []

    def pump_err(pipe, tail, r):  # relay a rank's stderr, keep its last lines for the post-mortem
        for raw in iter(pipe.readline, b""):
            line = raw.decode(errors="replace")
            tail.append(line)
            sys.stderr.write(line if r == 0 else "[rank %d] %s" % (r, line))
            sys.stderr.flush()
        pipe.close()

    def pump_out(pipe):
        out0.append(pipe.read())
        pipe.close()

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
        pr = subprocess.Popen(cmd, env=env, stderr=subprocess.PIPE,
                              stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL)
        procs.append(pr)
        tails.append(collections.deque(maxlen=25))
        th = threading.Thread(target=pump_err, args=(pr.stderr, tails[-1], r), daemon=True)
        th.start()
        pumps.append(th)
        if r == 0:
            th = threading.Thread(target=pump_out, args=(pr.stdout,), daemon=True)
            th.start()
            pumps.append(th)
    status = supervise(procs, deadline_s=deadline_s)
    for th in pumps:
        th.join(timeout=5)
    if status != 0:
        for r, tail in enumerate(tails):
            log("---- rank %d (exit %s), last lines of stderr:" % (r, procs[r].returncode))
            for line in tail:
                sys.stderr.write("    " + line)
        sys.stderr.flush()
    if out0 and out0[0]:
        sys.stdout.write(out0[0].decode(errors="replace"))
        sys.stdout.flush()
    return status


# ------------------------------------------------------------------------------- workload
class Bench(object):
    """One process = one rank.  Holds the control-plane group and measures workloads."""

    def __init__(self, args):
        self.args = args
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        # rehearsal on a one-GPU box: SPFM_DEVICE=0 SPFM_COMM=shm runs all ranks on device 0 with
        # the engine's host shared-memory exchange instead of RCCL (see spfm_comm_init_shm)
        self.local_rank = int(os.environ.get("SPFM_DEVICE", local_rank))
        if self.world == 1:
            # the concurrent-fits extra starts four fits side by side: the HIP runtime's hardware
            # queues must be configured before the first GPU call of the process
            from sparsepoly_amd import _capi

            _capi.ensure_hw_queues(4)
        import torch

        self.torch = torch
        self.dist = None
        if self.world > 1:
            import datetime

            import torch.distributed as dist

            # torch.distributed only carries control messages (unique id, barrier, max of the
            # timings): gloo, with a finite time-out so that a rank that dies in set-up turns
            # into an error on the others instead of a wait without end
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            torch.cuda.set_device(self.local_rank)
            tmo = float(os.environ.get("SPFM_BENCH_GLOO_TIMEOUT", 600))
            dist.init_process_group("gloo", rank=self.rank, world_size=self.world,
                                    timeout=datetime.timedelta(seconds=tmo))
            self.dist = dist
        self.ndev = int(os.environ.get("SPFM_BENCH_NDEV", self.world))
        self.devices_used = min(self.world, self.ndev)
        self.rehearsal = self.world > 1 and self.devices_used < self.world

    def fence(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    # ---- data ------------------------------------------------------------------------
    def make_data(self, n, d, sharded):
        from sparsepoly_amd import distributed as spdist
        from sparsepoly_amd.synth import make_problem

        t0 = time.time()
        if sharded:
            # O(1/N) per rank: only the own rows, straight from the counter-based generator (CSR)
            lo, hi = spdist.row_block(n, self.rank, self.world)
            X, y = make_problem(n, d, NNZ_PER_ROW, seed=0, row_range=(lo, hi))
            Xc = None
            t = self.torch.tensor([int(X.nnz)], dtype=self.torch.int64)
            self.dist.all_reduce(t)
            nnz = int(t.item())
        else:
            X, y = make_problem(n, d, NNZ_PER_ROW, seed=0)
            Xc = X.tocsc()
            Xc.sort_indices()
            nnz = Xc.nnz
        if self.rank == 0:
            log("data %dx%d nnz=%d: this rank's %d rows generated in %.1fs"
                % (n, d, nnz, X.shape[0], time.time() - t0))
        return dict(n=n, d=d, nnz=nnz, X=X, Xc=Xc, y=y, sharded=sharded,
                    data_s=round(time.time() - t0, 2))

    # ---- engine ----------------------------------------------------------------------
    def make_engine(self, data, cfg, precision, schedule, setup=None):
        from sparsepoly_amd import distributed as spdist
        from sparsepoly_amd.engine import HipEngine
        from sparsepoly_amd.schedule import Schedule

        n, d = data["n"], data["d"]
        K, DEGREE = cfg["k"], cfg["degree"]
        setup = {} if setup is None else setup
        P0 = 0.01 * np.random.RandomState(0).randn(DEGREE - 1, K, d)
        jf0 = np.arange(d, dtype=np.int32)
        eng = HipEngine(self.local_rank, precision)
        for kv in filter(None, os.environ.get("SPFM_OPTS", "").split(",")):  # e.g. prb_groups=32
            key, val = kv.split("=")
            eng.set_option(key, int(val))
        t1 = time.time()
        if data["sharded"]:
            spdist.init_engine_comm(eng)
            spdist.connect_peers(eng)  # persistent passes with the in-kernel xGMI exchange
        eng.set_data(data["X"] if data["sharded"] else data["Xc"], data["y"])
        eng.set_params(P0, np.zeros(d), np.ones(K))
        eng.configure(cfg["solver"], "squared", cfg["reg"], DEGREE)
        eng.init_pred(DEGREE, True, DEGREE == 3)
        setup["engine_s"] = round(time.time() - t1, 2)
        t1 = time.time()
        if data["sharded"]:
            # rank 0 colours the global structure once (the library's own policy for the step
            # width, decided from global inputs); everybody else installs the result
            if self.rank == 0:
                from sparsepoly_amd.synth import make_csr

                S = make_csr(n, d, NNZ_PER_ROW, seed=0, structure_only=True).tocsc()
                S.sort_indices()
                setup["structure_s"] = round(time.time() - t1, 2)
                order = eng.set_schedule(schedule, jf0, S)
                sch = eng.get_schedule(schedule)
                payload = [sch.order, sch.batch_ptr]
                del S
            else:
                payload = [None, None]
            self.dist.broadcast_object_list(payload, src=0)
            if self.rank != 0:
                order = eng.install_schedule(Schedule(payload[0], payload[1], schedule, (n, d)))
        else:
            order = eng.set_schedule(schedule, jf0)
        setup["schedule_s"] = round(time.time() - t1, 2)
        return eng, order, P0

    @staticmethod
    def iteration(eng, cfg, comps=None):
        """One pass of the reference's iteration body; `comps` limits the pcd component loop."""
        K, DEGREE = cfg["k"], cfg["degree"]
        ic = np.arange(K if comps is None else comps, dtype=np.int32)
        v = eng.cd_linear_epoch(cfg["alpha"])
        for deg in list(range(2, DEGREE)) + [DEGREE]:
            o = DEGREE - deg if deg != DEGREE else 0
            if cfg["solver"] == "pcd":
                v += eng.pcd_epoch(o, deg, cfg["beta"], cfg["gamma"], ETA0, ic)
            else:
                v += eng.pbcd_epoch(o, deg, cfg["beta"], cfg["gamma"], ETA0)
        return v

    # ---- one measured workload ---------------------------------------------------------
    def measure(self, data, cfg_id, precision, schedule, steps, warmup, want_y0=False):
        """warmup untimed + `steps` timed iterations (barrier + synchronize on both sides, MAX
        over ranks), then one profiled epoch for the roofline object.  Returns a dict."""
        torch, dist = self.torch, self.dist if data["sharded"] else None
        cfg = CONFIGS[cfg_id]
        K, DEGREE = cfg["k"], cfg["degree"]
        n, nnz = data["n"], data["nnz"]
        setup = {"data_s": data["data_s"]}
        eng, order, P0 = self.make_engine(data, cfg, precision, schedule, setup)
        y_pred0 = eng.get_y_pred() if want_y0 else None
        viols = []

        def warm_up(e):
            """The untimed steps; with several ranks all of them learn whether any rank failed."""
            ok, why = 1, ""
            try:
                for _ in range(warmup):
                    viols.append(self.iteration(e, cfg))
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
            # the in-kernel peer exchange did not come up on this node (the warm-up aborted on
            # some rank): every rank rebuilds its engine with the per-step collective instead and
            # the line says so
            if os.environ.get("SPFM_PEER", "1") == "0":
                raise RuntimeError("warm-up failed: " + why)
            log("rank %d: warm-up failed with the in-kernel peer exchange (%s); falling back to "
                "the per-step collective" % (self.rank, why))
            try:
                eng.close()
            except Exception:
                pass
            os.environ["SPFM_PEER"] = "0"
            del viols[:]
            eng, order, P0 = self.make_engine(data, cfg, precision, schedule, setup)
            ok, why = warm_up(eng)
            if not ok:
                raise RuntimeError("warm-up failed: " + why)
        n_batches = eng.n_batches
        if self.rank == 0:
            log("config %d, schedule '%s': %d dependent steps per sweep (set-up %.1fs)"
                % (cfg_id, schedule, n_batches, setup["engine_s"] + setup["schedule_s"]))

        def fence():
            torch.cuda.synchronize()
            if dist is not None:
                dist.barrier()
                torch.cuda.synchronize()

        fence()
        t_start = time.perf_counter()
        for _ in range(steps):
            viols.append(self.iteration(eng, cfg))
        fence()
        elapsed = time.perf_counter() - t_start
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        ms_per_step = 1e3 * elapsed / max(steps, 1)
        loss_after = eng.loss_sum()
        P_end, _ = eng.get_params()
        nnz_frac_P = float((P_end != 0).mean())

        # ---- roofline of the dominant kernel (profiled epoch, outside the timed region): HIP
        # events on the engine's stream around every launch of that kernel
        tsz = 4 if precision == "f32" else 8
        ic = np.arange(K, dtype=np.int32)
        roof = None
        eng.profile_reset()
        eng.profile_enable(True)
        if cfg["solver"] == "pcd":
            persistent = bool(eng.get_option("persistent_active"))
            eng.pcd_epoch(0, DEGREE, cfg["beta"], cfg["gamma"], ETA0,
                          ic[:4] if persistent else ic[:2])
            which = 0
            if persistent:
                # one launch = one component pass; per column entry: row 4 + value T + (yhat, y)
                # read 2T + yhat write T + A[i,1..m-1] read and write 2T(m-1)
                # (the wide pass -- steps of more than 64 columns, DESIGN 3d -- moves the same)
                kname = (("pcdwe_kernel" if eng.get_option("wide_ep") else "pcdw_kernel")
                         if eng.get_option("wide_active") else "pcd_prb_kernel")
                bytes_per_nnz = 4 + 4 * tsz + 2 * tsz * (DEGREE - 1)
            else:
                kname, bytes_per_nnz = "pcd_grad_kernel", 4 + 3 * tsz + tsz * (DEGREE - 1)
        else:
            eng.pbcd_epoch(0, DEGREE, cfg["beta"], cfg["gamma"], ETA0)
            persistent = bool(eng.get_option("pbprb_active"))
            which = 2
            # one launch = one pbcd epoch; per entry: row 4 + value T + (yhat, y) 2T + yhat write
            # T + A[i,1..m-1,:] read and write 2T(m-1)k
            kname = "pbcd_prb_kernel" if persistent else "pbcd_grad_kernel"
            bytes_per_nnz = (4 + 4 * tsz + 2 * tsz * (DEGREE - 1) * K) if persistent else \
                (4 + 3 * tsz + tsz * (DEGREE - 1) * K)
        eng.profile_enable(False)
        g_ms, g_launch, g_nnz = eng.profile_get(which)
        if g_launch > 0 and g_ms > 0:
            from sparsepoly_amd import _capi

            tag = _capi.build_tag()
            avg_us = 1e3 * g_ms / g_launch
            bytes_per_launch = bytes_per_nnz * g_nnz / g_launch
            achieved = bytes_per_launch / (avg_us * 1e-6) / 1e9
            traffic, traffic_src = None, "not collected for this build of the library (%s)" % tag
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", TRAFFIC_FILE)))
                ent = tj.get("config%d" % cfg_id, {}).get(kname)
                if (ent and tj.get("engine_tag") == tag and not RESIZED and self.world == 1
                        and precision == "f32" and schedule == "colored"):
                    traffic = round(1024.0 * (ent["fetch_kb_per_launch"]
                                              + ent["write_kb_per_launch"]), 1)
                    traffic_src = "profiles/%s (rocprofv3 --pmc passes, library build %s)" \
                        % (TRAFFIC_FILE, tag)
            except Exception:
                pass
            roof = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2),
                    "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "traffic": traffic, "traffic_source": traffic_src,
                    "avg_launch_us": round(avg_us, 3),
                    "alg_bytes_per_launch": round(bytes_per_launch, 1),
                    "launches_timed": int(g_launch)}
            if persistent:
                roof["dependent_steps_per_launch"] = n_batches
                roof["us_per_dependent_step_in_kernel"] = round(avg_us / n_batches, 3)
                if cfg["solver"] == "pcd":
                    # 0 = row state in global memory, 1 / 2 = row block in LDS
                    # (wide pass: 2 = the blocks' first rows in LDS, the others in global memory)
                    roof["row_block_in_lds"] = int(eng.get_option(
                        "wide_lds_active" if eng.get_option("wide_active") else "prb_lds_active"))
        # what really ran: the ranks of the engine's communicator, the exchange it used, whether
        # a persistent pass had to be redone on the multi-kernel engine; per-rank set-up cost
        res = dict(
            cfg_id=cfg_id, n=n, d=data["d"], nnz=nnz, schedule=schedule, precision=precision,
            ms_per_step=ms_per_step, elapsed=elapsed, steps=steps, viols=viols,
            loss_after=float(loss_after), nnz_frac_P=nnz_frac_P, n_batches=n_batches, roof=roof,
            ranks_seen=int(eng.get_option("n_ranks")),
            peer_exchange=bool(eng.get_option("peer_ready")),
            fallbacks=int(eng.get_option("persistent_fallbacks")),
            order=order, P0=P0, y_pred0=y_pred0)
        setup["peak_rss_gb"] = round(resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1048576.0, 2)
        setup["rank"] = self.rank
        setups = [setup]
        if dist is not None:
            setups = [None] * self.world
            dist.all_gather_object(setups, setup)
        res["setups"] = setups
        b_alg = alg_bytes(cfg, n, nnz, tsz)
        res["b_alg"] = b_alg
        res["iter_gbs"] = b_alg / (ms_per_step * 1e-3) / 1e9
        res["steps_per_iter"] = (1 + (K * (DEGREE - 1) if cfg["solver"] == "pcd"
                                      else (DEGREE - 1))) * n_batches
        eng.close()
        return res

    # ---- summaries -------------------------------------------------------------------
    @staticmethod
    def workload_string(cfg, n, d, nnz):
        s = "%s: degree=%d n_components=%d regularizer=%s solver=%s fit_linear=True " \
            "fit_lower=explicit on %dx%d CSR nnz=%d (~50/row)" \
            % (cfg["name"], cfg["degree"], cfg["k"], cfg["reg"], cfg["solver"], n, d, nnz)
        if RESIZED:
            s += " -- RESIZED by SPFM_BENCH_N/D: not the %dx%d matrix BASELINE.json's metric " \
                 "is quoted on" % (BASE_N, BASE_D)
        return s

    def brief(self, res):
        """What `other_configs` / `weak` carry for a measured workload."""
        cfg = CONFIGS[res["cfg_id"]]
        return {"workload": self.workload_string(cfg, res["n"], res["d"], res["nnz"]),
                "metric": "pcd_epochs_per_sec" if cfg["solver"] == "pcd" else "pbcd_epochs_per_sec",
                "ms_per_iteration": round(res["ms_per_step"], 3),
                "iterations_per_s": round(1e3 / res["ms_per_step"], 4),
                "iterations_timed": res["steps"],
                "schedule": res["schedule"], "dependent_steps_per_sweep": res["n_batches"],
                "us_per_dependent_step": round(1e3 * res["ms_per_step"] / res["steps_per_iter"], 3),
                "alpha": cfg["alpha"], "beta": cfg["beta"], "gamma": cfg["gamma"],
                "iteration_alg_GBs": round(res["iter_gbs"], 2),
                "iteration_alg_frac_of_hbm_peak": round(res["iter_gbs"] / HBM_PEAK_GBS, 5),
                "roofline": res["roof"],
                "viol": [round(float(v), 6) for v in res["viols"]],
                "sum_loss_after": round(res["loss_after"], 6),
                "nonzero_frac_P_after": round(res["nnz_frac_P"], 4),
                "persistent_fallbacks": res["fallbacks"]}


def cpu_baseline(data, res):
    """The oracle (test infrastructure, f64, one thread) on one full iteration of the workload
    `res` measured, in the same column order."""
    from oracle import oracle as orc

    cfg = CONFIGS[res["cfg_id"]]
    K, DEGREE = cfg["k"], cfg["degree"]
    n, d, Xc, y = data["n"], data["d"], data["Xc"], data["y"]
    orc.build()
    ds = orc.CSC(Xc)
    Pc = np.ascontiguousarray(res["P0"].copy())
    wc = np.zeros(d)
    yp = np.ascontiguousarray(res["y_pred0"])
    cn = np.asarray(Xc.multiply(Xc).sum(axis=0)).ravel()
    regc = orc.Regularizer(cfg["reg"])
    jf = np.ascontiguousarray(res["order"])
    lams = np.ones(K)
    ic = np.arange(K, dtype=np.int32)
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
    return {"value": round(1.0 / cpu_s, 6), "unit": "epochs/s", "cores": 1, "kind": "port",
            "sample": "one full iteration (cd_linear + every component pass / block epoch) of "
                      "the same %dx%d workload in the same column order: %.1f s" % (n, d, cpu_s),
            "host_cpus": os.cpu_count()}


def single_gpu_extras(b, data, head, args):
    """other_configs, exact_schedule, f64, concurrent_fits: N = 1 only, outside the timed region."""
    torch = b.torch
    extras = {}
    skip = os.environ.get("SPFM_BENCH_SKIP", "").split(",")  # diagnostics: leave extras out
    cfg_id = head["cfg_id"]
    cfg = CONFIGS[cfg_id]
    # ---- the other single-GPU BASELINE configurations on the same matrix
    if "other" not in skip:
        others = {}
        for oc in sorted(CONFIGS):
            if oc == cfg_id:
                continue
            r = b.measure(data, oc, args.precision, args.schedule, max(3, args.steps), 1)
            others["config%d" % oc] = b.brief(r)
            log("config %d: %.1f ms per iteration" % (oc, r["ms_per_step"]))
        extras["other_configs"] = others
    # ---- the estimators' default schedule ('exact': the reference's own order): a WHOLE iteration
    if "exact" not in skip and args.schedule != "exact":
        ex = {}
        for oc in sorted(CONFIGS):
            if oc != cfg_id and "other" in skip:
                continue
            t1 = time.time()
            eng, _, _ = b.make_engine(data, CONFIGS[oc], args.precision, "exact")
            ts = time.time() - t1
            nb2 = eng.n_batches
            b.iteration(eng, CONFIGS[oc], comps=1 if CONFIGS[oc]["solver"] == "pcd" else None)
            torch.cuda.synchronize()  # entry streams built, kernels loaded
            t1 = time.perf_counter()
            b.iteration(eng, CONFIGS[oc])
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t1)
            ex["config%d" % oc] = {
                "dependent_steps_per_sweep": nb2, "schedule_build_s": round(ts, 2),
                # pcd (degree 2, 3) and cd_linear run the reference order as merged steps whose
                # shared rows the chains replay; 0 = strict steps
                "merged_steps_per_sweep": int(eng.get_option("relax_steps")),
                "ms_per_iteration": round(ms, 1), "measured": "one whole iteration, timed",
                "persistent_fallbacks": int(eng.get_option("persistent_fallbacks"))}
            eng.close()
        own = ex.pop("config%d" % cfg_id)
        own["is_estimator_default"] = True
        if ex:
            own["other_configs"] = ex
        extras["exact_schedule"] = own
    if args.precision == "f32" and "f64" not in skip:
        e3, _, _ = b.make_engine(data, cfg, "f64", args.schedule)
        b.iteration(e3, cfg)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        b.iteration(e3, cfg)
        torch.cuda.synchronize()
        extras["f64"] = {"ms_per_iteration": round(1e3 * (time.perf_counter() - t1), 2),
                         "schedule": args.schedule}
        e3.close()
    # ---- four independent fits of the same workload at once (a regularisation path): one handle,
    # stream and host thread per fit, their persistent passes side by side on disjoint CUs
    if cfg["solver"] == "pcd" and "concurrent" not in skip:
        from sparsepoly_amd.engine import co_tenancy, hw_queue_report

        F, its = 4, 3
        with co_tenancy(F):
            engs = [b.make_engine(data, cfg, args.precision, args.schedule)[0] for _ in range(F)]
        for e in engs:
            b.iteration(e, cfg)  # builds the entry streams, warms up
        torch.cuda.synchronize()
        bar = threading.Barrier(F + 1)
        spent = [0.0] * F

        def fit_loop(f):
            bar.wait()
            t = time.perf_counter()
            for _ in range(its):
                b.iteration(engs[f], cfg)
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
        b_alg = head["b_alg"]
        extras["concurrent_fits"] = {
            "fits": F, "iterations_each": its,
            "ms_per_iteration_per_fit": [round(1e3 * t / its, 1) for t in spent],
            "aggregate_epochs_per_s": round(F * its / wall, 3),
            "aggregate_GBps": round(F * its * b_alg / wall / 1e9, 1),
            "aggregate_frac_of_hbm_peak": round(F * its * b_alg / wall / 1e9 / HBM_PEAK_GBS, 4),
            "persistent_fallbacks": fb,
            "hw_queues": hw_queue_report(),
            "note": "independent fits (different models, same matrix); each equals its solo run "
                    "bit for bit (tests/test_hip_concurrent.py); `value` above is ONE fit"}
    return extras


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip other_configs, exact-schedule, f64 and concurrent-fits measurements")
    ap.add_argument("--precision", default="f32", choices=["f32", "f64"])
    ap.add_argument("--schedule", default="colored", choices=["colored", "exact"])
    ap.add_argument("--scaling", default="both", choices=["both", "weak", "strong"],
                    help="N > 1: strong = the 1M x 100k matrix of the config sharded over the "
                         "ranks (`value` of the default line); weak = N times the rows AND "
                         "columns (rows sharded; same dependent steps per sweep, fixed work per "
                         "GPU and step; the family that ends in BASELINE configs[4] = 10M x 1M "
                         "on 8 GPUs); both (default) = strong as `value`, weak as an extra")
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

    b = Bench(args)
    world, rank = b.world, b.rank
    if args.gpus != world and rank == 0:
        log("--gpus %d but WORLD_SIZE=%d: the line reports the %d rank(s) that really ran"
            % (args.gpus, world, world))
    from sparsepoly_amd import _capi

    ENGINE_TAG = _capi.build_tag()  # hash of the library's sources, written in at build time
    cfg = CONFIGS[args.config]

    extras = {}
    cpu = None
    if world == 1:
        data = b.make_data(N_SAMPLES, N_FEATURES, sharded=False)
        head = b.measure(data, args.config, args.precision, args.schedule, args.steps, args.warmup,
                         want_y0=not args.no_cpu_baseline)
        value = args.steps / head["elapsed"]
        scaling = "weak"
        if not args.no_extras:
            extras = single_gpu_extras(b, data, head, args)
        if not args.no_cpu_baseline:
            cpu = cpu_baseline(data, head)
        note = "single GPU: " + ("the configuration BASELINE.json's metric is quoted on"
                                 if not RESIZED else
                                 "a RESIZED workload (SPFM_BENCH_N/D), not BASELINE's configuration")
    else:
        # the multiplier of the weak family: N shares of the config -- but only when every rank
        # has a GPU of its own; ranks that time-slice one device are a rehearsal of the code
        # path, not N GPUs' worth of work
        fams = ["strong", "weak"] if args.scaling == "both" else [args.scaling]
        results = {}
        for fam in fams:
            sc = world if fam == "weak" else 1
            data = b.make_data(N_SAMPLES * sc, N_FEATURES * sc, sharded=True)
            results[fam] = b.measure(data, args.config, args.precision, args.schedule, args.steps,
                                     args.warmup)
            del data
        head_fam = fams[0]
        head = results[head_fam]
        mult = 1 if b.rehearsal else world
        if head_fam == "weak":
            value = mult * args.steps / head["elapsed"]
            note = ("weak scaling: %d x the rows and columns of the config (each rank owns a "
                    "%d-row shard); value = %d x epochs/s of the %dx%d problem"
                    % (world, N_SAMPLES, mult, head["n"], head["d"]))
        else:
            value = args.steps / head["elapsed"]
            note = ("strong scaling: the config's %dx%d matrix sharded by rows over %d ranks -- "
                    "the same epoch on N GPUs" % (head["n"], head["d"], world))
        scaling = head_fam
        if "weak" in results and head_fam != "weak":
            w = results["weak"]
            wb = b.brief(w)
            wb["value"] = round(mult * args.steps / w["elapsed"], 4)
            wb["unit"] = "epochs/s"
            wb["note"] = ("weak scaling: %d x the rows and columns of the config (each rank owns a "
                          "%d-row shard of the %dx%d problem); value = %d x iterations/s of that "
                          "problem, i.e. config-sized shares of work per second"
                          % (world, N_SAMPLES, w["n"], w["d"], mult))
            wb["peer_exchange"] = w["peer_exchange"]
            wb["setup_per_rank"] = w["setups"]
            extras["weak"] = wb
        if b.rehearsal:
            note += "; REHEARSAL: %d ranks time-slice %d device(s)" % (world, b.devices_used)
        # ---- one independent fit per GPU: nothing is exchanged (a parameter grid fanned out over
        # the devices, sparsepoly_amd.concurrent.fit_concurrently(devices=...))
        if not b.rehearsal and not args.no_extras and args.scaling == "both":
            data = b.make_data(N_SAMPLES, N_FEATURES, sharded=False)
            r = b.measure(data, args.config, args.precision, args.schedule, args.steps, 1)
            del data
            t = b.torch.tensor([r["elapsed"]], dtype=b.torch.float64)
            b.dist.all_reduce(t, op=b.dist.ReduceOp.MAX)
            extras["independent_fits"] = {
                "fits": world, "one_per_gpu": True,
                "aggregate_epochs_per_s": round(world * args.steps / float(t.item()), 4),
                "ms_per_iteration_slowest": round(1e3 * float(t.item()) / args.steps, 3),
                "note": "every rank runs the whole config on its own GPU (different models of a "
                        "grid); no exchange, so this scales with N by construction"}

    if rank == 0:
        par = "single GPU"
        if world > 1:
            par = ("rows sharded over %d ranks (%s), per-step exchange: %s; control plane gloo, "
                   "communicator %s"
                   % (head["ranks_seen"],
                      "one process per GPU" if not b.rehearsal
                      else "REHEARSAL: %d processes on %d device(s)" % (world, b.devices_used),
                      "in-kernel peer-mapped slabs (no collective)"
                      if head["peer_exchange"] else "one all-reduce per dependent step",
                      "host-shm (ranks share a device)"
                      if os.environ.get("SPFM_COMM") == "shm" else "RCCL"))
        out = {
            "metric": "pcd_epochs_per_sec" if cfg["solver"] == "pcd" else "pbcd_epochs_per_sec",
            "value": round(value, 4),
            "unit": "epochs/s",
            # devices that really ran ranks; the rank count is `config.ranks`
            "n_gpus": b.devices_used if world > 1 else 1,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(head["ms_per_step"], 3),
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": "f32" if args.precision == "f32" else "f64",
            "data": "synthetic",
            "rehearsal": bool(b.rehearsal),
            "config": {"workload": b.workload_string(cfg, head["n"], head["d"], head["nnz"]),
                       "scaling_note": note,
                       "schedule": args.schedule,
                       "dependent_steps_per_sweep": head["n_batches"],
                       "alpha": cfg["alpha"], "beta": cfg["beta"], "gamma": cfg["gamma"],
                       "parallelism": par,
                       "ranks": head["ranks_seen"],
                       "devices_used": b.devices_used if world > 1 else 1,
                       "launcher": os.environ.get("SPFM_BENCH_LAUNCHER", "external"),
                       "persistent_fallbacks": head["fallbacks"],
                       "setup_per_rank": head["setups"]},
            "roofline": head["roof"],
            "cpu_baseline": cpu,
            "iteration_alg_GBs": round(head["iter_gbs"], 2),
            "iteration_alg_frac_of_hbm_peak": round(head["iter_gbs"] / HBM_PEAK_GBS, 5),
            "us_per_dependent_step": round(1e3 * head["ms_per_step"] / head["steps_per_iter"], 3),
            "viol": [round(float(v), 6) for v in head["viols"]],
            "sum_loss_after": round(head["loss_after"], 6),
            "nonzero_frac_P_after": round(head["nnz_frac_P"], 4),
            "engine_tag": ENGINE_TAG,
        }
        out.update(extras)
        print(json.dumps(out), file=json_out, flush=True)
    if b.dist is not None:
        b.dist.barrier()
        b.dist.destroy_process_group()


if __name__ == "__main__":
    main()
