"""The multi-GPU protocol (contiguous row shards, per-step exchange of the
column partial sums, replicated chain) executed by the real engine: two processes share the one
GPU of the test box.  Two exchange paths: the multi-kernel engine with a per-step all-reduce
through the host shared-memory communicator (`spfm_comm_init_shm`; RCCL refuses two ranks on one
device), and the persistent passes with the in-kernel exchange through IPC-mapped slabs
(`spfm_peer_alloc` / `spfm_peer_connect`) -- two persistent kernels co-resident on the GPU,
each writing its per-step totals into the other's slab.  The sharded result must equal
the single-process multi-kernel engine and the oracle, and be bit-identical on both ranks.
Needs a real MI355X: ``pytest -m gpu``."""
import multiprocessing as mp
import os
import sys

import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {
    # tag: solver, regularizer, degree, k, loss, beta, gamma
    "pcd_sql12": ("pcd", "squaredl12", 2, 6, "squared", 10.0, 1e-3),
    "pcd_ti3": ("pcd", "omegati", 3, 4, "logistic", 10.0, 1e-4),
    "pbcd_cs": ("pbcd", "omegacs", 2, 6, "squared", 1.0, 1e-2),
    # sparse and wide: colour classes of hundreds of columns -> the wide persistent passes
    "pcd_wide": ("pcd", "squaredl12", 2, 4, "squared", 10.0, 1e-3),
}


def _problem(loss, case=None):
    sys.path.insert(0, ROOT)
    from sparsepoly_amd.synth import make_problem

    if case == "pcd_wide":
        X, y = make_problem(6000, 4000, 4, seed=3)
        return sp.csr_matrix(X), y
    X, y = make_problem(6000, 500, 20, seed=3)
    if loss != "squared":
        y = np.where(y > np.median(y), 1.0, -1.0)
    return sp.csr_matrix(X), y


def _run(case, world, rank, shm_name, precision, engine="multi_kernel", hq=None):
    """One rank of the sharded run (world == 1: the whole problem, no communicator).
    engine = 'multi_kernel': per-step all-reduce through the host communicator;
    'persistent_peer': the persistent passes with the in-kernel exchange through IPC-mapped
    slabs (hq = per-rank queues that carry the 64-byte handles between the two processes)."""
    sys.path.insert(0, ROOT)
    from sparsepoly_amd.engine import HipEngine, canonical_csc

    solver, reg, degree, k, loss, beta, gamma = CASES[case]
    X, y = _problem(loss, case)
    n, d = X.shape
    lo, hi = (n * rank) // world, (n * (rank + 1)) // world
    Xg = canonical_csc(X)
    eng = HipEngine(0, precision)
    if engine == "multi_kernel":
        eng.set_option("persistent", 0)   # the multi-kernel engine in both runs
    elif engine == "persistent_peer_128":
        # persistent pbcd pass with 128 row workgroups per rank: two co-resident kernels fill the
        # 256 CUs; half of the workgroups own a slot (cross-GPU stage in the owners)
        eng.set_option("pbprb_groups", 128)
        eng.set_option("pcdw_groups", 64)
    else:
        eng.set_option("pbprb_groups", 64)  # two co-resident persistent kernels: 2 x 64 CUs
        eng.set_option("pcdw_groups", 64)
    if world > 1:
        eng.comm_init_shm(shm_name, world, rank)
        if engine.startswith("persistent_peer"):
            mine = eng.peer_alloc()
            for r in range(world):
                if r != rank:
                    hq[r].put((rank, mine))
            handles = {rank: mine}
            while len(handles) < world:
                r, h = hq[rank].get(timeout=120)
                handles[r] = h
            eng.peer_connect(world, rank, [handles[r] for r in range(world)])
    eng.set_data(canonical_csc(X[lo:hi]), y[lo:hi])
    P0 = 0.01 * np.random.RandomState(0).randn(degree - 1, k, d)
    eng.set_params(P0, np.zeros(d), np.ones(k))
    eng.configure(solver, loss, reg, degree)
    eng.init_pred(degree, True, degree == 3)
    order = eng.set_schedule("colored", np.arange(d, dtype=np.int32), Xg)  # GLOBAL conflicts
    ic = np.arange(k, dtype=np.int32)
    viol, losses = [], []
    for _ in range(2):
        v = eng.cd_linear_epoch(0.1)
        for deg in list(range(2, degree)) + [degree]:
            o = degree - deg if deg != degree else 0
            v += (eng.pcd_epoch(o, deg, beta, gamma, 1.0, ic) if solver == "pcd"
                  else eng.pbcd_epoch(o, deg, beta, gamma, 1.0))
        viol.append(v)
        losses.append(eng.loss_sum())       # all-reduced over the shards
    P, w = eng.get_params()
    yp = eng.get_y_pred()
    active = (eng.get_option("persistent_active"), eng.get_option("pbprb_active"))
    extra = (eng.get_option("pbprb_groups"), eng.get_option("persistent_fallbacks"))
    eng.close()
    return dict(P=P, w=w, viol=np.array(viol), loss=np.array(losses), y_pred=yp, order=order,
                rows=(lo, hi), active=active, extra=extra)


def _worker(case, world, rank, shm_name, precision, q, engine, hq):
    try:
        q.put((rank, _run(case, world, rank, shm_name, precision, engine, hq)))
    except Exception as e:  # surface the failure in the parent
        q.put((rank, repr(e)))


@pytest.mark.parametrize("case,engine",
                         [(c, e) for c in sorted(CASES) for e in ("multi_kernel", "persistent_peer")]
                         + [("pbcd_cs", "persistent_peer_128")])
def test_two_row_shards_on_one_gpu(oracle, case, engine):
    solver, reg, degree, k, loss, beta, gamma = CASES[case]
    shm_name = "/spfm_test_%d_%s_%s" % (os.getpid(), case, engine)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    hq = [ctx.Queue(), ctx.Queue()]
    procs = [ctx.Process(target=_worker, args=(case, 2, r, shm_name, "f64", q, engine, hq))
             for r in (0, 1)]
    for p in procs:
        p.start()
    got = {}
    try:
        for _ in procs:
            rank, res = q.get(timeout=240)
            got[rank] = res
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
        try:
            os.unlink("/dev/shm" + shm_name)
        except OSError:
            pass
    for r in (0, 1):
        assert isinstance(got[r], dict), got[r]
    a, b = got[0], got[1]
    # replicated state is bit-identical on both ranks
    assert np.array_equal(a["P"], b["P"]) and np.array_equal(a["w"], b["w"])
    assert np.array_equal(a["viol"], b["viol"]) and np.array_equal(a["loss"], b["loss"])
    assert np.array_equal(a["order"], b["order"])
    # equals the unsharded engine (summation order of the partial sums differs: 1e-10)
    one = _run(case, 1, 0, None, "f64", engine)
    if engine.startswith("persistent_peer"):  # the persistent passes really ran, with two ranks
        assert a["active"] == ((1, 0) if solver == "pcd" else (1, 1)), a["active"]
        assert a["extra"][1] == 0, a["extra"]       # no fallback to the multi-kernel engine
    assert np.array_equal(one["order"], a["order"])
    np.testing.assert_allclose(a["P"], one["P"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(a["w"], one["w"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(a["viol"], one["viol"], rtol=1e-10)
    np.testing.assert_allclose(a["loss"], one["loss"], rtol=1e-10)
    yp = np.concatenate([a["y_pred"], b["y_pred"]])
    np.testing.assert_allclose(yp, one["y_pred"], rtol=0, atol=1e-9)
    assert a["rows"] == (0, 3000) and b["rows"] == (3000, 6000)
    # ... and the oracle replaying the coloured order
    X, y = _problem(loss, case)
    fm = oracle.OracleFM(degree=degree, loss=loss, n_components=k, solver=solver, regularizer=reg,
                         alpha=0.1, beta=beta, gamma=gamma, tol=0, max_iter=2,
                         feature_order=a["order"])
    fm.fit(X, y, P_init=0.01 * np.random.RandomState(0).randn(degree - 1, k, X.shape[1]),
           lams_init=np.ones(k))
    np.testing.assert_allclose(a["P"], fm.P_, rtol=0, atol=1e-8)
    np.testing.assert_allclose(a["viol"], [h[0] for h in fm.history], rtol=1e-9)


def test_two_row_shards_wide_pass_float_rows_in_lds():
    """The per-rank shape of BASELINE configs[4] on several GPUs: the wide pass in its
    entry-parallel form with the shard's rows in LDS (float storage, squared loss) and the
    cross-GPU stage of the vslot owners through the peer-mapped slabs.  Both ranks bit-identical;
    equal to the one-rank float run up to the order of the partial sums and float rounding."""
    case, engine = "pcd_wide", "persistent_peer"
    shm_name = "/spfm_test_%d_wide_f32" % os.getpid()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    hq = [ctx.Queue(), ctx.Queue()]
    procs = [ctx.Process(target=_worker, args=(case, 2, r, shm_name, "f32", q, engine, hq))
             for r in (0, 1)]
    for p in procs:
        p.start()
    got = {}
    try:
        for _ in procs:
            rank, res = q.get(timeout=240)
            got[rank] = res
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
        try:
            os.unlink("/dev/shm" + shm_name)
        except OSError:
            pass
    for r in (0, 1):
        assert isinstance(got[r], dict), got[r]
    a, b = got[0], got[1]
    assert np.array_equal(a["P"], b["P"]) and np.array_equal(a["w"], b["w"])
    assert np.array_equal(a["viol"], b["viol"]) and np.array_equal(a["order"], b["order"])
    assert a["active"][0] == 1 and a["extra"][1] == 0, (a["active"], a["extra"])
    one = _run(case, 1, 0, None, "f32", engine)
    assert np.array_equal(one["order"], a["order"])
    np.testing.assert_allclose(a["P"], one["P"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(a["w"], one["w"], rtol=0, atol=2e-5)
    np.testing.assert_allclose(a["viol"], one["viol"], rtol=2e-5)
    yp = np.concatenate([a["y_pred"], b["y_pred"]])
    np.testing.assert_allclose(yp, one["y_pred"], rtol=0, atol=2e-4)


def _est_worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        import warnings

        import torch.distributed as dist

        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SPFM_COMM="shm")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from sparsepoly_amd import SparseFactorizationMachineClassifier

        X, y = _problem("logistic")
        est = SparseFactorizationMachineClassifier(
            degree=2, loss="logistic", n_components=5, solver="pcd", regularizer="squaredl12",
            beta=10.0, gamma=1e-3, max_iter=2, tol=0, random_state=0, schedule="colored",
            precision="f64", device=0, distributed=True)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            est.fit(X, y)
        q.put((rank, dict(P=est.P_, w=est.w_, n_iter=est.n_iter_)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        q.put((rank, repr(e)))


def test_estimator_distributed_flag_two_ranks_one_gpu():
    """The estimator's distributed=True path (torch.distributed for rank/world and the
    bootstrap message, engine communicator for the data path) with two gloo ranks."""
    import socket
    import warnings

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_est_worker, args=(r, 2, port, q)) for r in (0, 1)]
    for p in procs:
        p.start()
    got = {}
    try:
        for _ in procs:
            rank, res = q.get(timeout=240)
            got[rank] = res
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    for r in (0, 1):
        assert isinstance(got[r], dict), got[r]
    assert np.array_equal(got[0]["P"], got[1]["P"]) and np.array_equal(got[0]["w"], got[1]["w"])
    sys.path.insert(0, ROOT)
    from sparsepoly_amd import SparseFactorizationMachineClassifier

    X, y = _problem("logistic")
    one = SparseFactorizationMachineClassifier(
        degree=2, loss="logistic", n_components=5, solver="pcd", regularizer="squaredl12",
        beta=10.0, gamma=1e-3, max_iter=2, tol=0, random_state=0, schedule="colored",
        precision="f64", device=0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        one.fit(X, y)
    np.testing.assert_allclose(got[0]["P"], one.P_, rtol=0, atol=1e-10)
    np.testing.assert_allclose(got[0]["w"], one.w_, rtol=0, atol=1e-10)
    assert got[0]["n_iter"] == one.n_iter_


# ------------------------------------------------------------------ psgd, rows sharded
PSGD_CASES = {
    # tag: regularizer, degree, k, loss, batch_size, fit_lower (explicit: two orders of P)
    "l1_d2": ("l1", 2, 6, "squared", 64, None),
    "sql12_d3": ("squaredl12", 3, 4, "logistic", 100, "explicit"),
    "l21_ragged": ("l21", 2, 5, "squared", 4999, None),   # a short last minibatch (1001 rows)
    "sql21_tiny": ("squaredl21", 2, 4, "squared", 3, None),  # minibatches with no row of a rank
}


def _run_psgd(case, world, rank, shm_name):
    """One rank of a sharded psgd run (spfm_psgd_epoch_sharded): every rank gets the global
    visiting order, forms the gradient of its own rows of each minibatch; the gradients are
    all-reduced through the host communicator.  world == 1: spfm_psgd_epoch."""
    sys.path.insert(0, ROOT)
    from sparsepoly_amd.engine import HipEngine, canonical_csc

    reg, degree, k, loss, bs, fit_lower = PSGD_CASES[case]
    X, y = _problem(loss)
    n, d = X.shape
    lo, hi = (n * rank) // world, (n * (rank + 1)) // world
    eng = HipEngine(0, "f64")
    if world > 1:
        eng.comm_init_shm(shm_name, world, rank)
    eng.set_data(canonical_csc(X[lo:hi]), y[lo:hi])
    n_orders = degree - 1 if fit_lower == "explicit" else 1
    P0 = 0.01 * np.random.RandomState(0).randn(n_orders, k, d)
    eng.set_params(P0, np.zeros(d), np.ones(k))
    eng.configure("psgd", loss, reg, degree)
    rng = np.random.RandomState(5)
    order = np.arange(n, dtype=np.int32)
    it, losses = 1, []
    for _ in range(3):
        rng.shuffle(order)
        sl, it = eng.psgd_epoch(degree, 1e-3, 1e-2, 1e-3, 0.05, "optimal", 1.0, bs, order, True, it,
                                row_lo=lo if world > 1 else None)
        losses.append(sl)
    P, w = eng.get_params()
    eng.close()
    return dict(P=P, w=w, loss=np.array(losses), it=it)


def _psgd_worker(case, world, rank, shm_name, q):
    try:
        q.put((rank, _run_psgd(case, world, rank, shm_name)))
    except Exception as e:
        q.put((rank, repr(e)))


@pytest.mark.parametrize("case", sorted(PSGD_CASES))
def test_psgd_two_row_shards_on_one_gpu(case):
    """Data-parallel psgd (the reference's minibatch loop, psgd.py:125-199, with the minibatch
    gradient summed over the ranks): replicated parameters bit-identical on both ranks, equal to
    the one-rank run up to the order of the gradient sums, and to the oracle."""
    reg, degree, k, loss, bs, fit_lower = PSGD_CASES[case]
    shm_name = "/spfm_test_%d_psgd_%s" % (os.getpid(), case)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_psgd_worker, args=(case, 2, r, shm_name, q)) for r in (0, 1)]
    for p in procs:
        p.start()
    got = {}
    try:
        for _ in procs:
            rank, res = q.get(timeout=240)
            got[rank] = res
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
        try:
            os.unlink("/dev/shm" + shm_name)
        except OSError:
            pass
    for r in (0, 1):
        assert isinstance(got[r], dict), got[r]
    a, b = got[0], got[1]
    assert np.array_equal(a["P"], b["P"]) and np.array_equal(a["w"], b["w"])
    assert np.array_equal(a["loss"], b["loss"]) and a["it"] == b["it"]
    one = _run_psgd(case, 1, 0, None)
    assert a["it"] == one["it"]
    np.testing.assert_allclose(a["P"], one["P"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(a["w"], one["w"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(a["loss"], one["loss"], rtol=1e-10)
    assert np.abs(a["P"]).max() > 0


def _psgd_est_worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        import warnings

        import torch.distributed as dist

        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), SPFM_COMM="shm")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from sparsepoly_amd import SparseFactorizationMachineRegressor

        X, y = _problem("squared")
        est = SparseFactorizationMachineRegressor(
            degree=2, n_components=5, solver="psgd", regularizer="squaredl12", alpha=1e-3,
            beta=1e-2, gamma=1e-3, max_iter=3, tol=-1, n_iter_no_change=100, batch_size=128,
            eta0=0.05, shuffle=True, random_state=0, precision="f64", device=0, distributed=True)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            est.fit(X, y)
        q.put((rank, dict(P=est.P_, w=est.w_, it=est.it_, n_iter=est.n_iter_)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:
        q.put((rank, repr(e)))


def test_psgd_estimator_distributed_two_ranks_one_gpu():
    """solver='psgd' with distributed=True through the estimator (two gloo ranks, host
    communicator): same model on both ranks, equal to the one-GPU fit."""
    import socket
    import warnings

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_psgd_est_worker, args=(r, 2, port, q)) for r in (0, 1)]
    for p in procs:
        p.start()
    got = {}
    try:
        for _ in procs:
            rank, res = q.get(timeout=240)
            got[rank] = res
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    for r in (0, 1):
        assert isinstance(got[r], dict), got[r]
    assert np.array_equal(got[0]["P"], got[1]["P"]) and np.array_equal(got[0]["w"], got[1]["w"])
    sys.path.insert(0, ROOT)
    from sparsepoly_amd import SparseFactorizationMachineRegressor

    X, y = _problem("squared")
    one = SparseFactorizationMachineRegressor(
        degree=2, n_components=5, solver="psgd", regularizer="squaredl12", alpha=1e-3,
        beta=1e-2, gamma=1e-3, max_iter=3, tol=-1, n_iter_no_change=100, batch_size=128,
        eta0=0.05, shuffle=True, random_state=0, precision="f64", device=0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        one.fit(X, y)
    np.testing.assert_allclose(got[0]["P"], one.P_, rtol=0, atol=1e-10)
    np.testing.assert_allclose(got[0]["w"], one.w_, rtol=0, atol=1e-10)
    assert (got[0]["it"], got[0]["n_iter"]) == (one.it_, one.n_iter_)
