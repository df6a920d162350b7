"""The WIDE persistent passes (steps of up to 512 columns, one thread per column slot;
csrc/spfm_pcdw.hip.h) for degree-2 pcd and cd_linear: a sparse, wide matrix whose colour classes
hold hundreds of columns, against the oracle in the reported order and against the other engines
(64-column persistent passes, multi-kernel), for several workgroup counts (owner rounds, empty row
blocks), rows in LDS and in global memory, all losses and the three pcd regularizers.
Needs a real MI355X: ``pytest -m gpu``."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _problem(loss, n=6000, d=3000, per_row=4, seed=11):
    rng = np.random.RandomState(seed)
    rows = np.repeat(np.arange(n), per_row)
    cols = rng.randint(0, d, size=n * per_row)
    vals = rng.randn(n * per_row).astype(np.float32).astype(np.float64)
    X = sp.csr_matrix((vals, (rows, cols)), shape=(n, d))
    X.sum_duplicates()
    X.sort_indices()
    y = rng.randn(n).astype(np.float32).astype(np.float64)
    if loss != "squared":
        y = np.where(y > 0, 1.0, -1.0)
    return X, y


def _run(X, y, loss, reg, precision, options, k=5, epochs=2, beta=10.0, gamma=1e-3):
    from sparsepoly_amd.engine import HipEngine

    d = X.shape[1]
    eng = HipEngine(0, precision)
    for key, val in options.items():
        eng.set_option(key, val)
    eng.set_data(X, y)
    P0 = 0.05 * np.random.RandomState(1).randn(1, k, d)
    eng.set_params(P0, np.zeros(d), np.where(np.arange(k) % 2 == 0, 1.0, -1.0))
    eng.configure("pcd", loss, reg, 2)
    eng.init_pred(2, True, False)
    order = eng.set_schedule("colored", np.arange(d, dtype=np.int32))
    sched = eng.get_schedule()
    ic = np.arange(k, dtype=np.int32)
    viol = []
    for _ in range(epochs):
        viol.append(eng.cd_linear_epoch(0.5) + eng.pcd_epoch(0, 2, beta, gamma, 1.0, ic))
    P, w = eng.get_params()
    out = dict(order=order, viol=np.array(viol), P=P, w=w, y_pred=eng.get_y_pred(),
               wide=eng.get_option("wide_active"), lds=eng.get_option("wide_lds_active"),
               max_step=int(np.diff(sched.batch_ptr).max()), steps=eng.n_batches, P0=P0)
    eng.close()
    return out


@pytest.mark.parametrize("loss,reg", [("squared", "squaredl12"), ("logistic", "omegati"),
                                      ("squared_hinge", "l1"), ("squared", "l1")])
def test_wide_pass_matches_oracle(oracle, loss, reg):
    X, y = _problem(loss)
    r = _run(X, y, loss, reg, "f64", {})
    assert r["wide"] == 1 and 64 < r["max_step"] <= 512, (r["wide"], r["max_step"])
    k = 5
    fm = oracle.OracleFM(degree=2, loss=loss, n_components=k, solver="pcd", regularizer=reg,
                         alpha=0.5, beta=10.0, gamma=1e-3, tol=0, max_iter=2, fit_linear=True,
                         feature_order=r["order"])
    fm.fit(X, y, P_init=r["P0"], lams_init=np.where(np.arange(k) % 2 == 0, 1.0, -1.0))
    np.testing.assert_allclose(r["viol"], [h[0] for h in fm.history], rtol=1e-9)
    np.testing.assert_allclose(r["P"], fm.P_, rtol=0, atol=1e-8)
    np.testing.assert_allclose(r["w"], fm.w_, rtol=0, atol=1e-8)
    np.testing.assert_allclose(r["y_pred"], fm.y_pred_, rtol=0, atol=1e-7)


@pytest.mark.parametrize("options", [{"pcdw_groups": 1}, {"pcdw_groups": 7}, {"pcdw_groups": 32},
                                     {"pcdw_groups": 100}, {"prb_lds": 0}, {"persistent": 0},
                                     {"wide_ep": 0}, {"wide_ep": 0, "pcdw_groups": 7}])
def test_wide_pass_engine_options(options):
    """Same schedule, other workgroup counts / row residency / the multi-kernel engine: results
    agree to reduction-order rounding.  float storage uses LDS rows for the squared loss."""
    X, y = _problem("squared")
    a = _run(X, y, "squared", "squaredl12", "f64", {})
    b = _run(X, y, "squared", "squaredl12", "f64", options)
    np.testing.assert_array_equal(a["order"], b["order"])
    np.testing.assert_allclose(a["viol"], b["viol"], rtol=1e-10)
    np.testing.assert_allclose(a["P"], b["P"], rtol=0, atol=1e-10)
    np.testing.assert_allclose(a["y_pred"], b["y_pred"], rtol=0, atol=1e-9)


def test_wide_pass_f32_lds_rows_and_narrow_engine():
    """float storage: the row blocks live in LDS (squared loss); against the f64 run, and against
    the 64-column persistent passes (option wide=0: same algorithm, six times the steps)."""
    X, y = _problem("squared")
    a = _run(X, y, "squared", "squaredl12", "f64", {})
    b = _run(X, y, "squared", "squaredl12", "f32", {})
    c = _run(X, y, "squared", "squaredl12", "f32", {"prb_lds": 0})
    assert b["lds"] == 1 and c["lds"] == 0
    for r in (b, c):
        np.testing.assert_array_equal(a["order"], r["order"])
        np.testing.assert_allclose(a["viol"], r["viol"], rtol=2e-5)
        np.testing.assert_allclose(a["P"], r["P"], rtol=0, atol=1e-4)
    n = _run(X, y, "squared", "squaredl12", "f64", {"wide": 0})
    assert n["wide"] == 0 and n["max_step"] <= 64 and n["steps"] > a["steps"]


@pytest.mark.parametrize("ep", [1, 0])
@pytest.mark.parametrize("lds_rows", [1, 7, 40])
@pytest.mark.parametrize("groups", [None, 5])
def test_wide_pass_first_rows_of_a_block_in_lds(oracle, lds_rows, groups, ep):
    """LR = 2 (a row block too large for LDS -- BASELINE configs[4] on one GPU): the first
    `wide_lds_rows` rows of every block live in LDS (residual form), the others in global memory;
    an entry's row state comes from wherever its row lives.  The option caps the LDS rows so that a
    small problem takes the path.  With zero targets the residual IS the prediction, so the three
    forms (mixed, all global, all LDS) run the same arithmetic on the same floats: bit-identical.
    With real targets: equal to float rounding, and to the oracle."""
    X, y = _problem("squared")
    opts = {"wide_ep": ep} if groups is None else {"pcdw_groups": groups, "wide_ep": ep}
    for yy, exact in ((np.zeros_like(y), True), (y, False)):
        hyb = _run(X, yy, "squared", "squaredl12", "f32", dict(opts, wide_lds_rows=lds_rows))
        glob = _run(X, yy, "squared", "squaredl12", "f32", dict(opts, prb_lds=0))
        lds = _run(X, yy, "squared", "squaredl12", "f32", opts)
        assert (hyb["lds"], glob["lds"], lds["lds"]) == (2, 0, 1)
        assert np.abs(hyb["P"] - hyb["P0"]).max() > 1e-4  # it trained
        for r in (glob, lds):
            np.testing.assert_array_equal(hyb["order"], r["order"])
            if exact:
                for key in ("P", "w", "viol", "y_pred"):
                    np.testing.assert_array_equal(hyb[key], r[key], err_msg=key)
            else:
                np.testing.assert_allclose(hyb["viol"], r["viol"], rtol=2e-5)
                np.testing.assert_allclose(hyb["P"], r["P"], rtol=0, atol=2e-5)
                np.testing.assert_allclose(hyb["y_pred"], r["y_pred"], rtol=0, atol=2e-4)
    k = 5
    fm = oracle.OracleFM(degree=2, loss="squared", n_components=k, solver="pcd",
                         regularizer="squaredl12", alpha=0.5, beta=10.0, gamma=1e-3, tol=0,
                         max_iter=2, fit_linear=True, feature_order=hyb["order"])
    fm.fit(X, y, P_init=hyb["P0"], lams_init=np.where(np.arange(k) % 2 == 0, 1.0, -1.0))
    np.testing.assert_allclose(hyb["viol"], [h[0] for h in fm.history], rtol=2e-5)
    np.testing.assert_allclose(hyb["P"], fm.P_, rtol=0, atol=1e-4)


def test_moderately_wide_classes_run_as_64_column_steps(oracle):
    """Mean class width below the threshold (`wide_min_cols` = 110 columns when the 64-column pass
    keeps its rows in LDS, 0.72 x that otherwise -- double storage here): the engine colours again
    with 64 columns per class and runs the 64-column passes (a wide step costs about twice a narrow
    one); `wide_min_cols=0` keeps the wide pass.  Both equal the oracle in their
    reported orders."""
    X, y = _problem("squared", per_row=10)  # first fit: 51 classes, 59 columns on average, <= 96
    k = 5
    runs = {"policy": _run(X, y, "squared", "squaredl12", "f64", {}),
            "wide": _run(X, y, "squared", "squaredl12", "f64", {"wide_min_cols": 0})}
    assert runs["policy"]["wide"] == 0 and runs["policy"]["max_step"] <= 64
    assert runs["wide"]["wide"] == 1 and runs["wide"]["max_step"] > 64
    assert runs["policy"]["steps"] > runs["wide"]["steps"]
    for r in runs.values():
        fm = oracle.OracleFM(degree=2, loss="squared", n_components=k, solver="pcd",
                             regularizer="squaredl12", alpha=0.5, beta=10.0, gamma=1e-3, tol=0,
                             max_iter=2, fit_linear=True, feature_order=r["order"])
        fm.fit(X, y, P_init=r["P0"], lams_init=np.where(np.arange(k) % 2 == 0, 1.0, -1.0))
        np.testing.assert_allclose(r["viol"], [h[0] for h in fm.history], rtol=1e-9)
        np.testing.assert_allclose(r["P"], fm.P_, rtol=0, atol=1e-8)



@pytest.mark.parametrize("ep", [1, 0])
@pytest.mark.parametrize("precision,groups,lds_rows",
                         [("f64", 4, -1), ("f32", 4, 0), ("f32", 2, 0), ("f64", 256, -1),
                          ("f32", 4, -1), ("f32", 2, -1)])
def test_wide_pass_many_entries_per_thread_rows_in_global_memory(oracle, precision, groups, lds_rows,
                                                                 ep):
    """The memory path of the wide pass that BASELINE configs[4] takes at 10M rows, at a size the
    oracle replays in a second: 240k x 16k with 60-entry columns and few row blocks, so a thread
    (= one column slot of one row block) holds 15 (4 blocks) / 30 (2 blocks) entries per step --
    more than the `kPcdwEPT` = 8 it keeps in registers, the rest goes through the two dependent
    global loads in the gradient and again in the scatter -- and a workgroup gathers ~5 000 /
    ~10 000 rows per step (config 5: 430).  The row blocks (60k rows and more) do not fit LDS:
    packed 16-byte row records in global memory (float; with the blocks' first rows in LDS when
    the library chooses) / `yy` + `A` (double).  Classes hold up
    to 477 columns.  256 blocks: the register path on the same matrix.  Against the oracle in the
    reported order (pcd.py:97-135, cd_linear.py:8-33).  ep = 1: the entry-parallel form
    (pcdwe_kernel: 1 024 entries of a (workgroup, step) through its LDS table, the other thousands
    walked by their slots' threads), ep = 0: the thread-per-column form (pcdw_kernel)."""
    X, y = _problem("squared", n=240_000, d=16_000, per_row=4, seed=5)
    k = 3
    r = _run(X, y, "squared", "squaredl12", precision,
             {"pcdw_groups": groups, "wide_lds_rows": lds_rows, "wide_ep": ep}, k=k)
    # float storage, squared loss: the library's own choice (wide_lds_rows = -1) keeps the first
    # ~17k rows of every block in LDS (mode 2) -- entries beyond the registers then take the
    # mixed path too; 0 keeps every row in global memory
    want = 2 if (precision == "f32" and lds_rows != 0) else 0
    if want == 2 and groups == 2 and ep == 1:
        want = 0  # (120k rows per block: less than an eighth of them would fit beside the tables)
    assert r["wide"] == 1 and r["lds"] == want and 256 < r["max_step"] <= 512, \
        (r["wide"], r["lds"], r["max_step"])
    col_len = np.diff(X.tocsc().indptr)
    if groups <= 4:
        assert col_len.mean() / groups > 12  # entries per thread and step, on average
    fm = oracle.OracleFM(degree=2, loss="squared", n_components=k, solver="pcd",
                         regularizer="squaredl12", alpha=0.5, beta=10.0, gamma=1e-3, tol=0,
                         max_iter=2, fit_linear=True, feature_order=r["order"])
    fm.fit(X, y, P_init=r["P0"], lams_init=np.where(np.arange(k) % 2 == 0, 1.0, -1.0))
    ref_viol = [h[0] for h in fm.history]
    if precision == "f64":
        np.testing.assert_allclose(r["viol"], ref_viol, rtol=1e-9)
        np.testing.assert_allclose(r["P"], fm.P_, rtol=0, atol=1e-8)
        np.testing.assert_allclose(r["w"], fm.w_, rtol=0, atol=1e-8)
        np.testing.assert_allclose(r["y_pred"], fm.y_pred_, rtol=0, atol=1e-7)
    else:
        np.testing.assert_allclose(r["viol"], ref_viol, rtol=2e-5)
        np.testing.assert_allclose(r["P"], fm.P_, rtol=0, atol=1e-4)
        np.testing.assert_allclose(r["w"], fm.w_, rtol=0, atol=1e-4)
        scale = max(1.0, float(np.abs(fm.y_pred_).max()))
        np.testing.assert_allclose(r["y_pred"], fm.y_pred_, rtol=0, atol=2e-4 * scale)


@pytest.mark.parametrize("pos_a,pos_b", [(10, 150), (3, 40), (70, 71), (127, 128)])
def test_forced_omegati_clip_inside_a_wide_step(oracle, pos_a, pos_b):
    """omegati.py:97-98 (the clip of _dcache at 0) FORCED inside one wide step whose columns are
    spread over several waves of the chain (pcd_chain_waves): the construction of
    tests/test_hip_branches.py -- two empty columns a, b whose magnitudes make the running cache
    minus |p_b| negative at column b -- placed at chosen positions of a step of 200 pairwise
    row-disjoint columns (a and b in different 64-column chunks, in one chunk, or adjacent across
    a chunk boundary).  The device's clip counter must tick and the result must equal the
    oracle's."""
    from sparsepoly_amd.engine import HipEngine

    ta, tb = 6206275.0, 0.0013605052372440696
    na, nb = 5 * ta, 5 * tb
    assert (na + nb) - na < nb
    d = 200
    rng = np.random.RandomState(5)
    data_cols = [j for j in range(d) if j not in (pos_a, pos_b)]
    n = len(data_cols)
    X = sp.csc_matrix((rng.randn(n), (np.arange(n), np.array(data_cols))), shape=(n, d))
    y = rng.randn(n)
    P0 = np.zeros((1, 1, d))
    P0[0, 0, pos_a], P0[0, 0, pos_b] = na, -nb
    eng = HipEngine(0, "f64")
    eng.set_option("wide_min_cols", 0)
    eng.set_data(X, y)
    eng.set_params(P0, np.zeros(d), np.ones(1))
    eng.configure("pcd", "squared", "omegati", 2)
    eng.init_pred(2, False, False)
    y0 = eng.get_y_pred()
    eng.set_schedule("colored", np.arange(d, dtype=np.int32))
    sched = eng.get_schedule()
    assert eng.get_option("wide_active") == 1 and eng.n_batches == 1
    assert np.array_equal(sched.order, np.arange(d))       # one step, the natural order
    eng.debug_branch_counts(reset=True)
    v = eng.pcd_epoch(0, 2, 1.0, 0.05, 1.0, np.arange(1, dtype=np.int32))
    counts = eng.debug_branch_counts(reset=True)
    P, _ = eng.get_params()
    yp = eng.get_y_pred()
    eng.close()
    assert counts["omegati_clip"] > 0
    ds = oracle.CSC(X)
    regc = oracle.Regularizer("omegati")
    regc.init_cache_pcd(2, d, 1)
    Po = np.ascontiguousarray(P0[0].copy())
    ypo = np.ascontiguousarray(y0.copy())
    A = np.zeros((n, 3))
    vo = oracle.pcd_epoch(Po, ds, y, ypo, np.ones(1), 2, 1.0, 0.05, 1.0, regc, "squared", A,
                          np.arange(1, dtype=np.int32), np.arange(d, dtype=np.int32))
    np.testing.assert_allclose(v, vo, rtol=1e-12)
    np.testing.assert_allclose(P[0], Po, rtol=0, atol=1e-12)
    np.testing.assert_allclose(yp, ypo, rtol=0, atol=1e-12)
    assert P[0, 0, pos_a] == 0.0 and P[0, 0, pos_b] == 0.0


def test_wide_pass_config5_block_shape_rows_partly_in_lds_is_deterministic():
    """BASELINE configs[4]'s per-workgroup shape at an eighth of its size: 32 workgroups of 39 063
    rows (312 KB as (A, residual): the blocks' first ~14k rows in LDS, the others in global
    memory), wide classes, one cd_linear epoch
    and three component passes.  The incrementally maintained prediction equals the recomputed
    one on every row, two runs are bit-identical, and the entry-parallel form agrees with the
    thread-per-column form to float rounding.  (Regression: the prologue read the LDS image of the
    row block before all of it had been written -- a handful of rows per run, different ones each
    time, only with blocks that fill the LDS.)"""
    import os
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    "tools"))
    import bench_c5

    from sparsepoly_amd.synth import make_problem

    X, y = make_problem(1_250_000, 250_000, 50, seed=0)  # 250-entry columns: classes of ~200
    Xc = X.tocsc()
    Xc.sort_indices()
    P0 = 0.01 * np.random.RandomState(0).randn(1, bench_c5.K, Xc.shape[1])
    runs = [bench_c5.run_engine(Xc, y, P0, 3, dict(pcdw_groups=32, wide_min_cols=0, wide_ep=ep),
                                reps=0) for ep in (1, 1, 0)]
    for r in runs:
        assert r["info"]["wide_active"] == 1 and r["info"]["wide_rows_in_lds"] == 2, r["info"]
        scale = max(1.0, float(np.abs(r["y_recomputed"]).max()))
        np.testing.assert_allclose(r["y_incremental"], r["y_recomputed"], rtol=0, atol=2e-4 * scale)
    a, b, c = runs
    for key in ("P", "w", "y_pred"):
        np.testing.assert_array_equal(a[key], b[key], err_msg=key)
    np.testing.assert_allclose(a["v"], c["v"], rtol=1e-6)
    np.testing.assert_allclose(a["P"], c["P"], rtol=0, atol=1e-5)
