"""CPU-side checks: the C-ABI library loads and exports every symbol include/spfm.h
declares, host-only schedule construction, the synthetic generator, sklearn plumbing of
the estimators, and loud failure without a GPU.  No device compute here."""
import inspect
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "spfm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(spfm_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from sparsepoly_amd import _capi

    lib = _capi.load()
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), "libspfm_hip.so lacks %s" % name
    assert sorted(_capi.SYMBOLS) == declared


def test_product_never_imports_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "sparsepoly_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.replace("oracle/", "").replace("the oracle", "") \
                    or "import oracle" not in src, f
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f


def _has_gpu():
    import ctypes

    from sparsepoly_amd import _capi

    lib = _capi.load()
    h = ctypes.c_void_p()
    rc = lib.spfm_create(ctypes.byref(h), 0, 0)
    if rc == 0:
        lib.spfm_destroy(h)
    return rc == 0


def test_fit_fails_loudly_without_gpu():
    """No CPU fallback: without a HIP device fit() raises instead of computing elsewhere."""
    if _has_gpu():
        pytest.skip("a GPU is present")
    from sparsepoly_amd import SparseFactorizationMachineRegressor
    from sparsepoly_amd.engine import SpfmError

    X = np.random.RandomState(0).randn(20, 4)
    with pytest.raises(SpfmError):
        SparseFactorizationMachineRegressor(max_iter=1).fit(X, X[:, 0])
    est = SparseFactorizationMachineRegressor()
    est.P_, est.w_, est.lams_ = np.zeros((1, 2, 4)), np.zeros(4), np.ones(2)
    with pytest.raises(SpfmError):
        est.predict(X)


@pytest.mark.parametrize("mode", ["exact", "colored"])
@pytest.mark.parametrize("max_batch", [0, 1, 5])
def test_schedule_is_a_partition_into_row_disjoint_batches(mode, max_batch):
    from sparsepoly_amd.schedule import build_schedule

    rng = np.random.RandomState(3)
    X = sp.random(500, 120, density=0.03, format="csc", random_state=rng)
    X.data[:] = 1.0
    jf = rng.permutation(120).astype(np.int32)
    order, bp = build_schedule(X, mode, jf, max_batch)
    assert sorted(order.tolist()) == list(range(120))
    assert bp[0] == 0 and bp[-1] == 120 and np.all(np.diff(bp) > 0)
    if mode == "exact":
        np.testing.assert_array_equal(order, jf)
    if max_batch:
        assert np.diff(bp).max() <= max_batch
    for b in range(len(bp) - 1):
        rows = np.concatenate([X.indices[X.indptr[j]:X.indptr[j + 1]]
                               for j in order[bp[b]:bp[b + 1]]] + [np.empty(0, np.int32)])
        assert len(rows) == len(np.unique(rows)), "batch %d shares a row" % b
    if mode == "exact" and not max_batch:
        # maximal runs: the first column of every batch conflicts with the previous batch
        for b in range(1, len(bp) - 1):
            prev = np.concatenate([X.indices[X.indptr[j]:X.indptr[j + 1]]
                                   for j in order[bp[b - 1]:bp[b]]])
            j = order[bp[b]]
            assert np.intersect1d(prev, X.indices[X.indptr[j]:X.indptr[j + 1]]).size > 0


@pytest.mark.parametrize("max_batch", [0, 3])
def test_parallel_colouring_is_the_sequential_first_fit(max_batch, monkeypatch):
    """Big inputs take the threaded colouring (csrc/spfm_schedule.cpp, schedule_colored_parallel:
    d >= 4096 and nnz >= 2^20).  It must return exactly the first-fit colouring of the given
    order -- restated here with Python sets -- for any number of threads."""
    from sparsepoly_amd.schedule import build_schedule

    rng = np.random.RandomState(5)
    n, d, per_row = 66000, 4200, 16
    cols = rng.randint(0, d, size=(n, per_row))
    rows = np.repeat(np.arange(n), per_row)
    X = sp.csc_matrix((np.ones(n * per_row), (rows, cols.ravel())), shape=(n, d))
    X.sum_duplicates()
    X.sort_indices()
    assert X.nnz >= (1 << 20)
    jf = rng.permutation(d).astype(np.int32)
    got = {}
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("SPFM_THREADS", threads)
        order, bp = build_schedule(X, "colored", jf, max_batch)
        got[threads] = (order.copy(), bp.copy())
    for threads in ("3", "8"):
        np.testing.assert_array_equal(got[threads][0], got["1"][0])
        np.testing.assert_array_equal(got[threads][1], got["1"][1])
    # first fit in Python: colours present in each row, lowest colour absent from all rows of j
    row_colours = [set() for _ in range(n)]
    classes = []
    for j in jf:
        rj = X.indices[X.indptr[j]:X.indptr[j + 1]]
        taken = set().union(*[row_colours[i] for i in rj]) if len(rj) else set()
        c = 0
        while c < len(classes) and (c in taken or (max_batch and len(classes[c]) >= max_batch)):
            c += 1
        if c == len(classes):
            classes.append([])
        classes[c].append(int(j))
        for i in rj:
            row_colours[i].add(c)
    want_order = np.array([j for cl in classes for j in cl], dtype=np.int32)
    want_bp = np.cumsum([0] + [len(cl) for cl in classes]).astype(np.int32)
    np.testing.assert_array_equal(got["1"][0], want_order)
    np.testing.assert_array_equal(got["1"][1], want_bp)


def test_schedule_rejects_non_permutation():
    from sparsepoly_amd.schedule import build_schedule

    X = sp.random(50, 10, density=0.2, format="csc", random_state=0)
    with pytest.raises(ValueError):
        build_schedule(X, "exact", np.zeros(10, dtype=np.int32))


def test_colored_schedule_empty_and_full_columns():
    from sparsepoly_amd.schedule import build_schedule

    Xd = np.zeros((30, 6))
    Xd[:, 0] = 1.0            # full column: its own batch
    Xd[::2, 1] = 1.0
    Xd[1::2, 2] = 1.0         # disjoint from column 1
    Xd[:5, 5] = 1.0
    order, bp = build_schedule(sp.csc_matrix(Xd), "colored")
    batches = [set(order[bp[b]:bp[b + 1]].tolist()) for b in range(len(bp) - 1)]
    assert any(b == {0, 3, 4} for b in batches)  # empty columns conflict with nothing
    assert any({1, 2} <= b for b in batches)


def test_synth_generator_is_deterministic_and_f32_exact():
    from sparsepoly_amd.synth import make_problem

    X1, y1 = make_problem(2000, 300, 20, seed=5)
    X2, y2 = make_problem(2000, 300, 20, seed=5, )
    assert (X1 != X2).nnz == 0 and np.array_equal(y1, y2)
    assert X1.has_sorted_indices and np.diff(X1.indptr).max() <= 20
    np.testing.assert_array_equal(X1.data, X1.data.astype(np.float32).astype(np.float64))
    np.testing.assert_array_equal(y1, y1.astype(np.float32).astype(np.float64))
    assert abs(X1.data.mean()) < 0.05 and abs(X1.data.std() - 1) < 0.05
    # chunked generation gives the same matrix
    from sparsepoly_amd.synth import make_csr

    X3 = make_csr(2000, 300, 20, seed=5, chunk_rows=333)
    assert (X1 != X3).nnz == 0


def test_estimator_signature_matches_reference_order():
    """Drop-in: the reference's keywords come first, in the reference's order, with the
    reference's defaults (sparse_factorization_machines.py:631-658, :866-894)."""
    from sklearn.base import clone

    from sparsepoly_amd import (SparseFactorizationMachineClassifier,
                                SparseFactorizationMachineRegressor)

    ref_reg = ["degree", "n_components", "solver", "regularizer", "alpha", "beta", "gamma", "mean",
               "tol", "fit_lower", "fit_linear", "warm_start", "init_lambdas", "max_iter",
               "shuffle", "batch_size", "eta0", "learning_rate", "power_t", "n_iter_no_change",
               "verbose", "callback", "n_calls", "random_state"]
    defaults = dict(degree=2, n_components=2, solver="pcd", regularizer="squaredl12", alpha=1,
                    beta=1, gamma=1, mean=False, tol=1e-6, fit_lower="explicit", fit_linear=True,
                    warm_start=False, init_lambdas="ones", max_iter=100, shuffle=False,
                    batch_size="auto", eta0=1.0, learning_rate="optimal", power_t=1.0,
                    n_iter_no_change=5, verbose=False, callback=None, n_calls=10,
                    random_state=None)
    sig = inspect.signature(SparseFactorizationMachineRegressor.__init__)
    names = [p for p in sig.parameters if p != "self"]
    assert names[:len(ref_reg)] == ref_reg
    for k, v in defaults.items():
        assert sig.parameters[k].default == v
    sigc = inspect.signature(SparseFactorizationMachineClassifier.__init__)
    namesc = [p for p in sigc.parameters if p != "self"]
    assert namesc[:2] == ["degree", "loss"] and namesc[2:len(ref_reg) + 1] == ref_reg[1:]
    assert sigc.parameters["loss"].default == "squared_hinge"
    est = SparseFactorizationMachineClassifier(degree=3, loss="logistic", gamma=0.5)
    c = clone(est)
    assert c.get_params() == est.get_params() and c.loss == "logistic"
    assert SparseFactorizationMachineRegressor().loss == "squared"


def test_package_exports_reference_names():
    import sparsepoly_amd as sa

    for name in ["L1", "L21", "OmegaCS", "OmegaTI", "SquaredL12", "SquaredL21",
                 "SparseAllSubsetsClassifier", "SparseAllSubsetsRegressor",
                 "SparseFactorizationMachineClassifier", "SparseFactorizationMachineRegressor"]:
        assert hasattr(sa, name)
    from sparsepoly_amd.regularizer import REGULARIZATION

    assert list(REGULARIZATION) == ["squaredl12", "squaredl21", "l1", "l21", "omegati", "omegacs"]


def test_validation_errors_raised_before_any_device_work():
    """These ValueErrors/TypeErrors come from the host layer (reference base.py:17-34,
    :126-142; sparse_factorization_machines.py:396-404,428-430) and need no GPU."""
    from sparsepoly_amd import (SparseFactorizationMachineClassifier,
                                SparseFactorizationMachineRegressor)

    X = np.random.RandomState(0).randn(12, 3)
    y = X[:, 0]
    with pytest.raises(ValueError, match="Regularizer nope not supported"):
        SparseFactorizationMachineRegressor(regularizer="nope").fit(X, y)
    with pytest.raises(ValueError, match="Loss function nope not supported"):
        SparseFactorizationMachineClassifier(loss="nope").fit(X, np.sign(y))
    with pytest.raises(ValueError, match="Lambdas must be initialized"):
        SparseFactorizationMachineRegressor(init_lambdas="nope").fit(X, y)
    with pytest.raises(ValueError, match="Solver nope is not supported."):
        SparseFactorizationMachineRegressor(solver="nope").fit(X, y)
    with pytest.raises(TypeError, match="Only binary targets supported"):
        SparseFactorizationMachineClassifier().fit(X, y)


def test_row_block_partition():
    from sparsepoly_amd.distributed import row_block

    for n, w in [(10, 3), (7, 8), (1000003, 8), (5, 1)]:
        blocks = [row_block(n, r, w) for r in range(w)]
        assert blocks[0][0] == 0 and blocks[-1][1] == n
        for (a, b), (c, d) in zip(blocks[:-1], blocks[1:]):
            assert b == c and b >= a
        sizes = [b - a for a, b in blocks]
        assert max(sizes) - min(sizes) <= 1


def test_schedule_object_roundtrip(tmp_path):
    from sparsepoly_amd.schedule import Schedule

    X = sp.random(400, 90, density=0.04, format="csr", random_state=2)
    s1 = Schedule.build(X, "colored")
    assert s1.n_batches >= 1 and sorted(s1.order.tolist()) == list(range(90))
    assert np.diff(s1.batch_ptr).max() <= 64
    s1.save(tmp_path / "sched.npz")
    s2 = Schedule.load(tmp_path / "sched.npz")
    np.testing.assert_array_equal(s1.order, s2.order)
    np.testing.assert_array_equal(s1.batch_ptr, s2.batch_ptr)
    assert s2.mode == "colored" and s2.shape == (400, 90) and s2.nnz == X.nnz
    assert "n_batches=%d" % s1.n_batches in repr(s2)
