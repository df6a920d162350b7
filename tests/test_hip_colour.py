"""The first-fit colouring on the device (csrc/spfm_colour.hip) against the host form
(csrc/spfm_schedule.cpp): the same visiting order must give the same order and the same batch
boundaries, for uniform and skewed matrices, natural and shuffled visiting orders, class caps of
64 and 512 columns, empty columns and rows -- and the epochs on top are then bit-identical."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


def _matrix(n, d, per_row, seed, zipf=False, empty=True):
    rng = np.random.RandomState(seed)
    rows = np.repeat(np.arange(n), per_row)
    if zipf:  # a few very frequent columns
        p = 1.0 / np.arange(1, d + 1) ** 0.9
        cols = rng.choice(d, size=n * per_row, p=p / p.sum())
    else:
        cols = rng.randint(0, d, size=n * per_row)
    X = sp.csr_matrix((rng.randn(n * per_row), (rows, cols)), shape=(n, d))
    X.sum_duplicates()
    if empty:
        keep = np.ones(d)
        keep[[0, d // 2, d - 1]] = 0
        X = sp.csr_matrix(X @ sp.diags(keep))
        X.eliminate_zeros()
    X.sort_indices()
    return X, rng.randn(n)


def _schedule(X, y, device, order, solver="pcd", degree=2, options=()):
    from sparsepoly_amd.engine import HipEngine

    eng = HipEngine(0, "f32")
    eng.set_option("colour_device", device)
    for k, v in options:
        eng.set_option(k, v)
    eng.set_data(X, y)
    d = X.shape[1]
    k = 3
    eng.set_params(0.01 * np.random.RandomState(0).randn(degree - 1, k, d), np.zeros(d), np.ones(k))
    eng.configure(solver, "squared", "l1", degree)
    eng.init_pred(degree, True, degree > 2)
    o = eng.set_schedule("colored", order)
    used = eng.get_option("colour_device_used")
    sched = eng.get_schedule("colored")
    v = eng.cd_linear_epoch(1.0)
    if solver == "pcd":
        v += eng.pcd_epoch(0, degree, 5.0, 1e-3, 1.0, np.arange(k, dtype=np.int32))
    else:
        v += eng.pbcd_epoch(0, degree, 5.0, 1e-3, 1.0)
    P, w = eng.get_params()
    eng.close()
    return used, o, sched.batch_ptr, v, P, w


CASES = [
    # n, d, per_row, zipf, shuffled, solver, degree, options
    (30000, 5000, 40, False, False, "pcd", 2, ()),             # 64- or 512-column classes (policy)
    (30000, 5000, 40, False, True, "pcd", 2, ()),
    (60000, 4500, 20, True, False, "pcd", 2, ()),              # skewed columns
    (30000, 5000, 40, False, False, "pcd", 3, ()),             # degree 3: 64-column classes
    (30000, 5000, 40, False, True, "pbcd", 2, ()),
    (200000, 60000, 6, False, False, "pcd", 2, (("wide_min_cols", 0),)),  # wide classes (<= 512)
    (40000, 4096, 30, False, True, "pcd", 2, (("wide", 0),)),
]


@pytest.mark.parametrize("n,d,per_row,zipf,shuffled,solver,degree,options", CASES)
def test_device_colouring_equals_host_colouring(n, d, per_row, zipf, shuffled, solver, degree,
                                                options):
    X, y = _matrix(n, d, per_row, seed=n % 97 + d % 13, zipf=zipf)
    assert X.nnz >= (1 << 20)
    order = np.arange(d, dtype=np.int32)
    if shuffled:
        np.random.RandomState(4).shuffle(order)
    dev = _schedule(X, y, 1, order, solver, degree, options)
    host = _schedule(X, y, 0, order, solver, degree, options)
    assert dev[0] == 1 and host[0] == 0
    assert np.array_equal(dev[1], host[1])          # the order
    assert np.array_equal(dev[2], host[2])          # the batch boundaries
    assert dev[3] == host[3]
    assert np.array_equal(dev[4], host[4]) and np.array_equal(dev[5], host[5])
    # a colouring: no two columns of a class share a row
    Xc = X.tocsc()
    bp, o = dev[2], dev[1]
    for b in (0, len(bp) // 2, len(bp) - 2):
        cols = o[bp[b]:bp[b + 1]]
        rows = np.concatenate([Xc.indices[Xc.indptr[j]:Xc.indptr[j + 1]] for j in cols])
        assert len(rows) == len(np.unique(rows))


def test_small_problems_and_too_many_colours_stay_on_the_host():
    X, y = _matrix(3000, 300, 8, seed=1)                       # below the size threshold
    used = _schedule(X, y, 1, np.arange(300, dtype=np.int32))[0]
    assert used == 0
    # 300 000 columns in classes of 64: more than 3 500 classes in any colouring
    X, y = _matrix(60000, 300000, 20, seed=2, empty=False)
    r = _schedule(X, y, 1, np.arange(300000, dtype=np.int32), options=(("wide", 0),))
    assert r[0] == 0 and len(r[2]) - 1 >= 300000 // 64
