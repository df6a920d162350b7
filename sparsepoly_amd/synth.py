"""Synthetic sparse regression problems for benchmarks and large-size tests
(SURVEY.md section 8d): counter-based hash RNG, so every process (and the CPU
baseline) regenerates identical data from (seed, shape) without shipping files.

Row i draws ``nnz_per_row`` column ids ``hash(seed, i, t) mod d`` (duplicates
merged, so nnz/row <= nnz_per_row); values are N(0,1) (Box-Muller on the hash
stream) rounded to float32-representable numbers, so float64 and float32 engines
read identical inputs.  ``y = ANOVA_2(x; P*) + 0.1 N(0,1)`` with a planted
block-sparse ``P*`` of ``k_true`` components.
"""
import numpy as np
import scipy.sparse as sp

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(x):
    """splitmix64 finaliser on uint64 arrays"""
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
    x ^= x >> np.uint64(30)
    x = (x * np.uint64(0xBF58476D1CE4E5B9)) & _M64
    x ^= x >> np.uint64(27)
    x = (x * np.uint64(0x94D049BB133111EB)) & _M64
    x ^= x >> np.uint64(31)
    return x


def _uniform(keys, stream):
    h = _mix(keys ^ _mix(np.uint64(stream) * np.uint64(0xD1342543DE82EF95) + np.uint64(1)))
    return ((h >> np.uint64(11)).astype(np.float64) + 0.5) / float(1 << 53)


def _normal(keys, stream):
    u1 = _uniform(keys, 2 * stream)
    u2 = _uniform(keys, 2 * stream + 1)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


def make_csr(n, d, nnz_per_row=50, seed=0, chunk_rows=1 << 18, row_range=None,
             structure_only=False):
    """(n, d) CSR matrix, sorted column indices, float64 values that are exactly
    float32-representable.  ``row_range=(lo, hi)`` draws only the rows [lo, hi) of that
    matrix (shape (hi - lo, d)): the generator is counter based, so a rank of a row-sharded
    run builds its shard without the rest.  ``structure_only`` skips the values (int8 ones):
    what a colouring needs."""
    lo, hi = (0, n) if row_range is None else (int(row_range[0]), int(row_range[1]))
    if not 0 <= lo <= hi <= n:
        raise ValueError("row_range outside [0, n]")
    indptr = np.zeros(hi - lo + 1, dtype=np.int64)
    idx_parts, val_parts = [], []
    with np.errstate(over="ignore"):
        for r0 in range(lo, hi, chunk_rows):
            r1 = min(hi, r0 + chunk_rows)
            rows = np.repeat(np.arange(r0, r1, dtype=np.uint64), nnz_per_row)
            t = np.tile(np.arange(nnz_per_row, dtype=np.uint64), r1 - r0)
            h = _mix((rows * np.uint64(nnz_per_row) + t) ^ _mix(np.uint64(seed)))
            cols = h % np.uint64(d)
            key = np.unique(rows * np.uint64(d) + cols)  # sorted by (row, col), de-duplicated
            rr = (key // np.uint64(d)).astype(np.int64)
            cc = (key % np.uint64(d)).astype(np.int32)
            if structure_only:
                vals = np.ones(key.shape[0], dtype=np.int8)
            else:
                vals = _normal(key ^ _mix(np.uint64(seed) + np.uint64(77)), 1)
                vals = vals.astype(np.float32).astype(np.float64)
            indptr[r0 - lo + 1:r1 - lo + 1] = np.bincount(rr - r0, minlength=r1 - r0)
            idx_parts.append(cc)
            val_parts.append(vals)
    np.cumsum(indptr, out=indptr)
    if not idx_parts:
        idx_parts, val_parts = [np.zeros(0, np.int32)], [np.zeros(0, np.float64)]
    X = sp.csr_matrix((np.concatenate(val_parts), np.concatenate(idx_parts), indptr),
                      shape=(hi - lo, d))
    X.has_sorted_indices = True
    return X


def planted_target(X, k_true=8, seed=0, noise=0.1, block=0.02, row0=0):
    """y = sum_s ANOVA_2(p*_s, x) + noise * N(0,1); P* block-sparse (each component
    lives on a random `block` fraction of the features).  ``row0``: global index of X's
    first row (row shards draw the noise of their own rows)."""
    n, d = X.shape
    rng = np.random.RandomState(seed + 1000)
    P = np.zeros((k_true, d))
    for s in range(k_true):
        sup = rng.rand(d) < block
        P[s, sup] = rng.randn(int(sup.sum())) * 0.5
    XP = np.asarray(X @ P.T)
    X2 = X.multiply(X).tocsr()
    y = 0.5 * (XP ** 2 - np.asarray(X2 @ (P ** 2).T)).sum(axis=1)
    with np.errstate(over="ignore"):
        z = _normal(np.arange(row0, row0 + n, dtype=np.uint64)
                    ^ _mix(np.uint64(seed) + np.uint64(4242)), 3)
    y = y + noise * z
    return y.astype(np.float32).astype(np.float64), P


def make_problem(n, d, nnz_per_row=50, seed=0, k_true=8, row_range=None):
    """(X, y) of the (n, d) problem, or -- ``row_range=(lo, hi)`` -- its rows [lo, hi) only:
    identical to ``X[lo:hi], y[lo:hi]`` of the whole problem at 1/world of the work."""
    X = make_csr(n, d, nnz_per_row, seed, row_range=row_range)
    y, P_true = planted_target(X, k_true, seed, row0=0 if row_range is None else row_range[0])
    return X, y
