"""world_size-2 test (gloo, CPU) of the multi-GPU protocol of DESIGN.md section 6:
rows sharded in contiguous blocks, the per-step column partial sums all-reduced, the
chain (step size, prox, regularizer cache) replicated on every rank.  The device engine
cannot run here, so a NumPy double of one engine step (same batch schedule from the
library's host-only spfm_schedule_build, same helpers from sparsepoly_amd.distributed)
stands in for the kernels; the claim checked is that the sharded sweep equals the
unsharded oracle sweep in the same order, and that all ranks end bit-identical."""
import os
import socket
import sys

import numpy as np
import pytest
import scipy.sparse as sp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    rng = np.random.RandomState(12)
    X = sp.random(240, 40, density=0.12, random_state=rng, data_rvs=rng.randn, format="csr")
    y = rng.randn(240)
    P0 = 0.01 * rng.randn(1, 5, 40)
    lams = np.sign(rng.randn(5))
    return X, y, P0, lams


def _worker(rank, world, port, regname, out_dir):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as orc
    from sparsepoly_amd import distributed as spdist
    from sparsepoly_amd.schedule import build_schedule

    X, y, P0, lams = _problem()
    n, d = X.shape
    k = P0.shape[1]
    beta, gamma, eta, alpha = 10.0, 0.05, 1.0, 0.1
    # control-plane helpers used by the real path
    token = spdist.broadcast_bytes(b"x" * 128 if rank == 0 else None, 0)
    assert token == b"x" * 128
    lo, hi = spdist.row_block(n)
    Xc = sp.csc_matrix(X)
    Xc.sort_indices()
    order, bp = build_schedule(Xc, "colored")          # GLOBAL structure on every rank
    Xl = sp.csc_matrix(X[lo:hi])
    Xl.sort_indices()
    yl = y[lo:hi]
    P = P0[0].copy()
    w = np.zeros(d)
    cn = np.asarray(Xl.multiply(Xl).sum(axis=0)).ravel()
    t = torch.from_numpy(cn)
    dist.all_reduce(t)                                  # col_norm_sq is global
    y_pred = orc.poly_predict(X, P, lams, 2)[lo:hi].copy()
    reg = orc.Regularizer(regname)
    reg.init_cache_pcd(2, d, k)

    def col(j):
        sl = slice(Xl.indptr[j], Xl.indptr[j + 1])
        return Xl.indices[sl], Xl.data[sl]

    viols = []
    for epoch in range(2):
        viol = 0.0
        for b in range(len(bp) - 1):                    # cd_linear (cd_linear.py:8-33)
            cols = order[bp[b]:bp[b + 1]]
            part = np.zeros(len(cols))
            for q, j in enumerate(cols):
                i, x = col(j)
                part[q] = np.sum((y_pred[i] - yl[i]) * x)
            tp = torch.from_numpy(part)
            dist.all_reduce(tp)
            for q, j in enumerate(cols):
                u = (part[q] + alpha * w[j]) / (cn[j] + alpha)
                w[j] -= u
                viol += abs(u)
                i, x = col(j)
                y_pred[i] -= u * x
        for s in range(k):                              # pcd_epoch (pcd.py:71-137)
            A1 = np.asarray(Xl @ P[s]).ravel()
            reg.compute_cache_pcd(P, 2, s)
            for b in range(len(bp) - 1):
                cols = order[bp[b]:bp[b + 1]]
                part = np.zeros((len(cols), 2))
                for q, j in enumerate(cols):
                    i, x = col(j)
                    dA = x * (A1[i] - P[s, j] * x)
                    part[q] = (np.sum((y_pred[i] - yl[i]) * dA), np.sum(dA * dA))
                tp = torch.from_numpy(part)
                dist.all_reduce(tp)                     # the only exchange of the step
                for q, j in enumerate(cols):            # replicated chain
                    p_old = P[s, j]
                    inv = part[q, 1] * 1.0 + beta
                    upd = (part[q, 0] * lams[s] + beta * p_old) / inv
                    p_new = reg.prox_cd(p_old - eta * upd, eta * gamma / inv, 2, j)
                    P[s, j] = p_new
                    reg.update_cache_pcd(P, 2, s, j)
                    dl = p_old - p_new
                    viol += abs(dl)
                    i, x = col(j)
                    dA = x * (A1[i] - p_old * x)
                    A1[i] -= dl * x
                    y_pred[i] -= lams[s] * dl * dA
        viols.append(viol)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), P=P, w=w, viol=np.array(viols),
             order=order, lo=lo, hi=hi, y_pred=y_pred)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("regname", ["squaredl12", "omegati"])
def test_row_sharded_sweep_equals_unsharded_oracle(oracle, regname, tmp_path):
    import torch.multiprocessing as mp

    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, regname, str(tmp_path)), nprocs=world, join=True)
    r0 = np.load(tmp_path / "rank0.npz")
    r1 = np.load(tmp_path / "rank1.npz")
    # replicated state is bit-identical on all ranks
    np.testing.assert_array_equal(r0["P"], r1["P"])
    np.testing.assert_array_equal(r0["w"], r1["w"])
    np.testing.assert_array_equal(r0["viol"], r1["viol"])
    assert (r0["lo"], r0["hi"], r1["lo"], r1["hi"]) == (0, 120, 120, 240)
    X, y, P0, lams = _problem()
    fm = oracle.OracleFM(degree=2, n_components=5, solver="pcd", regularizer=regname, alpha=0.1,
                         beta=10.0, gamma=0.05, tol=0, max_iter=2, feature_order=r0["order"])
    fm.fit(X, y, P_init=P0, lams_init=lams)
    np.testing.assert_allclose(r0["P"], fm.P_[0], rtol=0, atol=1e-10)
    np.testing.assert_allclose(r0["w"], fm.w_, rtol=0, atol=1e-10)
    np.testing.assert_allclose(r0["viol"], [h[0] for h in fm.history], rtol=1e-10)
    np.testing.assert_allclose(np.concatenate([r0["y_pred"], r1["y_pred"]]), fm.y_pred_,
                               rtol=0, atol=1e-9)
