"""BASELINE configs[4] -- degree=2, n_components=30, pcd on the 10M x 1M synthetic CSR -- at its
full size on one GPU (the matrix fits: CSC + CSR images 8 GB).  Checks, for the engine that takes
a colour class as ONE dependent step (multi-kernel; the persistent pass caps a step at 64
columns): steps per sweep = number of colours (~2 750), the incremental
prediction equals the recomputed one after a cd_linear epoch + 2 component passes, and the result
equals the CPU oracle run on the same passes in the reported order (f32 storage tolerances).
``SPFM_C5_N`` / ``SPFM_C5_D`` shrink the problem for a quick run.  Needs a real MI355X."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config5_full_size_single_gpu(oracle):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_c5

    from sparsepoly_amd.synth import make_problem

    n = int(os.environ.get("SPFM_C5_N", 10_000_000))
    d = int(os.environ.get("SPFM_C5_D", 1_000_000))
    X, y = make_problem(n, d, 50, seed=0)
    Xc = X.tocsc()
    Xc.sort_indices()
    del X
    P0 = 0.01 * np.random.RandomState(0).randn(1, bench_c5.K, d)
    r = bench_c5.run_engine(Xc, y, P0, 2, {"persistent": 0}, reps=0)
    assert sorted(set(np.diff(np.sort(r["order"])))) == [1]          # a permutation
    if n == 10_000_000 and d == 1_000_000:
        assert 2000 < r["info"]["steps_per_sweep"] < 3500, r["info"]  # colours, not colours x 6
    scale = max(1.0, float(np.abs(r["y_recomputed"]).max()))
    np.testing.assert_allclose(r["y_incremental"], r["y_recomputed"], rtol=0, atol=2e-4 * scale)
    ref = bench_c5.run_oracle(Xc, y, P0, r["y0"], r["order"], 2)
    np.testing.assert_allclose(r["v_lin"], ref["v_lin"], rtol=1e-5)
    np.testing.assert_allclose(r["v"], ref["v"], rtol=1e-5)
    np.testing.assert_allclose(r["P"][0], ref["P"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(r["w"], ref["w"], rtol=0, atol=1e-4)
    np.testing.assert_allclose(r["y_pred"], ref["y_pred"], rtol=0, atol=2e-4 * scale)
