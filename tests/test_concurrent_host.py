"""Host logic of the concurrent-fit front end (sparsepoly_amd/concurrent.py): argument checks
that must fire before anything touches the GPU, the co-tenancy context, the hardware-queue
default.  CPU only."""
import os

import numpy as np
import pytest
import scipy.sparse as sp


def test_argument_checks_come_before_any_device_call():
    from sparsepoly_amd import SparseFactorizationMachineRegressor
    from sparsepoly_amd.concurrent import fit_concurrently, fit_path

    X = sp.random(30, 8, density=0.3, format="csr", random_state=0)
    y = np.arange(30.0)
    est = SparseFactorizationMachineRegressor(max_iter=1)
    with pytest.raises(ValueError, match="same length"):
        fit_path(est, X, y, gamma=[1.0, 2.0], beta=[1.0])
    with pytest.raises(ValueError, match="not a parameter"):
        fit_path(est, X, y, no_such_parameter=[1.0])
    with pytest.raises(ValueError, match="at least one"):
        fit_path(est, X, y)
    with pytest.raises(ValueError, match="one entry per estimator"):
        fit_concurrently([est], [X, X], [y, y])
    with pytest.raises(ValueError, match="max_concurrent"):
        fit_concurrently([est], X, y, max_concurrent=0)
    with pytest.raises(ValueError, match="one GPU"):
        fit_concurrently([SparseFactorizationMachineRegressor(distributed=True)], X, y)
    assert fit_concurrently([], X, y) == []
    assert hasattr(est, "fit_path")


def test_co_tenancy_context_nests_and_restores():
    from sparsepoly_amd import engine as E

    assert E._CO_TENANTS == 1
    with E.co_tenancy(4):
        assert E._CO_TENANTS == 4
        with E.co_tenancy(2):
            assert E._CO_TENANTS == 2
        assert E._CO_TENANTS == 4
    assert E._CO_TENANTS == 1
    with pytest.raises(RuntimeError):
        with E.co_tenancy(3):
            raise RuntimeError("x")
    assert E._CO_TENANTS == 1


def test_hardware_queue_default_is_set_before_the_runtime_starts():
    import sparsepoly_amd._capi  # noqa: F401

    assert int(os.environ["GPU_MAX_HW_QUEUES"]) >= 4


def test_shared_schedule_is_computed_once_and_installed_by_the_others():
    import threading

    from sparsepoly_amd import engine as E

    calls = {"compute": 0, "install": 0}
    lock = threading.Lock()

    def compute():
        with lock:
            calls["compute"] += 1
        return "order", "schedule"

    def install(s):
        assert s == "schedule"
        with lock:
            calls["install"] += 1
        return "order"

    # outside a sharing context: everybody computes
    assert E.shared_schedule(("k",), compute, install) == "order"
    assert calls == {"compute": 1, "install": 0}
    out = []
    with E.co_tenancy(4, share_schedules=True):
        th = [threading.Thread(target=lambda: out.append(E.shared_schedule(("k",), compute, install)))
              for _ in range(4)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert E.shared_schedule(("other",), compute, install) == "order"   # another key: computed
    assert out == ["order"] * 4
    assert calls == {"compute": 3, "install": 3}
    assert E._SHARED_SCHEDULES is None


def test_shared_schedule_leader_failure_lets_the_followers_compute():
    import threading

    from sparsepoly_amd import engine as E

    state = {"n": 0}
    gate = threading.Event()

    def compute():
        state["n"] += 1
        if state["n"] == 1:
            gate.wait(1.0)
            raise RuntimeError("leader fails")
        return "order", "schedule"

    res = []

    def run():
        try:
            res.append(E.shared_schedule(("k",), compute, lambda s: "installed"))
        except RuntimeError:
            res.append("error")

    with E.co_tenancy(2, share_schedules=True):
        a = threading.Thread(target=run)
        a.start()
        import time
        time.sleep(0.1)
        b = threading.Thread(target=run)
        b.start()
        gate.set()
        a.join()
        b.join()
    assert sorted(res) == ["error", "order"]
