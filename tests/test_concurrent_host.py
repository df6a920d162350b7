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
    with pytest.raises(ValueError, match="independent fits"):
        fit_concurrently([SparseFactorizationMachineRegressor(distributed=True)], X, y)
    with pytest.raises(ValueError, match="distinct device ids"):
        fit_concurrently([est], X, y, devices=[0, 0])
    with pytest.raises(ValueError, match="distinct device ids"):
        fit_concurrently([est], X, y, devices=[])
    assert fit_concurrently([], X, y) == []
    assert hasattr(est, "fit_path")


def test_co_tenancy_context_nests_and_restores():
    from sparsepoly_amd import engine as E

    def tenants():
        t = E.current_tenancy()
        return 1 if t is None else t.n

    assert tenants() == 1
    with E.co_tenancy(4):
        assert tenants() == 4
        with E.co_tenancy(2):
            assert tenants() == 2
        assert tenants() == 4
    assert tenants() == 1
    with pytest.raises(RuntimeError):
        with E.co_tenancy(3):
            raise RuntimeError("x")
    assert tenants() == 1


def test_overlapping_concurrent_calls_from_different_threads_keep_their_own_shares():
    """Two `co_tenancy` blocks open at the same time in two threads: each thread sees its own
    tenant count (the state is per call, bound to the threads that take part), and a thread that
    takes part in neither sees none."""
    import threading

    from sparsepoly_amd import engine as E

    seen = {}
    inside = threading.Barrier(2)

    def call(name, n):
        with E.co_tenancy(n):
            inside.wait(5)            # both blocks are open now
            seen[name] = E.current_tenancy().n
            inside.wait(5)
        seen[name + "_after"] = E.current_tenancy()

    a = threading.Thread(target=call, args=("a", 4))
    b = threading.Thread(target=call, args=("b", 2))
    a.start()
    b.start()
    a.join()
    b.join()
    assert seen == {"a": 4, "b": 2, "a_after": None, "b_after": None}
    assert E.current_tenancy() is None
    # a worker of a fan-out sees its device, the shared caches are the call's
    root = E.Tenancy(4, share_schedules=True, share_data=True)
    w = root.on_device(3)
    assert (w.n, w.device) == (4, 3) and w.schedules is root.schedules and w.images is root.images


def test_hardware_queues_are_configured_on_demand_not_at_import(monkeypatch):
    """Importing the package leaves GPU_MAX_HW_QUEUES alone; asking for concurrent fits exports it
    while the HIP runtime is not up yet, and warns -- instead of silently running two fits per
    queue -- when it is too late."""
    import warnings

    from sparsepoly_amd import _capi

    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    monkeypatch.setattr(_capi, "hip_initialised", lambda: False)
    assert _capi.ensure_hw_queues(4) == 8 and os.environ["GPU_MAX_HW_QUEUES"] == "8"
    monkeypatch.setenv("GPU_MAX_HW_QUEUES", "16")
    assert _capi.ensure_hw_queues(4) == 16 and os.environ["GPU_MAX_HW_QUEUES"] == "16"
    # runtime already initialised by the host program (any PyTorch user) with the default 4
    monkeypatch.delenv("GPU_MAX_HW_QUEUES", raising=False)
    monkeypatch.setattr(_capi, "hip_initialised", lambda: True)
    with pytest.warns(RuntimeWarning, match="hardware queues"):
        assert _capi.ensure_hw_queues(4) == 4
    assert "GPU_MAX_HW_QUEUES" not in os.environ
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert _capi.ensure_hw_queues(1) == 4          # a single fit needs no warning
        monkeypatch.setenv("GPU_MAX_HW_QUEUES", "8")
        assert _capi.ensure_hw_queues(4) == 8          # enough queues: quiet


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
    with E.co_tenancy(4, share_schedules=True) as ten:
        def member():  # a worker thread of the call: bound to its state
            E.bind_tenancy(ten)
            out.append(E.shared_schedule(("k",), compute, install))

        th = [threading.Thread(target=member) for _ in range(4)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert E.shared_schedule(("other",), compute, install) == "order"   # another key: computed
    assert out == ["order"] * 4
    assert calls == {"compute": 3, "install": 3}
    assert E.current_tenancy() is None


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

    box = {}

    def run():
        E.bind_tenancy(box["ten"])
        try:
            res.append(E.shared_schedule(("k",), compute, lambda s: "installed"))
        except RuntimeError:
            res.append("error")

    with E.co_tenancy(2, share_schedules=True) as ten:
        box["ten"] = ten
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


def test_fan_out_over_devices_binds_every_worker_to_its_device():
    """fit_concurrently(devices=[...]): per device `max_concurrent` worker threads, each bound to
    its device and to the call's shared caches; every estimator is fitted exactly once; an
    estimator's own `device` is not consulted.  Stand-in estimators: no GPU needed."""
    import threading

    from sparsepoly_amd import engine as E
    from sparsepoly_amd.concurrent import fit_concurrently

    seen = []
    lock = threading.Lock()
    gate = threading.Barrier(6)  # all six workers are alive at once

    class Stub(object):
        distributed = False
        warm_start = False
        solver = "pcd"
        device = 7  # must be ignored

        def __init__(self, i):
            self.i = i

        def fit(self, X, y):
            ten = E.current_tenancy()
            try:
                gate.wait(10)
            except threading.BrokenBarrierError:
                pass
            with lock:
                seen.append((self.i, ten.device, ten.n, ten.schedules is not None,
                             ten.images is not None, threading.current_thread().name))
            return self

    ests = [Stub(i) for i in range(12)]
    out = fit_concurrently(ests, "X", "y", max_concurrent=2, devices=[0, 3, 5])
    assert out == ests
    assert sorted(s[0] for s in seen) == list(range(12))
    assert {s[1] for s in seen} == {0, 3, 5}
    assert all(s[2] == 2 and s[3] and s[4] for s in seen)
    # two workers per device
    per_dev = {}
    for s in seen:
        per_dev.setdefault(s[1], set()).add(s[5])
    assert all(len(v) == 2 for v in per_dev.values())
    assert E.current_tenancy() is None
    # without `devices` the workers carry no device: the estimator's own is used
    seen.clear()
    gate = threading.Barrier(2)
    fit_concurrently([Stub(0), Stub(1)], "X", "y", max_concurrent=2)
    assert {s[1] for s in seen} == {None}
