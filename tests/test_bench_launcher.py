"""bench.py's own N-rank launcher (`python bench.py --gpus N` without torchrun): it must never
hang on a rank that dies in set-up.  CPU only: the ranks are stand-in child processes."""
import os
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _child(code):
    return [sys.executable, "-c", code]


def test_first_failing_rank_ends_the_job_within_seconds(capfd):
    """Rank 1 dies during set-up (exit 3) while rank 0 would sit in a collective for ten minutes:
    the launcher returns rank 1's status in well under 30 s, rank 0 is terminated (not left
    holding the GPU), and the ranks' last stderr lines are printed."""
    import bench

    code = ("import os, sys, time\n"
            "r = int(os.environ['RANK'])\n"
            "print('rank', r, 'of', os.environ['WORLD_SIZE'], 'set-up', file=sys.stderr, flush=True)\n"
            "if r == 1:\n"
            "    time.sleep(1.0); print('out of memory (say)', file=sys.stderr, flush=True); sys.exit(3)\n"
            "time.sleep(600)\n")
    t0 = time.time()
    status = bench.launch_ranks(3, [], ndev=3, child_cmd=_child(code))
    took = time.time() - t0
    assert status == 3
    assert took < 30, took
    err = capfd.readouterr().err
    assert "rank 1 exited with status 3" in err and "out of memory (say)" in err
    # no child of ours is left behind
    out = subprocess.run(["ps", "-o", "pid,args", "--ppid", str(os.getpid())], capture_output=True,
                         text=True).stdout
    assert "time.sleep(600)" not in out, out


def test_all_ranks_ok_relays_rank0_line_and_sets_the_rendezvous_env(capfd):
    import bench

    code = ("import os, sys, json\n"
            "r = int(os.environ['RANK'])\n"
            "assert os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0\n"
            "assert os.environ['LOCAL_RANK'] == str(r % 2) and os.environ['SPFM_BENCH_NDEV'] == '2'\n"
            "if r == 0: print(json.dumps({'value': 1.5, 'world': os.environ['WORLD_SIZE']}))\n"
            "else: print('not relayed')\n")
    assert bench.launch_ranks(2, [], ndev=2, child_cmd=_child(code)) == 0
    out = capfd.readouterr().out
    assert out.strip() == '{"value": 1.5, "world": "2"}'


def test_ranks_that_share_a_device_are_told_so(capfd):
    """Fewer devices than ranks: a rehearsal -- host-shm communicator, CU shares, SPFM_DEVICE."""
    import bench

    code = ("import os\n"
            "assert os.environ['SPFM_COMM'] == 'shm' and os.environ['SPFM_DEVICE'] == '0'\n"
            "assert 'prb_groups=' in os.environ['SPFM_OPTS']\n")
    assert bench.launch_ranks(2, [], ndev=1, child_cmd=_child(code)) == 0
    assert "REHEARSAL" in capfd.readouterr().err


def test_deadline_stops_ranks_that_never_end():
    import bench

    t0 = time.time()
    status = bench.launch_ranks(2, [], ndev=2, child_cmd=_child("import time; time.sleep(600)"),
                                deadline_s=1.5)
    assert status == 124 and time.time() - t0 < 20


def test_signal_to_the_launcher_is_forwarded_to_the_ranks(tmp_path):
    """SIGTERM to `python bench.py --gpus 2` (the driver's time limit) must not leave the ranks
    running: the launcher forwards it and exits 128 + 15."""
    marker = tmp_path / "rank_pids"
    child = ("import os, time\n"
             "open(%r, 'a').write(str(os.getpid()) + '\\n')\n"
             "time.sleep(600)\n" % str(marker))
    prog = ("import sys; sys.path.insert(0, %r); import bench\n"
            "sys.exit(bench.launch_ranks(2, [], ndev=2, child_cmd=[sys.executable, '-c', %r]))\n"
            % (ROOT, child))
    pr = subprocess.Popen([sys.executable, "-c", prog], stderr=subprocess.DEVNULL)
    t0 = time.time()
    while time.time() - t0 < 20 and (not marker.exists() or len(marker.read_text().split()) < 2):
        time.sleep(0.1)
    pids = [int(p) for p in marker.read_text().split()]
    assert len(pids) == 2
    pr.terminate()
    assert pr.wait(timeout=20) == 128 + 15
    time.sleep(0.2)
    for pid in pids:
        with pytest.raises(ProcessLookupError):
            os.kill(pid, 0)
