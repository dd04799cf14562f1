"""bench.py as a launcher: `--gpus N` must start N ranks itself (or fail loudly), never report
n_gpus = 1 from a run that was asked for more.  No GPU is touched here: the child processes are
stand-ins and the device count is patched."""
import io
import json
import os
import subprocess
import sys

import pytest

import bench

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class FakeProc:
    def __init__(self, rank, env, line, rc):
        self.rank, self.env, self.rc = rank, env, rc
        self.stdout = io.BytesIO(line.encode()) if rank == 0 else None
        self.killed = False

    def wait(self, timeout=None):
        return self.rc

    def poll(self):
        return self.rc

    def kill(self):
        self.killed = True


def _patch(monkeypatch, ndev, rcs, line='{"n_gpus": %d}'):
    started = []

    def popen(cmd, env=None, stdout=None):
        r = int(env["RANK"])
        assert cmd[0] == sys.executable and cmd[1].endswith("bench.py")
        p = FakeProc(r, env, "RCCL banner noise\n" + (line % int(env["WORLD_SIZE"])) + "\n", rcs[r])
        started.append(p)
        return p
    monkeypatch.setattr(bench, "visible_gpus", lambda: ndev)
    monkeypatch.setattr(bench.subprocess, "Popen", popen)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    return started


def test_launcher_starts_one_rank_per_gpu(monkeypatch, capsys):
    started = _patch(monkeypatch, ndev=4, rcs=[0, 0, 0, 0])
    rc = bench.main(["--gpus", "4", "--steps", "3", "--warmup", "1", "--scaling", "strong"])
    assert rc == 0 and len(started) == 4
    ports = {p.env["MASTER_PORT"] for p in started}
    assert len(ports) == 1 and int(ports.pop()) > 1024           # one free port, not a fixed default
    for r, p in enumerate(started):
        assert (p.env["RANK"], p.env["LOCAL_RANK"], p.env["WORLD_SIZE"]) == (str(r), str(r), "4")
        assert p.env["MASTER_ADDR"] == "127.0.0.1" and p.env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        assert "QNN_DIST_BACKEND" not in p.env or p.env["QNN_DIST_BACKEND"] == os.environ.get("QNN_DIST_BACKEND")
    out = capsys.readouterr().out.strip().splitlines()
    assert out == ['{"n_gpus": 4}']                              # exactly rank 0's result line


def test_launcher_refuses_more_ranks_than_gpus(monkeypatch, capsys):
    started = _patch(monkeypatch, ndev=1, rcs=[0, 0])
    rc = bench.main(["--gpus", "2"])
    assert rc == 2 and started == []
    cap = capsys.readouterr()
    assert cap.out == "" and "--gpus 2 but this box has 1 GPU" in cap.err


def test_launcher_rehearsal_uses_gloo(monkeypatch, capsys):
    started = _patch(monkeypatch, ndev=1, rcs=[0, 0])
    rc = bench.main(["--gpus", "2", "--rehearse"])
    assert rc == 0 and len(started) == 2
    assert all(p.env["QNN_DIST_BACKEND"] == "gloo" for p in started)


def test_launcher_propagates_a_failed_rank(monkeypatch, capsys):
    started = _patch(monkeypatch, ndev=2, rcs=[0, 3])
    rc = bench.main(["--gpus", "2"])
    assert rc == 3
    assert capsys.readouterr().out == ""                          # no result line from a failed job


def test_rank_rejects_world_size_mismatch(monkeypatch):
    monkeypatch.setenv("WORLD_SIZE", "3")
    monkeypatch.setenv("RANK", "0")
    with pytest.raises(SystemExit) as e:
        bench.main(["--gpus", "2"])
    assert "disagrees with WORLD_SIZE=3" in str(e.value)


def test_real_process_on_a_box_without_enough_gpus_fails_loudly():
    """The documented command, for real: on this (GPU-less) container `--gpus 2` must exit non-zero
    with an explanation on stderr and print no JSON line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box really has two GPUs")
    assert p.returncode != 0 and p.stdout.strip() == ""
    assert "--gpus 2 but this box has" in p.stderr


def _child(code):
    return subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE)


def test_supervise_real_children_failed_rank_ends_the_job_quickly(capsys):
    """REAL child processes: rank 1 exits 3 at once while rank 0 would sit in the rendezvous (a 60 s sleep holding its
    stdout pipe open).  The launcher must come back with 3 within seconds, kill rank 0 and print no JSON."""
    import time
    r0 = _child("import time, sys; sys.stdout.write('partial'); sys.stdout.flush(); time.sleep(60)")
    r1 = subprocess.Popen([sys.executable, "-c", "import sys; sys.exit(3)"])
    t0 = time.time()
    rc = bench.supervise([r0, r1], deadline_s=120.0)
    dt = time.time() - t0
    assert rc == 3 and dt < 5.0, (rc, dt)
    assert r0.poll() is not None                                  # rank 0 was killed, not left behind
    assert capsys.readouterr().out == ""


def test_supervise_real_children_deadline(capsys):
    import time
    r0 = _child("import time; time.sleep(60)")
    r1 = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(60)"])
    t0 = time.time()
    rc = bench.supervise([r0, r1], deadline_s=1.0)
    assert rc == 124 and time.time() - t0 < 8.0
    assert r0.poll() is not None and r1.poll() is not None
    assert capsys.readouterr().out == ""


def test_supervise_real_children_success_relays_last_line(capsys):
    r0 = _child("print('RCCL noise'); print('{\"n_gpus\": 2}')")
    r1 = subprocess.Popen([sys.executable, "-c", "pass"])
    assert bench.supervise([r0, r1], deadline_s=60.0) == 0
    assert capsys.readouterr().out.strip().splitlines() == ['{"n_gpus": 2}']
