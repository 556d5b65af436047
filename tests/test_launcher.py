"""bench.py --gpus N started bare (no torchrun around it) launches its own ranks: one process per GPU with the
environment torch.distributed.run would give them.  --dry-launch rehearses that on CPU (gloo in RCCL's place): the parent
never imports torch, the children rendezvous on 127.0.0.1, rank 0's JSON line is relayed, a dead rank fails the run."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=e, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("n", [2, 4])
def test_bare_gpus_n_launches_n_ranks(n):
    r = _run(["--gpus", str(n), "--dry-launch"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout  # stdout carries exactly rank 0's JSON line
    out = json.loads(lines[0])
    assert out["world"] == n and out["ranks_seen"] == list(range(1, n + 1))  # every rank joined with its own LOCAL_RANK
    assert out["master"].startswith("127.0.0.1:")


def test_a_dead_rank_fails_the_launch():
    r = _run(["--gpus", "2", "--dry-launch"], env={"CREC_DRY_FAIL_RANK": "1"}, timeout=600)
    assert r.returncode != 0
    assert "rank 1 exited with 3" in r.stderr


def test_under_a_launcher_it_does_not_relaunch():
    """with WORLD_SIZE set (torch.distributed.run's environment) the script is a rank, not a launcher"""
    r = _run(["--gpus", "1", "--dry-launch"], env={"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 0 and json.loads(r.stdout)["world"] == 1


def test_the_drivers_own_launch_line():
    """`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2`:
    the script is a worker of the launcher (whose store it joins as a client), rank 0's line is the only thing on stdout"""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    e = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-launch"],
                       env=e, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["world"] == 2 and out["ranks_seen"] == [1, 2]
