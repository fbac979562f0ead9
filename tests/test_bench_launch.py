"""bench.py's launch logic on CPU (no GPU, no rendering: --dry-run): `python bench.py --gpus N` starts its own rank
processes, the launcher form (torch.distributed.run) takes the same path, exactly ONE JSON line reaches stdout, a rank
that dies or a peer that never arrives ends the run with a non-zero exit code instead of a hang."""
import json
import os
import socket
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
SMALL = ["--dry-run", "--width", "64", "--height", "48", "--steps", "5", "--warmup", "2"]


def clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}


def one_json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("n,partition", [(2, "equal"), (3, "balanced")])
def test_self_spawn_dry_run(n, partition):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--partition", partition] + SMALL, env=clean_env(), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    out = one_json_line(r.stdout)
    assert out["n_gpus"] == n and out["data"] == "dry-run" and out["value"] is None
    assert out["config"]["launcher"] == "self-spawn" and out["config"]["exchange_verified"] is True
    if partition == "balanced":
        assert sum(out["config"]["strip_plan"]) == 48 and len(out["config"]["strip_plan"]) == n


def test_launcher_form_dry_run():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), BENCH, "--gpus", "2"] + SMALL
    r = subprocess.run(cmd, env=clean_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    out = one_json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["config"]["exchange_verified"] is True and out["config"]["launcher"] != "self-spawn"


def test_dead_rank_ends_the_run():
    """Rank 1 exits before the rendezvous: the parent must stop rank 0 (stuck waiting for its peer) and exit non-zero."""
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run-fail-rank", "1"] + SMALL, env=clean_env(), capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and time.monotonic() - t0 < 60
    assert "rank 1 exited with status 7" in r.stderr, r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_missing_peer_times_out():
    """A rank whose peers never start gives up after --timeout (watchdog) with exit code 124."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(clean_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--timeout", "5"] + SMALL, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and time.monotonic() - t0 < 60, (r.returncode, r.stderr[-2000:])


def test_world_size_mismatch_is_refused():
    env = dict(clean_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4"] + SMALL, env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "WORLD_SIZE 2 != --gpus 4" in r.stderr
