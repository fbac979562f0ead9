"""The N > 1 code path of bench.py on a one-GPU box: process group on backend nccl (= RCCL) with a single rank, the
two-slot FrameGather, and the per-slot hipGraph replay.  The gathered frame must equal the directly rendered one
(bench.py checks it and exits 3 otherwise).  Real multi-rank behaviour is covered on CPU by test_sharding_gloo.py."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("graph,inflight", [("on", 1), ("off", 1), ("off", 2)])
def test_bench_gather_path_single_rank_rccl(built_lib, graph, inflight):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-gather", "--graph", graph, "--frames-in-flight", str(inflight), "--steps", "7",
           "--warmup", "3", "--width", "640", "--height", "360", "--shadow-dim", "512", "--cube-dim", "64",
           "--no-cpu-baseline", "--no-producers"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "gathered frame == directly rendered frame: True" in r.stderr, r.stderr[-3000:]
    assert "capture failed" not in r.stderr, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 1 and out["value"] > 0
    assert out["config"]["launch"] == ("hipGraph replay" if graph == "on" else "eager")
    assert out["config"]["frames_in_flight"] == inflight
