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


@pytest.mark.parametrize("graph,inflight,partition", [("on", 1, "equal"), ("off", 1, "equal"), ("off", 2, "equal"), ("off", 2, "balanced")])
def test_bench_gather_path_single_rank_rccl(built_lib, graph, inflight, partition):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-gather", "--graph", graph, "--frames-in-flight", str(inflight), "--partition", partition, "--steps", "7",
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


def test_bench_json_contract(built_lib):
    """The one JSON line of bench.py carries every key of the driver's contract, with sane types (small frame, bounded CPU leg)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "5", "--warmup", "2", "--width", "640", "--height", "360",
           "--shadow-dim", "512", "--cube-dim", "64", "--cpu-band-rows", "32", "--no-producers"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    for k, t in (("metric", str), ("value", (int, float)), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int),
                 ("ms_per_step", (int, float)), ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict),
                 ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(out[k], t), (k, out[k])
    assert out["vs_baseline"] is None and out["n_gpus"] == 1 and out["steps"] == 5 and out["warmup"] == 2
    assert out["unit"] == "Mpixels/s" and out["scaling"] == "strong" and out["data"] == "synthetic" and "workload" in out["config"]
    rf = out["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert "traffic" in rf and rf["achieved"] > 0
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "Mpixels/s" and isinstance(cb["sample"], str)
    assert abs(out["value"] - 640 * 360 / (out["ms_per_step"] * 1e-3) / 1e6) / out["value"] < 0.01
