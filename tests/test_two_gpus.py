"""The real multi-rank exchange, the day two GPUs are visible (the reference itself is single-adapter, NodeMask 0:
CRYCHIC.cpp:96,105).  Every test here is `-m gpu` and skips when torch.cuda.device_count() < 2 -- on the one-GPU box that is
always (RCCL refuses two ranks on one device), on the driver's 8-GPU node it is the first execution of comm.cpp's
ncclAllGather / ragged ncclBroadcast group and of bench.py's cross-rank frame check with more than one rank.

Each rank is a subprocess with a timeout; a stuck rank ends the test with a non-zero exit (bench.py's watchdog exits 124, the
test's own timeout kills the rest by their exact PIDs), never a hang.  Nothing here reads /root/reference or the oracle: the
single-GPU frame of the same run is the checker, bit for bit."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def gpu_count():
    import torch
    return torch.cuda.device_count()          # counts devices without initialising HIP in this process


def need(n):
    have = gpu_count()
    if have < n:
        pytest.skip("needs %d visible GPUs, found %d" % (n, have))


def run_all(procs, timeout):
    """Wait for every rank; on a timeout or a failed rank end the others by PID.  Returns [(returncode, stdout, stderr)]."""
    import time
    deadline = time.time() + timeout
    out = [None] * len(procs)
    failed = False
    while any(o is None for o in out):
        for i, p in enumerate(procs):
            if out[i] is None and p.poll() is not None:
                so, se = p.communicate()
                out[i] = (p.returncode, so, se)
                failed = failed or p.returncode != 0
        if (failed or time.time() > deadline) and any(o is None for o in out):
            for i, p in enumerate(procs):
                if out[i] is None:
                    p.kill()
                    so, se = p.communicate()
                    out[i] = (-9 if p.returncode is None else p.returncode, so, se + "\n[killed by the test: %s]" % ("a peer failed" if failed else "timeout"))
            break
        time.sleep(0.05)
    return out


@pytest.mark.parametrize("n", [2, 4])
@pytest.mark.parametrize("partition,parts", [("balanced", 1), ("equal", 1), ("balanced", 3)])
def test_bench_two_or_more_ranks_on_rccl(built_lib, n, partition, parts):
    """`python bench.py --gpus N` (self-spawning launcher): equal strips + in-place ncclAllGather -> frame check on every rank ->
    (balanced) measured ragged plan + grouped ncclBroadcasts -> the same check.  bench.py exits non-zero if the gathered frame
    differs between ranks or from the frame one GPU renders alone.  parts = 3: the overlapped exchange (crychic_draw_hot_path_shared)."""
    need(n)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "2", "--width", "640", "--height", "360",
           "--shadow-dim", "512", "--cube-dim", "64", "--partition", partition, "--exchange-parts", str(parts), "--no-cpu-baseline", "--no-producers", "--timeout", "300"]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["value"] > 0 and out["scaling"] == "strong"
    assert out["config"]["exchange_verified"] is True
    assert ("crychic_allgather_frame" if parts == 1 else "crychic_draw_hot_path_shared") in out["config"]["exchange"], out["config"]["exchange"]
    if partition == "balanced" and parts == 1:
        assert "ncclBroadcast" in out["config"]["exchange"] or out["config"]["partition"].startswith("equal (fallback"), out["config"]


@pytest.mark.parametrize("n", [2, 3])
@pytest.mark.parametrize("ragged", [False, True, "parts"])
def test_cpp_host_one_process_per_gpu(built_lib, tmp_path, n, ragged):
    """tests/cpp/mgpu_driver `rank` x N (no Python in the ranks): crychic_comm_unique_id -> file rendezvous -> crychic_comm_create ->
    CRYCHIC::JoinNode -> Draw; every rank's gathered frame == the frame rank 0 rendered alone, byte for byte.  "parts": ragged strips
    with crychic_draw_hot_path_shared's overlapped exchange (lighting in three row ranges, each travelling on the side stream)."""
    need(n)
    from test_cpp_veneer import build_driver
    exe = build_driver("mgpu_driver")
    W, H = 256, 144
    d = str(tmp_path)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([exe, "rank", str(n), str(r), os.path.join(d, "id.bin"), d, str(W), str(H)] + {True: ["ragged"], "parts": ["ragged", "parts=3"]}.get(ragged, []),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) for r in range(n)]
    res = run_all(procs, 300)
    assert all(rc == 0 for rc, _, _ in res), "\n".join("rank %d rc %d: %s %s" % (i, rc, so[-500:], se[-1500:]) for i, (rc, so, se) in enumerate(res))
    single = np.fromfile(os.path.join(d, "frame_single.bin"), dtype=np.uint8)
    assert single.size == W * H * 4 and len(np.unique(single)) > 50
    for r in range(n):
        got = np.fromfile(os.path.join(d, "frame_%d.bin" % r), dtype=np.uint8)
        assert np.array_equal(single, got), "rank %d: %d bytes differ from the single-GPU frame" % (r, int((single != got).sum()))


@pytest.mark.parametrize("n", [2, 4])
def test_cpp_host_one_thread_all_gpus(built_lib, tmp_path, n):
    """tests/cpp/mgpu_driver `all` N: one process owning N GPUs (crychic_comm_create_all / crychic_allgather_frame_all)."""
    need(n)
    from test_cpp_veneer import build_driver
    exe = build_driver("mgpu_driver")
    W, H = 256, 144
    d = str(tmp_path)
    r = subprocess.run([exe, "all", str(n), d, str(W), str(H)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stdout + r.stderr
    single = np.fromfile(os.path.join(d, "frame_single.bin"), dtype=np.uint8)
    for k in range(n):
        assert np.array_equal(single, np.fromfile(os.path.join(d, "frame_%d.bin" % k), dtype=np.uint8)), k


def test_a_stuck_rank_ends_the_run(built_lib):
    """Two ranks under the external launcher form, one of which never starts: the other must exit non-zero within its timeout
    (the watchdog), not hang the node."""
    need(2)
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--width", "320", "--height", "180",
           "--shadow-dim", "256", "--cube-dim", "32", "--no-cpu-baseline", "--no-producers", "--timeout", "40"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=200)
    assert r.returncode != 0
