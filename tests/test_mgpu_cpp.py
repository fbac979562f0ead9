"""A C++ host (no Python, no torch in the process) rendering one frame across GPUs: tests/cpp/mgpu_driver.cpp drives the
veneer's CRYCHIC::Initialize / Update / Draw per rank, CRYCHIC::JoinNode puts the RCCL exchange of the C ABI
(crychic_comm_create / crychic_allgather_frame) behind Draw.  On the one-GPU box only world size 1 can run on RCCL (RCCL
refuses two ranks on one device); the strip arithmetic for N > 1 is covered by the strip-decomposition parity tests and
the exchange plumbing by test_bench_launch.py / test_sharding_gloo.py."""
import os
import subprocess

import numpy as np
import pytest

from test_cpp_veneer import build_driver

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_mgpu_driver_compiles(built_lib):
    assert os.path.exists(build_driver("mgpu_driver"))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["rank", "rank-ragged", "rank-parts", "all"])
def test_cpp_host_single_rank_rccl(built_lib, tmp_path, mode):
    exe = build_driver("mgpu_driver")
    W, H = 256, 144
    d = str(tmp_path)
    if mode == "all":
        cmd = [exe, "all", "1", d, str(W), str(H)]
    else:
        cmd = [exe, "rank", "1", "0", os.path.join(d, "id.bin"), d, str(W), str(H)] + {"rank-ragged": ["ragged"], "rank-parts": ["parts=3"]}.get(mode, [])     # parts=3: crychic_draw_hot_path_shared's
        #                                                                  side-stream exchange, lighting in three row ranges
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    single = np.fromfile(os.path.join(d, "frame_single.bin"), dtype=np.uint8)
    got = np.fromfile(os.path.join(d, "frame_0.bin"), dtype=np.uint8)
    assert single.size == W * H * 4 and np.array_equal(single, got)
    assert single.reshape(H, W, 4)[..., 3].min() == 255 and len(np.unique(single)) > 50      # a real image, not a cleared buffer


@pytest.mark.gpu
def test_cpp_host_reports_comm_errors(built_lib, tmp_path):
    """A second rank that cannot exist on a one-GPU box (device ordinal 1) fails with an exception text, not a crash or a hang."""
    exe = build_driver("mgpu_driver")
    d = str(tmp_path)
    open(os.path.join(d, "id.bin"), "wb").write(bytes(128))
    r = subprocess.run([exe, "rank", "2", "1", os.path.join(d, "id.bin"), d, "64", "48"], capture_output=True, text=True, timeout=120)
    import torch
    if torch.cuda.device_count() < 2:
        assert r.returncode == 1 and "exception" in r.stderr
