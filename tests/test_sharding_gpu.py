"""The N > 1 code path of bench.py on a one-GPU box: gloo control plane, the RCCL strip exchange behind the C ABI
(crychic_allgather_frame) with a single rank, frames in flight.  The gathered frame must equal the directly rendered one
(bench.py checks it and exits non-zero otherwise).  Rank spawning and multi-rank plumbing are covered on CPU by
test_bench_launch.py and test_sharding_gloo.py."""
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


@pytest.mark.parametrize("exchange,inflight,partition,parts", [("abi", 1, "equal", 1), ("abi", 2, "equal", 1), ("abi", 2, "balanced", 1), ("torch", 2, "equal", 1),
                                                                ("abi", 1, "equal", 3), ("abi", 2, "balanced", 4)])
def test_bench_gather_path_single_rank_rccl(built_lib, exchange, inflight, partition, parts):
    """bench.py's N > 1 path with one rank: gloo control plane, the RCCL exchange behind the C ABI (or the torch nccl
    fallback), frames in flight; bench.py itself compares the gathered frame with a direct render and exits non-zero on a
    difference (config.exchange_verified).  parts > 1: crychic_draw_hot_path_shared -- the lighting pass in row ranges, each range's
    exchange on the communicator's side stream."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-gather", "--exchange", exchange, "--frames-in-flight", str(inflight),
           "--exchange-parts", str(parts),
           "--partition", partition, "--steps", "7", "--warmup", "3", "--width", "640", "--height", "360", "--shadow-dim", "512", "--cube-dim", "64",
           "--no-cpu-baseline", "--no-producers", "--timeout", "400"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]      # native chatter must not reach stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 1 and out["value"] > 0
    assert out["config"]["exchange_verified"] is True
    assert out["config"]["frames_in_flight"] == inflight
    assert ("crychic_allgather_frame" in out["config"]["exchange"]) == (exchange == "abi" and parts == 1), out["config"]["exchange"]
    assert ("crychic_draw_hot_path_shared" in out["config"]["exchange"]) == (parts > 1), out["config"]["exchange"]


def test_bench_balanced_plan_falls_back_to_equal(built_lib):
    """The default (balanced) strip plan whose gathered frame fails its check: every rank switches to equal strips on the same
    communicator and the check runs again (the failure is injected; one rank on RCCL)."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-gather", "--fail-first-check", "--steps", "5", "--warmup", "2",
           "--width", "640", "--height", "360", "--shadow-dim", "512", "--cube-dim", "64", "--no-cpu-baseline", "--no-producers", "--timeout", "400"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert out["config"]["exchange_verified"] is True
    assert out["config"]["partition"].startswith("equal (fallback"), out["config"]["partition"]
    assert "falling back to equal strips" in r.stderr


@pytest.mark.parametrize("n", [2, 3])
def test_bench_measured_strip_plan_multi_rank(built_lib, n):
    """The part of an N > 1 run that needs neither several GPUs nor RCCL, with real ranks: self-spawn, gloo rendezvous, the
    balanced strip plan re-cut from the strip times the ranks measure and share, rank 0's plan adopted by all, K timed frames,
    max over ranks.  The ranks share GPU 0 and skip the exchange (RCCL refuses two ranks on one device): --plan-rehearsal."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--plan-rehearsal", "--steps", "5", "--warmup", "2", "--width", "640",
           "--height", "360", "--shadow-dim", "512", "--cube-dim", "64", "--no-cpu-baseline", "--no-producers", "--timeout", "400"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    out = json.loads(lines[0])
    cfg = out["config"]
    assert out["n_gpus"] == n and "REHEARSAL" in out["data"] and cfg["partition"] == "balanced" and cfg["launcher"] == "self-spawn"
    plan = cfg["strip_plan"]
    assert len(plan) == n and sum(plan) == 360 and all(p >= 2 and p % 2 == 0 for p in plan)
    # (which strip ends up taller is not asserted: at this size a strip is all launch overhead and the ranks share one GPU)
    assert out["value"] > 0 and out["ms_per_step"] > 0


def test_allgather_frame_abi_single_rank(built_lib):
    """crychic_comm_* / crychic_allgather_frame through ctypes with one rank: communicator from a rendezvous id, equal and
    explicit bounds, the stream-ordered barrier, error paths (CRYCHIC_E_COMM / INVALID_ARG never crash)."""
    import ctypes as C
    import torch
    from crychic_renderer_amd import Context, sharding
    lib = built_lib.lib
    ctx = Context(0)
    W, H = 64, 48
    ex = sharding.StripExchange(ctx, W, H, 1, 0, sharding.StripExchange.new_unique_id(), slots=2)
    assert (ex.row0, ex.rows) == (0, H) and lib.crychic_comm_size(ex.handle) == 1 and lib.crychic_comm_rank(ex.handle) == 0
    frame = torch.randint(0, 255, (H, W, 4), dtype=torch.uint8, device=ctx.device)
    ex.strip_buffer(0).copy_(frame)
    ex.launch(0)
    ex.barrier()
    ex.wait_all()
    assert torch.equal(ex.frame(0), frame)
    # explicit bounds that do not tile the frame are refused before anything is enqueued
    bad = (C.c_uint32 * 2)(0, H - 2)
    assert lib.crychic_allgather_frame(ex.handle, C.c_void_p(frame.data_ptr()), W, H, bad, None) == -1
    assert b"strips cover" in lib.crychic_last_error()
    assert lib.crychic_allgather_frame(None, C.c_void_p(frame.data_ptr()), W, H, None, None) == -1
    # the overlapped form refuses a part count outside 1 .. 8, a null communicator and a strip that is not the rank's own
    from crychic_renderer_amd._lib import FrameDesc, SsaoConstants, PassConstants
    f = FrameDesc()
    f.W, f.H, f.row0, f.rows = W, H, 0, H
    for parts in (0, 9):
        assert lib.crychic_draw_hot_path_shared(ex.handle, C.byref(SsaoConstants()), C.byref(PassConstants()), C.byref(f), None, parts, None) == -1
        assert b"nparts" in lib.crychic_last_error()
    assert lib.crychic_draw_hot_path_shared(None, C.byref(SsaoConstants()), C.byref(PassConstants()), C.byref(f), None, 2, None) == -1
    f.rows = H - 2
    assert lib.crychic_draw_hot_path_shared(ex.handle, C.byref(SsaoConstants()), C.byref(PassConstants()), C.byref(f), None, 2, None) == -1
    assert b"is not rank 0's strip" in lib.crychic_last_error()
    # a rank outside the communicator size is refused without touching RCCL
    h = C.c_void_p()
    assert lib.crychic_comm_create(ctx.handle, 2, 2, (C.c_uint8 * 128)(), C.byref(h)) == -1 and not h.value
    ex.close()
    # single-process form (ncclCommInitAll) with the one visible GPU
    comms = (C.c_void_p * 1)()
    ctxs = (C.c_void_p * 1)(ctx.handle)
    built_lib.check(lib.crychic_comm_create_all(ctxs, 1, comms))
    frames = (C.c_void_p * 1)(frame.data_ptr())
    built_lib.check(lib.crychic_allgather_frame_all(comms, 1, frames, W, H, None, None))
    torch.cuda.synchronize()
    lib.crychic_comm_destroy(comms[0])


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
    assert "traffic" in rf and rf["achieved"] > 0 and rf["scope"] == "frame"
    assert rf["traffic"] is None                      # not the workload the committed PMC passes were taken on
    assert [k["kernel"].split()[0].rstrip(",") for k in rf["kernels"]] == ["SSAO", "blur", "light_kernel"]
    assert rf["traffic_source"] is None and rf["frac_by_traffic"] is None
    dk = rf["dominant_kernel"]
    assert dk["kernel"] == "light_kernel" and 0 < dk["frac_on_shaded_pixels"] <= dk["frac"] < 1
    # N = 1 default: one frame at a time, with the labelled legs of the same run beside it
    cfg = out["config"]
    assert cfg["frames_in_flight"] == 1 and cfg["frame_ms_median_hipevent"] > 0
    for name in ("throughput_3_in_flight", "pcf_intended", "camera_covered", "cube_mip_chain"):
        assert cfg[name]["ms_per_frame"] > 0 and cfg[name]["Mpixels_per_s"] > 0 and 0 < cfg[name]["hbm_roofline_frac"] < 1, name
    assert cfg["camera_covered"]["covered_pixel_fraction"] > 0.999 and cfg["pcf_intended"]["light_ms"] > 0
    assert cfg["cube_mip_chain"]["cube_levels"] == 7 and cfg["cube_mip_chain"]["light_ms"] > 0          # --cube-dim 64
    assert all(k["ms"] > 0 and k["achieved_GBs"] > 0 for k in rf["kernels"])
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "Mpixels/s" and isinstance(cb["sample"], str)
    assert abs(out["value"] - 640 * 360 / (out["ms_per_step"] * 1e-3) / 1e6) / out["value"] < 0.01
