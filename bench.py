#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the CRYCHIC hot path (SSAO + 2*blurCount bilateral sweeps + deferred PBR lighting)
on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W            (N > 1: this process starts the N rank processes itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W               (one rank per GPU from a launcher: same code path)

A "step" is one frame: every input plane (G-buffer, view normals, depth, 4 shadow cascades, cubemap, noise) is
already resident in HBM; the step runs crychic_draw_hot_path for this rank's row strip and, for N > 1, the exchange
of the composed RGBA8 strips (strong scaling of ONE frame; the only collective).  The exchange is
crychic_allgather_frame -- RCCL over xGMI behind the C ABI (csrc/comm.cpp), enqueued on the stream that rendered
the strip; torch.distributed (gloo) carries only control traffic (rendezvous id, barriers, the max over ranks).
Workload = BASELINE.json configs[2]: 3840x2160, 3 directional lights, blurCount 4, cascade PCF.  The PCF radius
follows the reference shader as written (Common.hlsl:305 unsigned division => 16 coincident taps); --pcf intended
benches the 2.5-texel variant instead.  Frames run ONE AT A TIME on one stream -- the reference submits every frame to
one direct queue over one G-buffer / ambient-map set (Common/d3dApp.cpp:484-486), and SURVEY.md 8(d) defines the metric
on the wall time of one frame -- so `value` = K frames / the wall time of the K-frame region.  In the same run (N = 1)
the line also carries, all labelled and none of them `value`: the median of per-frame HIP-event times; a throughput leg
with three frames in flight on three streams, each pipeline on its OWN copy of every input plane; the same frame with
the evidently intended 16-tap PCF; and a camera pitched down until every pixel is covered (no sky).

Rank 0 prints ONE JSON line: metric/value/... plus "roofline" (frame level per SURVEY.md 8d, with a per-kernel
break-down measured by HIP events on the launch stream) and "cpu_baseline" (the CPU oracle on a band of the frame).

No launch is ever retried inside a process that has touched the GPU and nothing is re-exec'ed: a failed or stuck
rank ends the run with a non-zero exit code (--timeout).
"""
import argparse
import ctypes as C
import datetime
import hashlib
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
PMC_PROFILE = os.path.join(ROOT, "profiles", "r04_pmc_counters.json")


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--blur-count", type=int, default=4)
    ap.add_argument("--lights", type=int, default=3)
    ap.add_argument("--shadow-dim", type=int, default=4096)
    ap.add_argument("--cube-dim", type=int, default=256)
    ap.add_argument("--pcf", choices=["literal", "intended"], default="literal")
    ap.add_argument("--camera", choices=["reference", "covered"], default="reference",
                    help="covered: the camera pitched down until no pixel is sky (informational: every G-buffer texel is read)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-producers", action="store_true", help="skip the informational timing of the producer passes")
    ap.add_argument("--frames-in-flight", type=int, default=1,
                    help="independent frame pipelines (own stream, own ambient / edge workspace and, at N = 1, own copies of every "
                         "input plane) taking the K frames in turn.  1 (default) = one frame at a time, the metric's definition; "
                         "more = a throughput experiment (at N = 1 the default run reports one as config.throughput_3_in_flight)")
    ap.add_argument("--no-legs", action="store_true", help="skip the informational legs of the N = 1 line (throughput, intended PCF, covered camera)")
    ap.add_argument("--force-gather", action="store_true",
                    help="run the strip exchange even with one rank: exercises the N > 1 code path on a 1-GPU box")
    ap.add_argument("--exchange-parts", type=int, default=1,
                    help="N > 1: crychic_draw_hot_path_shared -- the lighting pass in N row ranges, each range's exchange on a side "
                         "stream while the next is lit (SURVEY 8e); 1 = the strip, then one exchange (default)")
    ap.add_argument("--exchange", choices=["abi", "torch"], default="abi",
                    help="abi: crychic_allgather_frame (RCCL behind the C ABI); torch: torch.distributed nccl all_gather (fallback)")
    ap.add_argument("--partition", choices=["equal", "balanced"], default="balanced",
                    help="N > 1: balanced (default) = cost-aware strip heights, seeded from the depth plane's coverage and re-cut from "
                         "measured strip times, one group of in-place ncclBroadcasts; falls back to equal if the plan cannot be made or "
                         "the gathered frame fails its check.  equal = H/N rows each, one in-place ncclAllGather")
    ap.add_argument("--balance-iters", type=int, default=4, help="measured re-balancing steps of the strip plan before the warm-up")
    ap.add_argument("--covered-weight", type=float, default=3.0, help="cost of a fully covered row pair relative to a sky row pair")
    ap.add_argument("--strip", default="", help="N:R -- time only the row strip rank R of N would render (no exchange): what one rank "
                                                "of an N-GPU run computes per frame, measurable on a 1-GPU box")
    ap.add_argument("--point-lights", type=int, default=0, help="extension (BASELINE configs[4]): n x n point-light grid, e.g. 8")
    ap.add_argument("--dump-scene", default="", help="write the input planes + constants for tools/prof_driver and exit")
    ap.add_argument("--cpu-band-rows", type=int, default=0, help="full-res rows of the CPU baseline band (0 = auto)")
    ap.add_argument("--timeout", type=float, default=900.0, help="seconds after which a rank (and the spawning parent) gives up and "
                                                                 "exits non-zero instead of waiting on a stuck peer")
    ap.add_argument("--dry-run", action="store_true",
                    help="plumbing rehearsal WITHOUT a GPU and without rendering: rank spawning, rendezvous, the strip plan and a gloo "
                         "exchange of synthetic strips; prints a line marked data=dry-run (not a measurement)")
    ap.add_argument("--dry-run-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)   # tests: this rank exits 7 before the rendezvous
    ap.add_argument("--fail-first-check", action="store_true", help=argparse.SUPPRESS)      # tests: the first frame check reports a mismatch
    ap.add_argument("--plan-rehearsal", action="store_true", help=argparse.SUPPRESS)        # tests: N ranks share GPU 0 and render their strips
    #                                                                                         without the exchange (RCCL refuses two ranks on one
    #                                                                                         device): rendezvous, measured strip plan, timing
    return ap.parse_args()


# ---- launching ----------------------------------------------------------------------------------------------------------
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher.  This parent never initialises HIP: it builds the library (hipcc, no
    GPU needed), starts one child per rank with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, relays rank 0's JSON line,
    and exits non-zero as soon as any child does (or after --timeout), ending the others by their exact PIDs."""
    if not args.dry_run:
        from crychic_renderer_amd import build
        build.build(verbose=False)
    port = free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), CRYCHIC_BENCH_LAUNCHER="self-spawn")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    lines = []
    reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout), daemon=True)
    reader.start()
    deadline = time.monotonic() + args.timeout + 30.0
    code = 0
    while True:
        states = [p.poll() for p in procs]
        bad = [(r, s) for r, s in enumerate(states) if s not in (None, 0)]
        if bad:
            code = bad[0][1] if bad[0][1] > 0 else 1
            print("bench.py: rank %d exited with status %d; stopping the other ranks" % bad[0], file=sys.stderr, flush=True)
            break
        if all(s == 0 for s in states):
            break
        if time.monotonic() > deadline:
            code = 124
            print("bench.py: ranks still running after %.0f s; stopping them" % (args.timeout + 30.0), file=sys.stderr, flush=True)
            break
        time.sleep(0.1)
    if code:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(timeout=5.0)
    for ln in lines:
        sys.stdout.write(ln)
    sys.stdout.flush()
    if not code and not any(ln.lstrip().startswith("{") for ln in lines):
        print("bench.py: rank 0 printed no result line", file=sys.stderr, flush=True)
        code = 1
    raise SystemExit(code)


def private_stdout():
    """Native libraries (gloo, RCCL) print progress chatter on fd 1; the contract is ONE JSON line there.  Point fd 1 at
    stderr for the life of the rank process and return a private handle on the real stdout for that line."""
    sys.stdout.flush()
    real = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)
    return real


def arm_watchdog(seconds, rank):
    """A stuck collective (a peer that never joined) cannot be recovered from inside the process: give up loudly."""
    def fire():
        print("bench.py rank %d: no result after %.0f s (stuck exchange or peer?) -- exiting 124" % (rank, seconds), file=sys.stderr, flush=True)
        os._exit(124)
    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t


def init_control_plane(args, world, rank):
    """gloo process group for control traffic only.  Finite timeout: a missing peer raises instead of hanging."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")      # the container hostname may not resolve
    dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=min(args.timeout, 300.0)))
    return dist


# ---- dry run (no GPU, no rendering) ----------------------------------------------------------------------------------------
def dry_run(args, world, rank, result_out):
    """Everything around the kernels, on CPU: rendezvous, strip plan, frames-in-flight slot rotation, the exchange (gloo
    stands in for RCCL) and the verification that every rank ends up with every strip.  Strips carry a synthetic pattern."""
    import torch
    from crychic_renderer_amd import sharding
    dist = init_control_plane(args, world, rank)
    W, H = args.width, args.height
    dev = torch.device("cpu")
    bounds = None
    if args.partition == "balanced":
        depth = torch.full((H, W), 0xFFFFFF, dtype=torch.int32)
        depth[H // 2:] = 0x7FFFFF            # lower half covered
        bounds = sharding.StripBalancer(depth, world, args.covered_weight).bounds()
    gather = sharding.FrameGather(W, H, world, rank, dev, bounds=bounds)
    row0, rows = gather.row0, gather.rows

    def pattern(frame, r):
        r0, rn = bounds[r] if bounds else sharding.strip_rows(H, world, r)
        return r0, rn, (17 * frame + 29 * r + 3) % 251

    dist.barrier()
    t0 = time.perf_counter()
    n = args.warmup + args.steps
    for i in range(n):
        buf = gather.strip_buffer(i)
        buf[row0:row0 + rows] = pattern(i, rank)[2]
        gather.launch(i)
    gather.wait_all()
    dist.barrier()
    dt = time.perf_counter() - t0
    ok = True
    for i in range(max(0, n - gather.SLOTS), n):
        f = gather.frame(i)
        for r in range(world):
            r0, rn, v = pattern(i, r)
            ok = ok and bool((f[r0:r0 + rn] == v).all())
    flag = torch.tensor([0 if ok else 1], dtype=torch.int32)
    dist.all_reduce(flag, op=dist.ReduceOp.MAX)
    tmax = torch.tensor([dt], dtype=torch.float64)
    dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "DRY RUN (no rendering, not a measurement)", "value": None, "unit": "Mpixels/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(float(tmax) / max(n, 1) * 1e3, 4),
                          "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u8", "data": "dry-run",
                          "config": {"workload": "synthetic strips %dx%d" % (W, H), "launcher": os.environ.get("CRYCHIC_BENCH_LAUNCHER", "external"),
                                     "partition": args.partition, "strip_plan": [b[1] for b in bounds] if bounds else None,
                                     "exchange": "gloo stand-in", "exchange_verified": bool(int(flag) == 0)}}), file=result_out, flush=True)
    dist.destroy_process_group()
    if int(flag):
        raise SystemExit("bench.py --dry-run: a rank is missing a peer's strip")


# ---- measurement helpers ----------------------------------------------------------------------------------------------------
def cpu_baseline(planes, args, pcf_radius):
    """The CPU oracle (oracle/, OpenMP over rows, all host cores) on a horizontal band through the middle of the
    same frame; band size is calibrated so the leg takes roughly 10-30 s.  kind = "port": the reference has no CPU
    implementation (D3D12 + HLSL only), the oracle is this repo's literal restatement."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib
    import scene_util
    orc = oracle_lib.load()
    p = scene_util.np_planes(planes)
    consts = planes["consts"]
    scb = oracle_lib.as_oracle_cb(consts.ssao_cb, oracle_lib.OrSsaoConstants)
    pcb = oracle_lib.as_oracle_cb(consts.pass_cb, oracle_lib.OrPassConstants)
    W, H = args.width, args.height
    bc = args.blur_count

    def run_band(rows_full):
        rows_full = max(2, min(H, rows_full) & ~1)
        r0 = ((H - rows_full) // 2) & ~1
        h0, hn = r0 // 2, rows_full // 2
        t0 = time.perf_counter()
        a0 = orc.ssao(scb, p["normal"], p["depth"], p["randvec"], h0, hn)
        cur = a0
        for _ in range(bc):
            cur = orc.blur(scb, p["normal"], p["depth"], cur, True, h0, hn)
            cur = orc.blur(scb, p["normal"], p["depth"], cur, False, h0, hn)
        orc.deferred_light(pcb, p["g0"], p["g1"], p["g2"], p["depth"], cur, p["shadow"], p["cube"], args.lights, pcf_radius,
                           row0=r0, rows=rows_full)
        return rows_full, time.perf_counter() - t0

    rows, dt = run_band(args.cpu_band_rows or 64)
    reps = 1
    if not args.cpu_band_rows:
        target = 15.0
        want = int(rows * target / max(dt, 1e-3))
        if want > rows * 2:
            rows, dt = run_band(min(H, want))
        while rows == H and dt * reps < 3.0 and reps < 16:   # many-core host: the whole frame is short, average a few
            dt = (dt * reps + run_band(H)[1]) / (reps + 1)
            reps += 1
    return {"value": round(rows * W / dt / 1e6, 3), "unit": "Mpixels/s", "cores": int(orc.lib.or_num_threads()), "kind": "port",
            "sample": "oracle (C, OpenMP) SSAO + %d blur sweeps + lighting on a %d-row band (%d x %d px) of the same "
                      "frame, %.1f s%s" % (2 * bc, rows, W, rows, dt, " (mean of %d runs)" % reps if reps > 1 else "")}


def kernel_source_hash():
    """Identifies the hot-path kernels a committed counter profile was taken on: sha256 over their sources + the build flags."""
    from crychic_renderer_amd import build
    h = hashlib.sha256()
    for name in ("blur_tiles.hpp", "devmath.hpp", "gamma_pow.inc", "kernels.hip", "kernels.hpp", "light_core.hpp", "ssao_core.hpp"):
        with open(os.path.join(build.CSRC, name), "rb") as f:
            h.update(name.encode() + b"\0" + f.read())
    h.update(" ".join(build.FLAGS).encode())
    return h.hexdigest()[:16]


def pmc_profile(args, world):
    """Hardware-counter figures per launch from the committed rocprofv3 PMC passes (tools/pmc_passes.sh ->
    profiles/r02_pmc_counters.json).  PMC collection cannot run inside this process, so the figures apply only to the
    exact workload AND the exact kernel sources they were measured on: any mismatch returns None (printed as null)."""
    if world != 1 or args.strip or args.point_lights:
        return None
    try:
        with open(PMC_PROFILE) as f:
            prof = json.load(f)
    except (OSError, ValueError):
        return None
    wl = prof.get("workload", {})
    mine = {"width": args.width, "height": args.height, "blur_count": args.blur_count, "lights": args.lights, "pcf": args.pcf,
            "shadow_dim": args.shadow_dim, "camera": args.camera}
    if any(wl.get(k) != v for k, v in mine.items()) or prof.get("kernel_source_hash") != kernel_source_hash():
        return None
    return prof.get("kernels")


def time_producers(ctx, planes, args, torch, row0, rows):
    """Informational (not part of `value`): the producer passes of the same frame on the device -- 4 shadow cascades,
    normals+depth and G-buffer of the reference's 100-box + grid scene through the HIP rasteriser.
    A rank that lights only rows [row0, row0 + rows) produces its G-buffer for those rows only (depth + normals and the
    shadow cascades stay whole: SSAO taps and shadow lookups of a strip reach outside it)."""
    from crychic_renderer_amd import SceneGeometry, geometry as g
    from crychic_renderer_amd._lib import PassConstants
    import numpy as np
    consts = planes["consts"]
    W, H, SD = args.width, args.height, args.shadow_dim
    geo = SceneGeometry(ctx, g.cascade_scene_items(), g.reference_materials(), g.procedural_textures(64))
    sgeo = SceneGeometry(ctx, g.cascade_scene_items(shadow_layer=True))
    dev = ctx.device
    shadow = torch.zeros((4, SD, SD), dtype=torch.int32, device=dev)
    depth = torch.zeros((H, W), dtype=torch.int32, device=dev)
    normal = torch.zeros((H, W, 4), dtype=torch.float16, device=dev)
    gb = [torch.zeros((H, W, 4), dtype=torch.float32, device=dev) for _ in range(3)]
    cbs = []
    for k in range(4):
        cb = PassConstants()
        cb.ViewProj[:] = list((consts.light_view[k].astype(np.float32) @ consts.light_proj[k].astype(np.float32)).T.reshape(-1))
        cbs.append(cb)

    shadow_planes = [shadow[k] for k in range(4)]

    g_rows = (row0, rows) if rows < H else None

    def run():
        sgeo.DrawSceneToShadowMaps(cbs, shadow_planes)
        geo.DrawNormalsDepthAndGBuffer(consts.pass_cb, normal, gb, depth, g_rows=g_rows)

    run()
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    n = 5
    t_sh = t_cam = 0.0
    for _ in range(n):
        e0.record()
        sgeo.DrawSceneToShadowMaps(cbs, shadow_planes)
        e1.record()
        geo.DrawNormalsDepthAndGBuffer(consts.pass_cb, normal, gb, depth, g_rows=g_rows)
        e2.record()
        torch.cuda.synchronize()
        t_sh += e0.elapsed_time(e1) / n
        t_cam += e1.elapsed_time(e2) / n
    # the rasterised planes and the analytic (ray-cast) planes the hot path is benchmarked on describe the same frame
    cov_agree = float(((depth != 0xFFFFFF) == (planes["depth"] != 0xFFFFFF)).float().mean())
    return {"shadow_4x%d" % SD: round(t_sh, 3), "normals_depth+gbuffer": round(t_cam, 3), "gbuffer_rows": list(g_rows) if g_rows else None,
            "triangles": int(geo.triangles),
            "coverage_agreement_with_analytic_scene": round(cov_agree, 6)}


def dump_scene(d, planes, args, pcf_radius):
    """Raw planes + constant buffers for the torch-free profiling driver (tools/prof_driver.cpp)."""
    os.makedirs(d, exist_ok=True)
    for k in ("depth", "normal", "g0", "g1", "g2", "cube", "randvec"):
        planes[k].cpu().numpy().tofile(os.path.join(d, k + ".bin"))
    for i in range(4):
        planes["shadow"][i].cpu().numpy().tofile(os.path.join(d, "shadow%d.bin" % i))
    c = planes["consts"]
    open(os.path.join(d, "ssao_cb.bin"), "wb").write(bytes(c.ssao_cb))
    open(os.path.join(d, "pass_cb.bin"), "wb").write(bytes(c.pass_cb))
    with open(os.path.join(d, "meta.txt"), "w") as f:
        f.write("%d %d %d %d %d %d %r 0\n" % (args.width, args.height, args.shadow_dim, args.cube_dim, args.blur_count, args.lights,
                                             float(pcf_radius)))
    print("scene dumped to", d)


def baseline_config_label(W, H, args, world):
    """The BASELINE.json config whose size and pass counts this run matches, or a plain statement that it matches none."""
    if args.point_lights:
        return "BASELINE configs[4] (%dx%d + %d point lights)%s" % (W, H, args.point_lights ** 2, "" if (W, H) == (7680, 4320) else " at a non-BASELINE size")
    if (W, H) == (3840, 2160) and args.blur_count == 4:
        return "BASELINE configs[3]" if world > 1 else "BASELINE configs[2]"
    if (W, H) == (1920, 1080) and args.blur_count == 1:
        return "BASELINE configs[1]"
    return "no BASELINE config (custom size / blur count)"


def bench_camera(args):
    from crychic_renderer_amd import scene
    return scene.covered_camera(args.width, args.height) if args.camera == "covered" else scene.default_camera(args.width, args.height)


# ---- the benchmark proper -------------------------------------------------------------------------------------------------------
def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args)                      # never returns
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    watchdog = arm_watchdog(args.timeout, rank)
    result_out = private_stdout()
    if args.dry_run:
        if rank == args.dry_run_fail_rank:
            raise SystemExit(7)
        dry_run(args, world, rank, result_out)
        watchdog.cancel()
        return

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product has no CPU path); --dry-run rehearses the plumbing without one")
    if args.plan_rehearsal:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d: LOCAL_RANK %d but only %d GPU(s) visible" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_gather
    dist = init_control_plane(args, world, rank) if use_dist else None

    from crychic_renderer_amd import build
    if rank == 0:
        build.build(verbose=False)
    if use_dist:
        dist.barrier()
    from crychic_renderer_amd import Context, Crychic, scene
    from crychic_renderer_amd._lib import lib
    from crychic_renderer_amd import sharding

    W, H = args.width, args.height
    ctx = Context(local_rank)
    dev = ctx.device
    planes = scene.make_scene(W, H, shadow_dim=args.shadow_dim, cube_dim=args.cube_dim, device=str(dev),
                              consts=scene.Constants(W, H, args.shadow_dim, cam=bench_camera(args)))
    pcf_radius = lib.crychic_pcf_search_radius(args.shadow_dim, 1 if args.pcf == "literal" else 0)

    def new_app(scene_planes=None, own_inputs=False, radius=None):
        """A renderer over `scene_planes` (default: the benchmark scene).  own_inputs: on private copies of every input plane,
        so that pipelines running side by side cannot share cache fills of the same addresses."""
        pl = planes if scene_planes is None else scene_planes
        if own_inputs:
            pl = {k: (v.clone() if isinstance(v, torch.Tensor) else v) for k, v in pl.items()}
        a = Crychic(ctx, W, H, pl["randvec"], pl["cube"], shadow_dim=args.shadow_dim)
        a.load_scene(pl)
        a.blurCount, a.numDirLights, a.pcfSearchRadius = args.blur_count, args.lights, pcf_radius if radius is None else radius
        if args.point_lights:
            a.set_point_lights(scene.point_light_grid(args.point_lights))
        return a

    app = new_app()
    if args.dump_scene:
        dump_scene(args.dump_scene, planes, args, pcf_radius)
        return

    def all_ranks_ok(ok):
        """True when `ok` holds on every rank (control plane; a no-op without peers)."""
        if not use_dist:
            return bool(ok)
        flag = torch.tensor([0 if ok else 1], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        return int(flag) == 0

    # ---- strip plan -------------------------------------------------------------------------------------------------
    # balanced (default): start from the depth plane's coverage (a covered row costs ~3x a sky row), then let every rank time
    # its own strip for a few frames, share the times and re-cut (sharding.StripBalancer): the frame's cost sits in its lower
    # half, so H/N rows each would leave the sky ranks idle.  Rank 0's plan is broadcast after every step, so the ranks agree
    # on it by construction; if anything in this phase fails on any rank, all ranks fall back to equal strips together.
    # equal: H/N rows per rank, one in-place ncclAllGather.
    bounds = None
    if args.partition == "balanced" and (world > 1 or args.strip or args.force_gather):
        nstrips = int(args.strip.split(":")[0]) if args.strip else world

        def agreed(b):
            """rank 0's plan on every rank (control plane)"""
            if not use_dist:
                return b
            t = torch.tensor([v for rb in b for v in rb], dtype=torch.int64)
            dist.broadcast(t, src=0)
            return [(int(t[2 * r]), int(t[2 * r + 1])) for r in range(nstrips)]

        ok = True
        try:
            balancer = sharding.StripBalancer(planes["depth"], nstrips, args.covered_weight)
            bounds = balancer.bounds()
        except Exception as e:  # noqa: BLE001
            print("bench.py rank %d: no balanced strip plan (%s)" % (rank, e), file=sys.stderr, flush=True)
            ok = False
        if all_ranks_ok(ok):
            bounds = agreed(bounds)
            if world > 1:
                app.mBackBuffer = planes["out"]
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(args.balance_iters):
                    mine = torch.tensor([-1.0], dtype=torch.float64)
                    try:
                        r0_, rn_ = bounds[rank]
                        for _ in range(3):
                            app.Draw(r0_, rn_)
                        e0.record()
                        for _ in range(20):
                            app.Draw(r0_, rn_)
                        e1.record()
                        torch.cuda.synchronize()
                        mine[0] = e0.elapsed_time(e1) / 20.0
                    except Exception as e:  # noqa: BLE001
                        print("bench.py rank %d: strip timing failed (%s)" % (rank, e), file=sys.stderr, flush=True)
                    every = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
                    dist.all_gather(every, mine)
                    if any(float(v) <= 0.0 for v in every):
                        break                                            # keep the last agreed plan
                    bounds = agreed(balancer.update([float(v) for v in every]))
        else:
            bounds = None
            args.partition = "equal (fallback: no balanced plan)"
    # The exchange always starts on the plain plan -- H/N rows per rank, one in-place ncclAllGather -- and moves to the balanced
    # plan (ragged strips, one group of in-place ncclBroadcasts) only after the equal-strip frame has passed its check in this
    # same run; the balanced plan then has to pass the same check or the run goes back to equal strips.
    plan, bounds = bounds, None
    row0, rows = sharding.strip_rows(H, world, rank)
    if args.strip:
        sn, sr = (int(v) for v in args.strip.split(":"))
        row0, rows = plan[sr] if plan else sharding.strip_rows(H, sn, sr)
        bounds = plan
    elif args.plan_rehearsal and plan:        # no exchange to verify: the measured plan is what the ranks render
        bounds = plan
        row0, rows = plan[rank]

    # Frames in flight: 1 by default -- one frame at a time on one stream, the metric's definition (the reference's
    # frames go through one direct queue over one set of targets).  More is a labelled throughput experiment: further pipelines
    # with their own stream, ambient maps, workspace and (N = 1) their own copies of the input planes.
    nflight = max(1, min(args.frames_in_flight, 4))
    apps, streams = [app], [torch.cuda.current_stream(dev)]
    for _ in range(nflight - 1):
        apps.append(new_app(own_inputs=(world == 1)))
        streams.append(torch.cuda.Stream(device=dev))

    # ---- the exchange ------------------------------------------------------------------------------------------------------
    # abi: crychic_allgather_frame, enqueued on the pipeline's own stream (slot k belongs to stream k: stream order keeps a
    # slot's gather ahead of its next lighting pass, no host wait anywhere).  If the communicator cannot be created on every
    # rank, all ranks fall back together to torch.distributed's nccl all_gather (equal strips only).
    exchange, exchange_kind = None, None
    if use_dist and not args.strip and not args.plan_rehearsal:
        if args.exchange == "abi":
            idt = torch.zeros(128, dtype=torch.uint8)
            try:
                if rank == 0:
                    idt = torch.frombuffer(bytearray(sharding.StripExchange.new_unique_id()), dtype=torch.uint8).clone()
                ok = True
            except Exception as e:  # noqa: BLE001
                print("bench.py rank %d: no RCCL rendezvous id (%s)" % (rank, e), file=sys.stderr, flush=True)
                ok = False
            if all_ranks_ok(ok):
                dist.broadcast(idt, src=0)
                try:
                    exchange = sharding.StripExchange(ctx, W, H, world, rank, bytes(idt.numpy().tobytes()), bounds=bounds, slots=nflight)
                    for k in range(nflight):                 # first exchange of every slot, outside the timed region
                        exchange.launch(k, streams[k])
                    exchange.wait_all()
                    ok = True
                except Exception as e:  # noqa: BLE001
                    print("bench.py rank %d: crychic_allgather_frame unavailable (%s)" % (rank, e), file=sys.stderr, flush=True)
                    ok = False
                if all_ranks_ok(ok):
                    exchange_kind = "crychic_allgather_frame (C ABI, RCCL)"
                else:
                    if exchange is not None:
                        exchange.abort()
                    exchange = None
        if exchange is None:
            if plan is not None:
                plan = None
                args.partition = "equal (fallback: no RCCL communicator behind the C ABI)"
            nccl = dist.new_group(backend="nccl", timeout=datetime.timedelta(seconds=min(args.timeout, 300.0)))
            exchange = sharding.FrameGather(W, H, world, rank, dev, group=nccl)
            exchange_kind = "torch.distributed nccl all_gather_into_tensor (fallback)" if args.exchange == "abi" else "torch.distributed nccl all_gather_into_tensor"
    abi = isinstance(exchange, sharding.StripExchange)
    outs = [planes["out"]] + [torch.zeros_like(planes["out"]) for _ in range(nflight - 1)]

    def step(i):
        k = i % nflight
        with torch.cuda.stream(streams[k]):            # launches and the exchange bind to this pipeline's stream
            if exchange is None:
                apps[k].mBackBuffer = outs[k]
                apps[k].Draw(row0, rows)
            elif abi and args.exchange_parts > 1:
                exchange.draw(apps[k], k, args.exchange_parts)
            elif abi:
                apps[k].mBackBuffer = exchange.strip_buffer(k)
                apps[k].Draw(row0, rows)
                exchange.launch(k, streams[k])
            else:
                apps[k].mBackBuffer = exchange.strip_buffer(i)   # waits for the gather that last read this slot
                apps[k].Draw(row0, rows)
                exchange.launch(i)

    def fence():
        if exchange is not None:
            exchange.wait_all()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()                              # coarse (TCP), then tight: a one-word all-reduce on the GPU
            if abi:
                exchange.barrier()
                torch.cuda.synchronize()

    # Set-up, not part of W or K: code objects loaded and clocks ramped before the caller's warm-up count starts to matter
    # (with W = 1 the first timed frames would otherwise run at the idle clock).
    app.mBackBuffer = planes["out"]
    for _ in range(150):
        app.Draw(row0, rows)
    torch.cuda.synchronize()
    if app.blur_chain_timed_out():
        raise SystemExit("bench.py rank %d: the single-launch blur chain reported a timed-out wait (crychic_blur_chain_status)" % rank)

    def warm_and_check():
        """W warm-up frames, then: every rank must hold the same complete frame, and it must be the frame one GPU renders alone
        -- a checksum of the last gathered frame compared across ranks, the gathered frame compared with a full local render
        bit for bit."""
        for i in range(args.warmup):
            step(i)
        fence()
        last = args.warmup - 1
        f = exchange.frame(last % nflight if abi else last)
        chk = int((f.reshape(-1).to(torch.int64) * (torch.arange(f.numel(), device=dev, dtype=torch.int64) % 251 + 1)).sum())
        allchk = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(allchk, torch.tensor([chk], dtype=torch.int64))
        app.mBackBuffer = planes["out"]
        app.Draw(0, H)
        torch.cuda.synchronize()
        exchange_ok = all(int(c) == chk for c in allchk) and bool(torch.equal(f, planes["out"]))
        return exchange_ok, allchk

    exchange_ok = None
    if exchange is not None and args.warmup > 0:
        exchange_ok, allchk = warm_and_check()              # equal strips, in-place ncclAllGather
        if not all_ranks_ok(exchange_ok):
            raise SystemExit("bench.py rank %d: the gathered frame differs between ranks or from the single-GPU frame (checksums %s)"
                             % (rank, [int(c) for c in allchk]))
        if plan is not None and abi:
            # the equal-strip frame is right on every rank: now the balanced plan, under the same check
            bounds = plan
            exchange.set_bounds(bounds)
            row0, rows = bounds[rank]
            ok2, allchk = warm_and_check()
            if args.fail_first_check:
                args.fail_first_check, ok2 = False, False
            if not all_ranks_ok(ok2):
                print("bench.py rank %d: balanced strips failed the frame check (checksums %s); falling back to equal strips"
                      % (rank, [int(c) for c in allchk]), file=sys.stderr, flush=True)
                bounds = None
                exchange.set_bounds(None)
                row0, rows = sharding.strip_rows(H, world, rank)
                args.partition = "equal (fallback: the balanced plan failed the frame check)"
                exchange_ok, allchk = warm_and_check()
                if not all_ranks_ok(exchange_ok):
                    raise SystemExit("bench.py rank %d: the gathered frame differs between ranks or from the single-GPU frame (checksums %s)"
                                     % (rank, [int(c) for c in allchk]))
        elif plan is not None:
            args.partition = "equal (the torch.distributed exchange gathers equal strips only)"
        if abi:
            exchange_kind += ": " + ("one group of in-place ncclBroadcasts" if bounds else "in-place ncclAllGather")
            if args.exchange_parts > 1:
                exchange_kind = "crychic_draw_hot_path_shared (C ABI, RCCL): %d parts, each one group of in-place ncclBroadcasts on a side stream" % args.exchange_parts
    elif args.warmup > 0:
        for i in range(args.warmup):
            step(i)
        fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    dt = time.perf_counter() - t0
    if any(a.blur_chain_timed_out() for a in apps):
        raise SystemExit("bench.py rank %d: the single-launch blur chain reported a timed-out wait inside the timed region" % rank)
    if exchange is None and nflight > 1 and not torch.equal(outs[0][row0:row0 + rows], outs[1][row0:row0 + rows]):
        raise SystemExit("bench.py: the frame pipelines disagree")      # same inputs, same kernels: must be the same bytes
    if use_dist:
        tmax = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax)

    def frame_stats(a, row0_, rows_, n):
        """n frames one at a time on the current stream: (wall ms per frame over the region, median of per-frame HIP-event ms)."""
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for _ in range(5):
            a.Draw(row0_, rows_)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for e0, e1 in evs:
            e0.record()
            a.Draw(row0_, rows_)
            e1.record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t1) / n * 1e3
        per = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
        return wall, per[len(per) // 2]

    def leg(ms, npx_):
        return {"ms_per_frame": round(ms, 4), "Mpixels_per_s": round(npx_ / ms / 1e3, 1),
                "hbm_roofline_frac": round((59 + 14 * max(args.blur_count, 0)) * npx_ / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}

    # ---- per-frame HIP-event times of the same workload (median of >= 50 frames) ----
    app.mBackBuffer = planes["out"]
    nev = max(50, min(args.steps, 200))
    _, frame_ms_median = frame_stats(app, row0, rows, nev)

    # ---- per-pass timing of the same workload (HIP events recorded by the library on the launch stream) ----
    app.set_profiling(True)
    app.mBackBuffer = planes["out"]
    acc = {"ssao_ms": 0.0, "blur_ms": 0.0, "light_ms": 0.0, "total_ms": 0.0}
    nprof = max(5, min(50, args.steps))
    for _ in range(nprof):
        app.Draw(row0, rows)
        t = app.last_pass_times()
        for k in acc:
            acc[k] += t[k] / nprof
    app.set_profiling(False)

    # ---- N = 1 informational legs (labelled; none of them is `value`) ----
    legs = {}
    if world == 1 and not args.strip and exchange is None and not args.no_legs and not args.point_lights:
        nleg = max(50, min(args.steps, 100))      # the legs are measurements in their own right: >= 50 frames whatever --steps is
        # (b, first) the evidently intended PCF: 2.5-texel rotated Poisson disc, 16 distinct taps per cascade (Common.hlsl:305 with float division)
        if args.pcf == "literal":
            a2 = new_app(radius=lib.crychic_pcf_search_radius(args.shadow_dim, 0))
            a2.mBackBuffer = torch.zeros_like(planes["out"])
            wall, med = frame_stats(a2, row0, rows, nleg)
            a2.set_profiling(True)
            a2.Draw(row0, rows)
            lt = a2.last_pass_times()["light_ms"]
            a2.set_profiling(False)
            legs["pcf_intended"] = dict(leg(wall, W * H), frame_ms_median_hipevent=round(med, 4), light_ms=round(lt, 4))
            del a2
        # (d) the cube map with its mip chain bound, as the reference binds snowcube1024.dds' (CRYCHIC.cpp:1148-1151; the file is not in
        # the checkout and the benchmark's cube map is a procedural level 0, so `value` is quoted without a chain): trilinear
        # reflection and sky lookups, level of detail from the pixel quads (light_kernel<.., MIPS>)
        if not args.point_lights:
            from crychic_renderer_amd import geometry
            chain, nlev = geometry.cube_mip_chain(planes["cube"].cpu().numpy())
            a4 = new_app()
            a4.set_cube_map(torch.from_numpy(chain).to(dev), dim=int(planes["cube"].shape[1]), levels=nlev)
            a4.mBackBuffer = torch.zeros_like(planes["out"])
            wall, med = frame_stats(a4, row0, rows, nleg)
            a4.set_profiling(True)
            a4.Draw(row0, rows)
            lt = a4.last_pass_times()["light_ms"]
            a4.set_profiling(False)
            legs["cube_mip_chain"] = dict(leg(wall, W * H), frame_ms_median_hipevent=round(med, 4), light_ms=round(lt, 4), cube_levels=nlev)
            del a4, chain
        # (c, second) a camera pitched down until every pixel is covered: no sky, every G-buffer texel is read
        if args.camera == "reference":
            args.camera = "covered"
            cplanes = scene.make_scene(W, H, shadow_dim=args.shadow_dim, cube_dim=args.cube_dim, device=str(dev),
                                       consts=scene.Constants(W, H, args.shadow_dim, cam=bench_camera(args)))
            args.camera = "reference"
            a3 = new_app(scene_planes=cplanes)
            a3.mBackBuffer = torch.zeros_like(planes["out"])
            wall, med = frame_stats(a3, row0, rows, nleg)
            a3.set_profiling(True)
            acc3 = {"ssao_ms": 0.0, "blur_ms": 0.0, "light_ms": 0.0}
            for _ in range(10):
                a3.Draw(row0, rows)
                t = a3.last_pass_times()
                for k in acc3:
                    acc3[k] += t[k] / 10
            a3.set_profiling(False)
            ccov = float(((cplanes["depth"].to(torch.int64) & 0xFFFFFF) != 0xFFFFFF).float().mean())
            legs["camera_covered"] = dict(leg(wall, W * H), frame_ms_median_hipevent=round(med, 4), covered_pixel_fraction=round(ccov, 4),
                                          pass_ms={k: round(v, 4) for k, v in acc3.items()},
                                          light_hbm_frac=round(52.5 * W * H / (acc3["light_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
            del a3, cplanes
        # (a) throughput: three frames in flight on three streams, every pipeline on its own copy of every input plane (last: it
        # leaves three streams' worth of work behind, and the one-at-a-time legs should not start in its wake)
        if nflight == 1:
            pipes = [app] + [new_app(own_inputs=True) for _ in range(2)]
            pstreams = [torch.cuda.current_stream(dev)] + [torch.cuda.Stream(device=dev) for _ in range(2)]
            pouts = [planes["out"]] + [torch.zeros_like(planes["out"]) for _ in range(2)]
            for phase in range(2):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(nleg * 3 if phase else 9):
                    with torch.cuda.stream(pstreams[i % 3]):
                        pipes[i % 3].mBackBuffer = pouts[i % 3]
                        pipes[i % 3].Draw(row0, rows)
                torch.cuda.synchronize()
                t_thr = (time.perf_counter() - t1) / (nleg * 3) * 1e3
            if not (torch.equal(pouts[0], pouts[1]) and torch.equal(pouts[0], pouts[2])):
                raise SystemExit("bench.py: the frame pipelines disagree")      # same inputs, same kernels: must be the same bytes
            legs["throughput_3_in_flight"] = dict(leg(t_thr, W * H), note="three pipelines on three streams, each with private copies of all input planes")
            del pipes, pouts

    producer_ms = None
    if rank == 0 and not args.no_producers:
        producer_ms = time_producers(ctx, planes, args, torch, row0, rows)

    if rank == 0:
        npx = W * H
        strip_px = W * rows
        covered = float(((planes["depth"].to(torch.int64) & 0xFFFFFF) != 0xFFFFFF).float().mean())
        bc = max(args.blur_count, 0)
        # Algorithmic bytes: SSAO 6.5 B, each blur sweep 7 B, lighting 52.5 B per full-res pixel; shadow maps, cubemap
        # and the depth mask are not counted.  Per-launch figures use the pixels this rank's launch covers.
        frame_bytes = (59 + 14 * bc) * npx
        frame_gbs = frame_bytes / (dt / args.steps) / 1e9
        pmc = pmc_profile(args, world)

        def kernel_row(name, ms, bytes_per_px, pmc_key):
            alg = bytes_per_px * strip_px
            row = {"kernel": name, "ms": round(ms, 4), "algorithmic_MB": round(alg / 1e6, 1),
                   "achieved_GBs": round(alg / (ms * 1e-3) / 1e9, 1) if ms > 0 else None,
                   "hbm_frac": round(alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms > 0 else None}
            if ms > 0 and alg / (ms * 1e-3) / 1e9 > HBM_PEAK_GBS:
                row["note"] = "hbm_frac > 1: algorithmic bytes (SURVEY 8d), not bytes moved -- the exits skip most of them; see hbm_frac_by_traffic"
            c = (pmc or {}).get(pmc_key)
            if c:
                row.update({"traffic_MB": round(c["hbm_bytes_per_launch"] / 1e6, 1),
                            "hbm_frac_by_traffic": round(c["hbm_bytes_per_launch"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms > 0 else None,
                            "traffic_over_algorithmic": round(c["hbm_bytes_per_launch"] / alg, 3),
                            "valu_issue_frac": c.get("valu_issue_frac"), "l2_read_GBs": c.get("l2_read_GBs"),
                            "bound": c.get("bound")})
            return row

        kernels = [kernel_row("SSAO pass (depth_pairs_kernel + ssao_kernel)", acc["ssao_ms"], 6.5, "ssao"),
                   kernel_row("blur, %d sweeps (blur_pair_kernel + blur_replay_chain_kernel: iterations 1..%d in one launch)" % (2 * bc, max(bc - 1, 0)), acc["blur_ms"], 7.0 * 2 * bc, "blur"),
                   kernel_row("light_kernel", acc["light_ms"], 52.5, "light")]
        traffic = None
        if pmc and all(k in pmc for k in ("ssao", "blur", "light")):
            traffic = int(sum(pmc[k]["hbm_bytes_per_launch"] for k in ("ssao", "blur", "light")))
        # The dominant kernel, two ways: by the formula's letter (52.5 B for every pixel of the launch) and on the bytes the launch
        # really needs (a sky pixel reads its 4-byte depth texel and writes 4 bytes, nothing of the G-buffer).
        strip_cov = float(((planes["depth"][row0:row0 + rows].to(torch.int64) & 0xFFFFFF) != 0xFFFFFF).float().mean())
        light_s = acc["light_ms"] * 1e-3
        needed = (strip_cov * 52.5 + (1.0 - strip_cov) * 8.0) * strip_px
        dominant = {"kernel": "light_kernel", "ms": round(acc["light_ms"], 4), "algorithmic_MB": round(52.5 * strip_px / 1e6, 1),
                    "frac": round(52.5 * strip_px / light_s / 1e9 / HBM_PEAK_GBS, 4),
                    "needed_MB_on_shaded_pixels": round(needed / 1e6, 1),
                    "frac_on_shaded_pixels": round(needed / light_s / 1e9 / HBM_PEAK_GBS, 4)}
        frame_s = dt / args.steps
        out = {
            "metric": "Mpixels/s for G-buffer->SSAO+blur->deferred PBR lighting at 4K",
            "value": round(npx * args.steps / dt / 1e6, 2),
            "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic" if not args.plan_rehearsal else "synthetic (PLAN REHEARSAL: all ranks on one GPU, no exchange -- not a measurement)",
            "config": {"workload": "%s: %dx%d, box+grid scene, %d dir lights, 14-tap SSAO + %d-pass bilateral "
                                   "blur + cascade PCF (%s radius), 4x%d^2 D24 shadow maps%s" % (
                                       baseline_config_label(W, H, args, world), W, H, args.lights, args.blur_count, args.pcf, args.shadow_dim,
                                       "" if args.camera == "reference" else ", camera pitched down (no sky)"),
                       "launcher": os.environ.get("CRYCHIC_BENCH_LAUNCHER", "external (torch.distributed.run)" if world > 1 else "direct"),
                       "sharding": ("%s row strips x%d" % (args.partition, world)) if world > 1 else "single GPU",
                       "partition": args.partition if use_dist else None,
                       "exchange": exchange_kind, "exchange_verified": exchange_ok,
                       "frames_in_flight": nflight,          # 1: `value` is K frames one at a time on one stream
                       "frame_ms_median_hipevent": round(frame_ms_median, 4),      # median of per-frame HIP-event times, same workload
                       "strip_rows": rows, "strip_plan": [b[1] for b in bounds] if bounds and world > 1 else None,
                       "strip_only": args.strip or None,
                       "point_lights": args.point_lights * args.point_lights,
                       "covered_pixel_fraction": round(covered, 4),
                       "frame_algorithmic_MB": round(frame_bytes / 1e6, 1),
                       "pass_ms": {k: round(v, 4) for k, v in acc.items()},      # HIP events recorded by the library around each pass
                       # informational legs measured in this same run (N = 1); none of them is `value`
                       "throughput_3_in_flight": legs.get("throughput_3_in_flight"),
                       "pcf_intended": legs.get("pcf_intended"),
                       "camera_covered": legs.get("camera_covered"),
                       "cube_mip_chain": legs.get("cube_mip_chain"),
                       "producer_passes_ms": producer_ms,
                       # what one rank spends on a frame end to end: its producers (shadow cascades whole, G-buffer for its strip) + the step
                       "frame_ms_incl_producers_rank0": (round(dt / args.steps * 1e3 + producer_ms["shadow_4x%d" % args.shadow_dim]
                                                               + producer_ms["normals_depth+gbuffer"], 3) if producer_ms else None)},
            # Frame level per SURVEY.md 8d: algorithmic bytes of the whole frame / measured frame time (N = 1: this GPU's HBM;
            # N > 1: the aggregate over N GPUs against N x peak).  `traffic`: HBM bytes per frame (2*FETCH_SIZE + WRITE_SIZE, gfx950
            # correction) from the COMMITTED rocprofv3 PMC passes named by traffic_source -- counters cannot be collected inside this
            # process -- null unless they were taken on exactly these kernel sources and this workload; frac_by_traffic prices the
            # frame on those bytes instead of the algorithmic ones.
            "roofline": {"scope": "frame", "bound": "hbm", "achieved": round(frame_gbs, 1), "peak": HBM_PEAK_GBS * world,
                         "unit": "GB/s", "frac": round(frame_gbs / (HBM_PEAK_GBS * world), 4), "traffic": traffic,
                         "traffic_source": ("committed profile %s (kernel sources %s)" % (os.path.relpath(PMC_PROFILE, ROOT), kernel_source_hash())) if traffic else None,
                         "frac_by_traffic": round(traffic / frame_s / 1e9 / (HBM_PEAK_GBS * world), 4) if traffic else None,
                         "dominant_kernel": dominant,
                         "kernels": kernels},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(planes, args, pcf_radius)
        print(json.dumps(out), file=result_out, flush=True)
    if use_dist:
        dist.barrier()
        if abi:
            exchange.close()
        dist.destroy_process_group()
    watchdog.cancel()


if __name__ == "__main__":
    main()
