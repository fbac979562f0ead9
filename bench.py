#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the CRYCHIC hot path (SSAO + 2*blurCount bilateral sweeps + deferred PBR lighting)
on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one frame: every input plane (G-buffer, view normals, depth, 4 shadow cascades, cubemap, noise) is
already resident in HBM; the step runs crychic_draw_hot_path for this rank's row strip and, for N > 1, the RCCL
exchange of the composed RGBA8 strips (strong scaling of ONE frame, the only collective).
Consecutive frames are independent, so two frame pipelines (two HIP streams, each with its own ambient / edge
workspace, --frames-in-flight) alternate: `value` is steady-state throughput; config.pass_ms is the latency view
(one frame at a time on one stream).  N > 1: strips are cost-balanced from measured strip times (--partition).
Workload = BASELINE.json configs[2]: 3840x2160, 3 directional lights, blurCount 4, cascade PCF.
The PCF radius follows the reference shader as written (Common.hlsl:305 unsigned division => 16 coincident taps);
--pcf intended benches the 2.5-texel variant instead.

Rank 0 prints ONE JSON line: metric/value/... plus "roofline" (dominant kernel = deferred lighting, measured with
HIP events on the launch stream) and "cpu_baseline" (the CPU oracle timed on a bounded band of the same frame).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-producers", action="store_true", help="skip the informational timing of the producer passes")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="replay the frame's launches from a captured hipGraph (auto = off: on a 1/8 strip the replay measured "
                         "0.085 ms against 0.081 ms for eager launches, which the GPU already pipelines)")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="independent frame pipelines (own stream + own ambient / edge workspace) alternating frames: the head and "
                         "tail of one frame's kernels overlap the other's (measured: -11 %% per whole 4K frame, -27 %% per 1/8 strip).  "
                         "0 = auto: 1 at N = 1 (clean per-kernel durations for the roofline object), 4 for the strips of N > 1 "
                         "(1/8 strip: 0.103 ms with 1, 0.089 with 2, 0.074 with 3, 0.069 with 4)")
    ap.add_argument("--also-two-in-flight", action="store_true",
                    help="N = 1: additionally time the K frames with two frames in flight and report it as an informational field "
                         "(off by default: its co-scheduled kernels would stretch the per-kernel averages of a rocprofv3 trace of the run)")
    ap.add_argument("--force-gather", action="store_true",
                    help="run the strip all-gather (RCCL) even with one rank: exercises the N > 1 code path on a 1-GPU box")
    ap.add_argument("--partition", choices=["balanced", "equal"], default="balanced",
                    help="N > 1: balanced = cost-aware strip heights from the depth plane's coverage + point-to-point strip exchange; "
                         "equal = H/N rows each + one all_gather_into_tensor")
    ap.add_argument("--balance-iters", type=int, default=4, help="measured re-balancing steps of the strip plan before the warm-up")
    ap.add_argument("--covered-weight", type=float, default=3.0, help="cost of a fully covered row pair relative to a sky row pair")
    ap.add_argument("--strip", default="", help="N:R -- time only the row strip rank R of N would render (no gather): what one rank of an "
                                                "N-GPU run computes per frame, measurable on a 1-GPU box")
    ap.add_argument("--point-lights", type=int, default=0, help="extension (BASELINE configs[4]): n x n point-light grid, e.g. 8")
    ap.add_argument("--dump-scene", default="", help="write the input planes + constants for tools/prof_driver and exit")
    ap.add_argument("--cpu-band-rows", type=int, default=0, help="full-res rows of the CPU baseline band (0 = auto)")
    return ap.parse_args()


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


def pmc_traffic(args, world):
    """HBM bytes per light_kernel launch from the committed rocprofv3 PMC passes (profiles/r01_pmc_traffic.json:
    (2*FETCH_SIZE + WRITE_SIZE) * 1024, the gfx950 correction of MI355X_MICROARCH.md).  PMC collection cannot run
    inside this process, so the figure applies only to the exact workload it was measured on; otherwise null."""
    if world != 1 or (args.width, args.height, args.blur_count, args.lights, args.pcf, args.shadow_dim) != (3840, 2160, 4, 3, "literal", 4096):
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            return int(json.load(f)["kernels"]["cry::light_kernel<true>"]["hbm_bytes_per_launch"])
    except (OSError, KeyError, ValueError):
        return None


def time_producers(ctx, planes, args, torch):
    """Informational (not part of `value`): the producer passes of the same frame on the device -- 4 shadow cascades,
    normals+depth and G-buffer of the reference's 100-box + grid scene through the HIP rasteriser."""
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

    def run():
        sgeo.DrawSceneToShadowMaps(cbs, shadow_planes)
        geo.DrawNormalsDepthAndGBuffer(consts.pass_cb, normal, gb, depth)

    run()
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    n = 5
    t_sh = t_cam = 0.0
    for _ in range(n):
        e0.record()
        sgeo.DrawSceneToShadowMaps(cbs, shadow_planes)
        e1.record()
        geo.DrawNormalsDepthAndGBuffer(consts.pass_cb, normal, gb, depth)
        e2.record()
        torch.cuda.synchronize()
        t_sh += e0.elapsed_time(e1) / n
        t_cam += e1.elapsed_time(e2) / n
    # the rasterised planes and the analytic (ray-cast) planes the hot path is benchmarked on describe the same frame
    cov_agree = float(((depth != 0xFFFFFF) == (planes["depth"] != 0xFFFFFF)).float().mean())
    return {"shadow_4x%d" % SD: round(t_sh, 3), "normals_depth+gbuffer": round(t_cam, 3), "triangles": int(geo.triangles),
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


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product has no CPU path)")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_gather
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

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
    planes = scene.make_scene(W, H, shadow_dim=args.shadow_dim, cube_dim=args.cube_dim, device=str(dev))
    app = Crychic(ctx, W, H, planes["randvec"], planes["cube"], shadow_dim=args.shadow_dim)
    app.load_scene(planes)
    app.blurCount, app.numDirLights = args.blur_count, args.lights
    app.pcfSearchRadius = lib.crychic_pcf_search_radius(args.shadow_dim, 1 if args.pcf == "literal" else 0)
    if args.dump_scene:
        dump_scene(args.dump_scene, planes, args, app.pcfSearchRadius)
        return
    if args.point_lights:
        app.set_point_lights(scene.point_light_grid(args.point_lights))

    # ---- strip plan -------------------------------------------------------------------------------------------------
    # equal: H/N rows per rank.  balanced: start from the depth plane's coverage (a covered row costs ~3x a sky row), then
    # let every rank time its own strip for a few frames, all-gather the times and re-cut (sharding.StripBalancer): equal
    # strips leave the ranks that own the ground 1.6x slower than the ones that own the sky.
    bounds = None
    if args.partition == "balanced" and (world > 1 or args.strip or args.force_gather):
        balancer = sharding.StripBalancer(planes["depth"], int(args.strip.split(":")[0]) if args.strip else world, args.covered_weight)
        bounds = balancer.bounds()
    if bounds and world > 1:
        app.mBackBuffer = planes["out"]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(args.balance_iters):
            r0_, rn_ = bounds[rank]
            for _ in range(3):
                app.Draw(r0_, rn_)
            e0.record()
            for _ in range(20):
                app.Draw(r0_, rn_)
            e1.record()
            torch.cuda.synchronize()
            mine = torch.tensor([e0.elapsed_time(e1) / 20.0], device=dev, dtype=torch.float32)
            every = torch.zeros(world, device=dev, dtype=torch.float32)
            dist.all_gather_into_tensor(every, mine)
            bounds = balancer.update([float(v) for v in every.cpu()])      # same inputs, same plan on every rank
    row0, rows = bounds[rank] if bounds and not args.strip else sharding.strip_rows(H, world, rank)
    if args.strip:
        sn, sr = (int(v) for v in args.strip.split(":"))
        row0, rows = bounds[sr] if bounds else sharding.strip_rows(H, sn, sr)
    gather = sharding.FrameGather(W, H, world, rank, dev, bounds=bounds if not args.strip else None) if use_dist else None

    # hipGraph: one graph per back-buffer slot replays the ~10 kernel launches of a frame with a single host call.
    use_graph = args.graph == "on"      # auto = eager: replaying a captured strip measured 5 % slower than launching it eagerly
    graphs = {}
    if use_graph:
        # Captured up front, before any collective is in flight (thread-local capture mode: the RCCL watchdog thread
        # may query events meanwhile).  A failed capture falls back to eager launches of the same kernels.
        try:
            for buf in (gather.render if gather is not None else [planes["out"]]):
                app.mBackBuffer = buf
                app.Draw(row0, rows)              # warm: no lazy allocation inside the capture
                torch.cuda.synchronize()
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    app.Draw(row0, rows)
                graphs[buf.data_ptr()] = g
        except Exception as e:  # noqa: BLE001 -- any capture problem: report and keep going eagerly
            print("bench.py: hipGraph capture failed (%s); launching eagerly" % e, file=sys.stderr, flush=True)
            graphs, use_graph = {}, False
            torch.cuda.synchronize()

    # Frames in flight: consecutive frames are independent, so a second pipeline (own stream, own ambient / edge workspace,
    # same read-only input planes) lets the short kernels of one strip fill the dispatch gaps of the other.
    # auto: one frame at a time at N = 1 (the roofline object below needs kernel durations that are not stretched by a
    # co-running frame, and has to agree with a rocprofv3 trace of this very command), four for the short strips of N > 1
    nflight = args.frames_in_flight or (4 if rows < H else 1)
    nflight = 1 if use_graph else max(1, min(nflight, sharding.FrameGather.SLOTS if gather is not None else 4))
    apps, streams = [app], [torch.cuda.current_stream(dev)]
    for _ in range(nflight - 1):
        a = Crychic(ctx, W, H, planes["randvec"], planes["cube"], shadow_dim=args.shadow_dim)
        a.load_scene(planes)
        a.blurCount, a.numDirLights, a.pcfSearchRadius = app.blurCount, app.numDirLights, app.pcfSearchRadius
        if args.point_lights:
            a.set_point_lights(scene.point_light_grid(args.point_lights))
        apps.append(a)
        streams.append(torch.cuda.Stream(device=dev))
    outs = [planes["out"]] + [torch.zeros_like(planes["out"]) for _ in range(nflight - 1)]

    def draw(k, slot_buffer):
        if use_graph:
            graphs[slot_buffer.data_ptr()].replay()
        else:
            apps[k].mBackBuffer = slot_buffer
            apps[k].Draw(row0, rows)

    def step(i):
        k = i % nflight
        with torch.cuda.stream(streams[k]):            # launches, the exchange's pre-sync and its wait all bind to this stream
            if gather is None:
                draw(k, outs[k])
            else:
                draw(k, gather.strip_buffer(i))        # double-buffered so the exchange of frame i overlaps Draw(i+1)
                gather.launch(i)

    def fence():
        if gather is not None:
            gather.wait_all()
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    # First exchange under guard: if the point-to-point batch of the balanced plan fails on any rank (an exception, not a
    # hang), every rank falls back to equal strips + one all_gather_into_tensor, which uses the collective path only.
    if gather is not None and world > 1 and bounds is not None:
        failed = torch.zeros(1, device=dev, dtype=torch.int32)
        try:
            step(0)
            gather.wait_all()
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            print("bench.py rank %d: strip exchange failed (%s); falling back to equal strips" % (rank, e), file=sys.stderr, flush=True)
            failed.fill_(1)
        dist.all_reduce(failed, op=dist.ReduceOp.MAX)
        if int(failed.item()):
            bounds = None
            row0, rows = sharding.strip_rows(H, world, rank)
            gather = sharding.FrameGather(W, H, world, rank, dev)
            args.partition = "equal (fallback)"

    # Set-up, not part of W or K: code objects loaded and clocks ramped before the caller's warm-up count starts to matter
    # (with W = 1 the first timed frames would otherwise run at the idle clock).
    app.mBackBuffer = planes["out"]
    for _ in range(150):
        app.Draw(row0, rows)
    torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    exchange_ok = None
    if gather is not None and world > 1 and args.warmup > 0:
        # every rank must now hold the same complete frame: compare a checksum of the last gathered frame across ranks
        f = gather.frame(args.warmup - 1)
        chk = (f.reshape(-1).to(torch.int64) * (torch.arange(f.numel(), device=dev, dtype=torch.int64) % 251 + 1)).sum().reshape(1)
        allchk = torch.zeros(world, device=dev, dtype=torch.int64)
        dist.all_gather_into_tensor(allchk, chk)
        exchange_ok = bool((allchk == allchk[0]).all().item()) and bool((f[..., 3] == 255).all().item())
        if not exchange_ok:
            raise SystemExit("bench.py: the ranks disagree on the gathered frame (checksums %s)" % allchk.tolist())
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    dt = time.perf_counter() - t0
    if gather is None and nflight > 1 and not torch.equal(outs[0], outs[1]):
        raise SystemExit("bench.py: the two frame pipelines disagree")      # same inputs, same kernels: must be the same bytes
    if use_dist:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    # ---- informational at N = 1: the same K frames with two frames in flight (not `value`, see --frames-in-flight) ----
    overlapped = None
    if args.also_two_in_flight and world == 1 and not args.strip and nflight == 1 and not use_graph:
        app2 = Crychic(ctx, W, H, planes["randvec"], planes["cube"], shadow_dim=args.shadow_dim)
        app2.load_scene(planes)
        app2.blurCount, app2.numDirLights, app2.pcfSearchRadius = app.blurCount, app.numDirLights, app.pcfSearchRadius
        if args.point_lights:
            app2.set_point_lights(scene.point_light_grid(args.point_lights))
        app2.mBackBuffer = torch.zeros_like(planes["out"])
        app.mBackBuffer = planes["out"]
        pair = [(app, torch.cuda.current_stream(dev)), (app2, torch.cuda.Stream(device=dev))]
        for phase in range(2):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(args.steps if phase else 10):
                a_, st_ = pair[i % 2]
                with torch.cuda.stream(st_):
                    a_.Draw(row0, rows)
            torch.cuda.synchronize()
            t_ov = time.perf_counter() - t1
        if not torch.equal(app2.mBackBuffer, planes["out"]):
            raise SystemExit("bench.py: the two frame pipelines disagree")
        overlapped = {"Mpixels_per_s": round(W * H * args.steps / t_ov / 1e6, 1), "ms_per_frame": round(t_ov / args.steps * 1e3, 4)}

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

    producer_ms = None
    if rank == 0 and not args.no_producers:
        producer_ms = time_producers(ctx, planes, args, torch)

    if rank == 0:
        npx = W * H
        strip_px = W * rows
        # algorithmic bytes of the lighting pass: 48 B G0..G2 + 2 B/4 px ambient + 4 B RGBA8 out = 52.5 B per pixel
        # (depth mask, shadow cascades and cubemap are not counted)
        light_bytes = 52.5 * strip_px
        achieved = light_bytes / (acc["light_ms"] * 1e-3) / 1e9
        frame_bytes = (59 + 14 * args.blur_count) * npx
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
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: %dx%d, box+grid scene, %d dir lights, 14-tap SSAO + %d-pass bilateral "
                                   "blur + cascade PCF (%s radius), 4x%d^2 D24 shadow maps" % (W, H, args.lights, args.blur_count,
                                                                                             args.pcf, args.shadow_dim),
                       "sharding": ("%s row strips x%d + RCCL %s of RGBA8 strips" % (args.partition, world, "point-to-point exchange"
                                    if bounds else "all-gather")) if world > 1 else "single GPU",
                       "exchange_verified": exchange_ok, "frames_in_flight": nflight,
                       "two_frames_in_flight_informational": overlapped,
                       "strip_rows": rows, "strip_plan": [b[1] for b in bounds] if bounds and world > 1 else None,
                       "launch": "hipGraph replay" if use_graph else "eager",
                       "strip_only": args.strip or None,
                       "point_lights": args.point_lights * args.point_lights,
                       "frame_algorithmic_MB": round(frame_bytes / 1e6, 1),
                       "frame_hbm_roofline_frac": round(frame_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS * world, 4)
                       if world == 1 else None,
                       "pass_ms": {k: round(v, 4) for k, v in acc.items()},      # one frame at a time on one stream (latency view)
                       "producer_passes_ms": producer_ms,
                       "full_frame_ms_incl_producers": (round(dt / args.steps * 1e3 + producer_ms["shadow_4x%d" % args.shadow_dim]
                                                              + producer_ms["normals_depth+gbuffer"], 3)
                                                        if producer_ms and world == 1 else None)},
            "roofline": {"kernel": "light_kernel", "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": pmc_traffic(args, world)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(planes, args, app.pcfSearchRadius)
        print(json.dumps(out), flush=True)
    if use_dist:
        if args.force_gather and rank == 0:      # self-check of the gather path: the last gathered frame is the rendered one
            app.mBackBuffer = planes["out"]
            app.Draw(0, H) if world == 1 else None
            torch.cuda.synchronize()
            if world == 1:
                same = bool(torch.equal(gather.frame(args.steps - 1), planes["out"]))
                print("bench.py: gathered frame == directly rendered frame: %s" % same, file=sys.stderr, flush=True)
                if not same:
                    raise SystemExit(3)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
