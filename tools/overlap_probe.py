"""Experiment (not part of the product): how well do the passes of independent frames overlap on one MI355X?

Modes, each over the bench's 4K workload:
  whole F     F frame pipelines, every frame one crychic_draw_hot_path call on its own stream (bench.py --frames-in-flight F)
  pipeline    every SSAO + blur chain on ONE stream, every lighting pass on ANOTHER, events between them: lighting(n) runs
              beside chain(n + 1)
  split F     (--priorities) every frame = crychic_ssao_compute on a HIGH-priority stream, then crychic_deferred_light on a
              LOW-priority stream behind an event, and the reverse
Prints ms per frame for each (records: profiles/r02_overlap_probe.txt).  Usage: python tools/overlap_probe.py [--steps 200]
"""
import argparse
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--skip-whole", action="store_true")
    ap.add_argument("--priorities", action="store_true", help="also run the split-stream modes with stream priorities")
    args = ap.parse_args()
    import torch
    from crychic_renderer_amd import build
    build.build(verbose=False)
    from crychic_renderer_amd import Context, Crychic, scene
    from crychic_renderer_amd._lib import lib, check
    import bench

    W, H = args.width, args.height
    ctx = Context(0)
    dev = ctx.device
    bargs = argparse.Namespace(camera="reference", width=W, height=H)
    planes = scene.make_scene(W, H, shadow_dim=4096, cube_dim=256, device=str(dev),
                              consts=scene.Constants(W, H, 4096, cam=bench.bench_camera(bargs)))
    pcf = lib.crychic_pcf_search_radius(4096, 1)

    def new_app():
        a = Crychic(ctx, W, H, planes["randvec"], planes["cube"], shadow_dim=4096)
        a.load_scene(planes)
        a.blurCount, a.numDirLights, a.pcfSearchRadius = 4, 3, pcf
        a.mBackBuffer = torch.zeros_like(planes["out"])
        return a

    ref = new_app()
    for _ in range(150):
        ref.Draw()
    torch.cuda.synchronize()

    def run(label, step, n):
        for i in range(20):
            step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            step(i)
        torch.cuda.synchronize()
        print("%-28s %.4f ms / frame" % (label, (time.perf_counter() - t0) / n * 1e3), flush=True)

    for F in (() if args.skip_whole else (1, 2, 3, 4)):
        apps = [new_app() for _ in range(F)]
        streams = [torch.cuda.Stream(device=dev) for _ in range(F)]

        def whole(i, apps=apps, streams=streams, F=F):
            with torch.cuda.stream(streams[i % F]):
                apps[i % F].Draw()
        run("whole F=%d" % F, whole, args.steps)
        for a in apps:
            assert torch.equal(a.mBackBuffer, ref.mBackBuffer)

    # two-stage pipeline: every SSAO + blur chain on ONE stream, every lighting pass on ANOTHER; lighting(n) runs under chain(n + 1)
    shadow_ptrs = (C.c_void_p * 4)(*[planes["shadow"][i].data_ptr() for i in range(4)])
    for slots in (2, 3):
        apps = [new_app() for _ in range(slots)]
        sS, sL = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
        ev_ao = [torch.cuda.Event() for _ in range(slots)]
        ev_lit = [torch.cuda.Event() for _ in range(slots)]

        def pipe(i, apps=apps, sS=sS, sL=sL, ev_ao=ev_ao, ev_lit=ev_lit, slots=slots):
            k = i % slots
            a = apps[k]
            ss, g = a.mSsao, a.mDeferred.mGBuffer
            sS.wait_event(ev_lit[k])
            check(lib.crychic_ssao_compute(ctx.handle, C.byref(a.mSsaoCB), ss.mNormalMap.data_ptr(), a.mDepthStencilBuffer.data_ptr(),
                                           ss.mRandomVectorMap.data_ptr(), ss.mAmbientMap0.data_ptr(), ss.mAmbientMap1.data_ptr(),
                                           ss.mEdge.data_ptr(), W, H, 4, 0, H // 2, sS.cuda_stream))
            ev_ao[k].record(sS)
            sL.wait_event(ev_ao[k])
            check(lib.crychic_deferred_light(ctx.handle, C.byref(a.mMainPassCB), g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(),
                                             a.mDepthStencilBuffer.data_ptr(), ss.mAmbientMap0.data_ptr(), shadow_ptrs, 4096,
                                             a.mCubeMap.data_ptr(), 256, a.mBackBuffer.data_ptr(), None, W, H, 0, H, 3, pcf, int(a.flags),
                                             sL.cuda_stream))
            ev_lit[k].record(sL)
        run("two-stage pipeline, %d slots" % slots, pipe, args.steps)
        for a in apps:
            assert torch.equal(a.mBackBuffer, ref.mBackBuffer), "pipeline differs"
    if not args.priorities:
        return

    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    shadow_ptrs = (C.c_void_p * 4)(*[planes["shadow"][i].data_ptr() for i in range(4)])
    for F in (2, 3, 4):
        for (ps, pl, tag) in ((-1, 0, "ssao high / light low"), (0, -1, "ssao low / light high"), (0, 0, "equal priority")):
            apps = [new_app() for _ in range(F)]
            s_ssao = [torch.cuda.Stream(device=dev, priority=ps) for _ in range(F)]
            s_light = [torch.cuda.Stream(device=dev, priority=pl) for _ in range(F)]
            ev_ao = [torch.cuda.Event() for _ in range(F)]
            ev_lit = [torch.cuda.Event() for _ in range(F)]

            def split(i, apps=apps, s_ssao=s_ssao, s_light=s_light, ev_ao=ev_ao, ev_lit=ev_lit, F=F):
                k = i % F
                a = apps[k]
                ss, g = a.mSsao, a.mDeferred.mGBuffer
                s_ssao[k].wait_event(ev_lit[k])        # the previous frame of this slot still reads ambient0
                check(lib.crychic_ssao_compute(ctx.handle, C.byref(a.mSsaoCB), ss.mNormalMap.data_ptr(), a.mDepthStencilBuffer.data_ptr(),
                                               ss.mRandomVectorMap.data_ptr(), ss.mAmbientMap0.data_ptr(), ss.mAmbientMap1.data_ptr(),
                                               ss.mEdge.data_ptr(), W, H, 4, 0, H // 2, s_ssao[k].cuda_stream))
                ev_ao[k].record(s_ssao[k])
                s_light[k].wait_event(ev_ao[k])
                check(lib.crychic_deferred_light(ctx.handle, C.byref(a.mMainPassCB), g[0].data_ptr(), g[1].data_ptr(), g[2].data_ptr(),
                                                 a.mDepthStencilBuffer.data_ptr(), ss.mAmbientMap0.data_ptr(), shadow_ptrs, 4096,
                                                 a.mCubeMap.data_ptr(), 256, a.mBackBuffer.data_ptr(), None, W, H, 0, H, 3, pcf, int(a.flags),
                                                 s_light[k].cuda_stream))
                ev_lit[k].record(s_light[k])
            run("split F=%d %s" % (F, tag), split, args.steps)
            for a in apps:
                assert torch.equal(a.mBackBuffer, ref.mBackBuffer), "split pipeline differs"


if __name__ == "__main__":
    main()
