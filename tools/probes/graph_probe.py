#!/usr/bin/env python3
"""Probe (not product): one frame / one strip of an 8-way split, direct launches against hipGraph replay of the same calls.
    python tools/probes/graph_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from crychic_renderer_amd import Context, Crychic, scene, sharding
from crychic_renderer_amd._lib import lib

W, H, SD = 3840, 2160, 4096
ctx = Context(0)
planes = scene.make_scene(W, H, shadow_dim=SD, cube_dim=256, device=str(ctx.device), consts=scene.Constants(W, H, SD))
app = Crychic(ctx, W, H, planes["randvec"], planes["cube"], shadow_dim=SD)
app.load_scene(planes)
app.blurCount, app.numDirLights, app.pcfSearchRadius = 4, 3, lib.crychic_pcf_search_radius(SD, 1)
bal = sharding.StripBalancer(planes["depth"], 8, 3.0).bounds()


def timed(fn, n=300):
    for _ in range(20):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, (r0, rn) in [("frame", (0, H))] + [("strip 8:%d" % k, bal[k]) for k in (0, 3, 5)]:
    direct = timed(lambda: app.Draw(r0, rn))
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        app.Draw(r0, rn)
    replay = timed(g.replay)
    g4 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g4):
        for _ in range(4):
            app.Draw(r0, rn)
    replay4 = timed(g4.replay, 100) / 4
    print("%-10s rows %4d  direct %.4f ms   graph %.4f ms   graph of 4 frames %.4f ms/frame" % (name, rn, direct, replay, replay4), flush=True)
