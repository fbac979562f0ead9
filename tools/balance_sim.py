#!/usr/bin/env python3
"""One-GPU rehearsal of the N-rank strip balancer: renders every rank's strip in turn, feeds the measured times to
sharding.StripBalancer and reports max / mean strip time per iteration (what bounds the N-GPU frame rate before the
exchange).  usage: tools/balance_sim.py N [iterations] [initial covered-row weight]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from crychic_renderer_amd import Context, Crychic, scene, sharding
from crychic_renderer_amd._lib import lib

N = int(sys.argv[1]); iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5; w0 = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
W, H = 3840, 2160
ctx = Context(0)
planes = scene.make_scene(W, H, shadow_dim=4096, cube_dim=256, device=str(ctx.device))
app = Crychic(ctx, W, H, planes["randvec"], planes["cube"], shadow_dim=4096)
app.load_scene(planes)
app.blurCount, app.numDirLights = 4, 3
app.pcfSearchRadius = lib.crychic_pcf_search_radius(4096, 1)
app.mBackBuffer = planes["out"]


def strip_ms(row0, rows, n=30):
    for _ in range(5):
        app.Draw(row0, rows)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        app.Draw(row0, rows)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


bal = sharding.StripBalancer(planes["depth"], N, w0)
for name, bounds in (("equal", [sharding.strip_rows(H, N, r) for r in range(N)]), ("balanced", bal.bounds())):
    for it in range(iters if name == "balanced" else 1):
        t = [strip_ms(*b) for b in bounds]
        print("%s it%d max %.4f mean %.4f  rows %s" % (name, it, max(t), sum(t) / N, [b[1] for b in bounds]), flush=True)
        if name == "balanced":
            bounds = bal.update(t)
