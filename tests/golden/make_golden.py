#!/usr/bin/env python3
"""Generates tests/golden/frame_64x64.npz: the input planes + constant buffers of one seeded 64x64 frame and the
outputs the CPU oracle produces for them at every stage.  The reference ships no golden vectors and cannot be built
or run here (D3D12/HLSL), so these vectors pin the ORACLE'S OWN behaviour (regression) and give the HIP path a
scene-generator-independent target.  Run from the repo root:  python tests/golden/make_golden.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib  # noqa: E402
import scene_util  # noqa: E402
from crychic_renderer_amd._lib import lib  # noqa: E402

W, H, SD, CD = 64, 64, 64, 16
pl = scene_util.cpu_scene(W, H, SD, CD)
p = scene_util.np_planes(pl)
c = pl["consts"]
orc = oracle_lib.load()
scb = oracle_lib.as_oracle_cb(c.ssao_cb, oracle_lib.OrSsaoConstants)
pcb = oracle_lib.as_oracle_cb(c.pass_cb, oracle_lib.OrPassConstants)
out = {k: p[k] for k in ("depth", "normal", "g0", "g1", "g2", "shadow", "cube", "randvec")}
out["normal"] = p["normal"].view(np.uint16)
out["ssao_cb"] = np.frombuffer(bytes(c.ssao_cb), dtype=np.uint8)
out["pass_cb"] = np.frombuffer(bytes(c.pass_cb), dtype=np.uint8)
a = orc.ssao(scb, p["normal"], p["depth"], p["randvec"])
out["ssao"] = a
for i in range(2):
    a = orc.blur(scb, p["normal"], p["depth"], a, True); out["blur%d_h" % i] = a
    a = orc.blur(scb, p["normal"], p["depth"], a, False); out["blur%d_v" % i] = a
r_lit = lib.crychic_pcf_search_radius(SD, 1)
r_int = lib.crychic_pcf_search_radius(SD, 0)
out["pcf_radius"] = np.array([r_lit, r_int], dtype=np.float32)
out["lit_1light_literal"] = orc.deferred_light(pcb, p["g0"], p["g1"], p["g2"], p["depth"], a, p["shadow"], p["cube"], 1, r_lit)
o, rad = orc.deferred_light(pcb, p["g0"], p["g1"], p["g2"], p["depth"], a, p["shadow"], p["cube"], 3, r_int, sky=True, want_radiance=True)
out["lit_3light_intended_sky"] = o
out["radiance_3light_intended_sky"] = rad
out["lit_ssao_off"] = orc.deferred_light(pcb, p["g0"], p["g1"], p["g2"], p["depth"], None, p["shadow"], p["cube"], 1, r_lit)
path = os.path.join(ROOT, "tests", "golden", "frame_64x64.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path), "bytes")
