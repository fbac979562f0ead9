#!/usr/bin/env python3
"""Generates tests/golden/default_frame_800x600.npz: the frame the reference renders out of the box -- 800 x 600 client area
(Common/d3dApp.h:126-127), Ssao::ComputeSsao(..., 3) (CRYCHIC.cpp:221), the deferred shader's NUM_DIR_LIGHTS = 1
(Common.hlsl:6-8), the PCF radius exactly as Common.hlsl:305 computes it (0), sky layer on (CRYCHIC.cpp:278-279) -- as
the CPU oracle renders it from the deterministic analytic scene (crychic_renderer_amd.scene; the shadow cascades at 512^2
instead of 4096^2 to keep the CPU tier fast: a 4096^2 variant runs live against the oracle on the GPU box).  Holds the
oracle's outputs plus a digest of the inputs (so drift of the scene generator is told apart from drift of the oracle).
The reference holds no golden vectors: this pins the ORACLE.   python tests/golden/make_default_frame.py"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib  # noqa: E402
import scene_util  # noqa: E402

W, H, SD, CD, BLUR, LIGHTS = 800, 600, 512, 64, 3, 1


def inputs_digest(p, c):
    h = hashlib.sha256()
    for k in ("depth", "normal", "g0", "g1", "g2", "shadow", "cube", "randvec"):
        h.update(np.ascontiguousarray(p[k]).tobytes())
    h.update(bytes(c.ssao_cb)); h.update(bytes(c.pass_cb))
    return h.hexdigest()


def render(orc):
    pl = scene_util.cpu_scene(W, H, SD, CD)
    p = scene_util.np_planes(pl)
    c = pl["consts"]
    scb = oracle_lib.as_oracle_cb(c.ssao_cb, oracle_lib.OrSsaoConstants)
    pcb = oracle_lib.as_oracle_cb(c.pass_cb, oracle_lib.OrPassConstants)
    ao = orc.compute_ssao(scb, p["normal"], p["depth"], p["randvec"], BLUR)
    rgba = orc.deferred_light(pcb, p["g0"], p["g1"], p["g2"], p["depth"], ao, p["shadow"], p["cube"], LIGHTS, 0.0, sky=True)
    return pl, p, c, ao, rgba


if __name__ == "__main__":
    pl, p, c, ao, rgba = render(oracle_lib.load())
    path = os.path.join(ROOT, "tests", "golden", "default_frame_800x600.npz")
    np.savez_compressed(path, ao=ao, rgba8=rgba, inputs_sha256=np.frombuffer(inputs_digest(p, c).encode(), dtype=np.uint8))
    print("wrote", path, os.path.getsize(path), "bytes; covered", float(((p["depth"] & 0xFFFFFF) != 0xFFFFFF).mean()))
