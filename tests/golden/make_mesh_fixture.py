#!/usr/bin/env python3
"""Writes tests/golden/car_mesh.npz (vertices + indices of Models/car.txt as the product's loader parses it) and
tests/golden/c1_skull_256.npz (BASELINE configs[0]: 256x256, skull.txt at scale 0.4 + translate (0,1,0)
[CRYCHIC.cpp:1913], 1 directional light, SSAO off, rendered by the CPU oracle end to end).  The models are data
files of the reference checkout (read-only input); run here, where /root/reference exists:
    python tests/golden/make_mesh_fixture.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib  # noqa: E402
import raster_util  # noqa: E402
from crychic_renderer_amd import geometry as g  # noqa: E402

MODELS = "/root/reference/Models"
v, idx = g.load_mesh_text(os.path.join(MODELS, "car.txt"))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "car_mesh.npz"), vertices=v.view(np.uint8), indices=idx)
print("car:", len(v), "vertices", len(idx) // 3, "triangles")

orc = oracle_lib.load()
out = raster_util.render_c1(orc, os.path.join(MODELS, "skull.txt"))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "c1_skull_256.npz"), rgba8=out["rgba8"], depth=out["depth"],
                    covered=np.array([out["covered"]]), tris=np.array([out["tris"]]))
print("c1: covered %.4f, %d setup triangles" % (out["covered"], out["tris"]))
