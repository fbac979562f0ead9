"""Row f2: CRYCHIC::UpdateInstanceData's frustum culling (CRYCHIC.cpp:515-564).  The product restates DirectXCollision's
local-space plane test in float; the oracle states visibility geometrically (eight corners against the six clip planes, in
double).  They must agree on every instance that is not within rounding distance of a frustum plane, and a numpy
restatement gives a third opinion.  Parity unpinned: DirectXCollision itself is not part of the reference checkout."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib


def cameras():
    from crychic_renderer_amd import scene
    from crychic_renderer_amd._lib import Camera
    out = [scene.default_camera(1920, 1080)]
    rng = np.random.default_rng(7)
    for _ in range(12):
        c = Camera()
        c.pos[:] = [float(x) for x in rng.uniform([-30, 0.5, -30], [30, 12, 30])]
        look = rng.standard_normal(3); look /= np.linalg.norm(look)
        c.look[:] = [float(x) for x in look]
        c.up[:] = [0.0, 1.0, 0.0]
        c.fovY = float(rng.uniform(0.3, 1.4)); c.aspect = float(rng.choice([1.0, 16 / 9, 0.6]))
        c.nearZ = float(rng.choice([0.1, 1.0, 5.0])); c.farZ = float(rng.choice([40.0, 100.0, 1000.0]))
        out.append(c)
    return out


def numpy_visible(cam, center, extents, worlds, oracle):
    """Third opinion: clip-space corner test in float64 numpy, view / proj taken from the oracle's constant builder."""
    pcb = oracle_lib.OrPassConstants()
    st = np.zeros((4, 16), np.float32); dirs = np.zeros((3, 3), np.float32); dirs[:, 1] = -1
    ocam = oracle_lib.as_oracle_cb(cam, oracle_lib.OrCamera)
    oracle.lib.or_build_pass_constants(C.addressof(ocam), 64, 64, st.ctypes.data, dirs.ctypes.data, C.addressof(pcb))
    vp = np.array(pcb.ViewProj, dtype=np.float64).reshape(4, 4).T            # stored transposed
    corners = np.array([[sx, sy, sz, 0] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=np.float64)
    corners = corners * np.append(extents, 0.0) + np.append(center, 1.0)
    vis = []
    for w in worlds:
        h = corners @ w.astype(np.float64) @ vp
        s = np.stack([h[:, 2], h[:, 3] - h[:, 2], h[:, 3] - h[:, 0], h[:, 3] + h[:, 0], h[:, 3] - h[:, 1], h[:, 3] + h[:, 1]])
        vis.append(bool((s.max(axis=1) >= 0).all()))
    return np.array(vis)


def test_frustum_cull_matches_oracle_and_numpy(built_lib, oracle):
    from crychic_renderer_amd import geometry as g
    box = g.create_box(1.0, 1.0, 1.0, 3)
    center, extents = g.mesh_bounds(box[0])
    assert np.allclose(center, 0) and np.allclose(extents, 0.5)
    rng = np.random.default_rng(3)
    worlds = []
    for i in range(10):
        for j in range(10):
            worlds.append(g.world_matrix((1.6, 1.6, 1.6), ((-5 + i) * 5.0, 0.8, (-5 + j) * 5.0)).reshape(4, 4).T)
    for _ in range(60):                                   # uniformly scaled boxes anywhere (the DX transform assumes uniform scale)
        s = float(rng.uniform(0.2, 6.0))
        worlds.append(g.world_matrix((s, s, s), tuple(rng.uniform(-60, 60, 3))).reshape(4, 4).T)
    worlds = np.ascontiguousarray(np.stack(worlds), dtype=np.float32)
    n = len(worlds)
    cf = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
    total_culled = 0
    for cam in cameras():
        vis = np.zeros(n, np.uint8)
        got = built_lib.lib.crychic_frustum_cull(C.byref(cam), cf(center), cf(extents), worlds.ctypes.data, n, vis.ctypes.data)
        assert got == int(vis.sum())
        ovis = np.zeros(n, np.uint8); margin = np.zeros(n, np.float64)
        ocam = oracle_lib.as_oracle_cb(cam, oracle_lib.OrCamera)
        assert oracle.lib.or_frustum_cull(C.addressof(ocam), center.ctypes.data, extents.ctypes.data, worlds.ctypes.data, n,
                                          ovis.ctypes.data, margin.ctypes.data) == int(ovis.sum())
        clear = np.abs(margin) > 1e-3                     # not grazing a plane
        assert clear.sum() > n * 0.9
        assert np.array_equal(vis[clear], ovis[clear]), (vis != ovis).nonzero()
        assert np.array_equal(numpy_visible(cam, center, extents, worlds, oracle)[clear], ovis[clear].astype(bool))
        total_culled += n - int(vis.sum())
    assert total_culled > n                               # the cameras really cull something


def test_reference_scene_visible_count(built_lib):
    """Default camera (0, 2, -15) looking +z (CRYCHIC.cpp:46,114): boxes behind the eye or far off to the sides are dropped
    from the instance buffer -- and hence from the shadow pass too -- exactly like UpdateInstanceData does."""
    from crychic_renderer_amd import geometry as g, scene
    cam = scene.default_camera(1920, 1080)
    full = g.cascade_scene_items()
    culled = g.cascade_scene_items(cull_camera=cam)
    assert len(full[0][2]) == 100 and len(culled[1][2]) == 1           # the grid is always in view
    nb = len(culled[0][2])
    assert 40 < nb < 100, nb
    # every surviving box is one of the originals, order preserved
    orig = [bytes(r) for r in full[0][2]]
    kept = [orig.index(bytes(r)) for r in culled[0][2]]
    assert kept == sorted(kept)
    # no box in front of the camera within the view cone was dropped
    for k, inst in enumerate(full[0][2]):
        x, z = inst["World"][3], inst["World"][11]
        if z > -10 and abs(x) < 0.35 * (z + 15):
            assert k in kept, (k, x, z)
