"""Producer passes, CPU tier: the oracle rasteriser against the analytic ray-cast scene, the
product's geometry generators against the oracle's, and the kernels' own bodies (hostsim) against the oracle,
bit for bit, for the shadow / normal-depth / G-buffer passes."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib
import scene_util

REF_MODELS = "/root/reference/Models"


def scene_items_oracle(orc):
    from crychic_renderer_amd import geometry as g
    box = oracle_lib.create_box(orc, 1, 1, 1, 3)
    grid = oracle_lib.create_grid(orc, 20, 30, 60, 40)
    prod = g.cascade_scene_items()
    return [(box[0], box[1], prod[0][2]), (grid[0], grid[1], prod[1][2])]


def test_geometry_generators_match_oracle(built_lib, oracle):
    from crychic_renderer_amd import geometry as g
    for args in ((1.0, 1.0, 1.0, 3), (2.0, 0.5, 1.5, 0), (1.0, 2.0, 3.0, 1)):
        pv, pi = g.create_box(*args)
        ov, oi = oracle_lib.create_box(oracle, *args)
        assert pv.tobytes() == ov.tobytes() and np.array_equal(pi, oi)
    pv, pi = g.create_box(1, 1, 1, 3)
    assert len(pv) == 24 * 48 and len(pi) == 36 * 64          # every subdivision: x4 triangles, unshared vertices
    pv, pi = g.create_grid(20.0, 30.0, 60, 40)
    ov, oi = oracle_lib.create_grid(oracle, 20.0, 30.0, 60, 40)
    assert pv.tobytes() == ov.tobytes() and np.array_equal(pi, oi)
    assert len(pv) == 2400 and len(pi) == 59 * 39 * 6
    assert pv["Pos"][0].tolist() == [-10.0, 0.0, 15.0] and np.allclose(pv["Pos"][-1], [10.0, 0.0, -15.0], atol=1e-5)


@pytest.mark.skipif(not os.path.exists(REF_MODELS), reason="reference models are not on this machine")
def test_mesh_text_loader_matches_oracle(built_lib, oracle):
    from crychic_renderer_amd import geometry as g
    for name, nv, nt in (("skull.txt", 31076, 60339), ("car.txt", 1860, 1850)):
        pv, pi = g.load_mesh_text(os.path.join(REF_MODELS, name))
        ov, oi = oracle_lib.load_mesh_text(oracle, os.path.join(REF_MODELS, name))
        assert len(pv) == nv and len(pi) == 3 * nt
        assert np.array_equal(pi, oi) and pi.max() < nv
        assert pv.tobytes() == ov.tobytes()
        t = pv["TangentU"].astype(np.float64); n = pv["Normal"].astype(np.float64)
        assert np.allclose(np.linalg.norm(t, axis=1), 1.0, atol=1e-5)
        assert np.abs((t * n).sum(axis=1)).max() < 2e-3        # tangent orthogonal to the (nearly unit) normal


def test_oracle_rasteriser_agrees_with_analytic_scene(built_lib, oracle):
    """Two independent producers of the same planes: the ray-cast scene (crychic_renderer_amd.scene) and the triangle
    rasteriser.  Coverage must be identical, depth equal to float precision, normals exact."""
    W, H = 320, 180
    pl = scene_util.cpu_scene(W, H, 256, 32)
    p = scene_util.np_planes(pl)
    cs = pl["consts"]
    items = scene_items_oracle(oracle)
    view = np.array(cs.pass_cb.View, np.float32); vp = np.array(cs.pass_cb.ViewProj, np.float32)
    r = oracle_lib.rasterize(oracle, 1, view, vp, items, None, None, W, H)
    cov_a, cov_r = p["depth"] < 0xFFFFFF, r["depth"] < 0xFFFFFF
    # silhouette pixels may fall either way (the rasteriser snaps vertices to 1/256 px, the ray caster does not)
    assert (cov_a != cov_r).mean() < 5e-4
    both = cov_a & cov_r
    dd = np.abs(p["depth"][both].astype(np.int64) - r["depth"][both].astype(np.int64))
    assert (dd >= 1000).mean() < 1e-2 and np.median(dd) < 300     # 1000 LSB = 6e-5 of the depth range; silhouettes excepted
    nd = np.abs(p["normal"][both][:, :3].astype(np.float32) - r["normal"][both][:, :3].astype(np.float32)).max(axis=1)
    assert (nd > 0).mean() < 1e-2
    assert (r["normal"][~cov_r].astype(np.float32) == np.array([0, 0, 1, 0], np.float32)).all()


@pytest.mark.parametrize("W,H", [(160, 90), (97, 61)])
def test_kernel_bodies_match_oracle_all_passes(built_lib, oracle, hostsim, W, H):
    from crychic_renderer_amd import geometry as g
    pl = scene_util.cpu_scene(W, H, 128, 16)
    cs = pl["consts"]
    items = g.cascade_scene_items()
    mats = g.reference_materials()
    tex = g.procedural_textures(32)
    view = np.array(cs.pass_cb.View, np.float32); vp = np.array(cs.pass_cb.ViewProj, np.float32)
    for mode in (1, 2):
        a = oracle_lib.rasterize(oracle, mode, view, vp, items, mats.view(oracle_lib.MATERIAL_DT), tex, W, H)
        b = hostsim.rasterize(mode, view, vp, items, mats, tex, W, H)
        assert np.array_equal(a["depth"], b["depth"])
        if mode == 1:
            assert np.array_equal(a["normal"].view(np.uint16), b["normal"].view(np.uint16))
        else:
            for k in ("g0", "g1", "g2"):
                assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
            assert len(np.unique(a["g1"][..., 0])) > 20           # textures really modulate the albedo
    # shadow cascade 0 with the shadow PSO's bias (CRYCHIC.cpp:1601-1603)
    lvp = (cs.light_view[0].astype(np.float64) @ cs.light_proj[0].astype(np.float64)).astype(np.float32).T.reshape(-1)
    sitems = g.cascade_scene_items(shadow_layer=True)
    a = oracle_lib.rasterize(oracle, 0, view, lvp, sitems, None, None, 128, 128, 10000, 2.0)
    b = hostsim.rasterize(0, view, lvp, sitems, None, None, 128, 128, 10000, 2.0)
    assert np.array_equal(a["depth"], b["depth"]) and (a["depth"] < 0xFFFFFF).mean() > 0.2
    nob = oracle_lib.rasterize(oracle, 0, view, lvp, sitems, None, None, 128, 128, 0, 0.0)
    cov = a["depth"] < 0xFFFFFF
    assert (a["depth"][cov].astype(np.int64) - nob["depth"][cov].astype(np.int64)).min() >= 9999   # bias pushes depth away


def test_clipping_and_culling(built_lib, oracle, hostsim):
    """A big quad through the near plane (clipped), one behind the camera (rejected), one back-facing (culled) and two
    coplanar overlapping triangles (the earlier one wins under LESS)."""
    W, H = 64, 48
    cs = scene_util.cpu_scene(W, H, 128, 16)["consts"]
    view = np.array(cs.pass_cb.View, np.float32); vp = np.array(cs.pass_cb.ViewProj, np.float32)
    from crychic_renderer_amd import geometry as g
    V = np.zeros(11, g.VERTEX_DT)
    V["Normal"] = (0, 1, 0); V["TangentU"] = (1, 0, 0)
    V["Pos"][:4] = [(-50, 0, -40), (-50, 0, 60), (50, 0, 60), (50, 0, -40)]                 # ground quad crossing the near plane
    V["Pos"][4:7] = [(-1, 3, -30), (1, 3, -30), (0, 5, -30)]                                # behind the camera
    V["Pos"][7:11] = [(-2, 1, 5), (-2, 4, 5), (2, 4, 5), (2, 1, 5)]                         # wall facing the camera at z = 5
    idx = np.array([0, 1, 2, 0, 2, 3, 4, 5, 6, 7, 8, 9, 7, 9, 10, 7, 9, 8, 7, 8, 9], np.uint32)  # last: back-facing + duplicate
    inst = g.make_instances([g.world_matrix()], [0])
    items = [(V, idx, inst)]
    a = oracle_lib.rasterize(oracle, 1, view, vp, items, None, None, W, H)
    b = hostsim.rasterize(1, view, vp, items, None, None, W, H)
    assert np.array_equal(a["depth"], b["depth"]) and np.array_equal(a["normal"].view(np.uint16), b["normal"].view(np.uint16))
    cov = a["depth"] < 0xFFFFFF
    assert cov[H - 1].all()                                        # the clipped ground reaches the bottom row
    assert not cov[0].any()                                        # nothing above the horizon except the wall band
    assert cov[H // 2 - 6:H // 2, W // 2 - 2:W // 2 + 2].all()    # the wall is drawn (front-facing copy)
    assert a["tris"] == b["tris"]


def test_guard_band_clipping(built_lib, oracle, hostsim):
    """Triangles with vertices far outside the viewport are RENDERED, as D3D12 renders them (guard-band clipping; CRYCHIC.cpp:2473
    draws whatever the scene holds), not dropped: a ground quad whose far corners are 10^7 units away, a triangle with one vertex a
    million NDC units to the right, one crossing both the near plane and the band, one wholly outside the band.  Kernel bodies ==
    oracle on every plane, the visible part is covered, and the in-band scene of test_clipping_and_culling is untouched by the
    clipper (its planes equal a render with the huge primitives appended behind everything)."""
    W, H = 96, 64
    cs = scene_util.cpu_scene(W, H, 128, 16)["consts"]
    view = np.array(cs.pass_cb.View, np.float32); vp = np.array(cs.pass_cb.ViewProj, np.float32)
    from crychic_renderer_amd import geometry as g
    V = np.zeros(16, g.VERTEX_DT)
    V["Normal"] = (0, 1, 0); V["TangentU"] = (1, 0, 0)
    V["Pos"][:4] = [(-1e7, 0, -40), (-1e7, 0, 1e7), (1e7, 0, 1e7), (1e7, 0, -40)]          # ground to the horizon and far beyond the sides
    V["Pos"][4:7] = [(-2, 1, 5), (-2, 4, 5), (3e6, 1, 5)]                                   # wall triangle with a vertex 3e6 units to the right
    V["Pos"][7:10] = [(-4e6, 6, 8), (0, 9, 8), (4e6, 6, 8)]                                  # wide sliver above, both ends outside the band
    V["Pos"][10:13] = [(5e6, 1, 20), (5e6, 4, 20), (6e6, 1, 20)]                             # wholly outside (right of the band)
    V["Pos"][13:16] = [(-3e6, 0.5, -16), (0, 0.5, 30), (3e6, 0.5, -16)]                      # crosses the near plane AND the band
    idx = np.array([0, 1, 2, 0, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15], np.uint32)
    items = [(V, idx, g.make_instances([g.world_matrix()], [0]))]
    for mode in (1, 2):
        a = oracle_lib.rasterize(oracle, mode, view, vp, items, g.reference_materials().view(oracle_lib.MATERIAL_DT), None, W, H)
        b = hostsim.rasterize(mode, view, vp, items, g.reference_materials(), None, W, H)
        assert np.array_equal(a["depth"], b["depth"]) and a["tris"] == b["tris"]
        if mode == 1:
            assert np.array_equal(a["normal"].view(np.uint16), b["normal"].view(np.uint16))
        else:
            for k in ("g0", "g1", "g2"):
                assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
    cov = a["depth"] < 0xFFFFFF
    assert cov[H - 1].all() and cov[H // 2 + 2].all()           # the ground reaches the bottom row and the horizon, full width
    assert cov[H // 2 - 6, W // 2 - 3:].all()                     # the wall triangle runs off the right edge
    # the two horizontal primitives (ground y = 0, the big triangle y = 0.5 in front of it) interpolate their height exactly enough
    # after clipping: every pixel of the lower half shows one of the two planes
    y = a["g0"][H // 2 + 4:, :, 1]
    near = np.minimum(np.abs(y), np.abs(y - 0.5))
    assert near.max() < 1e-2 and (np.abs(y - 0.5) < 1e-2).sum() > 200
    assert (np.abs(a["g0"][cov][:, 1]) < 1e-2).sum() > 20          # and the ground itself shows towards the horizon


@pytest.mark.parametrize("W,H", [(160, 90), (97, 61)])
def test_mipped_textures_kernel_bodies_match_oracle(built_lib, oracle, hostsim, W, H):
    """Material textures with a mip chain go through the anisotropic sampler (gsamAnisotropicWrap, CRYCHIC.cpp:2631-2638 --
    the kernel D3D leaves open is defined in raster_core.hpp / or_raster.c): kernel bodies == oracle on the G-buffer planes."""
    from crychic_renderer_amd import geometry as g
    cs = scene_util.cpu_scene(W, H, 128, 16)["consts"]
    items, mats = g.cascade_scene_items(), g.reference_materials()
    tex = [g.box_mips(t) for t in g.procedural_textures(64)]
    tex[1] = tex[1][:3]                                     # a truncated chain: the lod clamps to the last stored level
    view = np.array(cs.pass_cb.View, np.float32); vp = np.array(cs.pass_cb.ViewProj, np.float32)
    a = oracle_lib.rasterize(oracle, 2, view, vp, items, mats.view(oracle_lib.MATERIAL_DT), tex, W, H)
    b = hostsim.rasterize(2, view, vp, items, mats, tex, W, H)
    for k in ("g0", "g1", "g2"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
    plain = oracle_lib.rasterize(oracle, 2, view, vp, items, mats.view(oracle_lib.MATERIAL_DT), [t[0] for t in tex], W, H)
    cov = a["depth"] < 0xFFFFFF
    assert (a["g1"][cov] != plain["g1"][cov]).mean() > 0.2       # the chain is really used (minified ground and boxes)
    assert np.array_equal(a["g0"].view(np.uint32), plain["g0"].view(np.uint32))   # position / metalness do not depend on textures


def test_anisotropic_sampler_picks_sensible_levels(built_lib, oracle):
    """Known answers for the sampler's definition: a texture whose level k is the solid grey 16 k.  Seen on a ground plane that
    recedes to the horizon, the sampled level (= albedo * 255 / 16) is 0 where the texture is magnified, grows monotonically
    (within probe noise) up the screen, and never exceeds log2 of the level-0 footprint of the pixel."""
    from crychic_renderer_amd import geometry as g
    W, H = 128, 96
    cs = scene_util.cpu_scene(W, H, 128, 16)["consts"]
    size = 256
    levels = [np.full((max(1, size >> k), max(1, size >> k), 4), min(255, 16 * k), np.uint8) for k in range(9)]
    grid = g.create_grid(20.0, 30.0, 60, 40)
    mats = g.reference_materials().copy()
    mats["DiffuseAlbedo"][:] = 1.0
    mats["DiffuseMapIndex"][:] = 0; mats["NormalMapIndex"][:] = 1
    items = [(grid[0], grid[1], g.make_instances([g.world_matrix((3, 3, 3))], [0]))]
    view = np.array(cs.pass_cb.View, np.float32); vp = np.array(cs.pass_cb.ViewProj, np.float32)
    r = oracle_lib.rasterize(oracle, 2, view, vp, items, mats.view(oracle_lib.MATERIAL_DT), [levels, None], W, H)
    cov = r["depth"] < 0xFFFFFF
    lvl = r["g1"][..., 0] * 255.0 / 16.0
    col = W // 2
    rows = np.where(cov[:, col])[0]
    prof = lvl[rows, col]                                   # top of the ground (far) first, bottom (near) last
    assert prof[-1] < 1.5 and prof[0] > prof[-1] + 1.0      # near: about level 0-1; far: clearly higher levels
    assert (np.diff(prof) < 0.26).all()                     # going down the screen the level never jumps up
    assert prof.max() <= 8.0 and (lvl[cov] >= -1e-6).all()
    # TexTransform / tiling: the grid's TexC spans [0, 1] over 60 x 90 world units, 256 texels -> a few texels per world unit;
    # at the bottom row one pixel covers well under a texel vertically: magnified, level 0 exactly
    assert abs(prof[-1]) < 1e-6 or prof[-1] < 1.5
