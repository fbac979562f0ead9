"""Producer passes on the GPU: the HIP rasteriser through the C ABI against the oracle rasteriser,
bit for bit on every plane, then the whole frame (4 shadow cascades -> normals/depth -> G-buffer -> SSAO -> blur ->
lighting) produced entirely on the device against the all-CPU oracle frame."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib
import raster_util

torch = pytest.importorskip("torch")
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx(built_lib):
    if not torch.cuda.is_available():
        pytest.fail("GPU tests selected but no HIP device is visible")
    from crychic_renderer_amd import Context
    c = Context(0)
    yield c
    c.close()


def device_frame(ctx, built_lib, consts, items, shadow_items, materials, textures, W, H, SD, cube, blur_count, ndl, radius, sky=True):
    from crychic_renderer_amd import Crychic, LIGHT_SKY, SceneGeometry
    dev = ctx.device
    geo = SceneGeometry(ctx, items, materials, textures)
    sgeo = SceneGeometry(ctx, shadow_items)
    app = Crychic(ctx, W, H, torch.from_numpy(consts.randvec.copy()).to(dev), torch.from_numpy(cube).to(dev), shadow_dim=SD)
    app.mMainPassCB, app.mSsaoCB = consts.pass_cb, consts.ssao_cb
    shadow_cb = built_lib.PassConstants()
    cbs = []
    for k in range(4):                                     # DrawSceneToShadowMap: one pass constant slot per cascade
        shadow_cb.ViewProj[:] = list(raster_util.light_viewproj_t(consts, k))
        sgeo.DrawSceneToShadowMap(shadow_cb, app.mShadowMap.mShadowMap[k])
        cb = built_lib.PassConstants(); cb.ViewProj[:] = list(shadow_cb.ViewProj); cbs.append(cb)
    # the fused four-cascade pass must reproduce the four separate passes bit for bit
    fused = [torch.zeros_like(app.mShadowMap.mShadowMap[k]) for k in range(4)]
    sgeo.DrawSceneToShadowMaps(cbs, fused)
    torch.cuda.synchronize()
    for k in range(4):
        assert torch.equal(fused[k], app.mShadowMap.mShadowMap[k]), "cascade %d" % k
    geo.DrawNormalsAndDepth(app.mMainPassCB, app.mSsao.mNormalMap, app.mDepthStencilBuffer)
    geo.DrawGBuffer(app.mMainPassCB, app.mDeferred.mGBuffer, app.mDepthStencilBuffer)
    # the fused pass (one rasterisation, both pixel shaders) must reproduce the two passes bit for bit
    n2 = torch.zeros_like(app.mSsao.mNormalMap); d2 = torch.zeros_like(app.mDepthStencilBuffer)
    g2 = [torch.zeros_like(g) for g in app.mDeferred.mGBuffer]
    geo.DrawNormalsDepthAndGBuffer(app.mMainPassCB, n2, g2, d2)
    torch.cuda.synchronize()
    assert torch.equal(n2.view(torch.int16), app.mSsao.mNormalMap.view(torch.int16)) and torch.equal(d2, app.mDepthStencilBuffer)
    for a, b in zip(g2, app.mDeferred.mGBuffer):
        assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    # Strip-limited producers (a rank of an N-GPU frame): inside the rows every plane equals the unscissored pass bit for bit,
    # G-buffer texels outside the rows are left untouched; the fused form still fills depth + normals for the whole frame.
    for (r0, rn) in ((0, H), (H // 4 & ~1, H // 2 & ~1), (H - 2, 2)):
        poison = 0x7FC01234
        n3 = torch.zeros_like(n2); d3 = torch.zeros_like(d2)
        g3 = [torch.full_like(g, float("nan")).view(torch.int32).fill_(poison).view(torch.float32) for g in app.mDeferred.mGBuffer]
        geo.DrawNormalsDepthAndGBuffer(app.mMainPassCB, n3, g3, d3, g_rows=(r0, rn))
        g4 = [torch.full_like(g, float("nan")).view(torch.int32).fill_(poison).view(torch.float32) for g in app.mDeferred.mGBuffer]
        d4 = torch.full_like(d2, 0x12345)
        geo.DrawGBuffer(app.mMainPassCB, g4, d4, g_rows=(r0, rn))
        torch.cuda.synchronize()
        assert torch.equal(n3.view(torch.int16), app.mSsao.mNormalMap.view(torch.int16)) and torch.equal(d3, app.mDepthStencilBuffer)
        assert torch.equal(d4[r0:r0 + rn], app.mDepthStencilBuffer[r0:r0 + rn])
        outside = torch.ones(H, dtype=torch.bool, device=dev); outside[r0:r0 + rn] = False
        assert bool((d4[outside] == 0x12345).all())
        for a, b, full in zip(g3, g4, app.mDeferred.mGBuffer):
            for t in (a, b):
                assert torch.equal(t[r0:r0 + rn].view(torch.int32), full[r0:r0 + rn].view(torch.int32))
                assert bool((t[outside].view(torch.int32) == poison).all())
    app.blurCount, app.numDirLights, app.pcfSearchRadius, app.flags = blur_count, ndl, radius, (LIGHT_SKY if sky else 0)
    app.Draw()
    torch.cuda.synchronize()
    return app


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,SD,mips", [(256, 256, 512, False), (322, 190, 256, False), (322, 190, 256, True)])
def test_box_scene_all_on_device(ctx, built_lib, oracle, W, H, SD, mips):
    from crychic_renderer_amd import geometry as g, scene
    consts = raster_util.frame_constants(W, H, SD)
    items, sitems = g.cascade_scene_items(), g.cascade_scene_items(shadow_layer=True)
    mats, tex = g.reference_materials(), g.procedural_textures(64)
    if mips:        # material textures with mip chains: the anisotropic sampler of the G-buffer pass (gsamAnisotropicWrap)
        tex = [g.box_mips(t) for t in tex]
    cube = scene.make_cubemap(32, torch.device("cpu")).numpy()
    radius = built_lib.lib.crychic_pcf_search_radius(SD, 0)
    ref = raster_util.oracle_frame(oracle, consts, items, sitems, mats, tex, W, H, SD, cube, 3, 3, radius)
    app = device_frame(ctx, built_lib, consts, items, sitems, mats, tex, W, H, SD, cube, 3, 3, radius)
    u32 = lambda t: t.cpu().numpy().view(np.uint32)
    for k in range(4):
        assert np.array_equal(u32(app.mShadowMap.mShadowMap[k]), ref["shadow"][k]), "shadow cascade %d" % k
    assert np.array_equal(u32(app.mDepthStencilBuffer), ref["depth"])
    assert np.array_equal(app.mSsao.mNormalMap.cpu().numpy().view(np.uint16), ref["normal"].view(np.uint16))
    for i, k in enumerate(("g0", "g1", "g2")):
        assert np.array_equal(u32(app.mDeferred.mGBuffer[i]), ref[k].view(np.uint32)), k
    assert np.array_equal(app.mSsao.mAmbientMap0.cpu().numpy().view(np.uint16), ref["ao"])
    assert np.array_equal(app.mBackBuffer.cpu().numpy(), ref["rgba8"])
    assert (ref["shadow"] < 0xFFFFFF).mean() > 0.1 and len(np.unique(ref["rgba8"].reshape(-1, 4), axis=0)) > 200


@pytest.mark.gpu
def test_mesh_fixture_frame_on_device(ctx, built_lib, oracle):
    """A loaded triangle mesh (car.txt, committed as a fixture: the GPU box has no reference checkout) through the
    whole device pipeline, SSAO off and on, against the oracle."""
    from crychic_renderer_amd import geometry as g, scene
    d = np.load(os.path.join(GOLD, "car_mesh.npz"))
    v = d["vertices"].view(g.VERTEX_DT).reshape(-1); idx = d["indices"]
    W, H, SD = 256, 192, 256
    consts = raster_util.frame_constants(W, H, SD)
    inst = g.make_instances([g.world_matrix((1.5, 1.5, 1.5), (0.0, 1.0, -6.0))], [3])
    ground = g.create_grid(20.0, 30.0, 60, 40)
    items = [(v, idx, inst), (ground[0], ground[1], g.make_instances([g.world_matrix((3, 3, 3))], [1]))]
    cube = scene.make_cubemap(32, torch.device("cpu")).numpy()
    mats = g.reference_materials()
    for blur_count in (-1, 2):
        ref = raster_util.oracle_frame(oracle, consts, items, items, mats, None, W, H, SD, cube, blur_count, 1, 0.0)
        app = device_frame(ctx, built_lib, consts, items, items, mats, None, W, H, SD, cube, blur_count, 1, 0.0)
        assert np.array_equal(app.mDepthStencilBuffer.cpu().numpy().view(np.uint32), ref["depth"])
        assert np.array_equal(app.mBackBuffer.cpu().numpy(), ref["rgba8"]), blur_count
    assert (ref["depth"] < 0xFFFFFF).mean() > 0.3


@pytest.mark.gpu
def test_raster_argument_errors(ctx, built_lib):
    lib = built_lib.lib
    cb = built_lib.PassConstants()
    items = (built_lib.DrawItem * 1)()
    items[0].indexCount = 4
    d = torch.zeros((8, 8), dtype=torch.int32, device=ctx.device)
    ws = torch.zeros((1 << 16,), dtype=torch.uint8, device=ctx.device)
    rc = lib.crychic_draw_scene_to_shadow_map(ctx.handle, C.byref(cb), items, 1, C.c_void_p(d.data_ptr()), 8, 0, 0.0, C.c_void_p(ws.data_ptr()), ws.numel(), None)
    assert rc == -1 and b"triangle list" in lib.crychic_last_error()
    items[0].indexCount = 3; items[0].instanceCount = 1
    rc = lib.crychic_draw_scene_to_shadow_map(ctx.handle, C.byref(cb), items, 1, C.c_void_p(d.data_ptr()), 8, 0, 0.0, C.c_void_p(ws.data_ptr()), ws.numel(), None)
    assert rc == -1 and b"null buffer" in lib.crychic_last_error()
    rc = lib.crychic_draw_scene_to_shadow_map(ctx.handle, C.byref(cb), None, 0, C.c_void_p(d.data_ptr()), 8, 0, 0.0, C.c_void_p(ws.data_ptr()), 16, None)
    assert rc == -1 and b"workspace" in lib.crychic_last_error()


@pytest.mark.gpu
def test_huge_triangles_render_and_bad_input_is_reported(ctx, built_lib, oracle):
    """A triangle with a vertex a million NDC units outside the viewport is RENDERED (guard-band clipping, as D3D12 does), exactly
    as the oracle renders it, and nothing is flagged; what cannot be drawn -- a vertex index outside the vertex buffer, a
    non-finite position -- is REPORTED by crychic_raster_status instead of silently producing a wrong image."""
    import oracle_lib
    from crychic_renderer_amd import SceneGeometry, geometry as g
    lib, check = built_lib.lib, built_lib.check
    W = H = 64
    flags = C.c_uint32(123)
    cb = built_lib.PassConstants()
    eye = np.eye(4, dtype=np.float32)
    cb.ViewProj[:] = list(eye.reshape(-1)); cb.View[:] = list(eye.reshape(-1))
    depth = torch.zeros((H, W), dtype=torch.int32, device=ctx.device)

    def items(pts, idx=(0, 1, 2)):
        v = np.zeros(len(pts), dtype=g.VERTEX_DT)
        v["Pos"] = np.asarray(pts, dtype=np.float32); v["Normal"] = (0, 0, -1); v["TangentU"] = (1, 0, 0)
        both = list(idx) + [idx[0], idx[2], idx[1]]          # both windings: one of them faces the camera
        return [(v, np.asarray(both, dtype=np.uint32), g.make_instances([np.eye(4, dtype=np.float32).reshape(-1)], [0]))]

    def draw(it):
        SceneGeometry(ctx, it).DrawSceneToShadowMap(cb, depth, depth_bias=0, slope_bias=0.0)
        check(lib.crychic_raster_status(ctx.handle, None, C.byref(flags)))
        return flags.value, depth.cpu().numpy().view(np.uint32)

    # an ordinary clockwise triangle inside the clip volume: nothing to report, pixels covered
    st, d = draw(items([(-0.5, -0.5, 0.5), (-0.5, 0.5, 0.5), (0.5, -0.5, 0.5)]))
    assert st == 0 and (d != 0xFFFFFF).sum() > 100
    # vertices 1e6 NDC units away, z inside the clip range: clipped to the guard band and drawn, like the oracle draws them
    for pts in ([(-0.5, -0.5, 0.5), (-0.5, 0.5, 0.5), (1.0e6, -0.5, 0.5)], [(-1.0e6, -0.9, 0.25), (0.0, 2.0e6, 0.5), (1.0e6, -0.9, 0.75)],
                [(3.0e5, -0.5, 0.5), (3.0e5, 0.5, 0.5), (4.0e5, 0.0, 0.5)]):
        st, d = draw(items(pts))
        ref = oracle_lib.rasterize(oracle, 0, np.array(cb.View, np.float32), np.array(cb.ViewProj, np.float32), items(pts), None, None, W, H, 0, 0.0)
        assert st == 0 and np.array_equal(d, ref["depth"])
    assert (d == 0xFFFFFF).all()                                # the last one lies wholly outside: nothing drawn, nothing flagged
    st, d = draw(items([(-0.5, -0.5, 0.5), (-0.5, 0.5, 0.5), (1.0e6, -0.5, 0.5)]))
    assert (d[H // 2 + 4, W // 2:] != 0xFFFFFF).all()          # the long triangle runs off the right edge of the target
    # a non-finite position cannot be drawn: flagged
    st, d = draw(items([(-0.5, -0.5, 0.5), (-0.5, 0.5, 0.5), (np.nan, -0.5, 0.5)]))
    assert st & 1 and (d == 0xFFFFFF).all()
    # an index outside the 3-vertex buffer
    st, d = draw(items([(-0.5, -0.5, 0.5), (-0.5, 0.5, 0.5), (0.5, -0.5, 0.5)], idx=(0, 1, 7)))
    assert st & 2 and (d == 0xFFFFFF).all()


# ---- BASELINE configs[0]: CPU-only plumbing case ----------------------------------------------------------------------
def test_c1_skull_oracle_matches_golden(built_lib, oracle):
    """256x256, skull.txt + 1 directional light, SSAO off: rendered by the CPU oracle end to end (rasteriser included)
    and compared with the committed golden image.  Needs the reference's model file, present only in the build
    container."""
    path = "/root/reference/Models/skull.txt"
    if not os.path.exists(path):
        pytest.skip("reference models are not on this machine")
    gold = np.load(os.path.join(GOLD, "c1_skull_256.npz"))
    out = raster_util.render_c1(oracle, path)
    assert np.array_equal(out["depth"], gold["depth"])
    assert np.array_equal(out["rgba8"], gold["rgba8"])
    assert abs(out["covered"] - float(gold["covered"][0])) < 1e-9 and 0.02 < out["covered"] < 0.06
    assert (out["rgba8"][out["depth"] == 0xFFFFFF] == np.array([176, 196, 222, 255], np.uint8)).all()


def test_c1_skull_kernel_bodies(built_lib, oracle, hostsim):
    """The same case through the kernels' own bodies on the host (raster_core.hpp + light_core.hpp)."""
    path = "/root/reference/Models/skull.txt"
    if not os.path.exists(path):
        pytest.skip("reference models are not on this machine")
    from crychic_renderer_amd import geometry as g, scene
    W = H = 256
    consts = raster_util.frame_constants(W, H, 512)
    v, idx = g.load_mesh_text(path)
    items = raster_util.c1_items(v, idx)
    mats = g.reference_materials()
    view = np.array(consts.pass_cb.View, np.float32); vp = np.array(consts.pass_cb.ViewProj, np.float32)
    shadow = np.stack([hostsim.rasterize(0, view, raster_util.light_viewproj_t(consts, k), items, None, None, 512, 512, 10000, 2.0)["depth"]
                       for k in range(4)])
    nd = hostsim.rasterize(1, view, vp, items, mats, None, W, H)
    gb = hostsim.rasterize(2, view, vp, items, mats, None, W, H)
    cube = scene.make_cubemap(32, torch.device("cpu")).numpy()
    img = hostsim.light(consts.pass_cb, gb["g0"], gb["g1"], gb["g2"], nd["depth"], None, shadow, cube, 1, 0.0)
    gold = np.load(os.path.join(GOLD, "c1_skull_256.npz"))
    assert np.array_equal(nd["depth"], gold["depth"]) and np.array_equal(img, gold["rgba8"])
