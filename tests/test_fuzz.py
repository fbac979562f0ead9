"""Fuzz parity: random planes + perturbed constants (tests/fuzz_util.py) through the oracle, the kernel bodies compiled
for the host, and (gpu) the device.  Bar: AO and RGBA8 bit-exact; radiance bit-exact up to NaN payloads."""
import ctypes as C
import os

import numpy as np
import pytest

import fuzz_util
import oracle_lib

EXTRA = int(os.environ.get("CRY_FUZZ_EXTRA", "0"))      # a soak run: CRY_FUZZ_EXTRA=400 python -m pytest tests/test_fuzz.py


def with_chain(planes, seed):
    """Every other case binds the cube map's mip chain (trilinear lookups, level of detail from the pixel quads): the incoherent
    planes put NaN / inf / zero-length normals next to each other, i.e. into the quad derivatives."""
    if seed % 2 == 0:
        return planes["cube"], {}
    from crychic_renderer_amd import geometry as g
    chain, levels = g.cube_mip_chain(planes["cube"])
    return chain, dict(cube_dim=int(planes["cube"].shape[1]), cube_levels=levels)


def oracle_frame(oracle, planes, c, knobs, cube=None, chain_kw=None):
    scb = oracle_lib.as_oracle_cb(c.ssao_cb, oracle_lib.OrSsaoConstants)
    pcb = oracle_lib.as_oracle_cb(c.pass_cb, oracle_lib.OrPassConstants)
    ao = oracle.compute_ssao(scb, planes["normal"], planes["depth"], planes["randvec"], knobs["blurCount"]) if knobs["ssao_on"] else None
    out, rad = oracle.deferred_light(pcb, planes["g0"], planes["g1"], planes["g2"], planes["depth"], ao, planes["shadow"],
                                     planes["cube"] if cube is None else cube, knobs["numDirLights"], knobs["pcfSearchRadius"], sky=bool(knobs["sky"]),
                                     want_radiance=True, **(chain_kw or {}))
    return ao, out, rad


@pytest.mark.parametrize("seed", range(6 + EXTRA // 8))
def test_fuzz_kernel_bodies(built_lib, oracle, hostsim, seed):
    W, H, planes, c, knobs = fuzz_util.random_case(seed, built_lib)
    cube, chain_kw = with_chain(planes, seed)
    ao, out, rad = oracle_frame(oracle, planes, c, knobs, cube, chain_kw)
    if knobs["ssao_on"]:
        got, edge = hostsim.ssao(c.ssao_cb, planes["normal"], planes["depth"], planes["randvec"],
                                 int(built_lib.lib.crychic_edge_plane_bytes(W, H)))
        for _ in range(knobs["blurCount"]):
            got = hostsim.blur(c.ssao_cb, edge, got, W, H, True)
            got = hostsim.blur(c.ssao_cb, edge, got, W, H, False)
        assert np.array_equal(got, ao), knobs
    o2, r2 = hostsim.light(c.pass_cb, planes["g0"], planes["g1"], planes["g2"], planes["depth"], ao, planes["shadow"], cube,
                           knobs["numDirLights"], knobs["pcfSearchRadius"], flags=knobs["sky"], want_radiance=True, **chain_kw)
    assert np.array_equal(o2, out), knobs
    assert fuzz_util.same_floats(r2, rad), knobs


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(24 + EXTRA))
def test_fuzz_device(built_lib, oracle, seed):
    import torch
    from crychic_renderer_amd import Context, Crychic
    W, H, planes, c, knobs = fuzz_util.random_case(1000 + seed, built_lib)
    cube, chain_kw = with_chain(planes, seed)
    ao, out, rad = oracle_frame(oracle, planes, c, knobs, cube, chain_kw)
    ctx = Context(0)
    try:
        dev = {k: torch.from_numpy(np.ascontiguousarray(v).view(np.int32) if v.dtype == np.uint32 else np.ascontiguousarray(v)).to(ctx.device)
               for k, v in planes.items()}
        app = Crychic(ctx, W, H, dev["randvec"], dev["cube"], shadow_dim=planes["shadow"].shape[1])
        app.load_scene({**dev, "consts": c})
        lib, check = built_lib.lib, built_lib.check
        st = C.c_void_p(torch.cuda.current_stream(ctx.device).cuda_stream)
        a0, a1, edge = app.mSsao.mAmbientMap0, app.mSsao.mAmbientMap1, app.mSsao.mEdge
        if knobs["ssao_on"]:
            check(lib.crychic_ssao_compute(ctx.handle, C.byref(c.ssao_cb), C.c_void_p(dev["normal"].data_ptr()),
                                           C.c_void_p(dev["depth"].data_ptr()), C.c_void_p(dev["randvec"].data_ptr()),
                                           C.c_void_p(a0.data_ptr()), C.c_void_p(a1.data_ptr()), C.c_void_p(edge.data_ptr()),
                                           W, H, knobs["blurCount"], 0, H // 2, st))
            torch.cuda.synchronize()
            assert np.array_equal(a0.cpu().numpy().view(np.uint16), ao), knobs
        o = torch.zeros((H, W, 4), dtype=torch.uint8, device=ctx.device)
        r = torch.zeros((H, W, 4), dtype=torch.float32, device=ctx.device)
        sh = (C.c_void_p * 4)(*[dev["shadow"][k].data_ptr() for k in range(4)])
        dcube = torch.from_numpy(np.ascontiguousarray(cube)).to(ctx.device)
        check(lib.crychic_deferred_light(ctx.handle, C.byref(c.pass_cb), C.c_void_p(dev["g0"].data_ptr()), C.c_void_p(dev["g1"].data_ptr()),
                                         C.c_void_p(dev["g2"].data_ptr()), C.c_void_p(dev["depth"].data_ptr()),
                                         C.c_void_p(a0.data_ptr()) if knobs["ssao_on"] else None, sh, planes["shadow"].shape[1],
                                         C.c_void_p(dcube.data_ptr()), planes["cube"].shape[1], C.c_void_p(o.data_ptr()),
                                         C.c_void_p(r.data_ptr()), W, H, 0, H, knobs["numDirLights"], knobs["pcfSearchRadius"],
                                         knobs["sky"] | (chain_kw.get("cube_levels", 0) << 16), st))
        torch.cuda.synchronize()
        got = o.cpu().numpy()
        assert np.array_equal(got, out), "%s: %d channels differ" % (knobs, (got != out).sum())
        assert fuzz_util.same_floats(r.cpu().numpy(), rad), knobs
    finally:
        ctx.close()


# The smallest legal frames, single-row / single-column half-res maps, and the longest rows and columns a moderate pixel count
# allows (tall frames: many workgroup rows; wide frames: long rows of tiles, one cell row of every coarse map).
TINY = [(2, 2), (2, 4), (4, 2), (6, 6), (2, 70), (70, 2), (130, 2), (2, 130)]
LONG = [(16, 131072), (131072, 16), (2, 262140), (1048574, 2)]


@pytest.mark.parametrize("size", TINY)
def test_tiny_frames_kernel_bodies(built_lib, oracle, hostsim, size):
    W, H, planes, c, knobs = fuzz_util.random_case(77, built_lib, size=size)
    knobs["ssao_on"], knobs["blurCount"] = True, 2
    ao, out, rad = oracle_frame(oracle, planes, c, knobs)
    eb = int(built_lib.lib.crychic_edge_plane_bytes(W, H))
    got, _ = hostsim.compute_ssao(c.ssao_cb, planes["normal"], planes["depth"], planes["randvec"], eb, 2)
    assert np.array_equal(got, ao), size
    o2, r2 = hostsim.light(c.pass_cb, planes["g0"], planes["g1"], planes["g2"], planes["depth"], ao, planes["shadow"], planes["cube"],
                           knobs["numDirLights"], knobs["pcfSearchRadius"], flags=knobs["sky"], want_radiance=True)
    assert np.array_equal(o2, out), size


@pytest.mark.gpu
@pytest.mark.parametrize("size", TINY + LONG)
def test_tiny_and_long_frames_on_device(built_lib, oracle, size):
    """crychic_draw_hot_path (SSAO + 2 blur iterations + lighting) on frames of extreme shape == oracle; one size past the limit is
    refused with CRYCHIC_E_UNSUPPORTED rather than launched."""
    import torch
    from crychic_renderer_amd import Context, Crychic
    W, H, planes, c, knobs = fuzz_util.random_case(78, built_lib, size=size)
    knobs["ssao_on"], knobs["blurCount"] = True, 2
    ao, out, rad = oracle_frame(oracle, planes, c, knobs)
    ctx = Context(0)
    try:
        dev = {k: torch.from_numpy(np.ascontiguousarray(v).view(np.int32) if v.dtype == np.uint32 else np.ascontiguousarray(v)).to(ctx.device)
               for k, v in planes.items()}
        app = Crychic(ctx, W, H, dev["randvec"], dev["cube"], shadow_dim=planes["shadow"].shape[1])
        app.load_scene({**dev, "consts": c})
        app.blurCount, app.numDirLights, app.pcfSearchRadius, app.flags = 2, knobs["numDirLights"], knobs["pcfSearchRadius"], knobs["sky"]
        app.Draw()
        torch.cuda.synchronize()
        assert np.array_equal(app.mSsao.mAmbientMap0.cpu().numpy().view(np.uint16), ao), size
        assert np.array_equal(app.mBackBuffer.cpu().numpy(), out), size
    finally:
        ctx.close()


@pytest.mark.gpu
def test_oversized_frames_are_refused(built_lib):
    from crychic_renderer_amd import Context
    lib = built_lib.lib
    ctx = Context(0)
    try:
        _, _, _, c, _ = fuzz_util.random_case(79, built_lib, size=(4, 4))
        for W, H in ((2, 262142), (1 << 20, 2), (32768, 16384)):
            rc = lib.crychic_ssao(ctx.handle, C.byref(c.ssao_cb), C.c_void_p(8), C.c_void_p(8), C.c_void_p(8), C.c_void_p(8), None, W, H, 0, H // 2, None)
            assert rc == -4, (W, H, rc)
    finally:
        ctx.close()
