"""BASELINE configs[4] extension: point lights with tiled light culling (build-defined: the reference's point-light branch,
PBR.hlsl:109-124, is dead code -- parity is against this repo's oracle only, "parity unpinned")."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
import scene_util


def as_or_lights(lights):
    arr = (oracle_lib.OrLight * len(lights))()
    C.memmove(C.addressof(arr), C.addressof(lights), C.sizeof(arr))
    return arr


def lights_for_test():
    from crychic_renderer_amd import scene
    L = scene.point_light_grid(8)
    # make the set less regular: a few coloured / short-range / far-away lights
    L[3].Strength[:] = (3.0, 0.2, 0.2); L[3].FalloffEnd = 4.0
    L[10].Strength[:] = (0.1, 2.5, 0.3); L[10].FalloffStart = 0.5
    L[20].Position[:] = (0.0, 1.0, -9.0); L[20].Strength[:] = (1.5, 1.5, 0.2)
    L[63].Position[:] = (500.0, 500.0, 500.0)            # reaches nothing
    L[40].FalloffEnd = 200.0                              # reaches everything
    return L


@pytest.mark.parametrize("W,H", [(130, 70), (256, 144)])
def test_point_lights_kernel_body_matches_oracle(built_lib, oracle, hostsim, W, H):
    pl = scene_util.cpu_scene(W, H, 256, 32)
    p = scene_util.np_planes(pl); c = pl["consts"]
    pcb = oracle_lib.as_oracle_cb(c.pass_cb, oracle_lib.OrPassConstants)
    L = lights_for_test()
    ref, rref = oracle.deferred_light(pcb, p["g0"], p["g1"], p["g2"], p["depth"], None, p["shadow"], p["cube"], 1, 0.0,
                                      want_radiance=True, point_lights=as_or_lights(L))
    got, rgot = hostsim.light(c.pass_cb, p["g0"], p["g1"], p["g2"], p["depth"], None, p["shadow"], p["cube"], 1, 0.0,
                              want_radiance=True, point_lights=L)
    assert np.array_equal(got, ref) and np.array_equal(rgot.view(np.uint32), rref.view(np.uint32))
    base = oracle.deferred_light(pcb, p["g0"], p["g1"], p["g2"], p["depth"], None, p["shadow"], p["cube"], 1, 0.0)
    assert (ref.astype(np.int32) - base.astype(np.int32)).max() > 20      # the lights really add light
    assert (ref.astype(np.int32) >= base.astype(np.int32)).all()
    # a light out of range of everything changes nothing, bit for bit
    far = (oracle_lib.OrLight * 1)(); C.memmove(C.addressof(far), C.addressof(L[63]), 48)
    assert np.array_equal(oracle.deferred_light(pcb, p["g0"], p["g1"], p["g2"], p["depth"], None, p["shadow"], p["cube"], 1, 0.0,
                                                point_lights=far), base)


@pytest.mark.gpu
@pytest.mark.parametrize("W,H", [(256, 144), (322, 190)])
def test_tiled_point_lights_on_device(built_lib, oracle, W, H):
    import torch
    from crychic_renderer_amd import Context, Crychic
    ctx = Context(0)
    pl = scene_util.cpu_scene(W, H, 256, 32)
    p = scene_util.np_planes(pl); c = pl["consts"]
    dev = {k: torch.from_numpy(np.ascontiguousarray(v).view(np.int32) if v.dtype == np.uint32 else np.ascontiguousarray(v)).to(ctx.device)
           for k, v in p.items()}
    app = Crychic(ctx, W, H, dev["randvec"], dev["cube"], shadow_dim=256)
    app.load_scene({**dev, "consts": c})
    L = lights_for_test()
    scb = oracle_lib.as_oracle_cb(c.ssao_cb, oracle_lib.OrSsaoConstants)
    pcb = oracle_lib.as_oracle_cb(c.pass_cb, oracle_lib.OrPassConstants)
    ao = oracle.compute_ssao(scb, p["normal"], p["depth"], p["randvec"], 2)
    for radius_literal in (1, 0):
        app.blurCount, app.numDirLights = 2, 3
        app.pcfSearchRadius = built_lib.lib.crychic_pcf_search_radius(256, radius_literal)
        app.set_point_lights(L)
        app.Draw()
        torch.cuda.synchronize()
        ref = oracle.deferred_light(pcb, p["g0"], p["g1"], p["g2"], p["depth"], ao, p["shadow"], p["cube"], 3, app.pcfSearchRadius,
                                    point_lights=as_or_lights(L))
        assert np.array_equal(app.mBackBuffer.cpu().numpy(), ref), radius_literal
    # no lights -> the reference configuration again
    app.set_point_lights(None)
    app.Draw()
    torch.cuda.synchronize()
    ref0 = oracle.deferred_light(pcb, p["g0"], p["g1"], p["g2"], p["depth"], ao, p["shadow"], p["cube"], 3, app.pcfSearchRadius)
    assert np.array_equal(app.mBackBuffer.cpu().numpy(), ref0)
    ctx.close()


@pytest.mark.gpu
def test_c5_8k_point_lights_properties(built_lib, oracle):
    """BASELINE configs[4] size: 7680x4320, 64 point lights (8x8 grid, y = 3, falloff 1 -> 10) + PBR cubemap.
    Full-size properties (determinism, 8-strip decomposition == whole frame, lights only add light) and the oracle on a
    band of rows; the 8-GPU all-gather itself is covered by tests/test_sharding_gloo.py."""
    import torch
    from crychic_renderer_amd import Context, Crychic, scene
    W, H, SD = 7680, 4320, 4096          # the cascade size BASELINE and bench.py use
    ctx = Context(0)
    pl = scene.make_scene(W, H, shadow_dim=SD, cube_dim=256, device=str(ctx.device))
    c = pl["consts"]
    app = Crychic(ctx, W, H, pl["randvec"], pl["cube"], shadow_dim=SD)
    app.load_scene(pl)
    app.blurCount, app.numDirLights = 4, 3
    L = scene.point_light_grid(8)
    app.Draw()
    torch.cuda.synchronize()
    base = app.mBackBuffer.cpu().numpy().copy()
    app.set_point_lights(L)
    app.Draw()
    torch.cuda.synchronize()
    full = app.mBackBuffer.cpu().numpy().copy()
    ao = app.mSsao.mAmbientMap0.cpu().numpy().view(np.uint16).copy()
    assert (full.astype(np.int16) >= base.astype(np.int16)).all() and (full != base).any()
    app.Draw()
    torch.cuda.synchronize()
    assert np.array_equal(app.mBackBuffer.cpu().numpy(), full)
    app.mBackBuffer.zero_()
    for rank in range(8):
        r0, rn = C.c_uint32(), C.c_uint32()
        built_lib.check(built_lib.lib.crychic_strip_rows(H, 8, rank, C.byref(r0), C.byref(rn)))
        app.mSsao.mAmbientMap0.fill_(0x5A5A)
        app.Draw(r0.value, rn.value)
    torch.cuda.synchronize()
    assert np.array_equal(app.mBackBuffer.cpu().numpy(), full)
    del base
    # oracle on a band through the boxes (the AO plane is the device's own: its parity is test_c3 / test_compute_ssao)
    p = scene_util.np_planes(pl)
    pcb = oracle_lib.as_oracle_cb(c.pass_cb, oracle_lib.OrPassConstants)
    band0, rows = H // 2 + 200, 32
    ref = oracle.deferred_light(pcb, p["g0"], p["g1"], p["g2"], p["depth"], ao, p["shadow"], p["cube"], 3, app.pcfSearchRadius,
                                point_lights=as_or_lights(L), row0=band0, rows=rows)
    assert np.array_equal(full[band0:band0 + rows], ref[band0:band0 + rows])
    scb = oracle_lib.as_oracle_cb(c.ssao_cb, oracle_lib.OrSsaoConstants)
    ref_ssao = oracle.ssao(scb, p["normal"], p["depth"], p["randvec"], band0 // 2, rows // 2)
    a0 = torch.zeros((H // 2, W // 2), dtype=torch.int16, device=ctx.device)
    built_lib.check(built_lib.lib.crychic_ssao(ctx.handle, C.byref(c.ssao_cb), C.c_void_p(pl["normal"].data_ptr()),
                                               C.c_void_p(pl["depth"].data_ptr()), C.c_void_p(pl["randvec"].data_ptr()),
                                               C.c_void_p(a0.data_ptr()), None, W, H, band0 // 2, rows // 2,
                                               C.c_void_p(torch.cuda.current_stream(ctx.device).cuda_stream)))
    torch.cuda.synchronize()
    sl = slice(band0 // 2, band0 // 2 + rows // 2)
    assert np.array_equal(a0.cpu().numpy().view(np.uint16)[sl], ref_ssao[sl])
    ctx.close()
