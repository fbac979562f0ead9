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
