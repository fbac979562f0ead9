"""Host constant builders of the product (csrc/host_constants.cpp: the f2 row of SURVEY.md 8) against the oracle's
independent restatement of CRYCHIC.cpp:634-937 and Ssao.cpp:37-68,392-462."""
import ctypes as C

import numpy as np

import oracle_lib


def cam_pair(built_lib, W, H):
    from crychic_renderer_amd import scene
    cam = scene.default_camera(W, H)
    ocam = oracle_lib.OrCamera()
    C.memmove(C.addressof(ocam), C.addressof(cam), C.sizeof(cam))
    return cam, ocam


def struct_floats(s):
    return np.frombuffer(bytes(s), dtype=np.float32)


def test_rand_offsets_randvec_identical(built_lib, oracle):
    s1, s2 = C.c_uint32(1), C.c_uint32(1)
    o1 = ((C.c_float * 4) * 14)(); o2 = np.zeros((14, 4), np.float32)
    built_lib.lib.crychic_build_offset_vectors(C.byref(s1), o1)
    oracle.lib.or_build_offset_vectors(C.byref(s2), o2.ctypes.data)
    assert np.allclose(np.array(o1, dtype=np.float32), o2, atol=1e-7) and s1.value == s2.value
    for rtl in (0, 1):
        t1 = np.zeros((256, 256, 4), np.uint8); t2 = np.zeros_like(t1)
        a, b = C.c_uint32(s1.value), C.c_uint32(s1.value)
        built_lib.lib.crychic_build_random_vector_texture(C.byref(a), rtl, t1.ctypes.data)
        oracle.lib.or_build_random_vector_texture(C.byref(b), rtl, t2.ctypes.data)
        assert np.array_equal(t1, t2) and a.value == b.value
        assert (t1[..., 3] == 0).all() and 100 < t1[..., :3].mean() < 155
    # first texel: rand 9961, 491, 2995 -> v.z = 9961/32767 when arguments are evaluated right to left
    assert t1[0, 0, 0] == round(9961 / 32767 * 255)


def test_gauss_weights_identical(built_lib, oracle):
    w1 = (C.c_float * 11)(); w2 = np.zeros(11, np.float32)
    assert built_lib.lib.crychic_calc_gauss_weights(2.5, w1, 11) == 11
    oracle.lib.or_calc_gauss_weights(2.5, w2.ctypes.data, 11)
    assert np.array_equal(np.array(w1, dtype=np.float32), w2)
    assert built_lib.lib.crychic_calc_gauss_weights(3.0, w1, 11) < 0     # radius 6 > MaxBlurRadius (Ssao.cpp:45)
    assert built_lib.lib.crychic_calc_gauss_weights(2.5, w1, 5) < 0


def test_constant_buffers_match_oracle(built_lib, oracle):
    from crychic_renderer_amd import scene
    for (W, H, sd) in ((1920, 1080, 4096), (256, 256, 512), (3840, 2160, 4096)):
        cam, ocam = cam_pair(built_lib, W, H)
        consts = scene.Constants(W, H, sd, cam)
        ld = np.asarray(scene.BASE_LIGHT_DIRS, np.float32)
        lv = np.zeros((4, 16), np.float32); lp = np.zeros((4, 16), np.float32); st = np.zeros((4, 16), np.float32)
        oracle.lib.or_cascade_shadow_transforms(C.addressof(ocam), ld[0].ctypes.data, sd, lv.ctypes.data, lp.ctypes.data, st.ctypes.data)
        assert np.allclose(consts.light_view.reshape(4, 16), lv, rtol=2e-5, atol=2e-4)
        assert np.allclose(consts.light_proj.reshape(4, 16), lp, rtol=2e-5, atol=2e-4)
        assert np.allclose(consts.shadow_transform.reshape(4, 16), st, rtol=2e-5, atol=2e-4)
        opass = oracle_lib.OrPassConstants()
        oracle.lib.or_build_pass_constants(C.addressof(ocam), W, H, st.ctypes.data, ld.ctypes.data, C.addressof(opass))
        a, b = struct_floats(consts.pass_cb), struct_floats(opass)
        assert np.allclose(a, b, rtol=2e-5, atol=2e-4)
        off = np.array(consts.offsets, dtype=np.float32)
        ossao = oracle_lib.OrSsaoConstants()
        oracle.lib.or_build_ssao_constants(C.addressof(ocam), W, H, off.ctypes.data, C.addressof(ossao))
        assert np.allclose(struct_floats(consts.ssao_cb), struct_floats(ossao), rtol=1e-6, atol=1e-6)


def test_reference_values(built_lib):
    """Values the reference hard-codes."""
    from crychic_renderer_amd import scene
    c = scene.Constants(800, 600, 4096)
    s, p = c.ssao_cb, c.pass_cb
    assert (s.OcclusionRadius, s.OcclusionFadeStart, s.OcclusionFadeEnd, s.SurfaceEpsilon) == (0.5, np.float32(0.2), 1.0, np.float32(0.05))
    assert s.InvRenderTargetSize[0] == np.float32(1 / 400) and s.InvRenderTargetSize[1] == np.float32(1 / 300)
    assert tuple(s.RenderTargetSize) == (0.0, 0.0)                       # never written (FrameResource.h:61)
    assert (p.NearZ, p.FarZ) == (1.0, 1000.0)                            # sic, CRYCHIC.cpp:854-855
    assert np.allclose(list(p.AmbientLight), [0.4, 0.4, 0.6, 1.0])
    assert np.allclose(list(p.Lights[0].Strength), [2.4, 2.4, 2.5]) and np.allclose(list(p.Lights[2].Strength), [0, 0, 0])
    assert p.Lights[5].SpotPower == 64.0 and p.Lights[5].FalloffEnd == 10.0
    # transposed storage: HLSL gProj[3][2] = mem[11] = B, gProj[2][2] = mem[10] = A  (Ssao.hlsl:110-115)
    assert abs(s.Proj[10] - 100.0 / 99.0) < 1e-6 and abs(s.Proj[11] + 100.0 / 99.0) < 1e-6 and s.Proj[14] == 1.0
    # shadow transforms map the cascade centre into [0,1]^3
    for k in range(4):
        M = np.array(c.shadow_transform[k], dtype=np.float64)
        assert abs(M[0, 0]) > 0 and M[3, 3] == 1.0
    assert built_lib.lib.crychic_pcf_search_radius(4096, 1) == 0.0
    assert abs(built_lib.lib.crychic_pcf_search_radius(4096, 0) - 2.5 / 4096) < 1e-9
