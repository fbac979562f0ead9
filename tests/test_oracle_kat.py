"""Known-answer tests pinning the oracle.  The reference holds no golden vectors of its
own ("parity unpinned"); these are the analytically derived answers for the formulas the reference states."""
import ctypes as C
import math

import numpy as np

import oracle_lib


def test_gauss_weights_sigma_2_5(oracle):
    # Ssao.cpp:37-68, sigma = 2.5 (CRYCHIC.cpp:922)
    w = np.zeros(11, dtype=np.float32)
    n = oracle.lib.or_calc_gauss_weights(2.5, w.ctypes.data, 11)
    assert n == 11
    kat = [0.02219055, 0.04558900, 0.07981141, 0.11906464, 0.15136082, 0.16396722,
           0.15136082, 0.11906464, 0.07981141, 0.04558900, 0.02219055]
    assert np.allclose(w, kat, atol=2e-8)
    assert abs(float(w.sum()) - 1.0) < 2e-7
    assert np.array_equal(w, w[::-1])


def test_msvc_rand_and_offset_vectors(oracle):
    # MSVC CRT LCG, seed 1; Ssao.cpp:423-462
    st = C.c_uint32(1)
    off = np.zeros((14, 4), dtype=np.float32)
    oracle.lib.or_build_offset_vectors(C.byref(st), off.ctypes.data)
    lengths = np.linalg.norm(off[:, :3].astype(np.float64), axis=1)
    kat = [0.250938, 0.672689, 0.394978, 0.856555, 0.688757, 0.609905, 0.512719, 0.921972, 0.867130, 0.809954,
           0.380581, 0.894208, 0.782876, 0.635151]
    assert np.allclose(lengths, kat, atol=2e-6)
    # directions: 8 cube corners in opposite pairs, then -x,+x,-y,+y,-z,+z
    dirs = off[:, :3] / lengths[:, None]
    assert np.allclose(dirs[0], np.ones(3) / math.sqrt(3), atol=1e-6)
    assert np.allclose(dirs[1], -np.ones(3) / math.sqrt(3), atol=1e-6)
    assert np.allclose(dirs[8], [-1, 0, 0], atol=1e-6) and np.allclose(dirs[13], [0, 0, 1], atol=1e-6)
    assert (off[:, 3] == 0).all()
    # the next three rand() values feed the first random-vector texel
    nxt = [oracle.lib.or_msvc_rand(C.byref(st)) for _ in range(3)]
    assert nxt == [9961, 491, 2995]


def test_msvc_rand_first_values(oracle):
    st = C.c_uint32(1)
    assert [oracle.lib.or_msvc_rand(C.byref(st)) for _ in range(5)] == [41, 18467, 6334, 26500, 19169]


def test_depth_linearisation(oracle):
    # near 1, far 100: viewZ(0) = 1, viewZ(1) = 100, viewZ(0.5) = B / (0.5 - A)
    cb = oracle_lib.OrSsaoConstants()
    proj = np.zeros(16, dtype=np.float32)
    oracle.lib.or_mat_perspective_fov_lh(0.25 * math.pi, 16 / 9, 1.0, 100.0, proj.ctypes.data)
    cb.Proj[:] = list(proj.reshape(4, 4).T.reshape(-1))
    f = lambda z: oracle.lib.or_ndc_depth_to_view_depth(C.addressof(cb), z)
    assert abs(f(0.0) - 1.0) < 1e-6
    assert abs(f(1.0) - 100.0) < 2e-3
    assert abs(f(0.5) - 1.980198) < 1e-6
    assert abs(f(oracle.lib.or_d24_to_float(0xFFFFFF)) - 100.0) < 2e-3
    assert oracle.lib.or_d24_to_float(0xFFFFFF) == 1.0 and oracle.lib.or_d24_to_float(0) == 0.0
    assert oracle.lib.or_d24_to_float(0xAB123456) == oracle.lib.or_d24_to_float(0x123456)  # stencil byte ignored


def test_pcf_search_radius_quirk(oracle):
    # Common.hlsl:305: `5 / width / 2.0f` with uint width -> 0 for the 4096 map; intended 2.5 texels
    assert oracle.lib.or_pcf_search_radius(4096, 1) == 0.0
    assert oracle.lib.or_pcf_search_radius(4, 1) == 0.5
    assert abs(oracle.lib.or_pcf_search_radius(4096, 0) * 4096 - 2.5) < 1e-6


def test_detmath_accuracy(oracle):
    # the deterministic transcendentals are real sin/cos/log2/exp2/pow to a few ulp
    x = np.linspace(-200.0, 200.0, 200001).astype(np.float32)
    assert np.max(np.abs(oracle.eval_array(0, x) - np.sin(x.astype(np.float64)))) < 3e-6
    assert np.max(np.abs(oracle.eval_array(1, x) - np.cos(x.astype(np.float64)))) < 3e-6
    p = np.exp(np.linspace(-80, 80, 100001)).astype(np.float32)
    l2 = oracle.eval_array(2, p)
    assert np.max(np.abs(l2 - np.log2(p.astype(np.float64))) / np.maximum(1.0, np.abs(np.log2(p.astype(np.float64))))) < 2e-7
    e = np.linspace(-120, 120, 100001).astype(np.float32)
    assert np.max(np.abs(oracle.eval_array(3, e) / np.exp2(e.astype(np.float64)) - 1.0)) < 3e-7
    b = np.linspace(0.0, 1.0, 100001).astype(np.float32)
    g = oracle.eval_array(4, b, np.full_like(b, np.float32(1.0 / 2.2)))
    assert np.max(np.abs(g - b.astype(np.float64) ** (1.0 / 2.2))) < 3e-7
    assert g[0] == 0.0 and abs(g[-1] - 1.0) < 1e-7
    # special values
    assert np.isnan(oracle.lib.or_det_powf(-1.0, 0.5)) and np.isnan(oracle.lib.or_det_sinf(float("inf")))
    assert oracle.lib.or_det_powf(0.0, 0.4545) == 0.0


def test_pow_inv_gamma_definition(oracle):
    """The tone map's pow(x, 1/2.2) (DeferredShading.hlsl:90; oracle definition version 3, or_math.h or_pow_inv_gamma): table over
    the exponent field x degree-7 polynomial over the mantissa.  Real x^(1/2.2) to 3e-7 relative over the whole normal range, the
    stated special values, monotone, and both generated constant files are what tools/gen_gamma_pow.py prints."""
    import subprocess, sys, os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, os.path.join(root, "tools", "gen_gamma_pow.py"), "--check"], check=True)
    rng = np.random.default_rng(3)
    x = np.concatenate([np.linspace(0.0, 1.0, 200001), np.exp(rng.uniform(-87.0, 88.0, 200000)), [1.0, 2.0, 0.5, 1.17549435e-38, 3.4028234e38]]).astype(np.float32)
    g = oracle.eval_array(10, x).astype(np.float64)
    ref = x.astype(np.float64) ** (1.0 / 2.2)
    nz = x >= np.float32(1.17549435e-38)
    assert np.max(np.abs(g[nz] / ref[nz] - 1.0)) < 3e-7
    assert np.all(g[~nz] == 0.0)                                            # zero and subnormals -> 0
    xs = np.sort(x[:200001]); gs = oracle.eval_array(10, xs)
    assert np.all(np.diff(gs) >= 0.0)                                        # monotone on [0, 1]
    special = np.array([0.0, -0.0, 1e-40, -1e-40, np.inf, 1.0], dtype=np.float32)
    gsp = oracle.eval_array(10, special)
    assert np.array_equal(gsp[:4], np.zeros(4, np.float32)) and gsp[4] == np.inf and gsp[5] == 1.0
    bad = np.array([-1.0, -1.17549435e-38, -np.inf, np.nan], dtype=np.float32)
    assert np.all(np.isnan(oracle.eval_array(10, bad)))


def test_half_decode(oracle):
    bits = np.arange(65536, dtype=np.uint32)
    got = oracle.eval_array(9, bits.view(np.float32))
    ref = bits.astype(np.uint16).view(np.float16).astype(np.float32)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def _flat_case(W, H):
    import scene_util
    consts = scene_util.cpu_scene(W, H, 128, 16)["consts"]
    scb = oracle_lib.as_oracle_cb(consts.ssao_cb, oracle_lib.OrSsaoConstants)
    return consts, scb


def test_blur_invariants(oracle):
    W, H = 64, 48
    _, scb = _flat_case(W, H)
    rng = np.random.default_rng(5)
    depth = np.full((H, W), 8000000, dtype=np.uint32)
    normal = np.zeros((H, W, 4), dtype=np.float16); normal[..., 2] = -1
    const = np.full((H // 2, W // 2), 31337, dtype=np.uint16)
    for horz in (True, False):   # constant in => identical out
        assert np.array_equal(oracle.blur(scb, normal, depth, const, horz), const)
    # a pixel whose neighbours all fail the depth test passes through unchanged (c/W = w5*a/w5)
    amb = rng.integers(0, 65536, size=(H // 2, W // 2), dtype=np.uint16)
    depth2 = depth.copy()
    depth2[20:22, 30:32] = 2000000          # half-res pixel (15, 10) sits far in front of everything around it
    out = oracle.blur(scb, normal, depth2, amb, True)
    assert out[10, 15] == amb[10, 15]
    # symmetric weights, uniform geometry: interior of an H sweep over a ramp stays a ramp (linear => unchanged)
    ramp = (np.arange(W // 2, dtype=np.uint16)[None, :] * 1000 + np.zeros((H // 2, 1), dtype=np.uint16)).astype(np.uint16)
    out = oracle.blur(scb, normal, depth, ramp, True)
    assert np.abs(out[:, 6:-6].astype(np.int32) - ramp[:, 6:-6].astype(np.int32)).max() <= 1


def test_ssao_flat_wall_is_unoccluded(oracle):
    W, H = 96, 64
    consts, scb = _flat_case(W, H)
    depth = np.full((H, W), int(round(0.5 * 16777215)), dtype=np.uint32)
    normal = np.zeros((H, W, 4), dtype=np.float16); normal[..., 2] = -1
    out = oracle.ssao(scb, normal, depth, consts.randvec)
    assert (out[6:-6, 6:-6] == 65535).all()


def test_lighting_invariants(oracle):
    # black albedo + a light behind the surface: direct ~ 0, lit ~ shininess * F * cube
    import scene_util
    W, H = 16, 8
    pl = scene_util.cpu_scene(W, H, 128, 16)
    consts = pl["consts"]
    pcb = oracle_lib.as_oracle_cb(consts.pass_cb, oracle_lib.OrPassConstants)
    g0 = np.zeros((H, W, 4), np.float32); g0[..., :3] = (0, 0, 0); g0[..., 3] = 0.5
    g1 = np.zeros((H, W, 4), np.float32); g1[..., 3] = 0.5
    g2 = np.zeros((H, W, 4), np.float32); g2[..., :3] = (-0.57735, 0.57735, -0.57735); g2[..., 3] = 1  # N = light direction
    depth = np.full((H, W), 1000, dtype=np.uint32)
    shadow = np.full((4, 128, 128), 0xFFFFFF, dtype=np.uint32)
    cube = np.full((6, 16, 16, 4), 255, dtype=np.uint8)
    out, rad = oracle.deferred_light(pcb, g0, g1, g2, depth, None, shadow, cube, 1, 0.0, want_radiance=True)
    assert (out[..., 3] == 255).all() and np.isfinite(rad).all()
    # N.L <= 0 => nDotl clamps to 0.001: the direct term is tiny; reflection term dominates and is <= shininess
    assert rad[..., :3].max() <= 0.5 + 0.2 and rad[..., :3].min() > 0.0
    # uncovered pixels take the clear colour
    depth[:] = 0xFFFFFF
    out = oracle.deferred_light(pcb, g0, g1, g2, depth, None, shadow, cube, 1, 0.0)
    assert (out == np.array([176, 196, 222, 255], dtype=np.uint8)).all()


def test_shadow_compare_sampler(oracle):
    # LESS_EQUAL comparison then bilinear, border 0 (Appendix D)
    dim = 8
    s = np.full((dim, dim), 0xFFFFFF, dtype=np.uint32)
    s[:, :4] = 0                                       # left half at depth 0
    f = lambda u, v, ref: oracle.lib.or_sample_shadow_cmp(s.ctypes.data, dim, u, v, ref)
    assert f(0.75, 0.5, 0.5) == 1.0 and f(0.25, 0.5, 0.5) == 0.0
    assert abs(f(0.5, 0.5, 0.5) - 0.5) < 1e-7        # on the edge between texel 3 and 4 centres +0.5
    assert f(1.5, 0.5, 0.5) == 0.0                    # border texels compare as 0 (shadowed outside the map)
    assert f(0.75, 0.5, 0.0) == 1.0 and f(0.25, 0.5, 0.0) == 1.0  # ref 0 <= 0
