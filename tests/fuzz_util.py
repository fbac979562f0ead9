"""Random (incoherent) input planes and perturbed constants for the fuzz parity tests: nothing here looks like a scene --
the point is to drive every sampler / special-value path (border, clamp, NaN, inf, zero-length vectors, huge shadow
coordinates) through oracle, kernel bodies and the device with the same bits."""
import ctypes as C

import numpy as np


def random_case(seed, built_lib, size=None):
    """size = (W, H): force the frame size (the random draws that follow are the same either way)."""
    from crychic_renderer_amd import scene
    rng = np.random.default_rng(seed)
    W = int(rng.integers(17, 100)) * 2
    H = int(rng.integers(17, 70)) * 2
    if size is not None:
        W, H = size
    sd = int(rng.choice([16, 64, 130]))
    cd = int(rng.choice([2, 8, 33]))
    c = scene.Constants(W, H, shadow_dim=sd)
    # --- planes -------------------------------------------------------------------------------------------------
    depth = rng.integers(0, 1 << 24, size=(H, W), dtype=np.uint32)
    depth[rng.random((H, W)) < 0.25] = 0xFFFFFF                         # uncovered
    depth[rng.random((H, W)) < 0.05] = 0
    if rng.random() < 0.35 and W > 16 and H > 16:
        # a smooth depth field (tilted plane + a little noise) with spikes, holes and sky patches: the SSAO tap culling gets
        # blocks it can cull and blocks a single texel spoils
        yy, xx = np.mgrid[0:H, 0:W]
        lo, hi = sorted(rng.integers(1 << 20, (1 << 24) - 1, 2).tolist())
        ramp = lo + (hi - lo) * ((yy * float(rng.random()) + xx * float(rng.random())) / float(H + W))
        depth = (ramp + rng.integers(-40, 41, size=(H, W))).clip(0, 0xFFFFFE).astype(np.uint32)
        k = int(rng.integers(5, 60))
        depth[rng.integers(0, H, k), rng.integers(0, W, k)] = rng.integers(0, 1 << 24, k).astype(np.uint32)
        for _ in range(int(rng.integers(0, 5))):
            y0, x0 = int(rng.integers(0, H - 8)), int(rng.integers(0, W - 8))
            depth[y0:y0 + int(rng.integers(1, 9)), x0:x0 + int(rng.integers(1, 9))] = 0xFFFFFF
    depth |= rng.integers(0, 256, size=(H, W), dtype=np.uint32) << 24      # stencil byte must be ignored
    nrm = rng.standard_normal((H, W, 4)).astype(np.float16)
    sel = rng.random((H, W))
    nrm[sel < 0.03] = 0.0
    nrm[(sel >= 0.03) & (sel < 0.04), 0] = np.nan
    nrm[(sel >= 0.04) & (sel < 0.05), 1] = np.inf
    nrm[(sel >= 0.05) & (sel < 0.30)] = np.array([0.0, 0.0, -1.0, 0.0], dtype=np.float16)   # flat patches: blur accepts
    g0 = (rng.standard_normal((H, W, 4)) * 20.0).astype(np.float32); g0[..., 3] = rng.random((H, W))
    if rng.random() < 0.4:
        # coherent world positions (a tilted plane receding from the eye, a little noise): whole wavefronts then fall into one
        # shadow cascade, which is what the wave-uniform cascade path of the lighting kernel needs to run at all
        yy, xx = np.mgrid[0:H, 0:W].astype(np.float32)
        far = float(rng.choice([25.0, 60.0, 120.0]))
        eye = np.array(list(c.cam.pos), dtype=np.float32)
        g0[..., 0] = eye[0] + (xx / W - 0.5) * 6.0
        g0[..., 1] = eye[1] - 1.0 + rng.standard_normal((H, W)).astype(np.float32) * 0.05
        g0[..., 2] = eye[2] + 2.0 + (1.0 - yy / H) * far
    g1 = rng.random((H, W, 4)).astype(np.float32)
    g2 = rng.standard_normal((H, W, 4)).astype(np.float32)
    sel = rng.random((H, W))
    g2[sel < 0.02, :3] = 0.0
    g0[(sel >= 0.02) & (sel < 0.03), 0] = np.inf
    g0[(sel >= 0.03) & (sel < 0.04), 1] = np.nan
    g0[(sel >= 0.04) & (sel < 0.05), :3] = 1e30
    g1[(sel >= 0.05) & (sel < 0.06), 3] = 0.0
    g1[(sel >= 0.06) & (sel < 0.07), :3] = -2.0
    g0[(sel >= 0.07) & (sel < 0.08), :3] = np.array(list(c.cam.pos), dtype=np.float32)
    g1[(sel >= 0.08) & (sel < 0.09), 3] = np.nan
    shadow = rng.integers(0, 1 << 24, size=(4, sd, sd), dtype=np.uint32)
    shadow[:, : sd // 2] |= 0x00F00000                                     # mostly-lit half so compares go both ways
    cube = rng.integers(0, 256, size=(6, cd, cd, 4), dtype=np.uint8)
    randvec = rng.integers(0, 256, size=(256, 256, 4), dtype=np.uint8)
    planes = {"depth": depth, "normal": nrm, "g0": g0, "g1": g1, "g2": g2, "shadow": shadow, "cube": cube, "randvec": randvec}
    # --- constants ----------------------------------------------------------------------------------------------
    s, p = c.ssao_cb, c.pass_cb
    s.OcclusionRadius = float(rng.choice([0.05, 0.5, 3.0]))
    s.OcclusionFadeStart = float(rng.choice([0.0, 0.2, 1.0]))
    s.OcclusionFadeEnd = float(rng.choice([1.0, 2.0, s.OcclusionFadeStart]))     # start == end: division by zero
    s.SurfaceEpsilon = float(rng.choice([0.0, 0.05, 0.5]))
    if rng.random() < 0.35:            # a sheared projection: the kernels must leave their sparse-gProjTex fast path
        s.ProjTex[1] = float(rng.choice([0.05, -0.2]))
        s.ProjTex[13] = float(rng.choice([0.0, 0.01]))
    for i in range(3):
        d = rng.standard_normal(3)
        d = d / np.linalg.norm(d) if rng.random() > 0.1 else np.zeros(3)
        p.Lights[i].Direction[:] = [float(x) for x in d]
        p.Lights[i].Strength[:] = [float(x) for x in rng.choice([0.0, 0.1, 1.0, 2.4], size=3)]
    p.AmbientLight[:] = [float(x) for x in rng.random(4)]
    # shadow transforms: keep the cascade matrices but shrink them so random world positions land inside the maps
    for k in range(4):
        for j in range(16):
            p.ShadowTransforms[k][j] = float(p.ShadowTransforms[k][j] * rng.choice([0.2, 1.0]))
    knobs = {
        "blurCount": int(rng.integers(0, 7)), "numDirLights": int(rng.integers(0, 4)),
        "pcfSearchRadius": float(rng.choice([0.0, 2.5 / sd, 10.0 / sd])), "sky": int(rng.integers(0, 2)),
        "ssao_on": bool(rng.random() > 0.15),
    }
    return W, H, planes, c, knobs


def same_floats(a, b):
    """Bit equality, except that any NaN equals any NaN (x86 and gfx950 disagree on the sign / payload of generated NaNs)."""
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    an, bn = np.isnan(a), np.isnan(b)
    return bool(np.array_equal(an, bn) and np.array_equal(a.view(np.uint32)[~an], b.view(np.uint32)[~bn]))


def sky_probe_case(seed):
    """Inputs for the SSAO sky-shortcut probe (tests/test_hostsim_parity.py::test_sky_shortcut_is_exact and its device twin): a sky
    field with small patches of geometry within OcclusionFadeEnd of the far distance."""
    import oracle_lib
    from crychic_renderer_amd import scene
    rng = np.random.default_rng(1000 + seed)
    W, H = int(rng.choice([640, 770, 1024])), int(rng.choice([384, 400, 512]))       # several cells of the coarse geometry map each way
    c = scene.Constants(W, H, shadow_dim=64)
    scb = oracle_lib.as_oracle_cb(c.ssao_cb, oracle_lib.OrSsaoConstants)
    A, B = c.ssao_cb.Proj[10], c.ssao_cb.Proj[11]
    depth = np.full((H, W), 0xFFFFFF, dtype=np.uint32)
    # sky normals facing the camera (with the reference's clear value (0, 0, 1) an occluder in front of a far-plane pixel always
    # has dp = 0): random unit-ish vectors or (0, 0, -1), so that a tap landing on a patch really contributes
    normal = np.zeros((H, W, 4), dtype=np.float16); normal[..., 2] = -1.0
    if seed % 2:
        normal[..., :3] = rng.standard_normal((H, W, 3)).astype(np.float16)
    for _ in range(int(rng.integers(1, 5))):
        pw, ph = int(rng.integers(1, 24)), int(rng.integers(1, 16))
        x0, y0 = int(rng.integers(0, W - pw)), int(rng.integers(0, H - ph))
        if seed >= 12:      # hug the corner of a cell of the coarse geometry map (128 x 32 texels, columns offset by -2): worst case for the reach bound
            pw, ph = int(rng.integers(1, 3)), int(rng.integers(1, 3))
            x0 = min(W - pw, 128 * int(rng.integers(1, W // 128)) - 2 - int(rng.integers(0, 2)) * pw)
            y0 = min(H - ph, 32 * int(rng.integers(1, H // 32)) - int(rng.integers(0, 2)) * ph)
        vz = rng.uniform(99.0, 99.999, size=(ph, pw))                       # inside the occluding band of a pixel at the far distance
        zndc = A + B / vz
        depth[y0:y0 + ph, x0:x0 + pw] = np.round(zndc * 16777215.0).astype(np.uint32)
        normal[y0:y0 + ph, x0:x0 + pw, :3] = rng.standard_normal((ph, pw, 3)).astype(np.float16)
    if seed % 4 == 3:
        normal[rng.random((H, W)) < 0.01, 0] = np.inf                        # a non-finite normal keeps its wavefront off the shortcut
    randvec = rng.integers(0, 256, size=(256, 256, 4), dtype=np.uint8)
    if seed >= 12:
        randvec = (rng.integers(0, 2, size=(256, 256, 4), dtype=np.uint8) * 255).astype(np.uint8)      # |randVec| = sqrt(3): the longest reflected offsets
    return W, H, c, scb, depth, normal, randvec


def clear_cell_probe_case(seed):
    """Inputs for the clear-cell rule of the SSAO depth pass (ssao_core.hpp "clear cells"): sky next to geometry that hugs the near
    plane, an occlusion radius that puts taps at and behind the camera (q.z < 1e-3: never culled, landing anywhere -- mostly beyond
    the plane's edge, which is clear by definition) and a few non-finite normals (pixels that may not cull at all)."""
    import oracle_lib
    W, H, c, scb, depth, normal, randvec = sky_probe_case(seed)
    rng = np.random.default_rng(77 + seed)
    A, B = c.ssao_cb.Proj[10], c.ssao_cb.Proj[11]
    x0, x1 = W // 5, W // 5 + 90 + 8 * seed
    vz = rng.uniform(1.0, 1.8, size=(H // 2, x1 - x0))
    depth[H // 4:H // 4 + H // 2, x0:x1] = np.clip(np.round((A + B / vz) * 16777215.0), 0, 0xFFFFFF).astype(np.uint32)
    normal[H // 4:H // 4 + H // 2, x0:x1, :3] = rng.standard_normal((H // 2, x1 - x0, 3)).astype(np.float16)
    normal[rng.random((H, W)) < 0.002, 1] = np.nan
    c.ssao_cb.OcclusionRadius = 3.0
    scb = oracle_lib.as_oracle_cb(c.ssao_cb, oracle_lib.OrSsaoConstants)
    return W, H, c, scb, depth, normal, randvec


def cull_probe_case(seed, W=256, H=160):
    """Frames for the SSAO tap culling (ssao_core.hpp): the reference scene -- open ground, where most taps are culled -- with what
    the nearest-depth bound must not miss: single near texels (spikes a fraction of a block wide), texels just in front of their
    surroundings (around SurfaceEpsilon in view space), holes to the far plane, depth 0, a few small patches, NaN normals, and one
    of several SurfaceEpsilon values.  Returns (W, H, constants, depth, normal, randvec)."""
    import scene_util
    pl = scene_util.cpu_scene(W, H, 64, 8)
    p = scene_util.np_planes(pl)
    c = pl["consts"]
    rng = np.random.default_rng(1234 + seed)
    depth, normal = p["depth"].copy(), p["normal"].copy()
    if seed > 0:
        n = 60
        ys, xs = rng.integers(0, H, n), rng.integers(0, W, n)
        kind = rng.integers(0, 4, n)
        base = depth[ys, xs] & 0xFFFFFF
        spike = np.where(kind == 0, rng.integers(0, 1 << 22, n),
                np.where(kind == 1, base - rng.integers(0, 200, n).clip(max=base),
                np.where(kind == 2, 0xFFFFFF, 0))).astype(np.uint32)
        depth[ys, xs] = spike
        for _ in range(6):
            y0, x0 = int(rng.integers(0, H - 4)), int(rng.integers(0, W - 4))
            depth[y0:y0 + int(rng.integers(1, 5)), x0:x0 + int(rng.integers(1, 5))] = int(rng.integers(0, 1 << 24))
        normal[rng.integers(0, H, 40), rng.integers(0, W, 40), 0] = np.nan
        c.ssao_cb.SurfaceEpsilon = float([0.05, 0.0, 0.5, 0.05, 1e-4, 0.05][seed % 6])
    return W, H, c, depth, normal, p["randvec"]
