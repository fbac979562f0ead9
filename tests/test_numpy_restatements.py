"""Independent second opinion on the three hot-path shaders (SURVEY.md 8c iii): vectorised numpy restatements of
Ssao.hlsl:PS, SsaoBlur.hlsl:PS and DeferredShading.hlsl:PS written from the HLSL text in REAL arithmetic (float64, numpy's
own sin / pow / sqrt, no fused-anything, none of the oracle's helper definitions), compared with the C oracle under a stated
tolerance.  They check TRANSCRIPTION (a wrong sign, a swapped matrix index, a missing term, a sampler rule) and bound how far
the oracle's binary32 evaluation order (or_math.h, version 2) strays from the mathematics -- they do not check bits; the bit
contract is oracle == kernels (test_hostsim_parity.py, test_gpu_parity.py).

The reference holds no vectors of its own, so this is the only outside anchor for SSAO and lighting besides the App. C KATs."""
import numpy as np
import pytest

import oracle_lib
import scene_util

F = np.float64


# ---- D3D sampler rules, stated once more from scratch ---------------------------------------------
def bilinear(plane, u, v, mode, border=0.0):
    """plane: (H, W) or (H, W, C) float64; texel centres at (i + .5) / dim; weights (1 - f, f) from t = uv * dim - .5."""
    H, W = plane.shape[:2]
    tx, ty = u * W - 0.5, v * H - 0.5
    x0, y0 = np.floor(tx), np.floor(ty)
    fx, fy = tx - x0, ty - y0
    x0, y0 = x0.astype(np.int64), y0.astype(np.int64)

    def texel(xi, yi):
        if mode == "wrap":
            return plane[yi % H, xi % W]
        t = plane[np.clip(yi, 0, H - 1), np.clip(xi, 0, W - 1)]
        if mode == "border":
            inside = (xi >= 0) & (xi < W) & (yi >= 0) & (yi < H)
            t = np.where(inside if t.ndim == inside.ndim else inside[..., None], t, border)
        return t

    if plane.ndim == 3:
        fx, fy = fx[..., None], fy[..., None]
    top = texel(x0, y0) * (1 - fx) + texel(x0 + 1, y0) * fx
    bot = texel(x0, y0 + 1) * (1 - fx) + texel(x0 + 1, y0 + 1) * fx
    return top * (1 - fy) + bot * fy


def mul_row(v4, mem):
    """HLSL mul(rowvec, M) with M stored transposed (CRYCHIC.cpp:843-849): out[j] = sum_i v[i] * mem[4 j + i]."""
    M = np.asarray(mem, dtype=F).reshape(4, 4)
    return v4 @ M.T


def unorm_write(x, n):
    return np.floor(np.clip(np.nan_to_num(x, nan=0.0), 0.0, 1.0) * n + 0.5).astype(np.int64)


def normalize(v):
    return v / np.sqrt((v * v).sum(axis=-1, keepdims=True))


# ---- Ssao.hlsl:PS (Shaders/Ssao.hlsl:58-72, 76-108, 110-115, 117-199) --------------------------------------------------
def numpy_ssao(cb, normal_f16, depth_u32, randvec_u8):
    H, W = depth_u32.shape
    h2, w2 = H // 2, W // 2
    depth = (depth_u32 & 0xFFFFFF).astype(F) / 16777215.0
    nrm = normal_f16.astype(F)[..., :3]
    rnd = randvec_u8.astype(F)[..., :3] / 255.0
    A, B = F(cb.Proj[4 * 2 + 2]), F(cb.Proj[4 * 2 + 3])            # gProj[2][2], gProj[3][2]: M[r][c] = mem[4 c + r]
    ys, xs = np.mgrid[0:h2, 0:w2]
    u, v = (xs + 0.5) / w2, (ys + 0.5) / h2                            # TexC at the pixel centre
    posh = np.stack([2 * u - 1, 1 - 2 * v, np.zeros_like(u), np.ones_like(u)], axis=-1)
    ph = mul_row(posh, cb.InvProj)
    PosV = ph[..., :3] / ph[..., 3:4]
    n = normalize(nrm[np.clip(np.floor(v * H).astype(int), 0, H - 1), np.clip(np.floor(u * W).astype(int), 0, W - 1)])   # point / clamp
    pz = B / (bilinear(depth, u, v, "border", 1.0) - A)
    p = (pz / PosV[..., 2])[..., None] * PosV
    randVec = 2 * bilinear(rnd, 4 * u, 4 * v, "wrap") - 1
    occ_sum = np.zeros((h2, w2))
    eps, f0, f1, R = F(cb.SurfaceEpsilon), F(cb.OcclusionFadeStart), F(cb.OcclusionFadeEnd), F(cb.OcclusionRadius)
    for i in range(14):
        o = np.array([cb.OffsetVectors[i][k] for k in range(3)], dtype=F)
        offset = o - 2 * (randVec @ o)[..., None] * randVec          # reflect(o, randVec)
        flip = np.sign((offset * n).sum(-1))
        q = p + (flip * R)[..., None] * offset
        pq = mul_row(np.concatenate([q, np.ones((h2, w2, 1))], axis=-1), cb.ProjTex)
        pq = pq / pq[..., 3:4]
        rz = B / (bilinear(depth, pq[..., 0], pq[..., 1], "border", 1.0) - A)
        r = (rz / q[..., 2])[..., None] * q
        distZ = p[..., 2] - r[..., 2]
        with np.errstate(invalid="ignore", divide="ignore"):
            dp = np.maximum(np.nan_to_num((n * normalize(r - p)).sum(-1), nan=0.0), 0.0)
        occl = np.where(distZ > eps, np.clip((f1 - distZ) / (f1 - f0), 0.0, 1.0), 0.0)
        occ_sum += dp * occl
    access = 1.0 - occ_sum / 14.0
    return unorm_write(access ** 6, 65535)


# ---- SsaoBlur.hlsl:PS (Shaders/SsaoBlur.hlsl:85-146) ------------------------------------------------------------------------
def numpy_blur(cb, normal_f16, depth_u32, amb_u16, horizontal):
    H, W = depth_u32.shape
    h2, w2 = H // 2, W // 2
    w = np.array([cb.BlurWeights[i // 4][i % 4] for i in range(12)], dtype=F)
    A, B = F(cb.Proj[10]), F(cb.Proj[11])
    depth = (depth_u32 & 0xFFFFFF).astype(F) / 16777215.0
    nrm = normal_f16.astype(F)[..., :3]
    amb = amb_u16.astype(F) / 65535.0
    ys, xs = np.mgrid[0:h2, 0:w2]
    u0, v0 = (xs + 0.5) / w2, (ys + 0.5) / h2

    def taps(i):
        # tex = TexC + i * texOffset with texOffset = one half-res pixel along the sweep axis (SsaoBlur.hlsl:95-103)
        tx, ty = (xs + i, ys) if horizontal else (xs, ys + i)
        u, v = (tx + 0.5) / w2, (ty + 0.5) / h2
        fx, fy = np.clip(2 * tx + 1, 0, W - 1), np.clip(2 * ty + 1, 0, H - 1)     # point / clamp on the full-res normal map
        nz = B / (bilinear(depth, u, v, "border", 1.0) - A)
        return nrm[fy, fx], nz, amb[np.clip(ty, 0, h2 - 1), np.clip(tx, 0, w2 - 1)]

    cn, cz, ca = taps(0)
    color, total = w[5] * ca, np.full((h2, w2), w[5])
    for i in range(-5, 6):
        if i == 0:
            continue
        nn, nz, na = taps(i)
        ok = ((nn * cn).sum(-1) >= 0.8) & (np.abs(nz - cz) <= 0.2)
        color = color + np.where(ok, w[i + 5] * na, 0.0)
        total = total + np.where(ok, w[i + 5], 0.0)
    del u0, v0
    return unorm_write(color / total, 65535)


# ---- DeferredShading.hlsl:PS with PBR.hlsl, GBuffer.hlsl, LightingUtil.hlsl:52-60, Common.hlsl:263-317 (radius 0) ----------------
def cube_sample(cube_u8, d):
    """TextureCube.Sample, linear: D3D major-axis face selection, bilinear inside the face (clamp at the face edge)."""
    dim = cube_u8.shape[1]
    faces = cube_u8.astype(F) / 255.0
    ax, ay, az = np.abs(d[..., 0]), np.abs(d[..., 1]), np.abs(d[..., 2])
    isx, isy = (ax >= ay) & (ax >= az), ~((ax >= ay) & (ax >= az)) & (ay >= az)
    x, y, z = d[..., 0], d[..., 1], d[..., 2]
    face = np.where(isx, np.where(x >= 0, 0, 1), np.where(isy, np.where(y >= 0, 2, 3), np.where(z >= 0, 4, 5)))
    ma = np.where(isx, ax, np.where(isy, ay, az))
    sc = np.where(isx, np.where(x >= 0, -z, z), np.where(isy, x, np.where(z >= 0, x, -x)))
    tc = np.where(isx, -y, np.where(isy, np.where(y >= 0, z, -z), -y))
    u, v = 0.5 * (sc / ma + 1), 0.5 * (tc / ma + 1)
    out = np.zeros(d.shape[:-1] + (4,))
    for f in range(6):
        m = face == f
        if m.any():
            out[m] = bilinear(faces[f], u[m], v[m], "clamp")
    return out


def shadow_cmp(smap, u, v, ref):
    """SampleCmpLevelZero, LESS_EQUAL, linear, border 0: compare each texel, then filter."""
    dim = smap.shape[0]
    tx, ty = u * dim - 0.5, v * dim - 0.5
    x0, y0 = np.floor(tx), np.floor(ty)
    fx, fy = tx - x0, ty - y0
    x0, y0 = x0.astype(np.int64), y0.astype(np.int64)

    def c(xi, yi):
        inside = (xi >= 0) & (xi < dim) & (yi >= 0) & (yi < dim)
        t = np.where(inside, smap[np.clip(yi, 0, dim - 1), np.clip(xi, 0, dim - 1)], 0.0)
        return (ref <= t).astype(F)

    top = c(x0, y0) * (1 - fx) + c(x0 + 1, y0) * fx
    bot = c(x0, y0 + 1) * (1 - fx) + c(x0 + 1, y0 + 1) * fx
    return top * (1 - fy) + bot * fy


def numpy_light(cb, g0, g1, g2, depth_u32, ambient_u16, shadow_u32, cube_u8, num_dir_lights):
    """Covered pixels only; PCF with the radius the shader computes as written (Common.hlsl:305: 5 / width = 0 in uint)."""
    H, W = depth_u32.shape
    covered = (depth_u32 & 0xFFFFFF) < 0xFFFFFF
    G0, G1, G2 = g0.astype(F), g1.astype(F), g2.astype(F)
    pos, metal = G0[..., :3], G0[..., 3]
    albedo, rough = G1[..., :3], G1[..., 3]
    with np.errstate(invalid="ignore", divide="ignore"):
        N = normalize(G2[..., :3])
        eye = np.array(list(cb.EyePosW), dtype=F)
        toEye = eye - pos
        V = normalize(toEye)
        R0 = 0.04 + (albedo - 0.04) * metal[..., None]
        pos4 = np.concatenate([pos, np.ones((H, W, 1))], axis=-1)
        sp = mul_row(pos4, cb.ViewProjTex)
        amb = ambient_u16.astype(F) / 65535.0
        access = bilinear(amb, sp[..., 0] / sp[..., 3], sp[..., 1] / sp[..., 3], "clamp")
        ambient = access[..., None] * np.array(list(cb.AmbientLight), dtype=F)[:3] * albedo
        dist = np.sqrt((toEye * toEye).sum(-1))
        smaps = [(shadow_u32[k] & 0xFFFFFF).astype(F) / 16777215.0 for k in range(4)]
        pcf = []
        for k in range(4):
            s = mul_row(pos4, cb.ShadowTransforms[k])
            s = s / s[..., 3:4]
            pcf.append(shadow_cmp(smaps[k], s[..., 0], s[..., 1], s[..., 2]))      # 16 coincident taps / 16
        # `abs(distance - radius[j] < 5.0f)` is abs(bool): the blend branch runs whenever j < 3 and distance < radius[j]
        shadow0 = np.ones((H, W))
        shadow0 = np.where(dist < 100, pcf[3], shadow0)
        shadow0 = np.where(dist < 80, 0.5 * (pcf[2] + pcf[3]), shadow0)
        shadow0 = np.where(dist < 50, 0.5 * (pcf[1] + pcf[2]), shadow0)
        shadow0 = np.where(dist < 30, 0.5 * (pcf[0] + pcf[1]), shadow0)
        shin = (1 - rough) * 1.0
        PI = 3.1415926
        direct = np.zeros((H, W, 3))
        for i in range(num_dir_lights):
            Lg = cb.Lights[i]
            L = -np.array(list(Lg.Direction), dtype=F)
            Hv = normalize(V + L)
            hDotv = np.maximum((Hv * V).sum(-1), 0.001)
            nDotl = np.maximum((N * L).sum(-1), 0.001)
            nDotv = np.maximum((N * V).sum(-1), 0.001)
            a2 = rough * rough
            nDoth = np.maximum((N * Hv).sum(-1), 0.001)
            D = a2 / (PI * (nDoth * nDoth * (a2 - 1) + 1) ** 2)
            k = 0.125 * (rough + 1) * (rough + 1)
            Gs = (nDotv / (nDotv * (1 - k) + k)) * (nDotl / (nDotl * (1 - k) + k))
            Fr = R0 + (1 - R0) * (np.clip(1 - hDotv, 0, 1) ** 5)[..., None]       # GetBRDF: nDotv := hDotv (PBR.hlsl:58)
            fs = 0.25 * (D * Gs)[..., None] * Fr / (nDotl * hDotv)[..., None]
            fd = albedo / PI
            brdf = (1 - Fr) * (1 - metal)[..., None] * fd + Fr * fs
            sf = shadow0 if i == 0 else np.ones((H, W))
            direct += (sf ** 5)[..., None] * brdf * (np.array(list(Lg.Strength), dtype=F) * nDotl[..., None])
        direct = direct / (direct + 1)
        direct = np.where(direct > 0, np.abs(direct) ** (1 / 2.2), 0.0)
        lit = direct + ambient
        r = -V - 2 * ((N * -V).sum(-1))[..., None] * N                          # reflect(-view, N)
        refl = cube_sample(cube_u8, r)[..., :3]
        f0 = 1 - np.clip((N * r).sum(-1), 0, 1)
        lit = lit + (shin * 1.0)[..., None] * (R0 + (1 - R0) * (f0 ** 5)[..., None]) * refl
    lit = np.concatenate([lit, np.ones((H, W, 1))], axis=-1)
    return lit, covered


# ---- the comparisons ---------------------------------------------------------------------------------------------------------
CASES = [(64, 64), (130, 34)]


def _scene(W, H):
    pl = scene_util.cpu_scene(W, H, 512, 64)
    p = scene_util.np_planes(pl)
    c = pl["consts"]
    return p, c, oracle_lib.as_oracle_cb(c.ssao_cb, oracle_lib.OrSsaoConstants), oracle_lib.as_oracle_cb(c.pass_cb, oracle_lib.OrPassConstants)


@pytest.mark.parametrize("W,H", CASES)
def test_numpy_ssao_agrees_with_oracle(oracle, W, H):
    """Tolerance: a binary32 evaluation lands on the other side of an R16 rounding boundary for a few percent of texels and,
    rarely, of a `distZ > eps` / sign() decision (pow(access, 6) then amplifies it): >= 90 % identical, >= 97 % within 1 LSB,
    >= 99.5 % within 32 LSB, mean |difference| < 0.5 LSB."""
    p, c, scb, _ = _scene(W, H)
    ref = oracle.ssao(scb, p["normal"], p["depth"], p["randvec"]).astype(np.int64)
    got = numpy_ssao(scb, p["normal"], p["depth"], p["randvec"])
    d = np.abs(got - ref)
    assert ref.min() < 60000 < ref.max(), "the case must exercise occlusion"
    assert (d == 0).mean() >= 0.90 and (d <= 1).mean() >= 0.97 and (d <= 32).mean() >= 0.995 and d.mean() < 0.5, (
        (d == 0).mean(), (d <= 1).mean(), (d <= 32).mean(), d.mean(), d.max())


@pytest.mark.parametrize("W,H", CASES)
def test_numpy_blur_agrees_with_oracle(oracle, W, H):
    """Random ambient input (every tap matters).  >= 85 % identical, >= 99.5 % within 1 LSB (the rest: `dot >= 0.8` /
    `|dz| <= 0.2` decisions that flip between binary32 and real arithmetic)."""
    p, c, scb, _ = _scene(W, H)
    amb = np.random.default_rng(99).integers(0, 65536, size=(H // 2, W // 2), dtype=np.uint16)
    for horz in (True, False):
        ref = oracle.blur(scb, p["normal"], p["depth"], amb, horz).astype(np.int64)
        got = numpy_blur(scb, p["normal"], p["depth"], amb, horz)
        d = np.abs(got - ref)
        assert (d == 0).mean() >= 0.85 and (d <= 1).mean() >= 0.995, (horz, (d == 0).mean(), (d <= 1).mean(), d.max())


@pytest.mark.parametrize("W,H", CASES)
@pytest.mark.parametrize("lights", [1, 3])
def test_numpy_lighting_agrees_with_oracle(oracle, W, H, lights):
    """Radiance: |difference| <= 1e-5 + 1e-4 |ref| on >= 99.8 % of covered pixels (the rest sit on a PCF
    compare or cascade-distance decision); RGBA8: within 1 LSB on >= 99.8 %, identical on >= 97 %."""
    p, c, scb, pcb = _scene(W, H)
    ao = oracle.compute_ssao(scb, p["normal"], p["depth"], p["randvec"], 2)
    ref8, ref = oracle.deferred_light(pcb, p["g0"], p["g1"], p["g2"], p["depth"], ao, p["shadow"], p["cube"], lights, 0.0, want_radiance=True)
    lit, covered = numpy_light(pcb, p["g0"], p["g1"], p["g2"], p["depth"], ao, p["shadow"], p["cube"], lights)
    assert 0.2 < covered.mean() < 1.0
    rr, gg = ref[covered].astype(F), lit[covered]
    close = (np.abs(gg - rr) <= 1e-5 + 1e-4 * np.abs(rr)).all(axis=-1)
    assert close.mean() >= 0.998, (close.mean(), np.abs(gg - rr).max())
    d8 = np.abs(unorm_write(gg, 255) - ref8[covered].astype(np.int64))
    assert (d8 <= 1).mean() >= 0.998 and (d8 == 0).mean() >= 0.97, ((d8 <= 1).mean(), (d8 == 0).mean(), d8.max())


def test_zero_normal_is_a_definition(oracle):
    """Where the oracle DEFINES rather than restates (oracle/or_math.h "definitions that are not restatements"): normalize() of a
    zero vector.  The HLSL text gives 0 * rsqrt(0) = NaN, which saturate() / max() turn into 0: SSAO over a zero view normal has
    dp = max(dot(NaN, .), 0) = 0 for every tap => access 1 (the same 65535 either way), but a zero G-buffer normal lights to
    NaN radiance => BLACK after the UNORM write on D3D.  The oracle clamps the squared length first, so the normal stays (0,0,0):
    finite radiance, a visible pixel.  This test states both sides, so the documented difference cannot drift unnoticed."""
    W, H = 64, 64
    p, c, scb, pcb = _scene(W, H)
    g2 = p["g2"].copy()
    depth = p["depth"]
    zero = np.zeros((H, W), dtype=bool); zero[H // 2:, 16:48] = True
    zero &= (depth & 0xFFFFFF) != 0xFFFFFF                               # covered pixels only
    assert zero.sum() > 200
    g2[zero, :3] = 0.0
    ao = np.full((H // 2, W // 2), 65535, dtype=np.uint16)
    ref8, ref = oracle.deferred_light(pcb, p["g0"], p["g1"], g2, depth, ao, p["shadow"], p["cube"], 3, 0.0, want_radiance=True)
    with np.errstate(all="ignore"):
        lit, covered = numpy_light(pcb, p["g0"], p["g1"], g2, depth, ao, p["shadow"], p["cube"], 3)
    # the HLSL text: NaN radiance on the zero-normal pixels, black after the UNORM write
    assert np.isnan(lit[zero][:, :3]).all()
    assert (unorm_write(lit[zero][:, :3], 255) == 0).all()
    # the oracle's definition: finite radiance there (ambient + tone-mapped direct with a zero normal), not black
    assert np.isfinite(ref[zero]).all() and (ref8[zero][:, :3] > 0).all()
    # and everywhere else the two still agree as in test_numpy_lighting_agrees_with_oracle
    rest = covered & ~zero
    rr, gg = ref[rest].astype(F), lit[rest]
    assert (np.abs(gg - rr) <= 1e-5 + 1e-4 * np.abs(rr)).all(axis=-1).mean() >= 0.998
    # SSAO: a zero view normal gives access 1 under both readings
    normal = p["normal"].copy(); normal[:] = 0
    got = oracle.ssao(scb, normal, p["depth"], p["randvec"])
    assert (got == 65535).all()
