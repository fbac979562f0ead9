"""The sky cube map's mip chain.  The reference binds the whole chain of snowcube1024.dds
(CRYCHIC.cpp:1148-1151, MipLevels = GetDesc().MipLevels) and samples it MIN_MAG_MIP_LINEAR (CRYCHIC.cpp:2617-2622) in the lighting
pass (Shaders/DeferredShading.hlsl:95) and the sky (Shaders/sky.hlsl:46).  The file is not in the checkout and D3D leaves the
level-of-detail arithmetic to the hardware, so everything here is against this repo's oracle definition
(oracle/or_samplers.h "TextureCube.Sample with a mip chain") -- parity unpinned -- plus known answers derived from that definition:
  * the loader's level-after-level layout (product == oracle twin, both header generations, refusals),
  * or_cube_lod on hand-computable derivatives (one texel per pixel -> 0, four -> 2, clamp at the last level, the chain rule),
  * a chain whose level k is the grey 16 k: the sky of a small frame shows higher levels than the sky of a larger one, between
    levels where the filter is trilinear,
  * kernel bodies (hostsim) == oracle and device == oracle on a lit frame: reflections, sky, point lights, strips."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
from test_point_lights import as_or_lights
from test_textures import cube_header


def grey_chain(dim, levels=None):
    """Level k of every face is the grey 16 k (alpha 255 - k): which level a lookup used can be read off the colour."""
    n = int(np.log2(dim)) + 1 if levels is None else levels
    parts = []
    for k in range(n):
        d = max(dim >> k, 1)
        lv = np.empty((6, d, d, 4), np.uint8)
        lv[..., :3] = 16 * k
        lv[..., 3] = 255 - k
        parts.append(lv.reshape(-1))
    return np.concatenate(parts), n


def test_cube_chain_loader(built_lib, oracle, tmp_path):
    """crychic_load_dds_cube_rgba8_mips: the file stores face after face, each with its chain; the product wants level after level,
    each with its six faces.  Known answers (face k, level l -> 10 k + l), == the oracle's twin, DXT1 behind both headers, refusals."""
    from crychic_renderer_amd import geometry as g
    lib = built_lib.lib
    L = oracle.lib
    L.or_load_dds_cube_rgba8_mips.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]

    def oracle_chain(path):
        d, m = C.c_uint32(), C.c_uint32()
        assert L.or_load_dds_cube_rgba8_mips(path.encode(), None, 0, C.byref(d), C.byref(m)) == 0
        out = np.zeros(sum(6 * max(d.value >> k, 1) ** 2 * 4 for k in range(m.value)), np.uint8)
        assert L.or_load_dds_cube_rgba8_mips(path.encode(), out.ctypes.data, out.nbytes, C.byref(d), C.byref(m)) == 0
        return out, d.value, m.value

    def chain(face, dim, levels):
        out = b""
        for lv in range(levels):
            d = max(1, dim >> lv)
            px = np.zeros((d, d, 4), np.uint8)
            px[..., 0], px[..., 1], px[..., 2], px[..., 3] = 10 * face + lv, 100 + face, 200 + lv, 255 - face      # memory order B, G, R, A
            out += px.tobytes()
        return out
    p = tmp_path / "cube_argb.dds"
    p.write_bytes(cube_header(4, 3, None, (0xFF0000, 0xFF00, 0xFF, 0xFF000000)) + b"".join(chain(k, 4, 3) for k in range(6)))
    flat, dim, levels = g.load_dds_cube_mips(str(p))
    assert (dim, levels) == (4, 3) and flat.size == 6 * (16 + 4 + 1) * 4
    off = 0
    for lv in range(3):
        d = 4 >> lv
        level = flat[off:off + 6 * d * d * 4].reshape(6, d, d, 4)
        for k in range(6):
            assert (level[k] == np.array([200 + lv, 100 + k, 10 * k + lv, 255 - k], np.uint8)).all(), (lv, k)       # R, G, B, A
        off += 6 * d * d * 4
    of, od, ol = oracle_chain(str(p))
    assert (od, ol) == (4, 3) and np.array_equal(flat, of)
    assert np.array_equal(flat[:6 * 16 * 4].reshape(6, 4, 4, 4), g.load_dds_cube(str(p)))          # level 0 == the level-0 loader
    # DXT1, 8 x 8 faces with the full chain (8, 4, 2, 1), legacy and DX10 headers
    rng = np.random.default_rng(7)
    faces = [rng.integers(0, 256, sum(((max(1, 8 >> lv) + 3) // 4) ** 2 * 8 for lv in range(4)), dtype=np.uint8).tobytes() for _ in range(6)]
    legacy, dx10 = tmp_path / "c1.dds", tmp_path / "c1_dx10.dds"
    legacy.write_bytes(cube_header(8, 4, b"DXT1") + b"".join(faces))
    dx10.write_bytes(cube_header(8, 4, dx10=(71, 0x4)) + b"".join(faces))
    a, b = g.load_dds_cube_mips(str(legacy)), g.load_dds_cube_mips(str(dx10))
    assert a[1:] == (8, 4) and b[1:] == (8, 4) and np.array_equal(a[0], b[0]) and np.array_equal(a[0], oracle_chain(str(legacy))[0])
    assert a[0].size == 6 * (64 + 16 + 4 + 1) * 4
    # a file without a chain: one level
    one = tmp_path / "one.dds"
    one.write_bytes(cube_header(4, 1, b"DXT1") + bytes(8 * 6))
    assert g.load_dds_cube_mips(str(one))[1:] == (4, 1)
    # refusals: truncated payload, short buffer, a 2-D file, null outputs
    d, m = C.c_uint32(), C.c_uint32()
    cut = tmp_path / "cut.dds"
    cut.write_bytes(cube_header(8, 4, b"DXT1") + b"".join(faces)[:-8])
    buf = np.zeros(a[0].size, np.uint8)
    assert lib.crychic_load_dds_cube_rgba8_mips(str(cut).encode(), buf.ctypes.data, buf.nbytes, C.byref(d), C.byref(m)) == -1
    assert lib.crychic_load_dds_cube_rgba8_mips(str(legacy).encode(), buf.ctypes.data, buf.nbytes - 1, C.byref(d), C.byref(m)) == -1
    assert lib.crychic_load_dds_cube_rgba8_mips(str(legacy).encode(), None, 0, None, C.byref(m)) == -1
    assert lib.crychic_load_dds_cube_rgba8_mips(str(legacy).encode(), None, 0, C.byref(d), None) == -1
    # more levels than a 1 x 1 tail allows
    bad = tmp_path / "bad.dds"
    bad.write_bytes(cube_header(4, 5, b"DXT1") + bytes(8 * 6 * 5))
    assert lib.crychic_load_dds_cube_rgba8_mips(str(bad).encode(), None, 0, C.byref(d), C.byref(m)) == -4


def lod_of(oracle, dim, levels, r, ddx, ddy):
    L = oracle.lib
    L.or_sample_cube_lod.restype = C.c_float
    L.or_sample_cube_lod.argtypes = [C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    a, b, c = (np.asarray(v, np.float32) for v in (r, ddx, ddy))
    return float(L.or_sample_cube_lod(dim, levels, a.ctypes.data, b.ctypes.data, c.ctypes.data))


def test_cube_lod_known_answers(oracle):
    """The level of detail from hand-computable derivatives.  On the +X face sc = -r.z, tc = -r.y, u = 0.5 (sc / ma + 1): with
    r = (1, 0, 0) a derivative (0, 0, -2 k / dim) moves u by k texels of level 0."""
    dim, levels = 64, 7
    t = 2.0 / dim                                   # one level-0 texel in face coordinates [-1, 1]
    z = (0.0, 0.0, 0.0)
    assert lod_of(oracle, dim, levels, (1, 0, 0), z, z) == 0.0                                  # no change across the quad
    assert lod_of(oracle, dim, levels, (1, 0, 0), (0, 0, -t), z) == 0.0                         # one texel per pixel
    assert lod_of(oracle, dim, levels, (1, 0, 0), (0, 0, -0.5 * t), (0, 0.3 * t, 0)) == 0.0     # magnified
    assert abs(lod_of(oracle, dim, levels, (1, 0, 0), (0, 0, -4 * t), z) - 2.0) < 1e-6         # four texels per pixel
    assert abs(lod_of(oracle, dim, levels, (1, 0, 0), (0, 0, -t), (0, -8 * t, 0)) - 3.0) < 1e-6    # the larger axis decides
    assert abs(lod_of(oracle, dim, levels, (1, 0, 0), (0, 3 * t, -4 * t), z) - np.log2(5.0)) < 1e-6    # Euclidean length (3, 4) -> 5
    assert lod_of(oracle, dim, levels, (1, 0, 0), (0, 0, 10.0), z) == levels - 1.0             # clamp at the last level
    assert lod_of(oracle, dim, 1, (1, 0, 0), (0, 0, 10.0), z) == 0.0
    # every face, same magnitude: the derivative along each face's s axis
    for r, d in (((1, 0, 0), (0, 0, 1)), ((-1, 0, 0), (0, 0, 1)), ((0, 1, 0), (1, 0, 0)), ((0, -1, 0), (1, 0, 0)), ((0, 0, 1), (1, 0, 0)), ((0, 0, -1), (1, 0, 0))):
        assert abs(lod_of(oracle, dim, levels, r, tuple(4 * t * c for c in d), z) - 2.0) < 1e-6, r
    # the chain rule: scaling r leaves the face coordinate alone -- a derivative parallel to r does not move the lookup
    assert lod_of(oracle, dim, levels, (1.0, 0.25, -0.5), (0.5, 0.125, -0.25), z) == 0.0
    # ... and a change of the major axis alone does: u = -r.z / r.x, du = r.z / r.x^2 d(r.x) -> 0.5 * dim * 0.5 * 0.25 = 4 texels
    assert abs(lod_of(oracle, dim, levels, (1.0, 0.0, -0.5), (0.25, 0, 0), z) - 2.0) < 1e-6
    # a NaN derivative on the x axis: level 0 (the definition's rule); on the y axis alone it is ignored
    assert lod_of(oracle, dim, levels, (1, 0, 0), (np.nan, 0, 0), (0, 0, -4 * t)) == 0.0
    assert abs(lod_of(oracle, dim, levels, (1, 0, 0), (0, 0, -4 * t), (np.nan, 0, 0)) - 2.0) < 1e-6


def test_cube_level_lookup_known_answers(oracle):
    """or_cube_trilinear on the grey chain: integer lod -> that level's grey, lod + 0.5 -> halfway, the last level is a 1 x 1 face."""
    L = oracle.lib
    L.or_sample_cube_level.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_float, C.c_void_p]
    chain, levels = grey_chain(16)
    assert levels == 5
    r = np.asarray((0.3, -0.9, 0.2), np.float32)
    out = np.zeros(3, np.float32)
    for lod, want in ((0.0, 0.0), (1.0, 16.0), (2.5, 40.0), (3.25, 52.0), (4.0, 64.0)):
        L.or_sample_cube_level(chain.ctypes.data, 16, levels, r.ctypes.data, lod, out.ctypes.data)
        assert np.allclose(out * 255.0, want, atol=1e-3), (lod, out * 255.0)


def sky_frame(W, H):
    from crychic_renderer_amd import scene
    base = scene.Constants(W, H, shadow_dim=64)
    depth = np.full((H, W), 0xFFFFFF, np.uint32)
    z = np.zeros((H, W, 4), np.float32)
    shadow = np.full((4, 64, 64), 0xFFFFFF, np.uint32)
    return base.pass_cb, depth, z, shadow


def test_sky_uses_higher_levels_when_minified(oracle, hostsim):
    """All-sky frames over the grey chain of a 256-texel cube map: kernel bodies == oracle; a 32 x 24 frame sees several texels per
    pixel (levels around 2), a 256 x 192 frame under one (level 0 everywhere); the filter is trilinear (greys between levels)."""
    chain, levels = grey_chain(256)
    seen = {}
    for W, H in ((32, 24), (64, 48), (256, 192)):
        cb, depth, z, shadow = sky_frame(W, H)
        got = hostsim.light(cb, z, z, z, depth, None, shadow, chain, 1, 0.0, flags=1, cube_dim=256, cube_levels=levels)
        ref = oracle.deferred_light(oracle_lib.as_oracle_cb(cb, oracle_lib.OrPassConstants), z, z, z, depth, None, shadow, chain, 1, 0.0, sky=True,
                                    cube_dim=256, cube_levels=levels)
        assert np.array_equal(got, ref), (W, H)
        assert (got[..., 0] == got[..., 1]).all() and (got[..., 1] == got[..., 2]).all()
        seen[(W, H)] = got[..., 0].astype(np.float64) / 16.0             # the level each pixel looked at
    assert seen[(256, 192)].max() == 0.0
    small, mid = seen[(32, 24)], seen[(64, 48)]
    assert 1.0 < small.mean() < 4.0 and small.min() > 0.5
    assert abs((small.mean() - mid.mean()) - 1.0) < 0.2                   # twice the pixels per axis: one level lower
    assert (np.abs(small * 16.0 % 16.0) > 0).any()                        # trilinear: not every pixel sits on a level


@pytest.mark.parametrize("pcf,flags,points", [(0.0, 1, 0), (0.0, 1 | 0x100 | 0x200 | 0x400, 0), (2.5 / 256, 1, 0), (0.0, 1, 5)])
def test_lit_frame_kernel_bodies_equal_oracle(oracle, hostsim, pcf, flags, points):
    """Reflections + sky of a lit frame with a chain: kernel bodies == oracle, bit for bit (as written, with the FIX switches, with
    the intended PCF radius, with point lights); the chain changes the image against level 0 alone; a one-level chain does not."""
    import scene_util
    from crychic_renderer_amd import geometry as g, scene
    W, H = 96, 64
    planes = scene_util.cpu_scene(W, H, 256, 32)
    npl = scene_util.np_planes(planes)
    chain, levels = g.cube_mip_chain(npl["cube"])
    assert levels == 6 and chain.size == 6 * 4 * sum((32 >> k) ** 2 for k in range(6))
    cb = planes["consts"].pass_cb
    pcb = oracle_lib.as_oracle_cb(cb, oracle_lib.OrPassConstants)
    pl = scene.point_light_grid(points) if points else None
    opl = as_or_lights(pl) if pl is not None else None
    amb = np.full((H // 2, W // 2), 40000, np.uint16)
    kw = dict(cube_dim=32, cube_levels=levels)
    got = hostsim.light(cb, npl["g0"], npl["g1"], npl["g2"], npl["depth"], amb, npl["shadow"], chain, 3, pcf, flags=flags, point_lights=pl, **kw)
    ref = oracle.deferred_light(pcb, npl["g0"], npl["g1"], npl["g2"], npl["depth"], amb, npl["shadow"], chain, 3, pcf, sky=True, fixes=flags & ~1,
                                point_lights=opl, **kw)
    assert np.array_equal(got, ref)
    flat = oracle.deferred_light(pcb, npl["g0"], npl["g1"], npl["g2"], npl["depth"], amb, npl["shadow"], npl["cube"], 3, pcf, sky=True, fixes=flags & ~1,
                                 point_lights=opl)
    assert (ref != flat).any()
    one = oracle.deferred_light(pcb, npl["g0"], npl["g1"], npl["g2"], npl["depth"], amb, npl["shadow"], chain, 3, pcf, sky=True, fixes=flags & ~1,
                                point_lights=opl, cube_dim=32, cube_levels=1)
    assert np.array_equal(one, flat)


def noisy_normals(npl, seed=23):
    """The scene's G-buffer with per-pixel random normals: neighbouring reflection vectors point anywhere, the level of detail runs
    up to the 1 x 1 level."""
    rng = np.random.default_rng(seed)
    g2 = npl["g2"].copy()
    g2[..., :3] = rng.normal(size=g2[..., :3].shape).astype(np.float32)
    return g2


def test_noise_normals_reach_the_last_level(oracle, hostsim):
    """Random normals: lookups at every level including the 1 x 1 one (a face's single texel), kernel bodies == oracle; cutting
    the chain's last level off changes the image, so it was used."""
    import scene_util
    from crychic_renderer_amd import geometry as g
    W, H = 64, 48
    planes = scene_util.cpu_scene(W, H, 256, 16)
    npl = scene_util.np_planes(planes)
    rng = np.random.default_rng(5)
    cube = rng.integers(0, 256, (6, 16, 16, 4), dtype=np.uint8)           # a noisy cube map: its levels differ visibly
    chain, levels = g.cube_mip_chain(cube)
    assert levels == 5
    g2 = noisy_normals(npl)
    cb = planes["consts"].pass_cb
    pcb = oracle_lib.as_oracle_cb(cb, oracle_lib.OrPassConstants)
    got = hostsim.light(cb, npl["g0"], npl["g1"], g2, npl["depth"], None, npl["shadow"], chain, 3, 0.0, flags=1, cube_dim=16, cube_levels=levels)
    ref = oracle.deferred_light(pcb, npl["g0"], npl["g1"], g2, npl["depth"], None, npl["shadow"], chain, 3, 0.0, sky=True, cube_dim=16, cube_levels=levels)
    assert np.array_equal(got, ref)
    short = oracle.deferred_light(pcb, npl["g0"], npl["g1"], g2, npl["depth"], None, npl["shadow"], chain, 3, 0.0, sky=True, cube_dim=16, cube_levels=levels - 1)
    assert (short != ref).any()


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,points", [(96, 64, 0), (200, 90, 0), (70, 36, 6)])
def test_chain_on_device(built_lib, oracle, W, H, points):
    """crychic_deferred_light[_points] with CRYCHIC_LIGHT_CUBE_LEVELS on the device == oracle: widths that are no multiple of the
    32-pixel wavefront rows, sky + reflections, point lights (tile culling + the quad exchange in one kernel), the intended PCF
    radius, the frame in two strips; odd row0 is refused."""
    import torch
    import scene_util
    from crychic_renderer_amd import Context, geometry as g, scene
    lib, check = built_lib.lib, built_lib.check
    planes = scene_util.cpu_scene(W, H, 256, 64)
    npl = scene_util.np_planes(planes)
    chain, levels = g.cube_mip_chain(npl["cube"])
    cb = planes["consts"].pass_cb
    pcb = oracle_lib.as_oracle_cb(cb, oracle_lib.OrPassConstants)
    amb = np.random.default_rng(3).integers(20000, 65535, (H // 2, W // 2)).astype(np.uint16)
    pl = scene.point_light_grid(points) if points else None
    if W == 200:
        npl["g2"] = noisy_normals(npl)            # every level of the chain, the 1 x 1 one included, and wavefronts that are not flat
    ctx = Context(0)
    try:
        def dev(a):
            a = np.ascontiguousarray(a)
            return torch.from_numpy(a.view(np.int32) if a.dtype == np.uint32 else (a.view(np.int16) if a.dtype == np.uint16 else a)).to(ctx.device)
        d = {k: dev(npl[k]) for k in ("g0", "g1", "g2", "depth", "shadow")}
        dchain, damb = dev(chain), dev(amb)
        shadow_ptrs = (C.c_void_p * 4)(*[d["shadow"][k].data_ptr() for k in range(4)])
        dpl = torch.from_numpy(np.frombuffer(bytes(pl), np.uint8).copy()).to(ctx.device) if pl is not None else None
        P = lambda t: C.c_void_p(t.data_ptr())
        for pcf in (0.0, 2.5 / 256):
            flags = 1 | (levels << 16)
            out = torch.zeros((H, W, 4), dtype=torch.uint8, device=ctx.device)
            rad = torch.zeros((H, W, 4), dtype=torch.float32, device=ctx.device)

            def run(row0, rows):
                if pl is None:
                    return lib.crychic_deferred_light(ctx.handle, C.byref(cb), P(d["g0"]), P(d["g1"]), P(d["g2"]), P(d["depth"]), P(damb), shadow_ptrs, 256,
                                                      P(dchain), 64, P(out), P(rad), W, H, row0, rows, 3, pcf, flags, None)
                return lib.crychic_deferred_light_points(ctx.handle, C.byref(cb), P(d["g0"]), P(d["g1"]), P(d["g2"]), P(d["depth"]), P(damb), shadow_ptrs, 256,
                                                         P(dchain), 64, P(out), P(rad), W, H, row0, rows, 3, pcf, flags, P(dpl), len(pl), None)
            half = (H // 4) * 2
            check(run(0, half))
            check(run(half, H - half))
            torch.cuda.synchronize()
            ref, refrad = oracle.deferred_light(pcb, npl["g0"], npl["g1"], npl["g2"], npl["depth"], amb, npl["shadow"], chain, 3, pcf, sky=True,
                                                want_radiance=True, point_lights=as_or_lights(pl) if pl is not None else None,
                                                cube_dim=64, cube_levels=levels)
            assert np.array_equal(out.cpu().numpy(), ref), (pcf, int((out.cpu().numpy() != ref).sum()))
            assert np.array_equal(rad.cpu().numpy().view(np.uint32), refrad.view(np.uint32))
            flat = oracle.deferred_light(pcb, npl["g0"], npl["g1"], npl["g2"], npl["depth"], amb, npl["shadow"], npl["cube"], 3, pcf, sky=True,
                                         point_lights=as_or_lights(pl) if pl is not None else None)
            assert (ref != flat).any()
            assert run(1, 2) == -1 and b"whole pixel quads" in lib.crychic_last_error()
        # more levels than the face size allows
        bad = 1 | (9 << 16)
        assert lib.crychic_deferred_light(ctx.handle, C.byref(cb), P(d["g0"]), P(d["g1"]), P(d["g2"]), P(d["depth"]), P(damb), shadow_ptrs, 256,
                                          P(dchain), 64, P(out), None, W, H, 0, H, 3, 0.0, bad, None) == -1
    finally:
        ctx.close()


@pytest.mark.gpu
def test_hot_path_with_a_loaded_chain(built_lib, oracle, tmp_path):
    """A cube file with its chain -> crychic_load_dds_cube_rgba8_mips -> Crychic.set_cube_map -> Draw (the hot path: SSAO, blur, the
    lighting pass with the chain) == the oracle fed with its own decode of the same file."""
    import torch
    import scene_util
    from crychic_renderer_amd import Context, Crychic, geometry as g
    dim, nlev = 32, 6
    rng = np.random.default_rng(19)
    payload = b""
    for face in range(6):
        for lv in range(nlev):
            dd = max(1, dim >> lv)
            payload += rng.integers(0, 256, (dd, dd, 4), dtype=np.uint8).tobytes()
    p = tmp_path / "sky_chain.dds"
    p.write_bytes(cube_header(dim, nlev, None, (0xFF0000, 0xFF00, 0xFF, 0xFF000000)) + payload)
    chain, d, levels = g.load_dds_cube_mips(str(p))
    assert (d, levels) == (dim, nlev)
    L = oracle.lib
    L.or_load_dds_cube_rgba8_mips.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    ochain = np.zeros_like(chain)
    od, om = C.c_uint32(), C.c_uint32()
    assert L.or_load_dds_cube_rgba8_mips(str(p).encode(), ochain.ctypes.data, ochain.nbytes, C.byref(od), C.byref(om)) == 0
    assert np.array_equal(chain, ochain)
    W, H = 128, 96
    planes = scene_util.cpu_scene(W, H, 256, 8)
    npl = scene_util.np_planes(planes)
    ctx = Context(0)
    try:
        dev = {k: torch.from_numpy(np.ascontiguousarray(v).view(np.int32) if v.dtype == np.uint32 else np.ascontiguousarray(v)).to(ctx.device)
               for k, v in npl.items()}
        app = Crychic(ctx, W, H, dev["randvec"], dev["cube"], shadow_dim=256)
        app.load_scene({**dev, "consts": planes["consts"]})
        app.set_cube_map(torch.from_numpy(chain).to(ctx.device), dim=dim, levels=levels)
        app.blurCount, app.numDirLights, app.flags = 2, 3, 1
        app.Draw()
        torch.cuda.synchronize()
        scb = oracle_lib.as_oracle_cb(planes["consts"].ssao_cb, oracle_lib.OrSsaoConstants)
        pcb = oracle_lib.as_oracle_cb(planes["consts"].pass_cb, oracle_lib.OrPassConstants)
        amb = oracle.compute_ssao(scb, npl["normal"], npl["depth"], npl["randvec"], 2)
        ref = oracle.deferred_light(pcb, npl["g0"], npl["g1"], npl["g2"], npl["depth"], amb, npl["shadow"], ochain, 3, app.pcfSearchRadius, sky=True,
                                    cube_dim=dim, cube_levels=levels)
        assert np.array_equal(app.mBackBuffer.cpu().numpy(), ref)
        # the same frame in two strips (the quad rows stay inside a strip)
        full = app.mBackBuffer.clone()
        app.mBackBuffer.zero_()
        app.Draw(0, 40)
        app.Draw(40, H - 40)
        torch.cuda.synchronize()
        assert torch.equal(app.mBackBuffer, full)
    finally:
        ctx.close()
