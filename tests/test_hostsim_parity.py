"""CPU tier of the parity ladder: the per-pixel kernel bodies the gfx950 kernels inline (csrc/*_core.hpp), compiled
for the host, must reproduce the oracle bit for bit on seeded scenes -- ragged sizes included.  (The GPU tier,
tests/test_gpu_parity.py, repeats this through the C ABI on the real kernels.)"""
import numpy as np
import pytest

import oracle_lib
import scene_util

SIZES = [(64, 64), (130, 34), (256, 256)]


def setup(W, H, built_lib):
    pl = scene_util.cpu_scene(W, H, 512, 64)
    p = scene_util.np_planes(pl)
    c = pl["consts"]
    scb = oracle_lib.as_oracle_cb(c.ssao_cb, oracle_lib.OrSsaoConstants)
    pcb = oracle_lib.as_oracle_cb(c.pass_cb, oracle_lib.OrPassConstants)
    eb = int(built_lib.lib.crychic_edge_plane_bytes(W, H))
    return p, c, scb, pcb, eb


@pytest.mark.parametrize("W,H", SIZES)
def test_ssao_and_blur_chain(built_lib, oracle, hostsim, W, H):
    p, c, scb, pcb, eb = setup(W, H, built_lib)
    ref = oracle.ssao(scb, p["normal"], p["depth"], p["randvec"])
    got, edge = hostsim.ssao(c.ssao_cb, p["normal"], p["depth"], p["randvec"], eb)
    assert np.array_equal(got, ref)
    assert ref.min() < 65535
    # the taps on the raw D24 plane (no workspace) and on the decoded depth-pairs plane (default) are the same bits
    raw, edge_raw = hostsim.ssao(c.ssao_cb, p["normal"], p["depth"], p["randvec"], eb, pairs=False)
    assert np.array_equal(raw, ref)
    n12 = (W // 2) * (H // 2) * 12
    assert np.array_equal(edge[:n12], edge_raw[:n12])      # centre normals / depths identical either way
    # blur on the SSAO output and on noise (noise exercises every accept/reject combination)
    rng = np.random.default_rng(W * 7 + H)
    for start in (ref, rng.integers(0, 65536, size=ref.shape, dtype=np.uint16)):
        cur = start
        for horz in (True, False, True, False):
            r = oracle.blur(scb, p["normal"], p["depth"], cur, horz)
            g = hostsim.blur(c.ssao_cb, edge, cur, W, H, horz)
            assert np.array_equal(g, r), (horz, int((g != r).sum()))
            cur = r


def oracle_chain(oracle, scb, normal, depth, start, blur_count):
    cur = start
    for _ in range(blur_count):
        cur = oracle.blur(scb, normal, depth, cur, True)
        cur = oracle.blur(scb, normal, depth, cur, False)
    return cur


@pytest.mark.parametrize("W,H", [(64, 64), (130, 34), (256, 256), (300, 100)])
@pytest.mark.parametrize("blur_count", [1, 2, 3, 4, 5, 8])
def test_blur_chain_matches_oracle(built_lib, oracle, hostsim, W, H, blur_count):
    """The two-launch blur chain (blur_tiles.hpp: iteration 0 as one H + V launch, the others fused and replayed, tile by tile
    with recomputed aprons) equals the oracle's 2 * blurCount separate sweeps -- on the SSAO output and on noise (noise exercises
    every accept / reject combination, and map edges on every side of ragged tile grids)."""
    p, c, scb, pcb, eb = setup(W, H, built_lib)
    ref, edge = hostsim.ssao(c.ssao_cb, p["normal"], p["depth"], p["randvec"], eb)
    rng = np.random.default_rng(W * 7 + H + blur_count)
    for start in (ref, rng.integers(0, 65536, size=ref.shape, dtype=np.uint16)):
        want = oracle_chain(oracle, scb, p["normal"], p["depth"], start, blur_count)
        got = hostsim.blur_chain(c.ssao_cb, edge, start, W, H, blur_count)
        assert np.array_equal(got, want), (blur_count, int((got != want).sum()))


@pytest.mark.parametrize("blur_count", [1, 3, 4, 7])
def test_compute_ssao_row_strips(built_lib, oracle, hostsim, blur_count):
    """A strip of the final map (what one rank of N computes): the halo rows every stage recomputes, the tile grid anchored at
    absolute rows, tiles cut by the strip's edges."""
    W, H = 130, 200
    p, c, scb, pcb, eb = setup(W, H, built_lib)
    want = oracle.compute_ssao(scb, p["normal"], p["depth"], p["randvec"], blur_count)
    for row0, rows in ((0, 100), (37, 21), (50, 50), (83, 17)):
        got, _ = hostsim.compute_ssao(c.ssao_cb, p["normal"], p["depth"], p["randvec"], eb, blur_count, row0, rows)
        assert np.array_equal(got[row0:row0 + rows], want[row0:row0 + rows]), (row0, rows)


def test_row_ranges_match_full(built_lib, oracle, hostsim):
    W, H = 130, 34
    p, c, scb, pcb, eb = setup(W, H, built_lib)
    full = oracle.ssao(scb, p["normal"], p["depth"], p["randvec"])
    part = oracle.ssao(scb, p["normal"], p["depth"], p["randvec"], 5, 7)
    assert np.array_equal(part[5:12], full[5:12]) and (part[:5] == 0).all() and (part[12:] == 0).all()
    got, _ = hostsim.ssao(c.ssao_cb, p["normal"], p["depth"], p["randvec"], eb, 5, 7)
    assert np.array_equal(got[5:12], full[5:12])


@pytest.mark.parametrize("W,H", SIZES)
@pytest.mark.parametrize("ndl,literal,sky,ssao_on", [(1, 1, 0, 1), (3, 0, 1, 1), (3, 1, 0, 0), (2, 0, 0, 1)])
def test_deferred_light(built_lib, oracle, hostsim, W, H, ndl, literal, sky, ssao_on):
    p, c, scb, pcb, eb = setup(W, H, built_lib)
    amb = oracle.compute_ssao(scb, p["normal"], p["depth"], p["randvec"], 1) if ssao_on else None
    radius = built_lib.lib.crychic_pcf_search_radius(p["shadow"].shape[1], literal)
    ref, rref = oracle.deferred_light(pcb, p["g0"], p["g1"], p["g2"], p["depth"], amb, p["shadow"], p["cube"], ndl, radius,
                                      sky=bool(sky), want_radiance=True)
    got, rgot = hostsim.light(c.pass_cb, p["g0"], p["g1"], p["g2"], p["depth"], amb, p["shadow"], p["cube"], ndl, radius,
                              flags=sky, want_radiance=True)
    assert np.array_equal(got, ref)
    assert np.array_equal(rgot.view(np.uint32), rref.view(np.uint32))
    assert len(np.unique(ref.reshape(-1, 4), axis=0)) > 50   # a real image, not a constant


def test_degenerate_inputs_agree(built_lib, oracle, hostsim):
    """NaN / inf / zero-length inputs must take the same path on both sides (sampler and detmath special cases)."""
    W, H = 64, 64
    p, c, scb, pcb, eb = setup(W, H, built_lib)
    g0, g1, g2 = p["g0"].copy(), p["g1"].copy(), p["g2"].copy()
    depth = p["depth"].copy(); depth[:] = 1000
    g2[0:8] = 0.0                      # zero normal -> normalize gives NaN
    g0[8:16, :, :3] = 1e30             # far away -> huge shadow coordinates
    g0[16:24, :, 0] = np.inf
    g1[24:32, :, 3] = 0.0              # roughness 0 -> D = 0 * inf
    g0[32:40, :, :3] = np.array(list(c.cam.pos), dtype=np.float32)  # at the eye: view = 0/0
    g1[40:48, :, :3] = -1.0            # negative albedo
    radius = built_lib.lib.crychic_pcf_search_radius(p["shadow"].shape[1], 0)
    amb = np.full((H // 2, W // 2), 40000, dtype=np.uint16)
    ref, rref = oracle.deferred_light(pcb, g0, g1, g2, depth, amb, p["shadow"], p["cube"], 3, radius, want_radiance=True)
    got, rgot = hostsim.light(c.pass_cb, g0, g1, g2, depth, amb, p["shadow"], p["cube"], 3, radius, want_radiance=True)
    assert np.array_equal(got, ref)
    assert np.array_equal(rgot.view(np.uint32), rref.view(np.uint32))
    # SSAO with a degenerate normal map (zero / NaN normals) and depth 0 (viewZ = near)
    normal = p["normal"].copy(); normal[0:16] = 0; normal[16:20, :, 0] = np.nan
    d2 = p["depth"].copy(); d2[20:30] = 0
    r = oracle.ssao(scb, normal, d2, p["randvec"])
    g, _ = hostsim.ssao(c.ssao_cb, normal, d2, p["randvec"], eb)
    assert np.array_equal(g, r)


@pytest.mark.parametrize("fixes", [0x100, 0x200, 0x400, 0x700])
def test_quirk_fix_switches(built_lib, oracle, hostsim, fixes):
    """CRYCHIC_FIX_Q1 / Q3 / Q4 (crychic_hip.h): the evidently intended forms of the cascade-blend test, the specular
    denominator and the Fresnel factor.  Kernel bodies == oracle with the same switches, and every switch changes the picture."""
    W, H = 96, 64
    p, c, scb, pcb, eb = setup(W, H, built_lib)
    ao = oracle.compute_ssao(scb, p["normal"], p["depth"], p["randvec"], 1)
    base = oracle.deferred_light(pcb, p["g0"], p["g1"], p["g2"], p["depth"], ao, p["shadow"], p["cube"], 3, 0.0)
    for radius in (0.0, 2.5 / 512):
        ref, rref = oracle.deferred_light(pcb, p["g0"], p["g1"], p["g2"], p["depth"], ao, p["shadow"], p["cube"], 3, radius, fixes=fixes, want_radiance=True)
        got, rgot = hostsim.light(c.pass_cb, p["g0"], p["g1"], p["g2"], p["depth"], ao, p["shadow"], p["cube"], 3, radius, flags=fixes, want_radiance=True)
        assert np.array_equal(got, ref) and np.array_equal(rgot.view(np.uint32), rref.view(np.uint32))
        if radius == 0.0 and fixes != 0x100:        # Q1 only matters where the two blended cascades disagree
            assert (ref != base).mean() > 0.01


@pytest.mark.parametrize("seed", range(20))
def test_sky_shortcut_is_exact(built_lib, oracle, hostsim, seed):
    """The SSAO sky shortcut (ssao_core.hpp): a wavefront of clear-depth pixels whose tap neighbourhood holds no geometry writes
    65535 without running its taps.  Probe its reach bound: a sky field with small patches of geometry placed so close to the
    far plane that a sky pixel's tap landing on them DOES occlude (view depth within OcclusionFadeEnd of the far distance, and
    bilinear mixes of patch and sky texels), at random distances from everything else.  Kernel bodies with the shortcut ==
    oracle without one, the shortcut fires on most wavefronts, and some sky pixel is in fact occluded (so the probe bites)."""
    import fuzz_util
    W, H, c, scb, depth, normal, randvec = fuzz_util.sky_probe_case(seed)
    eb = int(built_lib.lib.crychic_edge_plane_bytes(W, H))
    ref = oracle.ssao(scb, normal, depth, randvec)
    got, _ = hostsim.ssao(c.ssao_cb, normal, depth, randvec, eb)
    skipped = int(hostsim.lib.hs_last_sky_waves())
    assert np.array_equal(got, ref), int((got != ref).sum())
    waves = (H // 2) * ((W // 2 + 63) // 64)
    assert skipped > 0.3 * waves, (skipped, waves)
    sky_half = (depth.reshape(H // 2, 2, W // 2, 2) == 0xFFFFFF).all(axis=(1, 3))
    if seed % 4 != 3:
        assert (ref[sky_half] < 65535).any(), "no sky pixel is occluded: the probe does not test the reach bound"
    # a SurfaceEpsilon too small for the argument switches the shortcut off (and the result is still the oracle's)
    c.ssao_cb.SurfaceEpsilon = 1.0e-6
    scb = oracle_lib.as_oracle_cb(c.ssao_cb, oracle_lib.OrSsaoConstants)
    got, _ = hostsim.ssao(c.ssao_cb, normal, depth, randvec, eb)
    assert int(hostsim.lib.hs_last_sky_waves()) == 0
    assert np.array_equal(got, oracle.ssao(scb, normal, depth, randvec))


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 4, 5])
def test_tap_culling_is_exact_and_bites(built_lib, oracle, hostsim, seed):
    """Tap culling (ssao_core.hpp): a tap whose footprint lies in blocks of the nearest-depth map that hold nothing in front of
    the pixel adds exactly +0 and is skipped.  The reference scene (open ground: most taps), then the same frame with what the
    bound must not miss: single near texels (spikes a fraction of a block wide), texels just inside / outside SurfaceEpsilon of
    their surroundings, holes to the far plane, depth 0; with three SurfaceEpsilon values.  Kernel bodies with culling == oracle
    without it, with and without the depth-pairs plane, and the cull fires on a large share of the taps."""
    import fuzz_util
    W, H, c, depth, normal, randvec = fuzz_util.cull_probe_case(seed)
    p = {"randvec": randvec}
    eb = int(built_lib.lib.crychic_edge_plane_bytes(W, H))
    scb = oracle_lib.as_oracle_cb(c.ssao_cb, oracle_lib.OrSsaoConstants)
    ref = oracle.ssao(scb, normal, depth, p["randvec"])
    got, _ = hostsim.ssao(c.ssao_cb, normal, depth, p["randvec"], eb)
    culled = int(hostsim.lib.hs_last_culled_taps())
    assert np.array_equal(got, ref), int((got != ref).sum())
    covered = int(((depth.reshape(H // 2, 2, W // 2, 2) & 0xFFFFFF) != 0xFFFFFF).any(axis=(1, 3)).sum())
    assert culled > (0.2 if c.ssao_cb.SurfaceEpsilon >= 0.01 else 0.005) * 14 * covered, (culled, covered)
    assert (ref < 65535).sum() > 50                                   # and the frame does have occlusion to get wrong
    nocull, _ = hostsim.ssao(c.ssao_cb, normal, depth, p["randvec"], eb, cull=False)
    assert int(hostsim.lib.hs_last_culled_taps()) == 0 and np.array_equal(nocull, ref)
    # constants the monotonicity argument does not cover switch the culling off (the result is still the oracle's)
    keep = (c.ssao_cb.Proj[4 * 2 + 3], c.ssao_cb.SurfaceEpsilon)
    try:
        c.ssao_cb.Proj[4 * 2 + 3] = -c.ssao_cb.Proj[4 * 2 + 3]
        scb = oracle_lib.as_oracle_cb(c.ssao_cb, oracle_lib.OrSsaoConstants)
        got, _ = hostsim.ssao(c.ssao_cb, normal, depth, p["randvec"], eb)
        assert int(hostsim.lib.hs_last_culled_taps()) == 0
        assert np.array_equal(got, oracle.ssao(scb, normal, depth, p["randvec"]))
    finally:
        c.ssao_cb.Proj[4 * 2 + 3], c.ssao_cb.SurfaceEpsilon = keep


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_clear_cells_are_never_read(built_lib, oracle, hostsim, seed):
    """Clear cells (ssao_core.hpp): the depth pass marks a cell of the nearest-depth map whose texels all hold the clear depth as
    +inf and leaves the pairs entries that only such cells can reach unwritten.  The host simulation poisons ALL of those entries.
    A frame of sky next to geometry that hugs the near plane, with an occlusion radius that puts taps at and behind the camera
    (q.z < 1e-3: never culled, landing anywhere -- mostly beyond the plane's edge, which is clear by definition), and non-finite
    normals (pixels that may not cull at all): kernel body == oracle, and such taps did land in clear cells."""
    import fuzz_util
    W, H, c, scb, depth, normal, randvec = fuzz_util.clear_cell_probe_case(seed)
    eb = int(built_lib.lib.crychic_edge_plane_bytes(W, H))
    want = oracle.ssao(scb, normal, depth, randvec)
    got, _ = hostsim.ssao(c.ssao_cb, normal, depth, randvec, eb)
    assert np.array_equal(got, want), int((got != want).sum())
    assert int(hostsim.lib.hs_last_clear_cell_taps()) > 100           # the rule was needed: taps evaluated on a clear cell's 1.0s
    assert int(hostsim.lib.hs_last_culled_taps()) > 1000
    assert (want < 65535).sum() > 50
    # strips: the row-limited pass leaves the same entries unwritten
    for row0, rows in ((0, H // 8), (H // 4, H // 8)):
        part, _ = hostsim.ssao(c.ssao_cb, normal, depth, randvec, eb, row0=row0, rows=rows)
        assert np.array_equal(part[row0:row0 + rows], want[row0:row0 + rows])


@pytest.mark.parametrize("seed", [0, 1, 5, 12, 14])
@pytest.mark.parametrize("blur_count", [1, 2, 4, 5])
def test_unoccluded_tile_exit_is_exact(built_lib, oracle, hostsim, seed, blur_count):
    """The first blur launch settles tiles whose whole neighbourhood (5 pixels per iteration) came out of the SSAO pass as 65535
    (ssao_core.hpp "unoccluded tiles"): it writes 65535, a centre-only decision and the tile's flag, and the fused replay launch
    skips flagged tiles altogether.  The chain must still equal the oracle's blurCount iterations, here on the sky-probe frames
    (large unoccluded areas next to occluded patches); a margin smaller than the reach of the sweeps does break it."""
    import fuzz_util
    W, H, c, scb, depth, normal, randvec = fuzz_util.sky_probe_case(seed)
    eb = int(built_lib.lib.crychic_edge_plane_bytes(W, H))
    want = oracle.compute_ssao(scb, normal, depth, randvec, blur_count)
    got, _ = hostsim.compute_ssao(c.ssao_cb, normal, depth, randvec, eb, blur_count)
    settled = int(hostsim.lib.hs_last_settled_tiles())
    assert np.array_equal(got, want), int((got != want).sum())
    if blur_count > 1:
        assert settled > 10
    noexit, _ = hostsim.compute_ssao(c.ssao_cb, normal, depth, randvec, eb, blur_count, use_exit=False)
    assert np.array_equal(noexit, want)


def test_unoccluded_tile_exit_margin_bites(built_lib, oracle, hostsim):
    """The exit's margin is not slack: with less than 5 pixels per iteration some settled tile is wrong on some probe frame."""
    import fuzz_util
    wrong = 0
    for seed in (0, 1, 5, 12, 14):
        W, H, c, scb, depth, normal, randvec = fuzz_util.sky_probe_case(seed)
        eb = int(built_lib.lib.crychic_edge_plane_bytes(W, H))
        want = oracle.compute_ssao(scb, normal, depth, randvec, 4)
        got, _ = hostsim.compute_ssao(c.ssao_cb, normal, depth, randvec, eb, 4, ones_margin=4)
        wrong += int((got != want).sum())
    assert wrong > 0


@pytest.mark.parametrize("blur_count", [2, 4])
def test_recycled_workspace_cannot_settle_a_tile(built_lib, oracle, hostsim, blur_count):
    """The edge workspace is caller-owned and nothing clears it.  Render frame A, then a DIFFERENT frame B over the same,
    uncleared workspace -- with a fresh stamp, as api.cpp draws one per frame, and, the worst case, with the very stamp frame A
    ran with (a recycled allocation that served another context): every word of the unoccluded-wavefront map and every tile flag
    a frame looks at was written by that frame, so frame B equals the oracle either way."""
    import fuzz_util
    W, H, c, scb, depth_a, normal, randvec = fuzz_util.sky_probe_case(0)
    depth_b, normal_b = np.ascontiguousarray(depth_a[::-1, ::-1]), np.ascontiguousarray(normal[::-1, ::-1])     # the patches elsewhere
    assert not np.array_equal(depth_a, depth_b)
    eb = int(built_lib.lib.crychic_edge_plane_bytes(W, H))
    want_b = oracle.compute_ssao(scb, normal_b, depth_b, randvec, blur_count)
    for stamp_b in (8, 7):
        _, edge = hostsim.compute_ssao(c.ssao_cb, normal, depth_a, randvec, eb, blur_count, stamp=7)
        assert int(hostsim.lib.hs_last_settled_tiles()) > 10
        got, _ = hostsim.compute_ssao(c.ssao_cb, normal_b, depth_b, randvec, eb, blur_count, edge=edge, stamp=stamp_b)
        assert np.array_equal(got, want_b), (stamp_b, int((got != want_b).sum()))
    # and a workspace pre-filled with the upcoming stamp in every word of both maps
    edge = np.zeros((eb,), dtype=np.uint8)
    edge.view(np.uint32)[:] = 9
    got, _ = hostsim.compute_ssao(c.ssao_cb, normal_b, depth_b, randvec, eb, blur_count, edge=edge, stamp=9)
    assert np.array_equal(got, want_b)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_dark_light_skip_is_exact(built_lib, oracle, hostsim, seed):
    """A directional light with Strength (0, 0, 0) -- the reference's third (CRYCHIC.cpp:863-864) -- is skipped where the pixel's
    inputs are bounded (light_core.hpp "dark lights") and evaluated everywhere else.  G-buffer values on both sides of every bound
    of the guard (roughness 0 / 0.03 / 10 / huge, albedo and metalness around 16, non-finite positions and normals), direction
    lengths at the ends of the accepted range and outside it, a NaN strength: kernel bodies == oracle (RGBA8 and radiance bits)."""
    import copy
    W, H = 96, 64
    p, c, scb, pcb, eb = setup(W, H, built_lib)
    rng = np.random.default_rng(77 + seed)
    g0, g1, g2 = p["g0"].copy(), p["g1"].copy(), p["g2"].copy()
    depth = p["depth"].copy(); depth[:] = 1000                       # every pixel covered
    n = W * H
    pick = lambda vals: rng.choice(np.asarray(vals, dtype=np.float32), size=(H, W))
    g1[..., 3] = pick([0.0, 0.02, 0.03, 0.031, 0.3, 0.8, 1.0, 9.99, 10.0, 10.5, 1e6, np.inf, np.nan, -0.5])
    for ch in range(3):
        g1[..., ch] = pick([0.0, 0.5, 0.9, 1.0, 15.9, 16.0, 16.5, -16.0, -17.0, 1e5, np.nan])
    g0[..., 3] = pick([0.0, 0.5, 1.0, 16.0, 16.01, -16.0, 1e9, np.nan])
    bad = rng.random((H, W)) < 0.1
    g0[bad, 0] = rng.choice(np.asarray([np.inf, -np.inf, np.nan, 2e30, 1e30], dtype=np.float32), size=int(bad.sum()))
    badn = rng.random((H, W)) < 0.1
    g2[badn, 1] = rng.choice(np.asarray([np.inf, np.nan, 0.0, 3.3e38], dtype=np.float32), size=int(badn.sum()))
    amb = rng.integers(0, 65536, size=(H // 2, W // 2), dtype=np.uint16)
    cb = copy.deepcopy(c.pass_cb)
    dirs = [(0.0, -0.707, -0.707), (0.0, -0.5001, 0.0), (0.0, -1.0004, 0.0), (0.0, -0.49, 0.0), (0.0, -1.01, 0.0), (np.nan, -1.0, 0.0)]
    for d in dirs:
        for strength in ((0.0, 0.0, 0.0), (-0.0, 0.0, -0.0), (0.0, np.nan, 0.0), (0.0, 1e-30, 0.0)):
            cb.Lights[2].Direction[:] = d
            cb.Lights[2].Strength[:] = strength
            ocb = oracle_lib.as_oracle_cb(cb, oracle_lib.OrPassConstants)
            ref, rref = oracle.deferred_light(ocb, g0, g1, g2, depth, amb, p["shadow"], p["cube"], 3, 0.0, want_radiance=True)
            got, rgot = hostsim.light(cb, g0, g1, g2, depth, amb, p["shadow"], p["cube"], 3, 0.0, want_radiance=True)
            assert np.array_equal(got, ref), (d, strength, int((got != ref).sum()))
            assert np.array_equal(rgot.view(np.uint32), rref.view(np.uint32)), (d, strength)
    # the guard bounds the G-buffer, not the eye: a huge, an infinite and a NaN EyePosW (the view vector's dot products all pass
    # through maxnn(), which is what keeps the skip and the short reciprocals exact there)
    cb.Lights[2].Direction[:] = dirs[0]
    cb.Lights[2].Strength[:] = (0.0, 0.0, 0.0)
    for eye in ((3.0e30, 2.0, -1.0e25), (np.inf, 2.0, -15.0), (0.0, np.nan, -15.0)):
        cb.EyePosW[:] = eye
        ocb = oracle_lib.as_oracle_cb(cb, oracle_lib.OrPassConstants)
        ref, rref = oracle.deferred_light(ocb, g0, g1, g2, depth, amb, p["shadow"], p["cube"], 3, 0.0, want_radiance=True)
        got, rgot = hostsim.light(cb, g0, g1, g2, depth, amb, p["shadow"], p["cube"], 3, 0.0, want_radiance=True)
        assert np.array_equal(got, ref), (eye, int((got != ref).sum()))
        assert np.array_equal(rgot.view(np.uint32), rref.view(np.uint32)), eye


@pytest.mark.parametrize("margin", [0, 8, 40])
def test_row_limited_depth_pass_is_exact(built_lib, oracle, hostsim, margin):
    """A strip's depth pass visits the strip's rows and a margin only (ssao_core.hpp DepthPairsRows): everything else in the pairs
    plane and the coarse maps is poison here.  Taps that leave the visited rows gather from the raw plane and are never culled,
    sky wavefronts whose reach leaves them take no shortcut: the strip equals the oracle for any margin -- on the reference scene
    (near ground: wide tap discs) and on the sky-probe frames (patches just outside the visited rows)."""
    import fuzz_util
    W, H = 256, 512
    p, c, scb, pcb, eb = setup(W, H, built_lib)
    want = oracle.ssao(scb, p["normal"], p["depth"], p["randvec"])
    for row0, rows in ((0, 40), (100, 37), (200, 56), (180, 76)):
        got, _ = hostsim.ssao(c.ssao_cb, p["normal"], p["depth"], p["randvec"], eb, row0, rows, margin=margin)
        assert int(hostsim.lib.hs_last_unprepared_rows()) > 0          # the pass really was limited
        assert np.array_equal(got[row0:row0 + rows], want[row0:row0 + rows]), (row0, rows, int((got[row0:row0 + rows] != want[row0:row0 + rows]).sum()))
    for seed in (0, 12):
        W, H, c, scb, depth, normal, randvec = fuzz_util.sky_probe_case(seed)
        eb = int(built_lib.lib.crychic_edge_plane_bytes(W, H))
        want = oracle.compute_ssao(scb, normal, depth, randvec, 2)
        for row0, rows in ((0, H // 8), (H // 6, H // 7), (H // 2 - H // 9, H // 9)):
            got, _ = hostsim.compute_ssao(c.ssao_cb, normal, depth, randvec, eb, 2, row0, rows, margin=margin)
            assert np.array_equal(got[row0:row0 + rows], want[row0:row0 + rows]), (seed, row0, rows)
