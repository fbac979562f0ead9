"""Material textures: the product's DDS decoder against the oracle's, on hand-built DXT1 / DXT5 /
32-bit-mask files (known answers) and, where the reference checkout is present, on its six material textures."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

import oracle_lib

REF_TEX = "/root/reference/Textures"


def dds_header(w, h, fourcc=None, masks=None):
    pf_flags = 0x4 if fourcc else (0x41 if masks[3] else 0x40)
    pf = struct.pack("<II4sIIIII", 32, pf_flags, fourcc or b"\0\0\0\0", 0 if fourcc else 32, *(masks or (0, 0, 0, 0)))
    hdr = struct.pack("<IIIIIII", 124, 0x1007, h, w, 0, 0, 0) + b"\0" * 44 + pf + struct.pack("<IIIII", 0x1000, 0, 0, 0, 0)
    assert len(hdr) == 124
    return b"DDS " + hdr


def oracle_load(oracle, path):
    L = oracle.lib
    L.or_load_dds_rgba8.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    w, h = C.c_uint32(), C.c_uint32()
    assert L.or_load_dds_rgba8(path.encode(), None, 0, C.byref(w), C.byref(h)) == 0
    out = np.zeros((h.value, w.value, 4), np.uint8)
    assert L.or_load_dds_rgba8(path.encode(), out.ctypes.data, out.nbytes, C.byref(w), C.byref(h)) == 0
    return out


def test_known_blocks(built_lib, oracle, tmp_path):
    from crychic_renderer_amd import geometry as g
    # DXT1, one block: c0 = pure red (0xF800) > c1 = pure blue (0x001F): palette red, blue, 2/3 red + 1/3 blue, 1/3 + 2/3
    idx = 0
    for t in range(16):
        idx |= (t % 4) << (2 * t)
    p = tmp_path / "bc1.dds"
    p.write_bytes(dds_header(4, 4, b"DXT1") + struct.pack("<HHI", 0xF800, 0x001F, idx))
    img = g.load_dds(str(p))
    assert img.shape == (4, 4, 4)
    assert img[0, 0].tolist() == [255, 0, 0, 255] and img[0, 1].tolist() == [0, 0, 255, 255]
    assert img[0, 2].tolist() == [170, 0, 85, 255] and img[0, 3].tolist() == [85, 0, 170, 255]
    assert np.array_equal(img, oracle_load(oracle, str(p)))
    # DXT1 three-colour mode (c0 <= c1): index 3 is transparent black
    p.write_bytes(dds_header(4, 4, b"DXT1") + struct.pack("<HHI", 0x001F, 0xF800, idx))
    img = g.load_dds(str(p))
    assert img[0, 2].tolist() == [128, 0, 128, 255] and img[0, 3].tolist() == [0, 0, 0, 0]
    assert np.array_equal(img, oracle_load(oracle, str(p)))
    # DXT5: alpha endpoints 255 > 0 with all eight selectors in texels 0..7
    abits = 0
    for t in range(16):
        abits |= (t % 8) << (3 * t)
    p5 = tmp_path / "bc3.dds"
    p5.write_bytes(dds_header(4, 4, b"DXT5") + bytes([255, 0]) + abits.to_bytes(6, "little") + struct.pack("<HHI", 0xFFFF, 0x0000, idx))
    img = g.load_dds(str(p5))
    assert img[..., 3].reshape(-1)[:8].tolist() == [255, 0, 219, 182, 146, 109, 73, 36]
    assert img[0, 0, :3].tolist() == [255, 255, 255] and img[0, 2, :3].tolist() == [170, 170, 170]
    assert np.array_equal(img, oracle_load(oracle, str(p5)))
    # A8R8G8B8 masks: memory order B, G, R, A
    pm = tmp_path / "argb.dds"
    px = np.arange(2 * 3 * 4, dtype=np.uint8).reshape(2, 3, 4)
    pm.write_bytes(dds_header(3, 2, None, (0xFF0000, 0xFF00, 0xFF, 0xFF000000)) + px.tobytes())
    img = g.load_dds(str(pm))
    assert img.shape == (2, 3, 4) and img[0, 0].tolist() == [2, 1, 0, 3] and img[1, 2].tolist() == [22, 21, 20, 23]
    assert np.array_equal(img, oracle_load(oracle, str(pm)))
    # errors
    bad = tmp_path / "bad.dds"
    bad.write_bytes(b"nope")
    w, h = C.c_uint32(), C.c_uint32()
    assert built_lib.lib.crychic_load_dds_rgba8(str(bad).encode(), None, 0, C.byref(w), C.byref(h)) == -4
    assert built_lib.lib.crychic_load_dds_rgba8(b"/nonexistent.dds", None, 0, C.byref(w), C.byref(h)) == -1
    small = np.zeros(8, np.uint8)
    assert built_lib.lib.crychic_load_dds_rgba8(str(p5).encode(), small.ctypes.data, 8, C.byref(w), C.byref(h)) == -1


@pytest.mark.skipif(not os.path.exists(REF_TEX), reason="reference textures are not on this machine")
def test_reference_material_textures(built_lib, oracle):
    from crychic_renderer_amd import geometry as g
    tex = g.reference_textures(REF_TEX)
    shapes = [t.shape[:2] for t in tex]
    assert shapes == [(512, 512), (256, 256), (512, 512), (512, 512), (1, 1), (1, 1)]
    for name, t in zip(["bricks2.dds", "bricks2_nmap.dds", "tile.dds", "tile_nmap.dds", "white1x1.dds", "default_nmap.dds"], tex):
        assert np.array_equal(t, oracle_load(oracle, os.path.join(REF_TEX, name))), name
    assert tex[4][0, 0].tolist() == [255, 255, 255, 255]                 # white1x1
    assert tex[5][0, 0, 2] > 200                                         # default normal map points along +z
    assert tex[1][..., 2].mean() > 180 and 100 < tex[1][..., 0].mean() < 155   # a normal map: mostly +z, xy around 0.5
    assert tex[0].std() > 10                                             # bricks have contrast


@pytest.mark.skipif(not os.path.exists(REF_TEX), reason="reference textures are not on this machine")
def test_gbuffer_with_reference_textures(built_lib, oracle, hostsim):
    """The G-buffer pass sampling the real brick / tile textures: kernel bodies vs oracle, bit for bit."""
    import oracle_lib
    import scene_util
    from crychic_renderer_amd import geometry as g
    W, H = 160, 90
    cs = scene_util.cpu_scene(W, H, 128, 16)["consts"]
    tex = g.reference_textures(REF_TEX)
    items, mats = g.cascade_scene_items(), g.reference_materials()
    view = np.array(cs.pass_cb.View, np.float32); vp = np.array(cs.pass_cb.ViewProj, np.float32)
    a = oracle_lib.rasterize(oracle, 2, view, vp, items, mats.view(oracle_lib.MATERIAL_DT), tex, W, H)
    b = hostsim.rasterize(2, view, vp, items, mats, tex, W, H)
    for k in ("g0", "g1", "g2"):
        assert np.array_equal(a[k].view(np.uint32), b[k].view(np.uint32)), k
    cov = a["depth"] < 0xFFFFFF
    assert len(np.unique(a["g1"][cov][:, 0])) > 50                       # brick / tile texels modulate the albedo
    # the same with the mip chains the files carry, through the anisotropic sampler (what gsamAnisotropicWrap does in the reference)
    names = ["bricks2.dds", "bricks2_nmap.dds", "tile.dds", "tile_nmap.dds", "white1x1.dds", "default_nmap.dds"]
    mtex = [g.load_dds_mips(os.path.join(REF_TEX, n)) for n in names]
    assert [len(t) for t in mtex][:1] == [10]
    am = oracle_lib.rasterize(oracle, 2, view, vp, items, mats.view(oracle_lib.MATERIAL_DT), mtex, W, H)
    bm = hostsim.rasterize(2, view, vp, items, mats, mtex, W, H)
    for k in ("g0", "g1", "g2"):
        assert np.array_equal(am[k].view(np.uint32), bm[k].view(np.uint32)), k
    assert (am["g1"][cov][:, :3] != a["g1"][cov][:, :3]).mean() > 0.2     # minified bricks: the chain changes the albedo


def test_save_ppm_round_trip(built_lib, tmp_path):
    """Present stand-in (row f3; the reference hands the back buffer to IDXGISwapChain::Present, CRYCHIC.cpp:294-297):
    crychic_save_ppm writes binary P6 with the alpha channel dropped; reading it back returns the RGB bytes."""
    import ctypes as C
    lib = built_lib.lib
    rng = np.random.default_rng(11)
    for (w, h) in ((1, 1), (5, 3), (800, 600)):
        img = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
        path = tmp_path / ("f_%dx%d.ppm" % (w, h))
        assert lib.crychic_save_ppm(str(path).encode(), img.ctypes.data, w, h) == 0
        raw = path.read_bytes()
        header = b"P6\n%d %d\n255\n" % (w, h)
        assert raw.startswith(header) and len(raw) == len(header) + w * h * 3
        assert np.array_equal(np.frombuffer(raw[len(header):], dtype=np.uint8).reshape(h, w, 3), img[..., :3])
    assert lib.crychic_save_ppm(str(tmp_path / "no_such_dir" / "x.ppm").encode(), img.ctypes.data, 4, 4) < 0
    assert lib.crychic_save_ppm(None, img.ctypes.data, 4, 4) < 0 and lib.crychic_save_ppm(str(path).encode(), None, 4, 4) < 0
    assert lib.crychic_save_ppm(str(path).encode(), img.ctypes.data, 0, 4) < 0


def oracle_load_mips(oracle, path):
    L = oracle.lib
    L.or_load_dds_rgba8_mips.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
    w, h, n = C.c_uint32(), C.c_uint32(), C.c_uint32()
    rc = L.or_load_dds_rgba8_mips(path.encode(), None, 0, C.byref(w), C.byref(h), C.byref(n))
    if rc:
        return rc
    sizes = [(max(1, h.value >> k), max(1, w.value >> k)) for k in range(n.value)]
    flat = np.zeros(sum(a * b * 4 for a, b in sizes), np.uint8)
    rc = L.or_load_dds_rgba8_mips(path.encode(), flat.ctypes.data, flat.size, C.byref(w), C.byref(h), C.byref(n))
    if rc:
        return rc
    out, off = [], 0
    for a, b in sizes:
        out.append(flat[off:off + a * b * 4].reshape(a, b, 4)); off += a * b * 4
    return out


def mip_header(w, h, levels, fourcc=None, masks=None):
    hdr = bytearray(dds_header(w, h, fourcc, masks))
    struct.pack_into("<I", hdr, 8, 0x1007 | 0x20000)          # DDSD_MIPMAPCOUNT
    struct.pack_into("<I", hdr, 28, levels)
    return bytes(hdr)


def test_mip_chain_loader(built_lib, oracle, tmp_path):
    """crychic_load_dds_rgba8_mips: every level the file stores, back to back (DDSTextureLoader uploads the stored levels);
    product == oracle on DXT1 / DXT5 / 32-bit chains, non-square and non-multiple-of-four levels included; truncated chains and
    impossible level counts are refused."""
    from crychic_renderer_amd import geometry as g
    rng = np.random.default_rng(21)

    def payload(w, h, levels, block):
        out = b""
        for k in range(levels):
            lw, lh = max(1, w >> k), max(1, h >> k)
            n = ((lw + 3) // 4) * ((lh + 3) // 4) * block if block else lw * lh * 4
            out += rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        return out

    cases = [("d1.dds", 16, 8, 5, b"DXT1", None, 8), ("d5.dds", 32, 32, 6, b"DXT5", None, 16), ("d5_partial.dds", 32, 32, 3, b"DXT5", None, 16),
             ("rgba.dds", 12, 5, 4, None, (0x00FF0000, 0x0000FF00, 0x000000FF, 0xFF000000), 0)]
    for name, w, h, levels, fourcc, masks, block in cases:
        p = tmp_path / name
        p.write_bytes(mip_header(w, h, levels, fourcc, masks) + payload(w, h, levels, block))
        got = g.load_dds_mips(str(p))
        ref = oracle_load_mips(oracle, str(p))
        assert len(got) == levels == len(ref)
        for k in range(levels):
            assert got[k].shape == (max(1, h >> k), max(1, w >> k), 4) and np.array_equal(got[k], ref[k]), (name, k)
        assert np.array_equal(got[0], g.load_dds(str(p)))                 # level 0 is what the single-level entry point returns
    # a file without the mip flag has one level
    p = tmp_path / "single.dds"
    p.write_bytes(dds_header(8, 8, b"DXT1") + payload(8, 8, 1, 8))
    assert len(g.load_dds_mips(str(p))) == 1
    # truncated chain / more levels than a 1 x 1 tail allows
    lib = built_lib.lib
    w_, h_, n_ = C.c_uint32(), C.c_uint32(), C.c_uint32()
    p = tmp_path / "short.dds"
    p.write_bytes(mip_header(16, 16, 5, b"DXT1") + payload(16, 16, 2, 8))
    buf = np.zeros(16 * 16 * 4 * 2, np.uint8)
    assert lib.crychic_load_dds_rgba8_mips(str(p).encode(), buf.ctypes.data, buf.size, C.byref(w_), C.byref(h_), C.byref(n_)) < 0
    assert oracle_load_mips(oracle, str(p)) < 0
    p = tmp_path / "toomany.dds"
    p.write_bytes(mip_header(4, 4, 9, b"DXT1") + payload(4, 4, 3, 8))
    assert lib.crychic_load_dds_rgba8_mips(str(p).encode(), None, 0, C.byref(w_), C.byref(h_), C.byref(n_)) < 0
    assert lib.crychic_load_dds_rgba8_mips(str(p).encode(), None, 0, C.byref(w_), C.byref(h_), None) < 0


@pytest.mark.skipif(not os.path.exists(REF_TEX), reason="reference textures are not on this machine")
def test_reference_textures_mip_chains(built_lib, oracle):
    """The reference's material textures with the mip chains their files carry (bricks2.dds: DXT5 512^2, 10 levels)."""
    from crychic_renderer_amd import geometry as g
    for name, levels in (("bricks2.dds", 10), ("tile.dds", None), ("bricks2_nmap.dds", None), ("tile_nmap.dds", None), ("white1x1.dds", 1)):
        got = g.load_dds_mips(os.path.join(REF_TEX, name))
        ref = oracle_load_mips(oracle, os.path.join(REF_TEX, name))
        assert len(got) == len(ref) and (levels is None or len(got) == levels), (name, len(got))
        for a, b in zip(got, ref):
            assert np.array_equal(a, b), name
        if len(got) > 3:      # a real chain: each level is about the 2 x 2 average of the one below
            lo = got[1].astype(np.float32)
            hi = got[0].astype(np.float32).reshape(lo.shape[0], 2, lo.shape[1], 2, 4).mean(axis=(1, 3))
            assert np.abs(lo - hi).mean() < 12.0, name


def test_dds_header_with_absurd_size_is_refused(built_lib, tmp_path):
    """A header announcing 2^31 x 2^31 texels (width * height * 4 wraps to 0) must be refused before any size is computed from it
    (ADVICE r2); so is a zero-sized one."""
    import ctypes as C
    import struct
    lib = built_lib.lib
    def header(w, h):
        hdr = bytearray(128)
        hdr[0:4] = b"DDS "
        struct.pack_into("<7I", hdr, 4, 124, 0x1007, h, w, w * 4 & 0xFFFFFFFF, 0, 1)
        struct.pack_into("<8I", hdr, 76, 32, 0x41, 0, 32, 0x00FF0000, 0x0000FF00, 0x000000FF, 0xFF000000)      # 32-bit masks
        return bytes(hdr)
    for w, h in ((1 << 31, 1 << 31), (0, 4), (20000, 4)):
        path = str(tmp_path / ("bad_%d_%d.dds" % (w % 997, h % 997)))
        open(path, "wb").write(header(w, h) + b"\0" * 64)
        ww, hh = C.c_uint32(), C.c_uint32()
        buf = (C.c_uint8 * 64)()
        assert lib.crychic_load_dds_rgba8(path.encode(), buf, 64, C.byref(ww), C.byref(hh)) < 0, (w, h)
    path = str(tmp_path / "ok.dds")
    open(path, "wb").write(header(4, 4) + bytes(range(64)))
    ww, hh = C.c_uint32(), C.c_uint32()
    buf = (C.c_uint8 * 64)()
    assert lib.crychic_load_dds_rgba8(path.encode(), buf, 64, C.byref(ww), C.byref(hh)) == 0 and (ww.value, hh.value) == (4, 4)


# ---- the sky cube map (CRYCHIC.cpp:960,968: snowcube1024.dds; the file itself is not in the checkout) ---------------------------
def cube_header(dim, levels, fourcc=None, masks=None, dx10=None, all_faces=True):
    """Legacy cube header (DDSCAPS2_CUBEMAP + faces) or, with dx10 = (dxgiFormat, miscFlag), a DX10 one."""
    if dx10:
        pf = struct.pack("<II4sIIIII", 32, 0x4, b"DX10", 0, 0, 0, 0, 0)
    else:
        pf_flags = 0x4 if fourcc else (0x41 if masks[3] else 0x40)
        pf = struct.pack("<II4sIIIII", 32, pf_flags, fourcc or b"\0\0\0\0", 0 if fourcc else 32, *(masks or (0, 0, 0, 0)))
    caps2 = 0 if dx10 else (0x200 | (0xFC00 if all_faces else 0x0C00))
    hdr = struct.pack("<IIIIIII", 124, 0x1007 | (0x20000 if levels > 1 else 0), dim, dim, 0, 0, levels) + b"\0" * 44 + pf \
        + struct.pack("<IIIII", 0x1008 | (0x400000 if levels > 1 else 0), caps2, 0, 0, 0)
    assert len(hdr) == 124
    ext = struct.pack("<IIIII", dx10[0], 3, dx10[1], 1, 0) if dx10 else b""
    return b"DDS " + hdr + ext


def oracle_cube(oracle, path):
    L = oracle.lib
    L.or_load_dds_cube_rgba8.argtypes = [C.c_char_p, C.c_void_p, C.c_size_t, C.c_void_p]
    d = C.c_uint32()
    assert L.or_load_dds_cube_rgba8(path.encode(), None, 0, C.byref(d)) == 0
    out = np.zeros((6, d.value, d.value, 4), np.uint8)
    assert L.or_load_dds_cube_rgba8(path.encode(), out.ctypes.data, out.nbytes, C.byref(d)) == 0
    return out


def test_cube_map_loader(built_lib, oracle, tmp_path):
    """crychic_load_dds_cube_rgba8: six faces in D3D order, each followed in the file by its own mip tail (which the loader skips),
    for uncompressed, DXT1 and DXT5 payloads and both header generations; known answers, and == the oracle's decoder."""
    from crychic_renderer_amd import geometry as g
    rng = np.random.default_rng(5)
    # 32-bit A8R8G8B8, 4 x 4 faces with 3 levels (4, 2, 1): face k is filled with the value 10 k + level
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
    cube = g.load_dds_cube(str(p))
    assert cube.shape == (6, 4, 4, 4)
    for k in range(6):
        assert (cube[k] == np.array([200, 100 + k, 10 * k, 255 - k], np.uint8)).all(), k       # R, G, B, A of level 0
    assert np.array_equal(cube, oracle_cube(oracle, str(p)))
    # DXT1 and DXT5, 8 x 8 faces with the full chain (8, 4, 2, 1): random blocks; the same payload behind a DX10 header
    for fourcc, dxgi, blk in ((b"DXT1", 71, 8), (b"DXT5", 77, 16)):
        def level_bytes(d):
            return ((d + 3) // 4) ** 2 * blk
        faces = [rng.integers(0, 256, sum(level_bytes(max(1, 8 >> lv)) for lv in range(4)), dtype=np.uint8).tobytes() for _ in range(6)]
        legacy, dx10 = tmp_path / ("cube_%s.dds" % fourcc.decode()), tmp_path / ("cube_%s_dx10.dds" % fourcc.decode())
        legacy.write_bytes(cube_header(8, 4, fourcc) + b"".join(faces))
        dx10.write_bytes(cube_header(8, 4, dx10=(dxgi, 0x4)) + b"".join(faces))
        a, b = g.load_dds_cube(str(legacy)), g.load_dds_cube(str(dx10))
        assert a.shape == (6, 8, 8, 4) and np.array_equal(a, b)
        assert np.array_equal(a, oracle_cube(oracle, str(legacy))) and np.array_equal(b, oracle_cube(oracle, str(dx10)))
        # face k's level 0 is what the 2-D decoder makes of the same blocks
        flat = tmp_path / "flat.dds"
        for k in (0, 5):
            flat.write_bytes(dds_header(8, 8, fourcc) + faces[k][:level_bytes(8)])
            assert np.array_equal(a[k], g.load_dds(str(flat))), (fourcc, k)
    # refusals: a 2-D file, a cube with missing faces, a truncated cube, a buffer that is too small; the 2-D loaders refuse a cube
    lib = built_lib.lib
    d, w, h = C.c_uint32(), C.c_uint32(), C.c_uint32()
    flat.write_bytes(dds_header(4, 4, b"DXT1") + bytes(8))
    assert lib.crychic_load_dds_cube_rgba8(str(flat).encode(), None, 0, C.byref(d)) == -4
    part = tmp_path / "part.dds"
    part.write_bytes(cube_header(4, 1, b"DXT1", all_faces=False) + bytes(8 * 6))
    assert lib.crychic_load_dds_cube_rgba8(str(part).encode(), None, 0, C.byref(d)) == -4
    cut = tmp_path / "cut.dds"
    cut.write_bytes(cube_header(4, 1, b"DXT1") + bytes(8 * 5))
    buf = np.zeros(6 * 4 * 4 * 4, np.uint8)
    assert lib.crychic_load_dds_cube_rgba8(str(cut).encode(), buf.ctypes.data, buf.nbytes, C.byref(d)) == -1
    assert lib.crychic_load_dds_cube_rgba8(str(p).encode(), buf.ctypes.data, buf.nbytes - 1, C.byref(d)) == -1
    assert lib.crychic_load_dds_rgba8(str(p).encode(), None, 0, C.byref(w), C.byref(h)) == -4


def test_cube_face_order_known_answer(built_lib, oracle, hostsim, tmp_path):
    """Face order and orientation of the sky cube map, end to end (the reference holds no decoded fixture and snowcube1024.dds is not
    in the checkout, so this is a derived known answer, not a pinned one): a hand-built six-face A8R8G8B8 DDS with one colour per
    face, plus a black marker texel next to each face's centre on its (s, t) = (0, 0) side, goes through crychic_load_dds_cube_rgba8 and the sky lookup
    (sky.hlsl:21-47) of an all-sky frame whose camera looks along +X, -X, +Y, -Y, +Z, -Z: the centre pixel shows D3D's face 0 .. 5
    (DDS order +X, -X, +Y, -Y, +Z, -Z), kernel bodies == oracle, and the marker corner sits where D3D's cube-face axes put it."""
    import copy
    from crychic_renderer_amd import geometry as g, scene
    from crychic_renderer_amd._lib import Camera, PassConstants, lib
    dim = 8
    colours = [(250, 10, 10), (10, 250, 10), (10, 10, 250), (250, 250, 10), (250, 10, 250), (10, 250, 250)]      # R, G, B per face
    payload = b""
    for k, (r, gg, b) in enumerate(colours):
        px = np.zeros((dim, dim, 4), np.uint8)
        px[..., 0], px[..., 1], px[..., 2], px[..., 3] = b, gg, r, 255          # memory order B, G, R, A
        px[dim // 2 - 1, dim // 2 - 1] = (0, 0, 0, 255)                         # the texel up and to the left of the face centre (towards s = t = 0)
        payload += px.tobytes()
    p = tmp_path / "faces.dds"
    p.write_bytes(cube_header(dim, 1, None, (0xFF0000, 0xFF00, 0xFF, 0xFF000000)) + payload)
    cube = g.load_dds_cube(str(p))
    assert np.array_equal(cube, oracle_cube(oracle, str(p)))
    W = H = 32
    base = scene.Constants(W, H, shadow_dim=64)
    views = [((1, 0, 0), (0, 1, 0)), ((-1, 0, 0), (0, 1, 0)), ((0, 1, 0), (0, 0, -1)), ((0, -1, 0), (0, 0, 1)), ((0, 0, 1), (0, 1, 0)), ((0, 0, -1), (0, 1, 0))]
    depth = np.full((H, W), 0xFFFFFF, np.uint32)
    z = np.zeros((H, W, 4), np.float32)
    shadow = np.full((4, 64, 64), 0xFFFFFF, np.uint32)
    dirs = np.asarray(scene.BASE_LIGHT_DIRS, dtype=np.float32)
    for face, (look, up) in enumerate(views):
        cam = copy.deepcopy(base.cam)
        cam.pos[:] = (0.0, 0.0, 0.0); cam.look[:] = look; cam.up[:] = up
        cb = PassConstants()
        assert lib.crychic_update_main_pass_cb(C.byref(cam), W, H, base.shadow_transform.ctypes.data, dirs.ctypes.data, C.byref(cb)) == 0
        got = hostsim.light(cb, z, z, z, depth, None, shadow, cube, 1, 0.0, flags=1)
        ref = oracle.deferred_light(oracle_lib.as_oracle_cb(cb, oracle_lib.OrPassConstants), z, z, z, depth, None, shadow, cube, 1, 0.0, sky=True)
        assert np.array_equal(got, ref), face
        centre = got[H // 2 + 4, W // 2 + 4, :3]          # just down-right of the centre: clear of the marker's filter footprint
        assert tuple(int(v) for v in centre) == colours[face], (face, centre)
        # D3D cube-face axes: (s, t) = (0, 0) is the face's top-left as seen from inside with the face's own "up":
        # for the +-X and +-Z faces with up = +Y that is the image's top-left corner; +Y is seen with up = -Z, -Y with up = +Z
        dark = (got[..., :3].astype(int).sum(-1) < 60)
        ys, xs = np.nonzero(dark)
        assert len(ys) > 0 and ys.mean() < H / 2 and xs.mean() < W / 2, (face, ys.mean() if len(ys) else None, xs.mean() if len(xs) else None)


@pytest.mark.gpu
def test_loaded_cube_map_lights_a_frame(built_lib, oracle, tmp_path):
    """A cube map that went through the DDS loader is the plane the lighting pass samples: sky + reflections of a small frame ==
    the oracle fed with the oracle's decode of the same file."""
    import torch
    import scene_util
    from crychic_renderer_amd import Context, Crychic, geometry as g
    rng = np.random.default_rng(11)
    dim = 16
    faces = [rng.integers(0, 256, (dim // 4) ** 2 * 8, dtype=np.uint8).tobytes() for _ in range(6)]
    p = tmp_path / "sky.dds"
    p.write_bytes(cube_header(dim, 1, b"DXT1") + b"".join(faces))
    cube = g.load_dds_cube(str(p))
    assert np.array_equal(cube, oracle_cube(oracle, str(p)))
    W, H = 128, 96
    planes = scene_util.cpu_scene(W, H, 256, 8)
    npl = scene_util.np_planes(planes)
    npl["cube"] = cube
    ctx = Context(0)
    try:
        dev = {k: torch.from_numpy(np.ascontiguousarray(v).view(np.int32) if v.dtype == np.uint32 else np.ascontiguousarray(v)).to(ctx.device)
               for k, v in npl.items()}
        app = Crychic(ctx, W, H, dev["randvec"], dev["cube"], shadow_dim=256)
        app.load_scene({**dev, "consts": planes["consts"]})
        app.blurCount, app.numDirLights, app.flags = 2, 3, 1           # CRYCHIC_LIGHT_SKY: uncovered pixels show the cube map
        app.Draw()
        torch.cuda.synchronize()
        scb = oracle_lib.as_oracle_cb(planes["consts"].ssao_cb, oracle_lib.OrSsaoConstants)
        pcb = oracle_lib.as_oracle_cb(planes["consts"].pass_cb, oracle_lib.OrPassConstants)
        amb = oracle.compute_ssao(scb, npl["normal"], npl["depth"], npl["randvec"], 2)
        ref = oracle.deferred_light(pcb, npl["g0"], npl["g1"], npl["g2"], npl["depth"], amb, npl["shadow"], cube, 3, app.pcfSearchRadius, sky=True)
        assert np.array_equal(app.mBackBuffer.cpu().numpy(), ref)
        assert (npl["depth"] & 0xFFFFFF == 0xFFFFFF).any()             # the frame does show sky
    finally:
        ctx.close()
