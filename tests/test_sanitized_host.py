"""The host-side C++ of the product (csrc/host_constants.cpp, host_geometry.cpp, host_textures.cpp: constant builders,
culling, mesh / DDS loaders) compiled with AddressSanitizer + UndefinedBehaviourSanitizer and driven through the same entry
points with the same arguments as the shipped library; results must match the shipped build byte for byte.  Runs only under
tools/sanitize.sh (CRYCHIC_SANITIZE=1 with the sanitizer runtime preloaded); any sanitizer report aborts the process."""
import ctypes as C
import os
import struct

import numpy as np
import pytest

import hostsim_lib

pytestmark = pytest.mark.skipif(not hostsim_lib.SANITIZE, reason="sanitizer tier: run tools/sanitize.sh")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "crychic_renderer_amd", "csrc")


@pytest.fixture(scope="module")
def san(built_lib):
    path = hostsim_lib.build_sanitized("libcrychic_host.so", [os.path.join(CSRC, f) for f in ("host_constants.cpp", "host_geometry.cpp", "host_textures.cpp")])
    lib = C.CDLL(path)
    for name, (res, args) in built_lib.PROTOTYPES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    return lib


def both(built_lib, san, call):
    a, b = call(built_lib.lib), call(san)
    assert a == b
    return a


def test_constant_builders(built_lib, san):
    from crychic_renderer_amd import scene
    for sigma in (0.5, 1.0, 2.5):
        def gauss(lib):
            w = (C.c_float * 11)()
            n = lib.crychic_calc_gauss_weights(C.c_float(sigma), w, 11)
            return n, bytes(w)
        both(built_lib, san, gauss)
    assert san.crychic_calc_gauss_weights(C.c_float(3.0), (C.c_float * 11)(), 11) < 0          # radius > MaxBlurRadius
    assert san.crychic_calc_gauss_weights(C.c_float(2.5), (C.c_float * 4)(), 4) < 0            # capacity too small

    def noise(lib):
        st = C.c_uint32(1)
        off = ((C.c_float * 4) * 14)()
        lib.crychic_build_offset_vectors(C.byref(st), off)
        tex = np.zeros((256, 256, 4), dtype=np.uint8)
        lib.crychic_build_random_vector_texture(C.byref(st), 1, tex.ctypes.data)
        return bytes(off), tex.tobytes(), st.value
    both(built_lib, san, noise)

    for (W, H, sd) in ((800, 600, 4096), (64, 36, 130)):
        cam = scene.default_camera(W, H)

        def cbs(lib):
            lv, lp, st = (np.zeros((4, 4, 4), dtype=np.float32) for _ in range(3))
            ld = (C.c_float * 3)(0.57735, -0.57735, 0.57735)
            assert lib.crychic_update_cascade_shadow_transform(C.byref(cam), ld, sd, lv.ctypes.data, lp.ctypes.data, st.ctypes.data) == 0
            p, s = built_lib.PassConstants(), built_lib.SsaoConstants()
            dirs = np.asarray(scene.BASE_LIGHT_DIRS, dtype=np.float32)
            assert lib.crychic_update_main_pass_cb(C.byref(cam), W, H, st.ctypes.data, dirs.ctypes.data, C.byref(p)) == 0
            off = ((C.c_float * 4) * 14)()
            assert lib.crychic_update_ssao_cb(C.byref(cam), W, H, C.cast(off, C.c_void_p), C.byref(s)) == 0
            return lv.tobytes(), lp.tobytes(), st.tobytes(), bytes(p), bytes(s)
        both(built_lib, san, cbs)


def test_culling_and_geometry(built_lib, san, tmp_path):
    from crychic_renderer_amd import scene
    cam = scene.default_camera(800, 600)
    rng = np.random.default_rng(3)
    worlds = np.tile(np.eye(4, dtype=np.float32), (200, 1, 1))
    worlds[:, 3, :3] = rng.uniform(-60, 60, size=(200, 3)).astype(np.float32)

    def cull(lib):
        vis = np.zeros(200, dtype=np.uint8)
        c, e = (C.c_float * 3)(0, 0, 0), (C.c_float * 3)(0.5, 0.5, 0.5)
        n = lib.crychic_frustum_cull(C.byref(cam), c, e, worlds.ctypes.data, 200, vis.ctypes.data)
        return n, vis.tobytes()
    n, _ = both(built_lib, san, cull)
    assert 0 < n < 200

    def shapes(lib):
        out = []
        for make, args in ((lib.crychic_create_box, (C.c_float(1), C.c_float(1), C.c_float(1), 3)), (lib.crychic_create_grid, (C.c_float(20), C.c_float(30), 60, 40))):
            ni = C.c_uint32()
            nv = make(*args, None, 0, None, 0, C.byref(ni))
            v = np.zeros((nv, 11), dtype=np.float32); idx = np.zeros(ni.value, dtype=np.uint32)
            assert make(*args, v.ctypes.data, nv, idx.ctypes.data, ni.value, C.byref(ni)) == nv
            assert make(*args, v.ctypes.data, nv - 1, idx.ctypes.data, ni.value, C.byref(ni)) < 0      # capacity check, no overrun
            out.append((v.tobytes(), idx.tobytes()))
        return out
    both(built_lib, san, shapes)

    mesh = tmp_path / "m.txt"
    verts = rng.standard_normal((50, 6))
    tris = rng.integers(0, 50, size=(80, 3))
    mesh.write_text("VertexCount: 50\nTriangleCount: 80\nVertexList (pos, normal)\n{\n" + "".join("\t%f %f %f %f %f %f\n" % tuple(r) for r in verts) +
                    "}\nTriangleList\n{\n" + "".join("\t%d %d %d\n" % tuple(t) for t in tris) + "}\n")

    def load(lib):
        nv, ni = C.c_uint32(), C.c_uint32()
        assert lib.crychic_load_mesh_text(str(mesh).encode(), None, 0, None, 0, C.byref(nv), C.byref(ni)) >= 0
        v = np.zeros((nv.value, 11), dtype=np.float32); idx = np.zeros(ni.value, dtype=np.uint32)
        assert lib.crychic_load_mesh_text(str(mesh).encode(), v.ctypes.data, nv.value, idx.ctypes.data, ni.value, C.byref(nv), C.byref(ni)) >= 0
        assert lib.crychic_load_mesh_text(str(mesh).encode(), v.ctypes.data, nv.value - 1, idx.ctypes.data, ni.value, C.byref(nv), C.byref(ni)) < 0
        return nv.value, ni.value, v.tobytes(), idx.tobytes()
    assert both(built_lib, san, load)[:2] == (50, 240)
    # malformed files are refused, not over-read
    bad = tmp_path / "bad.txt"
    bad.write_text("VertexCount: 5\nTriangleCount: 9\nVertexList (pos, normal)\n{\n 1 2 3\n")
    nv, ni = C.c_uint32(), C.c_uint32()
    assert san.crychic_load_mesh_text(str(bad).encode(), None, 0, None, 0, C.byref(nv), C.byref(ni)) < 0 or nv.value == 5


def _dds(fourcc, w, h, payload, rgb_masks=None):
    hdr = bytearray(128)
    hdr[0:4] = b"DDS "
    struct.pack_into("<IIIIIII", hdr, 4, 124, 0x1007, h, w, 0, 0, 1)
    if fourcc:
        struct.pack_into("<II4s", hdr, 76, 32, 0x4, fourcc)
    else:
        struct.pack_into("<IIIIIIII", hdr, 76, 32, 0x41, 0, 32, *rgb_masks)
    return bytes(hdr) + payload


def test_dds_decoder(built_lib, san, tmp_path):
    rng = np.random.default_rng(5)
    cases = {"dxt1.dds": _dds(b"DXT1", 16, 8, rng.integers(0, 256, 16 * 8 // 2, dtype=np.uint8).tobytes()),
             "dxt5.dds": _dds(b"DXT5", 8, 8, rng.integers(0, 256, 8 * 8, dtype=np.uint8).tobytes()),
             "argb.dds": _dds(None, 5, 3, rng.integers(0, 256, 5 * 3 * 4, dtype=np.uint8).tobytes(), (0x00FF0000, 0x0000FF00, 0x000000FF, 0xFF000000)),
             "short.dds": _dds(b"DXT5", 64, 64, b"\x00" * 100),      # truncated payload
             "tiny.dds": b"DDS \x7c"}
    for name, blob in cases.items():
        path = tmp_path / name
        path.write_bytes(blob)

        def dec(lib):
            w, h = C.c_uint32(), C.c_uint32()
            rc = lib.crychic_load_dds_rgba8(str(path).encode(), None, 0, C.byref(w), C.byref(h))
            if rc < 0:
                return rc, None
            buf = np.zeros(w.value * h.value * 4, dtype=np.uint8)
            assert lib.crychic_load_dds_rgba8(str(path).encode(), buf.ctypes.data, buf.size - 1, C.byref(w), C.byref(h)) < 0     # capacity check
            rc = lib.crychic_load_dds_rgba8(str(path).encode(), buf.ctypes.data, buf.size, C.byref(w), C.byref(h))
            return rc, buf.tobytes()
        rc, _ = both(built_lib, san, dec)
        assert (rc < 0) == name.startswith(("short", "tiny")), (name, rc)
    ppm = tmp_path / "o.ppm"
    img = rng.integers(0, 256, (3, 5, 4), dtype=np.uint8)
    assert san.crychic_save_ppm(str(ppm).encode(), img.ctypes.data, 5, 3) == 0
    assert ppm.read_bytes() == b"P6\n5 3\n255\n" + img[..., :3].tobytes()
