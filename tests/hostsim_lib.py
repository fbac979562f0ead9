"""Builds and loads tests/hostsim/libhostsim.so: the product's per-pixel kernel bodies compiled for the host
(TEST HARNESS ONLY -- lets the CPU-only tier compare the kernel text with the oracle bit for bit)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "hostsim", "hostsim.cpp")
LIB = os.path.join(ROOT, "tests", "hostsim", "libhostsim.so")
CSRC = os.path.join(ROOT, "crychic_renderer_amd", "csrc")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


SANITIZE = os.environ.get("CRYCHIC_SANITIZE") == "1"      # tools/sanitize.sh: ASan + UBSan build of the kernel bodies
SAN_DIR = os.path.join(ROOT, "tests", "hostsim", "_san")
SAN_FLAGS = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined,float-cast-overflow", "-fno-sanitize-recover=all", "-shared-libsan"]


def build_sanitized(name, sources, extra=()):
    """clang ASan + UBSan shared object under tests/hostsim/_san (loaded with the sanitizer runtime preloaded: tools/sanitize.sh)."""
    os.makedirs(SAN_DIR, exist_ok=True)
    out = os.path.join(SAN_DIR, name)
    deps = list(sources) + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")] + [os.path.join(ROOT, "include", "crychic_hip.h")]
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.run([CLANG, "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-mfma"] + SAN_FLAGS + list(extra) +
                       ["-I", os.path.join(ROOT, "include"), "-I", CSRC] + list(sources) + ["-o", out], check=True)
    return out


def build():
    if SANITIZE:
        return build_sanitized("libhostsim.so", [SRC])
    deps = [SRC] + [os.path.join(CSRC, f) for f in ("devmath.hpp", "gamma_pow.inc", "ssao_core.hpp", "blur_tiles.hpp", "light_core.hpp", "raster_core.hpp")]
    if not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        subprocess.run([CLANG, "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-mfma",
                        "-I", os.path.join(ROOT, "include"), "-I", CSRC, SRC, "-o", LIB], check=True)
    return LIB


class HostSim:
    def __init__(self):
        self.lib = L = C.CDLL(build())
        f, u32, i, vp = C.c_float, C.c_uint32, C.c_int, C.c_void_p
        for n in ("hs_d24_to_float", "hs_unorm16_to_float", "hs_unorm8_to_float"):
            getattr(L, n).restype = f; getattr(L, n).argtypes = [u32]
        L.hs_half_to_float.restype = f; L.hs_half_to_float.argtypes = [C.c_uint16]
        for n in ("hs_det_sin", "hs_det_cos", "hs_det_log2", "hs_det_exp2"):
            getattr(L, n).restype = f; getattr(L, n).argtypes = [f]
        L.hs_det_pow.restype = f; L.hs_det_pow.argtypes = [f, f]
        L.hs_nrand.restype = f; L.hs_nrand.argtypes = [f, f]
        L.hs_check_d24_threshold.restype = u32
        L.hs_eval_array.argtypes = [i, C.c_size_t, vp, vp, vp]
        L.hs_ssao.argtypes = [vp, vp, vp, vp, vp, vp, u32, u32, u32, u32]
        L.hs_ssao_path.argtypes = [vp, vp, vp, vp, vp, vp, u32, u32, u32, u32, i]
        L.hs_last_sky_waves.restype = u32
        L.hs_last_culled_taps.restype = u32
        L.hs_last_clear_cell_taps.restype = u32
        L.hs_blur.argtypes = [vp, vp, vp, vp, u32, u32, i, u32, u32]
        L.hs_blur_chain.argtypes = [vp, vp, vp, vp, u32, u32, i, u32, u32, i, i]
        L.hs_last_settled_tiles.restype = u32
        L.hs_blur_chain_ssao_plane.restype = i; L.hs_blur_chain_ssao_plane.argtypes = [i]
        L.hs_blur_chain_ssao_rows.argtypes = [i, u32, u32, u32, vp, vp]
        L.hs_set_stamp.argtypes = [u32]
        L.hs_set_prep_margin.argtypes = [i]
        L.hs_last_unprepared_rows.restype = u32
        L.hs_rasterize.restype = i
        L.hs_rasterize.argtypes = [i, vp, vp, vp, u32, vp, u32, vp, u32, u32, u32, i, f, vp, vp, vp, vp, vp]
        L.hs_light.argtypes = [vp, vp, vp, vp, vp, vp, vp, u32, vp, u32, vp, vp, u32, u32, u32, u32, i, f, u32, vp, u32]

    def eval_array(self, kind, a, b=None):
        a = np.ascontiguousarray(a); out = np.zeros(a.shape, dtype=np.float32)
        b = np.ascontiguousarray(b) if b is not None else a
        self.lib.hs_eval_array(kind, a.size, a.ctypes.data, b.ctypes.data, out.ctypes.data)
        return out

    def ssao(self, cb, normal_f16, depth_u32, randvec_u8, edge_bytes, row0=0, rows=None, emit=True, pairs=True, cull=True, edge=None, stamp=1, margin=-1):
        """pairs=True: the taps gather from the decoded depth-pairs plane (the product's path when it has a workspace), and with
        cull=True skip the ones the nearest-depth map proves to add nothing; pairs=False: from the raw D24 plane.
        edge: a workspace to (re)use as it is -- like the device, nothing clears it; stamp: the frame stamp of this call;
        margin: texel rows the depth pass visits beyond the rows of the call (-1: the product's default)."""
        H, W = depth_u32.shape
        rows = H // 2 - row0 if rows is None else rows
        out = np.zeros((H // 2, W // 2), dtype=np.uint16)
        if edge is None:
            edge = np.zeros((edge_bytes,), dtype=np.uint8)
        n = np.ascontiguousarray(normal_f16.view(np.uint16)); d = np.ascontiguousarray(depth_u32); r = np.ascontiguousarray(randvec_u8)
        self.lib.hs_set_stamp(int(stamp))
        self.lib.hs_set_prep_margin(int(margin))
        self.lib.hs_ssao_path(C.addressof(cb), n.ctypes.data, d.ctypes.data, r.ctypes.data, out.ctypes.data if emit else None,
                              edge.ctypes.data, W, H, row0, rows, (1 if cull else 2) if pairs else 0)
        return out, edge

    def blur(self, cb, edge, ambient_in, W, H, horizontal, row0=0, rows=None):
        rows = H // 2 - row0 if rows is None else rows
        out = np.zeros((H // 2, W // 2), dtype=np.uint16)
        a = np.ascontiguousarray(ambient_in)
        self.lib.hs_blur(C.addressof(cb), edge.ctypes.data, a.ctypes.data, out.ctypes.data, W, H, 1 if horizontal else 0, row0, rows)
        return out

    def compute_ssao(self, cb, normal_f16, depth_u32, randvec_u8, edge_bytes, blur_count, row0=0, rows=None, use_exit=True, ones_margin=None,
                     edge=None, stamp=1, margin=-1):
        """Ssao::ComputeSsao as api.cpp sequences it (SSAO pass, then the two-launch blur chain of blur_tiles.hpp through the
        kernels' own tile bodies).  Returns (ambient0, edge); rows [row0, row0 + rows) of ambient0 are the result.
        ones_margin: the reach the unoccluded-tile exit assumes (default: what api.cpp passes, 5 per iteration)."""
        H, W = depth_u32.shape
        h2 = H // 2
        rows = h2 - row0 if rows is None else rows
        r0, rn = C.c_uint32(), C.c_uint32()
        self.lib.hs_blur_chain_ssao_rows(blur_count, row0, rows, h2, C.addressof(r0), C.addressof(rn))
        out, edge = self.ssao(cb, normal_f16, depth_u32, randvec_u8, edge_bytes, r0.value, rn.value, edge=edge, stamp=stamp, margin=margin)
        planes = [np.full((h2, W // 2), 0xABCD, dtype=np.uint16), np.full((h2, W // 2), 0xABCD, dtype=np.uint16)]     # junk where nothing writes
        sp = self.lib.hs_blur_chain_ssao_plane(blur_count)
        planes[sp][r0.value:r0.value + rn.value] = out[r0.value:r0.value + rn.value]
        self.lib.hs_blur_chain(C.addressof(cb), edge.ctypes.data, planes[0].ctypes.data, planes[1].ctypes.data, W, H, blur_count, row0, rows,
                               1 if use_exit else 0, 5 * blur_count if ones_margin is None else int(ones_margin))
        return planes[0], edge

    def blur_chain(self, cb, edge, ambient_in, W, H, blur_count, use_exit=False):
        """The blur chain alone on a caller-made ambient map (whole frame, no exit unless the workspace's map belongs to it)."""
        h2 = H // 2
        planes = [np.full((h2, W // 2), 0xABCD, dtype=np.uint16), np.full((h2, W // 2), 0xABCD, dtype=np.uint16)]
        planes[self.lib.hs_blur_chain_ssao_plane(blur_count)][:] = ambient_in
        self.lib.hs_blur_chain(C.addressof(cb), edge.ctypes.data, planes[0].ctypes.data, planes[1].ctypes.data, W, H, blur_count, 0, h2,
                               1 if use_exit else 0, 5 * blur_count)
        return planes[0]

    def rasterize(self, mode, view_t, viewproj_t, items, materials, textures, W, H, depth_bias=0, slope_bias=0.0):
        from crychic_renderer_amd._lib import DrawItem, Texture
        arr = (DrawItem * len(items))()
        keep = []
        for k, (v, idx, inst) in enumerate(items):
            v = np.ascontiguousarray(v); idx = np.ascontiguousarray(idx); inst = np.ascontiguousarray(inst)
            keep += [v, idx, inst]
            arr[k] = DrawItem(v.ctypes.data, len(v), idx.ctypes.data, len(idx), 0, 0, inst.ctypes.data, len(inst))
        tex = (Texture * max(1, len(textures or [])))()
        for k, t in enumerate(textures or []):
            if t is not None:
                from crychic_renderer_amd.geometry import texture_levels
                flat, tw, th, levels = texture_levels(t); keep.append(flat)
                tex[k] = Texture(flat.ctypes.data, tw, th, levels)
        mats = np.ascontiguousarray(materials) if materials is not None else None
        depth = np.zeros((H, W), np.uint32)
        normal = np.zeros((H, W, 4), np.uint16) if mode == 1 else None
        g = [np.zeros((H, W, 4), np.float32) for _ in range(3)] if mode == 2 else [None] * 3
        view_t = np.ascontiguousarray(view_t, np.float32); viewproj_t = np.ascontiguousarray(viewproj_t, np.float32)
        n = self.lib.hs_rasterize(mode, view_t.ctypes.data, viewproj_t.ctypes.data, arr, len(items),
                                  mats.ctypes.data if mats is not None else None, len(mats) if mats is not None else 0,
                                  tex if textures else None, len(textures or []), W, H, depth_bias, slope_bias, depth.ctypes.data,
                                  normal.ctypes.data if normal is not None else None, *[x.ctypes.data if x is not None else None for x in g])
        if n < 0:
            raise RuntimeError("hs_rasterize failed (%d)" % n)
        return {"depth": depth, "normal": normal.view(np.float16) if normal is not None else None, "g0": g[0], "g1": g[1], "g2": g[2], "tris": n}

    def light(self, cb, g0, g1, g2, depth_u32, ambient, shadow_u32, cube_u8, num_dir_lights, pcf_radius, flags=0,
              want_radiance=False, point_lights=None, cube_dim=None, cube_levels=0):
        H, W = depth_u32.shape
        flags = int(flags) | ((int(cube_levels) & 15) << 16)
        out = np.zeros((H, W, 4), dtype=np.uint8)
        rad = np.zeros((H, W, 4), dtype=np.float32) if want_radiance else None
        g0, g1, g2 = (np.ascontiguousarray(g) for g in (g0, g1, g2))
        d = np.ascontiguousarray(depth_u32); s = np.ascontiguousarray(shadow_u32); c = np.ascontiguousarray(cube_u8)
        a = np.ascontiguousarray(ambient) if ambient is not None else None
        sh = (C.c_void_p * 4)(*[s[k].ctypes.data for k in range(4)])
        self.lib.hs_light(C.addressof(cb), g0.ctypes.data, g1.ctypes.data, g2.ctypes.data, d.ctypes.data,
                          a.ctypes.data if a is not None else None, sh, s.shape[1], c.ctypes.data, int(cube_dim or c.shape[1]),
                          out.ctypes.data, rad.ctypes.data if rad is not None else None, W, H, 0, H, num_dir_lights,
                          pcf_radius, flags, C.addressof(point_lights) if point_lights is not None else None,
                          len(point_lights) if point_lights is not None else 0)
        return (out, rad) if want_radiance else out


_HS = None


def load():
    global _HS
    if _HS is None:
        _HS = HostSim()
    return _HS
