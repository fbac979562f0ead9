"""ctypes binding of libcrychic_hip.so (include/crychic_hip.h).

The library is the product: if it has not been built (python -m crychic_renderer_amd.build, or
__graft_entry__.build()) importing this module raises -- there is no Python or CPU fallback.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# CRYCHIC_LIB: a probe build of the same library (tools/probes/variants.sh builds them with extra -D flags); never set in tests or bench runs
LIB_PATH = os.environ.get("CRYCHIC_LIB") or os.path.join(HERE, "libcrychic_hip.so")

MAX_LIGHTS = 16
LIGHT_SKY = 1
COMM_ID_BYTES = 128


class Light(C.Structure):
    _fields_ = [("Strength", C.c_float * 3), ("FalloffStart", C.c_float), ("Direction", C.c_float * 3),
                ("FalloffEnd", C.c_float), ("Position", C.c_float * 3), ("SpotPower", C.c_float)]


class PassConstants(C.Structure):
    _fields_ = [("View", C.c_float * 16), ("InvView", C.c_float * 16), ("Proj", C.c_float * 16),
                ("InvProj", C.c_float * 16), ("ViewProj", C.c_float * 16), ("InvViewProj", C.c_float * 16),
                ("ViewProjTex", C.c_float * 16), ("ShadowTransforms", (C.c_float * 16) * 12),
                ("EyePosW", C.c_float * 3), ("cbPerObjectPad1", C.c_float), ("RenderTargetSize", C.c_float * 2),
                ("InvRenderTargetSize", C.c_float * 2), ("NearZ", C.c_float), ("FarZ", C.c_float),
                ("TotalTime", C.c_float), ("DeltaTime", C.c_float), ("AmbientLight", C.c_float * 4),
                ("Lights", Light * MAX_LIGHTS)]


class SsaoConstants(C.Structure):
    _fields_ = [("Proj", C.c_float * 16), ("InvProj", C.c_float * 16), ("ProjTex", C.c_float * 16),
                ("OffsetVectors", (C.c_float * 4) * 14), ("BlurWeights", (C.c_float * 4) * 3),
                ("RenderTargetSize", C.c_float * 2), ("InvRenderTargetSize", C.c_float * 2),
                ("OcclusionRadius", C.c_float), ("OcclusionFadeStart", C.c_float), ("OcclusionFadeEnd", C.c_float),
                ("SurfaceEpsilon", C.c_float)]


class Camera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("look", C.c_float * 3), ("up", C.c_float * 3), ("fovY", C.c_float),
                ("aspect", C.c_float), ("nearZ", C.c_float), ("farZ", C.c_float)]


class FrameDesc(C.Structure):
    _fields_ = [("W", C.c_uint32), ("H", C.c_uint32), ("blurCount", C.c_int), ("numDirLights", C.c_int),
                ("pcfSearchRadius", C.c_float), ("flags", C.c_uint32), ("row0", C.c_uint32), ("rows", C.c_uint32),
                ("normal_dev", C.c_void_p), ("depth_dev", C.c_void_p), ("randvec_dev", C.c_void_p),
                ("g0_dev", C.c_void_p), ("g1_dev", C.c_void_p), ("g2_dev", C.c_void_p),
                ("shadow_dev", C.c_void_p * 4), ("shadowDim", C.c_uint32), ("cube_dev", C.c_void_p),
                ("cubeDim", C.c_uint32), ("ambient0_dev", C.c_void_p), ("ambient1_dev", C.c_void_p),
                ("edge_dev", C.c_void_p), ("out_rgba8_dev", C.c_void_p),
                ("point_lights_dev", C.c_void_p), ("numPointLights", C.c_uint32)]


class DrawItem(C.Structure):
    _fields_ = [("vertices_dev", C.c_void_p), ("vertexCount", C.c_uint32), ("indices_dev", C.c_void_p), ("indexCount", C.c_uint32),
                ("startIndexLocation", C.c_uint32), ("baseVertexLocation", C.c_int32), ("instances_dev", C.c_void_p),
                ("instanceCount", C.c_uint32)]


class Texture(C.Structure):
    _fields_ = [("rgba8_dev", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("mipLevels", C.c_uint32)]


class PassTimes(C.Structure):
    _fields_ = [("ssao_ms", C.c_float), ("blur_ms", C.c_float), ("light_ms", C.c_float), ("total_ms", C.c_float)]


assert C.sizeof(Light) == 48 and C.sizeof(PassConstants) == 2048 and C.sizeof(SsaoConstants) == 496

_u32, _i, _f, _vp, _sz = C.c_uint32, C.c_int, C.c_float, C.c_void_p, C.c_size_t
_P = C.POINTER

# name -> (restype, argtypes); every symbol include/crychic_hip.h declares
PROTOTYPES = {
    "crychic_ctx_create": (_i, [_i, _P(_vp)]),
    "crychic_ctx_destroy": (None, [_vp]),
    "crychic_last_error": (C.c_char_p, []),
    "crychic_version": (C.c_char_p, []),
    "crychic_ctx_device_name": (C.c_char_p, [_vp]),
    "crychic_calc_gauss_weights": (_i, [_f, _P(_f), _i]),
    "crychic_msvc_rand": (_i, [_P(_u32)]),
    "crychic_build_offset_vectors": (None, [_P(_u32), _P(C.c_float * 4)]),
    "crychic_build_random_vector_texture": (None, [_P(_u32), _i, _vp]),
    "crychic_update_cascade_shadow_transform": (_i, [_P(Camera), _P(_f), _u32, _vp, _vp, _vp]),
    "crychic_update_main_pass_cb": (_i, [_P(Camera), _u32, _u32, _vp, _vp, _P(PassConstants)]),
    "crychic_update_ssao_cb": (_i, [_P(Camera), _u32, _u32, _vp, _P(SsaoConstants)]),
    "crychic_pcf_search_radius": (_f, [_u32, _i]),
    "crychic_edge_plane_bytes": (_sz, [_u32, _u32]),
    "crychic_ssao": (_i, [_vp, _P(SsaoConstants), _vp, _vp, _vp, _vp, _vp, _u32, _u32, _u32, _u32, _vp]),
    "crychic_ssao_edges": (_i, [_vp, _P(SsaoConstants), _vp, _vp, _vp, _u32, _u32, _u32, _u32, _vp]),
    "crychic_ssao_blur": (_i, [_vp, _P(SsaoConstants), _vp, _vp, _vp, _u32, _u32, _i, _u32, _u32, _vp]),
    "crychic_ssao_compute": (_i, [_vp, _P(SsaoConstants), _vp, _vp, _vp, _vp, _vp, _vp, _u32, _u32, _i, _u32, _u32,
                                  _vp]),
    "crychic_deferred_light": (_i, [_vp, _P(PassConstants), _vp, _vp, _vp, _vp, _vp, _P(_vp), _u32, _vp, _u32, _vp,
                                    _vp, _u32, _u32, _u32, _u32, _i, _f, _u32, _vp]),
    "crychic_deferred_light_points": (_i, [_vp, _P(PassConstants), _vp, _vp, _vp, _vp, _vp, _P(_vp), _u32, _vp, _u32, _vp,
                                           _vp, _u32, _u32, _u32, _u32, _i, _f, _u32, _vp, _u32, _vp]),
    "crychic_draw_hot_path": (_i, [_vp, _P(SsaoConstants), _P(PassConstants), _P(FrameDesc), _vp]),
    "crychic_frustum_cull": (_i, [_P(Camera), _P(_f), _P(_f), _vp, _u32, _vp]),
    "crychic_ctx_set_profiling": (_i, [_vp, _i]),
    "crychic_ctx_last_pass_times": (_i, [_vp, _P(PassTimes)]),
    "crychic_strip_rows": (_i, [_u32, _i, _i, _P(_u32), _P(_u32)]),
    "crychic_comm_unique_id": (_i, [_vp]),
    "crychic_comm_create": (_i, [_vp, _i, _i, _vp, _P(_vp)]),
    "crychic_comm_create_all": (_i, [_P(_vp), _i, _P(_vp)]),
    "crychic_comm_destroy": (None, [_vp]),
    "crychic_comm_abort": (_i, [_vp]),
    "crychic_comm_rank": (_i, [_vp]),
    "crychic_comm_size": (_i, [_vp]),
    "crychic_comm_async_error": (_i, [_vp]),
    "crychic_allgather_frame": (_i, [_vp, _vp, _u32, _u32, _P(_u32), _vp]),
    "crychic_allgather_frame_all": (_i, [_P(_vp), _i, _P(_vp), _u32, _u32, _P(_u32), _P(_vp)]),
    "crychic_comm_barrier": (_i, [_vp, _vp]),
    "crychic_draw_hot_path_shared": (_i, [_vp, _P(SsaoConstants), _P(PassConstants), _P(FrameDesc), _P(_u32), _u32, _vp]),
    "crychic_create_box": (_i, [_f, _f, _f, _u32, _vp, _u32, _vp, _u32, _P(_u32)]),
    "crychic_create_grid": (_i, [_f, _f, _u32, _u32, _vp, _u32, _vp, _u32, _P(_u32)]),
    "crychic_load_mesh_text": (_i, [C.c_char_p, _vp, _u32, _vp, _u32, _P(_u32), _P(_u32)]),
    "crychic_load_dds_rgba8": (_i, [C.c_char_p, _vp, _sz, _P(_u32), _P(_u32)]),
    "crychic_load_dds_rgba8_mips": (_i, [C.c_char_p, _vp, _sz, _P(_u32), _P(_u32), _P(_u32)]),
    "crychic_load_dds_cube_rgba8": (_i, [C.c_char_p, _vp, _sz, _P(_u32)]),
    "crychic_load_dds_cube_rgba8_mips": (_i, [C.c_char_p, _vp, C.c_size_t, _P(_u32), _P(_u32)]),
    "crychic_save_ppm": (_i, [C.c_char_p, _vp, _u32, _u32]),
    "crychic_raster_workspace_bytes": (_sz, [C.c_uint64, _u32, _u32]),
    "crychic_raster_status": (_i, [_vp, _vp, _P(_u32)]),
    "crychic_blur_chain_status": (_i, [_vp, _vp, _P(_u32)]),
    "crychic_draw_scene_to_shadow_map": (_i, [_vp, _P(PassConstants), _P(DrawItem), _u32, _vp, _u32, _i, _f, _vp, _sz, _vp]),
    "crychic_draw_scene_to_shadow_maps": (_i, [_vp, _vp, _u32, _P(DrawItem), _u32, _vp, _u32, _i, _f, _vp, _sz, _vp]),
    "crychic_draw_normals_and_depth": (_i, [_vp, _P(PassConstants), _P(DrawItem), _u32, _vp, _vp, _u32, _u32, _vp, _sz, _vp]),
    "crychic_draw_gbuffer": (_i, [_vp, _P(PassConstants), _P(DrawItem), _u32, _vp, _u32, _P(Texture), _u32, _vp, _vp, _vp, _vp,
                                  _u32, _u32, _vp, _sz, _vp]),
    "crychic_draw_normals_depth_and_gbuffer": (_i, [_vp, _P(PassConstants), _P(DrawItem), _u32, _vp, _u32, _P(Texture), _u32, _vp, _vp, _vp, _vp, _vp,
                                               _u32, _u32, _vp, _sz, _vp]),
    "crychic_draw_gbuffer_rows": (_i, [_vp, _P(PassConstants), _P(DrawItem), _u32, _vp, _u32, _P(Texture), _u32, _vp, _vp, _vp, _vp,
                                       _u32, _u32, _u32, _u32, _vp, _sz, _vp]),
    "crychic_draw_normals_depth_and_gbuffer_rows": (_i, [_vp, _P(PassConstants), _P(DrawItem), _u32, _vp, _u32, _P(Texture), _u32, _vp, _vp, _vp, _vp,
                                                        _vp, _u32, _u32, _u32, _u32, _vp, _sz, _vp]),
}


class CrychicError(RuntimeError):
    """Mirror of the reference's DxException (Common/d3dUtil.h:132-144) for a failed C-ABI call."""

    def __init__(self, status, what):
        super().__init__("crychic status %d: %s" % (status, what))
        self.status = status


def _preload_hip_runtime():
    """PyTorch wheels bundle their own libamdhip64.so (same soname as /opt/rocm's).  Device pointers handed to the C
    ABI come from torch's allocator, so both must live in ONE HIP runtime: load torch's copy first, then the
    dynamic linker resolves libcrychic_hip.so's NEEDED libamdhip64.so.7 to it by soname.  Without torch (pure C++
    callers) the library simply binds to /opt/rocm's runtime."""
    try:
        import torch  # noqa: F401  (imports its bundled runtime)
    except ImportError:
        return
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def load():
    _preload_hip_runtime()
    # Rebuild a missing / stale library when the toolchain is present (a checkout whose sources are newer than the .so);
    # this is a build step, not a fallback: without hipcc and without a library the import fails below.
    try:
        from . import build as _build
        if _build._stale() and os.path.exists(os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")):
            _build.build(verbose=False)
    except Exception as e:  # a failed rebuild must not be hidden behind an old binary
        if os.path.exists(LIB_PATH):
            raise ImportError("libcrychic_hip.so is stale and rebuilding it failed: %s" % e)
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libcrychic_hip.so is not built (%s). Run `python -m crychic_renderer_amd.build`; "
            "crychic_renderer_amd has no CPU/Python fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = load()


def check(status):
    if status < 0:
        raise CrychicError(status, lib.crychic_last_error().decode("utf-8", "replace"))
    return status
