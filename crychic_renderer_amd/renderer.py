"""Python mirror of the reference's pass objects over the C ABI.

Class and method names follow the reference (Ssao.h:10-125, DeferredShading.h:4-45, ShadowMap.h:4-47,
CRYCHIC.h:56-190) so that the parity tests read like calls into the reference; D3D12 handles become torch
tensors that own HBM (PyTorch is used for device memory and streams only -- every pixel is computed by
libcrychic_hip.so).
"""
import ctypes as C

import torch

from . import _lib
from ._lib import DrawItem, FrameDesc, PassConstants, PassTimes, SsaoConstants, Texture, check, lib


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(None)


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class Context:
    """One GPU, one context (D3DApp::InitDirect3D, Common/d3dApp.cpp:415-479)."""

    def __init__(self, device_ordinal=0):
        self.handle = C.c_void_p()
        check(lib.crychic_ctx_create(int(device_ordinal), C.byref(self.handle)))
        self.device = torch.device("cuda", int(device_ordinal))

    @property
    def device_name(self):
        return lib.crychic_ctx_device_name(self.handle).decode()

    def close(self):
        if self.handle:
            lib.crychic_ctx_destroy(self.handle)
            self.handle = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Ssao:
    """Ssao.h:10-125.  Owns the view-normal map, the two half-res ambient maps, the random-vector map and the
    half-res edge workspace; ComputeSsao runs SSAO + blurCount x (H, V) blur on the caller's stream."""

    MaxBlurRadius = 5  # Ssao.h:24

    def __init__(self, ctx, width, height, randvec):
        self.ctx = ctx
        self.mRandomVectorMap = randvec
        self.OnResize(width, height)

    def OnResize(self, newWidth, newHeight):  # Ssao.cpp:164-183
        if newWidth % 2 or newHeight % 2:
            raise _lib.CrychicError(-1, "frame size must be even")
        self.mRenderTargetWidth, self.mRenderTargetHeight = newWidth, newHeight
        dev = self.ctx.device
        w2, h2 = newWidth // 2, newHeight // 2
        self.mNormalMap = torch.zeros((newHeight, newWidth, 4), device=dev, dtype=torch.float16)
        self.mNormalMap[..., 2] = 1.0  # clear (0,0,1,0), Ssao.cpp:317
        self.mAmbientMap0 = torch.full((h2, w2), -1, device=dev, dtype=torch.int16)  # R16_UNORM 1.0, Ssao.cpp:333
        self.mAmbientMap1 = torch.full((h2, w2), -1, device=dev, dtype=torch.int16)
        self.mEdge = torch.zeros((int(lib.crychic_edge_plane_bytes(newWidth, newHeight)),), device=dev, dtype=torch.uint8)

    def SsaoMapWidth(self):  # Ssao.cpp:22-25
        return self.mRenderTargetWidth // 2

    def SsaoMapHeight(self):  # Ssao.cpp:27-30
        return self.mRenderTargetHeight // 2

    @staticmethod
    def CalcGaussWeights(sigma):  # Ssao.cpp:37-68
        w = (C.c_float * 11)()
        n = check(lib.crychic_calc_gauss_weights(float(sigma), w, 11))
        return [w[i] for i in range(n)]

    def NormalMap(self):  # Ssao.cpp:70-73
        return self.mNormalMap

    def AmbientMap(self):  # Ssao.cpp:75-78
        return self.mAmbientMap0

    def ComputeSsao(self, depth, ssao_cb, blurCount, row0=0, rows=None):  # Ssao.cpp:185-229
        W, H = self.mRenderTargetWidth, self.mRenderTargetHeight
        rows = H // 2 - row0 if rows is None else rows
        check(lib.crychic_ssao_compute(self.ctx.handle, C.byref(ssao_cb), _ptr(self.mNormalMap), _ptr(depth),
                                       _ptr(self.mRandomVectorMap), _ptr(self.mAmbientMap0), _ptr(self.mAmbientMap1),
                                       _ptr(self.mEdge), W, H, int(blurCount), int(row0), int(rows),
                                       _stream(self.ctx.device)))

    def BlurAmbientMap(self, ssao_cb, horzBlur, row0=0, rows=None):  # Ssao.cpp:245-293
        W, H = self.mRenderTargetWidth, self.mRenderTargetHeight
        rows = H // 2 - row0 if rows is None else rows
        src, dst = (self.mAmbientMap0, self.mAmbientMap1) if horzBlur else (self.mAmbientMap1, self.mAmbientMap0)
        check(lib.crychic_ssao_blur(self.ctx.handle, C.byref(ssao_cb), _ptr(self.mEdge), _ptr(src), _ptr(dst), W, H,
                                    1 if horzBlur else 0, int(row0), int(rows), _stream(self.ctx.device)))


class DeferredShading:
    """DeferredShading.h:4-45: owns the G-buffer planes (R32G32B32A32_FLOAT, CRYCHIC.cpp:56-58).  The reference
    allocates four; GBuffer3 carries no information (GBuffer.hlsl:29) and is not allocated here."""

    def __init__(self, ctx, width, height):
        self.ctx = ctx
        self.OnResize(width, height)

    def OnResize(self, newWidth, newHeight):  # DeferredShading.cpp:79-93
        self.mWidth, self.mHeight = newWidth, newHeight
        self.mGBuffer = [torch.zeros((newHeight, newWidth, 4), device=self.ctx.device, dtype=torch.float32) for _ in range(3)]

    def Width(self):
        return self.mWidth

    def Height(self):
        return self.mHeight

    def Resource(self, index):  # DeferredShading.cpp:30-33
        return self.mGBuffer[index]


class ShadowMap:
    """ShadowMap.h:4-47: the cascade depth maps (D24 in uint32).  The reference allocates 12 x 12 slices of which
    4 cascades are valid (CRYCHIC.cpp:644)."""

    def __init__(self, ctx, width, height):
        if width != height:
            raise _lib.CrychicError(-1, "shadow maps are square")
        self.ctx = ctx
        self.mWidth, self.mHeight = width, height
        self.mShadowMap = torch.full((4, height, width), 0xFFFFFF, device=ctx.device, dtype=torch.int32)  # depth 1.0

    def Width(self):
        return self.mWidth

    def Height(self):
        return self.mHeight

    def Resource(self, index):
        return self.mShadowMap[index]


class Crychic:
    """Headless CRYCHIC (CRYCHIC.h:56-190): Draw() issues the hot part of CRYCHIC::Draw's deferred branch
    (CRYCHIC.cpp:220-221, 238-279) on this GPU's stream for the full-res rows [row0, row0 + rows)."""

    def __init__(self, ctx, width, height, randvec, cube, shadow_dim=4096):
        self.ctx = ctx
        self.mClientWidth, self.mClientHeight = width, height
        self.mShadowMap = ShadowMap(ctx, shadow_dim, shadow_dim)
        self.mSsao = Ssao(ctx, width, height, randvec)
        self.mDeferred = DeferredShading(ctx, width, height)
        self.mCubeMap = cube
        self.mCubeMapLevels = 1      # set_cube_map: > 1 = mCubeMap is a flat mip chain (CRYCHIC_LIGHT_CUBE_LEVELS), mCubeMapSize its level-0 face size
        self.mCubeMapSize = None
        self.mDepthStencilBuffer = torch.full((height, width), 0xFFFFFF, device=ctx.device, dtype=torch.int32)
        self.mBackBuffer = torch.zeros((height, width, 4), device=ctx.device, dtype=torch.uint8)
        self.mMainPassCB = PassConstants()
        self.mSsaoCB = SsaoConstants()
        self.blurCount = 3           # CRYCHIC.cpp:221
        self.numDirLights = 1        # NUM_DIR_LIGHTS of the deferred shader, Common.hlsl:6-8
        self.pcfSearchRadius = lib.crychic_pcf_search_radius(shadow_dim, 1)  # Common.hlsl:305 as written
        self.flags = 0
        self.mPointLights = None     # extension: torch uint8 tensor holding an array of Light structs (48 B each)
        self._desc = None

    def load_scene(self, planes):
        """Install externally produced input planes (scene.make_scene) in place of the producer passes."""
        self.mDepthStencilBuffer = planes["depth"]
        self.mSsao.mNormalMap = planes["normal"]
        self.mDeferred.mGBuffer = [planes["g0"], planes["g1"], planes["g2"]]
        self.mShadowMap.mShadowMap = planes["shadow"]
        self.mCubeMap = planes["cube"]
        self.mCubeMapLevels, self.mCubeMapSize = 1, None
        self.mSsao.mRandomVectorMap = planes["randvec"]
        self.mMainPassCB = planes["consts"].pass_cb
        self.mSsaoCB = planes["consts"].ssao_cb
        self._desc = None

    def frame_desc(self, row0=0, rows=None):
        W, H = self.mClientWidth, self.mClientHeight
        f = FrameDesc()
        f.W, f.H = W, H
        f.blurCount, f.numDirLights = int(self.blurCount), int(self.numDirLights)
        f.pcfSearchRadius, f.flags = float(self.pcfSearchRadius), int(self.flags) | ((int(self.mCubeMapLevels) & 15) << 16 if self.mCubeMapLevels > 1 else 0)
        f.row0, f.rows = int(row0), int(H - row0 if rows is None else rows)
        f.normal_dev = self.mSsao.mNormalMap.data_ptr()
        f.depth_dev = self.mDepthStencilBuffer.data_ptr()
        f.randvec_dev = self.mSsao.mRandomVectorMap.data_ptr()
        f.g0_dev, f.g1_dev, f.g2_dev = (g.data_ptr() for g in self.mDeferred.mGBuffer)
        for i in range(4):
            f.shadow_dev[i] = self.mShadowMap.mShadowMap[i].data_ptr()
        f.shadowDim = self.mShadowMap.Width()
        f.cube_dev, f.cubeDim = self.mCubeMap.data_ptr(), int(self.mCubeMapSize or self.mCubeMap.shape[1])
        f.ambient0_dev = self.mSsao.mAmbientMap0.data_ptr()
        f.ambient1_dev = self.mSsao.mAmbientMap1.data_ptr()
        f.edge_dev = self.mSsao.mEdge.data_ptr()
        f.out_rgba8_dev = self.mBackBuffer.data_ptr()
        if self.mPointLights is not None:
            f.point_lights_dev = self.mPointLights.data_ptr()
            f.numPointLights = self.mPointLights.numel() // 48
        return f

    def Draw(self, row0=0, rows=None, shared=None):  # CRYCHIC.cpp:172-306 (hot part)
        """shared = (communicator handle, bounds array or None, parts): crychic_draw_hot_path_shared -- the strip AND its exchange,
        the lighting pass in `parts` row ranges that travel while the next is lit (sharding.StripExchange.draw)."""
        # The descriptor only changes when a plane is re-allocated or a knob is turned: keep it across frames so the
        # per-frame host cost is one FFI call (matters once a strip takes tens of microseconds on 8 GPUs).  The key holds
        # every device pointer and size frame_desc() reads, so replacing any plane object invalidates the cached descriptor.
        ssao, sm = self.mSsao, self.mShadowMap.mShadowMap
        key = (row0, rows, self.mBackBuffer.data_ptr(), ssao.mAmbientMap0.data_ptr(), ssao.mAmbientMap1.data_ptr(), ssao.mEdge.data_ptr(),
               ssao.mNormalMap.data_ptr(), ssao.mRandomVectorMap.data_ptr(), self.mDepthStencilBuffer.data_ptr(),
               self.mDeferred.mGBuffer[0].data_ptr(), self.mDeferred.mGBuffer[1].data_ptr(), self.mDeferred.mGBuffer[2].data_ptr(),
               sm.data_ptr(), int(sm.shape[-1]), self.mCubeMap.data_ptr(), int(self.mCubeMapSize or self.mCubeMap.shape[1]), int(self.mCubeMapLevels),
               self.blurCount, self.numDirLights, self.pcfSearchRadius, self.flags,
               0 if self.mPointLights is None else self.mPointLights.data_ptr())
        if self._desc is None:
            self._desc = {}
        f = self._desc.get(key)
        if f is None:
            if len(self._desc) > 16:
                self._desc.clear()
            f = self._desc[key] = self.frame_desc(row0, rows)
        if shared is not None:
            check(lib.crychic_draw_hot_path_shared(shared[0], C.byref(self.mSsaoCB), C.byref(self.mMainPassCB), C.byref(f), shared[1],
                                                   int(shared[2]), _stream(self.ctx.device)))
            return
        check(lib.crychic_draw_hot_path(self.ctx.handle, C.byref(self.mSsaoCB), C.byref(self.mMainPassCB), C.byref(f),
                                        _stream(self.ctx.device)))

    def set_cube_map(self, cube, dim=None, levels=1):
        """The sky cube map: a 6 x dim x dim x 4 uint8 tensor (level 0 alone), or -- with `levels` > 1 -- the flat mip chain
        geometry.cube_mip_chain / load_dds_cube_mips produce (the reference binds the whole chain, CRYCHIC.cpp:1148-1151): the
        reflection and sky lookups are then trilinear (CRYCHIC_LIGHT_CUBE_LEVELS)."""
        self.mCubeMap = cube
        self.mCubeMapLevels = int(levels)
        self.mCubeMapSize = int(dim) if dim is not None else None
        self._desc = None

    def set_point_lights(self, lights):
        """Extension: `lights` is a ctypes array of Light (or None); copied to the device."""
        if lights is None or len(lights) == 0:
            self.mPointLights = None
        else:
            import numpy as np
            host = np.frombuffer(bytes(lights), dtype=np.uint8).copy()
            self.mPointLights = torch.from_numpy(host).to(self.ctx.device)
        self._desc = None

    def set_profiling(self, enabled):
        check(lib.crychic_ctx_set_profiling(self.ctx.handle, 1 if enabled else 0))

    def blur_chain_timed_out(self):
        """crychic_blur_chain_status: True if a workgroup of the most recent single-launch blur chain gave up waiting for a neighbour
        (synchronises the stream; the frame's ambient map would be wrong)."""
        flag = C.c_uint32(0)
        check(lib.crychic_blur_chain_status(self.ctx.handle, _stream(self.ctx.device), C.byref(flag)))
        return bool(flag.value)

    def last_pass_times(self):
        t = PassTimes()
        check(lib.crychic_ctx_last_pass_times(self.ctx.handle, C.byref(t)))
        return {"ssao_ms": t.ssao_ms, "blur_ms": t.blur_ms, "light_ms": t.light_ms, "total_ms": t.total_ms}


class SceneGeometry:
    """Device-resident vertex / index / instance / material / texture buffers of a set of render items, i.e. what
    CRYCHIC::BuildShapeGeometry + BuildMaterials + the per-frame InstanceBuffers hold (CRYCHIC.cpp:1250-1445, 515-592)."""

    def __init__(self, ctx, items, materials=None, textures=None):
        import numpy as np
        dev = ctx.device
        self.ctx = ctx
        self._keep = []
        self.items = (DrawItem * len(items))()
        self.triangles = 0
        for k, (v, idx, inst) in enumerate(items):
            tv = torch.from_numpy(np.ascontiguousarray(v).view(np.uint8).copy()).to(dev)
            ti = torch.from_numpy(np.ascontiguousarray(idx).view(np.int32).copy()).to(dev)
            tn = torch.from_numpy(np.ascontiguousarray(inst).view(np.uint8).copy()).to(dev)
            self._keep += [tv, ti, tn]
            self.items[k] = DrawItem(tv.data_ptr(), len(v), ti.data_ptr(), len(idx), 0, 0, tn.data_ptr(), len(inst))
            self.triangles += (len(idx) // 3) * len(inst)
        self.materials = None
        self.n_materials = 0
        if materials is not None:
            self.materials = torch.from_numpy(np.ascontiguousarray(materials).view(np.uint8).copy()).to(dev)
            self.n_materials = len(materials)
        self.textures = None
        self.n_textures = 0
        if textures:
            self.textures = (Texture * len(textures))()
            for k, t in enumerate(textures):
                if t is None:
                    continue
                from .geometry import texture_levels
                flat, tw, th, levels = texture_levels(t)          # one array = level 0 only; a list = a mip chain
                tt = torch.from_numpy(flat.copy()).to(dev)
                self._keep.append(tt)
                self.textures[k] = Texture(tt.data_ptr(), tw, th, levels)
            self.n_textures = len(textures)
        self._ws = {}

    def workspace(self, W, H, targets=1):
        key = (W, H, targets)
        if key not in self._ws:
            n = int(lib.crychic_raster_workspace_bytes(self.triangles * targets, W, H))
            self._ws[key] = torch.zeros((n,), dtype=torch.uint8, device=self.ctx.device)
        return self._ws[key]

    def DrawSceneToShadowMap(self, pass_cb, shadow_plane, depth_bias=10000, slope_bias=2.0):  # CRYCHIC.cpp:2477-2510, 1601-1603
        dim = int(shadow_plane.shape[0])
        ws = self.workspace(dim, dim)
        check(lib.crychic_draw_scene_to_shadow_map(self.ctx.handle, C.byref(pass_cb), self.items, len(self.items), _ptr(shadow_plane), dim,
                                                   int(depth_bias), float(slope_bias), _ptr(ws), ws.numel(), _stream(self.ctx.device)))

    def DrawSceneToShadowMaps(self, pass_cbs, shadow_planes, depth_bias=10000, slope_bias=2.0):
        """All cascades in one rasteriser pass (crychic_draw_scene_to_shadow_maps); bit-identical to one call per cascade."""
        n = len(pass_cbs)
        dim = int(shadow_planes[0].shape[0])
        ws = self.workspace(dim, dim, n)
        cbs = (type(pass_cbs[0]) * n)(*pass_cbs)
        ptrs = (C.c_void_p * n)(*[p.data_ptr() for p in shadow_planes])
        check(lib.crychic_draw_scene_to_shadow_maps(self.ctx.handle, C.cast(cbs, C.c_void_p), n, self.items, len(self.items), C.cast(ptrs, C.c_void_p), dim,
                                                    int(depth_bias), float(slope_bias), _ptr(ws), ws.numel(), _stream(self.ctx.device)))

    def DrawNormalsAndDepth(self, pass_cb, normal_map, depth):  # CRYCHIC.cpp:2512-2543
        H, W = int(depth.shape[0]), int(depth.shape[1])
        ws = self.workspace(W, H)
        check(lib.crychic_draw_normals_and_depth(self.ctx.handle, C.byref(pass_cb), self.items, len(self.items), _ptr(normal_map), _ptr(depth),
                                                 W, H, _ptr(ws), ws.numel(), _stream(self.ctx.device)))

    def DrawNormalsDepthAndGBuffer(self, pass_cb, normal, gbuffer, depth, g_rows=None):
        """DrawNormalsAndDepth + DrawGBuffer on one rasterisation (same items, same ViewProj => same visibility); every plane is
        bit-identical to the two separate passes.  g_rows = (row0, rows): a rank's strip -- depth and normals for the whole
        frame, G0..G2 for those rows only (crychic_draw_normals_depth_and_gbuffer_rows)."""
        H, W = int(depth.shape[0]), int(depth.shape[1])
        ws = self.workspace(W, H)
        if g_rows is not None:
            check(lib.crychic_draw_normals_depth_and_gbuffer_rows(self.ctx.handle, C.byref(pass_cb), self.items, len(self.items), _ptr(self.materials),
                                                                  self.n_materials, self.textures, self.n_textures, _ptr(normal), _ptr(gbuffer[0]),
                                                                  _ptr(gbuffer[1]), _ptr(gbuffer[2]), _ptr(depth), W, H, int(g_rows[0]), int(g_rows[1]),
                                                                  _ptr(ws), ws.numel(), _stream(self.ctx.device)))
            return
        check(lib.crychic_draw_normals_depth_and_gbuffer(self.ctx.handle, C.byref(pass_cb), self.items, len(self.items), _ptr(self.materials),
                                                         self.n_materials, self.textures, self.n_textures, _ptr(normal), _ptr(gbuffer[0]),
                                                         _ptr(gbuffer[1]), _ptr(gbuffer[2]), _ptr(depth), W, H, _ptr(ws), ws.numel(),
                                                         _stream(self.ctx.device)))

    def DrawGBuffer(self, pass_cb, gbuffer, depth, g_rows=None):  # CRYCHIC.cpp:2545-2571; g_rows = (row0, rows): scissored to a strip
        H, W = int(depth.shape[0]), int(depth.shape[1])
        ws = self.workspace(W, H)
        if g_rows is not None:
            check(lib.crychic_draw_gbuffer_rows(self.ctx.handle, C.byref(pass_cb), self.items, len(self.items), _ptr(self.materials), self.n_materials,
                                                self.textures, self.n_textures, _ptr(gbuffer[0]), _ptr(gbuffer[1]), _ptr(gbuffer[2]), _ptr(depth), W, H,
                                                int(g_rows[0]), int(g_rows[1]), _ptr(ws), ws.numel(), _stream(self.ctx.device)))
            return
        check(lib.crychic_draw_gbuffer(self.ctx.handle, C.byref(pass_cb), self.items, len(self.items), _ptr(self.materials), self.n_materials,
                                       self.textures, self.n_textures, _ptr(gbuffer[0]), _ptr(gbuffer[1]), _ptr(gbuffer[2]), _ptr(depth), W, H,
                                       _ptr(ws), ws.numel(), _stream(self.ctx.device)))
