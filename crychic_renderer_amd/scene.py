"""Synthetic input planes for the CRYCHIC hot path (SURVEY.md 8d "Synthetic inputs").

The reference fills its G-buffer, normal/depth and shadow-map targets through the D3D12 rasteriser
(GeometryPass.hlsl, DrawNormals.hlsl, Shadows.hlsl).  For benchmarks and parity tests the same planes are
produced analytically here: the live scene of CRYCHIC::BuildCascadeShadowRenderItems (CRYCHIC.cpp:2274-2378) --
a 60 x 90 ground grid at y = 0 and 10 x 10 boxes of side 1.6 centred at (5i-25, 0.8, 5j-25) -- is ray-cast per
pixel (camera planes) and per shadow texel (orthographic cascades, with the shadow PSO's depth bias,
CRYCHIC.cpp:1601-1603).  Everything is torch, so the 4K / 4 x 4096^2 case is generated on the GPU in seconds and
the small test cases on the CPU.  Constant buffers come from the library's host builders (csrc/host_constants.cpp).
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from ._lib import Camera, PassConstants, SsaoConstants, lib

BASE_LIGHT_DIRS = ((0.57735, -0.57735, 0.57735), (-0.57735, -0.57735, 0.57735), (0.0, -0.707, -0.707))  # CRYCHIC.h:173-177
GROUND_HALF_X, GROUND_HALF_Z = 30.0, 45.0   # CreateGrid(20, 30) scaled by 3 (CRYCHIC.cpp:1254, 2371)
BOX_HALF = 0.8                              # CreateBox(1,1,1) scaled by 1.6 (CRYCHIC.cpp:1253, 2342)


def default_camera(W, H):
    """Camera of CRYCHIC::Initialize / OnResize (CRYCHIC.cpp:46,114)."""
    cam = Camera()
    cam.pos[:] = (0.0, 2.0, -15.0)
    cam.look[:] = (0.0, 0.0, 1.0)
    cam.up[:] = (0.0, 1.0, 0.0)
    cam.fovY = 0.25 * math.pi
    cam.aspect = float(W) / float(H)
    cam.nearZ = 1.0
    cam.farZ = 100.0
    return cam


def covered_camera(W, H):
    """The benchmark's "covered" camera (bench.py --camera covered): the eye moved into the lane between two box columns and
    pitched down 25 degrees, so that the top edge of the view still meets the ground grid and nothing is nearer than the near
    plane -- every pixel is covered (no sky) and every G-buffer texel is read."""
    cam = default_camera(W, H)
    a = math.radians(25.0)
    cam.pos[:] = (2.5, 2.0, -15.0)
    cam.look[:] = (0.0, -math.sin(a), math.cos(a))
    cam.up[:] = (0.0, math.cos(a), math.sin(a))
    return cam


class Constants:
    """SsaoConstants + PassConstants + the light matrices of one frame (UpdateSsaoCB / UpdateMainPassCB /
    UpdateCascadeShadowTransform, CRYCHIC.cpp:634-937), built by the library's host code."""

    def __init__(self, W, H, shadow_dim=4096, cam=None):
        self.W, self.H, self.shadow_dim = W, H, shadow_dim
        self.cam = cam or default_camera(W, H)
        # The Ssao object is the first user of rand() in the process (CRYCHIC.cpp:51, Ssao.cpp:18-19).
        self.rand_state = C.c_uint32(1)
        self.offsets = ((C.c_float * 4) * 14)()
        lib.crychic_build_offset_vectors(C.byref(self.rand_state), self.offsets)
        self.randvec = np.zeros((256, 256, 4), dtype=np.uint8)
        lib.crychic_build_random_vector_texture(C.byref(self.rand_state), 1, self.randvec.ctypes.data)
        self.light_view = np.zeros((4, 4, 4), dtype=np.float32)
        self.light_proj = np.zeros((4, 4, 4), dtype=np.float32)
        self.shadow_transform = np.zeros((4, 4, 4), dtype=np.float32)
        ld = (C.c_float * 3)(*BASE_LIGHT_DIRS[0])
        _lib.check(lib.crychic_update_cascade_shadow_transform(
            C.byref(self.cam), ld, shadow_dim, self.light_view.ctypes.data, self.light_proj.ctypes.data,
            self.shadow_transform.ctypes.data))
        self.pass_cb = PassConstants()
        dirs = np.asarray(BASE_LIGHT_DIRS, dtype=np.float32)
        _lib.check(lib.crychic_update_main_pass_cb(C.byref(self.cam), W, H, self.shadow_transform.ctypes.data,
                                                   dirs.ctypes.data, C.byref(self.pass_cb)))
        self.ssao_cb = SsaoConstants()
        _lib.check(lib.crychic_update_ssao_cb(C.byref(self.cam), W, H, C.cast(self.offsets, C.c_void_p),
                                              C.byref(self.ssao_cb)))


def _box_centres(device, dtype):
    i = torch.arange(10, device=device, dtype=dtype)
    cx = ((-5 + i) * 5.0).repeat_interleave(10)       # instance i*10+j: x from i, z from j (CRYCHIC.cpp:2338-2346)
    cz = ((-5 + i) * 5.0).repeat(10)
    cy = torch.full_like(cx, 0.8)
    return torch.stack([cx, cy, cz], dim=1)            # (100, 3); material index = i % 2


def raycast(O, D, chunk=1 << 17):
    """Nearest hit of rays O + t*D (N x 3, float64) with the scene.  Returns t (inf = miss), unit normal,
    material id (0/1 = box columns, 3 = ground; -1 = miss)."""
    N = O.shape[0]
    dev, dt = O.device, O.dtype
    t_out = torch.full((N,), float("inf"), device=dev, dtype=dt)
    n_out = torch.zeros((N, 3), device=dev, dtype=dt)
    m_out = torch.full((N,), -1, device=dev, dtype=torch.int64)
    centres = _box_centres(dev, dt)
    bmin, bmax = centres - BOX_HALF, centres + BOX_HALF
    for s in range(0, N, chunk):
        o, d = O[s:s + chunk], D[s:s + chunk]
        n = o.shape[0]
        # ground y = 0
        dy = d[:, 1]
        tg = torch.where(dy != 0, -o[:, 1] / dy, torch.full_like(dy, float("inf")))
        px, pz = o[:, 0] + tg * d[:, 0], o[:, 2] + tg * d[:, 2]
        okg = (tg > 0) & (px.abs() <= GROUND_HALF_X) & (pz.abs() <= GROUND_HALF_Z) & torch.isfinite(tg)
        tg = torch.where(okg, tg, torch.full_like(tg, float("inf")))
        # boxes: slabs
        inv = 1.0 / d
        t0 = (bmin[None] - o[:, None]) * inv[:, None]
        t1 = (bmax[None] - o[:, None]) * inv[:, None]
        tlo, thi = torch.minimum(t0, t1), torch.maximum(t0, t1)
        tnear, axis = tlo.max(dim=2)
        tfar = thi.min(dim=2).values
        hit = (tfar >= tnear) & (tnear > 0)
        tb = torch.where(hit, tnear, torch.full_like(tnear, float("inf")))
        tbest, ibox = tb.min(dim=1)
        ax = axis.gather(1, ibox[:, None])[:, 0]
        nb = torch.zeros((n, 3), device=dev, dtype=dt)
        sgn = -torch.sign(d.gather(1, ax[:, None])[:, 0])
        nb.scatter_(1, ax[:, None], sgn[:, None])
        use_box = tbest < tg
        t = torch.where(use_box, tbest, tg)
        ng = torch.zeros((n, 3), device=dev, dtype=dt)
        ng[:, 1] = torch.where(o[:, 1] > 0, 1.0, -1.0).to(dt)
        nrm = torch.where(use_box[:, None], nb, ng)
        mat = torch.where(use_box, (ibox // 10) % 2, torch.full_like(ibox, 3))
        miss = ~torch.isfinite(t)
        mat = torch.where(miss, torch.full_like(mat, -1), mat)
        nrm = torch.where(miss[:, None], torch.zeros_like(nrm), nrm)
        t_out[s:s + n], n_out[s:s + n], m_out[s:s + n] = t, nrm, mat
    return t_out, n_out, m_out


def _camera_basis(cam, dev, dt):
    L = torch.tensor(list(cam.look), device=dev, dtype=dt)
    L = L / L.norm()
    up = torch.tensor(list(cam.up), device=dev, dtype=dt)
    R = torch.linalg.cross(up, L)
    R = R / R.norm()
    U = torch.linalg.cross(L, R)
    return R, U, L


def _camera_planes(consts, device):
    W, H, cam = consts.W, consts.H, consts.cam
    dt = torch.float64
    eye = torch.tensor(list(cam.pos), device=device, dtype=dt)
    R, U, L = _camera_basis(cam, device, dt)
    th = math.tan(0.5 * cam.fovY)
    xs = (torch.arange(W, device=device, dtype=dt) + 0.5) / W * 2.0 - 1.0
    ys = 1.0 - (torch.arange(H, device=device, dtype=dt) + 0.5) / H * 2.0
    vx = (xs * cam.aspect * th)[None, :].expand(H, W).reshape(-1)
    vy = (ys * th)[:, None].expand(H, W).reshape(-1)
    D = vx[:, None] * R[None] + vy[:, None] * U[None] + L[None]
    O = eye[None].expand_as(D).contiguous()
    t, nrm, mat = raycast(O, D)
    hit = torch.isfinite(t)
    tt = torch.where(hit, t, torch.zeros_like(t))
    pos = O + tt[:, None] * D
    zview = tt                                    # D has unit component along L
    A = cam.farZ / (cam.farZ - cam.nearZ)
    B = -cam.nearZ * cam.farZ / (cam.farZ - cam.nearZ)
    zndc = A + B / torch.where(hit, zview, torch.ones_like(zview))
    d24 = torch.round(zndc.clamp(0.0, 1.0) * 16777215.0).to(torch.int64)
    d24 = torch.where(hit & (zview >= cam.nearZ) & (zview <= cam.farZ), d24, torch.full_like(d24, 0xFFFFFF))
    covered = d24 < 0xFFFFFF

    nview = torch.stack([nrm @ R, nrm @ U, nrm @ L], dim=1)
    normal = torch.zeros((H * W, 4), device=device, dtype=torch.float32)
    normal[:, 2] = 1.0                             # clear (0,0,1,0), Ssao.cpp:317
    normal[covered, :3] = nview[covered].to(torch.float32)
    normal = normal.to(torch.float16)

    # materials: bricks0 (1,1,1) rough .3 / tile0 (.9,.9,.9) rough .7 / skullMat (1,1,1) rough .8; metalness is never
    # assigned, so MaterialData's default 0.5 reaches every texel (CRYCHIC.cpp:1770-1805, FrameResource.h:25)
    base = torch.ones((H * W, 3), device=device, dtype=dt)
    base[mat == 1] = 0.9
    rough = torch.full((H * W,), 0.8, device=device, dtype=dt)
    rough[mat == 0] = 0.3
    rough[mat == 1] = 0.7
    # procedural stand-in for the diffuse / normal textures (bricks2.dds, tile.dds; CRYCHIC.cpp:954-959)
    tex = 0.75 + 0.25 * torch.sin(3.1 * pos[:, 0] + 1.7 * pos[:, 1]) * torch.cos(2.3 * pos[:, 2] - 0.9 * pos[:, 1])
    tint = torch.stack([tex, 0.9 * tex + 0.1, 0.8 * tex + 0.2], dim=1)
    albedo = base * tint
    bump = 0.12 * torch.stack([torch.sin(9.0 * pos[:, 0] + 4.0 * pos[:, 2]), torch.sin(7.0 * pos[:, 1]),
                               torch.cos(8.0 * pos[:, 2] - 3.0 * pos[:, 0])], dim=1)
    nW = nrm + bump * (1.0 - nrm.abs())

    g0 = torch.zeros((H * W, 4), device=device, dtype=torch.float32)   # cleared to black (CRYCHIC.cpp:2554)
    g1 = torch.zeros_like(g0)
    g2 = torch.zeros_like(g0)
    g0[covered, :3] = pos[covered].to(torch.float32)
    g0[covered, 3] = 0.5
    g1[covered, :3] = albedo[covered].to(torch.float32)
    g1[covered, 3] = rough[covered].to(torch.float32)
    g2[covered, :3] = nW[covered].to(torch.float32)
    g2[covered, 3] = 1.0
    return {
        "depth": d24.to(torch.int32).reshape(H, W),
        "normal": normal.reshape(H, W, 4),
        "g0": g0.reshape(H, W, 4), "g1": g1.reshape(H, W, 4), "g2": g2.reshape(H, W, 4),
    }


def _shadow_cascade(consts, k, device):
    dim = consts.shadow_dim
    dt = torch.float64
    V = torch.tensor(consts.light_view[k], device=device, dtype=dt)     # row-vector matrices
    P = torch.tensor(consts.light_proj[k], device=device, dtype=dt)
    Vinv = torch.linalg.inv(V)
    Pinv = torch.linalg.inv(P)
    ld = torch.tensor(BASE_LIGHT_DIRS[0], device=device, dtype=dt)
    ld = ld / ld.norm()
    out = torch.empty((dim, dim), device=device, dtype=torch.int32)
    rows_per = max(1, (1 << 19) // dim)
    xs = (torch.arange(dim, device=device, dtype=dt) + 0.5) / dim * 2.0 - 1.0
    bias_const = 10000.0 / 16777216.0                                    # DepthBias = 10000 on D24
    texel_view = 2.0 / (P[0, 0] * dim)
    for r0 in range(0, dim, rows_per):
        r1 = min(dim, r0 + rows_per)
        ys = 1.0 - (torch.arange(r0, r1, device=device, dtype=dt) + 0.5) / dim * 2.0
        n = (r1 - r0) * dim
        ndc = torch.zeros((n, 4), device=device, dtype=dt)
        ndc[:, 0] = xs[None, :].expand(r1 - r0, dim).reshape(-1)
        ndc[:, 1] = ys[:, None].expand(r1 - r0, dim).reshape(-1)
        ndc[:, 3] = 1.0
        ow = (ndc @ Pinv) @ Vinv
        O = ow[:, :3].contiguous()
        D = ld[None].expand(n, 3).contiguous()
        t, nrm, _ = raycast(O, D)
        hit = torch.isfinite(t)
        depth = torch.where(hit, t, torch.zeros_like(t)) * P[2, 2]
        nl = nrm @ V[:3, :3]                                            # normal in light view space
        nz = torch.where(nl[:, 2].abs() > 1e-6, nl[:, 2].abs(), torch.full_like(nl[:, 2], 1e-6))
        slope = torch.maximum(nl[:, 0].abs(), nl[:, 1].abs()) / nz * texel_view * P[2, 2]
        depth = depth + bias_const + 2.0 * slope                        # SlopeScaledDepthBias = 2.0
        d24 = torch.round(depth.clamp(0.0, 1.0) * 16777215.0).to(torch.int64)
        d24 = torch.where(hit, d24, torch.full_like(d24, 0xFFFFFF))
        out[r0:r1] = d24.to(torch.int32).reshape(r1 - r0, dim)
    return out


def make_cubemap(dim, device):
    """Procedural stand-in for Textures/snowcube1024.dds (absent from the checkout): sky gradient + sun glow."""
    dt = torch.float32
    u = (torch.arange(dim, device=device, dtype=dt) + 0.5) / dim * 2.0 - 1.0
    sc = u[None, :].expand(dim, dim)
    tc = u[:, None].expand(dim, dim)
    one = torch.ones_like(sc)
    dirs = [(one, -tc, -sc), (-one, -tc, sc), (sc, one, tc), (sc, -one, -tc), (sc, -tc, one), (-sc, -tc, -one)]
    sun = torch.tensor([-d for d in BASE_LIGHT_DIRS[0]], device=device, dtype=dt)
    sun = sun / sun.norm()
    faces = []
    for dx, dy, dz in dirs:
        d = torch.stack([dx, dy, dz], dim=-1)
        d = d / d.norm(dim=-1, keepdim=True)
        up = d[..., 1].clamp(0.0, 1.0)[..., None]
        horizon = torch.tensor([0.8, 0.85, 0.9], device=device, dtype=dt)
        zenith = torch.tensor([0.2, 0.4, 0.8], device=device, dtype=dt)
        col = horizon + (zenith - horizon) * up
        ground = torch.tensor([0.35, 0.33, 0.30], device=device, dtype=dt)
        col = torch.where(d[..., 1:2] < 0, ground + (horizon - ground) * (1.0 + d[..., 1:2]).clamp(0, 1) ** 8, col)
        glow = (d @ sun).clamp(0.0, 1.0)[..., None] ** 64
        col = (col + glow * torch.tensor([1.0, 0.9, 0.7], device=device, dtype=dt)).clamp(0.0, 1.0)
        rgba = torch.cat([col, torch.ones_like(col[..., :1])], dim=-1)
        faces.append(torch.round(rgba * 255.0).to(torch.uint8))
    return torch.stack(faces, dim=0).contiguous()


def make_scene(W, H, shadow_dim=4096, cube_dim=256, device="cpu", consts=None):
    """All input planes of one frame plus the output / workspace planes, as torch tensors on `device`."""
    consts = consts or Constants(W, H, shadow_dim)
    dev = torch.device(device)
    planes = _camera_planes(consts, dev)
    planes["shadow"] = torch.stack([_shadow_cascade(consts, k, dev) for k in range(4)], dim=0).contiguous()
    planes["cube"] = make_cubemap(cube_dim, dev)
    planes["randvec"] = torch.from_numpy(consts.randvec.copy()).to(dev)
    w2, h2 = W // 2, H // 2
    planes["ambient0"] = torch.zeros((h2, w2), device=dev, dtype=torch.int16)
    planes["ambient1"] = torch.zeros((h2, w2), device=dev, dtype=torch.int16)
    planes["edge"] = torch.zeros((int(lib.crychic_edge_plane_bytes(W, H)),), device=dev, dtype=torch.uint8)
    planes["out"] = torch.zeros((H, W, 4), device=dev, dtype=torch.uint8)
    planes["consts"] = consts
    return planes


def point_light_grid(n=8, y=3.0, falloff_start=1.0, falloff_end=10.0, strength=1.0, extent=24.5):
    """BASELINE configs[4] lights: n x n point lights on a regular grid at height y over the box field,
    linear falloff start..end, white strength -- seed-free.  Returns a ctypes array of Light."""
    from ._lib import Light
    arr = (Light * (n * n))()
    for i in range(n):
        for j in range(n):
            L = arr[i * n + j]
            L.Strength[:] = (strength, strength, strength)
            L.FalloffStart, L.FalloffEnd = falloff_start, falloff_end
            L.Direction[:] = (0.0, -1.0, 0.0)
            L.Position[:] = (-extent + 2.0 * extent * i / (n - 1), y, -extent + 2.0 * extent * j / (n - 1))
            L.SpotPower = 64.0
    return arr
