"""Meshes, instances and materials of the reference's live scene for the producer passes.

Host-side data comes from the library's geometry functions (csrc/host_geometry.cpp); the layouts are the
reference's structured-buffer ABI (FrameResource.h:7-27,69-75)."""
import ctypes as C

import numpy as np

from ._lib import check, lib

VERTEX_DT = np.dtype([("Pos", "<f4", 3), ("Normal", "<f4", 3), ("TexC", "<f4", 2), ("TangentU", "<f4", 3)])
INSTANCE_DT = np.dtype([("World", "<f4", 16), ("TexTransform", "<f4", 16), ("MaterialIndex", "<u4"), ("pad", "<u4", 3)])
MATERIAL_DT = np.dtype([("DiffuseAlbedo", "<f4", 4), ("FresnelR0", "<f4", 3), ("Roughness", "<f4"), ("MatTransform", "<f4", 16),
                        ("DiffuseMapIndex", "<u4"), ("NormalMapIndex", "<u4"), ("Metalness", "<f4"), ("pad", "<u4")])
assert VERTEX_DT.itemsize == 44 and INSTANCE_DT.itemsize == 144 and MATERIAL_DT.itemsize == 112

IDENTITY = np.eye(4, dtype=np.float32).reshape(-1)


def _two_pass(fn, *args):
    ni = C.c_uint32()
    nv = check(fn(*args, None, 0, None, 0, C.byref(ni)))
    v = np.zeros(nv, VERTEX_DT)
    idx = np.zeros(ni.value, np.uint32)
    check(fn(*args, v.ctypes.data, nv, idx.ctypes.data, ni.value, C.byref(ni)))
    return v, idx


def create_box(width, height, depth, num_subdivisions):
    """GeometryGenerator::CreateBox (Common/GeometryGenerator.cpp:10-101)."""
    return _two_pass(lib.crychic_create_box, float(width), float(height), float(depth), int(num_subdivisions))


def create_grid(width, depth, m, n):
    """GeometryGenerator::CreateGrid (Common/GeometryGenerator.cpp:551-614)."""
    return _two_pass(lib.crychic_create_grid, float(width), float(depth), int(m), int(n))


def load_mesh_text(path):
    """Models/*.txt as CRYCHIC::BuildSkullGeometry reads them (CRYCHIC.cpp:1447-1557)."""
    nv, ni = C.c_uint32(), C.c_uint32()
    check(lib.crychic_load_mesh_text(path.encode(), None, 0, None, 0, C.byref(nv), C.byref(ni)))
    v = np.zeros(nv.value, VERTEX_DT)
    idx = np.zeros(ni.value, np.uint32)
    check(lib.crychic_load_mesh_text(path.encode(), v.ctypes.data, nv.value, idx.ctypes.data, ni.value, C.byref(nv), C.byref(ni)))
    return v, idx


def world_matrix(scale=(1, 1, 1), translate=(0, 0, 0)):
    """XMMatrixScaling(s) * XMMatrixTranslation(t), stored transposed like UpdateInstanceData does (CRYCHIC.cpp:546)."""
    m = np.diag([scale[0], scale[1], scale[2], 1.0]).astype(np.float32)
    m[3, :3] = translate
    return m.T.reshape(-1).copy()


def make_instances(worlds, material_indices):
    inst = np.zeros(len(worlds), INSTANCE_DT)
    for k, (w, mi) in enumerate(zip(worlds, material_indices)):
        inst[k]["World"] = w
        inst[k]["TexTransform"] = IDENTITY
        inst[k]["MaterialIndex"] = mi
    return inst


def mesh_bounds(vertices):
    """BoundingBox of a mesh as the reference builds it (min / max of the positions -> center, extents; CRYCHIC.cpp:1318-1337)."""
    pos = vertices["Pos"]
    lo, hi = pos.min(axis=0), pos.max(axis=0)
    return (0.5 * (lo + hi)).astype(np.float32), (0.5 * (hi - lo)).astype(np.float32)


def frustum_cull(cam, vertices, instances):
    """CRYCHIC::UpdateInstanceData (CRYCHIC.cpp:515-564, culling enabled as in CRYCHIC.h:188): the instances whose
    bounding box is not DISJOINT from the camera frustum, in their original order (crychic_frustum_cull)."""
    if len(instances) == 0:
        return instances
    center, extents = mesh_bounds(vertices)
    # the instance records hold World transposed (GPU layout); the test wants the row-vector matrix back
    worlds = np.ascontiguousarray(instances["World"].reshape(len(instances), 4, 4).transpose(0, 2, 1), dtype=np.float32)
    vis = np.zeros(len(instances), np.uint8)
    n = lib.crychic_frustum_cull(C.byref(cam), center.ctypes.data_as(C.POINTER(C.c_float)), extents.ctypes.data_as(C.POINTER(C.c_float)),
                                 worlds.ctypes.data, len(instances), vis.ctypes.data)
    check(min(n, 0))
    return instances[vis.astype(bool)]


def reference_materials():
    """CRYCHIC::BuildMaterials (CRYCHIC.cpp:1768-1821); Metalness is never assigned, so the MaterialData default 0.5
    reaches the GPU (FrameResource.h:25, CRYCHIC.cpp:578-586)."""
    rows = [((1.0, 1.0, 1.0, 1.0), (0.1, 0.1, 0.1), 0.3, 0, 1),      # bricks0
            ((0.9, 0.9, 0.9, 1.0), (0.2, 0.2, 0.2), 0.7, 2, 3),      # tile0
            ((0.0, 0.0, 0.0, 1.0), (0.98, 0.97, 0.95), 0.1, 4, 5),   # mirror0
            ((1.0, 1.0, 1.0, 1.0), (0.6, 0.6, 0.6), 0.8, 4, 5),      # skullMat
            ((1.0, 1.0, 1.0, 1.0), (0.1, 0.1, 0.1), 1.0, 6, 7)]      # sky
    m = np.zeros(len(rows), MATERIAL_DT)
    for k, (alb, r0, rough, dmap, nmap) in enumerate(rows):
        m[k]["DiffuseAlbedo"] = alb
        m[k]["FresnelR0"] = r0
        m[k]["Roughness"] = rough
        m[k]["MatTransform"] = IDENTITY
        m[k]["DiffuseMapIndex"] = dmap
        m[k]["NormalMapIndex"] = nmap
        m[k]["Metalness"] = 0.5
    return m


def cascade_scene_items(shadow_layer=False, cull_camera=None):
    """The Opaque (or OpaqueShadow) render items of CRYCHIC::BuildCascadeShadowRenderItems[WithShadow]
    (CRYCHIC.cpp:2322-2375, 2380-2435): 100 instanced boxes + the grid.  Returns [(vertices, indices, instances), ...].
    With `cull_camera` the instance lists are what UpdateInstanceData uploads for that camera (frustum culling on)."""
    box = create_box(1.0, 1.0, 1.0, 3)              # CRYCHIC.cpp:1253
    grid = create_grid(20.0, 30.0, 60, 40)          # CRYCHIC.cpp:1254
    worlds, mats = [], []
    for i in range(10):
        for j in range(10):
            worlds.append(world_matrix((1.6, 1.6, 1.6), ((-5 + i) * 5.0, 0.8, (-5 + j) * 5.0)))
            mats.append(i % 3 if shadow_layer else i % 2)
    box_inst = make_instances(worlds, mats)
    grid_inst = make_instances([world_matrix((3.0, 3.0, 3.0))], [1 if shadow_layer else 3])
    if cull_camera is not None:
        box_inst = frustum_cull(cull_camera, box[0], box_inst)
        grid_inst = frustum_cull(cull_camera, grid[0], grid_inst)
    return [(box[0], box[1], box_inst), (grid[0], grid[1], grid_inst)]


def procedural_textures(size=64):
    """Stand-ins for the DDS material textures (row f4 decodes the real ones): index 0/2/4/6 diffuse, 1/3/5/7 normal maps
    (the heap order of CRYCHIC::LoadTextures, CRYCHIC.cpp:954-959)."""
    y, x = np.mgrid[0:size, 0:size]
    out = []
    for k in range(8):
        t = np.zeros((size, size, 4), np.uint8)
        if k % 2 == 0:   # diffuse: brick / tile pattern
            mortar = ((x % 16) < 1) | ((y % 8) < 1) if k == 0 else ((x % 32) < 2) | ((y % 32) < 2)
            base = np.array([[200, 120, 90], [230, 230, 235], [255, 255, 255], [255, 255, 255]][k // 2], np.uint8)
            t[..., :3] = base
            t[mortar, :3] = (base * 0.6).astype(np.uint8)
            t[..., 3] = 255
        else:            # normal map: gentle bumps around (0.5, 0.5, 1)
            nx = 128 + (20 * np.sin(2 * np.pi * x / 16.0)).astype(np.int32)
            ny = 128 + (20 * np.cos(2 * np.pi * y / 8.0)).astype(np.int32)
            t[..., 0] = nx; t[..., 1] = ny; t[..., 2] = 250; t[..., 3] = 255
        out.append(t)
    return out


def box_mips(level0):
    """A mip chain for an H x W x 4 uint8 image by 2 x 2 box filtering with round-to-nearest, down to 1 x 1 (the reference's DDS
    files carry their own chains; this is for procedural test textures).  Returns the list of levels."""
    levels = [np.ascontiguousarray(level0)]
    while levels[-1].shape[0] > 1 or levels[-1].shape[1] > 1:
        a = levels[-1].astype(np.uint32)
        h, w = a.shape[0], a.shape[1]
        a = a[: max(1, h - h % 2) if h > 1 else 1, : max(1, w - w % 2) if w > 1 else 1]
        if h > 1:
            a = a[0::2] + a[1::2]
        if w > 1:
            a = a[:, 0::2] + a[:, 1::2]
        div = (2 if h > 1 else 1) * (2 if w > 1 else 1)
        levels.append(((a + div // 2) // div).astype(np.uint8))
    return levels


def cube_mip_chain(cube, levels=None):
    """The mip chain of a 6 x dim x dim x 4 uint8 cube map by 2 x 2 box filtering of each face (box_mips), flattened into the layout
    CRYCHIC_LIGHT_CUBE_LEVELS announces: level after level, each level the six faces of max(dim >> level, 1)^2 texels.  Returns (flat
    uint8 array, number of levels); `levels` caps the chain (default: down to 1 x 1)."""
    faces = [box_mips(cube[f]) for f in range(6)]
    n = len(faces[0]) if levels is None else min(int(levels), len(faces[0]))
    return np.concatenate([faces[f][k].reshape(-1) for k in range(n) for f in range(6)]), n


def texture_levels(t):
    """(flat uint8 array of all levels back to back, width, height, mipLevels) for a texture given as one H x W x 4 array or as a
    list of level arrays (level k = max(1, W >> k) x max(1, H >> k))."""
    if isinstance(t, (list, tuple)):
        w, h = t[0].shape[1], t[0].shape[0]
        for k, lv in enumerate(t):
            assert lv.shape[:2] == (max(1, h >> k), max(1, w >> k)) and lv.dtype == np.uint8, (k, lv.shape)
        return np.concatenate([np.ascontiguousarray(lv).reshape(-1) for lv in t]), w, h, len(t)
    t = np.ascontiguousarray(t)
    return t.reshape(-1), t.shape[1], t.shape[0], 1


def load_dds_mips(path):
    """A reference material texture with the mip chain its file stores, as a list of level arrays (crychic_load_dds_rgba8_mips)."""
    w, h, n = C.c_uint32(), C.c_uint32(), C.c_uint32()
    check(lib.crychic_load_dds_rgba8_mips(path.encode(), None, 0, C.byref(w), C.byref(h), C.byref(n)))
    sizes = [(max(1, h.value >> k), max(1, w.value >> k)) for k in range(n.value)]
    flat = np.zeros(sum(a * b * 4 for a, b in sizes), np.uint8)
    check(lib.crychic_load_dds_rgba8_mips(path.encode(), flat.ctypes.data, flat.size, C.byref(w), C.byref(h), C.byref(n)))
    out, off = [], 0
    for a, b in sizes:
        out.append(flat[off:off + a * b * 4].reshape(a, b, 4)); off += a * b * 4
    return out


def load_dds(path):
    """A reference material texture as an H x W x 4 uint8 array (crychic_load_dds_rgba8, row f4)."""
    w, h = C.c_uint32(), C.c_uint32()
    check(lib.crychic_load_dds_rgba8(path.encode(), None, 0, C.byref(w), C.byref(h)))
    out = np.zeros((h.value, w.value, 4), np.uint8)
    check(lib.crychic_load_dds_rgba8(path.encode(), out.ctypes.data, out.nbytes, C.byref(w), C.byref(h)))
    return out


def load_dds_cube(path):
    """A DDS cube map as the 6 x dim x dim x 4 uint8 plane the lighting pass takes (crychic_load_dds_cube_rgba8): level 0 of the
    faces +X, -X, +Y, -Y, +Z, -Z."""
    d = C.c_uint32()
    check(lib.crychic_load_dds_cube_rgba8(path.encode(), None, 0, C.byref(d)))
    out = np.zeros((6, d.value, d.value, 4), np.uint8)
    check(lib.crychic_load_dds_cube_rgba8(path.encode(), out.ctypes.data, out.nbytes, C.byref(d)))
    return out


def load_dds_cube_mips(path):
    """A DDS cube map with the mip chain the file stores (crychic_load_dds_cube_rgba8_mips): (flat uint8 array in the level-after-level
    layout of CRYCHIC_LIGHT_CUBE_LEVELS, dim, mipLevels)."""
    d, m = C.c_uint32(), C.c_uint32()
    check(lib.crychic_load_dds_cube_rgba8_mips(path.encode(), None, 0, C.byref(d), C.byref(m)))
    out = np.zeros(sum(6 * max(d.value >> k, 1) ** 2 * 4 for k in range(m.value)), np.uint8)
    check(lib.crychic_load_dds_cube_rgba8_mips(path.encode(), out.ctypes.data, out.nbytes, C.byref(d), C.byref(m)))
    return out, d.value, m.value


def reference_textures(texture_dir):
    """gTextureMaps in heap order (CRYCHIC::LoadTextures, CRYCHIC.cpp:954-959): bricks2, bricks2_nmap, tile, tile_nmap,
    white1x1, default_nmap."""
    import os
    names = ["bricks2.dds", "bricks2_nmap.dds", "tile.dds", "tile_nmap.dds", "white1x1.dds", "default_nmap.dds"]
    return [load_dds(os.path.join(texture_dir, n)) for n in names]
