"""Scene assembly shared by the producer-pass tests: constants, draw items and the end-to-end oracle frame."""
import numpy as np

import oracle_lib


def frame_constants(W, H, shadow_dim):
    from crychic_renderer_amd import scene
    return scene.Constants(W, H, shadow_dim)


def light_viewproj_t(consts, k):
    """Transposed ViewProj of cascade k as UpdateShadowPassCB stores it (CRYCHIC.cpp:877,889)."""
    return (consts.light_view[k].astype(np.float32) @ consts.light_proj[k].astype(np.float32)).T.reshape(-1).copy()


def oracle_frame(orc, consts, items, shadow_items, materials, textures, W, H, shadow_dim, cube, blur_count, num_dir_lights,
                 pcf_radius, sky=True, cube_dim=None, cube_levels=0):
    """The whole reference frame on the CPU: 4 shadow cascades, normals+depth, G-buffer, ComputeSsao, lighting.  cube_levels > 1:
    `cube` is a flat mip chain of a cube map with cube_dim-texel faces."""
    view = np.array(consts.pass_cb.View, np.float32)
    vp = np.array(consts.pass_cb.ViewProj, np.float32)
    mats = materials.view(oracle_lib.MATERIAL_DT) if materials is not None else None
    shadow = np.stack([oracle_lib.rasterize(orc, 0, view, light_viewproj_t(consts, k), shadow_items, None, None, shadow_dim,
                                            shadow_dim, 10000, 2.0)["depth"] for k in range(4)])
    nd = oracle_lib.rasterize(orc, 1, view, vp, items, mats, textures, W, H)
    gb = oracle_lib.rasterize(orc, 2, view, vp, items, mats, textures, W, H)
    scb = oracle_lib.as_oracle_cb(consts.ssao_cb, oracle_lib.OrSsaoConstants)
    pcb = oracle_lib.as_oracle_cb(consts.pass_cb, oracle_lib.OrPassConstants)
    ao = orc.compute_ssao(scb, nd["normal"], nd["depth"], consts.randvec, blur_count) if blur_count >= 0 else None
    rgba = orc.deferred_light(pcb, gb["g0"], gb["g1"], gb["g2"], nd["depth"], ao, shadow, cube, num_dir_lights, pcf_radius, sky=sky,
                              cube_dim=cube_dim, cube_levels=cube_levels)
    return {"shadow": shadow, "normal": nd["normal"], "depth": nd["depth"], "g0": gb["g0"], "g1": gb["g1"], "g2": gb["g2"], "ao": ao,
            "rgba8": rgba, "tris": nd["tris"]}


def c1_items(mesh_v, mesh_i):
    """BASELINE configs[0]: the skull at scale 0.4, translate (0, 1, 0) (CRYCHIC.cpp:1913) with skullMat (index 3)."""
    from crychic_renderer_amd import geometry as g
    inst = g.make_instances([g.world_matrix((0.4, 0.4, 0.4), (0.0, 1.0, 0.0))], [3])
    return [(mesh_v, mesh_i, inst)]


def render_c1(orc, skull_path, W=256, H=256, shadow_dim=512):
    from crychic_renderer_amd import geometry as g, scene
    import torch
    consts = frame_constants(W, H, shadow_dim)
    v, idx = oracle_lib.load_mesh_text(orc, skull_path)
    items = c1_items(v, idx)
    cube = scene.make_cubemap(32, torch.device("cpu")).numpy()
    fr = oracle_frame(orc, consts, items, items, g.reference_materials(), None, W, H, shadow_dim, cube, -1, 1, 0.0, sky=False)
    fr["covered"] = float((fr["depth"] < 0xFFFFFF).mean())
    return fr
