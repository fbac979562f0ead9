"""python -m crychic_renderer_amd.demo [--size WxH] [--out frame.ppm] [--textures DIR] [--cube FILE.dds]

Renders one frame of the reference's live scene entirely on the GPU -- 4 shadow cascades, view normals + depth, G-buffer
(HIP rasteriser), SSAO + blur, deferred lighting + sky -- and writes it as PPM (the headless stand-in for Present)."""
import argparse
import ctypes as C

import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1280x720")
    ap.add_argument("--out", default="frame.ppm")
    ap.add_argument("--shadow-dim", type=int, default=2048)
    ap.add_argument("--textures", default="", help="directory with the reference's DDS textures (else procedural stand-ins)")
    ap.add_argument("--cube", default="", help="a DDS cube map for the sky and the reflections, e.g. the reference's Textures/snowcube1024.dds "
                                               "(else a procedural one)")
    a = ap.parse_args()
    W, H = (int(v) for v in a.size.lower().split("x"))
    import torch
    from . import Context, Crychic, LIGHT_SKY, SceneGeometry, check, geometry as g, lib, scene
    from ._lib import PassConstants
    ctx = Context(0)
    consts = scene.Constants(W, H, a.shadow_dim)
    tex = g.reference_textures(a.textures) if a.textures else g.procedural_textures(64)
    geo = SceneGeometry(ctx, g.cascade_scene_items(), g.reference_materials(), tex)
    sgeo = SceneGeometry(ctx, g.cascade_scene_items(shadow_layer=True))
    cube = scene.make_cubemap(256, ctx.device)
    app = Crychic(ctx, W, H, torch.from_numpy(consts.randvec.copy()).to(ctx.device), cube, shadow_dim=a.shadow_dim)
    if a.cube:            # with the mip chain the file stores, as the reference binds it (CRYCHIC.cpp:1148-1151)
        chain, dim, levels = g.load_dds_cube_mips(a.cube)
        app.set_cube_map(torch.from_numpy(chain).to(ctx.device), dim=dim, levels=levels)
    app.mMainPassCB, app.mSsaoCB = consts.pass_cb, consts.ssao_cb
    for k in range(4):
        cb = PassConstants()
        cb.ViewProj[:] = list((consts.light_view[k].astype(np.float32) @ consts.light_proj[k].astype(np.float32)).T.reshape(-1))
        sgeo.DrawSceneToShadowMap(cb, app.mShadowMap.mShadowMap[k])
    geo.DrawNormalsAndDepth(app.mMainPassCB, app.mSsao.mNormalMap, app.mDepthStencilBuffer)
    geo.DrawGBuffer(app.mMainPassCB, app.mDeferred.mGBuffer, app.mDepthStencilBuffer)
    app.blurCount, app.numDirLights, app.flags = 3, 1, LIGHT_SKY        # the reference's settings (CRYCHIC.cpp:221, Common.hlsl:6-8)
    app.Draw()
    torch.cuda.synchronize()
    img = np.ascontiguousarray(app.mBackBuffer.cpu().numpy())
    check(lib.crychic_save_ppm(a.out.encode(), img.ctypes.data, W, H))
    print("wrote %s (%dx%d) on %s" % (a.out, W, H, ctx.device_name))


if __name__ == "__main__":
    main()
