"""Third opinion on the bilateral blur: an independent vectorised numpy restatement of SsaoBlur.hlsl:85-146 against
the scalar C oracle (SURVEY.md 8c iii).  float32 throughout, same accumulation order => bit-exact."""
import numpy as np

import oracle_lib
import scene_util


def numpy_blur(cb, normal_f16, depth_u32, amb_u16, horizontal):
    H, W = depth_u32.shape
    h2, w2 = H // 2, W // 2
    f32 = np.float32
    w = np.array([cb.BlurWeights[i // 4][i % 4] for i in range(12)], dtype=f32)
    A, B = f32(cb.Proj[10]), f32(cb.Proj[11])
    d = (depth_u32 & 0xFFFFFF).astype(f32) / f32(16777215.0)
    # centre depth: bilinear at the 2x2 corner = lerp(lerp(a,b,.5), lerp(c,d,.5), .5)
    t00, t10, t01, t11 = d[0::2, 0::2], d[0::2, 1::2], d[1::2, 0::2], d[1::2, 1::2]
    top = t00 + f32(0.5) * (t10 - t00); bot = t01 + f32(0.5) * (t11 - t01)
    zndc = top + f32(0.5) * (bot - top)
    vz = B / (zndc - A)
    border_vz = B / (f32(1.0) - A)
    nrm = normal_f16[1::2, 1::2, :3].astype(f32)            # texel (2x+1, 2y+1)
    gcol = normal_f16[1::2, 0, :3].astype(f32)              # clamp target for x + i < 0: texel (0, 2y+1)
    grow = normal_f16[0, 1::2, :3].astype(f32)              # clamp target for y + i < 0: texel (2x+1, 0)
    a = amb_u16.astype(f32) / f32(65535.0)
    color = w[5] * a
    total = np.full((h2, w2), w[5], dtype=f32)
    ys, xs = np.mgrid[0:h2, 0:w2]
    for i in range(-5, 6):
        if i == 0:
            continue
        tx = xs + (i if horizontal else 0); ty = ys + (0 if horizontal else i)
        cx = np.clip(tx, 0, w2 - 1); cy = np.clip(ty, 0, h2 - 1)
        nn = nrm[cy, cx]
        nn = np.where((tx < 0)[..., None], gcol[cy], nn)
        nn = np.where((ty < 0)[..., None], grow[cx], nn)
        inside = (tx >= 0) & (tx < w2) & (ty >= 0) & (ty < h2)
        nz = np.where(inside, vz[cy, cx], border_vz)
        dot = (nn[..., 0] * nrm[..., 0] + nn[..., 1] * nrm[..., 1]) + nn[..., 2] * nrm[..., 2]
        ok = (dot >= f32(0.8)) & (np.abs(nz - vz) <= f32(0.2))
        color = np.where(ok, color + w[i + 5] * a[cy, cx], color)
        total = np.where(ok, total + w[i + 5], total)
    out = color / total
    out = np.where(out > 0, np.where(out < 1, out, f32(1.0)), f32(0.0)).astype(f32)
    return (out * f32(65535.0) + f32(0.5)).astype(np.uint16)


def test_numpy_blur_agrees_with_oracle(oracle):
    for (W, H) in ((64, 64), (130, 34)):
        pl = scene_util.cpu_scene(W, H, 512, 64)
        p = scene_util.np_planes(pl)
        scb = oracle_lib.as_oracle_cb(pl["consts"].ssao_cb, oracle_lib.OrSsaoConstants)
        rng = np.random.default_rng(99)
        amb = rng.integers(0, 65536, size=(H // 2, W // 2), dtype=np.uint16)
        for horz in (True, False):
            ref = oracle.blur(scb, p["normal"], p["depth"], amb, horz)
            got = numpy_blur(scb, p["normal"], p["depth"], amb, horz)
            assert np.array_equal(got, ref), (W, H, horz, int((got != ref).sum()))
