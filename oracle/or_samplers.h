/*
 * or_samplers.h -- D3D fixed-function sampler emulation for the oracle (TEST INFRASTRUCTURE).
 *
 * Rules: texel i has its centre at (i + 0.5) / dim; bilinear at uv uses
 * t = uv*dim - 0.5 (one mad), i0 = floor(t), f = t - i0, texels i0 and i0+1 with weights (1-f) and f, evaluated
 * as lerp(a, b, f) = mad(f, b - a, a), x first then y; the address mode is applied per texel.
 * Non-finite or far-out-of-range coordinates are DEFINED to address only out-of-range texels.
 */
#ifndef OR_SAMPLERS_H
#define OR_SAMPLERS_H

#include "or_math.h"

/* D24_UNORM -> float: u24 / (2^24 - 1)  (ShadowMap.cpp:94 R24_UNORM_X8 view, Common/d3dApp.cpp depth SRV). */
static inline float or_d24(uint32_t v) { return (float)(v & 0x00FFFFFFu) / 16777215.0f; }
static inline float or_unorm16(uint16_t v) { return (float)v / 65535.0f; }
static inline float or_unorm8(uint8_t v) { return (float)v / 255.0f; }
/* UNORM write: floor(saturate(x) * (2^n - 1) + 0.5), the scale-and-bias as one mad */
static inline uint16_t or_to_unorm16(float x) { return (uint16_t)fmaf(or_saturate(x), 65535.0f, 0.5f); }
static inline uint8_t or_to_unorm8(float x) { return (uint8_t)fmaf(or_saturate(x), 255.0f, 0.5f); }

typedef struct or_bilin {
    int i0, j0;   /* top-left texel (may be out of range) */
    float fx, fy; /* weights of the +1 texels */
} or_bilin;

/* Clamp the (already floored) coordinate into [-2, dim+1] so the int conversion is defined; anything
 * non-finite maps to -2 (fully out of range). */
static inline int or_texel_index(float fl, uint32_t dim)
{
    if (!(fl >= -2.0f)) return -2;
    if (fl > (float)dim + 1.0f) return (int)dim + 1;
    return (int)fl;
}

static inline or_bilin or_bilinear_setup(float u, float v, uint32_t w, uint32_t h)
{
    or_bilin b;
    float tx = fmaf(u, (float)w, -0.5f);
    float ty = fmaf(v, (float)h, -0.5f);
    float flx = floorf(tx), fly = floorf(ty);
    b.fx = tx - flx;
    b.fy = ty - fly;
    b.i0 = or_texel_index(flx, w);
    b.j0 = or_texel_index(fly, h);
    if (!(b.fx == b.fx) || !(b.fy == b.fy) || fabsf(tx) == INFINITY || fabsf(ty) == INFINITY) {
        b.i0 = -2; b.j0 = -2; b.fx = 0.0f; b.fy = 0.0f;
    }
    return b;
}

static inline float or_bilerp(float a, float b, float c, float d, float fx, float fy)
{
    float top = or_lerp(a, b, fx);
    float bot = or_lerp(c, d, fx);
    return or_lerp(top, bot, fy);
}

/* gsamDepthMap: linear, border = opaque white (CRYCHIC.cpp:1057-1066). */
static inline float or_depth_texel_border1(const uint32_t* d, uint32_t W, uint32_t H, int x, int y)
{
    if (x < 0 || y < 0 || x >= (int)W || y >= (int)H) return 1.0f;
    return or_d24(d[(size_t)y * W + (size_t)x]);
}
static inline float or_depth_linear_border(const uint32_t* d, uint32_t W, uint32_t H, float u, float v)
{
    or_bilin b = or_bilinear_setup(u, v, W, H);
    float t00 = or_depth_texel_border1(d, W, H, b.i0, b.j0);
    float t10 = or_depth_texel_border1(d, W, H, b.i0 + 1, b.j0);
    float t01 = or_depth_texel_border1(d, W, H, b.i0, b.j0 + 1);
    float t11 = or_depth_texel_border1(d, W, H, b.i0 + 1, b.j0 + 1);
    return or_bilerp(t00, t10, t01, t11, b.fx, b.fy);
}

/* gsamShadow: comparison LESS_EQUAL (ref <= texel), min-mag linear, border opaque black = 0
 * (CRYCHIC.cpp:2649-2658).  Compare each texel first, then filter. */
static inline float or_shadow_texel_cmp(const uint32_t* s, uint32_t dim, int x, int y, float ref)
{
    float t = (x < 0 || y < 0 || x >= (int)dim || y >= (int)dim) ? 0.0f : or_d24(s[(size_t)y * dim + (size_t)x]);
    return (ref <= t) ? 1.0f : 0.0f;
}
static inline float or_shadow_cmp_linear(const uint32_t* s, uint32_t dim, float u, float v, float ref)
{
    or_bilin b = or_bilinear_setup(u, v, dim, dim);
    float c00 = or_shadow_texel_cmp(s, dim, b.i0, b.j0, ref);
    float c10 = or_shadow_texel_cmp(s, dim, b.i0 + 1, b.j0, ref);
    float c01 = or_shadow_texel_cmp(s, dim, b.i0, b.j0 + 1, ref);
    float c11 = or_shadow_texel_cmp(s, dim, b.i0 + 1, b.j0 + 1, ref);
    return or_bilerp(c00, c10, c01, c11, b.fx, b.fy);
}

/* gsamLinearWrap on the 256x256 RGBA8 random-vector map (CRYCHIC.cpp:1068-1073). */
static inline int or_wrap(int i, int dim) { int m = i % dim; return m < 0 ? m + dim : m; }
static inline void or_randvec_linear_wrap(const uint8_t* rv, float u, float v, float rgb[3])
{
    /* wrap the coordinate first so the int conversion stays in range */
    float uw = u - floorf(u), vw = v - floorf(v);
    or_bilin b = or_bilinear_setup(uw, vw, 256, 256);
    int x0 = or_wrap(b.i0, 256), x1 = or_wrap(b.i0 + 1, 256);
    int y0 = or_wrap(b.j0, 256), y1 = or_wrap(b.j0 + 1, 256);
    for (int c = 0; c < 3; ++c) {
        float t00 = or_unorm8(rv[((size_t)y0 * 256 + x0) * 4 + c]);
        float t10 = or_unorm8(rv[((size_t)y0 * 256 + x1) * 4 + c]);
        float t01 = or_unorm8(rv[((size_t)y1 * 256 + x0) * 4 + c]);
        float t11 = or_unorm8(rv[((size_t)y1 * 256 + x1) * 4 + c]);
        rgb[c] = or_bilerp(t00, t10, t01, t11, b.fx, b.fy);
    }
}

/* gsamLinearClamp on the half-res R16_UNORM ambient map (CRYCHIC.cpp:2624-2629). */
static inline int or_clampi(int i, int lo, int hi) { return i < lo ? lo : (i > hi ? hi : i); }
static inline float or_ambient_linear_clamp(const uint16_t* a, uint32_t w2, uint32_t h2, float u, float v)
{
    or_bilin b = or_bilinear_setup(u, v, w2, h2);
    int x0 = or_clampi(b.i0, 0, (int)w2 - 1), x1 = or_clampi(b.i0 + 1, 0, (int)w2 - 1);
    int y0 = or_clampi(b.j0, 0, (int)h2 - 1), y1 = or_clampi(b.j0 + 1, 0, (int)h2 - 1);
    float t00 = or_unorm16(a[(size_t)y0 * w2 + x0]);
    float t10 = or_unorm16(a[(size_t)y0 * w2 + x1]);
    float t01 = or_unorm16(a[(size_t)y1 * w2 + x0]);
    float t11 = or_unorm16(a[(size_t)y1 * w2 + x1]);
    return or_bilerp(t00, t10, t01, t11, b.fx, b.fy);
}

/* TextureCube.Sample with gsamLinearWrap (DeferredShading.hlsl:95, sky.hlsl:46): D3D face selection
 * (major axis, ties x >= y >= z), then bilinear inside the face with clamp-to-edge (seams are not
 * filtered across faces -- oracle definition). Faces: +X,-X,+Y,-Y,+Z,-Z. */
static inline void or_cube_linear(const uint8_t* cube, uint32_t dim, const float r[3], float* out, int nch)
{
    float ax = fabsf(r[0]), ay = fabsf(r[1]), az = fabsf(r[2]);
    int face; float sc, tc, ma;
    if (ax >= ay && ax >= az) { ma = ax; if (r[0] >= 0.0f) { face = 0; sc = -r[2]; tc = -r[1]; } else { face = 1; sc = r[2]; tc = -r[1]; } }
    else if (ay >= az)        { ma = ay; if (r[1] >= 0.0f) { face = 2; sc = r[0]; tc = r[2]; } else { face = 3; sc = r[0]; tc = -r[2]; } }
    else                      { ma = az; if (r[2] >= 0.0f) { face = 4; sc = r[0]; tc = -r[1]; } else { face = 5; sc = -r[0]; tc = -r[1]; } }
    float inv = or_rcp(ma);
    float u = fmaf(0.5f, sc * inv, 0.5f);       /* 0.5 * (sc / ma + 1) */
    float v = fmaf(0.5f, tc * inv, 0.5f);
    or_bilin b = or_bilinear_setup(u, v, dim, dim);
    int x0 = or_clampi(b.i0, 0, (int)dim - 1), x1 = or_clampi(b.i0 + 1, 0, (int)dim - 1);
    int y0 = or_clampi(b.j0, 0, (int)dim - 1), y1 = or_clampi(b.j0 + 1, 0, (int)dim - 1);
    const uint8_t* f = cube + (size_t)face * dim * dim * 4;
    for (int c = 0; c < nch; ++c) {
        float t00 = or_unorm8(f[((size_t)y0 * dim + x0) * 4 + c]);
        float t10 = or_unorm8(f[((size_t)y0 * dim + x1) * 4 + c]);
        float t01 = or_unorm8(f[((size_t)y1 * dim + x0) * 4 + c]);
        float t11 = or_unorm8(f[((size_t)y1 * dim + x1) * 4 + c]);
        out[c] = or_bilerp(t00, t10, t01, t11, b.fx, b.fy);
    }
}

/* ---- TextureCube.Sample with a mip chain (MIN_MAG_MIP_LINEAR: CRYCHIC.cpp:2617-2622; the whole chain is bound, :1148-1151) ----
 * D3D specifies what goes into the level-of-detail of a cube lookup -- the screen-space derivatives of the direction, carried to
 * the selected face -- and leaves the arithmetic to the hardware, so this is an ORACLE DEFINITION (parity unpinned):
 *   face, (sc, tc, ma) of r as in or_cube_linear; for a derivative d of r (ddx or ddy) the same selection gives (dsc, dtc) and
 *   dma = sign(r_major) * d_major; the face coordinate u = 0.5 (sc / ma + 1) then moves by (chain rule)
 *       du = (dsc - (sc / ma) dma) / ma * 0.5 dim      [level-0 texels per pixel], dv alike;
 *   rho^2 = max(du_x^2 + dv_x^2, du_y^2 + dv_y^2);  lod = min(0.5 log2(min(rho^2, 3e38)), levels - 1), 0 unless rho^2 > 1
 *   (the deterministic log2 of or_math.h);  the two levels floor(lod), floor(lod) + 1 are each filtered as or_cube_linear
 *   filters level 0 and mixed with one mad: c0 + frac * (c1 - c0); frac == 0 takes the lower level alone.
 * The chain is stored level after level, each level six faces of max(dim >> level, 1)^2 RGBA8 texels. */
static inline uint32_t or_cube_level_dim(uint32_t dim, uint32_t level) { uint32_t d = dim >> level; return d ? d : 1u; }
static inline size_t or_cube_level_offset(uint32_t dim, uint32_t level)
{
    size_t off = 0;
    for (uint32_t k = 0; k < level; ++k) { size_t d = or_cube_level_dim(dim, k); off += 6u * d * d * 4u; }
    return off;
}
static inline void or_cube_face_delta(const float r[3], const float d[3], float half_dim, float* du, float* dv)
{
    float ax = fabsf(r[0]), ay = fabsf(r[1]), az = fabsf(r[2]);
    float sc, tc, ma, dsc, dtc, dma;
    if (ax >= ay && ax >= az) { ma = ax; if (r[0] >= 0.0f) { sc = -r[2]; tc = -r[1]; dsc = -d[2]; dtc = -d[1]; dma = d[0]; } else { sc = r[2]; tc = -r[1]; dsc = d[2]; dtc = -d[1]; dma = -d[0]; } }
    else if (ay >= az)        { ma = ay; if (r[1] >= 0.0f) { sc = r[0]; tc = r[2]; dsc = d[0]; dtc = d[2]; dma = d[1]; } else { sc = r[0]; tc = -r[2]; dsc = d[0]; dtc = -d[2]; dma = -d[1]; } }
    else                      { ma = az; if (r[2] >= 0.0f) { sc = r[0]; tc = -r[1]; dsc = d[0]; dtc = -d[1]; dma = d[2]; } else { sc = -r[0]; tc = -r[1]; dsc = -d[0]; dtc = -d[1]; dma = -d[2]; } }
    float inv = or_rcp(ma);
    *du = fmaf(-(sc * inv), dma, dsc) * inv * half_dim;
    *dv = fmaf(-(tc * inv), dma, dtc) * inv * half_dim;
}
static inline float or_cube_lod(uint32_t dim, uint32_t levels, const float r[3], const float ddx[3], const float ddy[3])
{
    float half_dim = 0.5f * (float)dim, ux, vx, uy, vy;
    or_cube_face_delta(r, ddx, half_dim, &ux, &vx);
    or_cube_face_delta(r, ddy, half_dim, &uy, &vy);
    float rx = fmaf(ux, ux, vx * vx), ry = fmaf(uy, uy, vy * vy);
    float rho2 = (ry > rx) ? ry : rx;               /* a NaN rx makes rho2 NaN, a NaN ry is ignored; either way: */
    if (!(rho2 > 1.0f)) return 0.0f;                /* ... magnified, or undefined -> level 0 */
    float lod = 0.5f * or_det_log2_normal(rho2 < 3.0e38f ? rho2 : 3.0e38f);
    float top = (float)(levels - 1u);
    return lod < top ? lod : top;
}
static inline void or_cube_trilinear(const uint8_t* chain, uint32_t dim, uint32_t levels, const float r[3], float lod, float* out, int nch)
{
    uint32_t l0 = (uint32_t)lod;
    float frac = lod - (float)l0;
    float c0[4], c1[4];
    or_cube_linear(chain + or_cube_level_offset(dim, l0), or_cube_level_dim(dim, l0), r, c0, nch);
    if (frac == 0.0f || l0 + 1u >= levels) { for (int c = 0; c < nch; ++c) out[c] = c0[c]; return; }
    or_cube_linear(chain + or_cube_level_offset(dim, l0 + 1u), or_cube_level_dim(dim, l0 + 1u), r, c1, nch);
    for (int c = 0; c < nch; ++c) out[c] = fmaf(frac, c1[c] - c0[c], c0[c]);
}

#endif
