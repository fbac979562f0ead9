// raster_core.hpp -- per-primitive / per-pixel bodies of the producer passes: the work the D3D12
// fixed-function pipeline does around Shaders/Shadows.hlsl, DrawNormals.hlsl and GeometryPass.hlsl for the
// DrawIndexedInstanced calls of CRYCHIC::DrawSceneToShadowMap / DrawNormalsAndDepth / DrawGBuffer
// (CRYCHIC.cpp:2473, 2477-2571).  Host+device inline (checked on the CPU against the oracle via tests/hostsim).
//
// Pipeline: vertex shader -> clip against 0 <= z <= w -> viewport + 1/256-pixel snap -> cull (clockwise = front) ->
// integer edge functions with the top-left rule -> depth (double, D24) -> 64-bit visibility key, minimum wins
// (depth LESS, ties to the earlier primitive) -> per-pixel resolve with perspective-correct attributes.
#pragma once
#include "devmath.hpp"
#include "crychic_hip.h"

namespace cry {

struct VsOut {          // 15 floats, interpolated as one block when clipping
    float posH[4];
    float posW[3];
    float normalW[3];
    float tangentW[3];
    float tex[2];
};
constexpr int kVsFloats = 15;

struct SetupTri {       // 192 B
    int32_t X[3], Y[3];             // 24.8 fixed-point screen position
    float z[3], invw[3];
    float posW[3][3], normalW[3][3], tangentW[3][3], tex[3][2];
    uint32_t matIndex;
    uint32_t pad;
    int64_t A2;                     // twice the signed area (> 0 in every listed slot; the host simulation marks empty slots with 0)
};

CRY_HD void mul3x3(const float v[3], const float* m, float out[3])
{
    for (int j = 0; j < 3; ++j) out[j] = (v[0] * m[4 * j + 0] + v[1] * m[4 * j + 1]) + v[2] * m[4 * j + 2];
}
CRY_HD void mul4x4(const float v[4], const float* m, float out[4])
{
    for (int j = 0; j < 4; ++j) out[j] = mulcol(v[0], v[1], v[2], v[3], m + 4 * j);
}

// VS of GeometryPass.hlsl:22-42 (superset of DrawNormals.hlsl:38-64 and Shadows.hlsl:21-44)
CRY_HD VsOut vertex_shader(const crychic_vertex& vin, const crychic_instance_data& inst, const crychic_material_data* mat,
                           const float* viewProj)
{
    VsOut o;
    const float p4[4] = { vin.Pos[0], vin.Pos[1], vin.Pos[2], 1.0f };
    float pw[4];
    mul4x4(p4, inst.World, pw);
    mul4x4(pw, viewProj, o.posH);
    o.posW[0] = pw[0]; o.posW[1] = pw[1]; o.posW[2] = pw[2];
    mul3x3(vin.Normal, inst.World, o.normalW);
    mul3x3(vin.TangentU, inst.World, o.tangentW);
    const float t4[4] = { vin.TexC[0], vin.TexC[1], 0.0f, 1.0f };
    float t1[4], t2[4];
    mul4x4(t4, inst.TexTransform, t1);
    if (mat) { mul4x4(t1, mat->MatTransform, t2); o.tex[0] = t2[0]; o.tex[1] = t2[1]; }
    else { o.tex[0] = t1[0]; o.tex[1] = t1[1]; }
    return o;
}

CRY_HD VsOut lerp_vertex(const VsOut& a, const VsOut& b, float t)
{
    VsOut o;
    const float* fa = a.posH; const float* fb = b.posH; float* fo = o.posH;   // the struct is 15 contiguous floats
    for (int i = 0; i < kVsFloats; ++i) fo[i] = fa[i] + t * (fb[i] - fa[i]);
    return o;
}

// Clipping.  D3D12 clips to 0 <= z <= w (DepthClipEnable, Common/d3dx12.h:203-216) and to a GUARD BAND in x and y: a triangle
// with a vertex far outside the viewport is rendered, not dropped (CRYCHIC.cpp:2473 draws whatever the scene holds).  The
// fixed-point edge functions below are defined up to +-2^22 pixels, so a polygon that leaves the band |x|, |y| <= g w with
// g = 2^21 / (dim / 2) NDC units (i.e. +-2^21 pixels around the viewport centre, half the representable range) is clipped to it;
// the new edges lie two million pixels outside the target and cannot touch a pixel centre.  A polygon inside the band is not
// touched, so every image rendered before this clipper existed is unchanged bit for bit.
// Planes: 0: z >= 0   1: w - z >= 0   2: g w + x >= 0   3: g w - x >= 0   4: g w + y >= 0   5: g w - y >= 0  (g = gx for 2, 3; gy for 4, 5).
constexpr int kClipPlanes = 6;
constexpr int kMaxPolyVerts = 3 + kClipPlanes + 1;      // Sutherland-Hodgman adds at most one vertex per plane
constexpr int kSlotsPerTriangle = 3 + kClipPlanes - 2;   // fan triangles of a 9-gon
CRY_HD float guard_band(uint32_t dim) { return 2097152.0f / (0.5f * (float)dim); }
CRY_HD float clip_distance(const VsOut& v, int plane, float gx, float gy)
{
    switch (plane) {
    case 0: return v.posH[2];
    case 1: return v.posH[3] - v.posH[2];
    case 2: return gx * v.posH[3] + v.posH[0];
    case 3: return gx * v.posH[3] - v.posH[0];
    case 4: return gy * v.posH[3] + v.posH[1];
    default: return gy * v.posH[3] - v.posH[1];
    }
}
// Sutherland-Hodgman against one plane; returns the new vertex count (<= n + 1).
CRY_HD int clip_plane(const VsOut* in, int n, int plane, VsOut* out, float gx = 0.0f, float gy = 0.0f)
{
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const VsOut& a = in[i];
        const VsOut& b = in[(i + 1 == n) ? 0 : i + 1];
        const float da = clip_distance(a, plane, gx, gy), db = clip_distance(b, plane, gx, gy);
        const bool ina = da >= 0.0f, inb = db >= 0.0f;
        if (ina) out[m++] = a;
        if (ina != inb) out[m++] = lerp_vertex(a, b, da / (da - db));
    }
    return m;
}
// The whole clipper for one triangle: `poly` holds the three vertices on entry and the clipped polygon on return (`tmp` is
// scratch; both kMaxPolyVerts long).  Only the planes some vertex is outside of are applied, in plane order.
CRY_HD int clip_triangle(VsOut* poly, VsOut* tmp, uint32_t W, uint32_t H)
{
    const float gx = guard_band(W), gy = guard_band(H);
    int n = 3;
    for (int plane = 0; plane < kClipPlanes && n >= 3; ++plane) {
        bool any = false;
        for (int i = 0; i < n; ++i) any = any | !(clip_distance(poly[i], plane, gx, gy) >= 0.0f);
        if (!any) continue;
        n = clip_plane(poly, n, plane, tmp, gx, gy);
        for (int i = 0; i < n; ++i) poly[i] = tmp[i];
    }
    return n;
}

CRY_HD int64_t orient2d(int32_t ax, int32_t ay, int32_t bx, int32_t by, int32_t px, int32_t py)
{
    return (int64_t)(bx - ax) * (int64_t)(py - ay) - (int64_t)(by - ay) * (int64_t)(px - ax);
}
// clockwise triangle on a y-down screen: left edges run upwards, the top edge runs to the right
CRY_HD bool is_top_left(int32_t ax, int32_t ay, int32_t bx, int32_t by) { return (by < ay) || (by == ay && bx > ax); }

// Viewport transform + snap + cull.  Returns false (and leaves A2 = 0) for culled / degenerate triangles; sets *overflow when a
// coordinate is outside the +-2^22 pixel range the fixed-point edge functions are defined for -- after clip_triangle that means
// a non-finite vertex position, nothing else.
CRY_HD bool setup_triangle(const VsOut& v0, const VsOut& v1, const VsOut& v2, uint32_t matIndex, uint32_t W, uint32_t H,
                           SetupTri& s, bool* overflow)
{
    const VsOut* v[3] = { &v0, &v1, &v2 };
    s.A2 = 0;
    for (int i = 0; i < 3; ++i) {
        const float invw = 1.0f / v[i]->posH[3];
        const float nx = v[i]->posH[0] * invw, ny = v[i]->posH[1] * invw;
        const float sx = (nx + 1.0f) * (0.5f * (float)W);
        const float sy = (1.0f - ny) * (0.5f * (float)H);
        if (!(__builtin_fabsf(sx) < 4194304.0f) || !(__builtin_fabsf(sy) < 4194304.0f)) { *overflow = true; return false; }
        s.X[i] = (int32_t)__builtin_floorf(sx * 256.0f + 0.5f);
        s.Y[i] = (int32_t)__builtin_floorf(sy * 256.0f + 0.5f);
        s.z[i] = v[i]->posH[2] * invw;
        s.invw[i] = invw;
        for (int c = 0; c < 3; ++c) { s.posW[i][c] = v[i]->posW[c]; s.normalW[i][c] = v[i]->normalW[c]; s.tangentW[i][c] = v[i]->tangentW[c]; }
        s.tex[i][0] = v[i]->tex[0]; s.tex[i][1] = v[i]->tex[1];
    }
    s.matIndex = matIndex;
    s.pad = 0;
    const int64_t a2 = orient2d(s.X[0], s.Y[0], s.X[1], s.Y[1], s.X[2], s.Y[2]);
    if (a2 <= 0) return false;   // back-facing (counter-clockwise on screen) or degenerate
    s.A2 = a2;
    return true;
}

struct PixelBox { int x0, y0, x1, y1; };   // inclusive pixel range whose centres can be covered
// Rows are limited to [yLo, yHi): the whole target, or the scissor of a strip-limited pass.
CRY_HD PixelBox triangle_box(const SetupTri& t, uint32_t W, uint32_t yLo, uint32_t yHi)
{
    int32_t minX = t.X[0], maxX = t.X[0], minY = t.Y[0], maxY = t.Y[0];
    for (int i = 1; i < 3; ++i) {
        minX = t.X[i] < minX ? t.X[i] : minX; maxX = t.X[i] > maxX ? t.X[i] : maxX;
        minY = t.Y[i] < minY ? t.Y[i] : minY; maxY = t.Y[i] > maxY ? t.Y[i] : maxY;
    }
    PixelBox b;
    b.x0 = (minX - 128 + 255) >> 8; b.x1 = (maxX - 128) >> 8;
    b.y0 = (minY - 128 + 255) >> 8; b.y1 = (maxY - 128) >> 8;
    b.x0 = b.x0 < 0 ? 0 : b.x0; b.y0 = b.y0 < (int)yLo ? (int)yLo : b.y0;
    b.x1 = b.x1 > (int)W - 1 ? (int)W - 1 : b.x1; b.y1 = b.y1 > (int)yHi - 1 ? (int)yHi - 1 : b.y1;
    return b;
}
CRY_HD PixelBox triangle_box(const SetupTri& t, uint32_t W, uint32_t H) { return triangle_box(t, W, 0u, H); }

// Depth bias of the shadow PSO (CRYCHIC.cpp:1601-1603): DepthBias * 2^-24 + SlopeScaledDepthBias * max |dz/dx|,|dz/dy|
CRY_HD double triangle_depth_bias(const SetupTri& t, int depthBias, float slopeScaledDepthBias)
{
    const double dz1 = (double)t.z[1] - (double)t.z[0], dz2 = (double)t.z[2] - (double)t.z[0];
    const double dzdx = (dz1 * (double)(t.Y[2] - t.Y[0]) - dz2 * (double)(t.Y[1] - t.Y[0])) / (double)t.A2 * 256.0;
    const double dzdy = (dz2 * (double)(t.X[1] - t.X[0]) - dz1 * (double)(t.X[2] - t.X[0])) / (double)t.A2 * 256.0;
    const double ax = __builtin_fabs(dzdx), ay = __builtin_fabs(dzdy);
    const double ms = ax > ay ? ax : ay;
    return (double)depthBias * (1.0 / 16777216.0) + (double)slopeScaledDepthBias * ms;
}

struct EdgeFlags { bool tl0, tl1, tl2; };
CRY_HD EdgeFlags triangle_edge_flags(const SetupTri& t)
{
    return EdgeFlags{ is_top_left(t.X[1], t.Y[1], t.X[2], t.Y[2]), is_top_left(t.X[2], t.Y[2], t.X[0], t.Y[0]),
                      is_top_left(t.X[0], t.Y[0], t.X[1], t.Y[1]) };
}

// Coverage + depth of pixel (px, py); returns the visibility key or ~0 when the pixel centre is not covered.
CRY_HD uint64_t fragment_key(const SetupTri& t, const EdgeFlags& e, double bias, int px, int py, uint32_t serial)
{
    const int32_t cx = px * 256 + 128, cy = py * 256 + 128;
    const int64_t w0 = orient2d(t.X[1], t.Y[1], t.X[2], t.Y[2], cx, cy);
    const int64_t w1 = orient2d(t.X[2], t.Y[2], t.X[0], t.Y[0], cx, cy);
    const int64_t w2 = orient2d(t.X[0], t.Y[0], t.X[1], t.Y[1], cx, cy);
    if ((w0 | w1 | w2) < 0) return ~0ull;
    if ((w0 == 0 && !e.tl0) || (w1 == 0 && !e.tl1) || (w2 == 0 && !e.tl2)) return ~0ull;
    const double invA2 = 1.0 / (double)t.A2;   // loop-invariant: one reciprocal per triangle (hoisted out of the pixel loops)
    const double l1 = (double)w1 * invA2, l2 = (double)w2 * invA2;
    double z = (double)t.z[0] + l1 * ((double)t.z[1] - (double)t.z[0]) + l2 * ((double)t.z[2] - (double)t.z[0]);
    z = z + bias;
    z = (z > 0.0) ? z : 0.0;
    z = (z > 1.0) ? 1.0 : z;
    const uint64_t d24 = (uint64_t)(z * 16777215.0 + 0.5);
    return (d24 << 32) | (uint64_t)serial;
}

constexpr uint64_t kVisClear = (uint64_t)0x00FFFFFFu << 32;   // depth 1.0, no primitive

// float -> half, round to nearest even (fp16 render-target write)
CRY_HD uint16_t float_to_half(float f)
{
    _Float16 h = (_Float16)f;
    uint16_t u;
    __builtin_memcpy(&u, &h, 2);
    return u;
}

// A material texture (gTextureMaps[], Common.hlsl:52): R8G8B8A8 levels stored back to back, level k = max(1, w >> k) x
// max(1, h >> k) texels; mipLevels 0 or 1 = level 0 only.  Same layout as crychic_texture.
struct Texture { const uint8_t* rgba8; uint32_t width, height, mipLevels; };

// One bilinear WRAP fetch inside level `level` (the level's own texel grid).
CRY_HD f4 sample_texture_level(const Texture& t, uint32_t level, float u, float v)
{
    uint32_t w = t.width, h = t.height;
    size_t off = 0;
    for (uint32_t k = 0; k < level; ++k) { off += (size_t)w * h; w = w > 1u ? w >> 1 : 1u; h = h > 1u ? h >> 1 : 1u; }
    const float uw = u - __builtin_floorf(u), vw = v - __builtin_floorf(v);
    const Bilin b = bilinear_setup(uw, vw, w, h);
    auto wrap = [](int i, int n) { int m = i % n; return (uint32_t)(m < 0 ? m + n : m); };
    const uint32_t x0 = wrap(b.i0, (int)w), x1 = wrap(b.i0 + 1, (int)w);
    const uint32_t y0 = wrap(b.j0, (int)h), y1 = wrap(b.j0 + 1, (int)h);
    const uint32_t* p = (const uint32_t*)t.rgba8 + off;
    const uint32_t t00 = p[y0 * w + x0], t10 = p[y0 * w + x1], t01 = p[y1 * w + x0], t11 = p[y1 * w + x1];
    f4 o;
    o.x = bilerp(unorm8_to_float(t00 & 255u), unorm8_to_float(t10 & 255u), unorm8_to_float(t01 & 255u), unorm8_to_float(t11 & 255u), b.fx, b.fy);
    o.y = bilerp(unorm8_to_float((t00 >> 8) & 255u), unorm8_to_float((t10 >> 8) & 255u), unorm8_to_float((t01 >> 8) & 255u), unorm8_to_float((t11 >> 8) & 255u), b.fx, b.fy);
    o.z = bilerp(unorm8_to_float((t00 >> 16) & 255u), unorm8_to_float((t10 >> 16) & 255u), unorm8_to_float((t01 >> 16) & 255u), unorm8_to_float((t11 >> 16) & 255u), b.fx, b.fy);
    o.w = bilerp(unorm8_to_float(t00 >> 24), unorm8_to_float(t10 >> 24), unorm8_to_float(t01 >> 24), unorm8_to_float(t11 >> 24), b.fx, b.fy);
    return o;
}

// Screen-space derivatives of the texture coordinate, as the 2 x 2 pixel quad provides them to Sample().
struct TexGrad { float dudx, dvdx, dudy, dvdy; };

// gsamAnisotropicWrap (CRYCHIC.cpp:2631-2638: D3D12_FILTER_ANISOTROPIC, MaxAnisotropy 8, WRAP) on a mip chain.  D3D leaves
// the anisotropic kernel to the implementation; this is the definition the oracle and the kernels share:
//   footprint axes in level-0 texels  Px = (dudx W, dvdx H), Py = (dudy W, dvdy H); major = the longer one
//   N = clamp(ceil(|major| / |minor|), 1, 8) probes spread evenly along the major axis, centred on (u, v)
//   lod = clamp(log2(|major| / N), 0, levels - 1); every probe is trilinear (two bilinear WRAP fetches, lerp by frac(lod))
//   result = the mean of the probes.
// A texture without a mip chain (mipLevels <= 1) is one bilinear fetch of level 0: what round 1 modelled.
CRY_HD f4 sample_texture(const Texture* tex, uint32_t nTextures, uint32_t index, bool isNormalMap, float u, float v, const TexGrad& g)
{
    if (!tex || index >= nTextures || !tex[index].rgba8) return isNormalMap ? f4{ 0.5f, 0.5f, 1.0f, 1.0f } : f4{ 1.0f, 1.0f, 1.0f, 1.0f };
    const Texture t = tex[index];
    if (t.mipLevels <= 1u) return sample_texture_level(t, 0u, u, v);
    const float W = (float)t.width, H = (float)t.height;
    const float pxu = g.dudx * W, pxv = g.dvdx * H, pyu = g.dudy * W, pyv = g.dvdy * H;
    const float lx2 = fma(pxv, pxv, pxu * pxu), ly2 = fma(pyv, pyv, pyu * pyu);
    const bool majorX = lx2 >= ly2;
    const float pmax = len_from_sq(majorX ? lx2 : ly2), pmin = len_from_sq(majorX ? ly2 : lx2);
    const float nf = __builtin_fminf(__builtin_fmaxf(__builtin_ceilf(pmax * rcp(pmin)), 1.0f), 8.0f);      // NaN -> 1
    const float rn = rcp(nf);
    const float lod = __builtin_fminf(__builtin_fmaxf(det_log2(pmax * rn), 0.0f), (float)(t.mipLevels - 1u));
    const float l0f = __builtin_floorf(lod), fl = lod - l0f;
    const uint32_t l0 = (uint32_t)l0f, l1 = l0 + 1u < t.mipLevels ? l0 + 1u : t.mipLevels - 1u;
    const float du = majorX ? g.dudx : g.dudy, dv = majorX ? g.dvdx : g.dvdy;
    const int N = (int)nf;
    f4 acc{ 0.0f, 0.0f, 0.0f, 0.0f };
    for (int k = 0; k < N; ++k) {
        const float s = fma((float)k + 0.5f, rn, -0.5f);
        const float uu = fma(s, du, u), vv = fma(s, dv, v);
        const f4 a = sample_texture_level(t, l0, uu, vv), b = sample_texture_level(t, l1, uu, vv);
        acc.x += lerpf(a.x, b.x, fl); acc.y += lerpf(a.y, b.y, fl); acc.z += lerpf(a.z, b.z, fl); acc.w += lerpf(a.w, b.w, fl);
    }
    return f4{ acc.x * rn, acc.y * rn, acc.z * rn, acc.w * rn };
}

struct ResolveOut {
    f3 normalV;         // mode bit 0 (DrawNormals.hlsl)
    f4 g0, g1, g2;      // mode bit 1 (GeometryPass.hlsl); mode 3 = both pixel shaders on the one visibility result
};

// Pixel stage for the winning primitive of pixel (px, py): perspective-correct attributes, then the pass's PS.
CRY_HD ResolveOut resolve_pixel(int mode, const SetupTri& t, int px, int py, const float* view,
                                const crychic_material_data* materials, uint32_t nMaterials, const Texture* textures,
                                uint32_t nTextures)
{
    const int32_t cx = px * 256 + 128, cy = py * 256 + 128;
    const double w0 = (double)orient2d(t.X[1], t.Y[1], t.X[2], t.Y[2], cx, cy);
    const double w1 = (double)orient2d(t.X[2], t.Y[2], t.X[0], t.Y[0], cx, cy);
    const double w2 = (double)orient2d(t.X[0], t.Y[0], t.X[1], t.Y[1], cx, cy);
    const double invA = 1.0 / (double)t.A2;
    const double q0 = (w0 * invA) * (double)t.invw[0], q1 = (w1 * invA) * (double)t.invw[1], q2 = (w2 * invA) * (double)t.invw[2];
    const double invqs = 1.0 / ((q0 + q1) + q2);   // perspective correction: one reciprocal per pixel
    auto interp = [&](float a0, float a1, float a2) { return (float)((((double)a0 * q0 + (double)a1 * q1) + (double)a2 * q2) * invqs); };

    ResolveOut r{};
    const f3 N = normalize3(f3{ interp(t.normalW[0][0], t.normalW[1][0], t.normalW[2][0]), interp(t.normalW[0][1], t.normalW[1][1], t.normalW[2][1]),
                                interp(t.normalW[0][2], t.normalW[1][2], t.normalW[2][2]) });   // DrawNormals.hlsl:85, GeometryPass.hlsl:58
    if (mode & 1) {
        const float n3[3] = { N.x, N.y, N.z };
        float nv[3];
        mul3x3(n3, view, nv);                                                                  // DrawNormals.hlsl:92
        r.normalV = f3{ nv[0], nv[1], nv[2] };
    }
    if (!(mode & 2)) return r;
    const f3 posW{ interp(t.posW[0][0], t.posW[1][0], t.posW[2][0]), interp(t.posW[0][1], t.posW[1][1], t.posW[2][1]),
                   interp(t.posW[0][2], t.posW[1][2], t.posW[2][2]) };
    const f3 tanW{ interp(t.tangentW[0][0], t.tangentW[1][0], t.tangentW[2][0]), interp(t.tangentW[0][1], t.tangentW[1][1], t.tangentW[2][1]),
                   interp(t.tangentW[0][2], t.tangentW[1][2], t.tangentW[2][2]) };
    const float tu = interp(t.tex[0][0], t.tex[1][0], t.tex[2][0]), tv = interp(t.tex[0][1], t.tex[1][1], t.tex[2][1]);
    // TexC of this primitive at another pixel of the quad (a helper invocation extrapolates the plane equations there)
    auto tex_at = [&](int qx, int qy, float& u, float& v) {
        const int32_t ax = qx * 256 + 128, ay = qy * 256 + 128;
        const double a0 = (double)orient2d(t.X[1], t.Y[1], t.X[2], t.Y[2], ax, ay), a1 = (double)orient2d(t.X[2], t.Y[2], t.X[0], t.Y[0], ax, ay);
        const double a2 = (double)orient2d(t.X[0], t.Y[0], t.X[1], t.Y[1], ax, ay);
        const double r0 = (a0 * invA) * (double)t.invw[0], r1 = (a1 * invA) * (double)t.invw[1], r2 = (a2 * invA) * (double)t.invw[2];
        const double ir = 1.0 / ((r0 + r1) + r2);
        u = (float)((((double)t.tex[0][0] * r0 + (double)t.tex[1][0] * r1) + (double)t.tex[2][0] * r2) * ir);
        v = (float)((((double)t.tex[0][1] * r0 + (double)t.tex[1][1] * r1) + (double)t.tex[2][1] * r2) * ir);
    };

    // GeometryPass.hlsl:44-66; a material index outside the buffer reads MaterialData's defaults (FrameResource.h:17-27)
    float albedoM[3] = { 1.0f, 1.0f, 1.0f }, roughness = 0.5f, metalness = 0.5f;
    uint32_t dmap = 0, nmap = 0;
    if (materials && t.matIndex < nMaterials) {
        const crychic_material_data& M = materials[t.matIndex];
        albedoM[0] = M.DiffuseAlbedo[0]; albedoM[1] = M.DiffuseAlbedo[1]; albedoM[2] = M.DiffuseAlbedo[2];
        roughness = M.Roughness; metalness = M.Metalness; dmap = M.DiffuseMapIndex; nmap = M.NormalMapIndex;
    }
    // Implicit derivatives of Sample(): differences inside the pixel's 2 x 2 quad (ddx along its row, ddy along its column).
    // Only a texture with a mip chain looks at them.
    TexGrad grad{ 0.0f, 0.0f, 0.0f, 0.0f };
    const bool mipped = textures && ((dmap < nTextures && textures[dmap].mipLevels > 1u) || (nmap < nTextures && textures[nmap].mipLevels > 1u));
    if (mipped) {
        const int qx = px & ~1, qy = py & ~1;
        float ua, va, ub, vb;
        tex_at(qx, py, ua, va); tex_at(qx + 1, py, ub, vb);
        grad.dudx = ub - ua; grad.dvdx = vb - va;
        tex_at(px, qy, ua, va); tex_at(px, qy + 1, ub, vb);
        grad.dudy = ub - ua; grad.dvdy = vb - va;
    }
    const f4 dtex = sample_texture(textures, nTextures, dmap, false, tu, tv, grad);             // :53
    const f4 ntex = sample_texture(textures, nTextures, nmap, true, tu, tv, grad);              // :60
    // NormalSampleToWorldSpace  Common.hlsl:112-128
    const f3 nT{ 2.0f * ntex.x - 1.0f, 2.0f * ntex.y - 1.0f, 2.0f * ntex.z - 1.0f };
    const float dtn = dot3(tanW, N);
    const f3 T = normalize3(f3{ tanW.x - dtn * N.x, tanW.y - dtn * N.y, tanW.z - dtn * N.z });
    const f3 B{ N.y * T.z - N.z * T.y, N.z * T.x - N.x * T.z, N.x * T.y - N.y * T.x };
    const f3 bumped{ (nT.x * T.x + nT.y * B.x) + nT.z * N.x, (nT.x * T.y + nT.y * B.y) + nT.z * N.y, (nT.x * T.z + nT.y * B.z) + nT.z * N.z };
    r.g0 = f4{ posW.x, posW.y, posW.z, metalness };                                             // GBuffer.hlsl:22-31
    r.g1 = f4{ albedoM[0] * dtex.x, albedoM[1] * dtex.y, albedoM[2] * dtex.z, roughness };
    r.g2 = f4{ bumped.x, bumped.y, bumped.z, 1.0f };
    return r;
}

}  // namespace cry
