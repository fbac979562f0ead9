/*
 * or_raster.c -- oracle restatement of the producer passes: what the D3D12 fixed-function
 * pipeline does around Shaders/Shadows.hlsl, DrawNormals.hlsl and GeometryPass.hlsl when CRYCHIC::DrawSceneToShadowMap,
 * DrawNormalsAndDepth and DrawGBuffer (CRYCHIC.cpp:2477-2571) issue DrawIndexedInstanced (CRYCHIC.cpp:2473), plus
 * GeometryGenerator::CreateBox/CreateGrid and the skull.txt loader.  TEST INFRASTRUCTURE, parity unpinned.
 *
 * Rasteriser definition (D3D11.3 functional spec rules; arithmetic fixed here):
 *   - clip against 0 <= z <= w (depth clip on) in clip space, Sutherland-Hodgman, new vertex = a + t*(b - a),
 *     t = da / (da - db), fan triangulation; then against a guard band of +-2^21 pixels in x and y (see clip_plane);
 *   - viewport: sx = (x/w + 1) * (W/2), sy = (1 - y/w) * (H/2), z = z/w; sx, sy snapped to 1/256 pixel
 *     (floor(v*256 + 0.5)); clockwise on screen = front, back faces culled;
 *   - coverage: 64-bit integer edge functions at pixel centres (+0.5), top-left rule;
 *   - depth: screen-space barycentric interpolation of z in double, + bias (shadow pass), clamp, D24 =
 *     floor(z * (2^24 - 1) + 0.5); test LESS against 1.0; equal depths keep the earlier primitive;
 *   - attributes: perspective-correct (weights lambda_i / w_i) in double, rounded to float once.
 */
#include "crychic_oracle.h"
#include "or_samplers.h"
#include <stdio.h>
#include <stdlib.h>

typedef struct vs_out {
    float posH[4], posW[3], normalW[3], tangentW[3], tex[2];
} vs_out;

typedef struct setup_tri {
    int32_t X[3], Y[3];
    float z[3], invw[3];
    float posW[3][3], normalW[3][3], tangentW[3][3], tex[3][2];
    uint32_t matIndex;
    int64_t A2;
} setup_tri;

static void mul3x3(const float v[3], const float mem[16], float out[3])
{
    for (int j = 0; j < 3; ++j) out[j] = (v[0] * mem[4 * j + 0] + v[1] * mem[4 * j + 1]) + v[2] * mem[4 * j + 2];
}

/* VS of GeometryPass.hlsl:22-42 / DrawNormals.hlsl:38-64 / Shadows.hlsl:21-44 (superset). */
static void vertex_shader(const or_vertex* vin, const or_instance_data* inst, const or_material_data* mat,
                          const float viewProj[16], vs_out* o)
{
    float p4[4] = { vin->Pos[0], vin->Pos[1], vin->Pos[2], 1.0f }, pw[4];
    or_mul_v4_m(p4, inst->World, pw);
    or_mul_v4_m(pw, viewProj, o->posH);
    o->posW[0] = pw[0]; o->posW[1] = pw[1]; o->posW[2] = pw[2];
    mul3x3(vin->Normal, inst->World, o->normalW);
    mul3x3(vin->TangentU, inst->World, o->tangentW);
    float t4[4] = { vin->TexC[0], vin->TexC[1], 0.0f, 1.0f }, t1[4], t2[4];
    or_mul_v4_m(t4, inst->TexTransform, t1);
    if (mat) { or_mul_v4_m(t1, mat->MatTransform, t2); o->tex[0] = t2[0]; o->tex[1] = t2[1]; }
    else { o->tex[0] = t1[0]; o->tex[1] = t1[1]; }
}

static void lerp_vertex(const vs_out* a, const vs_out* b, float t, vs_out* o)
{
    const float* fa = (const float*)a; const float* fb = (const float*)b; float* fo = (float*)o;
    for (size_t i = 0; i < sizeof(vs_out) / sizeof(float); ++i) fo[i] = fa[i] + t * (fb[i] - fa[i]);
}

/* Clipping: 0 <= z <= w (DepthClipEnable, Common/d3dx12.h:203-216) and a guard band in x and y -- D3D12 renders a triangle with
 * a vertex far outside the viewport (CRYCHIC.cpp:2473), it does not drop it.  The band is |x|, |y| <= g w with
 * g = 2^21 / (dim / 2) NDC units (+-2^21 pixels around the viewport centre: half of the +-2^22 pixels the fixed-point edge
 * functions are defined for), a build-defined choice where D3D leaves the extent to the hardware; a polygon inside the band is
 * not touched.  Planes: 0: z >= 0   1: w - z >= 0   2: g w + x >= 0   3: g w - x >= 0   4: g w + y >= 0   5: g w - y >= 0. */
#define OR_CLIP_PLANES 6
#define OR_MAX_POLY (3 + OR_CLIP_PLANES + 1)
static float guard_band(uint32_t dim) { return 2097152.0f / (0.5f * (float)dim); }
static float clip_distance(const vs_out* v, int plane, float gx, float gy)
{
    switch (plane) {
    case 0: return v->posH[2];
    case 1: return v->posH[3] - v->posH[2];
    case 2: return gx * v->posH[3] + v->posH[0];
    case 3: return gx * v->posH[3] - v->posH[0];
    case 4: return gy * v->posH[3] + v->posH[1];
    default: return gy * v->posH[3] - v->posH[1];
    }
}
/* Sutherland-Hodgman against one plane; dist(v) >= 0 is inside. */
static int clip_plane(const vs_out* in, int n, int plane, vs_out* out, float gx, float gy)
{
    int m = 0;
    for (int i = 0; i < n; ++i) {
        const vs_out* a = &in[i]; const vs_out* b = &in[(i + 1) % n];
        float da = clip_distance(a, plane, gx, gy), db = clip_distance(b, plane, gx, gy);
        int ina = da >= 0.0f, inb = db >= 0.0f;
        if (ina) out[m++] = *a;
        if (ina != inb) {
            float t = da / (da - db);
            lerp_vertex(a, b, t, &out[m++]);
        }
    }
    return m;
}
/* only the planes some vertex is outside of are applied, in plane order */
static int clip_triangle(vs_out* poly, vs_out* tmp, uint32_t W, uint32_t H)
{
    const float gx = guard_band(W), gy = guard_band(H);
    int n = 3;
    for (int plane = 0; plane < OR_CLIP_PLANES && n >= 3; ++plane) {
        int any = 0;
        for (int i = 0; i < n; ++i) any |= !(clip_distance(&poly[i], plane, gx, gy) >= 0.0f);
        if (!any) continue;
        n = clip_plane(poly, n, plane, tmp, gx, gy);
        for (int i = 0; i < n; ++i) poly[i] = tmp[i];
    }
    return n;
}

static int is_top_left(int32_t ax, int32_t ay, int32_t bx, int32_t by)
{
    /* clockwise triangle, y down: left edges run upwards, the top edge runs to the right */
    return (by < ay) || (by == ay && bx > ax);
}
static int64_t orient(int32_t ax, int32_t ay, int32_t bx, int32_t by, int32_t px, int32_t py)
{
    return (int64_t)(bx - ax) * (int64_t)(py - ay) - (int64_t)(by - ay) * (int64_t)(px - ax);
}

typedef struct tri_list { setup_tri* t; size_t n, cap; int overflow; } tri_list;

static void emit_triangle(tri_list* L, const vs_out* v0, const vs_out* v1, const vs_out* v2, uint32_t matIndex,
                          uint32_t W, uint32_t H)
{
    const vs_out* v[3] = { v0, v1, v2 };
    setup_tri s;
    for (int i = 0; i < 3; ++i) {
        float invw = 1.0f / v[i]->posH[3];
        float nx = v[i]->posH[0] * invw, ny = v[i]->posH[1] * invw;
        float sx = (nx + 1.0f) * (0.5f * (float)W);
        float sy = (1.0f - ny) * (0.5f * (float)H);
        if (!(fabsf(sx) < 4194304.0f) || !(fabsf(sy) < 4194304.0f)) { L->overflow = 1; return; }
        s.X[i] = (int32_t)floorf(sx * 256.0f + 0.5f);
        s.Y[i] = (int32_t)floorf(sy * 256.0f + 0.5f);
        s.z[i] = v[i]->posH[2] * invw;
        s.invw[i] = invw;
        memcpy(s.posW[i], v[i]->posW, 12); memcpy(s.normalW[i], v[i]->normalW, 12);
        memcpy(s.tangentW[i], v[i]->tangentW, 12); memcpy(s.tex[i], v[i]->tex, 8);
    }
    s.A2 = orient(s.X[0], s.Y[0], s.X[1], s.Y[1], s.X[2], s.Y[2]);
    if (s.A2 <= 0) return; /* back-facing or degenerate */
    s.matIndex = matIndex;
    if (L->n == L->cap) {
        L->cap = L->cap ? L->cap * 2 : 4096;
        L->t = (setup_tri*)realloc(L->t, L->cap * sizeof(setup_tri));
    }
    L->t[L->n++] = s;
}

/* float -> half, round to nearest even (the fp16 render-target write of DrawNormals.hlsl:93). */
uint16_t or_float_to_half(float f)
{
    uint32_t x = or_float_to_bits(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | ((ax > 0x7F800000u) ? 0x200u | ((ax >> 13) & 0x3FFu) : 0u));
    if (ax >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);          /* rounds to >= 65520 -> inf */
    if (ax < 0x33000001u) return (uint16_t)sign;                        /* < 2^-25 (or == 2^-25 ties to even 0) */
    int e = (int)(ax >> 23) - 127;
    uint32_t m = (ax & 0x007FFFFFu) | 0x00800000u;
    int shift;
    uint32_t he;
    if (e < -14) { shift = 13 + (-14 - e); he = 0; }                    /* subnormal half */
    else { shift = 13; he = (uint32_t)(e + 15); }
    uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    if (he == 0) return (uint16_t)(sign | q);                           /* q may carry into the exponent: correct */
    return (uint16_t)(sign | (((he << 10) + (q - 0x400u)) & 0x7FFFu));  /* q in [0x400, 0x800]: carry handled by + */
}

/* One bilinear WRAP fetch in level `level` of a mip chain stored level after level. */
static void sample_level(const or_texture* t, uint32_t level, float u, float v, float out[4])
{
    uint32_t w = t->width, h = t->height;
    const uint8_t* base = t->rgba8;
    for (uint32_t k = 0; k < level; ++k) { base += (size_t)w * h * 4; w = w > 1 ? w >> 1 : 1; h = h > 1 ? h >> 1 : 1; }
    float uw = u - floorf(u), vw = v - floorf(v);
    or_bilin b = or_bilinear_setup(uw, vw, w, h);
    int x0 = or_wrap(b.i0, (int)w), x1 = or_wrap(b.i0 + 1, (int)w);
    int y0 = or_wrap(b.j0, (int)h), y1 = or_wrap(b.j0 + 1, (int)h);
    for (int c = 0; c < 4; ++c) {
        float t00 = or_unorm8(base[((size_t)y0 * w + x0) * 4 + c]);
        float t10 = or_unorm8(base[((size_t)y0 * w + x1) * 4 + c]);
        float t01 = or_unorm8(base[((size_t)y1 * w + x0) * 4 + c]);
        float t11 = or_unorm8(base[((size_t)y1 * w + x1) * 4 + c]);
        out[c] = or_bilerp(t00, t10, t01, t11, b.fx, b.fy);
    }
}

/* gsamAnisotropicWrap: D3D12_FILTER_ANISOTROPIC, MaxAnisotropy 8, WRAP (CRYCHIC.cpp:2631-2638), used by GeometryPass.hlsl:53,60.
 * D3D specifies the sampler's inputs (the quad's derivatives) but leaves the anisotropic kernel to the hardware; the oracle
 * DEFINES it (DESIGN.md "Oracle definitions"): footprint axes in level-0 texels Px = (dudx W, dvdx H), Py = (dudy W, dvdy H);
 * N = clamp(ceil(|major| / |minor|), 1, 8) probes spread evenly along the major axis and centred on (u, v);
 * lod = clamp(log2(|major| / N), 0, levels - 1); each probe trilinear; the result is the mean of the probes.
 * grad = { dudx, dvdx, dudy, dvdy }.  Without a mip chain: one bilinear fetch of level 0. */
static void sample_texture(const or_texture* tex, uint32_t nTextures, uint32_t index, int isNormalMap, float u, float v,
                           const float grad[4], float out[4])
{
    if (!tex || index >= nTextures || !tex[index].rgba8) {
        if (isNormalMap) { out[0] = 0.5f; out[1] = 0.5f; out[2] = 1.0f; out[3] = 1.0f; }
        else { out[0] = out[1] = out[2] = out[3] = 1.0f; }
        return;
    }
    const or_texture* t = &tex[index];
    if (t->mipLevels <= 1) { sample_level(t, 0, u, v, out); return; }
    float W = (float)t->width, H = (float)t->height;
    float pxu = grad[0] * W, pxv = grad[1] * H, pyu = grad[2] * W, pyv = grad[3] * H;
    float lx2 = fmaf(pxv, pxv, pxu * pxu), ly2 = fmaf(pyv, pyv, pyu * pyu);
    int majorX = lx2 >= ly2;
    float pmax = or_len(majorX ? lx2 : ly2), pmin = or_len(majorX ? ly2 : lx2);
    float nf = fminf(fmaxf(ceilf(pmax * or_rcp(pmin)), 1.0f), 8.0f);
    float rn = or_rcp(nf);
    float lod = fminf(fmaxf(or_det_log2f_(pmax * rn), 0.0f), (float)(t->mipLevels - 1));
    float l0f = floorf(lod), fl = lod - l0f;
    uint32_t l0 = (uint32_t)l0f, l1 = l0 + 1 < t->mipLevels ? l0 + 1 : t->mipLevels - 1;
    float du = majorX ? grad[0] : grad[2], dv = majorX ? grad[1] : grad[3];
    int N = (int)nf;
    float acc[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
    for (int k = 0; k < N; ++k) {
        float s = fmaf((float)k + 0.5f, rn, -0.5f);
        float uu = fmaf(s, du, u), vv = fmaf(s, dv, v);
        float a[4], bb[4];
        sample_level(t, l0, uu, vv, a);
        sample_level(t, l1, uu, vv, bb);
        for (int c = 0; c < 4; ++c) acc[c] += or_lerp(a[c], bb[c], fl);
    }
    for (int c = 0; c < 4; ++c) out[c] = acc[c] * rn;
}

/* TexC of primitive T at pixel (qx, qy): the plane equations evaluated at a pixel that may lie outside the triangle (what a
 * helper invocation of the 2 x 2 quad computes). */
static void tex_at(const setup_tri* T, int qx, int qy, float uv[2])
{
    int32_t ax = qx * 256 + 128, ay = qy * 256 + 128;
    double a0 = (double)orient(T->X[1], T->Y[1], T->X[2], T->Y[2], ax, ay);
    double a1 = (double)orient(T->X[2], T->Y[2], T->X[0], T->Y[0], ax, ay);
    double a2 = (double)orient(T->X[0], T->Y[0], T->X[1], T->Y[1], ax, ay);
    double invA = 1.0 / (double)T->A2;
    double r0 = (a0 * invA) * (double)T->invw[0], r1 = (a1 * invA) * (double)T->invw[1], r2 = (a2 * invA) * (double)T->invw[2];
    double ir = 1.0 / ((r0 + r1) + r2);
    for (int c = 0; c < 2; ++c)
        uv[c] = (float)((((double)T->tex[0][c] * r0 + (double)T->tex[1][c] * r1) + (double)T->tex[2][c] * r2) * ir);
}

int or_rasterize(int mode, const float view[16], const float viewProj[16], const or_draw_item* items, uint32_t nItems,
                 const or_material_data* materials, uint32_t nMaterials, const or_texture* textures, uint32_t nTextures,
                 uint32_t W, uint32_t H, int depthBias, float slopeScaledDepthBias, uint32_t* depth_out,
                 uint16_t* normal_out, float* g0, float* g1, float* g2)
{
    tri_list L = { 0, 0, 0, 0 };
    /* ---- vertex shading, clipping, setup, in draw order ---- */
    for (uint32_t it = 0; it < nItems; ++it) {
        const or_draw_item* d = &items[it];
        vs_out* vs = (vs_out*)malloc((size_t)d->vertexCount * sizeof(vs_out));
        for (uint32_t inst = 0; inst < d->instanceCount; ++inst) {
            const or_instance_data* I = &d->instances[inst];
            const or_material_data* M = (materials && I->MaterialIndex < nMaterials) ? &materials[I->MaterialIndex] : NULL;
            for (uint32_t k = 0; k < d->vertexCount; ++k) vertex_shader(&d->vertices[k], I, M, viewProj, &vs[k]);
            for (uint32_t t = 0; t + 2 < d->indexCount; t += 3) {
                vs_out poly[OR_MAX_POLY], tmp[OR_MAX_POLY];
                for (int c = 0; c < 3; ++c) {
                    int64_t vi = (int64_t)d->indices[d->startIndexLocation + t + c] + d->baseVertexLocation;
                    if (vi < 0 || vi >= (int64_t)d->vertexCount) { free(vs); free(L.t); return -1; }
                    poly[c] = vs[vi];
                }
                int n = clip_triangle(poly, tmp, W, H);
                for (int c = 1; c + 1 < n; ++c) emit_triangle(&L, &poly[0], &poly[c], &poly[c + 1], I->MaterialIndex, W, H);
            }
        }
        free(vs);
    }
    if (L.overflow) { free(L.t); return -1; }

    /* ---- coverage + depth: visibility key = (d24 << 32) | (serial + 1), minimum wins ---- */
    size_t npx = (size_t)W * H;
    uint64_t* vis = (uint64_t*)malloc(npx * sizeof(uint64_t));
    for (size_t i = 0; i < npx; ++i) vis[i] = (uint64_t)0x00FFFFFFu << 32;
    for (size_t s = 0; s < L.n; ++s) {
        const setup_tri* T = &L.t[s];
        int32_t minX = T->X[0], maxX = T->X[0], minY = T->Y[0], maxY = T->Y[0];
        for (int i = 1; i < 3; ++i) {
            if (T->X[i] < minX) minX = T->X[i];
            if (T->X[i] > maxX) maxX = T->X[i];
            if (T->Y[i] < minY) minY = T->Y[i];
            if (T->Y[i] > maxY) maxY = T->Y[i];
        }
        /* pixel px is a candidate when its centre px*256+128 lies in [min, max] */
        int x0 = (minX - 128 + 255) >> 8, x1 = (maxX - 128) >> 8, y0 = (minY - 128 + 255) >> 8, y1 = (maxY - 128) >> 8;
        if (x0 < 0) x0 = 0;
        if (y0 < 0) y0 = 0;
        if (x1 > (int)W - 1) x1 = (int)W - 1;
        if (y1 > (int)H - 1) y1 = (int)H - 1;
        if (x0 > x1 || y0 > y1) continue;
        int tl0 = is_top_left(T->X[1], T->Y[1], T->X[2], T->Y[2]);
        int tl1 = is_top_left(T->X[2], T->Y[2], T->X[0], T->Y[0]);
        int tl2 = is_top_left(T->X[0], T->Y[0], T->X[1], T->Y[1]);
        double bias = 0.0;
        if (mode == 0) {
            double dz1 = (double)T->z[1] - (double)T->z[0], dz2 = (double)T->z[2] - (double)T->z[0];
            double dzdx = (dz1 * (double)(T->Y[2] - T->Y[0]) - dz2 * (double)(T->Y[1] - T->Y[0])) / (double)T->A2 * 256.0;
            double dzdy = (dz2 * (double)(T->X[1] - T->X[0]) - dz1 * (double)(T->X[2] - T->X[0])) / (double)T->A2 * 256.0;
            double ms = fabs(dzdx) > fabs(dzdy) ? fabs(dzdx) : fabs(dzdy);
            bias = (double)depthBias * (1.0 / 16777216.0) + (double)slopeScaledDepthBias * ms;
        }
        const double invA2 = 1.0 / (double)T->A2;   /* barycentrics: one reciprocal per triangle, then products (all in double) */
        for (int py = y0; py <= y1; ++py) {
            for (int px = x0; px <= x1; ++px) {
                int32_t cx = px * 256 + 128, cy = py * 256 + 128;
                int64_t w0 = orient(T->X[1], T->Y[1], T->X[2], T->Y[2], cx, cy);
                int64_t w1 = orient(T->X[2], T->Y[2], T->X[0], T->Y[0], cx, cy);
                int64_t w2 = orient(T->X[0], T->Y[0], T->X[1], T->Y[1], cx, cy);
                if (w0 < 0 || w1 < 0 || w2 < 0) continue;
                if ((w0 == 0 && !tl0) || (w1 == 0 && !tl1) || (w2 == 0 && !tl2)) continue;
                double l1 = (double)w1 * invA2, l2 = (double)w2 * invA2;
                double z = (double)T->z[0] + l1 * ((double)T->z[1] - (double)T->z[0]) + l2 * ((double)T->z[2] - (double)T->z[0]);
                z = z + bias;
                if (!(z > 0.0)) z = 0.0;
                if (z > 1.0) z = 1.0;
                uint64_t d24 = (uint64_t)(z * 16777215.0 + 0.5);
                uint64_t key = (d24 << 32) | (uint64_t)(s + 1);
                size_t idx = (size_t)py * W + (size_t)px;
                if (key < vis[idx]) vis[idx] = key;
            }
        }
    }

    /* ---- resolve: depth plane + the pass's pixel shader on the winning primitive ---- */
#pragma omp parallel for schedule(dynamic, 8)
    for (int py = 0; py < (int)H; ++py) {
        for (uint32_t px = 0; px < W; ++px) {
            size_t idx = (size_t)py * W + px;
            uint64_t key = vis[idx];
            uint32_t serial = (uint32_t)(key & 0xFFFFFFFFu);
            depth_out[idx] = (uint32_t)(key >> 32);
            if (mode == 0) continue;
            if (serial == 0) {
                if (mode == 1) { uint16_t* o = normal_out + idx * 4; o[0] = 0; o[1] = 0; o[2] = 0x3C00; o[3] = 0; }
                else { for (int c = 0; c < 4; ++c) { g0[idx * 4 + c] = 0.0f; g1[idx * 4 + c] = 0.0f; g2[idx * 4 + c] = 0.0f; } }
                continue;
            }
            const setup_tri* T = &L.t[serial - 1];
            int32_t cx = (int32_t)px * 256 + 128, cy = py * 256 + 128;
            double w0 = (double)orient(T->X[1], T->Y[1], T->X[2], T->Y[2], cx, cy);
            double w1 = (double)orient(T->X[2], T->Y[2], T->X[0], T->Y[0], cx, cy);
            double w2 = (double)orient(T->X[0], T->Y[0], T->X[1], T->Y[1], cx, cy);
            double invA = 1.0 / (double)T->A2;
            double q0 = (w0 * invA) * (double)T->invw[0], q1 = (w1 * invA) * (double)T->invw[1], q2 = (w2 * invA) * (double)T->invw[2];
            double invqs = 1.0 / ((q0 + q1) + q2);   /* perspective correction: one reciprocal per pixel */
#define OR_INTERP(a0, a1, a2) ((float)((((double)(a0) * q0 + (double)(a1) * q1) + (double)(a2) * q2) * invqs))
            float nW[3], N[3];
            for (int c = 0; c < 3; ++c) nW[c] = OR_INTERP(T->normalW[0][c], T->normalW[1][c], T->normalW[2][c]);
            or_normalize3(nW, N);                                        /* DrawNormals.hlsl:85 / GeometryPass.hlsl:58 */
            if (mode == 1) {
                float nv[3];
                mul3x3(N, view, nv);                                     /* DrawNormals.hlsl:92 */
                uint16_t* o = normal_out + idx * 4;
                o[0] = or_float_to_half(nv[0]); o[1] = or_float_to_half(nv[1]); o[2] = or_float_to_half(nv[2]); o[3] = 0;
                continue;
            }
            float posW[3], tanW[3], tex[2];
            for (int c = 0; c < 3; ++c) posW[c] = OR_INTERP(T->posW[0][c], T->posW[1][c], T->posW[2][c]);
            for (int c = 0; c < 3; ++c) tanW[c] = OR_INTERP(T->tangentW[0][c], T->tangentW[1][c], T->tangentW[2][c]);
            for (int c = 0; c < 2; ++c) tex[c] = OR_INTERP(T->tex[0][c], T->tex[1][c], T->tex[2][c]);
#undef OR_INTERP
            /* GeometryPass.hlsl:44-66 */
            or_material_data defmat;
            memset(&defmat, 0, sizeof defmat);
            defmat.DiffuseAlbedo[0] = defmat.DiffuseAlbedo[1] = defmat.DiffuseAlbedo[2] = defmat.DiffuseAlbedo[3] = 1.0f;
            defmat.Roughness = 0.5f; defmat.Metalness = 0.5f;
            const or_material_data* M = (materials && T->matIndex < nMaterials) ? &materials[T->matIndex] : &defmat;
            /* implicit derivatives of Sample(): differences inside the pixel's 2 x 2 quad (ddx along its row, ddy along its column) */
            float grad[4] = { 0.0f, 0.0f, 0.0f, 0.0f };
            {
                int qx = (int)px & ~1, qy = py & ~1;
                float a[2], b[2];
                tex_at(T, qx, py, a); tex_at(T, qx + 1, py, b);
                grad[0] = b[0] - a[0]; grad[1] = b[1] - a[1];
                tex_at(T, (int)px, qy, a); tex_at(T, (int)px, qy + 1, b);
                grad[2] = b[0] - a[0]; grad[3] = b[1] - a[1];
            }
            float dtex[4], ntex[4];
            sample_texture(textures, nTextures, M->DiffuseMapIndex, 0, tex[0], tex[1], grad, dtex);   /* :53 */
            sample_texture(textures, nTextures, M->NormalMapIndex, 1, tex[0], tex[1], grad, ntex);    /* :60 */
            float albedo[3] = { M->DiffuseAlbedo[0] * dtex[0], M->DiffuseAlbedo[1] * dtex[1], M->DiffuseAlbedo[2] * dtex[2] };
            /* NormalSampleToWorldSpace  Common.hlsl:112-128 */
            float nT[3] = { 2.0f * ntex[0] - 1.0f, 2.0f * ntex[1] - 1.0f, 2.0f * ntex[2] - 1.0f };
            float dtn = or_dot3(tanW, N);
            float tt[3] = { tanW[0] - dtn * N[0], tanW[1] - dtn * N[1], tanW[2] - dtn * N[2] }, Tn[3];
            or_normalize3(tt, Tn);
            float B[3] = { N[1] * Tn[2] - N[2] * Tn[1], N[2] * Tn[0] - N[0] * Tn[2], N[0] * Tn[1] - N[1] * Tn[0] };
            float bumped[3];
            for (int c = 0; c < 3; ++c) bumped[c] = (nT[0] * Tn[c] + nT[1] * B[c]) + nT[2] * N[c];
            /* EncodePBRToGBuffer  GBuffer.hlsl:22-31 */
            g0[idx * 4 + 0] = posW[0]; g0[idx * 4 + 1] = posW[1]; g0[idx * 4 + 2] = posW[2]; g0[idx * 4 + 3] = M->Metalness;
            g1[idx * 4 + 0] = albedo[0]; g1[idx * 4 + 1] = albedo[1]; g1[idx * 4 + 2] = albedo[2]; g1[idx * 4 + 3] = M->Roughness;
            g2[idx * 4 + 0] = bumped[0]; g2[idx * 4 + 1] = bumped[1]; g2[idx * 4 + 2] = bumped[2]; g2[idx * 4 + 3] = 1.0f;
        }
    }
    int n = (int)L.n;
    free(vis);
    free(L.t);
    return n;
}

/* ---- GeometryGenerator (Common/GeometryGenerator.cpp) ---------------------------------------------------------- */
static or_vertex mkv(float px, float py, float pz, float nx, float ny, float nz, float tx, float ty, float tz, float u, float v)
{
    or_vertex r = { { px, py, pz }, { nx, ny, nz }, { u, v }, { tx, ty, tz } };
    return r;
}
static void norm3_(const float a[3], float o[3])
{
    float l = sqrtf(or_dot3_host(a, a));
    o[0] = a[0] / l; o[1] = a[1] / l; o[2] = a[2] / l;
}
/* GeometryGenerator.cpp:277-305 */
static or_vertex midpoint(const or_vertex* a, const or_vertex* b)
{
    or_vertex m;
    float n[3], t[3];
    for (int c = 0; c < 3; ++c) { m.Pos[c] = 0.5f * (a->Pos[c] + b->Pos[c]); n[c] = 0.5f * (a->Normal[c] + b->Normal[c]); t[c] = 0.5f * (a->TangentU[c] + b->TangentU[c]); }
    norm3_(n, m.Normal);
    norm3_(t, m.TangentU);
    for (int c = 0; c < 2; ++c) m.TexC[c] = 0.5f * (a->TexC[c] + b->TexC[c]);
    return m;
}

int or_create_box(float width, float height, float depth, uint32_t numSubdivisions, or_vertex* vout, uint32_t vcap,
                  uint32_t* iout, uint32_t icap, uint32_t* nIdx)
{
    float w2 = 0.5f * width, h2 = 0.5f * height, d2 = 0.5f * depth;
    or_vertex v[24] = {
        mkv(-w2, -h2, -d2, 0, 0, -1, 1, 0, 0, 0, 1), mkv(-w2, +h2, -d2, 0, 0, -1, 1, 0, 0, 0, 0), mkv(+w2, +h2, -d2, 0, 0, -1, 1, 0, 0, 1, 0), mkv(+w2, -h2, -d2, 0, 0, -1, 1, 0, 0, 1, 1),
        mkv(-w2, -h2, +d2, 0, 0, 1, -1, 0, 0, 1, 1), mkv(+w2, -h2, +d2, 0, 0, 1, -1, 0, 0, 0, 1), mkv(+w2, +h2, +d2, 0, 0, 1, -1, 0, 0, 0, 0), mkv(-w2, +h2, +d2, 0, 0, 1, -1, 0, 0, 1, 0),
        mkv(-w2, +h2, -d2, 0, 1, 0, 1, 0, 0, 0, 1), mkv(-w2, +h2, +d2, 0, 1, 0, 1, 0, 0, 0, 0), mkv(+w2, +h2, +d2, 0, 1, 0, 1, 0, 0, 1, 0), mkv(+w2, +h2, -d2, 0, 1, 0, 1, 0, 0, 1, 1),
        mkv(-w2, -h2, -d2, 0, -1, 0, -1, 0, 0, 1, 1), mkv(+w2, -h2, -d2, 0, -1, 0, -1, 0, 0, 0, 1), mkv(+w2, -h2, +d2, 0, -1, 0, -1, 0, 0, 0, 0), mkv(-w2, -h2, +d2, 0, -1, 0, -1, 0, 0, 1, 0),
        mkv(-w2, -h2, +d2, -1, 0, 0, 0, 0, -1, 0, 1), mkv(-w2, +h2, +d2, -1, 0, 0, 0, 0, -1, 0, 0), mkv(-w2, +h2, -d2, -1, 0, 0, 0, 0, -1, 1, 0), mkv(-w2, -h2, -d2, -1, 0, 0, 0, 0, -1, 1, 1),
        mkv(+w2, -h2, -d2, 1, 0, 0, 0, 0, 1, 0, 1), mkv(+w2, +h2, -d2, 1, 0, 0, 0, 0, 1, 0, 0), mkv(+w2, +h2, +d2, 1, 0, 0, 0, 0, 1, 1, 0), mkv(+w2, -h2, +d2, 1, 0, 0, 0, 0, 1, 1, 1)
    };
    uint32_t nv = 24, ni = 36;
    or_vertex* V = (or_vertex*)malloc(sizeof(or_vertex) * nv);
    uint32_t* I = (uint32_t*)malloc(sizeof(uint32_t) * ni);
    memcpy(V, v, sizeof v);
    for (uint32_t f = 0; f < 6; ++f) { uint32_t b = 4 * f; uint32_t q[6] = { b, b + 1, b + 2, b, b + 2, b + 3 }; memcpy(I + 6 * f, q, sizeof q); }
    if (numSubdivisions > 6u) numSubdivisions = 6u;
    for (uint32_t s = 0; s < numSubdivisions; ++s) {                   /* Subdivide, GeometryGenerator.cpp:214-275 */
        uint32_t nt = ni / 3;
        or_vertex* V2 = (or_vertex*)malloc(sizeof(or_vertex) * nt * 6);
        uint32_t* I2 = (uint32_t*)malloc(sizeof(uint32_t) * nt * 12);
        for (uint32_t i = 0; i < nt; ++i) {
            or_vertex a = V[I[i * 3 + 0]], b = V[I[i * 3 + 1]], c = V[I[i * 3 + 2]];
            or_vertex m0 = midpoint(&a, &b), m1 = midpoint(&b, &c), m2 = midpoint(&a, &c);
            or_vertex six[6] = { a, b, c, m0, m1, m2 };
            memcpy(V2 + i * 6, six, sizeof six);
            uint32_t k[12] = { i * 6 + 0, i * 6 + 3, i * 6 + 5, i * 6 + 3, i * 6 + 4, i * 6 + 5, i * 6 + 5, i * 6 + 4, i * 6 + 2, i * 6 + 3, i * 6 + 1, i * 6 + 4 };
            memcpy(I2 + i * 12, k, sizeof k);
        }
        free(V); free(I);
        V = V2; I = I2; nv = nt * 6; ni = nt * 12;
    }
    int rc = (int)nv;
    if (nIdx) *nIdx = ni;
    if (vout && iout) { if (nv > vcap || ni > icap) rc = -1; else { memcpy(vout, V, sizeof(or_vertex) * nv); memcpy(iout, I, sizeof(uint32_t) * ni); } }
    free(V); free(I);
    return rc;
}

/* GeometryGenerator.cpp:551-614 */
int or_create_grid(float width, float depth, uint32_t m, uint32_t n, or_vertex* v, uint32_t vcap, uint32_t* idx, uint32_t icap, uint32_t* nIdx)
{
    uint32_t vc = m * n, fc = (m - 1) * (n - 1) * 2;
    if (nIdx) *nIdx = fc * 3;
    if (!v || !idx) return (int)vc;
    if (vc > vcap || fc * 3 > icap) return -1;
    float halfWidth = 0.5f * width, halfDepth = 0.5f * depth;
    float dx = width / (float)(n - 1), dz = depth / (float)(m - 1), du = 1.0f / (float)(n - 1), dv = 1.0f / (float)(m - 1);
    for (uint32_t i = 0; i < m; ++i) {
        float z = halfDepth - (float)i * dz;
        for (uint32_t j = 0; j < n; ++j) {
            float x = -halfWidth + (float)j * dx;
            v[i * n + j] = mkv(x, 0.0f, z, 0, 1, 0, 1, 0, 0, (float)j * du, (float)i * dv);
        }
    }
    uint32_t k = 0;
    for (uint32_t i = 0; i < m - 1; ++i)
        for (uint32_t j = 0; j < n - 1; ++j) {
            idx[k] = i * n + j; idx[k + 1] = i * n + j + 1; idx[k + 2] = (i + 1) * n + j;
            idx[k + 3] = (i + 1) * n + j; idx[k + 4] = i * n + j + 1; idx[k + 5] = (i + 1) * n + j + 1;
            k += 6;
        }
    return (int)vc;
}

/* CRYCHIC::BuildSkullGeometry  CRYCHIC.cpp:1447-1557 */
int or_load_mesh_text(const char* path, or_vertex* v, uint32_t vcap, uint32_t* idx, uint32_t icap, uint32_t* nVerts, uint32_t* nIdx)
{
    FILE* f = fopen(path, "r");
    if (!f) return -1;
    char tok[64];
    unsigned vcount = 0, tcount = 0;
    if (fscanf(f, "%63s %u", tok, &vcount) != 2 || fscanf(f, "%63s %u", tok, &tcount) != 2) { fclose(f); return -1; }
    for (int i = 0; i < 4; ++i) if (fscanf(f, "%63s", tok) != 1) { fclose(f); return -1; }   /* "VertexList (pos, normal) {" */
    if (nVerts) *nVerts = vcount;
    if (nIdx) *nIdx = 3 * tcount;
    if (!v || !idx) { fclose(f); return (int)vcount; }
    if (vcount > vcap || 3 * tcount > icap) { fclose(f); return -1; }
    for (unsigned i = 0; i < vcount; ++i) {
        or_vertex* p = &v[i];
        if (fscanf(f, "%f %f %f %f %f %f", &p->Pos[0], &p->Pos[1], &p->Pos[2], &p->Normal[0], &p->Normal[1], &p->Normal[2]) != 6) { fclose(f); return -1; }
        p->TexC[0] = p->TexC[1] = 0.0f;
        const float* N = p->Normal;
        float up[3] = { 0.0f, 1.0f, 0.0f }, c[3];
        if (fabsf(or_dot3_host(N, up)) < 1.0f - 0.001f) {                    /* :1489-1493: T = normalize(up x N) */
            c[0] = up[1] * N[2] - up[2] * N[1]; c[1] = up[2] * N[0] - up[0] * N[2]; c[2] = up[0] * N[1] - up[1] * N[0];
        } else {                                                        /* :1494-1499: up = +z, T = normalize(N x up) */
            float u2[3] = { 0.0f, 0.0f, 1.0f };
            c[0] = N[1] * u2[2] - N[2] * u2[1]; c[1] = N[2] * u2[0] - N[0] * u2[2]; c[2] = N[0] * u2[1] - N[1] * u2[0];
        }
        norm3_(c, p->TangentU);
    }
    for (int i = 0; i < 3; ++i) if (fscanf(f, "%63s", tok) != 1) { fclose(f); return -1; }   /* "} TriangleList {" */
    for (unsigned i = 0; i < 3 * tcount; ++i) if (fscanf(f, "%u", &idx[i]) != 1) { fclose(f); return -1; }
    fclose(f);
    return (int)vcount;
}
