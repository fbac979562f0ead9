/*
 * or_ssao.c -- oracle restatement of Shaders/Ssao.hlsl and Shaders/SsaoBlur.hlsl plus the
 * Ssao::ComputeSsao sequencing (TEST INFRASTRUCTURE, parity unpinned: see crychic_oracle.h).
 */
#include "crychic_oracle.h"
#include "or_samplers.h"

/* Ssao.hlsl:110-115  viewZ = gProj[3][2] / (z_ndc - gProj[2][2]); HLSL M[r][c] = mem[4c + r]. */
static inline float ndc_to_view(const or_ssao_constants* cb, float z_ndc)
{
    return or_div(cb->Proj[4 * 2 + 3], z_ndc - cb->Proj[4 * 2 + 2]);
}
float or_ndc_depth_to_view_depth(const or_ssao_constants* cb, float z_ndc) { return ndc_to_view(cb, z_ndc); }

/* Full-screen-quad texture coordinate of half-res pixel (x, y): the rasteriser interpolates gTexCoords
 * (Ssao.hlsl:41-49,62) to the pixel centre. */
static inline void pixel_uv(uint32_t x, uint32_t y, uint32_t w2, uint32_t h2, float* u, float* v)
{
    *u = or_div((float)x + 0.5f, (float)w2);
    *v = or_div((float)y + 0.5f, (float)h2);
}

/* gsamPointClamp fetch of the full-res normal map at the centre of half-res pixel (xi, yi); (xi, yi) may
 * lie outside the half-res map (blur taps).  uv*W = 2*xi + 1 lies exactly on a texel edge; the oracle
 * DEFINES the result as texel floor(2*xi + 1) = 2*xi + 1, and CLAMP addressing then
 * acts on that full-res texel index (xi = -1 -> texel 0, not texel 1). */
static inline void normal_point(const uint16_t* normal, uint32_t W, uint32_t H, int xi, int yi, float n[3])
{
    int tx = or_clampi(2 * xi + 1, 0, (int)W - 1);
    int ty = or_clampi(2 * yi + 1, 0, (int)H - 1);
    const uint16_t* t = normal + ((size_t)ty * W + (size_t)tx) * 4;
    n[0] = or_half_bits_to_float(t[0]);
    n[1] = or_half_bits_to_float(t[1]);
    n[2] = or_half_bits_to_float(t[2]);
}

/* gsamDepthMap (linear, border 1.0) at the centre of half-res pixel (xi, yi): t = 2*xi + 0.5 exactly, so
 * texels 2xi, 2xi+1 (and rows 2yi, 2yi+1) with weights 0.5 -- valid for even W, H (all configs).  Pixels
 * outside the half-res map address only border texels. */
static inline float depth_at_half_pixel(const uint32_t* depth, uint32_t W, uint32_t H, int xi, int yi)
{
    float t00 = or_depth_texel_border1(depth, W, H, 2 * xi, 2 * yi);
    float t10 = or_depth_texel_border1(depth, W, H, 2 * xi + 1, 2 * yi);
    float t01 = or_depth_texel_border1(depth, W, H, 2 * xi, 2 * yi + 1);
    float t11 = or_depth_texel_border1(depth, W, H, 2 * xi + 1, 2 * yi + 1);
    return or_bilerp(t00, t10, t01, t11, 0.5f, 0.5f);
}

/* Ssao.hlsl:76-108 */
static inline float occlusion_function(const or_ssao_constants* cb, float distZ)
{
    float occlusion = 0.0f;
    if (distZ > cb->SurfaceEpsilon) {
        float fadeLength = cb->OcclusionFadeEnd - cb->OcclusionFadeStart;
        occlusion = or_saturate(or_div(cb->OcclusionFadeEnd - distZ, fadeLength));
    }
    return occlusion;
}

/* Ssao.hlsl:117-199 for one half-res pixel. */
static uint16_t ssao_pixel(const or_ssao_constants* cb, const uint16_t* normal, const uint32_t* depth,
                           const uint8_t* randvec, uint32_t W, uint32_t H, uint32_t x, uint32_t y)
{
    uint32_t w2 = W / 2, h2 = H / 2;
    float u, v;
    pixel_uv(x, y, w2, h2, &u, &v);

    /* VS (Ssao.hlsl:58-72): PosH = (2u-1, 1-2v, 0, 1); PosV = mul(PosH, gInvProj).xyz / .w, evaluated at the
     * pixel centre (the interpolation of a projective-linear quantity over the quad). */
    float posh[4] = { fmaf(2.0f, u, -1.0f), fmaf(-2.0f, v, 1.0f), 0.0f, 1.0f };
    float ph[4];
    or_mul_v4_m(posh, cb->InvProj, ph);
    float rw = or_rcp(ph[3]);
    float PosV[3] = { ph[0] * rw, ph[1] * rw, ph[2] * rw };

    float nraw[3], n[3];
    normal_point(normal, W, H, (int)x, (int)y, nraw);
    or_normalize3(nraw, n);                                           /* :125 */
    float pz = ndc_to_view(cb, depth_at_half_pixel(depth, W, H, (int)x, (int)y)); /* :126-127 */

    float t = or_div(pz, PosV[2]);                                    /* :135 */
    float p[3] = { t * PosV[0], t * PosV[1], t * PosV[2] };

    float rv[3];
    or_randvec_linear_wrap(randvec, 4.0f * u, 4.0f * v, rv);          /* :138 */
    float randVec[3] = { fmaf(2.0f, rv[0], -1.0f), fmaf(2.0f, rv[1], -1.0f), fmaf(2.0f, rv[2], -1.0f) };

    float occlusionSum = 0.0f;
    for (int i = 0; i < 14; ++i) {                                    /* gSampleCount :39 */
        float offset[3];
        or_reflect3(cb->OffsetVectors[i], randVec, offset);          /* :148 */
        float flip = or_sign(or_dot3(offset, n));                    /* :151 */
        float fr = flip * cb->OcclusionRadius;
        float q[4] = { fmaf(fr, offset[0], p[0]), fmaf(fr, offset[1], p[1]), fmaf(fr, offset[2], p[2]), 1.0f }; /* :154 */
        float projQ[4];
        or_mul_v4_m(q, cb->ProjTex, projQ);                          /* :157 */
        float rq = or_rcp(projQ[3]);
        float qu = projQ[0] * rq, qv = projQ[1] * rq;                /* :158 */
        float rz = ndc_to_view(cb, or_depth_linear_border(depth, W, H, qu, qv)); /* :164-165 */
        float s = or_div(rz, q[2]);                                  /* :171 */
        float r[3] = { s * q[0], s * q[1], s * q[2] };
        float distZ = p[2] - r[2];                                   /* :185 */
        float d[3] = { r[0] - p[0], r[1] - p[1], r[2] - p[2] }, dn[3];
        or_normalize3(d, dn);
        float dp = or_max0(or_dot3(n, dn), 0.0f);                    /* :186 */
        occlusionSum = fmaf(dp, occlusion_function(cb, distZ), occlusionSum); /* :188-190 */
    }
    occlusionSum = or_div(occlusionSum, 14.0f);                       /* :193 */
    float access = 1.0f - occlusionSum;                               /* :195 */
    /* :198 saturate(pow(access, 6)): the literal-6 power is DEFINED as three multiplies. */
    float a2 = access * access, a4 = a2 * a2, a6 = a4 * a2;
    return or_to_unorm16(a6);
}

void or_ssao(const or_ssao_constants* cb, const uint16_t* normal, const uint32_t* depth, const uint8_t* randvec,
             uint32_t W, uint32_t H, uint16_t* ambient_out, uint32_t row0, uint32_t rows)
{
    uint32_t w2 = W / 2, h2 = H / 2;
    uint32_t row1 = row0 + rows; if (row1 > h2) row1 = h2;
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = (int)row0; y < (int)row1; ++y)
        for (uint32_t x = 0; x < w2; ++x)
            ambient_out[(size_t)y * w2 + x] = ssao_pixel(cb, normal, depth, randvec, W, H, x, (uint32_t)y);
}

/* SsaoBlur.hlsl:85-146 for one half-res pixel. */
static uint16_t blur_pixel(const or_ssao_constants* cb, const uint16_t* normal, const uint32_t* depth,
                           const uint16_t* in, uint32_t W, uint32_t H, int horizontal, int x, int y)
{
    int w2 = (int)(W / 2), h2 = (int)(H / 2);
    const float* blurWeights = &cb->BlurWeights[0][0];                /* :88-93 */
    const int gBlurRadius = 5;                                        /* :48 */
    int dx = horizontal ? 1 : 0, dy = horizontal ? 0 : 1;             /* :95-103 */

    float color = blurWeights[gBlurRadius] * or_unorm16(in[(size_t)y * w2 + x]);  /* :106 */
    float totalWeight = blurWeights[gBlurRadius];                     /* :107 */
    float centerNormal[3];
    normal_point(normal, W, H, x, y, centerNormal);                   /* :109 (not renormalised) */
    float centerDepth = ndc_to_view(cb, depth_at_half_pixel(depth, W, H, x, y)); /* :110-111 */

    for (int i = -gBlurRadius; i <= gBlurRadius; ++i) {               /* :113 */
        if (i == 0) continue;
        int tx = x + i * dx, ty = y + i * dy;                          /* tex = TexC + i*texOffset, in half-res pixels */
        float neighborNormal[3];
        normal_point(normal, W, H, tx, ty, neighborNormal);           /* point/clamp :121 */
        float neighborDepth = ndc_to_view(cb, depth_at_half_pixel(depth, W, H, tx, ty)); /* border :122-123 */
        if (or_dot3(neighborNormal, centerNormal) >= 0.8f && fabsf(neighborDepth - centerDepth) <= 0.2f) { /* :131-132 */
            float weight = blurWeights[i + gBlurRadius];
            int cx = or_clampi(tx, 0, w2 - 1), cy = or_clampi(ty, 0, h2 - 1);       /* point/clamp :137 */
            color = fmaf(weight, or_unorm16(in[(size_t)cy * w2 + cx]), color);
            totalWeight += weight;
        }
    }
    return or_to_unorm16(or_div(color, totalWeight));                 /* :145 */
}

void or_ssao_blur(const or_ssao_constants* cb, const uint16_t* normal, const uint32_t* depth,
                  const uint16_t* ambient_in, uint16_t* ambient_out, uint32_t W, uint32_t H, int horizontal,
                  uint32_t row0, uint32_t rows)
{
    uint32_t w2 = W / 2, h2 = H / 2;
    uint32_t row1 = row0 + rows; if (row1 > h2) row1 = h2;
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = (int)row0; y < (int)row1; ++y)
        for (int x = 0; x < (int)w2; ++x)
            ambient_out[(size_t)y * w2 + x] = blur_pixel(cb, normal, depth, ambient_in, W, H, horizontal, x, y);
}

/* Ssao::ComputeSsao + BlurAmbientMap  Ssao.cpp:185-243: the final AO lives in ambient0. */
void or_compute_ssao(const or_ssao_constants* cb, const uint16_t* normal, const uint32_t* depth,
                     const uint8_t* randvec, uint32_t W, uint32_t H, uint16_t* ambient0, uint16_t* ambient1,
                     int blurCount)
{
    uint32_t h2 = H / 2;
    or_ssao(cb, normal, depth, randvec, W, H, ambient0, 0, h2);
    for (int i = 0; i < blurCount; ++i) {
        or_ssao_blur(cb, normal, depth, ambient0, ambient1, W, H, 1, 0, h2); /* Ssao.cpp:240, 253-258 */
        or_ssao_blur(cb, normal, depth, ambient1, ambient0, W, H, 0, 0, h2); /* Ssao.cpp:241, 260-265 */
    }
}

float or_sample_depth_linear_border(const uint32_t* depth, uint32_t W, uint32_t H, float u, float v)
{
    return or_depth_linear_border(depth, W, H, u, v);
}
void or_sample_randvec(const uint8_t* randvec, float u, float v, float rgb[3]) { or_randvec_linear_wrap(randvec, u, v, rgb); }
float or_half_to_float(uint16_t h) { return or_half_bits_to_float(h); }
float or_d24_to_float(uint32_t d24) { return or_d24(d24); }
