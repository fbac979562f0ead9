/*
 * or_light.c -- oracle restatement of Shaders/DeferredShading.hlsl:PS with PBR.hlsl, GBuffer.hlsl,
 * LightingUtil.hlsl:52-60 and the cascaded-shadow Poisson PCF of Common.hlsl:167-183,263-317
 * (TEST INFRASTRUCTURE, parity unpinned: see crychic_oracle.h).  Reference quirks Q1-Q6, Q13 of
 * SURVEY.md are reproduced on purpose and marked below.
 */
#include "crychic_oracle.h"
#include "or_samplers.h"

#define OR_PI 3.1415926f /* PBR.hlsl:2 */

/* Common.hlsl:167-171.  `noise` is a scalar broadcast to float2, so abs(noise.x + noise.y) * 0.5 == noise. */
static inline float nrand(float u, float v)
{
    float d = fmaf(v, 78.233f * 2.0f, u * (12.9898f * 2.0f));
    float noise = or_frac(or_det_sinf_(d) * 43758.5453f);
    return fabsf(noise + noise) * 0.5f;
}
float or_nrand(float u, float v) { return nrand(u, v); }

/* Common.hlsl:173-183 */
static const float poissonDisk[16][2] = {
    { -0.94201624f, -0.39906216f }, { 0.94558609f, -0.76890725f },
    { -0.094184101f, -0.92938870f }, { 0.34495938f, 0.29387760f },
    { -0.91588581f, 0.45771432f }, { -0.81544232f, -0.87912464f },
    { -0.38277543f, 0.27676845f }, { 0.97484398f, 0.75648379f },
    { 0.44323325f, -0.97511554f }, { 0.53742981f, -0.47373420f },
    { -0.26496911f, -0.41893023f }, { 0.79197514f, 0.19090188f },
    { -0.24188840f, 0.99706507f }, { -0.81409955f, 0.91437590f },
    { 0.19984126f, 0.78641367f }, { 0.14383161f, -0.14100790f }
};

float or_pcf_search_radius(uint32_t width, int literal)
{
    /* Common.hlsl:305 `float search_radius = 5 / width / 2.0f;` with `uint width`: 5 / width is an
     * unsigned integer division (quirk Q2). */
    if (literal) return (float)(5u / width) / 2.0f;
    return 5.0f / (float)width / 2.0f;        /* evaluated on the host and handed to the pass as a constant */
}

/* Common.hlsl:263-317 */
static float pcf_poisson(const uint32_t* shadow, uint32_t dim, const float sp[4], float search_radius)
{
    float rw = or_rcp(sp[3]);
    float x = sp[0] * rw, y = sp[1] * rw, depth = sp[2] * rw;           /* :266-269 */
    float theta = nrand(x, y);                                          /* :301 */
    float cos_theta = or_det_cosf_(theta);
    float sin_theta = or_det_sinf_(theta);
    float percentLit = 0.0f;
    for (int i = 0; i < 16; ++i) {                                      /* N_SAMPLE :21, :308 */
        /* mul(poissonDisk[i], float2x2(c, s, -s, c))  :304,310 */
        float px = fmaf(poissonDisk[i][1], -sin_theta, poissonDisk[i][0] * cos_theta);
        float py = fmaf(poissonDisk[i][1], cos_theta, poissonDisk[i][0] * sin_theta);
        percentLit += or_shadow_cmp_linear(shadow, dim, fmaf(px, search_radius, x), fmaf(py, search_radius, y), depth); /* :311-313 */
    }
    return percentLit * 0.0625f;                                        /* :315 (/ 16 is exact) */
}
float or_pcf_poisson(const uint32_t* shadow, uint32_t dim, const float shadowPosH[4], float searchRadius)
{
    return pcf_poisson(shadow, dim, shadowPosH, searchRadius);
}
float or_sample_shadow_cmp(const uint32_t* shadow, uint32_t dim, float u, float v, float ref)
{
    return or_shadow_cmp_linear(shadow, dim, u, v, ref);
}

static inline float pow5(float x) { float x2 = x * x, x4 = x2 * x2; return x4 * x; }

/* PBR.hlsl:4-14 */
static inline float ndf_ggx(const float n[3], const float h[3], float a)
{
    float a2 = a * a;
    float nDoth = or_max0(or_dot3(n, h), 0.001f);
    float nDoth2 = nDoth * nDoth;
    float t = fmaf(nDoth2, a2 - 1.0f, 1.0f);
    float tmp = t * t;                      /* pow(x, 2) */
    float bottom = OR_PI * tmp;
    return a2 * or_rcp(bottom);             /* top * rcp(bottom) */
}
/* PBR.hlsl:16-21 */
static inline float geometry_schlick_ggx(float nDotvec, float k) { return or_div(nDotvec, fmaf(nDotvec, 1.0f - k, k)); }

/* One directional light: PBR.hlsl:72-88 (GetPBRDesc), :45-70 (GetBRDF), :99-106 (PBRShading loop body). */
#define OR_FIX_Q1 0x100 /* cascade blend only within 5 units of the cascade radius (Default.hlsl:131's form) */
#define OR_FIX_Q3 0x200 /* specular denominator nDotl * nDotv */
#define OR_FIX_Q4 0x400 /* brdf = kd * fd + fs */

static void pbr_light(const float lightDir[3], const float strength[3], const float albedo[3], float roughness, float metalness,
                      const float normal[3], const float view[3], float shadowTerm, int flags, float result[3])
{
    float vl[3] = { view[0] + lightDir[0], view[1] + lightDir[1], view[2] + lightDir[2] }, halfVec[3];
    or_normalize3(vl, halfVec);
    float hDotv = or_max0(or_dot3(halfVec, view), 0.001f);
    float nDotl = or_max0(or_dot3(normal, lightDir), 0.001f);
    float nDotv = or_max0(or_dot3(normal, view), 0.001f);

    float nDotvQ = hDotv;                    /* quirk Q3: PBR.hlsl:58 `float nDotv = pbrDesc.hDotv;` */
    float D = ndf_ggx(normal, halfVec, roughness);
    float fr = pow5(or_saturate(1.0f - nDotvQ)); /* FresnelSchlick :40-43 */
    float k = 0.125f * (roughness + 1.0f) * (roughness + 1.0f);
    float G = geometry_schlick_ggx(nDotv, k) * geometry_schlick_ggx(nDotl, k); /* true nDotv :29-38 */
    float s5 = shadowTerm;
    /* PBR.hlsl:66 `/ (nDotl * nDotv)` with the local nDotv = hDotv (Q3), the same quotient for every channel */
    float rdenom = or_rcp(nDotl * ((flags & OR_FIX_Q3) ? nDotv : nDotvQ));
    for (int c = 0; c < 3; ++c) {
        float f0 = or_lerp(0.04f, albedo[c], metalness);
        float F = fmaf(1.0f - f0, fr, f0);
        float fs = 0.25f * D * G * F;
        fs = fs * rdenom;
        float fd = albedo[c] * (1.0f / OR_PI);
        float ks = F;                         /* quirk Q4: F applied twice */
        float kd = (1.0f - F) * (1.0f - metalness);
        float brdf = (flags & OR_FIX_Q4) ? kd * fd + fs : fmaf(ks, fs, kd * fd);
        float irradiance = strength[c] * nDotl;
        result[c] = fmaf(s5 * brdf, irradiance, result[c]);
    }
}

static void pbr_dir_light(const or_light* L, const float albedo[3], float roughness, float metalness,
                          const float normal[3], const float view[3], float shadow, int flags, float result[3])
{
    float lightDir[3] = { -L->Direction[0], -L->Direction[1], -L->Direction[2] };   /* PBR.hlsl:101 */
    pbr_light(lightDir, L->Strength, albedo, roughness, metalness, normal, view, pow5(shadow) /* :105 */, flags, result);
}

/* BUILD-DEFINED EXTENSION (BASELINE configs[4], parity unpinned): the reference's point-light branch (PBR.hlsl:109-124) does
 * not compile (`pbr.nDotl`, :117) and its accumulation is commented out (:122).  The extension enables it as evidently
 * intended: l = Position - pos, d = |l|, range test d > FalloffEnd -> no contribution (LightingUtil.hlsl:104-105 and the
 * spot branch :133-137), l /= d, BRDF as for directional lights, strength * nDotl * CalcAttenuation, shadowFactor 1. */
static void pbr_point_light(const or_light* L, const float pos[3], const float albedo[3], float roughness, float metalness,
                            const float normal[3], const float view[3], int flags, float result[3])
{
    float l[3] = { L->Position[0] - pos[0], L->Position[1] - pos[1], L->Position[2] - pos[2] };
    float d = or_len(or_dot3(l, l));
    if (d > L->FalloffEnd) return;
    float rd = or_rcp(d);
    float ln[3] = { l[0] * rd, l[1] * rd, l[2] * rd };
    float att = or_saturate(or_div(L->FalloffEnd - d, L->FalloffEnd - L->FalloffStart));   /* CalcAttenuation, LightingUtil.hlsl:44-48 */
    /* GetPBRDesc / GetBRDF exactly as for a directional light (PBR.hlsl:72-88, 45-70), with l as the light direction */
    float vl[3] = { view[0] + ln[0], view[1] + ln[1], view[2] + ln[2] }, halfVec[3];
    or_normalize3(vl, halfVec);
    float hDotv = or_max0(or_dot3(halfVec, view), 0.001f);
    float nDotl = or_max0(or_dot3(normal, ln), 0.001f);
    float nDotv = or_max0(or_dot3(normal, view), 0.001f);
    float nDotvQ = hDotv;
    float D = ndf_ggx(normal, halfVec, roughness);
    float fr = pow5(or_saturate(1.0f - nDotvQ));
    float k = 0.125f * (roughness + 1.0f) * (roughness + 1.0f);
    float G = geometry_schlick_ggx(nDotv, k) * geometry_schlick_ggx(nDotl, k);
    float rdenom = or_rcp(nDotl * ((flags & OR_FIX_Q3) ? nDotv : nDotvQ));
    for (int c = 0; c < 3; ++c) {
        float f0 = or_lerp(0.04f, albedo[c], metalness);
        float F = fmaf(1.0f - f0, fr, f0);
        float fs = 0.25f * D * G * F;
        fs = fs * rdenom;
        float fd = albedo[c] * (1.0f / OR_PI);
        float kd = (1.0f - F) * (1.0f - metalness);
        float brdf = (flags & OR_FIX_Q4) ? kd * fd + fs : fmaf(F, fs, kd * fd);
        float lightStrength = L->Strength[c] * nDotl;      /* PBR.hlsl:118 */
        lightStrength = lightStrength * att;               /* :120 */
        result[c] = fmaf(1.0f * brdf, lightStrength, result[c]);   /* :122 with shadowFactor[i] = 1 */
    }
}

#define OR_CUBE_LEVELS(flags) (((uint32_t)(flags) >> 16) & 15u)   /* the cube map's mip levels (0 and 1: level 0 alone): CRYCHIC_LIGHT_CUBE_LEVELS */

static void cube4(const uint8_t* cube, uint32_t dim, const float dir[3], float rgba[4])
{
    or_cube_linear(cube, dim, dir, rgba, 4);
}

/* DeferredShading.hlsl:32,94 and GBuffer.hlsl:41 for the pixel at idx: r = reflect(-view, normalW) -- what light_pixel looks the
 * cube map up with, recomputed for the pixel's quad neighbours when the cube map has a mip chain. */
static void pixel_reflection(const or_pass_constants* cb, const float* g0, const float* g2, size_t idx, float r[3])
{
    const float* G0 = g0 + idx * 4; const float* G2 = g2 + idx * 4;
    float nraw[3] = { G2[0], G2[1], G2[2] }, normalW[3], view[3];
    or_normalize3(nraw, normalW);
    float toEye[3] = { cb->EyePosW[0] - G0[0], cb->EyePosW[1] - G0[1], cb->EyePosW[2] - G0[2] };
    or_normalize3(toEye, view);
    float negv[3] = { -view[0], -view[1], -view[2] };
    or_reflect3(negv, normalW, r);
}
static int covered_at(const uint32_t* depth, uint32_t W, uint32_t H, uint32_t x, uint32_t y)
{
    return x < W && y < H && (depth[(size_t)y * W + x] & 0x00FFFFFFu) < 0x00FFFFFFu;
}
/* The implicit derivatives of gCubeMap.Sample in the lighting pass (oracle definition, or_samplers.h): differences inside the
 * pixel's 2 x 2 quad (quads start on even pixel coordinates), right - left in the pixel's row and lower - upper in its column; a
 * neighbour that the pass does not shade (no geometry there, or outside the frame) contributes a zero derivative. */
static float reflection_lod(const or_pass_constants* cb, const float* g0, const float* g2, const uint32_t* depth, uint32_t cubeDim,
                            uint32_t levels, uint32_t W, uint32_t H, uint32_t x, uint32_t y, const float r[3])
{
    float ddx[3] = { 0.0f, 0.0f, 0.0f }, ddy[3] = { 0.0f, 0.0f, 0.0f }, n[3];
    if (covered_at(depth, W, H, x ^ 1u, y)) {
        pixel_reflection(cb, g0, g2, (size_t)y * W + (x ^ 1u), n);
        for (int c = 0; c < 3; ++c) ddx[c] = (x & 1u) ? r[c] - n[c] : n[c] - r[c];
    }
    if (covered_at(depth, W, H, x, y ^ 1u)) {
        pixel_reflection(cb, g0, g2, (size_t)(y ^ 1u) * W + x, n);
        for (int c = 0; c < 3; ++c) ddy[c] = (y & 1u) ? r[c] - n[c] : n[c] - r[c];
    }
    return or_cube_lod(cubeDim, levels, r, ddx, ddy);
}

static void light_pixel(const or_pass_constants* cb, const float* g0, const float* g1, const float* g2,
                        const uint16_t* ambient, const uint32_t* const shadow[4], uint32_t shadowDim,
                        const uint8_t* cube, uint32_t cubeDim, uint32_t W, uint32_t H, size_t idx,
                        int numDirLights, float pcfRadius, const or_light* pointLights, uint32_t numPointLights, int flags,
                        const uint32_t* depth, float lit[4])
{
    /* DeferredShading.hlsl:25-30: the anisotropic-wrap fetch at exact texel centres is the texel itself. */
    const float* G0 = g0 + idx * 4; const float* G1 = g1 + idx * 4; const float* G2 = g2 + idx * 4;
    float posW[3] = { G0[0], G0[1], G0[2] };          /* GBuffer.hlsl:37 */
    float metalness = G0[3];                          /* :38 (quirk Q5: always 0.5 in the reference scene) */
    float albedo[3] = { G1[0], G1[1], G1[2] };        /* :39 */
    float roughness = G1[3];                          /* :40 */
    float nraw[3] = { G2[0], G2[1], G2[2] }, normalW[3];
    or_normalize3(nraw, normalW);                     /* :41; G3 and G2.w ignored (Q13) */

    float toEye[3] = { cb->EyePosW[0] - posW[0], cb->EyePosW[1] - posW[1], cb->EyePosW[2] - posW[2] };
    float view[3];
    or_normalize3(toEye, view);                       /* DeferredShading.hlsl:32 */
    float fresnelR0[3];
    for (int c = 0; c < 3; ++c) fresnelR0[c] = or_lerp(0.04f, albedo[c], metalness); /* :35 */

    float pos4[4] = { posW[0], posW[1], posW[2], 1.0f };
    float ambientAccess = 1.0f;
    if (ambient) {
        float sp[4];
        or_mul_v4_m(pos4, cb->ViewProjTex, sp);       /* :40 */
        float rw = or_rcp(sp[3]);
        ambientAccess = or_ambient_linear_clamp(ambient, W / 2, H / 2, sp[0] * rw, sp[1] * rw); /* :41-42 */
    }
    float amb[4];
    for (int c = 0; c < 3; ++c) amb[c] = ambientAccess * cb->AmbientLight[c] * albedo[c]; /* :44 */
    amb[3] = ambientAccess * cb->AmbientLight[3] * 1.0f;

    float shadowFactors[OR_MAX_LIGHTS];
    for (int i = 0; i < OR_MAX_LIGHTS; ++i) shadowFactors[i] = 1.0f;  /* :46-51 */

    static const float radius[4] = { 30.0f, 50.0f, 80.0f, 100.0f };  /* :53 */
    float distance = or_len(or_dot3(toEye, toEye));                   /* :57 length(gEyePosW - posW) */
    for (int j = 0; j < 4; ++j) {
        /* :60 `abs(distance - radius[j] < 5.0f)` is abs() of a bool (quirk Q1): it is 1 whenever
         * distance < radius[j], so the blend branch is taken for every j < 3. */
        int blendTerm = (distance - radius[j] < 5.0f) ? 1 : 0;
        if (flags & OR_FIX_Q1) blendTerm = fabsf(distance - radius[j]) < 5.0f;   /* the intended test, Default.hlsl:131 */
        if (j < 3 && distance < radius[j] && blendTerm != 0) {
            float sp0[4], sp1[4];
            or_mul_v4_m(pos4, cb->ShadowTransforms[j], sp0);          /* :62 */
            or_mul_v4_m(pos4, cb->ShadowTransforms[j + 1], sp1);      /* :63 */
            float a = pcf_poisson(shadow[j], shadowDim, sp0, pcfRadius);
            float b = pcf_poisson(shadow[j + 1], shadowDim, sp1, pcfRadius);
            shadowFactors[0] = 0.5f * (a + b);                         /* :66 */
            break;
        } else if (distance < radius[j]) {
            float sp0[4];
            or_mul_v4_m(pos4, cb->ShadowTransforms[j], sp0);          /* :71 */
            shadowFactors[0] = pcf_poisson(shadow[j], shadowDim, sp0, pcfRadius); /* :72-73 */
            break;
        }
    }

    const float shininess = (1.0f - roughness) * 1.0f;                /* :84, normalW.a = 1 */

    float direct[3] = { 0.0f, 0.0f, 0.0f };
    for (int i = 0; i < numDirLights; ++i)                            /* PBR.hlsl:99-106; NUM_DIR_LIGHTS (Q6) */
        pbr_dir_light(&cb->Lights[i], albedo, roughness, metalness, normalW, view, shadowFactors[i], flags, direct);
    for (uint32_t i = 0; i < numPointLights; ++i)                     /* extension: NUM_POINT_LIGHTS lights from a separate buffer */
        pbr_point_light(&pointLights[i], posW, albedo, roughness, metalness, normalW, view, flags, direct);
    for (int c = 0; c < 3; ++c) {
        float d = or_div(direct[c], direct[c] + 1.0f);                /* :89 */
        d = or_pow_inv_gamma(d);                                      /* :90  pow(d, 1 / 2.2) */
        lit[c] = d + amb[c];                                          /* :92 */
    }
    /* w: directLight.w = 0 -> 0/(0+1) = 0 -> pow(0, 1/2.2) = 0 -> + ambient.a; then :99 overwrites with 1 */

    float negv[3] = { -view[0], -view[1], -view[2] }, r[3];
    or_reflect3(negv, normalW, r);                                    /* :94 */
    float refl[4];
    if (OR_CUBE_LEVELS(flags) > 1u) {                                 /* :95 with the chain bound (CRYCHIC.cpp:1148-1151) */
        uint32_t x = (uint32_t)(idx % W), y = (uint32_t)(idx / W);
        float lod = reflection_lod(cb, g0, g2, depth, cubeDim, OR_CUBE_LEVELS(flags), W, H, x, y, r);
        or_cube_trilinear(cube, cubeDim, OR_CUBE_LEVELS(flags), r, lod, refl, 4);
    } else
        cube4(cube, cubeDim, r, refl);                                /* :95 */
    float cosI = or_saturate(or_dot3(normalW, r));                    /* LightingUtil.hlsl:54 */
    float f0 = 1.0f - cosI;
    float f5 = f0 * f0 * f0 * f0 * f0;                                /* :57 */
    for (int c = 0; c < 3; ++c) {
        float fresnel = fmaf(1.0f - fresnelR0[c], f5, fresnelR0[c]);
        lit[c] = fmaf(shininess * fresnel, refl[c], lit[c]);          /* DeferredShading.hlsl:97 */
    }
    lit[3] = 1.0f;                                                    /* :99 */
}

/* Shaders/sky.hlsl:21-47: cubemap lookup along the view ray of the pixel (the sky sphere is centred on
 * the eye, so the interpolated PosL is parallel to the ray).  Ray = near-plane point of the pixel in view
 * space rotated by InvView. */
static void sky_direction(const or_pass_constants* cb, uint32_t W, uint32_t H, uint32_t x, uint32_t y, float dw[4])
{
    float u = or_div((float)x + 0.5f, (float)W), v = or_div((float)y + 0.5f, (float)H);
    float posh[4] = { fmaf(2.0f, u, -1.0f), fmaf(-2.0f, v, 1.0f), 0.0f, 1.0f }, ph[4];
    or_mul_v4_m(posh, cb->InvProj, ph);
    float rw = or_rcp(ph[3]);
    float pv[4] = { ph[0] * rw, ph[1] * rw, ph[2] * rw, 0.0f };
    or_mul_v4_m(pv, cb->InvView, dw);
}
static void sky_pixel(const or_pass_constants* cb, const uint8_t* cube, uint32_t cubeDim, uint32_t levels, uint32_t W, uint32_t H,
                      uint32_t x, uint32_t y, float out[4])
{
    float dw[4];
    sky_direction(cb, W, H, x, y, dw);
    if (levels > 1u) {
        /* sky.hlsl:46 with the chain: the quad neighbours' directions are those of their own pixels (the sky sphere covers the
         * whole quad, its far pixels running as helpers); outside the frame: zero derivative */
        float ddx[3] = { 0.0f, 0.0f, 0.0f }, ddy[3] = { 0.0f, 0.0f, 0.0f }, n[4];
        if ((x ^ 1u) < W) { sky_direction(cb, W, H, x ^ 1u, y, n); for (int c = 0; c < 3; ++c) ddx[c] = (x & 1u) ? dw[c] - n[c] : n[c] - dw[c]; }
        if ((y ^ 1u) < H) { sky_direction(cb, W, H, x, y ^ 1u, n); for (int c = 0; c < 3; ++c) ddy[c] = (y & 1u) ? dw[c] - n[c] : n[c] - dw[c]; }
        or_cube_trilinear(cube, cubeDim, levels, dw, or_cube_lod(cubeDim, levels, dw, ddx, ddy), out, 4);
        return;
    }
    cube4(cube, cubeDim, dw, out);
}

void or_deferred_light(const or_pass_constants* cb, const float* g0, const float* g1, const float* g2,
                       const uint32_t* depth, const uint16_t* ambient, const uint32_t* const shadow[4],
                       uint32_t shadowDim, const uint8_t* cube, uint32_t cubeDim, uint8_t* out_rgba8,
                       float* radiance_out, uint32_t W, uint32_t H, uint32_t row0, uint32_t rows,
                       int numDirLights, float pcfSearchRadius, int sky)
{
    or_deferred_light_points(cb, g0, g1, g2, depth, ambient, shadow, shadowDim, cube, cubeDim, out_rgba8, radiance_out, W, H, row0, rows,
                             numDirLights, pcfSearchRadius, sky, NULL, 0);
}

void or_deferred_light_points(const or_pass_constants* cb, const float* g0, const float* g1, const float* g2,
                              const uint32_t* depth, const uint16_t* ambient, const uint32_t* const shadow[4],
                              uint32_t shadowDim, const uint8_t* cube, uint32_t cubeDim, uint8_t* out_rgba8,
                              float* radiance_out, uint32_t W, uint32_t H, uint32_t row0, uint32_t rows,
                              int numDirLights, float pcfSearchRadius, int sky, const or_light* pointLights, uint32_t numPointLights)
{
    uint32_t row1 = row0 + rows; if (row1 > H) row1 = H;
    /* Colors::LightSteelBlue (CRYCHIC.cpp:247) */
    static const float clearColor[4] = { 0.690196097f, 0.768627524f, 0.870588303f, 1.0f };
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = (int)row0; y < (int)row1; ++y) {
        for (uint32_t x = 0; x < W; ++x) {
            size_t idx = (size_t)y * W + x;
            float lit[4];
            /* Coverage: the deferred draw re-rasterises the opaque geometry against a depth buffer cleared
             * to 1.0 with LESS (CRYCHIC.cpp:248,273), i.e. exactly the pixels whose normal/depth pass depth
             * is below the clear value. */
            if ((depth[idx] & 0x00FFFFFFu) < 0x00FFFFFFu)
                light_pixel(cb, g0, g1, g2, ambient, shadow, shadowDim, cube, cubeDim, W, H, idx, numDirLights,
                            pcfSearchRadius, pointLights, numPointLights, sky, depth, lit);
            else if (sky & 1)
                sky_pixel(cb, cube, cubeDim, OR_CUBE_LEVELS(sky), W, H, x, (uint32_t)y, lit);
            else
                for (int c = 0; c < 4; ++c) lit[c] = clearColor[c];
            if (radiance_out) for (int c = 0; c < 4; ++c) radiance_out[idx * 4 + c] = lit[c];
            for (int c = 0; c < 4; ++c) out_rgba8[idx * 4 + c] = or_to_unorm8(lit[c]);
        }
    }
}

void or_sample_cube(const uint8_t* cube, uint32_t dim, const float dir[3], float rgb[3]) { or_cube_linear(cube, dim, dir, rgb, 3); }
float or_sample_cube_lod(uint32_t dim, uint32_t levels, const float dir[3], const float ddx[3], const float ddy[3]) { return or_cube_lod(dim, levels, dir, ddx, ddy); }
void or_sample_cube_level(const uint8_t* chain, uint32_t dim, uint32_t levels, const float dir[3], float lod, float rgb[3])
{
    or_cube_trilinear(chain, dim, levels, dir, lod, rgb, 3);
}
float or_sample_ambient_linear_clamp(const uint16_t* ambient, uint32_t w2, uint32_t h2, float u, float v)
{
    return or_ambient_linear_clamp(ambient, w2, h2, u, v);
}
float or_det_sinf(float x) { return or_det_sinf_(x); }
float or_det_cosf(float x) { return or_det_cosf_(x); }
float or_det_log2f(float x) { return or_det_log2f_(x); }
float or_det_exp2f(float x) { return or_det_exp2f_(x); }
float or_det_powf(float x, float y) { return or_det_powf_(x, y); }

/* Batch evaluation of the scalar definitions (unit tests compare them with the product's device math compiled
 * for the host).  kind: 0 sin, 1 cos, 2 log2, 3 exp2, 4 pow(in, in2), 5 nrand(in, in2), 6 d24 decode (in = bits),
 * 7 unorm16 decode, 8 unorm8 decode, 9 half decode, 10 pow(in, 1/2.2) of the tone map. */
void or_eval_array(int kind, size_t n, const float* in, const float* in2, float* out)
{
    const uint32_t* bits = (const uint32_t*)in;
    for (size_t i = 0; i < n; ++i) {
        switch (kind) {
        case 0: out[i] = or_det_sinf_(in[i]); break;
        case 1: out[i] = or_det_cosf_(in[i]); break;
        case 2: out[i] = or_det_log2f_(in[i]); break;
        case 3: out[i] = or_det_exp2f_(in[i]); break;
        case 4: out[i] = or_det_powf_(in[i], in2[i]); break;
        case 5: out[i] = nrand(in[i], in2[i]); break;
        case 6: out[i] = or_d24(bits[i]); break;
        case 7: out[i] = or_unorm16((uint16_t)bits[i]); break;
        case 8: out[i] = or_unorm8((uint8_t)bits[i]); break;
        case 9: out[i] = or_half_bits_to_float((uint16_t)bits[i]); break;
        case 10: out[i] = or_pow_inv_gamma(in[i]); break;
        default: out[i] = 0.0f;
        }
    }
}
