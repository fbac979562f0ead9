// light_core.hpp -- per-pixel body of the deferred lighting kernel: Shaders/DeferredShading.hlsl:23-101 with
// GBuffer.hlsl:33-43, PBR.hlsl:4-107, LightingUtil.hlsl:52-60 and the rotated-Poisson cascade PCF of
// Common.hlsl:167-183,263-317.  The reference's quirks are reproduced, not fixed.
#pragma once
#include "devmath.hpp"
#include "crychic_hip.h"

namespace cry {

// The subset of cbPass (Common.hlsl:82-107) the pixel shader reads, in the reference's transposed layout.
struct LightParams {
    float ViewProjTex[16];
    float ShadowTransforms[4][16];
    float InvProj[16];
    float InvView[16];
    float EyePosW[3];
    float pcfSearchRadius;
    float AmbientLight[4];
    crychic_light Lights[CRYCHIC_MAX_LIGHTS];
    const uint32_t* shadow[4];
    uint32_t shadowDim;
    uint32_t cubeDim;
    uint32_t W, H;
    int numDirLights;
    uint32_t flags;
    const crychic_light* pointLights;   // extension (BASELINE configs[4]): NUM_POINT_LIGHTS lights in a device buffer
    uint32_t numPointLights;
    uint32_t shadowWIsOne;     // light_shadow_w_is_one(): every cascade's ShadowTransform has the w column (0, 0, 0, 1)
    uint32_t darkLights;       // light_dark_mask(): bit i = directional light i has Strength (0, 0, 0) and a sane direction
    float rcpW, rcpH;          // rcp((float)W), rcp((float)H)
    uint32_t unitLights;       // light_dark_lengths_ok(): every directional light in use has a finite direction no longer than 1.001
    // Interleaved copies of rows the kernels evaluate two at a time in packed fp32 (wave-uniform operands have to be adjacent
    // scalar registers for v_pk_fma_f32): ShadowPairs[J][k] = { ShadowTransforms[J][k], ShadowTransforms[J + 1][k] } for the x, y
    // and z rows (k < 12) of cascades J and J + 1, which every pixel nearer than 80 blends (cascade_fetch_uniform);
    // ScreenPairs[k] = { ViewProjTex[k], ViewProjTex[4 + k] }, the x and y rows of the projection to the ambient map.
    float ShadowPairs[3][12][2];
    float ScreenPairs[4][2];
    float halfDims[2];         // (float)(W / 2), (float)(H / 2)
    uint32_t cubeLevels;       // mip levels the cube map holds (CRYCHIC_LIGHT_CUBE_LEVELS; 0 / 1 = level 0 alone)
};
CRY_HD void light_params_derive(LightParams& P)
{
    for (int J = 0; J < 3; ++J)
        for (int k = 0; k < 12; ++k) { P.ShadowPairs[J][k][0] = P.ShadowTransforms[J][k]; P.ShadowPairs[J][k][1] = P.ShadowTransforms[J + 1][k]; }
    for (int k = 0; k < 4; ++k) { P.ScreenPairs[k][0] = P.ViewProjTex[k]; P.ScreenPairs[k][1] = P.ViewProjTex[4 + k]; }
    P.halfDims[0] = (float)(P.W / 2);
    P.halfDims[1] = (float)(P.H / 2);
}

constexpr uint32_t kMaxPointLights = 1024;   // tile masks live in LDS: 32 words

// Common.hlsl:167-171 (noise is a scalar broadcast, so abs(noise.x + noise.y) * 0.5 == noise)
CRY_HD float nrand(float u, float v)
{
    float d = fma(v, 78.233f * 2.0f, u * (12.9898f * 2.0f));
    float s = det_sin(d) * 43758.5453f;
    float noise = s - __builtin_floorf(s);
    return __builtin_fabsf(noise + noise) * 0.5f;
}

// Texture lookups come in two halves -- *_fetch computes the addresses and issues the loads, *_resolve decodes and filters --
// so that a caller can issue every gather of a pixel (ambient map, shadow cascades, cubemap) before it waits for the first:
// one memory round trip per pixel instead of one per lookup.
//
// gsamShadow: LESS_EQUAL comparison on each texel, then bilinear; BORDER colour 0  (CRYCHIC.cpp:2649-2658)
struct ShadowFetch { TexelPair p0, p1; Bilin b; };
template <bool FINITE = false>
CRY_HD ShadowFetch shadow_fetch(const uint32_t* __restrict__ s, uint32_t dim, float u, float v)
{
    ShadowFetch f;
    f.b = bilinear_setup<FINITE>(u, v, dim, dim);
    const uint32_t r0 = (uint32_t)clampi(f.b.j0, 0, (int)dim - 1), r1 = (uint32_t)clampi(f.b.j0 + 1, 0, (int)dim - 1);
    f.p0 = pair_at(s, r0, dim, f.b.i0);   // one 8-byte load per footprint row
    f.p1 = pair_at(s, r1, dim, f.b.i0);
    return f;
}
// WAVE_CHECK: let a wavefront whose footprints all lie inside the map (everything but the rim of a cascade) drop the four
// BORDER selects -- for callers with one lookup at a time; inside the 16-tap loop the branch would separate the taps' loads.
template <bool WAVE_CHECK = false>
CRY_HD float shadow_resolve(const ShadowFetch& f, uint32_t dim, float ref)
{
    const Bilin& b = f.b;
    bool xa = (uint32_t)b.i0 < dim, xb = (uint32_t)(b.i0 + 1) < dim;
    bool y0 = (uint32_t)b.j0 < dim, y1 = (uint32_t)(b.j0 + 1) < dim;
#if defined(__HIP_DEVICE_COMPILE__)
    if (WAVE_CHECK && __builtin_amdgcn_ballot_w64(!(xa & xb & y0 & y1)) == 0) xa = xb = y0 = y1 = true;
#endif
    // the BORDER colour 0 is D24 0: select on the integer texel, then decode unconditionally
    const float t00 = d24_to_float((xa && y0) ? f.p0.a : 0u);
    const float t10 = d24_to_float((xb && y0) ? f.p0.b : 0u);
    const float t01 = d24_to_float((xa && y1) ? f.p1.a : 0u);
    const float t11 = d24_to_float((xb && y1) ? f.p1.b : 0u);
    const float c00 = (ref <= t00) ? 1.0f : 0.0f, c10 = (ref <= t10) ? 1.0f : 0.0f;
    const float c01 = (ref <= t01) ? 1.0f : 0.0f, c11 = (ref <= t11) ? 1.0f : 0.0f;
    return bilerp(c00, c10, c01, c11, b.fx, b.fy);
}
CRY_HD float shadow_cmp_linear(const uint32_t* __restrict__ s, uint32_t dim, float u, float v, float ref)
{
    return shadow_resolve(shadow_fetch(s, dim, u, v), dim, ref);
}

// Common.hlsl:173-183, interleaved x,y
#define CRY_POISSON_TABLE                                                                                         \
    { -0.94201624f, -0.39906216f, 0.94558609f, -0.76890725f, -0.094184101f, -0.92938870f, 0.34495938f, 0.29387760f, \
      -0.91588581f, 0.45771432f, -0.81544232f, -0.87912464f, -0.38277543f, 0.27676845f, 0.97484398f, 0.75648379f,   \
      0.44323325f, -0.97511554f, 0.53742981f, -0.47373420f, -0.26496911f, -0.41893023f, 0.79197514f, 0.19090188f,   \
      -0.24188840f, 0.99706507f, -0.81409955f, 0.91437590f, 0.19984126f, 0.78641367f, 0.14383161f, -0.14100790f }

// Common.hlsl:305 as written (uint division) gives radius 0: every one of the 16 taps is uv + (+-0) == uv, so one filtered
// fetch `tap` is accumulated 16 times and divided by 16 (:308-315) -- bit-identical to the loop.
CRY_HD float pcf_zero_radius(float tap)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // fully lit / fully shadowed footprints (all but the shadow edges): 16 additions of 1.0 or +0.0 are exact, and so is the
    // division by 16 that follows -- the result is the tap itself.  Wave-uniform, so the edges still run the literal loop.
    if (__builtin_amdgcn_ballot_w64(!((tap == 0.0f) | (tap == 1.0f))) == 0) return tap;
#endif
    float percentLit = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) percentLit += tap;
    return percentLit * 0.0625f;
}

// The comparison of gsamShadow in the integer domain.  d24_to_float is non-decreasing in the texel, so `ref <= d24_to_float(t)` is
// `t >= T(ref)` with T(ref) = the number of D24 values that decode below ref (2^24 if all do: no texel passes; NaN ref: the same,
// as NaN <= x is false).  ref * (2^24 - 1), truncated, is within [-3, +2] of T (decode rounds by at most half a D24 step, the
// product by at most one), so T is that estimate minus 2 plus the number of the six candidates there that decode below ref --
// six decodes once per lookup position instead of four per tap.  Exhaustively checked against the definition on the host
// (tests/test_devmath_host.py::test_d24_threshold).
CRY_HD uint32_t d24_threshold(float ref)
{
    if (!(ref == ref)) return 0x01000000u;
    const float rc = __builtin_fminf(__builtin_fmaxf(ref, 0.0f), 1.0f);
    const int t0 = (int)(rc * 16777215.0f) - 2;
    const uint32_t base = t0 < 0 ? 0u : (uint32_t)t0;
    uint32_t T = base;
#pragma unroll
    for (uint32_t k = 0; k < 6u; ++k) {
        const uint32_t t = base + k;
        T += (t <= 0x00FFFFFFu && d24_to_float(t) < ref) ? 1u : 0u;
    }
    return T;
}
// The 16 rotated Poisson taps of one cascade lookup (Common.hlsl:301-315) when every tap's footprint lies inside the map and all
// coordinates are finite (pcf_taps_inside, wave-uniform at the caller): no BORDER selects, no NaN handling, the four compares of
// a footprint on the raw texels against d24_threshold(depth).  Same taps, same filter weights, same accumulation order as the
// general loop of pcf_poisson, hence the same bits.  (The two coordinates of a tap as one packed pair: measured 6 % slower.)
CRY_HD bool pcf_taps_inside(uint32_t dim, float x, float y, float depth, float radius)
{
    const float fd = (float)dim, reach = fma(1.42f * radius, fd, 1.0f);      // |rotated poissonDisk[i]| <= 1.415 (points in [-1, 1]^2)
    const float tx = fma(x, fd, -0.5f), ty = fma(y, fd, -0.5f);
    return (tx - reach >= 1.0f) & (tx + reach <= fd - 3.0f) & (ty - reach >= 1.0f) & (ty + reach <= fd - 3.0f) & (__builtin_fabsf(depth) < 1.0e30f) &
           (radius < 64.0f);
}
CRY_HD float pcf_poisson_inside(const uint32_t* __restrict__ s, uint32_t dim, float x, float y, float depth, float radius)
{
    const float theta = nrand(x, y);                          // :301
    const float c = det_cos(theta), sn = det_sin(theta);      // :302-303
    const uint32_t T = d24_threshold(depth);
    const float fd = (float)dim;
    const float P[32] = CRY_POISSON_TABLE;
    float percentLit = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {                            // :308
        const float px = fma(P[2 * i + 1], -sn, P[2 * i] * c); // mul(poissonDisk[i], float2x2(c, s, -s, c))
        const float py = fma(P[2 * i + 1], c, P[2 * i] * sn);
        const float tx = fma(fma(px, radius, x), fd, -0.5f), ty = fma(fma(py, radius, y), fd, -0.5f);     // bilinear_setup, in range
        const float flx = __builtin_floorf(tx), fly = __builtin_floorf(ty);
        const float fx = tx - flx, fy = ty - fly;
        const uint32_t t0 = mul24((uint32_t)(int)fly, dim) + (uint32_t)(int)flx;
        const RawPair r0 = load_pair(s, t0), r1 = load_pair(s, t0 + dim);
        const float c00 = (r0.lo & 0x00FFFFFFu) >= T ? 1.0f : 0.0f, c10 = (r0.hi & 0x00FFFFFFu) >= T ? 1.0f : 0.0f;
        const float c01 = (r1.lo & 0x00FFFFFFu) >= T ? 1.0f : 0.0f, c11 = (r1.hi & 0x00FFFFFFu) >= T ? 1.0f : 0.0f;
        percentLit += bilerp(c00, c10, c01, c11, fx, fy);     // :311-313
    }
    return percentLit * 0.0625f;                              // :315
}

// CalcCascadeShadowFactorWithPoisson  Common.hlsl:263-317
// ZERO_RADIUS is a compile-time promise that radius == 0 (the reference's own value): it removes the general tap loop
// from the instantiation the reference-literal configuration runs.
// W_ONE is a promise that spw == 1.0f exactly (orthographic light projection, bounded posW): x / 1 is x, so the three
// divisions of the perspective divide are skipped without changing a bit.
template <bool ZERO_RADIUS, bool W_ONE = false>
CRY_HD float pcf_poisson(const uint32_t* __restrict__ s, uint32_t dim, float spx, float spy, float spz, float spw,
                         float radius)
{
    const float rw = W_ONE ? 1.0f : rcp(spw);      // x / 1 == x * rcp(1): the W_ONE form drops an exact identity
    const float x = W_ONE ? spx : spx * rw, y = W_ONE ? spy : spy * rw, depth = W_ONE ? spz : spz * rw;  // :266-269
    float percentLit = 0.0f;
    if (ZERO_RADIUS || radius == 0.0f) {
        return pcf_zero_radius(shadow_cmp_linear(s, dim, x, y, depth));
    } else {
        bool inside = pcf_taps_inside(dim, x, y, depth, radius);
#if defined(__HIP_DEVICE_COMPILE__)
        inside = __builtin_amdgcn_ballot_w64(!inside) == 0;      // wave-uniform: the rim of a cascade runs the general loop
#endif
        if (inside) return pcf_poisson_inside(s, dim, x, y, depth, radius);
        const float theta = nrand(x, y);                          // :301
        const float c = det_cos(theta), sn = det_sin(theta);      // :302-303
        const float P[32] = CRY_POISSON_TABLE;
#pragma unroll
        for (int i = 0; i < 16; ++i) {                            // :308
            const float px = fma(P[2 * i + 1], -sn, P[2 * i] * c); // mul(poissonDisk[i], float2x2(c, s, -s, c))
            const float py = fma(P[2 * i + 1], c, P[2 * i] * sn);
            percentLit += shadow_cmp_linear(s, dim, fma(px, radius, x), fma(py, radius, y), depth);  // :311-313
        }
    }
    return percentLit * 0.0625f;                                  // :315
}

// The cascades' light projections are orthographic (CRYCHIC.cpp:804), so ShadowTransform = lightView * ortho * T has the
// w column (0, 0, 0, 1): shadowPosH.w = ((x*0 + y*0) + z*0) + 1 is exactly 1 for every finite posW.
CRY_HD bool light_shadow_w_is_one(const float (*T)[16])
{
    bool ok = true;
    for (int j = 0; j < 4; ++j) {
        ok = ok && T[j][12] == 0.0f && T[j][13] == 0.0f && T[j][14] == 0.0f && T[j][15] == 1.0f;
        for (int k = 0; k < 12; ++k) ok = ok && __builtin_fabsf(T[j][k]) < 1.0e12f;
    }
    return ok;
}

CRY_HD float pow5(float x) { float x2 = x * x, x4 = x2 * x2; return x4 * x; }

#define CRY_PBR_PI 3.1415926f  // PBR.hlsl:2

// GetPBRDesc (PBR.hlsl:72-88) + GetBRDF (:45-70) for light direction `lightDir`; adds scale * brdf * (strength * nDotl * att)
// in the shader's association order.  Directional lights: att = 1 is skipped (POINT = false).
// fixQ3 / fixQ4: the CRYCHIC_FIX_Q3 / Q4 forms (crychic_hip.h); both false = the reference as written.
// BOUNDED: a promise that the pixel passed light_dark_guard() and that the light direction is no longer than 1.001 (light_dark_mask's
// test) -- then every reciprocal below has a normal argument with a normal result (the Smith denominators lie in [0.13, 15.2],
// nDotl * nDotv in [1e-6, 1.01], pi * tt^2 in [2.5e-6, 3.2e4]: see "dark lights"), where rcp_normal IS rcp, two instructions shorter.
// ... in two steps: pbr_light_eval computes brdf and strength * nDotl (* att), pbr_light_add applies
// result = mad(scale * brdf, lightStrength, result) -- the shader's own operations in the shader's own order.  (Evaluating
// lights ahead of the shadow factor's resolve, to hide the cascades' round trip, was measured and lost: profiles/r04_experiments.txt.)
struct LightTerm { v2f brdfRG, lsRG; float brdfB, lsB; };
template <bool POINT, bool BOUNDED = false>
CRY_HD LightTerm pbr_light_eval(f3 lightDir, const float* strength, float att, f3 albedo, float roughness, float metalness, f3 normal,
                                f3 view, bool fixQ3 = false, bool fixQ4 = false)
{
    auto rcpB = [](float b) { return BOUNDED ? rcp_normal(b) : rcp(b); };
    const f3 halfVec = normalize3(f3{ view.x + lightDir.x, view.y + lightDir.y, view.z + lightDir.z });
    const float hDotv = maxnn(dot3(halfVec, view), 0.001f);
    const float nDotl = maxnn(dot3(normal, lightDir), 0.001f);
    const float nDotv = maxnn(dot3(normal, view), 0.001f);
    const float nDotvQ = hDotv;  // PBR.hlsl:58 (Q3)

    // NDF_GGX :4-14
    const float a2 = roughness * roughness;
    const float nDoth = maxnn(dot3(normal, halfVec), 0.001f);
    const float tt = fma(nDoth * nDoth, a2 - 1.0f, 1.0f);
    const float D = a2 * rcpB(CRY_PBR_PI * (tt * tt));
    // FresnelSchlick :40-43
    const float fr = pow5(saturate(1.0f - nDotvQ));
    // GeometrySmith :29-38 (true nDotv)
    const float k = 0.125f * (roughness + 1.0f) * (roughness + 1.0f);
    const float G = (nDotv * rcpB(fma(nDotv, 1.0f - k, k))) * (nDotl * rcpB(fma(nDotl, 1.0f - k, k)));      // a / b = a * rcp(b)
    const float rdenom = rcpB(nDotl * (fixQ3 ? nDotv : nDotvQ));
    const float invPi = 1.0f / CRY_PBR_PI;
    const float oneMinusMetal = 1.0f - metalness;

    // The three colour channels run the same expression tree: red and green as one packed pair (v_pk_*_f32: two channels per
    // issue slot), blue as scalars.  Every lane is the scalar expression bit for bit (devmath.hpp "two-wide packed fp32").
    const float c025DG = 0.25f * D * G;
    LightTerm t;
    {
        const v2f alb{ albedo.x, albedo.y }, str{ strength[0], strength[1] };
        const v2f f0 = fma2(splat(metalness), alb - 0.04f, splat(0.04f));       // lerp(0.04, albedo, metalness)
        const v2f F = fma2(1.0f - f0, splat(fr), f0);
        v2f fs = c025DG * F;
        fs = fs * rdenom;
        const v2f fd = alb * invPi;
        const v2f kd = (1.0f - F) * oneMinusMetal;
        t.brdfRG = fixQ4 ? kd * fd + fs : fma2(F, fs, kd * fd);                 // ks = F (Q4)
        t.lsRG = str * nDotl;                                                    // :104 / :118
        if (POINT) t.lsRG = t.lsRG * att;                                        // :120
    }
    {
        const float f0 = lerpf(0.04f, albedo.z, metalness);
        const float F = fma(1.0f - f0, fr, f0);
        float fs = c025DG * F;
        fs = fs * rdenom;
        const float fd = albedo.z * invPi;
        const float kd = (1.0f - F) * oneMinusMetal;
        t.brdfB = fixQ4 ? kd * fd + fs : fma(F, fs, kd * fd);
        t.lsB = strength[2] * nDotl;
        if (POINT) t.lsB = t.lsB * att;
    }
    return t;
}
CRY_HD void pbr_light_add(const LightTerm& t, float scale, f3& result)
{
    const v2f res = fma2(scale * t.brdfRG, t.lsRG, v2f{ result.x, result.y });   // :105 / :122
    result.x = res.x;
    result.y = res.y;
    result.z = fma(scale * t.brdfB, t.lsB, result.z);
}
template <bool POINT, bool BOUNDED = false>
CRY_HD void pbr_light(f3 lightDir, const float* strength, float att, f3 albedo, float roughness, float metalness, f3 normal,
                      f3 view, float scale, f3& result, bool fixQ3 = false, bool fixQ4 = false)
{
    pbr_light_add(pbr_light_eval<POINT, BOUNDED>(lightDir, strength, att, albedo, roughness, metalness, normal, view, fixQ3, fixQ4), scale, result);
}

// One directional light of PBRShading (PBR.hlsl:99-106).
template <bool BOUNDED = false>
CRY_HD void pbr_dir_light(const crychic_light& L, f3 albedo, float roughness, float metalness, f3 normal, f3 view,
                          float shadow, f3& result, bool fixQ3 = false, bool fixQ4 = false)
{
    pbr_light<false, BOUNDED>(f3{ -L.Direction[0], -L.Direction[1], -L.Direction[2] }, L.Strength, 1.0f, albedo, roughness, metalness, normal,
                              view, pow5(shadow), result, fixQ3, fixQ4);
}

// Point light, BUILD-DEFINED EXTENSION: the reference's branch (PBR.hlsl:109-124) is dead code; enabled as evidently
// intended -- range test d > FalloffEnd (LightingUtil.hlsl:104-105), l /= d, linear attenuation, shadowFactor 1.
CRY_HD void pbr_point_light(const crychic_light& L, f3 pos, f3 albedo, float roughness, float metalness, f3 normal, f3 view,
                            f3& result, bool fixQ3 = false, bool fixQ4 = false)
{
    const f3 l{ L.Position[0] - pos.x, L.Position[1] - pos.y, L.Position[2] - pos.z };
    const float d = len_from_sq(dot3(l, l));
    if (d > L.FalloffEnd) return;
    const float rd = rcp(d);
    const f3 ln{ l.x * rd, l.y * rd, l.z * rd };
    const float att = saturate(divf(L.FalloffEnd - d, L.FalloffEnd - L.FalloffStart));
    pbr_light<true>(ln, L.Strength, att, albedo, roughness, metalness, normal, view, 1.0f, result, fixQ3, fixQ4);
}

// TextureCube.Sample(gsamLinearWrap, r): D3D major-axis face selection (ties x >= y >= z), bilinear inside the
// face with clamp-to-edge.  Faces +X,-X,+Y,-Y,+Z,-Z, RGBA8.
struct CubeFetch { TexelPair p0, p1; float fx, fy; };
// The two rows of a footprint as loaded, before pair_at_clamped's picks: texels (cx, cx + 1) of rows y0, y1, and the column index
// i0 the picks need (cube_pick).  cube_fetch issues the loads and nothing else touches them until cube_resolve.
struct CubeRows { RawPair r0, r1; float fx, fy; int i0; };
// Where the two rows of a footprint start (texel indices into the cube map plane) and what the filter needs besides the texels.
struct CubeAddr { uint32_t t0, t1; float fx, fy; int i0; };
CRY_HD CubeAddr cube_address(uint32_t dim, f3 r)
{
    const float ax = __builtin_fabsf(r.x), ay = __builtin_fabsf(r.y), az = __builtin_fabsf(r.z);
    // major axis (ties x >= y >= z) and the face's (sc, tc) by selects: a wave whose lanes look at different faces stays converged
    const bool isx = (ax >= ay) & (ax >= az), isy = !isx & (ay >= az);
    const bool px = r.x >= 0.0f, py = r.y >= 0.0f, pz = r.z >= 0.0f;
    const bool ux = __builtin_unpredictable(isx), uy = __builtin_unpredictable(isy);      // selects, not branches
    const float ma = ux ? ax : (uy ? ay : az);
    const float sc = ux ? (px ? -r.z : r.z) : (uy ? r.x : (pz ? r.x : -r.x));
    const float tc = ux ? -r.y : (uy ? (py ? r.z : -r.z) : -r.y);
    const uint32_t face = ux ? (px ? 0u : 1u) : (uy ? (py ? 2u : 3u) : (pz ? 4u : 5u));
    const float rma = rcp(ma);
    const float u = fma(0.5f, sc * rma, 0.5f);      // 0.5 * (sc / ma + 1)
    const float v = fma(0.5f, tc * rma, 0.5f);
    const uint32_t faceRow = mul24(face, dim);   // faces are stacked: row index face*dim + y
    // Footprints in the interior of their face (floor(texel coordinate) in [0, dim - 2] on both axes, which a NaN fails), for the
    // whole wavefront: bilinear_setup's non-finite rule and the CLAMP of rows and columns cannot act, the rows are at
    // (face * dim + j0) * dim + i0.  A wavefront with a footprint on a face edge takes the general sampler's indices.  As in
    // ambient_fetch_projected the vote chooses how the ADDRESSES are computed; the loads follow the merge.
    const float fd = (float)dim;
    const float tx = fma(u, fd, -0.5f), ty = fma(v, fd, -0.5f);
    const float flx = __builtin_floorf(tx), fly = __builtin_floorf(ty);
    bool inside = (flx >= 0.0f) & (flx <= fd - 2.0f) & (fly >= 0.0f) & (fly <= fd - 2.0f);
#if defined(__HIP_DEVICE_COMPILE__)
    inside = __builtin_amdgcn_ballot_w64(!inside) == 0;
#endif
    CubeAddr f;
    uint32_t t0, t1;
    if (inside) {
        f.i0 = (int)flx;
        t0 = mul24(faceRow + (uint32_t)(int)fly, dim) + (uint32_t)f.i0;
        t1 = t0 + dim;
        f.fx = tx - flx;
        f.fy = ty - fly;
    } else {
        float us = u, vs = v;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(us), "+v"(vs));      // the general sampler's clamps stay on this side of the vote
#endif
        const Bilin b = bilinear_setup(us, vs, dim, dim);
        const uint32_t y0 = (uint32_t)clampi(b.j0, 0, (int)dim - 1), y1 = (uint32_t)clampi(b.j0 + 1, 0, (int)dim - 1);
        const uint32_t cx = (uint32_t)clampi(b.i0, 0, (int)dim - 2);
        t0 = mul24(faceRow + y0, dim) + cx;
        t1 = mul24(faceRow + y1, dim) + cx;
        f.i0 = b.i0;
        f.fx = b.fx;
        f.fy = b.fy;
    }
    f.t0 = t0;
    f.t1 = t1;
    return f;
}
CRY_HD CubeRows cube_load(const uint32_t* __restrict__ cube, const CubeAddr& a)
{
    return CubeRows{ load_pair(cube, a.t0), load_pair(cube, a.t1), a.fx, a.fy, a.i0 };
}
CRY_HD CubeRows cube_fetch(const uint32_t* __restrict__ cube, uint32_t dim, f3 r) { return cube_load(cube, cube_address(dim, r)); }
// pair_at_clamped's picks: texels clamp(i0) and clamp(i0 + 1) out of the loaded pair (cx, cx + 1), cx = clamp(i0, 0, dim - 2)
CRY_HD CubeFetch cube_pick(const CubeRows& c, uint32_t dim)
{
    const bool hiA = c.i0 > (int)dim - 2, loB = c.i0 < 0;
    CubeFetch f;
    f.p0 = TexelPair{ hiA ? c.r0.hi : c.r0.lo, loB ? c.r0.lo : c.r0.hi };
    f.p1 = TexelPair{ hiA ? c.r1.hi : c.r1.lo, loB ? c.r1.lo : c.r1.hi };
    f.fx = c.fx;
    f.fy = c.fy;
    return f;
}
// WANT_ALPHA = false (the reflection lookup of a lit pixel, which uses .rgb only): .w = 0.
// Every channel is bilerp of the four decoded texels, bit for bit; the work is arranged in packed pairs: red + green of one texel
// decode together (and blue + alpha), and without alpha the blue channel pairs the two texels of a COLUMN -- {b00, b01} and
// {b10, b11} -- so that both row filters are one packed lerp.
CRY_HD v2f unorm8_to_float2(uint32_t a, uint32_t b)
{
    const v2f t{ (float)a, (float)b };
    return fma2(t, splat(u2f(0x3B808081u)), t * u2f(0xAF7EFEFFu));       // unorm_decode per lane
}
template <bool WANT_ALPHA = true>
CRY_HD f4 cube_resolve(const CubeFetch& f)
{
    const uint32_t t00 = f.p0.a, t10 = f.p0.b, t01 = f.p1.a, t11 = f.p1.b;
    auto pair = [&](uint32_t sh) {
        const v2f a00 = unorm8_to_float2((t00 >> sh) & 255u, (t00 >> (sh + 8u)) & 255u), a10 = unorm8_to_float2((t10 >> sh) & 255u, (t10 >> (sh + 8u)) & 255u);
        const v2f a01 = unorm8_to_float2((t01 >> sh) & 255u, (t01 >> (sh + 8u)) & 255u), a11 = unorm8_to_float2((t11 >> sh) & 255u, (t11 >> (sh + 8u)) & 255u);
        const v2f top = lerp2(a00, a10, splat(f.fx)), bot = lerp2(a01, a11, splat(f.fx));
        return lerp2(top, bot, splat(f.fy));
    };
    const v2f rg = pair(0u);
    if (WANT_ALPHA) {
        const v2f ba = pair(16u);
        return f4{ rg.x, rg.y, ba.x, ba.y };
    }
    const v2f left = unorm8_to_float2((t00 >> 16) & 255u, (t01 >> 16) & 255u), right = unorm8_to_float2((t10 >> 16) & 255u, (t11 >> 16) & 255u);
    const v2f rows = lerp2(left, right, splat(f.fx));        // { top, bot }
    return f4{ rg.x, rg.y, lerpf(rows.x, rows.y, f.fy), 0.0f };
}
CRY_HD f4 cube_linear(const uint32_t* __restrict__ cube, uint32_t dim, f3 r) { return cube_resolve(cube_pick(cube_fetch(cube, dim, r), dim)); }

// ---- TextureCube.Sample with the mip chain bound (MIN_MAG_MIP_LINEAR, CRYCHIC.cpp:2617-2622, :1148-1151) -------------------------
// D3D leaves the level-of-detail arithmetic to the hardware; the definition used here is written out in DESIGN.md section 3 ("the cube
// map's mip chain") and restated independently by the test suite's CPU checker: derivatives of the direction inside
// the pixel's 2 x 2 quad, carried to the selected face by the chain rule, lod = min(0.5 log2(rho^2), levels - 1), two levels
// filtered as level 0 is and mixed with one mad.  The chain: level after level, each six faces of max(dim >> level, 1)^2 texels.
CRY_HD void cube_face_delta(f3 r, f3 d, float halfDim, float& du, float& dv)
{
    const float ax = __builtin_fabsf(r.x), ay = __builtin_fabsf(r.y), az = __builtin_fabsf(r.z);
    const bool isx = (ax >= ay) & (ax >= az), isy = !isx & (ay >= az);
    const bool px = r.x >= 0.0f, py = r.y >= 0.0f, pz = r.z >= 0.0f;
    const float ma = isx ? ax : (isy ? ay : az);
    const float sc = isx ? (px ? -r.z : r.z) : (isy ? r.x : (pz ? r.x : -r.x));
    const float tc = isx ? -r.y : (isy ? (py ? r.z : -r.z) : -r.y);
    const float dsc = isx ? (px ? -d.z : d.z) : (isy ? d.x : (pz ? d.x : -d.x));
    const float dtc = isx ? -d.y : (isy ? (py ? d.z : -d.z) : -d.y);
    const float dma = isx ? (px ? d.x : -d.x) : (isy ? (py ? d.y : -d.y) : (pz ? d.z : -d.z));
    const float rma = rcp(ma);
    du = fma(-(sc * rma), dma, dsc) * rma * halfDim;
    dv = fma(-(tc * rma), dma, dtc) * rma * halfDim;
}
CRY_HD float cube_lod(uint32_t dim, uint32_t levels, f3 r, f3 ddx, f3 ddy)
{
    const float halfDim = 0.5f * (float)dim;
    float ux, vx, uy, vy;
    cube_face_delta(r, ddx, halfDim, ux, vx);
    cube_face_delta(r, ddy, halfDim, uy, vy);
    const float rx = fma(ux, ux, vx * vx), ry = fma(uy, uy, vy * vy);
    const float rho2 = (ry > rx) ? ry : rx;
    if (!(rho2 > 1.0f)) return 0.0f;
    const float lod = 0.5f * det_log2_normal(rho2 < 3.0e38f ? rho2 : 3.0e38f);
    const float top = (float)(levels - 1u);
    return lod < top ? lod : top;
}
CRY_HD uint32_t cube_level_dim(uint32_t dim, uint32_t level) { const uint32_t d = dim >> level; return d ? d : 1u; }
// The footprint of a lookup at one level of the chain, for levels (and so level sizes) that differ from lane to lane: selects, no
// branch, so that a caller can issue the loads of two levels together.  A 1 x 1 level is its face's texel: the pair loaded is
// texels (min(face, 4), + 1) of the level -- inside it -- and `hiOnly` says which of the two is the face's (cube_level_pick); its
// four "texels" being one, the filter weights (computed as for any level) do not matter unless they are non-finite, and then they
// poison the result exactly as they do in the definition.
struct CubeLevelAddr { uint32_t t0, t1; float fx, fy; int i0; uint32_t d; bool one, hiOnly; };
CRY_HD CubeLevelAddr cube_level_address(uint32_t off, uint32_t d, f3 r)
{
    const float ax = __builtin_fabsf(r.x), ay = __builtin_fabsf(r.y), az = __builtin_fabsf(r.z);
    const bool isx = (ax >= ay) & (ax >= az), isy = !isx & (ay >= az);
    const bool px = r.x >= 0.0f, py = r.y >= 0.0f, pz = r.z >= 0.0f;
    const float ma = isx ? ax : (isy ? ay : az);
    const float sc = isx ? (px ? -r.z : r.z) : (isy ? r.x : (pz ? r.x : -r.x));
    const float tc = isx ? -r.y : (isy ? (py ? r.z : -r.z) : -r.y);
    const uint32_t face = isx ? (px ? 0u : 1u) : (isy ? (py ? 2u : 3u) : (pz ? 4u : 5u));
    const float rma = rcp(ma);
    const float u = fma(0.5f, sc * rma, 0.5f), v = fma(0.5f, tc * rma, 0.5f);
    const Bilin b = bilinear_setup(u, v, d, d);
    const uint32_t dd = d < 2u ? 2u : d;                // keeps the clamps in range for the 1 x 1 level, whose addresses are set below
    const uint32_t y0 = (uint32_t)clampi(b.j0, 0, (int)dd - 1), y1 = (uint32_t)clampi(b.j0 + 1, 0, (int)dd - 1);
    const uint32_t cx = (uint32_t)clampi(b.i0, 0, (int)dd - 2);
    const uint32_t faceRow = face * d;
    CubeLevelAddr a;
    a.one = d < 2u;
    a.hiOnly = face == 5u;
    const uint32_t single = off + (face < 4u ? face : 4u);
    a.t0 = a.one ? single : off + (faceRow + y0) * d + cx;
    a.t1 = a.one ? single : off + (faceRow + y1) * d + cx;
    a.fx = b.fx; a.fy = b.fy; a.i0 = b.i0; a.d = d;
    return a;
}
CRY_HD CubeFetch cube_level_pick(const CubeLevelAddr& a, RawPair r0, RawPair r1)
{
    const CubeFetch f = cube_pick(CubeRows{ r0, r1, a.fx, a.fy, a.i0 }, a.d);
    const uint32_t t = a.hiOnly ? r0.hi : r0.lo;
    return a.one ? CubeFetch{ TexelPair{ t, t }, TexelPair{ t, t }, a.fx, a.fy } : f;
}
// The two levels of a trilinear lookup: both footprints addressed, then the four loads, then the filters and the mad.
template <bool WANT_ALPHA>
CRY_HD f4 cube_trilinear(const uint32_t* __restrict__ chain, uint32_t dim, uint32_t levels, f3 r, float lod)
{
    const uint32_t l0 = (uint32_t)lod;
    const float frac = lod - (float)l0;
    const uint32_t l1 = l0 + 1u < levels ? l0 + 1u : l0;       // l1 == l0 or frac == 0: the mad returns c0 (texels are finite)
    uint32_t off0 = 0;                                          // texels before level l0
    for (uint32_t k = 0; k + 1u < levels; ++k) { const uint32_t d = cube_level_dim(dim, k); off0 += k < l0 ? 6u * d * d : 0u; }
    const uint32_t d0 = cube_level_dim(dim, l0), d1 = cube_level_dim(dim, l1);
    const uint32_t off1 = off0 + (l1 > l0 ? 6u * d0 * d0 : 0u);
    const CubeLevelAddr a0 = cube_level_address(off0, d0, r), a1 = cube_level_address(off1, d1, r);
    const RawPair p00 = load_pair(chain, a0.t0), p01 = load_pair(chain, a0.t1), p10 = load_pair(chain, a1.t0), p11 = load_pair(chain, a1.t1);
    const f4 c0 = cube_resolve<WANT_ALPHA>(cube_level_pick(a0, p00, p01));
    const f4 c1 = cube_resolve<WANT_ALPHA>(cube_level_pick(a1, p10, p11));
    return f4{ fma(frac, c1.x - c0.x, c0.x), fma(frac, c1.y - c0.y, c0.y), fma(frac, c1.z - c0.z, c0.z), WANT_ALPHA ? fma(frac, c1.w - c0.w, c0.w) : 0.0f };
}
// How light_pixel looks the cube map up: level 0 alone (the fetch in flight with the pixel's other gathers) ...
struct CubeLevel0 {
    struct Fetch { CubeRows c; };
    CRY_HD Fetch fetch(const LightParams& P, const uint32_t* __restrict__ cube, f3 r) const { return Fetch{ cube_fetch(cube, P.cubeDim, r) }; }
    CRY_HD f4 resolve(const LightParams& P, const uint32_t* __restrict__, const Fetch& f) const { return cube_resolve<false>(cube_pick(f.c, P.cubeDim)); }
};
// ... or the chain at the level of detail the caller derived from the pixel's quad (light_kernel<.., MIPS>).  `flat`: every lane of the
// wavefront that shades looks at level 0 alone (lod == 0 -- the lookup is magnified, which is most of a 4K frame): the mad with
// frac == 0 returns the level-0 filter bit for bit, so the wavefront takes CubeLevel0's lookup, in flight with its other gathers.
struct CubeChain {
    float lod;
    bool flat;
    struct Fetch { CubeRows c; f3 r; };
    CRY_HD Fetch fetch(const LightParams& P, const uint32_t* __restrict__ cube, f3 r) const
    {
        CubeAddr a{ 0u, 0u, 0.0f, 0.0f, 0 };
        if (flat) a = cube_address(P.cubeDim, r);            // the vote picks the addresses; the loads follow the merge
        return Fetch{ cube_load(cube, a), r };
    }
    CRY_HD f4 resolve(const LightParams& P, const uint32_t* __restrict__ cube, const Fetch& f) const
    {
        if (flat) return cube_resolve<false>(cube_pick(f.c, P.cubeDim));
        return cube_trilinear<false>(cube, P.cubeDim, P.cubeLevels, f.r, lod);
    }
};
// `flat` for a set of lanes: on the device a vote of the lanes that are active at the call, on the host the lane's own answer
CRY_HD bool cube_chain_flat(float lod)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(lod != 0.0f) == 0;
#else
    return lod == 0.0f;
#endif
}

// gsamLinearClamp on the half-res R16_UNORM ambient map  (CRYCHIC.cpp:2624-2629)
struct AmbientFetch { uint16_t t00, t10, t01, t11; float fx, fy; };
CRY_HD AmbientFetch ambient_fetch(const uint16_t* __restrict__ a, uint32_t w2, uint32_t h2, float u, float v)
{
    const Bilin b = bilinear_setup(u, v, w2, h2);
    const uint32_t x0 = (uint32_t)clampi(b.i0, 0, (int)w2 - 1), x1 = (uint32_t)clampi(b.i0 + 1, 0, (int)w2 - 1);
    const uint32_t y0 = (uint32_t)clampi(b.j0, 0, (int)h2 - 1), y1 = (uint32_t)clampi(b.j0 + 1, 0, (int)h2 - 1);
    const uint32_t r0 = mul24(y0, w2), r1 = mul24(y1, w2);
    return AmbientFetch{ load_at<uint16_t>(a, (r0 + x0) * 2u), load_at<uint16_t>(a, (r0 + x1) * 2u),
                         load_at<uint16_t>(a, (r1 + x0) * 2u), load_at<uint16_t>(a, (r1 + x1) * 2u), b.fx, b.fy };
}
// The lighting pass's own lookup (DeferredShading.hlsl:40-42): posW projected by gViewProjTex, x and y as one packed pair
// (LightParams::ScreenPairs; each lane is mulcol1 / bilinear_setup bit for bit), and the four texels as TWO 4-byte loads of
// {t(cx), t(cx + 1)} (2-byte aligned) from rows y0, y1.  When the footprint of every pixel of the wavefront lies inside the map
// (everything but the frame's rim and non-finite positions: a NaN fails the test) cx = i0 and nothing is clamped; otherwise the
// general sampler's indices (non-finite rule, CLAMP) with cx = clamp(i0, 0, w - 2), and the texels are picked from the loaded
// pairs afterwards.  The vote only chooses how ADDRESSES are computed: the loads themselves sit behind the merge, so that no wait
// for them is needed before the other gathers of the pixel have been issued (a branch that returns loaded data makes the
// compiler wait inside it: three serialised round trips per pixel, measured -- profiles/r04_experiments.txt).
// hasAO false: the 1 x 1 stand-in of light_pixel (the cubemap's first bytes): w = h = 1, both picks take the low half.
struct AmbientPairs { uint32_t d0, d1; float fx, fy; bool x0lo, x1lo; };      // t00 = x0lo ? lo(d0) : hi(d0), t10 = x1lo ? lo(d0) : hi(d0), ...
CRY_HD AmbientPairs ambient_fetch_projected(const LightParams& P, const uint16_t* __restrict__ a, bool hasAO, const uint16_t* __restrict__ standIn, f3 posW)
{
    const v2f c0{ P.ScreenPairs[0][0], P.ScreenPairs[0][1] }, c1{ P.ScreenPairs[1][0], P.ScreenPairs[1][1] };
    const v2f c2{ P.ScreenPairs[2][0], P.ScreenPairs[2][1] }, c3{ P.ScreenPairs[3][0], P.ScreenPairs[3][1] };
    const v2f s = fma2(splat(posW.z), c2, fma2(splat(posW.y), c1, splat(posW.x) * c0)) + c3;
    const float rsw = rcp(mulcol1(posW.x, posW.y, posW.z, P.ViewProjTex + 12));
    const v2f uv = s * rsw;
    // A one-texel-wide map (W = 2) has no pair to load -- its last row's dword would leave the plane: such a frame reads the
    // stand-in here (valid memory, never looked at) and samples the map in ambient_resolve, from the coordinates kept in fx, fy.
    const bool tiny = hasAO && P.W < 4u;
    if (tiny) hasAO = false;
    const uint32_t w2 = hasAO ? P.W / 2u : 1u, h2 = hasAO ? P.H / 2u : 1u;
    const v2f dims{ hasAO ? P.halfDims[0] : 1.0f, hasAO ? P.halfDims[1] : 1.0f };
    const v2f t = fma2(uv, dims, -0.5f);
    const v2f fl = floor2(t);
    bool inside = (fl.x >= 0.0f) & (fl.x <= dims.x - 2.0f) & (fl.y >= 0.0f) & (fl.y <= dims.y - 2.0f);
#if defined(__HIP_DEVICE_COMPILE__)
    inside = __builtin_amdgcn_ballot_w64(!inside) == 0;
#endif
    AmbientPairs f;
    uint32_t o0, o1;
    if (inside) {
        o0 = mul24((uint32_t)(int)fl.y, w2) + (uint32_t)(int)fl.x;
        o1 = o0 + w2;
        const v2f fr = t - fl;
        f.fx = fr.x; f.fy = fr.y; f.x0lo = true; f.x1lo = false;
    } else {
        float u = uv.x, v = uv.y;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(u), "+v"(v));      // the general sampler's clamps stay on this side of the vote
#endif
        const Bilin b = bilinear_setup(u, v, w2, h2);
        const int x0 = clampi(b.i0, 0, (int)w2 - 1), x1 = clampi(b.i0 + 1, 0, (int)w2 - 1);
        const int cx = clampi(b.i0, 0, (int)w2 - 2 > 0 ? (int)w2 - 2 : 0);
        const uint32_t y0 = (uint32_t)clampi(b.j0, 0, (int)h2 - 1), y1 = (uint32_t)clampi(b.j0 + 1, 0, (int)h2 - 1);
        o0 = mul24(y0, w2) + (uint32_t)cx;
        o1 = mul24(y1, w2) + (uint32_t)cx;
        f.fx = b.fx; f.fy = b.fy; f.x0lo = x0 == cx; f.x1lo = x1 == cx;
    }
    const uint16_t* src = hasAO ? a : standIn;
    f.d0 = load_at<uint32_t>(src, o0 * 2u);
    f.d1 = load_at<uint32_t>(src, o1 * 2u);
    if (tiny) { f.fx = uv.x; f.fy = uv.y; }
    return f;
}
CRY_HD float ambient_linear_clamp(const uint16_t* __restrict__ a, uint32_t w2, uint32_t h2, float u, float v);
CRY_HD float ambient_resolve(const LightParams& P, const uint16_t* __restrict__ a, const AmbientPairs& f)
{
    if (P.W < 4u) return ambient_linear_clamp(a, P.W / 2u, P.H / 2u, f.fx, f.fy);      // the one-texel-wide map (see the fetch)
#if defined(__HIP_DEVICE_COMPILE__)
    // a wavefront over unoccluded ground (most of them): texels of 65535 decode to 1.0 and lerp(1, 1, t) = mad(t, 0, 1) = 1 for the
    // finite weights bilinear_setup returns (both halves of both pairs all ones: the four texels, whichever halves they are)
    if (__builtin_amdgcn_ballot_w64((f.d0 & f.d1) != 0xFFFFFFFFu) == 0) return 1.0f;
#endif
    const uint32_t l0 = f.d0 & 0xFFFFu, h0 = f.d0 >> 16, l1 = f.d1 & 0xFFFFu, h1 = f.d1 >> 16;
    return bilerp(unorm16_to_float(f.x0lo ? l0 : h0), unorm16_to_float(f.x1lo ? l0 : h0), unorm16_to_float(f.x0lo ? l1 : h1),
                  unorm16_to_float(f.x1lo ? l1 : h1), f.fx, f.fy);
}
CRY_HD float ambient_resolve(const AmbientFetch& f)
{
#if defined(__HIP_DEVICE_COMPILE__)
    // a wavefront over unoccluded ground (most of them): four texels of 65535 decode to 1.0 and lerp(1, 1, t) = mad(t, 0, 1) = 1
    // for the finite weights bilinear_setup returns
    if (__builtin_amdgcn_ballot_w64((f.t00 & f.t10 & f.t01 & f.t11) != 0xFFFFu) == 0) return 1.0f;
#endif
    return bilerp(unorm16_to_float(f.t00), unorm16_to_float(f.t10), unorm16_to_float(f.t01), unorm16_to_float(f.t11), f.fx, f.fy);
}
CRY_HD float ambient_linear_clamp(const uint16_t* __restrict__ a, uint32_t w2, uint32_t h2, float u, float v)
{
    const AmbientFetch f = ambient_fetch(a, w2, h2, u, v);
    return bilerp(unorm16_to_float(f.t00), unorm16_to_float(f.t10), unorm16_to_float(f.t01), unorm16_to_float(f.t11), f.fx, f.fy);
}

// ---- dark lights -------------------------------------------------------------------------------------------------------
// A directional light whose Strength is exactly (0, 0, 0) -- the reference's own third light is one (CRYCHIC.cpp:863-864) -- adds
// fma(scale * brdf, 0 * nDotl, result) to each channel (PBR.hlsl:104-105): result itself whenever scale * brdf and nDotl are
// finite (x * 0 = +-0, and r + +-0 = r; a zero r can only change the sign of its zero, which no output can see: the tone map
// takes +0 and -0 to the same 0, DeferredShading.hlsl:89-90).  light_dark_mask() picks such lights on the host; the pixel side
// is light_dark_guard(): inputs bounded so that no term of GetBRDF (PBR.hlsl:45-70) can overflow or become NaN --
//   roughness in [0.03, 10]: a^2 >= 9e-4 keeps NDF_GGX's denominator (nDoth^2 (a^2 - 1) + 1)^2 away from 0 (nDoth <= 1 + ulps:
//     both vectors are normalised), D <= 400; k = (r + 1)^2 / 8 <= 15.2 keeps GeometrySchlickGGX's denominators positive for
//     nDotl <= 1.002, G <= 1e6;
//   |albedo|, |metalness| <= 16: f0, F <= 512, so F * fs <= 512 * (0.25 * 400 * 1e6 * 512 * 1e6) ~ 3e19;
//   G-buffer position and normal finite, |EyePosW - posW|^2 < 1e30 (light_pixel adds that test to the guard: the normalised view
//     vector is then finite and at most unit length -- the G-buffer bounds alone do not bound the eye), light direction finite with
//     length in [0.5, 1.001] (host): every dot product is finite, nDotl <= 1.002.
// A wavefront skips a dark light only if all its pixels pass the guard; otherwise it evaluates the light like any other.
CRY_HD uint32_t light_dark_mask(const crychic_light* L, int n)
{
    uint32_t m = 0;
    for (int i = 0; i < n && i < 32; ++i) {
        const float* s = L[i].Strength;
        const float* d = L[i].Direction;
        const float len2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
        if (s[0] == 0.0f && s[1] == 0.0f && s[2] == 0.0f && len2 >= 0.25f && len2 <= 1.002f) m |= 1u << i;       // NaN fails every test
    }
    return m;
}
CRY_HD bool light_dark_lengths_ok(const crychic_light* L, int n)
{
    for (int i = 0; i < n; ++i) {
        const float* d = L[i].Direction;
        if (!(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] <= 1.002f)) return false;      // NaN fails
    }
    return true;
}
CRY_HD bool light_dark_guard(f4a G0, f4a G1, f4a G2)
{
    const float pmax = 1.2676506e30f;
    return (G1.w >= 0.03f) & (G1.w <= 10.0f) & (__builtin_fabsf(G1.x) <= 16.0f) & (__builtin_fabsf(G1.y) <= 16.0f) & (__builtin_fabsf(G1.z) <= 16.0f) &
           (__builtin_fabsf(G0.w) <= 16.0f) & (__builtin_fabsf(G0.x) < pmax) & (__builtin_fabsf(G0.y) < pmax) & (__builtin_fabsf(G0.z) < pmax) &
           (__builtin_fabsf(G2.x) < 3.0e38f) & (__builtin_fabsf(G2.y) < 3.0e38f) & (__builtin_fabsf(G2.z) < 3.0e38f);
}

// Point-light iteration policies of light_pixels (extension).
struct NoPointLights {
    CRY_HD void operator()(f3, f3, float, float, f3, f3, f3&, bool, bool) const {}
};
// Iterates every point light of the buffer (what the oracle does); the tiled kernel substitutes a culled iteration.
struct AllPointLights {
    const crychic_light* lights; uint32_t n;
    CRY_HD void operator()(f3 pos, f3 albedo, float roughness, float metalness, f3 normal, f3 view, f3& result, bool fixQ3, bool fixQ4) const
    {
        for (uint32_t i = 0; i < n; ++i) pbr_point_light(lights[i], pos, albedo, roughness, metalness, normal, view, result, fixQ3, fixQ4);
    }
};

// DeferredShading.hlsl:53-76: cascade selection and the (blended) shadow factor of the first light for one pixel.
// `abs(distance - radius[j] < 5.0f)` is abs(bool) (Q1), true whenever distance < radius[j]: every pixel nearer than 80 blends
// cascades j and j+1.
CRY_HD int cascade_index(float distance)    // :53-58
{
    // the first radius the distance is below (radii ascend, so it is below all later ones as well); NaN -> 4
    return 4 - ((int)(distance < 30.0f) + (int)(distance < 50.0f) + (int)(distance < 80.0f) + (int)(distance < 100.0f));
}
template <bool ZERO_RADIUS>
CRY_HD float cascade_shadow(const LightParams& P, f3 posW, float distance, bool fixQ1)
{
    const int j = cascade_index(distance);
    if (j == 4) return 1.0f;
    // posW bounded => no product with a zero matrix entry is NaN / inf => shadowPosH.w == 1 (light_shadow_w_is_one)
    const float pmax = 1.2676506e30f;
    bool wOne = P.shadowWIsOne && __builtin_fabsf(posW.x) < pmax && __builtin_fabsf(posW.y) < pmax && __builtin_fabsf(posW.z) < pmax;
#if defined(__HIP_DEVICE_COMPILE__)
    wOne = __builtin_amdgcn_ballot_w64(!wOne) == 0;          // wave-uniform choice of the instantiation
#endif
    auto cascade = [&](int k) {
        const float* T = P.ShadowTransforms[k];
        const float spx = mulcol1(posW.x, posW.y, posW.z, T + 0), spy = mulcol1(posW.x, posW.y, posW.z, T + 4);
        const float spz = mulcol1(posW.x, posW.y, posW.z, T + 8);
        if (wOne) return pcf_poisson<ZERO_RADIUS, true>(P.shadow[k], P.shadowDim, spx, spy, spz, 1.0f, P.pcfSearchRadius);
        float spw = mulcol1(posW.x, posW.y, posW.z, T + 12);
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(spw));   // keeps the three divisions inside this (rare) branch: the optimiser would otherwise
                                        // speculate them above the wave-uniform test and select afterwards
#endif
        return pcf_poisson<ZERO_RADIUS, false>(P.shadow[k], P.shadowDim, spx, spy, spz, spw, P.pcfSearchRadius);
    };
    const float radiusJ = j == 0 ? 30.0f : (j == 1 ? 50.0f : (j == 2 ? 80.0f : 100.0f));
    const bool blend = j < 3 && (!fixQ1 || __builtin_fabsf(distance - radiusJ) < 5.0f);   // Q1: as written, every j < 3 blends
    // The cascades a pixel needs -- j, and j + 1 when it blends -- visited in a WAVE-UNIFORM loop over k: the transform and the map
    // of cascade k are then scalar operands (a per-lane k costs sixteen vector registers per matrix and per-lane map pointers), and
    // a wavefront whose pixels agree on j runs exactly the two lookups it ran before.  Same lookups per pixel, so the same bits.
    float a = 0.0f, b = 0.0f;
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
        const bool mine = j == k, next = blend & (j + 1 == k);
#if defined(__HIP_DEVICE_COMPILE__)
        if (__builtin_amdgcn_ballot_w64(mine | next) == 0) continue;
#endif
        if (mine | next) {
            const float v = cascade(k);
            a = mine ? v : a;
            b = next ? v : b;
        }
    }
    return blend ? 0.5f * (a + b) : a;              // :66 / :73
}

// The common case of cascade_shadow for a whole wavefront: every pixel in the same cascade J <= 2 (so J and J + 1 are blended,
// as written), orthographic light projections, the literal zero PCF radius.  The transforms are then wave-uniform (scalar
// registers) and the footprints of both cascades are fetched together -- one memory round trip where cascade_shadow takes one
// per lookup.  Same operations per pixel, so the same bits; returns false (nothing done) when the wavefront is not uniform.
//
// Both lookups run as ONE packed evaluation -- lane .x cascade J, lane .y cascade J + 1 (LightParams::ShadowPairs holds the two
// transforms interleaved): three matrix rows, the sampler's texel coordinates, the D24 decodes and the bilinear filter of the
// comparison results are v_pk_*_f32, each lane the scalar expression bit for bit (mulcol1, bilinear_setup<true>, d24_to_float,
// bilerp).  And when both footprints of every pixel of the wavefront lie inside their maps (everything but the rim of a
// cascade: floor(texel coordinate) in [0, dim - 2] on both axes) nothing is clamped, selected or range-tested: the footprint's
// rows are two 8-byte loads at (j0 * dim + i0), the BORDER colour cannot occur.  The rim takes shadow_fetch / shadow_resolve on
// the same coordinates.
CRY_HD v2f d24_to_float2(uint32_t a, uint32_t b)
{
    const v2f t{ (float)(a & 0x00FFFFFFu), (float)(b & 0x00FFFFFFu) };
    return fma2(t, splat(u2f(0x33800001u)), t * u2f(0xA77FFFFFu));      // unorm_decode per lane
}
// In two halves: cascade_uniform_fetch issues the loads (or declines: the wavefront is not uniform), cascade_uniform_resolve turns
// the texels into the factor -- so that the caller can issue every gather of the pixel before anything waits for one.
struct CascadeTexels {
    RawPair a0, a1, b0, b1;        // rows y0, y1 of cascade J (a) and J + 1 (b): texels (cx, cx + 1)
    v2f fx, fy, z;                 // filter weights and reference depths, lane .x = J, .y = J + 1
    bool inside;                   // wave-uniform on the device: every footprint inside its map (cx = i0, no BORDER texel)
    int ia, ja, ib, jb;            // rim only: the footprints' top-left texel indices (pair_at's picks, BORDER tests)
};
// cascade_uniform_test: may this wavefront take the packed path, and with which cascade J (wave-uniform)?  Asked before anything
// else of the pixel is computed, so that a wavefront that may not runs the general lookups (cascade_shadow) while little is live.
template <bool ZERO_RADIUS>
CRY_HD bool cascade_uniform_test(const LightParams& P, f3 posW, float distance, bool fixQ1, int& J)
{
    J = 0;
    if (!ZERO_RADIUS || fixQ1 || !P.shadowWIsOne) return false;
    // |posW| < 1e15 with the transforms' entries below 1e12 (light_shadow_w_is_one): shadowPosH.w == 1, and the shadow
    // coordinates stay below 4e27, so the sampler's texel coordinates (x 16384 at most) are finite: bilinear_setup<FINITE>
    const float pmax = 1.0e15f;
    const int j = cascade_index(distance);
    J = j;
#if defined(__HIP_DEVICE_COMPILE__)
    J = __builtin_amdgcn_readfirstlane(J);
#endif
    const bool ok = j == J && J <= 2 && __builtin_fabsf(posW.x) < pmax && __builtin_fabsf(posW.y) < pmax && __builtin_fabsf(posW.z) < pmax;
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(!ok) == 0;
#else
    return ok;
#endif
}
CRY_HD void cascade_uniform_fetch(const LightParams& P, f3 posW, int J, CascadeTexels& c)
{
    const float (*S)[2] = P.ShadowPairs[J];
    auto row = [&](int r) {        // mulcol1 of both cascades: fma(z, c2, fma(y, c1, x * c0)) + c3 per lane
        const v2f c0{ S[4 * r][0], S[4 * r][1] }, c1{ S[4 * r + 1][0], S[4 * r + 1][1] }, c2{ S[4 * r + 2][0], S[4 * r + 2][1] }, c3{ S[4 * r + 3][0], S[4 * r + 3][1] };
        return fma2(splat(posW.z), c2, fma2(splat(posW.y), c1, splat(posW.x) * c0)) + c3;
    };
    const v2f u = row(0), v = row(1);
    c.z = row(2);
    const uint32_t dim = P.shadowDim;
    const float fd = (float)dim;
    const v2f tx = fma2(u, fd, -0.5f), ty = fma2(v, fd, -0.5f);      // bilinear_setup<true>: fx = tx - floor(tx), no non-finite case
    const v2f flx = floor2(tx), fly = floor2(ty);
    c.fx = tx - flx;
    c.fy = ty - fly;
    const float lo = __builtin_fminf(__builtin_fminf(flx.x, flx.y), __builtin_fminf(fly.x, fly.y));
    const float hi = __builtin_fmaxf(__builtin_fmaxf(flx.x, flx.y), __builtin_fmaxf(fly.x, fly.y));
    bool inside = (lo >= 0.0f) & (hi <= fd - 2.0f);                   // all four finite (above)
#if defined(__HIP_DEVICE_COMPILE__)
    inside = __builtin_amdgcn_ballot_w64(!inside) == 0;
#endif
    c.inside = inside;
    // The vote chooses how the ADDRESSES are computed (ambient_fetch_projected explains why the loads follow the merge).
    uint32_t a0, a1, b0, b1;
    if (inside) {
        a0 = mul24((uint32_t)(int)fly.x, dim) + (uint32_t)(int)flx.x;
        b0 = mul24((uint32_t)(int)fly.y, dim) + (uint32_t)(int)flx.y;
        a1 = a0 + dim;
        b1 = b0 + dim;
        c.ia = c.ja = c.ib = c.jb = 0;
    } else {
        v2f fxs = flx, fys = fly;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(fxs), "+v"(fys));      // keeps the rim's clamps inside this branch (the optimiser would otherwise
                                                      // compute them ahead of the vote)
#endif
        c.ia = texel_index(fxs.x, dim); c.ja = texel_index(fys.x, dim);          // bilinear_setup<true>'s indices; shadow_fetch's rows, pair_at's column
        c.ib = texel_index(fxs.y, dim); c.jb = texel_index(fys.y, dim);
        const uint32_t ca = (uint32_t)clampi(c.ia, 0, (int)dim - 2), cb = (uint32_t)clampi(c.ib, 0, (int)dim - 2);
        a0 = mul24((uint32_t)clampi(c.ja, 0, (int)dim - 1), dim) + ca;
        a1 = mul24((uint32_t)clampi(c.ja + 1, 0, (int)dim - 1), dim) + ca;
        b0 = mul24((uint32_t)clampi(c.jb, 0, (int)dim - 1), dim) + cb;
        b1 = mul24((uint32_t)clampi(c.jb + 1, 0, (int)dim - 1), dim) + cb;
    }
    c.a0 = load_pair(P.shadow[J], a0);
    c.a1 = load_pair(P.shadow[J], a1);
    c.b0 = load_pair(P.shadow[J + 1], b0);
    c.b1 = load_pair(P.shadow[J + 1], b1);
}
// pair_at's picks out of a loaded pair (cx, cx + 1), cx = clamp(i0, 0, dim - 2)
CRY_HD TexelPair pair_pick(RawPair v, int i0, uint32_t dim)
{
    const int cx = clampi(i0, 0, (int)dim - 2);
    return TexelPair{ (i0 == cx) ? v.lo : v.hi, (i0 + 1 == cx) ? v.lo : v.hi };
}
CRY_HD float cascade_uniform_resolve(const LightParams& P, const CascadeTexels& c)
{
    float a, b;
    if (c.inside) {
        const v2f one = splat(1.0f), zero = splat(0.0f);
        const v2f c00 = select2(c.z <= d24_to_float2(c.a0.lo, c.b0.lo), one, zero), c10 = select2(c.z <= d24_to_float2(c.a0.hi, c.b0.hi), one, zero);
        const v2f c01 = select2(c.z <= d24_to_float2(c.a1.lo, c.b1.lo), one, zero), c11 = select2(c.z <= d24_to_float2(c.a1.hi, c.b1.hi), one, zero);
        const v2f t = lerp2(lerp2(c00, c10, c.fx), lerp2(c01, c11, c.fx), c.fy);      // bilerp per lane
        a = t.x;
        b = t.y;
    } else {
        const ShadowFetch f0{ pair_pick(c.a0, c.ia, P.shadowDim), pair_pick(c.a1, c.ia, P.shadowDim), Bilin{ c.ia, c.ja, c.fx.x, c.fy.x } };
        const ShadowFetch f1{ pair_pick(c.b0, c.ib, P.shadowDim), pair_pick(c.b1, c.ib, P.shadowDim), Bilin{ c.ib, c.jb, c.fx.y, c.fy.y } };
        a = shadow_resolve<true>(f0, P.shadowDim, c.z.x);
        b = shadow_resolve<true>(f1, P.shadowDim, c.z.y);
    }
#if defined(__HIP_DEVICE_COMPILE__)
    // pcf_zero_radius, both lookups under one vote: fully lit / fully shadowed footprints come out of the 16 additions unchanged
    if (__builtin_amdgcn_ballot_w64(!(((a == 0.0f) | (a == 1.0f)) & ((b == 0.0f) | (b == 1.0f)))) == 0) return 0.5f * (a + b);
#endif
    return 0.5f * (pcf_zero_radius(a) + pcf_zero_radius(b));     // :66
}

// DeferredShading.hlsl:23-101 for one covered pixel.  Every gather of the pixel -- ambient map, cubemap, shadow cascades -- is
// issued before the first is waited for (one memory round trip instead of one per lookup).
// FIX: a compile-time promise that P.flags may carry CRYCHIC_FIX_* bits; false = the reference as written, with no trace of
// the switches in the instantiation the benchmark runs.
// DeferredShading.hlsl:32,94 + GBuffer.hlsl:41: the direction light_pixel looks the cube map up with (the same operations in the same
// order; a kernel that needs it before the call -- the quad derivatives of the chain lookup -- gets the identical value).
CRY_HD f3 reflection_dir(const LightParams& P, f4a G0, f4a G2)
{
    const f3 normalW = normalize3(f3{ G2.x, G2.y, G2.z });
    const f3 toEye{ P.EyePosW[0] - G0.x, P.EyePosW[1] - G0.y, P.EyePosW[2] - G0.z };
    const f3 view = normalize3(toEye);
    return reflect3(f3{ -view.x, -view.y, -view.z }, normalW);
}

template <bool ZERO_RADIUS, class PointLights = NoPointLights, bool FIX = false, class Cube = CubeLevel0>
CRY_HD f4 light_pixel(const LightParams& P, f4a G0, f4a G1, f4a G2, const uint16_t* __restrict__ ambient,
                      const uint32_t* __restrict__ cube, PointLights pointLights = PointLights(), Cube cubeLookup = Cube())
{
    const bool fixQ1 = FIX && (P.flags & CRYCHIC_FIX_Q1), fixQ3 = FIX && (P.flags & CRYCHIC_FIX_Q3), fixQ4 = FIX && (P.flags & CRYCHIC_FIX_Q4);
    const f3 posW{ G0.x, G0.y, G0.z };                         // GBuffer.hlsl:37-41
    const float metalness = G0.w;
    const f3 albedo{ G1.x, G1.y, G1.z };
    const float roughness = G1.w;
    const f3 normalW = normalize3(f3{ G2.x, G2.y, G2.z });

    const f3 toEye{ P.EyePosW[0] - posW.x, P.EyePosW[1] - posW.y, P.EyePosW[2] - posW.z };
    // :53-76  cascade selection and shadow factor of light 0: a wavefront that cannot take the packed lookup settles the factor here
    const float d2Eye = dot3(toEye, toEye);
    const float distance = len_from_sq(d2Eye);
    float shadow0 = 1.0f;
    int cascadeJ;
    const bool packedCascades = cascade_uniform_test<ZERO_RADIUS>(P, posW, distance, fixQ1, cascadeJ);
    if (!packedCascades) shadow0 = cascade_shadow<ZERO_RADIUS>(P, posW, distance, fixQ1);
    const f3 view = normalize3(toEye);                          // :32
    const f3 R0{ lerpf(0.04f, albedo.x, metalness), lerpf(0.04f, albedo.y, metalness),
                 lerpf(0.04f, albedo.z, metalness) };           // :35

    // :40-42; without an ambient map the fetch still runs, on a 1 x 1 stand-in (the cubemap's first bytes), so that no branch
    // separates it from the other gathers
    const bool hasAO = ambient != nullptr;
    const AmbientPairs af = ambient_fetch_projected(P, ambient, hasAO, (const uint16_t*)cube, posW);
    const f3 r = reflect3(f3{ -view.x, -view.y, -view.z }, normalW);  // :94
    const typename Cube::Fetch cf = cubeLookup.fetch(P, cube, r);   // :95

    CascadeTexels ct;
    if (packedCascades) {
        cascade_uniform_fetch(P, posW, cascadeJ, ct);             // texels in flight with the ambient map's and the cubemap's
        shadow0 = cascade_uniform_resolve(P, ct);
    }

    const float ambientAccess = hasAO ? ambient_resolve(P, ambient, af) : 1.0f;
    // the reflection lookup is filtered here as well -- every gather of the pixel has arrived with the cascades' texels -- so that
    // what stays live across the lights is three colours and one Fresnel factor, not two texel pairs, weights and the vector
    const f4 refl = cubeLookup.resolve(P, cube, cf);
    const float f0 = 1.0f - saturate(dot3(normalW, r));         // LightingUtil.hlsl:54-57
    const float f5 = f0 * f0 * f0 * f0 * f0;
    const f3 amb{ ambientAccess * P.AmbientLight[0] * albedo.x, ambientAccess * P.AmbientLight[1] * albedo.y,
                  ambientAccess * P.AmbientLight[2] * albedo.z };  // :44

    const float shininess = (1.0f - roughness) * 1.0f;          // :84 (normalW.a == 1)

    // "dark lights" above: with a dark light in the list and bounded inputs on every pixel of the wavefront, the dark lights are
    // skipped and the others -- whose directions light_dark_lengths_ok() vouches for -- take the shorter reciprocals
    bool bounded = false;
    if (P.darkLights) {
        // ... and a finite view vector: |toEye|^2 < 1e30 (false for NaN) bounds EyePosW - posW, so `view` is finite and no longer than
        // 1 + ulps.  Without it an infinite EyePosW makes hDotv = +inf, and rcp_normal(+inf) is NaN where rcp(+inf) is 0 (found by
        // tests/test_gpu_parity.py::test_dark_light_skip_on_device's eye cases, round 4).
        bounded = light_dark_guard(G0, G1, G2) & (d2Eye < 1.0e30f);
#if defined(__HIP_DEVICE_COMPILE__)
        bounded = __builtin_amdgcn_ballot_w64(!bounded) == 0;   // wave-uniform: the loop below stays converged
#endif
    }
    const uint32_t dark = bounded ? P.darkLights : 0u;
    const bool shortRcp = bounded && P.unitLights;
    f3 direct{ 0.0f, 0.0f, 0.0f };
    for (int i = 0; i < P.numDirLights; ++i) {                  // PBR.hlsl:99-106; shadowFactors[i>0] == 1 (:46-51)
        if ((dark >> i) & 1u) continue;
        if (shortRcp) pbr_dir_light<true>(P.Lights[i], albedo, roughness, metalness, normalW, view, i == 0 ? shadow0 : 1.0f, direct, fixQ3, fixQ4);
        else pbr_dir_light<false>(P.Lights[i], albedo, roughness, metalness, normalW, view, i == 0 ? shadow0 : 1.0f, direct, fixQ3, fixQ4);
    }
    pointLights(posW, albedo, roughness, metalness, normalW, view, direct, fixQ3, fixQ4);   // extension; a no-op in the reference configuration
    f4 lit;
    const v2f d2{ direct.x, direct.y };
    const v2f tm = pow_inv_gamma2(d2 * rcp2(d2 + 1.0f)) + v2f{ amb.x, amb.y };     // :89-92 pow(x / (x + 1), 1 / 2.2), red and green packed
    lit.z = pow_inv_gamma(divf(direct.z, direct.z + 1.0f)) + amb.z;

    const v2f R02{ R0.x, R0.y };
    const v2f spec = fma2(shininess * fma2(1.0f - R02, splat(f5), R02), v2f{ refl.x, refl.y }, tm);  // :97
    lit.x = spec.x;
    lit.y = spec.y;
    lit.z = fma(shininess * fma(1.0f - R0.z, f5, R0.z), refl.z, lit.z);
    lit.w = 1.0f;                                               // :99
    return lit;
}

// sky.hlsl:21-47 for an uncovered pixel: cubemap lookup along the pixel's view ray.
CRY_HD f3 sky_direction(const LightParams& P, uint32_t x, uint32_t y)
{
    const float u = ((float)x + 0.5f) * P.rcpW, v = ((float)y + 0.5f) * P.rcpH;         // (x + 0.5) / W as a * rcp(b)
    const float hx = fma(2.0f, u, -1.0f), hy = fma(-2.0f, v, 1.0f);
    const float phx = mulcol(hx, hy, 0.0f, 1.0f, P.InvProj + 0);
    const float phy = mulcol(hx, hy, 0.0f, 1.0f, P.InvProj + 4);
    const float phz = mulcol(hx, hy, 0.0f, 1.0f, P.InvProj + 8);
    const float phw = mulcol(hx, hy, 0.0f, 1.0f, P.InvProj + 12);
    const float rphw = rcp(phw);
    const float vx = phx * rphw, vy = phy * rphw, vz = phz * rphw;
    return f3{ mulcol(vx, vy, vz, 0.0f, P.InvView + 0), mulcol(vx, vy, vz, 0.0f, P.InvView + 4),
               mulcol(vx, vy, vz, 0.0f, P.InvView + 8) };
}
CRY_HD f4 sky_pixel(const LightParams& P, const uint32_t* __restrict__ cube, uint32_t x, uint32_t y)
{
    return cube_linear(cube, P.cubeDim, sky_direction(P, x, y));
}
// The same with the chain: the quad neighbours' directions are those of their own pixels (outside the frame: zero derivative).
CRY_HD f4 sky_pixel_chain(const LightParams& P, const uint32_t* __restrict__ cube, uint32_t x, uint32_t y)
{
    const f3 d = sky_direction(P, x, y);
    f3 ddx{ 0.0f, 0.0f, 0.0f }, ddy{ 0.0f, 0.0f, 0.0f };
    if ((x ^ 1u) < P.W) { const f3 n = sky_direction(P, x ^ 1u, y); ddx = (x & 1u) ? f3{ d.x - n.x, d.y - n.y, d.z - n.z } : f3{ n.x - d.x, n.y - d.y, n.z - d.z }; }
    if ((y ^ 1u) < P.H) { const f3 n = sky_direction(P, x, y ^ 1u); ddy = (y & 1u) ? f3{ d.x - n.x, d.y - n.y, d.z - n.z } : f3{ n.x - d.x, n.y - d.y, n.z - d.z }; }
    const float lod = cube_lod(P.cubeDim, P.cubeLevels, d, ddx, ddy);
    if (cube_chain_flat(lod)) return cube_linear(cube, P.cubeDim, d);           // magnified for every sky lane of the wavefront: level 0 as before
    return cube_trilinear<true>(cube, P.cubeDim, P.cubeLevels, d, lod);
}

CRY_HD uint32_t pack_rgba8(f4 c)
{
    return float_to_unorm8(c.x) | (float_to_unorm8(c.y) << 8) | (float_to_unorm8(c.z) << 16) | (float_to_unorm8(c.w) << 24);
}

}  // namespace cry
