/*
 * or_math.h -- oracle arithmetic definitions, version 2 (TEST INFRASTRUCTURE, see crychic_oracle.h).
 *
 * Everything here is IEEE-754 binary32 with one fixed evaluation order.  HLSL leaves the precision of mul / dot / lerp /
 * mad contraction, `/`, rcp, rsqrt, sqrt, pow, sin and cos open (D3D11.3 functional spec: 1 ULP class operations, fused or
 * unfused mad at the compiler's choice; fxc emits mad and rcp-multiply freely), so the oracle DEFINES one evaluation:
 *
 *   - a*b+c chains are FUSED exactly where written: fmaf() below, nowhere else (build with -ffp-contract=off);
 *   - a / b      := a * or_rcp(b); or_rcp is the correctly rounded reciprocal with flush-to-zero on subnormal
 *                   inputs and results (what `v_rcp_f32` + one Newton step yields on gfx950 for EVERY binary32 input:
 *                   tools/exact_math_probe.hip, tests/test_gpu_exact_math.py);
 *   - length(v)  := sqrtf(clamp(dot(v,v), 2^-100, 2^100)), normalize(v) := v * (1 / length): IEEE sqrt and division on
 *                   an argument clamped to a range where neither can meet a subnormal (NaN -> 2^-100);
 *   - pow(x, y)  := the fixed exp2(y * log2(x)) kernels below; sin / cos: fixed Cody-Waite + polynomial kernels;
 *   - pow(x, 1/2.2), the tone map's only pow (DeferredShading.hlsl:90), := SCALE[E] * P(m - 1) (or_pow_inv_gamma below, version 3).
 *
 * Version 1 (round 1) used unfused chains and IEEE division everywhere; version 2 exists because those choices cost the
 * GPU kernels ~40 % of their instruction issue without being any closer to what a D3D12 driver computes.  DESIGN.md
 * "Oracle definitions" lists every definition.
 *
 * DEFINITIONS THAT ARE NOT RESTATEMENTS.  The items above fix precision HLSL leaves open.  Two go further and fix BEHAVIOUR on
 * inputs where HLSL's own result is different in kind -- they are build-defined semantics, not a reading of the reference:
 *   - normalize(v) of a zero / denormal-length (or NaN-length) vector.  HLSL: v * rsqrt(0) = 0 * inf = NaN, which the
 *     shaders' saturate() / max() then turn into 0 (a zero G-buffer or view normal renders BLACK on D3D: `normalize` at
 *     Ssao.hlsl:125, GBuffer.hlsl:41, DeferredShading.hlsl:32, PBR.hlsl:53).  Here: the squared length is clamped to
 *     [2^-100, 2^100] first (or_clamp_len2), so the result is the FINITE vector v * 2^50 -- (0,0,0) for an exact zero -- and the
 *     pixel gets ambient access 1 / finite radiance.  Unreachable with the reference's producers (normals are normalised in
 *     the vertex shaders, the normal map is cleared to (0,0,1,0), CRYCHIC.cpp:2526; the G-buffer is lit only under covered
 *     pixels), reachable through the C ABI with hand-made planes.  tests/test_numpy_restatements.py::
 *     test_zero_normal_is_a_definition shows the two behaviours side by side.
 *   - pow(x, y) outside x > 0 (or_pow: x = 0 or subnormal -> 0, x < 0 or NaN -> NaN; or_pow_inv_gamma: +-0 and subnormals of
 *     either sign -> 0, as D3D's denormal flush gives, negative normals and NaN -> NaN).  HLSL pow(x, y) = exp2(y * log2(x)) gives
 *     NaN for x < 0 as well, 0 for x = 0 and is unspecified on subnormals; the only call with a computed base is the tone map
 *     pow(direct / (direct + 1), 1 / 2.2) (DeferredShading.hlsl:89-90), whose base is >= 0 unless an input is negative or NaN.
 * Both are shared bit for bit by the kernels (csrc/devmath.hpp) and pinned by the fuzz tests (tests/test_fuzz.py: zero, NaN and
 * infinite vectors in every plane).
 */
#ifndef OR_MATH_H
#define OR_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

static inline float or_bits_to_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t or_float_to_bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* HLSL saturate: NaN -> 0. */
static inline float or_saturate(float x) { return (x > 0.0f) ? ((x < 1.0f) ? x : 1.0f) : 0.0f; }
/* HLSL max(x, c): returns the non-NaN operand. */
static inline float or_max0(float x, float c) { return (x > c) ? x : c; }
/* HLSL sign(): -1, 0, +1 (NaN -> 0). */
static inline float or_sign(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }
/* HLSL frac(x) = x - floor(x). */
static inline float or_frac(float x) { return x - floorf(x); }
/* HLSL lerp(a,b,t) = a + t*(b-a), as one mad. */
static inline float or_lerp(float a, float b, float t) { return fmaf(t, b - a, a); }

/* Reciprocal: correctly rounded 1/b with subnormal inputs and results flushed (to +-inf / +-0).  |b| > 2^126 is exactly
 * the set of normal b whose reciprocal is subnormal. */
static inline float or_rcp(float b)
{
    if (b != b) return b;
    float ab = fabsf(b);
    if (ab < 1.17549435e-38f) return copysignf(INFINITY, b);
    if (ab > 8.50705917e37f) return copysignf(0.0f, b);
    return 1.0f / b;
}
/* HLSL a / b. */
static inline float or_div(float a, float b) { return a * or_rcp(b); }
/* Squared length -> length / inverse length on the clamped argument (NaN -> the lower bound). */
static inline float or_clamp_len2(float d) { return (d != d) ? 7.8886090522101181e-31f : fminf(fmaxf(d, 7.8886090522101181e-31f), 1.2676506002282294e30f); }
static inline float or_len(float d2) { return sqrtf(or_clamp_len2(d2)); }
static inline float or_inv_len(float d2) { return 1.0f / sqrtf(or_clamp_len2(d2)); }

/* HOST code of the reference (DirectXMath on x86: XMVector3Dot and friends, SSE multiplies and adds) is plain unfused IEEE. */
static inline float or_dot3_host(const float a[3], const float b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
/* HLSL dot(a, b): x first, then two mads. */
static inline float or_dot3(const float a[3], const float b[3]) { return fmaf(a[2], b[2], fmaf(a[1], b[1], a[0] * b[0])); }
/* HLSL normalize(v) = v * rsqrt(dot(v,v)). */
static inline void or_normalize3(const float v[3], float out[3])
{
    float inv = or_inv_len(or_dot3(v, v));
    out[0] = v[0] * inv; out[1] = v[1] * inv; out[2] = v[2] * inv;
}
/* HLSL reflect(i, n) = i - 2*dot(n,i)*n  (n need not be unit: Ssao.hlsl:148 passes an un-normalised randVec). */
static inline void or_reflect3(const float i[3], const float n[3], float out[3])
{
    float d2 = 2.0f * or_dot3(n, i);
    out[0] = fmaf(-d2, n[0], i[0]); out[1] = fmaf(-d2, n[1], i[1]); out[2] = fmaf(-d2, n[2], i[2]);
}
/* HLSL mul(float4 v, float4x4 M) with M stored transposed in memory: out[j] = sum_i v[i]*mem[4j+i],
 * x first, then three mads. */
static inline float or_mul_v4_col(const float v[4], const float col[4])
{
    return fmaf(v[3], col[3], fmaf(v[2], col[2], fmaf(v[1], col[1], v[0] * col[0])));
}
static inline void or_mul_v4_m(const float v[4], const float mem[16], float out[4])
{
    for (int j = 0; j < 4; ++j) out[j] = or_mul_v4_col(v, mem + 4 * j);
}

/* IEEE half -> float (exact). */
static inline float or_half_bits_to_float(uint16_t h)
{
    uint32_t s = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1Fu;
    uint32_t m = h & 0x3FFu;
    if (e == 0) {
        if (m == 0) return or_bits_to_float(s);
        /* subnormal: m * 2^-24 */
        float f = (float)m * 5.9604644775390625e-8f;
        return (s ? -f : f);
    }
    if (e == 31) return or_bits_to_float(s | 0x7F800000u | (m << 13));
    return or_bits_to_float(s | ((e + 112u) << 23) | (m << 13));
}

/* ---- deterministic transcendentals ("detmath") ------------------------------------------------
 * HLSL sin/cos/pow/exp2/log2 are hardware approximations with no pinned bits, so the oracle fixes
 * one polynomial evaluation (Horner form, every step one mad).  The HIP kernels implement the same
 * recurrences and must agree bit for bit.                                                         */

static inline float or_det_sincos_core(float x, int want_cos)
{
    if (!(fabsf(x) < 8388608.0f)) return x - x; /* inf, NaN -> NaN; huge finite -> 0 */
    float k = nearbyintf(x * 0.636619772367581343f); /* round-half-even of x*2/pi */
    float r = fmaf(-k, 1.5703125f, x);               /* three-term Cody-Waite reduction by pi/2 */
    r = fmaf(-k, 4.837512969970703125e-4f, r);
    r = fmaf(-k, 7.54978995489188216e-8f, r);
    int q = ((int)k + want_cos) & 3;
    float r2 = r * r;
    float sp = fmaf(fmaf(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f);
    float s = fmaf(sp * r2, r, r);
    float cp = fmaf(fmaf(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f);
    float c = fmaf(cp * r2, r2, fmaf(-0.5f, r2, 1.0f));
    float v = (q & 1) ? c : s;
    return (q & 2) ? -v : v;
}
static inline float or_det_sinf_(float x) { return or_det_sincos_core(x, 0); }
static inline float or_det_cosf_(float x) { return or_det_sincos_core(x, 1); }

/* log2 of a NORMAL positive x: x = m * 2^e with m in [sqrt(1/2), sqrt(2)); log2(m) = s * P(s^2), s = (m-1)/(m+1)
 * (2/ln2 * atanh(s), minimax on |s| <= 0.1716: 7e-10 relative). */
static inline float or_det_log2_normal(float x)
{
    uint32_t ue = or_float_to_bits(x) - 0x3F3504F3u;
    float ef = (float)((int32_t)ue >> 23);
    float m = or_bits_to_float((ue & 0x007FFFFFu) + 0x3F3504F3u);
    float s = (m - 1.0f) * (1.0f / (m + 1.0f));     /* m + 1 in [1.7, 2.42]: IEEE reciprocal == or_rcp */
    float s2 = s * s;
    float p = fmaf(fmaf(fmaf(0.43174004554748535f, s2, 0.5767142176628113f), s2, 0.9617988467216492f), s2, 2.885390043258667f);
    return fmaf(s, p, ef);
}
/* 2^z for z in [-125, 127]: z = n + f, f in [-1/2, 1/2], degree-6 minimax (2e-9 relative), exact scaling. */
static inline float or_det_exp2_clamped(float z)
{
    float n = nearbyintf(z);
    float f = z - n;
    float p = fmaf(fmaf(fmaf(fmaf(fmaf(fmaf(0.00015406982856802642f, f, 0.0013400138122960925f), f, 0.009618260897696018f), f,
                                  0.05550328269600868f), f, 0.24022649228572845f), f, 0.6931471824645996f), f, 1.0f);
    return ldexpf(p, (int)n);
}
static inline float or_det_log2f_(float x)
{
    if (!(x >= 1.17549435e-38f)) return (x >= 0.0f) ? -INFINITY : or_bits_to_float(0x7FC00000u);   /* zero / subnormal; negative, NaN */
    return or_det_log2_normal(x);
}
static inline float or_det_exp2f_(float x)
{
    if (x != x) return x;
    return or_det_exp2_clamped(fminf(fmaxf(x, -125.0f), 127.0f));
}

/* HLSL pow(x, y) = exp2(y * log2(x)) for the exponents the path uses (0 < y <= 1): x below the smallest normal
 * (zero, subnormal) -> 0; negative or NaN base -> NaN; y * log2(x) is clamped to [-125, 127] (unreachable for y <= 1
 * except x = +inf, which therefore evaluates to 2^127). */
static inline float or_det_powf_(float x, float y)
{
    if (!(x >= 0.0f)) return or_bits_to_float(0x7FC00000u);
    if (!(x >= 1.17549435e-38f)) return 0.0f;
    float z = y * or_det_log2_normal(x);
    return or_det_exp2_clamped(fminf(fmaxf(z, -125.0f), 127.0f));
}

/* pow(x, 1/2.2), the tone map (DeferredShading.hlsl:89-90): x = m * 2^(E-127), m in [1, 2) read off the bit pattern;
 * SCALE[E] = RN(2^((E-127)/2.2)) (table over the biased exponent field; [0] = 0, [255] = +inf), P = degree-7 minimax of
 * (1 + u)^(1/2.2) on [0, 1), Horner, one mad per step.  Constants: or_gamma_pow.inc (tools/gen_gamma_pow.py).  <= 3e-7 relative. */
#include "or_gamma_pow.inc"
static inline float or_pow_inv_gamma(float x)
{
    static const float coef[8] = CRY_GAMMA_POW_COEFFS;
    static const float scale[256] = CRY_GAMMA_POW_SCALE;
    if (!(x > -1.17549435e-38f)) return or_bits_to_float(0x7FC00000u);      /* negative normal, NaN */
    uint32_t b = or_float_to_bits(x);
    float u = or_bits_to_float((b & 0x007FFFFFu) | 0x3F800000u) - 1.0f;
    float p = coef[7];
    for (int k = 6; k >= 0; --k) p = fmaf(p, u, coef[k]);
    return scale[(b >> 23) & 255u] * p;
}

#endif
