/*
 * or_math.h -- oracle arithmetic definitions (TEST INFRASTRUCTURE, see crychic_oracle.h).
 *
 * Everything here is IEEE-754 binary32 with one fixed evaluation order and no fused multiply-add
 * (build with -ffp-contract=off).  These are the oracle's DEFINITIONS of the HLSL intrinsics whose
 * precision D3D leaves open; DESIGN.md "Oracle definitions" lists them.
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
/* HLSL lerp(a,b,t) = a + t*(b-a). */
static inline float or_lerp(float a, float b, float t) { return a + t * (b - a); }

static inline float or_dot3(const float a[3], const float b[3]) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
/* HLSL normalize(v) = v * rsqrt(dot(v,v)); defined here as v * (1 / sqrt(dot)). */
static inline void or_normalize3(const float v[3], float out[3])
{
    float inv = 1.0f / sqrtf(or_dot3(v, v));
    out[0] = v[0] * inv; out[1] = v[1] * inv; out[2] = v[2] * inv;
}
/* HLSL reflect(i, n) = i - 2*dot(n,i)*n  (n need not be unit: Ssao.hlsl:148 passes an un-normalised randVec). */
static inline void or_reflect3(const float i[3], const float n[3], float out[3])
{
    float d2 = 2.0f * or_dot3(n, i);
    out[0] = i[0] - d2 * n[0]; out[1] = i[1] - d2 * n[1]; out[2] = i[2] - d2 * n[2];
}
/* HLSL mul(float4 v, float4x4 M) with M stored transposed in memory: out[j] = sum_i v[i]*mem[4j+i],
 * summed left to right. */
static inline void or_mul_v4_m(const float v[4], const float mem[16], float out[4])
{
    for (int j = 0; j < 4; ++j)
        out[j] = ((v[0] * mem[4 * j + 0] + v[1] * mem[4 * j + 1]) + v[2] * mem[4 * j + 2]) + v[3] * mem[4 * j + 3];
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
 * one polynomial evaluation (Cephes single-precision kernels, Horner form, no FMA).  The HIP
 * kernels implement the same recurrences and must agree bit for bit.                            */

static inline float or_det_sincos_core(float x, int want_cos)
{
    if (!(fabsf(x) < 8388608.0f)) return x - x; /* inf, NaN -> NaN; huge finite -> 0 */
    float k = nearbyintf(x * 0.636619772367581343f); /* round-half-even of x*2/pi */
    float r = x - k * 1.5703125f;
    r = r - k * 4.837512969970703125e-4f;
    r = r - k * 7.54978995489188216e-8f;
    int q = ((int)k + want_cos) & 3;
    float r2 = r * r;
    float s = ((-1.9515295891e-4f * r2 + 8.3321608736e-3f) * r2 - 1.6666654611e-1f) * r2 * r + r;
    float c = ((2.443315711809948e-5f * r2 - 1.388731625493765e-3f) * r2 + 4.166664568298827e-2f) * r2 * r2
              - 0.5f * r2 + 1.0f;
    float v = (q & 1) ? c : s;
    return (q & 2) ? -v : v;
}
static inline float or_det_sinf_(float x) { return or_det_sincos_core(x, 0); }
static inline float or_det_cosf_(float x) { return or_det_sincos_core(x, 1); }

static inline float or_det_log2f_(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return or_bits_to_float(0x7FC00000u);
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return x;
    uint32_t u = or_float_to_bits(x);
    int e = (int)(u >> 23) - 126; /* x = m * 2^e, m in [0.5, 1) */
    if ((u >> 23) == 0) {         /* subnormal: scale by 2^24 first */
        x = x * 16777216.0f;
        u = or_float_to_bits(x);
        e = (int)(u >> 23) - 126 - 24;
    }
    float m = or_bits_to_float((u & 0x007FFFFFu) | 0x3F000000u);
    if (m < 0.70710678118654752440f) { e -= 1; m = m + m - 1.0f; }
    else { m = m - 1.0f; }
    float z = m * m;
    float y = ((((((((7.0376836292e-2f * m - 1.1514610310e-1f) * m + 1.1676998740e-1f) * m - 1.2420140846e-1f) * m
                  + 1.4249322787e-1f) * m - 1.6668057665e-1f) * m + 2.0000714765e-1f) * m - 2.4999993993e-1f) * m
               + 3.3333331174e-1f) * m * z;
    y = y - 0.5f * z;
    /* log2(1+m) = (m + y) * log2(e), split as Cephes does */
    float r = y * 0.44269504088896340735992f;
    r = r + m * 0.44269504088896340735992f;
    r = r + y;
    r = r + m;
    r = r + (float)e;
    return r;
}

static inline float or_det_exp2f_(float x)
{
    if (x != x) return x;
    if (x >= 128.0f) return INFINITY;
    if (x < -126.0f) return 0.0f;
    float n = floorf(x + 0.5f);
    float f = x - n; /* [-0.5, 0.5] */
    float p = (((((1.535336188319500e-4f * f + 1.339887440266574e-3f) * f + 9.618437357674640e-3f) * f
                 + 5.550332471162809e-2f) * f + 2.402264791363012e-1f) * f + 6.931472028550421e-1f) * f + 1.0f;
    int ni = (int)n; /* [-126, 128] */
    if (ni > 127) { p = p * 2.0f; ni = 127; }
    return p * or_bits_to_float((uint32_t)(ni + 127) << 23);
}

/* HLSL pow(x, y) = exp2(y * log2(x)) for x > 0; pow(0, y>0) = 0; negative or NaN base -> NaN. */
static inline float or_det_powf_(float x, float y)
{
    if (x != x) return x;
    if (x < 0.0f) return or_bits_to_float(0x7FC00000u);
    if (x == 0.0f) return 0.0f;
    return or_det_exp2f_(y * or_det_log2f_(x));
}

#endif
