// devmath.hpp -- arithmetic building blocks of the CRYCHIC hot-path kernels (gfx950).
//
// Every function is IEEE-754 binary32 with a fixed evaluation order (DESIGN.md "Oracle definitions", version 2).  The
// translation units that include this header are built with -ffp-contract=off: a*b+c is fused exactly where fma() is
// written -- dot products, matrix rows, lerps and every Horner step -- and nowhere else.  Division is a * rcp(b) with rcp
// the correctly rounded, flush-to-zero reciprocal; lengths use the correctly rounded square root of an argument clamped
// to [2^-100, 2^100].  On the device those are v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 plus one correction step each, proven
// equal to the definitions for EVERY binary32 input by tools/exact_math_probe.hip (tests/test_gpu_exact_math.py runs it);
// on the host (tests/hostsim) they are the definitions themselves.  Transcendentals follow fixed polynomial recurrences,
// so results do not depend on a vendor math library.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define CRY_HD __host__ __device__ __forceinline__
#else
#define CRY_HD inline
#endif

namespace cry {

struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };
struct alignas(8) u2 { uint32_t x, y; };
struct alignas(16) u4 { uint32_t x, y, z, w; };
struct alignas(16) f4a { float x, y, z, w; };

CRY_HD uint32_t f2u(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
CRY_HD float u2f(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }

// HLSL saturate(): NaN -> 0.  fmin(fmax(x, 0), 1) is one v_max_f32 ... clamp on gfx950 (the select form costs four
// instructions).  It may return -0 where the select form returns +0; every use multiplies the result into a sum or
// quantises it, where the sign of a zero cannot reach an output bit.
CRY_HD float saturate(float x) { return __builtin_fminf(__builtin_fmaxf(x, 0.0f), 1.0f); }
CRY_HD float maxnn(float x, float c) { return (x > c) ? x : c; }            // HLSL max(): NaN loses
CRY_HD float signf(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }
CRY_HD float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
CRY_HD float lerpf(float a, float b, float t) { return fma(t, b - a, a); }

// ---- exact reciprocal / square root ------------------------------------------------------------------------------
// rcp(b): the correctly rounded 1/b; subnormal b -> +-inf, subnormal results -> +-0 (|b| > 2^126), NaN -> NaN.
// Device: v_rcp_f32 (1 ulp) + one Newton step is correctly rounded for every normal b with a normal reciprocal; whenever
// the seed is not a normal number (0, inf, NaN) it already is the defined result.  5 VALU instructions against 11 for the
// IEEE quotient expansion (v_div_scale x2, v_rcp, 4 fma, mul, v_div_fmas, v_div_fixup).
CRY_HD float rcp(float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float e = fma(-b, r0, 1.0f);
    const float r1 = fma(e, r0, r0);
    return __builtin_isnormal(r0) ? r1 : r0;
#else
    if (b != b) return b;
    const float ab = __builtin_fabsf(b);
    if (ab < 1.17549435e-38f) return __builtin_copysignf(__builtin_inff(), b);
    if (ab > 8.50705917e37f) return __builtin_copysignf(0.0f, b);
    return 1.0f / b;
#endif
}
// The same for an argument known to be a normal number with a normal reciprocal (2^-126 <= |b| <= 2^126): no select.
CRY_HD float rcp_normal(float b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float r0 = __builtin_amdgcn_rcpf(b);
    return fma(fma(-b, r0, 1.0f), r0, r0);
#else
    return 1.0f / b;
#endif
}
CRY_HD float divf(float a, float b) { return a * rcp(b); }     // HLSL a / b
// Squared length clamped to [2^-100, 2^100] (NaN -> 2^-100: v_med3_f32 returns the minimum when an operand is NaN).
CRY_HD float clamp_len2(float d)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_fmed3f(d, 7.8886090522101181e-31f, 1.2676506002282294e30f);
#else
    return (d != d) ? 7.8886090522101181e-31f : __builtin_fminf(__builtin_fmaxf(d, 7.8886090522101181e-31f), 1.2676506002282294e30f);
#endif
}
// sqrt of an argument in [2^-100, 2^100], correctly rounded: v_sqrt_f32 (1 ulp) + one Markstein correction with
// h = rsq/2 -- exhaustively equal to IEEE sqrtf on that range (the residual x - g*g would underflow below it).
CRY_HD float sqrt_clamped(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float g = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rsqf(x);
    return fma(fma(-g, g, x), h, g);
#else
    return __builtin_sqrtf(x);
#endif
}
CRY_HD float len_from_sq(float d2) { return sqrt_clamped(clamp_len2(d2)); }                  // length(v), d2 = dot(v, v)
CRY_HD float inv_len_from_sq(float d2) { return rcp_normal(sqrt_clamped(clamp_len2(d2))); }  // 1 / length(v)

CRY_HD float dot3(f3 a, f3 b) { return fma(a.z, b.z, fma(a.y, b.y, a.x * b.x)); }
CRY_HD f3 normalize3(f3 v)
{
    const float inv = inv_len_from_sq(dot3(v, v));
    return f3{ v.x * inv, v.y * inv, v.z * inv };
}
CRY_HD f3 reflect3(f3 i, f3 n)
{
    const float d2 = 2.0f * dot3(n, i);
    return f3{ fma(-d2, n.x, i.x), fma(-d2, n.y, i.y), fma(-d2, n.z, i.z) };
}
// HLSL mul(float4(v), M) for a matrix stored transposed (the reference's cbuffer layout): column j of the
// row-vector matrix is mem[4j .. 4j+3].
CRY_HD float mulcol(float x, float y, float z, float w, const float* col)
{
    return fma(w, col[3], fma(z, col[2], fma(y, col[1], x * col[0])));
}
// w == 1: mad(1, c3, acc) == acc + c3
CRY_HD float mulcol1(float x, float y, float z, const float* col)
{
    return fma(z, col[2], fma(y, col[1], x * col[0])) + col[3];
}

// ---- two-wide packed fp32 ---------------------------------------------------------------------------------------
// On gfx950 a wave64 VALU instruction occupies its SIMD for 4 cycles whether it is v_mul_f32 or v_pk_mul_f32 (measured:
// 4.1 cycles per VALU instruction in the VALU-bound SSAO loop), and the packed forms do two lanes' worth of IEEE-754
// work per instruction.  The tap loops therefore process independent work items in pairs: `v2f` arithmetic compiles to
// v_pk_mul_f32 / v_pk_add_f32 (never fused: -ffp-contract=off), each lane bit-identical to the scalar expression.
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));

CRY_HD v2f splat(float a) { return v2f{ a, a }; }
CRY_HD v2f select2(v2i m, v2f a, v2f b) { return m ? a : b; }          // per lane: mask all-ones -> a
CRY_HD v2f saturate2(v2f x) { return v2f{ saturate(x.x), saturate(x.y) }; }   // NaN -> 0
CRY_HD v2f max0_2(v2f x) { return select2(x > 0.0f, x, splat(0.0f)); }  // HLSL max(x, 0): NaN -> 0
// sign(): +-1 with the sign bit of x copied in (v_bfi_b32), 0 for +-0 and NaN -- the same values as (x > 0) - (x < 0)
CRY_HD float sign1(float x) { return ((x < 0.0f) | (x > 0.0f)) ? __builtin_copysignf(1.0f, x) : 0.0f; }
CRY_HD v2f sign2(v2f x) { return v2f{ sign1(x.x), sign1(x.y) }; }
CRY_HD v2f floor2(v2f x) { return v2f{ __builtin_floorf(x.x), __builtin_floorf(x.y) }; }
CRY_HD v2f sqrt2(v2f x) { return v2f{ __builtin_sqrtf(x.x), __builtin_sqrtf(x.y) }; }
CRY_HD v2f lerp2(v2f a, v2f b, v2f t) { return a + t * (b - a); }
// Two correctly rounded divisions at once.  The device path is LLVM's own f32 fdiv expansion (v_div_scale x2, v_rcp,
// Newton step, quotient refinement, v_div_fmas, v_div_fixup) written out so that the six FMA/MUL steps of the two
// quotients issue as packed instructions: 16 instead of 22 VALU instructions per pair, bit-identical to n / d
// (same operations, same special-case fix-up).  The host build simply divides.
CRY_HD v2f div2(v2f n, v2f d)
{
#if defined(__HIP_DEVICE_COMPILE__) && !defined(CRYCHIC_RELAXED_MATH_PROBE)   // probe build: tools/relaxed_math_probe.py
    bool na, nb, da, db;
    const v2f ds{ __builtin_amdgcn_div_scalef(n.x, d.x, false, &da), __builtin_amdgcn_div_scalef(n.y, d.y, false, &db) };
    const v2f ns{ __builtin_amdgcn_div_scalef(n.x, d.x, true, &na), __builtin_amdgcn_div_scalef(n.y, d.y, true, &nb) };
    const v2f r{ __builtin_amdgcn_rcpf(ds.x), __builtin_amdgcn_rcpf(ds.y) };
    const v2f nd = -ds;
    const v2f f0 = __builtin_elementwise_fma(nd, r, v2f{ 1.0f, 1.0f });
    const v2f f1 = __builtin_elementwise_fma(f0, r, r);
    const v2f m = ns * f1;
    const v2f f2 = __builtin_elementwise_fma(nd, m, ns);
    const v2f f3 = __builtin_elementwise_fma(f2, f1, m);
    const v2f f4 = __builtin_elementwise_fma(nd, f3, ns);
    return v2f{ __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(f4.x, f1.x, f3.x, na), d.x, n.x),
                __builtin_amdgcn_div_fixupf(__builtin_amdgcn_div_fmasf(f4.y, f1.y, f3.y, nb), d.y, n.y) };
#else
    return n / d;
#endif
}
CRY_HD v2f div2(float n, v2f d) { return div2(v2f{ n, n }, d); }
CRY_HD v2f div2(v2f n, float d) { return div2(n, v2f{ d, d }); }
struct f3x2 { v2f x, y, z; };                                          // two 3-vectors, component-packed
CRY_HD v2f dot3x2(f3x2 a, f3x2 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
CRY_HD f3x2 splat3(f3 a) { return f3x2{ splat(a.x), splat(a.y), splat(a.z) }; }

// ---- format decoders -------------------------------------------------------------------------------
// D24 -> float == (float)u / 16777215.0f for every u in [0, 2^24): q = u * 2^-24 is exact and the quotient
// is q * (1 + 2^-24 + ...), i.e. q plus a correction strictly between 0.5 and 1 ulp(q), so the correctly
// rounded result is the next float above q.
CRY_HD float d24_to_float(uint32_t texel)
{
    uint32_t u = texel & 0x00FFFFFFu;
    float q = (float)u * 5.9604644775390625e-8f;
    return u2f(f2u(q) + (u != 0u ? 1u : 0u));   // u == 0: q is +0.0 and stays +0.0
}
// u / 65535.0f and u / 255.0f, correctly rounded, via one reciprocal multiply and one residual correction.
CRY_HD float unorm16_to_float(uint32_t u)
{
    const float c = 1.0f / 65535.0f;
    float a = (float)u;
    float t = a * c;
    float r = __builtin_fmaf(-t, 65535.0f, a);
    return __builtin_fmaf(r, c, t);
}
CRY_HD float unorm8_to_float(uint32_t u)
{
    const float c = 1.0f / 255.0f;
    float a = (float)u;
    float t = a * c;
    float r = __builtin_fmaf(-t, 255.0f, a);
    return __builtin_fmaf(r, c, t);
}
CRY_HD uint32_t float_to_unorm16(float x) { return (uint32_t)(saturate(x) * 65535.0f + 0.5f); }
CRY_HD uint32_t float_to_unorm8(float x) { return (uint32_t)(saturate(x) * 255.0f + 0.5f); }

CRY_HD float half_to_float(uint16_t h)
{
    _Float16 v;
    __builtin_memcpy(&v, &h, 2);
    return (float)v;
}

// ---- deterministic transcendentals -------------------------------------------------------------------
// sin/cos: 3-term Cody-Waite reduction by pi/2, degree-7 / degree-8 kernels on [-pi/4, pi/4].
CRY_HD float det_sincos(float x, int want_cos)
{
    float ax = __builtin_fabsf(x);
    if (!(ax < 8388608.0f)) return x - x;
    float k = __builtin_rintf(x * 0.636619772367581343f);
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
CRY_HD float det_sin(float x) { return det_sincos(x, 0); }
CRY_HD float det_cos(float x) { return det_sincos(x, 1); }

CRY_HD float det_log2(float x)
{
    if (x != x) return x;
    if (x < 0.0f) return u2f(0x7FC00000u);
    if (x == 0.0f) return u2f(0xFF800000u);
    if (x == u2f(0x7F800000u)) return x;
    uint32_t u = f2u(x);
    int e = (int)(u >> 23) - 126;
    if ((u >> 23) == 0) {
        x = x * 16777216.0f;
        u = f2u(x);
        e = (int)(u >> 23) - 126 - 24;
    }
    float m = u2f((u & 0x007FFFFFu) | 0x3F000000u);
    if (m < 0.70710678118654752440f) { e -= 1; m = m + m - 1.0f; }
    else { m = m - 1.0f; }
    float z = m * m;
    float y = ((((((((7.0376836292e-2f * m - 1.1514610310e-1f) * m + 1.1676998740e-1f) * m - 1.2420140846e-1f) * m
                  + 1.4249322787e-1f) * m - 1.6668057665e-1f) * m + 2.0000714765e-1f) * m - 2.4999993993e-1f) * m
               + 3.3333331174e-1f) * m * z;
    y = y - 0.5f * z;
    float r = y * 0.44269504088896340735992f;
    r = r + m * 0.44269504088896340735992f;
    r = r + y;
    r = r + m;
    r = r + (float)e;
    return r;
}

CRY_HD float det_exp2(float x)
{
    if (x != x) return x;
    if (x >= 128.0f) return u2f(0x7F800000u);
    if (x < -126.0f) return 0.0f;
    float n = __builtin_floorf(x + 0.5f);
    float f = x - n;
    float p = (((((1.535336188319500e-4f * f + 1.339887440266574e-3f) * f + 9.618437357674640e-3f) * f
                 + 5.550332471162809e-2f) * f + 2.402264791363012e-1f) * f + 6.931472028550421e-1f) * f + 1.0f;
    int ni = (int)n;
    if (ni > 127) { p = p * 2.0f; ni = 127; }
    return p * u2f((uint32_t)(ni + 127) << 23);
}

CRY_HD float det_pow(float x, float y)
{
    if (x != x) return x;
    if (x < 0.0f) return u2f(0x7FC00000u);
    if (x == 0.0f) return 0.0f;
    return det_exp2(y * det_log2(x));
}

// ---- bilinear addressing --------------------------------------------------------------
struct Bilin { int i0, j0; float fx, fy; };

CRY_HD int texel_index(float fl, uint32_t dim)
{
    // clamp(fl, -2, dim + 1) with NaN -> -2 (fmax returns the non-NaN operand); fl is already integral
    const float c = __builtin_fminf(__builtin_fmaxf(fl, -2.0f), (float)dim + 1.0f);
    return (int)c;
}
CRY_HD Bilin bilinear_setup(float u, float v, uint32_t w, uint32_t h)
{
    Bilin b;
    const float tx = u * (float)w - 0.5f;
    const float ty = v * (float)h - 0.5f;
    const float flx = __builtin_floorf(tx), fly = __builtin_floorf(ty);
    const float fx = tx - flx, fy = ty - fly;
    const bool bad = !(fx == fx) | !(fy == fy);   // non-finite coordinates address only out-of-range texels
    b.fx = bad ? 0.0f : fx;
    b.fy = bad ? 0.0f : fy;
    b.i0 = bad ? -2 : texel_index(flx, w);
    b.j0 = bad ? -2 : texel_index(fly, h);
    return b;
}
CRY_HD float bilerp(float t00, float t10, float t01, float t11, float fx, float fy)
{
    float top = lerpf(t00, t10, fx);
    float bot = lerpf(t01, t11, fx);
    return lerpf(top, bot, fy);
}
CRY_HD int clampi(int i, int lo, int hi) { return i < lo ? lo : (i > hi ? hi : i); }

// Plane addressing.  Texel indices are combined with 24-bit multiplies (full-rate v_mul_u32_u24; rows and widths are
// far below 2^24) into 32-bit BYTE offsets from a wave-uniform base, which the compiler turns into
// `global_load ... v_off, s[base]` (SGPR base + 32-bit VGPR offset) instead of quarter-rate 64-bit pointer arithmetic
// (v_mul_lo_u32 + v_lshl_add_u64 per texel: measured 4.8 cycles per VALU instruction in the SSAO tap loop before this).
// Every plane the kernels gather from is smaller than 4 GiB (checked at the API).
CRY_HD uint32_t mul24(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return a * b;
#endif
}
template <class T>
CRY_HD T load_at(const void* __restrict__ base, uint32_t byteOffset)
{
    T v;
    __builtin_memcpy(&v, (const char*)base + byteOffset, sizeof(T));
    return v;
}

// Two horizontally adjacent texels of a row-major 32-bit plane with one 8-byte load (the address is only 4-byte
// aligned: gfx950 global loads allow that).  `row` is an in-range row index, width >= 2.
struct TexelPair { uint32_t a, b; };
struct RawPair { uint32_t lo, hi; };
CRY_HD RawPair load_pair(const uint32_t* __restrict__ plane, uint32_t texelIndex)
{
    return load_at<RawPair>(plane, texelIndex * 4u);
}
// BORDER / range-checked variant: returns texels i0 and i0+1 where they exist (the caller replaces out-of-range
// ones by the border value; what is returned for them is unspecified but always read from valid memory).
CRY_HD TexelPair pair_at(const uint32_t* __restrict__ plane, uint32_t row, uint32_t width, int i0)
{
    const int cx = clampi(i0, 0, (int)width - 2);
    const RawPair v = load_pair(plane, mul24(row, width) + (uint32_t)cx);
    TexelPair t;
    t.a = (i0 == cx) ? v.lo : v.hi;      // i0 == width-1 -> hi
    t.b = (i0 + 1 == cx) ? v.lo : v.hi;  // i0 == -1      -> lo
    return t;
}
// CLAMP variant: texels clamp(i0) and clamp(i0+1).
CRY_HD TexelPair pair_at_clamped(const uint32_t* __restrict__ plane, uint32_t row, uint32_t width, int i0)
{
    const int cx = clampi(i0, 0, (int)width - 2);
    const RawPair v = load_pair(plane, mul24(row, width) + (uint32_t)cx);
    TexelPair t;
    t.a = (i0 > (int)width - 2) ? v.hi : v.lo;
    t.b = (i0 < 0) ? v.lo : v.hi;
    return t;
}

}  // namespace cry
