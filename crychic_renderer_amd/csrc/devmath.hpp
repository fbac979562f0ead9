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
    // host simulation: an argument outside the promise is answered with NaN, so that a caller whose guard does not cover it shows
    // up in the CPU tier (the device's v_rcp_f32 + Newton step returns NaN or garbage there: +inf gives fma(-inf, 0, 1) = NaN)
    const float ab = __builtin_fabsf(b);
    if (!(ab >= 1.17549435e-38f && ab <= 8.50705917e37f)) return u2f(0x7FC00000u);
    return 1.0f / b;
#endif
}
CRY_HD float divf(float a, float b) { return a * rcp(b); }     // HLSL a / b
// Squared length clamped to [2^-100, 2^100] (NaN -> 2^-100: v_med3_f32 returns the minimum when an operand is a quiet NaN;
// the argument is always an arithmetic result -- a dot product -- so it is never a signaling one).
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
// On gfx950 v_pk_fma_f32 issues in 4.6 cycles against 4.2 for v_fma_f32 (tools/valu_rate.hip): two mads for the price of
// one.  The SSAO tap loop therefore processes two taps at a time: `v2f` arithmetic compiles to v_pk_fma_f32 /
// v_pk_mul_f32 / v_pk_add_f32, each lane bit-identical to the scalar expression (fused only where fma2 is written).
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));

CRY_HD v2f splat(float a) { return v2f{ a, a }; }
CRY_HD v2f select2(v2i m, v2f a, v2f b) { return m ? a : b; }          // per lane: mask all-ones -> a
CRY_HD v2f saturate2(v2f x) { return v2f{ saturate(x.x), saturate(x.y) }; }   // NaN -> 0
CRY_HD v2f max0_2(v2f x) { return select2(x > 0.0f, x, splat(0.0f)); }  // HLSL max(x, 0): NaN -> 0
// sign(): +-1 with the sign bit of x copied in (v_bfi_b32), 0 for +-0 and NaN -- the same values as (x > 0) - (x < 0)
CRY_HD float sign1(float x) { return __builtin_islessgreater(x, 0.0f) ? __builtin_copysignf(1.0f, x) : 0.0f; }      // v_cmp_lg_f32: false for 0 and NaN
CRY_HD v2f sign2(v2f x) { return v2f{ sign1(x.x), sign1(x.y) }; }
CRY_HD v2f floor2(v2f x) { return v2f{ __builtin_floorf(x.x), __builtin_floorf(x.y) }; }
CRY_HD v2f sqrt2(v2f x) { return v2f{ __builtin_sqrtf(x.x), __builtin_sqrtf(x.y) }; }
CRY_HD v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }      // v_pk_fma_f32: two mads per instruction
CRY_HD v2f fma2(v2f a, float b, v2f c) { return fma2(a, splat(b), c); }
CRY_HD v2f fma2(float a, v2f b, v2f c) { return fma2(splat(a), b, c); }
CRY_HD v2f fma2(v2f a, v2f b, float c) { return fma2(a, b, splat(c)); }
CRY_HD v2f fma2(v2f a, float b, float c) { return fma2(a, splat(b), splat(c)); }
CRY_HD v2f lerp2(v2f a, v2f b, v2f t) { return fma2(t, b - a, a); }
// rcp() of two values: the transcendental seeds are scalar instructions, the Newton step is packed.
CRY_HD v2f rcp2(v2f b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const v2f r0{ __builtin_amdgcn_rcpf(b.x), __builtin_amdgcn_rcpf(b.y) };
    const v2f e = fma2(-b, r0, 1.0f);
    const v2f r1 = fma2(e, r0, r0);
    return v2f{ __builtin_isnormal(r0.x) ? r1.x : r0.x, __builtin_isnormal(r0.y) ? r1.y : r0.y };
#else
    return v2f{ rcp(b.x), rcp(b.y) };
#endif
}
CRY_HD v2f inv_len_from_sq2(v2f d2)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const v2f x{ clamp_len2(d2.x), clamp_len2(d2.y) };
    const v2f g{ __builtin_amdgcn_sqrtf(x.x), __builtin_amdgcn_sqrtf(x.y) };
    const v2f h = 0.5f * v2f{ __builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y) };
    const v2f sq = fma2(fma2(-g, g, x), h, g);
    const v2f r0{ __builtin_amdgcn_rcpf(sq.x), __builtin_amdgcn_rcpf(sq.y) };
    return fma2(fma2(-sq, r0, 1.0f), r0, r0);
#else
    return v2f{ inv_len_from_sq(d2.x), inv_len_from_sq(d2.y) };
#endif
}
struct f3x2 { v2f x, y, z; };                                          // two 3-vectors, component-packed
CRY_HD v2f dot3x2(f3x2 a, f3x2 b) { return fma2(a.z, b.z, fma2(a.y, b.y, a.x * b.x)); }
CRY_HD f3x2 splat3(f3 a) { return f3x2{ splat(a.x), splat(a.y), splat(a.z) }; }

// ---- format decoders -------------------------------------------------------------------------------
// u / (2^n - 1), correctly rounded, for every u the format can hold: with c = 1 / (2^n - 1) split into c_hi = RN(c) and
// c_lo = RN(c - c_hi), fma(u, c_hi, u * c_lo) equals the IEEE quotient (float)u / (2^n - 1) -- checked exhaustively on the
// host for n = 8, 16 and 24 (tests/test_devmath_host.py).  Three instructions with the conversion (v_cvt_f32_u32, v_mul_f32,
// v_fma_f32) against eleven for the division.
CRY_HD float unorm_decode(uint32_t u, float c_hi, float c_lo)
{
    const float a = (float)u;
    return fma(a, c_hi, a * c_lo);
}
CRY_HD float d24_to_float(uint32_t texel) { return unorm_decode(texel & 0x00FFFFFFu, u2f(0x33800001u), u2f(0xA77FFFFFu)); }
CRY_HD float unorm16_to_float(uint32_t u) { return unorm_decode(u, u2f(0x37800080u), u2f(0x27800080u)); }
CRY_HD float unorm8_to_float(uint32_t u) { return unorm_decode(u, u2f(0x3B808081u), u2f(0xAF7EFEFFu)); }
CRY_HD uint32_t float_to_unorm16(float x) { return (uint32_t)fma(saturate(x), 65535.0f, 0.5f); }
CRY_HD uint32_t float_to_unorm8(float x) { return (uint32_t)fma(saturate(x), 255.0f, 0.5f); }

CRY_HD float half_to_float(uint16_t h)
{
    _Float16 v;
    __builtin_memcpy(&v, &h, 2);
    return (float)v;
}

// ---- deterministic transcendentals -------------------------------------------------------------------
// sin/cos: 3-term Cody-Waite reduction by pi/2, degree-7 / degree-8 kernels on [-pi/4, pi/4], every step one mad.
CRY_HD float det_sincos(float x, int want_cos)
{
    float ax = __builtin_fabsf(x);
    if (!(ax < 8388608.0f)) return x - x;
    float k = __builtin_rintf(x * 0.636619772367581343f);
    float r = fma(-k, 1.5703125f, x);
    r = fma(-k, 4.837512969970703125e-4f, r);
    r = fma(-k, 7.54978995489188216e-8f, r);
    int q = ((int)k + want_cos) & 3;
    float r2 = r * r;
    float sp = fma(fma(-1.9515295891e-4f, r2, 8.3321608736e-3f), r2, -1.6666654611e-1f);
    float s = fma(sp * r2, r, r);
    float cp = fma(fma(2.443315711809948e-5f, r2, -1.388731625493765e-3f), r2, 4.166664568298827e-2f);
    float c = fma(cp * r2, r2, fma(-0.5f, r2, 1.0f));
    float v = (q & 1) ? c : s;
    return (q & 2) ? -v : v;
}
CRY_HD float det_sin(float x) { return det_sincos(x, 0); }
CRY_HD float det_cos(float x) { return det_sincos(x, 1); }

// log2 of a normal positive x: x = m 2^e, m in [sqrt(1/2), sqrt(2)); log2(m) = s P(s^2), s = (m - 1) / (m + 1)
CRY_HD float det_log2_normal(float x)
{
    const uint32_t ue = f2u(x) - 0x3F3504F3u;
    const float ef = (float)((int32_t)ue >> 23);
    const float m = u2f((ue & 0x007FFFFFu) + 0x3F3504F3u);
    const float s = (m - 1.0f) * rcp_normal(m + 1.0f);
    const float s2 = s * s;
    const float p = fma(fma(fma(0.43174004554748535f, s2, 0.5767142176628113f), s2, 0.9617988467216492f), s2, 2.885390043258667f);
    return fma(s, p, ef);
}
// 2^z for z in [-125, 127]
CRY_HD float det_exp2_clamped(float z)
{
    const float n = __builtin_rintf(z);
    const float f = z - n;
    const float p = fma(fma(fma(fma(fma(fma(0.00015406982856802642f, f, 0.0013400138122960925f), f, 0.009618260897696018f), f,
                                    0.05550328269600868f), f, 0.24022649228572845f), f, 0.6931471824645996f), f, 1.0f);
    return __builtin_ldexpf(p, (int)n);
}
CRY_HD float clampf(float x, float lo, float hi) { return __builtin_fminf(__builtin_fmaxf(x, lo), hi); }
CRY_HD float det_log2(float x)
{
    if (!(x >= 1.17549435e-38f)) return (x >= 0.0f) ? u2f(0xFF800000u) : u2f(0x7FC00000u);
    return det_log2_normal(x);
}
CRY_HD float det_exp2(float x)
{
    if (x != x) return x;
    return det_exp2_clamped(clampf(x, -125.0f, 127.0f));
}
// HLSL pow(x, y) for 0 < y <= 1: below the smallest normal -> 0; negative or NaN -> NaN.  Branch-free:
// the kernel runs on a safe stand-in and the specials are selected afterwards.
CRY_HD float det_pow(float x, float y)
{
    const bool normal = x >= 1.17549435e-38f;
    const float z = y * det_log2_normal(normal ? x : 1.0f);
    const float r = det_exp2_clamped(clampf(z, -125.0f, 127.0f));
    return normal ? r : ((x >= 0.0f) ? 0.0f : u2f(0x7FC00000u));
}

CRY_HD v2f rcp_normal2(v2f b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const v2f r0{ __builtin_amdgcn_rcpf(b.x), __builtin_amdgcn_rcpf(b.y) };
    return fma2(fma2(-b, r0, 1.0f), r0, r0);
#else
    return v2f{ rcp_normal(b.x), rcp_normal(b.y) };
#endif
}

// ---- the tone map's pow(x, 1 / 2.2)  (DeferredShading.hlsl:89-90) -----------------------------------------------------------
// Oracle definition, version 3: with x = m 2^(E-127), m in [1, 2) read off the bit pattern,
//     pow(x, 1/2.2) := SCALE[E] * P(m - 1),
// SCALE[E] = RN(2^((E-127)/2.2)) from a 256-entry table indexed by the biased exponent field, P the degree-7 minimax
// polynomial of (1 + u)^(1/2.2) on [0, 1) in Horner form, one mad per step (constants: gamma_pow.inc, generated by
// tools/gen_gamma_pow.py; <= 3e-7 relative, tests/test_oracle_kat.py).  SCALE[0] = 0: zero and subnormal bases of either sign
// give 0 (D3D flushes denormals); SCALE[255] = inf: pow(+inf) = inf; a negative normal base or NaN gives NaN.
// Round 1-3 evaluated exp2(y * log2 x) with two range reductions, a reciprocal and nine mads per value (det_pow above, still the
// definition of the general pow): 35 instructions per channel against 15 + one 4-byte load from a 1 KB table here.
#include "gamma_pow.inc"
static constexpr float kGammaPowCoef[8] = CRY_GAMMA_POW_COEFFS;
static constexpr float kGammaPowScale[256] = CRY_GAMMA_POW_SCALE;
CRY_HD float pow_inv_gamma(float x)
{
    const uint32_t b = f2u(x);
    const float scale = kGammaPowScale[(b >> 23) & 255u];
    const float u = u2f((b & 0x007FFFFFu) | 0x3F800000u) - 1.0f;       // exact
    float p = kGammaPowCoef[7];
#pragma unroll
    for (int k = 6; k >= 0; --k) p = fma(p, u, kGammaPowCoef[k]);
    const float r = scale * p;
    return (x > -1.17549435e-38f) ? r : u2f(0x7FC00000u);              // false for negative normals and NaN
}
// two values at once (the red and green channel): bit manipulation and table lookups per lane, the polynomial packed; each lane
// is bit-identical to pow_inv_gamma
CRY_HD v2f pow_inv_gamma2(v2f x)
{
    const uint32_t b0 = f2u(x.x), b1 = f2u(x.y);
    const v2f scale{ kGammaPowScale[(b0 >> 23) & 255u], kGammaPowScale[(b1 >> 23) & 255u] };
    const v2f u = v2f{ u2f((b0 & 0x007FFFFFu) | 0x3F800000u), u2f((b1 & 0x007FFFFFu) | 0x3F800000u) } - 1.0f;
    v2f p = splat(kGammaPowCoef[7]);
#pragma unroll
    for (int k = 6; k >= 0; --k) p = fma2(p, u, splat(kGammaPowCoef[k]));
    const v2f r = scale * p;
    return v2f{ (x.x > -1.17549435e-38f) ? r.x : u2f(0x7FC00000u), (x.y > -1.17549435e-38f) ? r.y : u2f(0x7FC00000u) };
}

// ---- bilinear addressing --------------------------------------------------------------
struct Bilin { int i0, j0; float fx, fy; };

CRY_HD int texel_index(float fl, uint32_t dim)
{
    // clamp(fl, -2, dim) with NaN -> -2 (fmax returns the non-NaN operand); fl is already integral.  A footprint whose
    // top-left texel lies further out than that consists of out-of-range texels only, and so does its clamped stand-in.
#if defined(__HIP_DEVICE_COMPILE__)
    return (int)__builtin_amdgcn_fmed3f(fl, -2.0f, (float)dim);      // one v_med3_f32; a quiet NaN comes out as the minimum, like clamp_len2
#else
    const float c = __builtin_fminf(__builtin_fmaxf(fl, -2.0f), (float)dim);
    return (int)c;
#endif
}
// FINITE: a promise that u and v are finite and small enough for tx, ty to be finite (the caller bounded what they come from):
// the non-finite case below cannot occur and its tests are dropped.
template <bool FINITE = false>
CRY_HD Bilin bilinear_setup(float u, float v, uint32_t w, uint32_t h)
{
    Bilin b;
    const float tx = fma(u, (float)w, -0.5f);
    const float ty = fma(v, (float)h, -0.5f);
    const float flx = __builtin_floorf(tx), fly = __builtin_floorf(ty);
    const float fx = tx - flx, fy = ty - fly;
    const bool bad = !FINITE && (!(fx == fx) | !(fy == fy));   // non-finite coordinates address only out-of-range texels
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
