// exact_math_probe.hip -- exhaustive check (every positive normal binary32) of the refinement sequences that turn the
// gfx950 approximate v_rcp_f32 / v_rsq_f32 / v_sqrt_f32 into exactly specified functions the CPU oracle can restate:
//   rcp(x)   := 1.0f / x                       (IEEE correctly rounded)
//   sqrt(x)  := sqrtf(x)                       (IEEE correctly rounded)
//   rsqrt(x) := (float)(1.0 / sqrt((double)x)) (the oracle's definition; fp64 here, IEEE on the GPU as well)
// The reference value of each is computed on the GPU by the compiler's IEEE expansions (fp32 division / sqrt with
// -fhip-fp32-correctly-rounded-divide-sqrt, the default; fp64 for rsqrt).  Prints, per candidate, the number of inputs
// whose bits differ and the first few of them.  Build: hipcc -O2 --offload-arch=gfx950 -ffp-contract=off
#include <hip/hip_runtime.h>
#include "../crychic_renderer_amd/csrc/devmath.hpp"
#include <cstdint>
#include <cstdio>
#include <cstring>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }

// ---- candidates -------------------------------------------------------------------------------------------------
__device__ __forceinline__ float rcp_hw(float b) { return __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float rcp_nr1(float b)
{
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r0, 1.0f);
    return __builtin_fmaf(e, r0, r0);
}
__device__ __forceinline__ float rcp_nr2(float b)
{
    const float r1 = rcp_nr1(b);
    const float e = __builtin_fmaf(-b, r1, 1.0f);
    return __builtin_fmaf(e, r1, r1);
}
__device__ __forceinline__ float rsq_hw(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float rsq_nr1(float x)      // residual from the rounded product
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    const float r = __builtin_fmaf(-g, h, 0.5f);
    return __builtin_fmaf(y, r, y);
}
__device__ __forceinline__ float rsq_nr1x(float x)     // residual with the product's rounding error folded in
{
    const float y = __builtin_amdgcn_rsqf(x);
    const float g = x * y, h = 0.5f * y;
    const float ge = __builtin_fmaf(x, y, -g);
    float r = __builtin_fmaf(-g, h, 0.5f);
    r = __builtin_fmaf(-ge, h, r);
    return __builtin_fmaf(y, r, y);
}
__device__ __forceinline__ float rsq_nr2x(float x)     // the same step twice
{
    float y = rsq_nr1x(x);
    const float g = x * y, h = 0.5f * y;
    const float ge = __builtin_fmaf(x, y, -g);
    float r = __builtin_fmaf(-g, h, 0.5f);
    r = __builtin_fmaf(-ge, h, r);
    return __builtin_fmaf(y, r, y);
}
__device__ __forceinline__ float sqrt_hw(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float sqrt_gs(float x)      // Goldschmidt step + Markstein correction, no range scaling
{
    const float y = __builtin_amdgcn_rsqf(x);
    float g = x * y, h = 0.5f * y;
    const float r = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r, g);
    h = __builtin_fmaf(h, r, h);
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}
__device__ __forceinline__ float sqrt_short(float x)   // hardware sqrt + one Markstein correction
{
    const float g = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rsqf(x);
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}

// ---- total (branch-free) forms: specials come from the hardware seed, selected by its class ---------------------
__device__ __forceinline__ bool is_normal(float x) { return __builtin_isnormal(x); }
__device__ __forceinline__ float rcp_total(float b)
{
    const float r0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, r0, 1.0f);
    const float r1 = __builtin_fmaf(e, r0, r0);
    return is_normal(r0) ? r1 : r0;
}
__device__ __forceinline__ float sqrt_total(float x)
{
    const float g = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rsqf(x);
    const float d = __builtin_fmaf(-g, g, x);
    const float s = __builtin_fmaf(d, h, g);
    return is_normal(g) ? s : g;
}
__device__ __forceinline__ float invsqrt_total(float x)     // rcp(sqrt(x)) with one select
{
    const float g = __builtin_amdgcn_sqrtf(x);
    const float y = __builtin_amdgcn_rsqf(x);
    const float h = 0.5f * y;
    const float d = __builtin_fmaf(-g, g, x);
    const float s = __builtin_fmaf(d, h, g);
    const float r0 = __builtin_amdgcn_rcpf(s);
    const float e = __builtin_fmaf(-s, r0, 1.0f);
    const float r1 = __builtin_fmaf(e, r0, r0);
    return (is_normal(g) && is_normal(r0)) ? r1 : y;
}
// definitions the CPU oracle can state with IEEE operations only (flush-to-zero on subnormal inputs and results)
__device__ __forceinline__ float def_rcp(float b)
{
    if (b != b) return b;
    const float ab = __builtin_fabsf(b);
    if (ab < 1.17549435e-38f) return __builtin_copysignf(__builtin_inff(), b);
    if (ab > 8.50705917e37f) return __builtin_copysignf(0.0f, b);       // |1/b| < 2^-126
    return 1.0f / b;
}
__device__ __forceinline__ float def_sqrt(float x)
{
    if (x != x) return x;
    if (__builtin_fabsf(x) < 1.17549435e-38f) return __builtin_copysignf(0.0f, x);
    return __builtin_sqrtf(x);
}
__device__ __forceinline__ float def_invsqrt(float x)
{
    if (x != x) return x;
    if (__builtin_fabsf(x) < 1.17549435e-38f) return __builtin_copysignf(__builtin_inff(), x);
    return def_rcp(__builtin_sqrtf(x));
}

// the oracle's definitions of the two length primitives (or_math.h), IEEE operations only
__device__ __forceinline__ float def_clamp_len2(float d) { return (d != d) ? 7.8886090522101181e-31f : __builtin_fminf(__builtin_fmaxf(d, 7.8886090522101181e-31f), 1.2676506002282294e30f); }
__device__ __forceinline__ float def_len(float d) { return __builtin_sqrtf(def_clamp_len2(d)); }
__device__ __forceinline__ float def_inv_len(float d) { return 1.0f / __builtin_sqrtf(def_clamp_len2(d)); }

// ---- references -------------------------------------------------------------------------------------------------
__device__ __forceinline__ float ref_rcp(float x) { return 1.0f / x; }
__device__ __forceinline__ float ref_sqrt(float x) { return __builtin_sqrtf(x); }
__device__ __forceinline__ float ref_rsqrt(float x) { return (float)(1.0 / __builtin_sqrt((double)x)); }

struct Tally { unsigned long long bad; unsigned long long maxulp; uint32_t first[8]; };

template <int WHICH>
__global__ __launch_bounds__(256) void sweep(Tally* t, uint64_t lo, uint64_t hi)
{
    unsigned long long bad = 0, maxulp = 0;
    for (uint64_t u = lo + (uint64_t)blockIdx.x * 256u + threadIdx.x; u < hi; u += (uint64_t)gridDim.x * 256u) {
        float x = u2f((uint32_t)u);
        // The length primitives only ever see an arithmetic result (a dot product), and the hardware never produces a signaling
        // NaN: v_med3_f32 treats the two NaN kinds differently, so the sweep quiets them as any preceding mad would.
        if (WHICH >= 14 && x != x) x = u2f(f2u(x) | 0x00400000u);
        float got, ref;
        if (WHICH == 0) { got = rcp_hw(x); ref = ref_rcp(x); }
        else if (WHICH == 1) { got = rcp_nr1(x); ref = ref_rcp(x); }
        else if (WHICH == 2) { got = rcp_nr2(x); ref = ref_rcp(x); }
        else if (WHICH == 3) { got = rsq_hw(x); ref = ref_rsqrt(x); }
        else if (WHICH == 4) { got = rsq_nr1(x); ref = ref_rsqrt(x); }
        else if (WHICH == 5) { got = rsq_nr1x(x); ref = ref_rsqrt(x); }
        else if (WHICH == 6) { got = rsq_nr2x(x); ref = ref_rsqrt(x); }
        else if (WHICH == 7) { got = sqrt_hw(x); ref = ref_sqrt(x); }
        else if (WHICH == 8) { got = sqrt_gs(x); ref = ref_sqrt(x); }
        else if (WHICH == 9) { got = sqrt_short(x); ref = ref_sqrt(x); }
        else if (WHICH == 10) { got = rcp_total(x); ref = def_rcp(x); }
        else if (WHICH == 11) { got = sqrt_total(x); ref = def_sqrt(x); }
        else if (WHICH == 12) { got = invsqrt_total(x); ref = def_invsqrt(x); }
        else if (WHICH == 13) { got = cry::rcp(x); ref = def_rcp(x); }                    // the shipped text (csrc/devmath.hpp)
        else if (WHICH == 14) { got = cry::len_from_sq(x); ref = def_len(x); }
        else { got = cry::inv_len_from_sq(x); ref = def_inv_len(x); }
        uint32_t a = f2u(got), b = f2u(ref);
        if (got != got) a = 0x7FC00000u;       // any NaN == any NaN
        if (ref != ref) b = 0x7FC00000u;
        if (a != b) {
            const unsigned long long d = a > b ? a - b : b - a;
            if (d > maxulp) maxulp = d;
            if (bad == 0) { const unsigned long long k = atomicAdd(&t->bad, 0ull); if (k < 8) t->first[k & 7] = (uint32_t)u; }
            ++bad;
        }
    }
    if (bad) { atomicAdd(&t->bad, bad); atomicMax(&t->maxulp, maxulp); }
}

template <int WHICH>
static int run(const char* name, uint64_t lo, uint64_t hi, Tally* d)
{
    Tally z;
    memset(&z, 0, sizeof z);
    CHK(hipMemcpy(d, &z, sizeof z, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(sweep<WHICH>, dim3(4096), dim3(256), 0, 0, d, lo, hi);
    CHK(hipDeviceSynchronize());
    CHK(hipMemcpy(&z, d, sizeof z, hipMemcpyDeviceToHost));
    printf("%-34s inputs [%08llx, %08llx): %llu differ, max distance %llu ulp", name, (unsigned long long)lo, (unsigned long long)hi, z.bad, z.maxulp);
    if (z.bad) { printf("; e.g."); for (int i = 0; i < 8 && z.first[i]; ++i) printf(" %08x", z.first[i]); }
    printf("\n");
    return 0;
}

int main(int argc, char** argv)
{
    Tally* d;
    CHK(hipMalloc(&d, sizeof(Tally)));
    const uint32_t cls8[][2] = { { 0x00000000u, 0x00800000u }, { 0x00800000u, 0x7F000000u }, { 0x7F000000u, 0x7F800000u }, { 0x7F800000u, 0x80000000u },
                                 { 0x80000000u, 0x80800000u }, { 0x80800000u, 0xFF000000u }, { 0xFF000000u, 0xFF800000u }, { 0xFF800000u, 0xFFFFFFFFu } };
    if (argc > 1 && strcmp(argv[1], "shipped") == 0) {     // what tests/test_gpu_exact_math.py asserts on
        for (auto& c : cls8) {
            if (run<13>("rcp (devmath.hpp) vs or_rcp", c[0], c[1], d)) return 1;
            if (run<14>("len_from_sq vs or_len", c[0], c[1], d)) return 1;
            if (run<15>("inv_len_from_sq vs or_inv_len", c[0], c[1], d)) return 1;
        }
        // 0xFFFFFFFF itself (a NaN): the half-open ranges above stop just short of it
        if (run<13>("rcp (devmath.hpp) vs or_rcp", 0xFFFFFFFFull, 0x100000000ull, d)) return 1;
        CHK(hipFree(d));
        return 0;
    }
    // full positive normal range, and the range the kernels promise (2^-100 .. 2^100: neither the input nor the result is subnormal)
    const uint32_t ranges[2][2] = { { 0x00800000u, 0x7F800000u }, { 0x0D800000u, 0x71800000u } };
    for (int r = 0; r < 2; ++r) {
        const uint32_t lo = ranges[r][0], hi = ranges[r][1];
        printf("---- range %d\n", r);
        if (run<0>("v_rcp_f32", lo, hi, d)) return 1;
        if (run<1>("v_rcp_f32 + 1 Newton step", lo, hi, d)) return 1;
        if (run<2>("v_rcp_f32 + 2 Newton steps", lo, hi, d)) return 1;
        if (run<3>("v_rsq_f32", lo, hi, d)) return 1;
        if (run<4>("v_rsq_f32 + step (rounded product)", lo, hi, d)) return 1;
        if (run<5>("v_rsq_f32 + step (exact residual)", lo, hi, d)) return 1;
        if (run<6>("v_rsq_f32 + 2 steps (exact)", lo, hi, d)) return 1;
        if (run<7>("v_sqrt_f32", lo, hi, d)) return 1;
        if (run<8>("v_rsq_f32 Goldschmidt+Markstein", lo, hi, d)) return 1;
        if (run<9>("v_sqrt_f32 + Markstein", lo, hi, d)) return 1;
    }
    printf("---- total functions, every binary32 bit pattern, by class\n");
    const uint32_t cls[][2] = { { 0x00000000u, 0x00800000u }, { 0x00800000u, 0x7F000000u }, { 0x7F000000u, 0x7F800000u }, { 0x7F800000u, 0x80000000u },
                                { 0x80000000u, 0x80800000u }, { 0x80800000u, 0xFF000000u }, { 0xFF000000u, 0xFF800000u }, { 0xFF800000u, 0xFFFFFFFFu } };
    for (auto& c : cls) {
        if (run<10>("rcp_total vs def_rcp", c[0], c[1], d)) return 1;
        if (run<11>("sqrt_total vs def_sqrt", c[0], c[1], d)) return 1;
        if (run<12>("invsqrt_total vs def_invsqrt", c[0], c[1], d)) return 1;
    }
    CHK(hipFree(d));
    return 0;
}
