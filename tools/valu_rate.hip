// valu_rate.hip -- issue-rate microbenchmark for the VALU instructions the hot-path kernels are made of (gfx950).
// Every lane runs N unrolled copies of one instruction over 8 independent register chains; 8 waves per SIMD.
// Prints SIMD cycles per wave64 instruction (wall time x clock / instructions per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));
constexpr int ITER = 4096;

#define KERNEL(NAME, DECL, BODY, SINK)                                                     \
    __global__ __launch_bounds__(256) void NAME(float* out, float a, float b) {            \
        DECL;                                                                               \
        for (int i = 0; i < ITER; ++i) { BODY; }                                           \
        out[blockIdx.x * 256 + threadIdx.x] = SINK;                                        \
    }

#define R8 float r0 = a + threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7
#define S8 (r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7)
#define OP8(ASM) asm volatile(ASM "\n" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b))

KERNEL(k_mul, R8, OP8("v_mul_f32 %0, %0, %8\nv_mul_f32 %1, %1, %8\nv_mul_f32 %2, %2, %8\nv_mul_f32 %3, %3, %8\nv_mul_f32 %4, %4, %8\nv_mul_f32 %5, %5, %8\nv_mul_f32 %6, %6, %8\nv_mul_f32 %7, %7, %8"), S8)
KERNEL(k_add, R8, OP8("v_add_f32 %0, %0, %8\nv_add_f32 %1, %1, %8\nv_add_f32 %2, %2, %8\nv_add_f32 %3, %3, %8\nv_add_f32 %4, %4, %8\nv_add_f32 %5, %5, %8\nv_add_f32 %6, %6, %8\nv_add_f32 %7, %7, %8"), S8)
KERNEL(k_fma, R8, OP8("v_fma_f32 %0, %0, %8, %8\nv_fma_f32 %1, %1, %8, %8\nv_fma_f32 %2, %2, %8, %8\nv_fma_f32 %3, %3, %8, %8\nv_fma_f32 %4, %4, %8, %8\nv_fma_f32 %5, %5, %8, %8\nv_fma_f32 %6, %6, %8, %8\nv_fma_f32 %7, %7, %8, %8"), S8)
KERNEL(k_rcp, R8, OP8("v_rcp_f32 %0, %0\nv_rcp_f32 %1, %1\nv_rcp_f32 %2, %2\nv_rcp_f32 %3, %3\nv_rcp_f32 %4, %4\nv_rcp_f32 %5, %5\nv_rcp_f32 %6, %6\nv_rcp_f32 %7, %7"), S8)
KERNEL(k_sqrt, R8, OP8("v_sqrt_f32 %0, %0\nv_sqrt_f32 %1, %1\nv_sqrt_f32 %2, %2\nv_sqrt_f32 %3, %3\nv_sqrt_f32 %4, %4\nv_sqrt_f32 %5, %5\nv_sqrt_f32 %6, %6\nv_sqrt_f32 %7, %7"), S8)
KERNEL(k_cndmask, R8, OP8("v_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\nv_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\nv_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc"), S8)
KERNEL(k_divfixup, R8, OP8("v_div_fixup_f32 %0, %0, %8, %8\nv_div_fixup_f32 %1, %1, %8, %8\nv_div_fixup_f32 %2, %2, %8, %8\nv_div_fixup_f32 %3, %3, %8, %8\nv_div_fixup_f32 %4, %4, %8, %8\nv_div_fixup_f32 %5, %5, %8, %8\nv_div_fixup_f32 %6, %6, %8, %8\nv_div_fixup_f32 %7, %7, %8, %8"), S8)
KERNEL(k_divscale, R8, OP8("v_div_scale_f32 %0, vcc, %0, %8, %0\nv_div_scale_f32 %1, vcc, %1, %8, %1\nv_div_scale_f32 %2, vcc, %2, %8, %2\nv_div_scale_f32 %3, vcc, %3, %8, %3\nv_div_scale_f32 %4, vcc, %4, %8, %4\nv_div_scale_f32 %5, vcc, %5, %8, %5\nv_div_scale_f32 %6, vcc, %6, %8, %6\nv_div_scale_f32 %7, vcc, %7, %8, %7"), S8)
KERNEL(k_mulu24, R8, OP8("v_mul_u32_u24 %0, %0, %8\nv_mul_u32_u24 %1, %1, %8\nv_mul_u32_u24 %2, %2, %8\nv_mul_u32_u24 %3, %3, %8\nv_mul_u32_u24 %4, %4, %8\nv_mul_u32_u24 %5, %5, %8\nv_mul_u32_u24 %6, %6, %8\nv_mul_u32_u24 %7, %7, %8"), S8)
KERNEL(k_mullo, R8, OP8("v_mul_lo_u32 %0, %0, %8\nv_mul_lo_u32 %1, %1, %8\nv_mul_lo_u32 %2, %2, %8\nv_mul_lo_u32 %3, %3, %8\nv_mul_lo_u32 %4, %4, %8\nv_mul_lo_u32 %5, %5, %8\nv_mul_lo_u32 %6, %6, %8\nv_mul_lo_u32 %7, %7, %8"), S8)
KERNEL(k_cvt, R8, OP8("v_cvt_f32_u32 %0, %0\nv_cvt_f32_u32 %1, %1\nv_cvt_f32_u32 %2, %2\nv_cvt_f32_u32 %3, %3\nv_cvt_f32_u32 %4, %4\nv_cvt_f32_u32 %5, %5\nv_cvt_f32_u32 %6, %6\nv_cvt_f32_u32 %7, %7"), S8)
KERNEL(k_floor, R8, OP8("v_floor_f32 %0, %0\nv_floor_f32 %1, %1\nv_floor_f32 %2, %2\nv_floor_f32 %3, %3\nv_floor_f32 %4, %4\nv_floor_f32 %5, %5\nv_floor_f32 %6, %6\nv_floor_f32 %7, %7"), S8)

KERNEL(k_fmac, R8, OP8("v_fmac_f32 %0, %8, %8\nv_fmac_f32 %1, %8, %8\nv_fmac_f32 %2, %8, %8\nv_fmac_f32 %3, %8, %8\nv_fmac_f32 %4, %8, %8\nv_fmac_f32 %5, %8, %8\nv_fmac_f32 %6, %8, %8\nv_fmac_f32 %7, %8, %8"), S8)
KERNEL(k_mul64, R8, OP8("v_mul_f32_e64 %0, %0, %8\nv_mul_f32_e64 %1, %1, %8\nv_mul_f32_e64 %2, %2, %8\nv_mul_f32_e64 %3, %3, %8\nv_mul_f32_e64 %4, %4, %8\nv_mul_f32_e64 %5, %5, %8\nv_mul_f32_e64 %6, %6, %8\nv_mul_f32_e64 %7, %7, %8"), S8)
KERNEL(k_max, R8, OP8("v_max_f32 %0, %0, %8\nv_max_f32 %1, %1, %8\nv_max_f32 %2, %2, %8\nv_max_f32 %3, %3, %8\nv_max_f32 %4, %4, %8\nv_max_f32 %5, %5, %8\nv_max_f32 %6, %6, %8\nv_max_f32 %7, %7, %8"), S8)
KERNEL(k_and, R8, OP8("v_and_b32 %0, %0, %8\nv_and_b32 %1, %1, %8\nv_and_b32 %2, %2, %8\nv_and_b32 %3, %3, %8\nv_and_b32 %4, %4, %8\nv_and_b32 %5, %5, %8\nv_and_b32 %6, %6, %8\nv_and_b32 %7, %7, %8"), S8)
KERNEL(k_addu, R8, OP8("v_add_u32 %0, %0, %8\nv_add_u32 %1, %1, %8\nv_add_u32 %2, %2, %8\nv_add_u32 %3, %3, %8\nv_add_u32 %4, %4, %8\nv_add_u32 %5, %5, %8\nv_add_u32 %6, %6, %8\nv_add_u32 %7, %7, %8"), S8)
KERNEL(k_lshl, R8, OP8("v_lshlrev_b32 %0, 1, %0\nv_lshlrev_b32 %1, 1, %1\nv_lshlrev_b32 %2, 1, %2\nv_lshlrev_b32 %3, 1, %3\nv_lshlrev_b32 %4, 1, %4\nv_lshlrev_b32 %5, 1, %5\nv_lshlrev_b32 %6, 1, %6\nv_lshlrev_b32 %7, 1, %7"), S8)
KERNEL(k_cmp, R8, OP8("v_cmp_gt_f32 vcc, %0, %8\nv_cmp_gt_f32 vcc, %1, %8\nv_cmp_gt_f32 vcc, %2, %8\nv_cmp_gt_f32 vcc, %3, %8\nv_cmp_gt_f32 vcc, %4, %8\nv_cmp_gt_f32 vcc, %5, %8\nv_cmp_gt_f32 vcc, %6, %8\nv_cmp_gt_f32 vcc, %7, %8"), S8)
KERNEL(k_cmp64, R8, OP8("v_cmp_gt_f32_e64 s[20:21], %0, %8\nv_cmp_gt_f32_e64 s[20:21], %1, %8\nv_cmp_gt_f32_e64 s[20:21], %2, %8\nv_cmp_gt_f32_e64 s[20:21], %3, %8\nv_cmp_gt_f32_e64 s[20:21], %4, %8\nv_cmp_gt_f32_e64 s[20:21], %5, %8\nv_cmp_gt_f32_e64 s[20:21], %6, %8\nv_cmp_gt_f32_e64 s[20:21], %7, %8"), S8)
KERNEL(k_cmpcnd, R8, OP8("v_cmp_gt_f32 vcc, %0, %8\nv_cndmask_b32 %0, %0, %8, vcc\nv_cmp_gt_f32 vcc, %1, %8\nv_cndmask_b32 %1, %1, %8, vcc\nv_cmp_gt_f32 vcc, %2, %8\nv_cndmask_b32 %2, %2, %8, vcc\nv_cmp_gt_f32 vcc, %3, %8\nv_cndmask_b32 %3, %3, %8, vcc\nv_cmp_gt_f32 vcc, %4, %8\nv_cndmask_b32 %4, %4, %8, vcc\nv_cmp_gt_f32 vcc, %5, %8\nv_cndmask_b32 %5, %5, %8, vcc\nv_cmp_gt_f32 vcc, %6, %8\nv_cndmask_b32 %6, %6, %8, vcc\nv_cmp_gt_f32 vcc, %7, %8\nv_cndmask_b32 %7, %7, %8, vcc"), S8)
KERNEL(k_cnd64, R8, OP8("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\nv_cndmask_b32_e64 %1, %1, %8, s[20:21]\nv_cndmask_b32_e64 %2, %2, %8, s[20:21]\nv_cndmask_b32_e64 %3, %3, %8, s[20:21]\nv_cndmask_b32_e64 %4, %4, %8, s[20:21]\nv_cndmask_b32_e64 %5, %5, %8, s[20:21]\nv_cndmask_b32_e64 %6, %6, %8, s[20:21]\nv_cndmask_b32_e64 %7, %7, %8, s[20:21]"), S8)
KERNEL(k_med3, R8, OP8("v_med3_f32 %0, %0, %8, %8\nv_med3_f32 %1, %1, %8, %8\nv_med3_f32 %2, %2, %8, %8\nv_med3_f32 %3, %3, %8, %8\nv_med3_f32 %4, %4, %8, %8\nv_med3_f32 %5, %5, %8, %8\nv_med3_f32 %6, %6, %8, %8\nv_med3_f32 %7, %7, %8, %8"), S8)
KERNEL(k_cvti, R8, OP8("v_cvt_i32_f32 %0, %0\nv_cvt_i32_f32 %1, %1\nv_cvt_i32_f32 %2, %2\nv_cvt_i32_f32 %3, %3\nv_cvt_i32_f32 %4, %4\nv_cvt_i32_f32 %5, %5\nv_cvt_i32_f32 %6, %6\nv_cvt_i32_f32 %7, %7"), S8)
KERNEL(k_frexp, R8, OP8("v_frexp_mant_f32 %0, %0\nv_frexp_mant_f32 %1, %1\nv_frexp_mant_f32 %2, %2\nv_frexp_mant_f32 %3, %3\nv_frexp_mant_f32 %4, %4\nv_frexp_mant_f32 %5, %5\nv_frexp_mant_f32 %6, %6\nv_frexp_mant_f32 %7, %7"), S8)
KERNEL(k_ldexp, R8, OP8("v_ldexp_f32 %0, %0, 1\nv_ldexp_f32 %1, %1, 1\nv_ldexp_f32 %2, %2, 1\nv_ldexp_f32 %3, %3, 1\nv_ldexp_f32 %4, %4, 1\nv_ldexp_f32 %5, %5, 1\nv_ldexp_f32 %6, %6, 1\nv_ldexp_f32 %7, %7, 1"), S8)
KERNEL(k_mov, R8, OP8("v_mov_b32 %0, %8\nv_mov_b32 %1, %8\nv_mov_b32 %2, %8\nv_mov_b32 %3, %8\nv_mov_b32 %4, %8\nv_mov_b32 %5, %8\nv_mov_b32 %6, %8\nv_mov_b32 %7, %8"), S8)

#define P4 v2f p0 = { a + threadIdx.x, a }, p1 = p0 + 1.0f, p2 = p0 + 2.0f, p3 = p0 + 3.0f, p4 = p0 + 4.0f, p5 = p0 + 5.0f, p6 = p0 + 6.0f, p7 = p0 + 7.0f; v2f pb = { b, b }
#define PS ((p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7).x)
#define POP8(ASM) asm volatile(ASM "\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb))
KERNEL(k_pkmul, P4, POP8("v_pk_mul_f32 %0, %0, %8\nv_pk_mul_f32 %1, %1, %8\nv_pk_mul_f32 %2, %2, %8\nv_pk_mul_f32 %3, %3, %8\nv_pk_mul_f32 %4, %4, %8\nv_pk_mul_f32 %5, %5, %8\nv_pk_mul_f32 %6, %6, %8\nv_pk_mul_f32 %7, %7, %8"), PS)
KERNEL(k_pkadd, P4, POP8("v_pk_add_f32 %0, %0, %8\nv_pk_add_f32 %1, %1, %8\nv_pk_add_f32 %2, %2, %8\nv_pk_add_f32 %3, %3, %8\nv_pk_add_f32 %4, %4, %8\nv_pk_add_f32 %5, %5, %8\nv_pk_add_f32 %6, %6, %8\nv_pk_add_f32 %7, %7, %8"), PS)
KERNEL(k_pkfma, P4, POP8("v_pk_fma_f32 %0, %0, %8, %8\nv_pk_fma_f32 %1, %1, %8, %8\nv_pk_fma_f32 %2, %2, %8, %8\nv_pk_fma_f32 %3, %3, %8, %8\nv_pk_fma_f32 %4, %4, %8, %8\nv_pk_fma_f32 %5, %5, %8, %8\nv_pk_fma_f32 %6, %6, %8, %8\nv_pk_fma_f32 %7, %7, %8, %8"), PS)

typedef void (*kern_t)(float*, float, float);
int main()
{
    float* out;
    const int blocks = 256 * 8;   // 8 workgroups of 4 waves per CU -> 8 waves per SIMD
    CHK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const double clk = prop.clockRate * 1e3;   // Hz
    struct { const char* name; kern_t k; } ks[] = {
        { "v_mul_f32", k_mul }, { "v_add_f32", k_add }, { "v_fma_f32", k_fma }, { "v_pk_mul_f32", k_pkmul }, { "v_pk_add_f32", k_pkadd },
        { "v_pk_fma_f32", k_pkfma }, { "v_rcp_f32", k_rcp }, { "v_sqrt_f32", k_sqrt }, { "v_cndmask_b32", k_cndmask },
        { "v_div_scale_f32", k_divscale }, { "v_div_fixup_f32", k_divfixup }, { "v_mul_u32_u24", k_mulu24 }, { "v_mul_lo_u32", k_mullo },
        { "v_cvt_f32_u32", k_cvt }, { "v_floor_f32", k_floor },
        { "v_fmac_f32 (VOP2)", k_fmac }, { "v_mul_f32_e64", k_mul64 }, { "v_max_f32", k_max }, { "v_and_b32", k_and }, { "v_add_u32", k_addu },
        { "v_lshlrev_b32", k_lshl }, { "v_cmp_gt_f32 vcc", k_cmp }, { "v_cmp_gt_f32_e64 sgpr", k_cmp64 }, { "v_cmp+v_cndmask (x2)", k_cmpcnd },
        { "v_cndmask_e64 sgpr", k_cnd64 }, { "v_med3_f32", k_med3 }, { "v_cvt_i32_f32", k_cvti }, { "v_frexp_mant_f32", k_frexp },
        { "v_ldexp_f32", k_ldexp }, { "v_mov_b32", k_mov },
    };
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    printf("clock %.0f MHz (nominal); cycles per wave64 instruction per SIMD at 8 waves/SIMD\n", clk / 1e6);
    for (auto& k : ks) {
        hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.9999f);
        CHK(hipDeviceSynchronize());
        CHK(hipEventRecord(e0));
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k.k, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.9999f);
        CHK(hipEventRecord(e1));
        CHK(hipEventSynchronize(e1));
        float ms;
        CHK(hipEventElapsedTime(&ms, e0, e1));
        const double instr_per_simd = 5.0 * (double)blocks * 4.0 /*waves*/ * ITER * 8.0 / (256.0 * 4.0);
        printf("%-18s %6.2f cycles/instr   (%.3f ms)\n", k.name, ms * 1e-3 * clk / instr_per_simd, ms / 5);
    }
    return 0;
}
