// kernels.hip -- gfx950 kernels of the CRYCHIC hot path and their stream-ordered launchers.
//
// Work decomposition (all kernels): one lane per output pixel, 64 consecutive pixels of one row per
// wavefront so every plane access is a contiguous 64-lane burst (16 B/lane on the fp32 G-buffer planes,
// 8 B/lane on the fp16 normal plane and the depth row pairs), 4 rows per 256-thread workgroup.
#include <hip/hip_runtime.h>
#include "kernels.hpp"
#include "ssao_core.hpp"
#include "light_core.hpp"

namespace cry {

// ---- SSAO ------------------------------------------------------------------------------------------------
// Shaders/Ssao.hlsl:117-199 over half-res rows [row0, row1).  EMIT_AO = false builds only the edge workspace.
template <bool EMIT_AO>
__global__ __launch_bounds__(256) void ssao_kernel(crychic_ssao_constants cb, const u2* __restrict__ normal,
                                                   const uint32_t* __restrict__ depth,
                                                   const uint32_t* __restrict__ randvec,
                                                   uint16_t* __restrict__ ambient, EdgePlane edge, uint32_t W,
                                                   uint32_t H, uint32_t row0, uint32_t row1)
{
    const uint32_t w2 = W / 2;
    const uint32_t x = blockIdx.x * 64u + (threadIdx.x & 63u);
    const uint32_t y = row0 + blockIdx.y * 4u + (threadIdx.x >> 6);
    if (x >= w2 || y >= row1) return;

    const SsaoCentre c = ssao_centre(cb, normal, depth, W, H, (int)x, (int)y);
    if (edge.nrm) {
        const uint32_t idx = y * w2 + x;
        edge.nrm[idx] = c.nrm_bits;
        edge.vz[idx] = c.vz;
        if (x == 0) edge.gcol[y] = normal[(2u * y + 1u) * W];   // texel (0, 2y+1)
        if (y == row0) edge.grow[x] = normal[2u * x + 1u];      // texel (2x+1, 0)
    }
    if (EMIT_AO) ambient[y * w2 + x] = (uint16_t)ssao_pixel(cb, c, depth, randvec, W, H, x, y);
}

// ---- bilateral blur ------------------------------------------------------------------------------------------
// Shaders/SsaoBlur.hlsl:85-146, one sweep.  A 64 x 16 output tile plus its 5-pixel apron along the sweep axis is
// staged once in LDS as pre-decoded floats (normal.xyz + linear depth as one 16-byte entry, ambient as one dword), so
// the 11-tap window of every pixel is served by one ds_read_b128 + one ds_read_b32 per tap instead of 3 global
// fetches + 4 format conversions.  Consecutive lanes read consecutive 16-byte entries (conflict-free for both
// directions: the horizontal window slides along a staged row, the vertical one hops whole rows).
template <bool HORZ>
__global__ __launch_bounds__(256) void blur_kernel(crychic_ssao_constants cb, EdgePlane edge,
                                                   const uint16_t* __restrict__ in, uint16_t* __restrict__ out,
                                                   uint32_t W, uint32_t H, uint32_t row0, uint32_t row1)
{
    constexpr int BW = 64, BH = 16, R = 5;
    constexpr int SW = HORZ ? BW + 2 * R : BW;
    constexpr int SH = HORZ ? BH : BH + 2 * R;
    __shared__ f4a s_nz[SW * SH];
    __shared__ float s_a[SW * SH];

    const int w2 = (int)(W / 2), h2 = (int)(H / 2);
    const int x0 = (int)blockIdx.x * BW, y0 = (int)row0 + (int)blockIdx.y * BH;
    const int sx0 = HORZ ? x0 - R : x0, sy0 = HORZ ? y0 : y0 - R;
    const float borderZ = ndc_to_view(cb, 1.0f);

    for (int k = (int)threadIdx.x; k < SW * SH; k += 256) {
        const int ly = k / SW, lx = k - ly * SW;
        const BlurTap t = blur_fetch(edge, in, borderZ, w2, h2, sx0 + lx, sy0 + ly);
        s_nz[k] = f4a{ t.n.x, t.n.y, t.n.z, t.z };
        s_a[k] = t.a;
    }
    __syncthreads();

    const int tx = (int)(threadIdx.x & 63u), tyb = (int)(threadIdx.x >> 6);
    const int x = x0 + tx;
#pragma unroll
    for (int j = 0; j < BH / 4; ++j) {
        const int ty = tyb + 4 * j;
        const int y = y0 + ty;
        if (x < w2 && y < (int)row1) {
            const uint32_t v = blur_pixel(&cb.BlurWeights[0][0], [&](int i) {
                const int idx = HORZ ? ty * SW + tx + i : (ty + i) * SW + tx;
                const f4a q = s_nz[idx];
                return BlurTap{ f3{ q.x, q.y, q.z }, q.w, s_a[idx] };
            });
            out[(uint32_t)y * (uint32_t)w2 + (uint32_t)x] = (uint16_t)v;
        }
    }
}

// ---- deferred lighting -----------------------------------------------------------------------------------------
// Shaders/DeferredShading.hlsl:23-101 as a full-screen pass over rows [row0, row1), masked by depth < 1.
template <bool ZERO_RADIUS>
__global__ __launch_bounds__(256) void light_kernel(LightParams P, const f4a* __restrict__ g0,
                                                    const f4a* __restrict__ g1, const f4a* __restrict__ g2,
                                                    const uint32_t* __restrict__ depth,
                                                    const uint16_t* __restrict__ ambient,
                                                    const uint32_t* __restrict__ cube, uint32_t* __restrict__ out,
                                                    f4a* __restrict__ radiance, uint32_t row0, uint32_t row1)
{
    const uint32_t x = blockIdx.x * 64u + (threadIdx.x & 63u);
    const uint32_t y = row0 + blockIdx.y * 4u + (threadIdx.x >> 6);
    if (x >= P.W || y >= row1) return;
    const uint32_t idx = y * P.W + x;
    f4 lit;
    // coverage: the reference re-rasterises the opaque items with LESS against depth cleared to 1.0
    // (CRYCHIC.cpp:248,273) -- exactly the pixels whose normal/depth pass depth is below the clear value.
    if ((depth[idx] & 0x00FFFFFFu) < 0x00FFFFFFu) {
        lit = light_pixel<ZERO_RADIUS>(P, g0[idx], g1[idx], g2[idx], ambient, cube);
    } else if (P.flags & CRYCHIC_LIGHT_SKY) {
        lit = sky_pixel(P, cube, x, y);
    } else {
        lit = f4{ 0.690196097f, 0.768627524f, 0.870588303f, 1.0f };  // Colors::LightSteelBlue, CRYCHIC.cpp:247
    }
    if (radiance) radiance[idx] = f4a{ lit.x, lit.y, lit.z, lit.w };
    out[idx] = pack_rgba8(lit);
}

// ---- launchers ---------------------------------------------------------------------------------------------------
static inline dim3 grid_for(uint32_t width, uint32_t rows, uint32_t rows_per_block = 4u)
{
    return dim3((width + 63u) / 64u, (rows + rows_per_block - 1u) / rows_per_block, 1);
}

hipError_t launch_ssao(const crychic_ssao_constants& cb, const void* normal, const uint32_t* depth,
                       const uint8_t* randvec, uint16_t* ambient, void* edge_base, uint32_t W, uint32_t H,
                       uint32_t row0, uint32_t rows, bool emit_ao, hipStream_t stream)
{
    if (rows == 0) return hipSuccess;
    EdgePlane e{ nullptr, nullptr, nullptr, nullptr };
    if (edge_base) e = edge_plane_carve(edge_base, W, H);
    const dim3 grid = grid_for(W / 2, rows);
    if (emit_ao)
        hipLaunchKernelGGL(ssao_kernel<true>, grid, dim3(256), 0, stream, cb, (const u2*)normal, depth,
                           (const uint32_t*)randvec, ambient, e, W, H, row0, row0 + rows);
    else
        hipLaunchKernelGGL(ssao_kernel<false>, grid, dim3(256), 0, stream, cb, (const u2*)normal, depth,
                           (const uint32_t*)randvec, ambient, e, W, H, row0, row0 + rows);
    return hipGetLastError();
}

hipError_t launch_blur(const crychic_ssao_constants& cb, const void* edge_base, const uint16_t* in, uint16_t* out,
                       uint32_t W, uint32_t H, bool horizontal, uint32_t row0, uint32_t rows, hipStream_t stream)
{
    if (rows == 0) return hipSuccess;
    const EdgePlane e = edge_plane_carve(const_cast<void*>(edge_base), W, H);
    const dim3 grid = grid_for(W / 2, rows, 16u);
    if (horizontal)
        hipLaunchKernelGGL(blur_kernel<true>, grid, dim3(256), 0, stream, cb, e, in, out, W, H, row0, row0 + rows);
    else
        hipLaunchKernelGGL(blur_kernel<false>, grid, dim3(256), 0, stream, cb, e, in, out, W, H, row0, row0 + rows);
    return hipGetLastError();
}

hipError_t launch_light(const LightParams& P, const float* g0, const float* g1, const float* g2,
                        const uint32_t* depth, const uint16_t* ambient, const uint8_t* cube, uint8_t* out,
                        float* radiance, uint32_t row0, uint32_t rows, hipStream_t stream)
{
    if (rows == 0) return hipSuccess;
    const dim3 grid = grid_for(P.W, rows);
    if (P.pcfSearchRadius == 0.0f)
        hipLaunchKernelGGL(light_kernel<true>, grid, dim3(256), 0, stream, P, (const f4a*)g0, (const f4a*)g1,
                           (const f4a*)g2, depth, ambient, (const uint32_t*)cube, (uint32_t*)out, (f4a*)radiance, row0,
                           row0 + rows);
    else
        hipLaunchKernelGGL(light_kernel<false>, grid, dim3(256), 0, stream, P, (const f4a*)g0, (const f4a*)g1,
                           (const f4a*)g2, depth, ambient, (const uint32_t*)cube, (uint32_t*)out, (f4a*)radiance, row0,
                           row0 + rows);
    return hipGetLastError();
}

}  // namespace cry
