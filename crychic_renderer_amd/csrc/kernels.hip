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

// XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (each with a private 4 MiB L2), so with
// the natural order every XCD touches the whole frame: the SSAO depth taps then miss L2 eight times over
// (measured: 362 MB fetched per 4K launch against 50 MB of planes).  The remap hands each XCD stripes of STRIPE
// consecutive tile rows, interleaved over the frame with period 8*STRIPE: neighbouring tiles -- which share tap
// footprints and blur aprons -- share an L2, while cheap (sky) and expensive (near geometry) regions still spread
// over all XCDs (one contiguous band per XCD measured 1.8x SLOWER on the lighting pass: load imbalance).
// Placement is a performance hint only: any dispatch order produces the same pixels.
template <uint32_t STRIPE>
__device__ __forceinline__ void tile_origin(uint32_t& bx, uint32_t& by)
{
    const uint32_t nbx = gridDim.x, n = nbx * gridDim.y;
    uint32_t b = blockIdx.y * nbx + blockIdx.x;
    if (STRIPE > 0) {
        const uint32_t chunk = nbx * STRIPE;              // tiles per stripe
        const uint32_t full = (n / (chunk * 8u)) * (chunk * 8u);   // tiles covered by whole 8-stripe groups
        if (b < full) {
            const uint32_t xcd = b & 7u, k = b >> 3;      // k-th tile this XCD receives
            const uint32_t j = k / chunk, o = k - j * chunk;
            b = (j * 8u + xcd) * chunk + o;
        }                                                 // the tail keeps its ids (bijective)
    }
    by = b / nbx;
    bx = b - by * nbx;
}

// Two-dimensional variant for the SSAO pass: the (padded) grid is cut into super-tiles of SX x SY workgroup tiles; whole
// super-tiles are dealt to the XCDs, so the depth texels an XCD's taps reach form a compact block instead of a
// full-width stripe (at 7680 x 4320 a stripe of the depth plane no longer fits the 4 MiB L2).  gridDim is a multiple of
// (SX, SY); tiles outside the frame exit at once.
__device__ __forceinline__ void tile_origin_2d(uint32_t& bx, uint32_t& by, uint32_t SX, uint32_t SY)
{
    const uint32_t ncx = gridDim.x / SX, T = SX * SY, nst = ncx * (gridDim.y / SY);
    const uint32_t b = blockIdx.y * gridDim.x + blockIdx.x;
    const uint32_t full = (nst & ~7u) * T;                // workgroups in whole groups of eight super-tiles
    uint32_t st, o;
    if (b < full) {
        const uint32_t xcd = b & 7u, k = b >> 3;          // k-th workgroup this XCD receives
        const uint32_t j = k / T;
        o = k - j * T;
        st = j * 8u + xcd;
    } else {                                              // the tail keeps its order (bijective)
        st = b / T;
        o = b - st * T;
    }
    const uint32_t sty = st / ncx, stx = st - sty * ncx, oy = o / SX, ox = o - oy * SX;
    bx = stx * SX + ox;
    by = sty * SY + oy;
}

// ---- SSAO ------------------------------------------------------------------------------------------------
// Re-lays the D24 depth plane as decoded {d(x, y), d(x, y+1)} entries with a BORDER guard band (ssao_core.hpp "depth pairs"):
// one lane per two horizontally adjacent entries (a 16-byte store), rows y = -2 .. H, entries x = -2 .. W+1; a workgroup
// covers 8 rows (two per wave) x 128 entries = 16 blocks of 8 x 8 padded texels.
// By-products: the coarse geometry map of the sky shortcut and the nearest-depth map of the tap culling (ssao_core.hpp).
// A wave covers texel columns [128 blockIdx.x - 2, 128 blockIdx.x + 126) of its rows (through the .x / .z components of its
// entries); if any texel of a row lies below the clear depth it stamps the cell (blockIdx.x, y / 32) with this frame's stamp --
// a plain store: every writer of a cell stores the same value, and a stale or uninitialised word can only read as "geometry"
// (no shortcut), never as "sky".  The block minima go through a 4-lane shuffle and LDS; positions outside the padded plane
// count as the clear depth, which is what a footprint reaching them reads.
__global__ __launch_bounds__(256) void depth_pairs_kernel(const uint32_t* __restrict__ depth, f4a* __restrict__ pairs, uint32_t* __restrict__ geo,
                                                          float* __restrict__ zmin, uint32_t stamp, uint32_t W, uint32_t H, CullParams cull)
{
    __shared__ float s_min[4][16];
    const uint32_t halfPitch = depth_pairs_pitch(W) / 2u;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t px2 = blockIdx.x * 64u + lane;
    float m = 1.0f;
#pragma unroll
    for (uint32_t r = 0; r < 2u; ++r) {
        const uint32_t py = blockIdx.y * 8u + wave * 2u + r;
        bool geometry = false;
        if (py < H + 3u && px2 < halfPitch) {
            const f4a e = depth_pairs_entry2(depth, W, H, 2 * (int)px2 - 2, (int)py - 2);
            pairs[py * halfPitch + px2] = e;
            geometry = (e.x != 1.0f) | (e.z != 1.0f);              // texels (x, y), (x + 1, y): BORDER and clear-depth texels decode to 1.0
            m = __builtin_fminf(m, __builtin_fminf(e.x, e.z));
        }
        const uint32_t y = py - 2u;
        if (__builtin_amdgcn_ballot_w64(geometry) != 0 && lane == 0 && y < H)
            geo[(y >> 5) * geo_map_cols(W) + blockIdx.x] = stamp;
    }
    // 8 padded texel columns = 4 lanes
    m = __builtin_fminf(m, __shfl_xor(m, 1));
    m = __builtin_fminf(m, __shfl_xor(m, 2));
    if ((lane & 3u) == 0) s_min[wave][lane >> 2] = m;
    __syncthreads();
    if (threadIdx.x < 16u) {
        const uint32_t cx = blockIdx.x * 16u + threadIdx.x;
        if (cx < zmin_map_cols(W) && blockIdx.y < zmin_map_rows(H)) {
            const float mm = __builtin_fminf(__builtin_fminf(s_min[0][threadIdx.x], s_min[1][threadIdx.x]),
                                             __builtin_fminf(s_min[2][threadIdx.x], s_min[3][threadIdx.x]));
            zmin[blockIdx.y * zmin_map_cols(W) + cx] = zmin_cell_value(cull.A, cull.B, mm);
        }
    }
}

// The lookup map of the tap culling: per block position the smallest of the 2 x 2 block values starting there.
__global__ __launch_bounds__(256) void zmin_combine_kernel(const float* __restrict__ zmin, float* __restrict__ zcull, uint32_t cols, uint32_t rows)
{
    const uint32_t cx = blockIdx.x * 64u + (threadIdx.x & 63u), cy = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (cx < cols && cy < rows) zcull[cy * cols + cx] = zmin_combine(zmin, cols, rows, cx, cy);
}

// Shaders/Ssao.hlsl:117-199 over half-res rows [row0, row1).  EMIT_AO = false builds only the edge workspace.
// PAIRS: the depth taps read the pairs plane of the edge workspace (built by depth_pairs_kernel earlier on the stream).
template <bool EMIT_AO, bool PAIRS>
__global__ __launch_bounds__(256) void ssao_kernel(crychic_ssao_constants cb, const u2* __restrict__ normal,
                                                   const uint32_t* __restrict__ depth,
                                                   const uint32_t* __restrict__ randvec,
                                                   uint16_t* __restrict__ ambient, EdgePlane edge, uint32_t W,
                                                   uint32_t H, uint32_t row0, uint32_t row1, uint32_t SX, uint32_t SY, int sparseProjTex,
                                                   SkyReach sky, uint32_t stamp, int cullEnabled)
{
    const uint32_t w2 = W / 2;
    uint32_t bx, by;
    tile_origin_2d(bx, by, SX, SY);
    const uint32_t x = bx * 64u + (threadIdx.x & 63u);
    const uint32_t y = row0 + by * 4u + (threadIdx.x >> 6);
    if (x >= w2 || y >= row1) return;

    const DepthPairs dp{ edge.pairs, depth_pairs_pitch(W) };
    const DepthD24 dd{ depth, W, H };
    const SsaoCentre c = PAIRS ? ssao_centre(cb, normal, dp, W, H, (int)x, (int)y) : ssao_centre(cb, normal, depth, W, H, (int)x, (int)y);
    if (edge.nrm) {
        const uint32_t idx = y * w2 + x;
        edge.nrm[idx] = c.nrm_bits;
        edge.vz[idx] = c.vz;
        if (x == 0) edge.gcol[y] = normal[(2u * y + 1u) * W];   // texel (0, 2y+1)
        if (y == row0) edge.grow[x] = normal[2u * x + 1u];      // texel (2x+1, 0)
    }
    // Sky shortcut (ssao_core.hpp): every live lane of this wave is a sky pixel and no cell its taps can reach holds geometry
    if (EMIT_AO && PAIRS && sky.enabled && __builtin_amdgcn_ballot_w64(!c.sky) == 0) {
        const uint32_t x0 = bx * 64u, n = (w2 - x0) < 64u ? (w2 - x0) : 64u;
        const GeoCells g = ssao_sky_cells(sky, W, H, x0, n, y);
        const uint32_t ncx = g.cx1 - g.cx0 + 1u, ncells = ncx * (g.cy1 - g.cy0 + 1u), pitch = geo_map_cols(W);
        bool geometry = false;
        for (uint32_t k = threadIdx.x & 63u; k < ncells; k += n) {          // the wave's n live lanes (0 .. n-1) share the cells
            const uint32_t cy = k / ncx, cx = k - cy * ncx;
            geometry |= edge.geo[(g.cy0 + cy) * pitch + g.cx0 + cx] == stamp;
        }
        if (__builtin_amdgcn_ballot_w64(geometry) == 0) {
            ambient[y * w2 + x] = (uint16_t)0xFFFFu;
            if ((threadIdx.x & 63u) == 0) edge.ones[y * ones_map_cols(W) + bx] = stamp;
            return;
        }
    }
    if (EMIT_AO) {
        uint32_t v;
        if (PAIRS && cullEnabled) v = ssao_pixel(cb, c, dp, randvec, W, H, x, y, sparseProjTex != 0, ZminMap{ edge.zcull, zmin_map_cols(W) });
        else if (PAIRS) v = ssao_pixel(cb, c, dp, randvec, W, H, x, y, sparseProjTex != 0);
        else v = ssao_pixel(cb, c, dd, randvec, W, H, x, y, sparseProjTex != 0);
        ambient[y * w2 + x] = (uint16_t)v;
        // unoccluded-wavefront map (ssao_core.hpp "unoccluded tiles"): lane 0 is live whenever the wave is (x = 64 bx < w2)
        if (PAIRS && stamp != 0u && __builtin_amdgcn_ballot_w64(v != 0xFFFFu) == 0 && (threadIdx.x & 63u) == 0)
            edge.ones[y * ones_map_cols(W) + bx] = stamp;
    }
}

// ---- bilateral blur ------------------------------------------------------------------------------------------
// Shaders/SsaoBlur.hlsl:85-146, one sweep.  A 64 x 16 output tile plus its 5-pixel apron along the sweep axis is
// staged once in LDS as pre-decoded floats (normal.xyz + linear depth as one 16-byte entry, ambient as one dword), so
// the 11-tap window of every pixel is served by one ds_read_b128 + one ds_read_b32 per tap instead of 3 global
// fetches + 4 format conversions.  Consecutive lanes read consecutive 16-byte entries (conflict-free for both
// directions: the horizontal window slides along a staged row, the vertical one hops whole rows).
template <bool HORZ, bool RECORD>
__global__ __launch_bounds__(256) void blur_kernel(crychic_ssao_constants cb, EdgePlane edge,
                                                   const uint16_t* __restrict__ in, uint16_t* __restrict__ out,
                                                   uint32_t W, uint32_t H, uint32_t row0, uint32_t row1, uint32_t stamp, int onesMargin)
{
    constexpr int BW = 64, BH = 16, R = 5;
    constexpr int SW = HORZ ? BW + 2 * R : BW;
    constexpr int SH = HORZ ? BH : BH + 2 * R;
    __shared__ f4a s_nz[SW * SH];
    __shared__ float s_a[SW * SH];

    const int w2 = (int)(W / 2), h2 = (int)(H / 2);
    uint32_t bx, by;
    tile_origin<4>(bx, by);
    const int x0 = (int)bx * BW, y0 = (int)row0 + (int)by * BH;
    if (RECORD && stamp != 0u) {
        // Unoccluded tile (ssao_core.hpp): every wavefront of the SSAO pass within reach of this frame's sweeps wrote 65535
        const OnesRegion g = blur_ones_region((uint32_t)w2, (uint32_t)h2, x0, y0, BW, BH, onesMargin);
        const uint32_t ncol = g.c1 - g.c0 + 1u, ncell = ncol * (g.r1 - g.r0 + 1u), pitch = ones_map_cols(W);
        bool occluded = false;
        for (uint32_t k = threadIdx.x; k < ncell; k += 256u) {
            const uint32_t r = k / ncol, c = k - r * ncol;
            occluded |= edge.ones[(g.r0 + r) * pitch + g.c0 + c] != stamp;
        }
        if (!__syncthreads_or(occluded)) {
            // The recorded decision is "centre tap only": a neighbouring tile that replays these pixels (the fused replay sweeps
            // recompute their apron rows) then gets w5 * 1 * rcp(w5) -> 65535, the value every other decision would give too.
            const int tx = (int)(threadIdx.x & 63u), tyb = (int)(threadIdx.x >> 6), x = x0 + tx;
            uint16_t* __restrict__ mask_o = HORZ ? edge.mask_h : edge.mask_v;
            float* __restrict__ total_o = HORZ ? edge.total_h : edge.total_v;
#pragma unroll
            for (int j = 0; j < BH / 4; ++j) {
                const int y = y0 + tyb + 4 * j;
                if (x < w2 && y < (int)row1) {
                    const uint32_t p = (uint32_t)y * (uint32_t)w2 + (uint32_t)x;
                    out[p] = (uint16_t)0xFFFFu;
                    mask_o[p] = (uint16_t)(1u << 5);
                    total_o[p] = cb.BlurWeights[1][1];      // w[5]
                }
            }
            return;
        }
    }
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
    uint16_t* __restrict__ mask_out = HORZ ? edge.mask_h : edge.mask_v;
    float* __restrict__ total_out = HORZ ? edge.total_h : edge.total_v;
#pragma unroll
    for (int j = 0; j < BH / 4; ++j) {
        const int ty = tyb + 4 * j;
        const int y = y0 + ty;
        if (x < w2 && y < (int)row1) {
            const BlurOut o = blur_pixel_full(&cb.BlurWeights[0][0], [&](int i) {
                const int idx = HORZ ? ty * SW + tx + i : (ty + i) * SW + tx;
                const f4a q = s_nz[idx];
                return BlurTap{ f3{ q.x, q.y, q.z }, q.w, s_a[idx] };
            });
            const uint32_t p = (uint32_t)y * (uint32_t)w2 + (uint32_t)x;
            out[p] = (uint16_t)o.value;
            if (RECORD) { mask_out[p] = (uint16_t)o.mask; total_out[p] = o.total; }
        }
    }
}

// A later sweep of the same direction: the per-tap decisions and totalWeight recorded by the RECORD sweep are
// replayed, so only the ambient tile (+ apron) is staged and each tap costs one LDS read, one multiply-add and a
// select.  Bit-identical to blur_kernel (same float operations in the same order).
template <bool HORZ>
__global__ __launch_bounds__(256) void blur_replay_kernel(crychic_ssao_constants cb, EdgePlane edge,
                                                          const uint16_t* __restrict__ in, uint16_t* __restrict__ out,
                                                          uint32_t W, uint32_t H, uint32_t row0, uint32_t row1, int onesShortcut)
{
    constexpr int BW = 64, BH = 16, R = 5;
    constexpr int SW = HORZ ? BW + 2 * R : BW;
    constexpr int SH = HORZ ? BH : BH + 2 * R;
    __shared__ float s_a[SW * SH];

    const int w2 = (int)(W / 2), h2 = (int)(H / 2);
    uint32_t bx, by;
    tile_origin<4>(bx, by);
    const int x0 = (int)bx * BW, y0 = (int)row0 + (int)by * BH;
    const int sx0 = HORZ ? x0 - R : x0, sy0 = HORZ ? y0 : y0 - R;
    const int tx = (int)(threadIdx.x & 63u), tyb = (int)(threadIdx.x >> 6);
    const int x = x0 + tx;
    const uint16_t* __restrict__ mask_in = HORZ ? edge.mask_h : edge.mask_v;
    const float* __restrict__ total_in = HORZ ? edge.total_h : edge.total_v;
    bool allOne = onesShortcut != 0;
    for (int k = (int)threadIdx.x; k < SW * SH; k += 256) {
        const int ly = k / SW, lx = k - ly * SW;
        const int cx = clampi(sx0 + lx, 0, w2 - 1), cy = clampi(sy0 + ly, 0, h2 - 1);   // ambient: point / CLAMP
        const uint32_t raw = in[(uint32_t)cy * (uint32_t)w2 + (uint32_t)cx];
        allOne = allOne && raw == 0xFFFFu;
        s_a[k] = unorm16_to_float(raw);
    }
    // A window whose ambient values are all 1.0 blurs to exactly 1.0 whatever the recorded decisions are: the colour sum adds
    // the very weights the recorded total was built from, in the same order, so colour == total bit for bit and x / x = 1
    // for the finite positive totals that finite positive weights give (blur_weights_positive(), checked by the launcher).
    // Most of the frame (sky, unoccluded surfaces) is like that; the whole tile then skips masks, taps and divisions.
    if (__syncthreads_and(allOne)) {
#pragma unroll
        for (int j = 0; j < BH / 4; ++j) {
            const int y = y0 + tyb + 4 * j;
            if (x < w2 && y < (int)row1) out[(uint32_t)y * (uint32_t)w2 + (uint32_t)x] = (uint16_t)0xFFFFu;
        }
        return;
    }
    uint32_t m[BH / 4];
    float tot[BH / 4];
#pragma unroll
    for (int j = 0; j < BH / 4; ++j) {
        const int y = y0 + tyb + 4 * j;
        const bool live = (x < w2) && (y < (int)row1);
        const uint32_t p = live ? (uint32_t)y * (uint32_t)w2 + (uint32_t)x : 0u;
        m[j] = mask_in[p];
        tot[j] = total_in[p];
    }

#pragma unroll
    for (int j = 0; j < BH / 4; ++j) {
        const int ty = tyb + 4 * j;
        const int y = y0 + ty;
        if (x < w2 && y < (int)row1) {
            const uint32_t v = blur_pixel_replay(&cb.BlurWeights[0][0], m[j], tot[j], [&](int i) {
                return s_a[HORZ ? ty * SW + tx + i : (ty + i) * SW + tx];
            });
            out[(uint32_t)y * (uint32_t)w2 + (uint32_t)x] = (uint16_t)v;
        }
    }
}

// One whole replay iteration (horizontal then vertical sweep, SsaoBlur ping-pong 0 -> 1 -> 0 of Ssao.cpp:240-241) in one
// launch: the horizontal results of the tile's rows plus a 5-row apron stay in LDS -- quantised to R16_UNORM and decoded
// again exactly as the round trip through the ambient map does -- and feed the vertical sweep.  Bit-identical to
// blur_replay_kernel<true> followed by blur_replay_kernel<false>; `out` must not alias `in`.
__global__ __launch_bounds__(256) void blur_replay_pair_kernel(crychic_ssao_constants cb, EdgePlane edge,
                                                               const uint16_t* __restrict__ in, uint16_t* __restrict__ out,
                                                               uint32_t W, uint32_t H, uint32_t row0, uint32_t row1, int onesShortcut)
{
    constexpr int BW = 64, BH = 16, R = 5;
    constexpr int SW = BW + 2 * R, SH = BH + 2 * R;
    __shared__ float s_in[SW * SH];
    __shared__ float s_mid[BW * SH];

    const int w2 = (int)(W / 2), h2 = (int)(H / 2);
    uint32_t bx, by;
    tile_origin<4>(bx, by);
    const int x0 = (int)bx * BW, y0 = (int)row0 + (int)by * BH;
    const int tx = (int)(threadIdx.x & 63u), tyb = (int)(threadIdx.x >> 6);
    const int x = x0 + tx;
    bool allOne = onesShortcut != 0;
    for (int k = (int)threadIdx.x; k < SW * SH; k += 256) {
        const int ly = k / SW, lx = k - ly * SW;
        const int cx = clampi(x0 - R + lx, 0, w2 - 1), cy = clampi(y0 - R + ly, 0, h2 - 1);   // ambient: point / CLAMP
        const uint32_t raw = in[(uint32_t)cy * (uint32_t)w2 + (uint32_t)cx];
        allOne = allOne && raw == 0xFFFFu;
        s_in[k] = unorm16_to_float(raw);
    }
    if (__syncthreads_and(allOne)) {        // all-ones window: both sweeps return exactly 1.0 (see blur_replay_kernel)
#pragma unroll
        for (int j = 0; j < BH / 4; ++j) {
            const int y = y0 + tyb + 4 * j;
            if (x < w2 && y < (int)row1) out[(uint32_t)y * (uint32_t)w2 + (uint32_t)x] = (uint16_t)0xFFFFu;
        }
        return;
    }
    // the vertical sweep's recorded decisions of this thread's four outputs
    uint32_t m[BH / 4];
    float tot[BH / 4];
#pragma unroll
    for (int j = 0; j < BH / 4; ++j) {
        const int y = y0 + tyb + 4 * j;
        const bool live = (x < w2) && (y < (int)row1);
        const uint32_t p = live ? (uint32_t)y * (uint32_t)w2 + (uint32_t)x : 0u;
        m[j] = edge.mask_v[p];
        tot[j] = edge.total_v[p];
    }
    // horizontal sweep of rows y0-5 .. y0+BH+4 (CLAMPed: a vertical tap above / below the map reads the edge row's result)
    for (int k = (int)threadIdx.x; k < BW * SH; k += 256) {
        const int ly = k >> 6, lx = k & 63;
        const int cy = clampi(y0 - R + ly, 0, h2 - 1), cx = x0 + lx;
        if (cx < w2) {
            const uint32_t p = (uint32_t)cy * (uint32_t)w2 + (uint32_t)cx;
            const uint32_t v = blur_pixel_replay(&cb.BlurWeights[0][0], edge.mask_h[p], edge.total_h[p],
                                                 [&](int i) { return s_in[ly * SW + lx + i]; });
            s_mid[k] = unorm16_to_float(v);
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < BH / 4; ++j) {
        const int ty = tyb + 4 * j;
        const int y = y0 + ty;
        if (x < w2 && y < (int)row1) {
            const uint32_t v = blur_pixel_replay(&cb.BlurWeights[0][0], m[j], tot[j], [&](int i) { return s_mid[(ty + i) * BW + tx]; });
            out[(uint32_t)y * (uint32_t)w2 + (uint32_t)x] = (uint16_t)v;
        }
    }
}

// ---- deferred lighting -----------------------------------------------------------------------------------------
// Shaders/DeferredShading.hlsl:23-101 as a full-screen pass over rows [row0, row1), masked by depth < 1.
template <bool ZERO_RADIUS, bool FIX>
__global__ __launch_bounds__(256) void light_kernel(LightParams P, const f4a* __restrict__ g0,
                                                    const f4a* __restrict__ g1, const f4a* __restrict__ g2,
                                                    const uint32_t* __restrict__ depth,
                                                    const uint16_t* __restrict__ ambient,
                                                    const uint32_t* __restrict__ cube, uint32_t* __restrict__ out,
                                                    f4a* __restrict__ radiance, uint32_t row0, uint32_t row1)
{
    uint32_t bx, by;
    tile_origin<0>(bx, by);
    const uint32_t x = bx * 64u + (threadIdx.x & 63u);
    const uint32_t y = row0 + by * 4u + (threadIdx.x >> 6);
    if (x >= P.W || y >= row1) return;
    const uint32_t idx = y * P.W + x;
    f4 lit;
    // coverage: the reference re-rasterises the opaque items with LESS against depth cleared to 1.0
    // (CRYCHIC.cpp:248,273) -- exactly the pixels whose normal/depth pass depth is below the clear value.
    if ((depth[idx] & 0x00FFFFFFu) < 0x00FFFFFFu) {
        lit = light_pixel<ZERO_RADIUS, NoPointLights, FIX>(P, g0[idx], g1[idx], g2[idx], ambient, cube);
    } else if (P.flags & CRYCHIC_LIGHT_SKY) {
        lit = sky_pixel(P, cube, x, y);
    } else {
        lit = f4{ 0.690196097f, 0.768627524f, 0.870588303f, 1.0f };  // Colors::LightSteelBlue, CRYCHIC.cpp:247
    }
    if (radiance) radiance[idx] = f4a{ lit.x, lit.y, lit.z, lit.w };
    out[idx] = pack_rgba8(lit);
}

// ---- deferred lighting with point lights (extension, BASELINE configs[4]) ------------------------------------------
// Same pass plus NUM_POINT_LIGHTS point lights.  Tiled light culling in LDS: the 64 x 4-pixel tile of a workgroup
// reduces the world-space bounding box of its covered pixels (wave shuffles, then LDS), every lane then tests lights
// against the box (sphere of radius FalloffEnd vs AABB, conservatively inflated) and sets the light's bit in an LDS mask;
// each pixel finally walks the set bits in ascending index order -- the accumulation order of the un-culled loop -- and
// applies the exact per-pixel range test, so culling never changes a bit of the result.
template <bool ZERO_RADIUS>
__global__ __launch_bounds__(256) void light_points_kernel(LightParams P, const f4a* __restrict__ g0, const f4a* __restrict__ g1,
                                                           const f4a* __restrict__ g2, const uint32_t* __restrict__ depth,
                                                           const uint16_t* __restrict__ ambient, const uint32_t* __restrict__ cube,
                                                           uint32_t* __restrict__ out, f4a* __restrict__ radiance, uint32_t row0,
                                                           uint32_t row1)
{
    __shared__ float s_box[4][6];
    __shared__ uint32_t s_mask[kMaxPointLights / 32];
    uint32_t bx, by;
    tile_origin<0>(bx, by);
    const uint32_t x = bx * 64u + (threadIdx.x & 63u);
    const uint32_t y = row0 + by * 4u + (threadIdx.x >> 6);
    const bool inFrame = (x < P.W) && (y < row1);
    const uint32_t idx = inFrame ? y * P.W + x : 0u;
    const bool covered = inFrame && ((depth[idx] & 0x00FFFFFFu) < 0x00FFFFFFu);
    f4a G0{ 0, 0, 0, 0 };
    if (covered) G0 = g0[idx];

    // 1. tile bounding box of the covered pixels' world positions
    const float big = 3.0e38f;
    float lo[3] = { covered ? G0.x : big, covered ? G0.y : big, covered ? G0.z : big };
    float hi[3] = { covered ? G0.x : -big, covered ? G0.y : -big, covered ? G0.z : -big };
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            lo[c] = __builtin_fminf(lo[c], __shfl_xor(lo[c], off));
            hi[c] = __builtin_fmaxf(hi[c], __shfl_xor(hi[c], off));
        }
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63u) == 0) { for (int c = 0; c < 3; ++c) { s_box[wave][c] = lo[c]; s_box[wave][3 + c] = hi[c]; } }
    if (threadIdx.x < kMaxPointLights / 32) s_mask[threadIdx.x] = 0u;
    __syncthreads();
    float blo[3], bhi[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        blo[c] = __builtin_fminf(__builtin_fminf(s_box[0][c], s_box[1][c]), __builtin_fminf(s_box[2][c], s_box[3][c]));
        bhi[c] = __builtin_fmaxf(__builtin_fmaxf(s_box[0][3 + c], s_box[1][3 + c]), __builtin_fmaxf(s_box[2][3 + c], s_box[3][3 + c]));
    }
    // 2. cull: light l touches the tile if dist(Position, box) <= FalloffEnd (inflated: the per-pixel test is the exact one)
    const bool anyCovered = blo[0] <= bhi[0];
    if (anyCovered)
        for (uint32_t l = threadIdx.x; l < P.numPointLights; l += 256u) {
            const crychic_light L = P.pointLights[l];
            float d2 = 0.0f;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float p = L.Position[c];
                const float e = __builtin_fmaxf(__builtin_fmaxf(blo[c] - p, p - bhi[c]), 0.0f);
                d2 += e * e;
            }
            const float r = L.FalloffEnd * 1.0001f + 1.0e-3f;
            if (d2 <= r * r) atomicOr(&s_mask[l >> 5], 1u << (l & 31u));
        }
    __syncthreads();
    if (!inFrame) return;

    // 3. shade
    f4 lit;
    if (covered) {
        auto culled = [&](f3 pos, f3 albedo, float roughness, float metalness, f3 normal, f3 view, f3& result, bool fixQ3, bool fixQ4) {
            const uint32_t words = (P.numPointLights + 31u) >> 5;
            for (uint32_t w = 0; w < words; ++w) {
                uint32_t m = s_mask[w];
                while (m) {
                    const uint32_t b = (uint32_t)__builtin_ctz(m);
                    m &= m - 1u;
                    pbr_point_light(P.pointLights[w * 32u + b], pos, albedo, roughness, metalness, normal, view, result, fixQ3, fixQ4);
                }
            }
        };
        lit = light_pixel<ZERO_RADIUS, decltype(culled), true>(P, G0, g1[idx], g2[idx], ambient, cube, culled);
    } else if (P.flags & CRYCHIC_LIGHT_SKY) {
        lit = sky_pixel(P, cube, x, y);
    } else {
        lit = f4{ 0.690196097f, 0.768627524f, 0.870588303f, 1.0f };
    }
    if (radiance) radiance[idx] = f4a{ lit.x, lit.y, lit.z, lit.w };
    out[idx] = pack_rgba8(lit);
}

// ---- launchers ---------------------------------------------------------------------------------------------------
// All 11 blur weights finite and in (0, 1e30): every total the record sweep can produce is then finite and positive.
static bool blur_weights_positive(const crychic_ssao_constants& cb)
{
    const float* w = &cb.BlurWeights[0][0];
    for (int i = 0; i < 11; ++i)
        if (!(w[i] > 0.0f && w[i] < 1.0e30f)) return false;
    return true;
}

static inline dim3 grid_for(uint32_t width, uint32_t rows, uint32_t rows_per_block = 4u)
{
    return dim3((width + 63u) / 64u, (rows + rows_per_block - 1u) / rows_per_block, 1);
}

hipError_t launch_depth_pairs(const crychic_ssao_constants& cb, const uint32_t* depth, void* edge_base, uint32_t W, uint32_t H, uint32_t stamp,
                              hipStream_t stream)
{
    const EdgePlane e = edge_plane_carve(edge_base, W, H);
    // grid.x == geo_map_cols(W); 16 cells of the nearest-depth map per workgroup in x, one cell row in y
    const dim3 grid((depth_pairs_pitch(W) / 2u + 63u) / 64u, zmin_map_rows(H), 1);
    const CullParams cull = ssao_cull_params(cb);
    hipLaunchKernelGGL(depth_pairs_kernel, grid, dim3(256), 0, stream, depth, (f4a*)const_cast<void*>(e.pairs), e.geo, e.zmin, stamp, W, H, cull);
    if (cull.enabled) {
        const uint32_t cols = zmin_map_cols(W), rows = zmin_map_rows(H);
        hipLaunchKernelGGL(zmin_combine_kernel, dim3((cols + 63u) / 64u, (rows + 3u) / 4u, 1), dim3(256), 0, stream, e.zmin, e.zcull, cols, rows);
    }
    return hipGetLastError();
}

hipError_t launch_ssao(const crychic_ssao_constants& cb, const void* normal, const uint32_t* depth,
                       const uint8_t* randvec, uint16_t* ambient, void* edge_base, uint32_t W, uint32_t H,
                       uint32_t row0, uint32_t rows, bool emit_ao, bool use_pairs, uint32_t stamp, hipStream_t stream)
{
    if (rows == 0) return hipSuccess;
    EdgePlane e{};
    if (edge_base) e = edge_plane_carve(edge_base, W, H);
    if (use_pairs && !edge_base) return hipErrorInvalidValue;
    dim3 grid = grid_for(W / 2, rows);
    // super-tiles of SX x SY workgroup tiles (64 x 4 half-res pixels each) per XCD; the grid is padded to whole super-tiles
    // 4 x 32 tiles = 256 x 128 half-res pixels (512 x 256 depth texels, 1 MB of the pairs plane plus the reach of the taps
    // around it, against 4 MiB of L2): best of 11 shapes on the pairs plane at 4K (pass 112.2 us with 6 x 32, the best shape on
    // the raw D24 plane, 106.4 us with 4 x 32; full-width stripes of 16 tile rows were 208 us in round 1)
    uint32_t SX = 4u, SY = 32u;
    SX = SX > grid.x ? grid.x : SX;
    SY = SY > grid.y ? grid.y : SY;
    grid.x = (grid.x + SX - 1u) / SX * SX;
    grid.y = (grid.y + SY - 1u) / SY * SY;
    const int sparse = ssao_projtex_is_sparse(cb) ? 1 : 0;
    SkyReach sky = ssao_sky_reach(cb, W, H);
    if (!use_pairs || stamp == 0u) sky.enabled = 0;      // stamp 0: the caller did not build the geometry map
    const int cull = (use_pairs && stamp != 0u && ssao_cull_params(cb).enabled) ? 1 : 0;    // the nearest-depth map comes with the pairs plane
#define CRY_LAUNCH_SSAO(K) hipLaunchKernelGGL(K, grid, dim3(256), 0, stream, cb, (const u2*)normal, depth, (const uint32_t*)randvec, ambient, e, W, H, row0, row0 + rows, SX, SY, sparse, sky, stamp, cull)
    if (emit_ao && use_pairs) CRY_LAUNCH_SSAO((ssao_kernel<true, true>));
    else if (emit_ao) CRY_LAUNCH_SSAO((ssao_kernel<true, false>));
    else CRY_LAUNCH_SSAO((ssao_kernel<false, false>));
#undef CRY_LAUNCH_SSAO
    return hipGetLastError();
}

hipError_t launch_blur(const crychic_ssao_constants& cb, const void* edge_base, const uint16_t* in, uint16_t* out,
                       uint32_t W, uint32_t H, bool horizontal, BlurMode mode, uint32_t row0, uint32_t rows,
                       uint32_t stamp, int onesMargin, hipStream_t stream)
{
    if (rows == 0) return hipSuccess;
    const EdgePlane e = edge_plane_carve(const_cast<void*>(edge_base), W, H);
    const dim3 grid = grid_for(W / 2, rows, 16u);
    // the unoccluded-tile exit needs finite positive weights (x * rcp(x) of a finite positive total) and the SSAO pass's map
    if (mode != BlurMode::Record || !blur_weights_positive(cb)) stamp = 0u;
#define CRY_LAUNCH_BLUR(K) hipLaunchKernelGGL(K, grid, dim3(256), 0, stream, cb, e, in, out, W, H, row0, row0 + rows, stamp, onesMargin)
    if (mode == BlurMode::Replay) {
        const int ones = blur_weights_positive(cb) ? 1 : 0;
        if (horizontal) hipLaunchKernelGGL(blur_replay_kernel<true>, grid, dim3(256), 0, stream, cb, e, in, out, W, H, row0, row0 + rows, ones);
        else hipLaunchKernelGGL(blur_replay_kernel<false>, grid, dim3(256), 0, stream, cb, e, in, out, W, H, row0, row0 + rows, ones);
    } else if (mode == BlurMode::Record) {
        if (horizontal) CRY_LAUNCH_BLUR((blur_kernel<true, true>)); else CRY_LAUNCH_BLUR((blur_kernel<false, true>));
    } else {
        if (horizontal) CRY_LAUNCH_BLUR((blur_kernel<true, false>)); else CRY_LAUNCH_BLUR((blur_kernel<false, false>));
    }
#undef CRY_LAUNCH_BLUR
    return hipGetLastError();
}

hipError_t launch_blur_replay_pair(const crychic_ssao_constants& cb, const void* edge_base, const uint16_t* in, uint16_t* out,
                                   uint32_t W, uint32_t H, uint32_t row0, uint32_t rows, hipStream_t stream)
{
    if (rows == 0) return hipSuccess;
    const EdgePlane e = edge_plane_carve(const_cast<void*>(edge_base), W, H);
    hipLaunchKernelGGL(blur_replay_pair_kernel, grid_for(W / 2, rows, 16u), dim3(256), 0, stream, cb, e, in, out, W, H, row0, row0 + rows,
                       blur_weights_positive(cb) ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_light(const LightParams& P, const float* g0, const float* g1, const float* g2,
                        const uint32_t* depth, const uint16_t* ambient, const uint8_t* cube, uint8_t* out,
                        float* radiance, uint32_t row0, uint32_t rows, hipStream_t stream)
{
    if (rows == 0) return hipSuccess;
    const dim3 grid = grid_for(P.W, rows);
    if (P.numPointLights) {
        if (P.pcfSearchRadius == 0.0f)
            hipLaunchKernelGGL(light_points_kernel<true>, grid, dim3(256), 0, stream, P, (const f4a*)g0, (const f4a*)g1, (const f4a*)g2,
                               depth, ambient, (const uint32_t*)cube, (uint32_t*)out, (f4a*)radiance, row0, row0 + rows);
        else
            hipLaunchKernelGGL(light_points_kernel<false>, grid, dim3(256), 0, stream, P, (const f4a*)g0, (const f4a*)g1, (const f4a*)g2,
                               depth, ambient, (const uint32_t*)cube, (uint32_t*)out, (f4a*)radiance, row0, row0 + rows);
        return hipGetLastError();
    }
    const bool fix = (P.flags & (CRYCHIC_FIX_Q1 | CRYCHIC_FIX_Q3 | CRYCHIC_FIX_Q4)) != 0;
#define CRY_LAUNCH_LIGHT(K) hipLaunchKernelGGL(K, grid, dim3(256), 0, stream, P, (const f4a*)g0, (const f4a*)g1, (const f4a*)g2, depth, ambient, \
                                               (const uint32_t*)cube, (uint32_t*)out, (f4a*)radiance, row0, row0 + rows)
    if (P.pcfSearchRadius == 0.0f) { if (fix) CRY_LAUNCH_LIGHT((light_kernel<true, true>)); else CRY_LAUNCH_LIGHT((light_kernel<true, false>)); }
    else { if (fix) CRY_LAUNCH_LIGHT((light_kernel<false, true>)); else CRY_LAUNCH_LIGHT((light_kernel<false, false>)); }
#undef CRY_LAUNCH_LIGHT
    return hipGetLastError();
}

}  // namespace cry
