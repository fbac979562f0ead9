// kernels.hip -- gfx950 kernels of the CRYCHIC hot path and their stream-ordered launchers.
//
// Work decomposition (all kernels): one lane per output pixel, 64 consecutive pixels of one row per
// wavefront so every plane access is a contiguous 64-lane burst (16 B/lane on the fp32 G-buffer planes,
// 8 B/lane on the fp16 normal plane and the depth row pairs), 4 rows per 256-thread workgroup.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "kernels.hpp"
#include "ssao_core.hpp"
#include "blur_tiles.hpp"
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
// Re-lays the D24 depth plane as decoded {d(x, y), d(x, y+1)} entries with a BORDER guard band (ssao_core.hpp "depth pairs").
// One wavefront per 8 entry rows x 128 entry columns: a lane owns two horizontally adjacent entries of each of the 8 rows (one
// 16-byte store per row) and reads the 9 texel rows they are built from once (one 8-byte load per row).  Entry rows are
// py = 8 cy .. 8 cy + 7 of cell row cy (texel rows py - 2 .. py - 1), entry columns 128 seg .. 128 seg + 127 of segment seg.
// By-products, both without LDS or barriers:
//   * the nearest-depth map of the tap culling: cell (cx, cy) = smallest decoded texel of padded texels [8 cx, 8 cx + 8] x
//     [8 cy, 8 cy + 8] -- the 9 rows a lane holds, over the 4 lanes of the cell plus the first column of the next lane (the last
//     cell of a wavefront takes that column from one extra load); positions outside the plane count as the clear depth, which
//     is what a footprint reaching them reads;
//   * the coarse geometry map of the sky shortcut: a wavefront covers texel columns [128 seg - 2, 128 seg + 126) of its rows; if
//     any texel of rows belonging to cell row c lies below the clear depth it stamps cell (seg, c) with this frame's stamp -- a
//     plain store: every writer of a cell stores the same value, and a stale or uninitialised word can only read as "geometry"
//     (no shortcut), never as "sky".
template <bool WRITE_PAIRS>
__global__ __launch_bounds__(256) void depth_pairs_kernel(const uint32_t* __restrict__ depth, f4a* __restrict__ pairs, uint32_t* __restrict__ geo,
                                                          float* __restrict__ zcull, uint32_t stamp, uint32_t W, uint32_t H, CullParams cull,
                                                          uint32_t cellRow0)
{
    const uint32_t halfPitch = depth_pairs_pitch(W) / 2u;
    const uint32_t lane = threadIdx.x & 63u, seg = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (seg * 64u >= halfPitch) return;                          // wave-uniform
    const uint32_t px2 = seg * 64u + lane, cy = cellRow0 + blockIdx.y;
    const int x = 2 * (int)px2 - 2, y0 = 8 * (int)cy - 2;         // first texel column of the lane, first texel row of the wave
    const bool inx = (uint32_t)x < W;                             // W even, x even: x + 1 < W as well
    const uint32_t cx = inx ? (uint32_t)x : 0u;
    float r0[9], r1[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int y = y0 + k;
        RawPair t{ 0x00FFFFFFu, 0x00FFFFFFu };
        if (inx && (uint32_t)y < H) t = load_at<RawPair>(depth, (mul24((uint32_t)y, W) + cx) * 4u);
        r0[k] = d24_to_float(t.lo);
        r1[k] = d24_to_float(t.hi);
    }
    const bool live = px2 < halfPitch;
    // coarse geometry map: rows y0 .. y0 + 7 (row y0 + 8 is the next cell row's first) fall into at most two 32-row cells
    const int yFirst = y0 < 0 ? 0 : y0;
    const uint32_t c0 = (uint32_t)yFirst >> 5;
    bool geoA = false, geoB = false;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int y = y0 + k;
        const bool g = (r0[k] != 1.0f) | (r1[k] != 1.0f);          // BORDER and clear-depth texels decode to 1.0
        if ((uint32_t)y < H) { if (((uint32_t)y >> 5) == c0) geoA |= g; else geoB |= g; }
    }
    const bool anyA = __builtin_amdgcn_ballot_w64(geoA) != 0, anyB = __builtin_amdgcn_ballot_w64(geoB) != 0;
    if (lane == 0) {
        if (anyA) geo[c0 * geo_map_cols(W) + seg] = stamp;
        if (anyB) geo[(c0 + 1u) * geo_map_cols(W) + seg] = stamp;
    }
    // nearest-depth map
    float mFirst = r0[0], mSecond = r1[0];
#pragma unroll
    for (int k = 1; k < 9; ++k) { mFirst = __builtin_fminf(mFirst, r0[k]); mSecond = __builtin_fminf(mSecond, r1[k]); }
    float m = __builtin_fminf(mFirst, mSecond);
    m = __builtin_fminf(m, __shfl_xor(m, 1));
    m = __builtin_fminf(m, __shfl_xor(m, 2));                       // the cell's own 8 columns
    float next = __shfl(mFirst, (int)((lane | 3u) + 1u) & 63);      // first column of the next cell (lanes 60..63: replaced below)
    {
        // column 128 seg + 126 (texel), rows y0 .. y0 + 8: lanes 0 .. 8 load one texel each
        const int xe = 128 * (int)seg + 126, ye = y0 + (int)lane;
        float v = 1.0f;
        if (lane < 9u && (uint32_t)xe < W && (uint32_t)ye < H) v = d24_to_float(load_at<uint32_t>(depth, (mul24((uint32_t)ye, W) + (uint32_t)xe) * 4u));
        v = __builtin_fminf(v, __shfl_xor(v, 1));
        v = __builtin_fminf(v, __shfl_xor(v, 2));
        v = __builtin_fminf(v, __shfl_xor(v, 4));
        v = __builtin_fminf(v, __shfl_xor(v, 8));
        v = __shfl(v, 0);
        if (lane >= 60u) next = v;
    }
    m = __builtin_fminf(m, next);                                   // the same value in the four lanes of a cell
    // The pairs plane.  A lane's entries (2 px2, 2 px2 + 1) are read by footprints whose left entry is 2 px2 - 1, 2 px2 or 2 px2 + 1:
    // its own cell, and for the first lane of a cell the cell to the left as well.  Entries that only clear cells can reach are
    // read by nobody (ssao_core.hpp "clear cells") and are not written.  The first cell of a wavefront always writes: its left
    // neighbour belongs to another wavefront.
    const float left = __shfl(m, (int)(lane - 1u) & 63);
    const bool unread = cull.clear && m == 1.0f && lane >= 4u && ((lane & 3u) != 0u || left == 1.0f);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint32_t py = 8u * cy + (uint32_t)k;
        if (WRITE_PAIRS && live && !unread && py < H + 3u) pairs[py * halfPitch + px2] = f4a{ r0[k], r0[k + 1], r1[k], r1[k + 1] };
    }
    const uint32_t cellX = seg * 16u + (lane >> 2);
    if ((lane & 3u) == 0 && cellX < zmin_map_cols(W)) zcull[cy * zmin_map_cols(W) + cellX] = zmin_cell_value(cull, m);
}

// Shaders/Ssao.hlsl:117-199 over half-res rows [row0, row1).  EMIT_AO = false builds only the edge workspace.
// PAIRS: the depth taps read the pairs plane of the edge workspace (built by depth_pairs_kernel earlier on the stream).
// MAPS: the depth pass of this frame built the coarse maps (geometry, nearest depth) that the sky shortcut and the tap culling read.
// ROWS: that pass was limited to the footprint rows [prepJ0, prepJ0 + prepNj) (a strip of a multi-GPU frame): taps that leave them
// gather from the raw plane (ssao_core.hpp DepthPairsRows).
struct PrepRows { int j0lo; uint32_t nj; };
template <bool EMIT_AO, bool PAIRS, bool MAPS, bool ROWS>
__global__ __launch_bounds__(256) void ssao_kernel(crychic_ssao_constants cb, const u2* __restrict__ normal,
                                                   const uint32_t* __restrict__ depth,
                                                   const uint32_t* __restrict__ randvec,
                                                   uint16_t* __restrict__ ambient, EdgePlane edge, uint32_t W,
                                                   uint32_t H, uint32_t row0, uint32_t row1, uint32_t SX, uint32_t SY, int sparseProjTex,
                                                   SkyReach sky, uint32_t stamp, int cullEnabled, PrepRows prep, HalfResScale hs)
{
    const uint32_t w2 = W / 2;
    uint32_t bx, by;
    tile_origin_2d(bx, by, SX, SY);
    const uint32_t x = bx * 64u + (threadIdx.x & 63u);
    const uint32_t y = row0 + by * 4u + (threadIdx.x >> 6);
    if (x >= w2 || y >= row1) return;

    const DepthPairs dp{ edge.pairs, depth_pairs_pitch(W) };
    const DepthD24 dd{ depth, W, H };
    // the pixel's own depth comes from the D24 plane (coalesced): its pairs entry may be one the depth pass did not write
    const SsaoCentre c = ssao_centre(cb, normal, depth, W, H, (int)x, (int)y);
    if (edge.nrm) {
        const uint32_t idx = y * w2 + x;
        edge.nrm[idx] = c.nrm_bits;
        edge.vz[idx] = c.vz;
        if (x == 0) edge.gcol[y] = normal[(2u * y + 1u) * W];   // texel (0, 2y+1)
        if (y == row0) edge.grow[x] = normal[2u * x + 1u];      // texel (2x+1, 0)
    }
    // Sky shortcut (ssao_core.hpp): every live lane of this wave is a sky pixel and no cell its taps can reach holds geometry
    if (EMIT_AO && MAPS && sky.enabled && __builtin_amdgcn_ballot_w64(!c.sky) == 0) {
        const uint32_t x0 = bx * 64u, n = (w2 - x0) < 64u ? (w2 - x0) : 64u;
        const GeoCells g = ssao_sky_cells(sky, W, H, x0, n, y);
        const uint32_t ncx = g.cx1 - g.cx0 + 1u, ncells = ncx * (g.cy1 - g.cy0 + 1u), pitch = geo_map_cols(W);
        bool geometry = false;
        for (uint32_t k = threadIdx.x & 63u; k < ncells; k += n) {          // the wave's n live lanes (0 .. n-1) share the cells
            const uint32_t cy = k / ncx, cx = k - cy * ncx;
            geometry |= edge.geo[(g.cy0 + cy) * pitch + g.cx0 + cx] == stamp;
        }
        if (g.known && __builtin_amdgcn_ballot_w64(geometry) == 0) {
            ambient[y * w2 + x] = (uint16_t)0xFFFFu;
            if ((threadIdx.x & 63u) == 0) edge.ones[y * ones_map_cols(W) + bx] = stamp;
            return;
        }
    }
    if (EMIT_AO) {
        uint32_t v;
        const ZminMap zm{ edge.zcull, zmin_map_cols(W) };
        if (ROWS && PAIRS) {      // && MAPS
            const DepthPairsRows dr{ dp, dd, prep.j0lo, prep.nj };
            if (cullEnabled) v = ssao_pixel(cb, c, dr, randvec, W, H, x, y, hs, sparseProjTex != 0, ZminMapRows{ zm, prep.j0lo, prep.nj });
            else v = ssao_pixel(cb, c, dr, randvec, W, H, x, y, hs, sparseProjTex != 0);
        }
        else if (ROWS) {          // MAPS without the pairs plane: the taps gather from the raw plane, the culling map is the strip's
            if (cullEnabled) v = ssao_pixel(cb, c, dd, randvec, W, H, x, y, hs, sparseProjTex != 0, ZminMapRows{ zm, prep.j0lo, prep.nj });
            else v = ssao_pixel(cb, c, dd, randvec, W, H, x, y, hs, sparseProjTex != 0);
        }
        else if (PAIRS && MAPS && cullEnabled) v = ssao_pixel(cb, c, dp, randvec, W, H, x, y, hs, sparseProjTex != 0, zm);
        else if (PAIRS) v = ssao_pixel(cb, c, dp, randvec, W, H, x, y, hs, sparseProjTex != 0);
        else if (MAPS && cullEnabled) v = ssao_pixel(cb, c, dd, randvec, W, H, x, y, hs, sparseProjTex != 0, zm);
        else v = ssao_pixel(cb, c, dd, randvec, W, H, x, y, hs, sparseProjTex != 0);
        ambient[y * w2 + x] = (uint16_t)v;
        // unoccluded-wavefront map (ssao_core.hpp "unoccluded tiles"): lane 0 is live whenever the wave is (x = 64 bx < w2).  The
        // word is written by EVERY wavefront that emits ambient values -- the stamp or 0 -- so no word of a row computed this
        // frame is ever stale.
        if (MAPS && stamp != 0u) {
            const bool allOnes = __builtin_amdgcn_ballot_w64(v != 0xFFFFu) == 0;
            if ((threadIdx.x & 63u) == 0) edge.ones[y * ones_map_cols(W) + bx] = allOnes ? stamp : 0u;
        }
    }
}

// ---- bilateral blur ------------------------------------------------------------------------------------------
// Shaders/SsaoBlur.hlsl:85-146, one self-contained sweep (crychic_ssao_blur == Ssao::BlurAmbientMap(cmdList, bool), Ssao.cpp:245-293).
// A 64 x 16 output tile plus its 5-pixel apron along the sweep axis is staged once in LDS as pre-decoded floats (normal.xyz +
// linear depth as one 16-byte entry, ambient as one dword), so the 11-tap window of every pixel is served by one ds_read_b128 +
// one ds_read_b32 per tap instead of 3 global fetches + 4 format conversions.  Consecutive lanes read consecutive 16-byte
// entries (conflict-free for both directions: the horizontal window slides along a staged row, the vertical one hops whole rows).
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
    uint32_t bx, by;
    tile_origin<4>(bx, by);
    const int x0 = (int)bx * BW, y0 = (int)row0 + (int)by * BH;
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

// The workgroup behind the tile bodies of blur_tiles.hpp.  COHERENT: the ambient planes are read and written with agent-scope
// (device-coherent) accesses -- the single-launch chain, whose tiles hand values to workgroups on other XCDs inside one kernel.
template <bool COHERENT = false>
struct BlockDevT {
    static constexpr int kLanes = 1;
    __device__ int tid() const { return (int)threadIdx.x; }
    __device__ int size() const { return (int)blockDim.x; }
    __device__ void sync() const { __syncthreads(); }
    __device__ bool all(bool p) const { return __syncthreads_and(p) != 0; }
    __device__ bool any(bool p) const { return __syncthreads_or(p) != 0; }
    __device__ int wave() const { return (int)(threadIdx.x >> 6); }
    __device__ int waves() const { return (int)(blockDim.x >> 6); }
    template <class F> __device__ void lanes(F f) const { f((int)(threadIdx.x & 63u)); }
    __device__ bool wave_all(bool p) const { return __builtin_amdgcn_ballot_w64(!p) == 0; }       // over the live lanes of the wavefront
    __device__ bool wave_leader() const { return (threadIdx.x & 63u) == 0; }
    // LDS operations of one wavefront execute in order; what has to be stopped is the compiler moving a lane's reads of OTHER
    // lanes' entries above its own writes (different addresses to it)
    __device__ void wave_sync() const
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    __device__ uint32_t amb_load(const uint16_t* p) const
    {
        if (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return *p;
    }
    __device__ void amb_store(uint16_t* p, uint32_t v) const
    {
        if (COHERENT) __hip_atomic_store(p, (uint16_t)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else *p = (uint16_t)v;
    }
};
using BlockDev = BlockDevT<false>;

// Tiles of the blur launches lie on an absolute 64 x 16 grid of the half-res map (so that the flags one launch leaves per tile
// mean the same tile to the next); a launch covers the tile rows that intersect its rows [row0, row1).
__device__ __forceinline__ BlurTileArgs blur_tile_args(const crychic_ssao_constants& cb, const EdgePlane& edge, const uint16_t* in, uint16_t* out,
                                                       uint32_t W, uint32_t H, uint32_t row0, uint32_t row1)
{
    // Natural order: consecutive tiles of a row go to different XCDs.  The blur's planes are small (they live in L2 / Infinity
    // Cache whatever the placement), but its cost is very uneven -- settled (sky, open ground) tiles cost nothing, the others sit
    // in the lower half of the frame -- and stripes of tile rows per XCD left one XCD with twice the work of the others.
    uint32_t bx = blockIdx.x, by = blockIdx.y;
    by += row0 / (uint32_t)kBlurTileH;
    BlurTileArgs a;
    a.w = &cb.BlurWeights[0][0];
    a.e = edge;
    a.in = in;
    a.out = out;
    a.w2 = (int)(W / 2);
    a.h2 = (int)(H / 2);
    a.x0 = (int)bx * kBlurTileW;
    a.y0 = (int)by * kBlurTileH;
    a.row0 = (int)row0;
    a.row1 = (int)row1;
    a.borderZ = ndc_to_view(cb, 1.0f);
    a.tileIndex = by * blur_tiles_x(W) + bx;
    return a;
}

// Iteration 0 of the blur: horizontal + vertical sweep of a tile in one launch (blur_tiles.hpp blur_pair_tile).
template <bool RECORD>
__global__ __launch_bounds__(512) void blur_pair_kernel(crychic_ssao_constants cb, EdgePlane edge, const uint16_t* __restrict__ in,
                                                        uint16_t* __restrict__ out, uint32_t W, uint32_t H, uint32_t row0, uint32_t row1,
                                                        uint32_t stamp, int onesMargin, int ssaoRow0, int ssaoRow1, uint32_t frameStamp)
{
    __shared__ f4a s_nz[kBlurPairSW * kBlurPairSH];
    __shared__ float s_a[kBlurPairSW * kBlurPairSH];
    __shared__ float s_mid[kBlurTileW * kBlurPairSH];
    __shared__ uint16_t s_hmask[kBlurTileW * kBlurTileH];
    const BlurTileArgs a = blur_tile_args(cb, edge, in, out, W, H, row0, row1);
    // The tile's counter of the single-launch chain that follows (blur_replay_chain_kernel): this frame's tag, no iteration done.
    // Written here, one launch earlier, so that every counter the chain polls was written in this frame -- whatever the workspace
    // held before, it cannot read as a neighbour's progress.
    if (RECORD && threadIdx.x == 0) edge.progress[a.tileIndex] = (unsigned long long)frameStamp << 8;
    blur_pair_tile<RECORD>(BlockDev{}, a, stamp, onesMargin, ssaoRow0, ssaoRow1, s_nz, s_a, s_mid, s_hmask);
}

// A later iteration of the blur, both sweeps, replayed (blur_tiles.hpp blur_replay_tile): 8 wavefronts per tile.
__global__ __launch_bounds__(512) void blur_replay_kernel(crychic_ssao_constants cb, EdgePlane edge, const uint16_t* __restrict__ in,
                                                          uint16_t* __restrict__ out, uint32_t W, uint32_t H, uint32_t row0, uint32_t row1,
                                                          uint32_t stamp, int onesShortcut)
{
    __shared__ float s_in[kBlurPairSW * kBlurPairSH];
    __shared__ uint32_t s_mask[kBlurPairSW * kBlurPairSH];
    __shared__ float s_mid[kBlurTileW * kBlurPairSH];
    __shared__ uint32_t s_rows[kBlurMaxWaves];
    const BlurTileArgs a = blur_tile_args(cb, edge, in, out, W, H, row0, row1);
    blur_replay_tile(BlockDev{}, a, stamp, onesShortcut != 0, s_in, s_mask, s_mid, s_rows);
}

// Iterations 1 .. blurCount - 1 of the chain in ONE launch: workgroup (x, y, z) replays iteration z + 1 of tile (x, y).  What the
// per-iteration launches get from the kernel boundary -- iteration j reads what iteration j - 1 wrote, and may overwrite the plane
// iteration j - 1 read -- is a per-tile dependency here: a tile's iteration j needs its own and its eight neighbours' iteration
// j - 1 COMPLETE (their outputs are its apron; their inputs are the plane it writes).  Every tile publishes its count of completed
// iterations, tagged with the frame stamp (a stale or uninitialised word never matches), with agent-scope release; a workgroup
// polls its up to nine predecessors with agent-scope acquire before it stages.  Forward progress: workgroups are dispatched in
// increasing (z, y, x) order and a workgroup only ever waits for workgroups of the layer below, which were dispatched before it
// and wait for nothing above them.  The poll is bounded all the same: a workgroup that gives up sets EdgePlane::progress's error
// word and goes on (wrong pixels, never a hang).  Tiles the pair launch settled neither work nor publish nor are waited for.
// Same tile body as blur_replay_kernel (blur_tiles.hpp blur_replay_tile): same bits.  What it buys is the launch boundaries: two
// of the chain's four drain-and-refill gaps, and a tile's iteration j + 1 can start while other tiles are still in iteration j.
struct BlurChainPlan {
    uint16_t* plane[2];
    uint32_t row0[8], row1[8];      // half-res rows of its output iteration z + 1 owes
    uint32_t in[8];                 // plane index it reads (writes the other)
    uint32_t tileRow0;              // first tile row of the grid (absolute 64 x 16 grid)
};
__global__ __launch_bounds__(512) void blur_replay_chain_kernel(crychic_ssao_constants cb, EdgePlane edge, BlurChainPlan plan, uint32_t W, uint32_t H,
                                                                uint32_t stamp, uint32_t exitStamp, int onesShortcut)
{
    __shared__ float s_in[kBlurPairSW * kBlurPairSH];
    __shared__ uint32_t s_mask[kBlurPairSW * kBlurPairSH];
    __shared__ float s_mid[kBlurTileW * kBlurPairSH];
    __shared__ uint32_t s_rows[kBlurMaxWaves];
    const uint32_t it = blockIdx.z, tx = blockIdx.x, ty = plan.tileRow0 + blockIdx.y, ntx = blur_tiles_x(W);
    const uint32_t tile = ty * ntx + tx;
    if (exitStamp != 0u && edge.tiles[tile] == exitStamp) return;                     // settled: 65535 in both planes, for good
    const unsigned long long tag = (unsigned long long)stamp << 8;
    if (it > 0u) {
        // wait for the 3 x 3 neighbourhood's iteration `it` (count >= it), lanes 0..8 of the first wavefront one tile each
        if (threadIdx.x < 9u) {
            const int nx = (int)tx + (int)(threadIdx.x % 3u) - 1, ny = (int)ty + (int)(threadIdx.x / 3u) - 1;
            const bool inGrid = nx >= 0 && nx < (int)ntx && ny >= (int)plan.tileRow0 && ny < (int)(plan.tileRow0 + gridDim.y);
            if (inGrid) {
                const uint32_t n = (uint32_t)ny * ntx + (uint32_t)nx;
                if (!(exitStamp != 0u && edge.tiles[n] == exitStamp)) {
                    bool ok = false;
                    for (int spin = 0; spin < (1 << 16) && !ok; ++spin) {
                        // relaxed polls (a coherent load, no cache invalidation per poll: an acquire here flushed the caches under the
                        // workgroups that were doing the work -- 14x slower); ONE acquire fence after the wait, below
                        const unsigned long long v = __hip_atomic_load(edge.progress + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = (v >> 8) == (tag >> 8) && (v & 255ull) >= (unsigned long long)it;
                        if (!ok) __builtin_amdgcn_s_sleep(16);
                    }
                    if (!ok) edge.progress[(size_t)ntx * blur_tiles_y(H)] = tag | 255ull;      // the error word (one past the tiles)
                }
            }
        }
        __syncthreads();          // the staging loads below are device-coherent themselves: nothing to invalidate
    }
    const uint32_t row0 = plan.row0[it], row1 = plan.row1[it];
    const bool mine = ty * (uint32_t)kBlurTileH < row1 && (ty + 1u) * (uint32_t)kBlurTileH > row0;
    if (mine) {
        BlurTileArgs a;
        a.w = &cb.BlurWeights[0][0];
        a.e = edge;
        a.in = plan.plane[plan.in[it]];
        a.out = plan.plane[plan.in[it] ^ 1u];
        a.w2 = (int)(W / 2);
        a.h2 = (int)(H / 2);
        a.x0 = (int)tx * kBlurTileW;
        a.y0 = (int)ty * kBlurTileH;
        a.row0 = (int)row0;
        a.row1 = (int)row1;
        a.borderZ = ndc_to_view(cb, 1.0f);
        a.tileIndex = tile;
        blur_replay_tile(BlockDevT<true>{}, a, 0u, onesShortcut != 0, s_in, s_mask, s_mid, s_rows);
    }
    // The tile's outputs were device-coherent stores: once every wavefront's stores are acknowledged (workgroup-scope release:
    // s_waitcnt) they are visible to every XCD, and the count may be published -- no L2 write-back (an agent-scope release fence
    // per workgroup writes back the XCD's whole L2: measured 390 us for the launch)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(edge.progress + tile, tag | (unsigned long long)(it + 1u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- deferred lighting -----------------------------------------------------------------------------------------
// Shaders/DeferredShading.hlsl:23-101 as a full-screen pass over rows [row0, row1), masked by depth < 1.
// MIPS (the cube map holds a mip chain, P.cubeLevels > 1): a wavefront covers 32 x 2 pixels instead of 64 x 1, so that every 2 x 2
// quad of the frame lies inside one wavefront -- the x neighbour is lane ^ 1, the y neighbour lane ^ 32 -- and the level of detail of
// the reflection lookup comes from the neighbours' reflection vectors without a second pass (light_core.hpp "TextureCube.Sample
// with the mip chain bound").
template <bool MIPS>
__device__ __forceinline__ void light_tile_pixel(uint32_t bx, uint32_t by, uint32_t row0, uint32_t& x, uint32_t& y)
{
    if (MIPS) {
        const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63u;
        x = bx * 64u + (wave & 1u) * 32u + (lane & 31u);
        y = row0 + by * 4u + (wave >> 1) * 2u + (lane >> 5);
    } else {
        x = bx * 64u + (threadIdx.x & 63u);
        y = row0 + by * 4u + (threadIdx.x >> 6);
    }
}
// The level of detail of this lane's reflection lookup; every lane of the wavefront calls it (converged).  `r` is the lane's
// reflection vector when `covered`.  A neighbour the pass does not shade contributes a zero derivative (the oracle's definition).
__device__ __forceinline__ float quad_reflection_lod(const LightParams& P, bool covered, f3 r, uint32_t x, uint32_t y)
{
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(covered);
    const uint32_t lane = threadIdx.x & 63u;
    const f3 nx{ __shfl_xor(r.x, 1), __shfl_xor(r.y, 1), __shfl_xor(r.z, 1) };
    const f3 ny{ __shfl_xor(r.x, 32), __shfl_xor(r.y, 32), __shfl_xor(r.z, 32) };
    const bool hasX = (mask >> (lane ^ 1u)) & 1ull, hasY = (mask >> (lane ^ 32u)) & 1ull;
    f3 ddx{ 0.0f, 0.0f, 0.0f }, ddy{ 0.0f, 0.0f, 0.0f };
    if (hasX) ddx = (x & 1u) ? f3{ r.x - nx.x, r.y - nx.y, r.z - nx.z } : f3{ nx.x - r.x, nx.y - r.y, nx.z - r.z };
    if (hasY) ddy = (y & 1u) ? f3{ r.x - ny.x, r.y - ny.y, r.z - ny.z } : f3{ ny.x - r.x, ny.y - r.y, ny.z - r.z };
    return cube_lod(P.cubeDim, P.cubeLevels, r, ddx, ddy);
}

template <bool ZERO_RADIUS, bool FIX, bool MIPS = false>
__global__ __launch_bounds__(256) void light_kernel(LightParams P, const f4a* __restrict__ g0,
                                                    const f4a* __restrict__ g1, const f4a* __restrict__ g2,
                                                    const uint32_t* __restrict__ depth,
                                                    const uint16_t* __restrict__ ambient,
                                                    const uint32_t* __restrict__ cube, uint32_t* __restrict__ out,
                                                    f4a* __restrict__ radiance, uint32_t row0, uint32_t row1)
{
    uint32_t bx, by;
    tile_origin<0>(bx, by);
    uint32_t x, y;
    light_tile_pixel<MIPS>(bx, by, row0, x, y);
    if (MIPS) {
        const bool in = x < P.W && y < row1;
        const uint32_t idx = in ? y * P.W + x : 0u;
        const bool covered = in && (depth[idx] & 0x00FFFFFFu) < 0x00FFFFFFu;
        f4a G0{ 0, 0, 0, 0 }, G2{ 0, 0, 0, 0 };
        f3 r{ 0.0f, 0.0f, 0.0f };
        if (covered) { G0 = g0[idx]; G2 = g2[idx]; r = reflection_dir(P, G0, G2); }
        const float lod = quad_reflection_lod(P, covered, r, x, y);
        if (!in) return;
        f4 lit;
        if (covered) lit = light_pixel<ZERO_RADIUS, NoPointLights, FIX, CubeChain>(P, G0, g1[idx], G2, ambient, cube, NoPointLights(), CubeChain{ lod, cube_chain_flat(lod) });
        else if (P.flags & CRYCHIC_LIGHT_SKY) lit = sky_pixel_chain(P, cube, x, y);
        else lit = f4{ 0.690196097f, 0.768627524f, 0.870588303f, 1.0f };
        if (radiance) radiance[idx] = f4a{ lit.x, lit.y, lit.z, lit.w };
        out[idx] = pack_rgba8(lit);
        return;
    }
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
template <bool ZERO_RADIUS, bool MIPS = false>
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
    uint32_t x, y;
    light_tile_pixel<MIPS>(bx, by, row0, x, y);             // MIPS: 32 x 2 pixels per wavefront (quads inside a wavefront)
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
    float lod = 0.0f;
    f4a G2{ 0, 0, 0, 0 };
    if (MIPS) {                                              // every lane, converged: the quad neighbours' reflection vectors
        f3 r{ 0.0f, 0.0f, 0.0f };
        if (covered) { G2 = g2[idx]; r = reflection_dir(P, G0, G2); }
        lod = quad_reflection_lod(P, covered, r, x, y);
    }
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
        if (MIPS) lit = light_pixel<ZERO_RADIUS, decltype(culled), true, CubeChain>(P, G0, g1[idx], G2, ambient, cube, culled, CubeChain{ lod, cube_chain_flat(lod) });
        else lit = light_pixel<ZERO_RADIUS, decltype(culled), true>(P, G0, g1[idx], g2[idx], ambient, cube, culled);
    } else if (P.flags & CRYCHIC_LIGHT_SKY) {
        lit = MIPS ? sky_pixel_chain(P, cube, x, y) : sky_pixel(P, cube, x, y);
    } else {
        lit = f4{ 0.690196097f, 0.768627524f, 0.870588303f, 1.0f };
    }
    if (radiance) radiance[idx] = f4a{ lit.x, lit.y, lit.z, lit.w };
    out[idx] = pack_rgba8(lit);
}

// ---- launchers ---------------------------------------------------------------------------------------------------
// Margin (depth texel rows) of a row-limited depth pass around the rows of the call (ssao_core.hpp depth_pass_cell_rows).  Any value
// gives the same pixels (tests: margins 0, 8, 40 with everything unvisited poisoned); CRYCHIC_DEPTH_MARGIN is the tuning knob the
// strips rehearsal sweeps (tools/strips_rehearsal.sh), unset = the default rule (H / 11, at least 64).
static int depth_margin()
{
    static const int m = [] { const char* e = getenv("CRYCHIC_DEPTH_MARGIN"); return e ? atoi(e) : -1; }();
    return m;
}
static inline dim3 grid_for(uint32_t width, uint32_t rows, uint32_t rows_per_block = 4u)
{
    return dim3((width + 63u) / 64u, (rows + rows_per_block - 1u) / rows_per_block, 1);
}

hipError_t launch_depth_pairs(const crychic_ssao_constants& cb, const uint32_t* depth, void* edge_base, uint32_t W, uint32_t H, uint32_t stamp,
                              bool writePairs, uint32_t row0, uint32_t rows, hipStream_t stream)
{
    const EdgePlane e = edge_plane_carve(edge_base, W, H);
    uint32_t c0, cn;
    depth_pass_cell_rows(H, row0, rows, &c0, &cn, depth_margin());
    if (cn == 0) return hipSuccess;
    // a wavefront per 128 entry columns x 8 entry rows (= one row of 16 cells of the nearest-depth map), four wavefronts per workgroup
    const uint32_t segs = (depth_pairs_pitch(W) / 2u + 63u) / 64u;
    const dim3 grid((segs + 3u) / 4u, cn, 1);
#define CRY_LAUNCH_DP(K) hipLaunchKernelGGL(K, grid, dim3(256), 0, stream, depth, (f4a*)const_cast<void*>(e.pairs), e.geo, e.zcull, stamp, W, H, \
                                            ssao_cull_params(cb), c0)
    if (writePairs) CRY_LAUNCH_DP(depth_pairs_kernel<true>); else CRY_LAUNCH_DP(depth_pairs_kernel<false>);
#undef CRY_LAUNCH_DP
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
    // super-tiles of SX x SY workgroup tiles (64 x 4 half-res pixels each) per XCD; the grid is padded to whole super-tiles.
    // 2 x 16 tiles = 128 x 64 half-res pixels.  Round 2 ran 4 x 32 (every tap gathered: L2 locality of the gathers decided); with
    // the tap culling only one tap in seven gathers and the balance between XCDs -- sky is cheap, near ground is not -- weighs
    // more: 4K 81.7 -> 77.4 us for the pass, 1080p 33.4 -> 31.7, covered camera equal, 8K +1.5 % (natural order: 77.5 / 31.4 /
    // +2 % / +3 %; profiles/r03_supertile_sweep.txt)
    uint32_t SX = 2u, SY = 16u;
    SX = SX > grid.x ? grid.x : SX;
    SY = SY > grid.y ? grid.y : SY;
    grid.x = (grid.x + SX - 1u) / SX * SX;
    grid.y = (grid.y + SY - 1u) / SY * SY;
    const int sparse = ssao_projtex_is_sparse(cb) ? 1 : 0;
    SkyReach sky = ssao_sky_reach(cb, W, H);
    const bool maps = edge_base && stamp != 0u;        // stamp 0: the caller did not run the depth pass
    if (!maps) sky.enabled = 0;
    const int cull = (maps && ssao_cull_params(cb).enabled) ? 1 : 0;
    // what launch_depth_pairs prepared for these rows: everything, or footprint rows j0 with 8 c0 <= j0 + 2 < 8 (c0 + cn)
    uint32_t c0, cn;
    depth_pass_cell_rows(H, row0, rows, &c0, &cn, depth_margin());
    const bool limitedRows = maps && (c0 > 0u || c0 + cn < zmin_map_rows(H));
    const bool limited = limitedRows && use_pairs;
    const PrepRows prep{ 8 * (int)c0 - 2, 8u * cn };
    if (limitedRows) {                  // geometry-map cells are known for the texel rows the pass visited (8 entry rows per cell row)
        sky.y0 = 8 * (int)c0 - 2 < 0 ? 0 : 8 * (int)c0 - 2;
        sky.y1 = 8 * (int)(c0 + cn) - 2 > (int)H ? (int)H : 8 * (int)(c0 + cn) - 2;
    }
#define CRY_LAUNCH_SSAO(K) hipLaunchKernelGGL(K, grid, dim3(256), 0, stream, cb, (const u2*)normal, depth, (const uint32_t*)randvec, ambient, e, W, H, row0, row0 + rows, SX, SY, sparse, sky, stamp, cull, prep, half_res_scale(W, H))
    if (!emit_ao) CRY_LAUNCH_SSAO((ssao_kernel<false, false, false, false>));
    else if (limited) CRY_LAUNCH_SSAO((ssao_kernel<true, true, true, true>));
    else if (limitedRows) CRY_LAUNCH_SSAO((ssao_kernel<true, false, true, true>));
    else if (use_pairs && maps) CRY_LAUNCH_SSAO((ssao_kernel<true, true, true, false>));
    else if (use_pairs) CRY_LAUNCH_SSAO((ssao_kernel<true, true, false, false>));
    else if (maps) CRY_LAUNCH_SSAO((ssao_kernel<true, false, true, false>));
    else CRY_LAUNCH_SSAO((ssao_kernel<true, false, false, false>));
#undef CRY_LAUNCH_SSAO
    return hipGetLastError();
}

hipError_t launch_blur(const crychic_ssao_constants& cb, const void* edge_base, const uint16_t* in, uint16_t* out,
                       uint32_t W, uint32_t H, bool horizontal, uint32_t row0, uint32_t rows, hipStream_t stream)
{
    if (rows == 0) return hipSuccess;
    const EdgePlane e = edge_plane_carve(const_cast<void*>(edge_base), W, H);
    const dim3 grid = grid_for(W / 2, rows, 16u);
    if (horizontal) hipLaunchKernelGGL(blur_kernel<true>, grid, dim3(256), 0, stream, cb, e, in, out, W, H, row0, row0 + rows);
    else hipLaunchKernelGGL(blur_kernel<false>, grid, dim3(256), 0, stream, cb, e, in, out, W, H, row0, row0 + rows);
    return hipGetLastError();
}

// tile rows of the absolute 64 x 16 grid that intersect half-res rows [row0, row0 + rows)
static inline dim3 blur_tile_grid(uint32_t W, uint32_t row0, uint32_t rows)
{
    const uint32_t t0 = row0 / (uint32_t)kBlurTileH, t1 = (row0 + rows - 1u) / (uint32_t)kBlurTileH;
    return dim3(blur_tiles_x(W), t1 - t0 + 1u, 1);
}

hipError_t launch_blur_pair(const crychic_ssao_constants& cb, const void* edge_base, const uint16_t* in, uint16_t* out, uint32_t W, uint32_t H,
                            uint32_t row0, uint32_t rows, bool record, uint32_t stamp, int onesMargin, uint32_t ssaoRow0, uint32_t ssaoRows,
                            hipStream_t stream)
{
    if (rows == 0) return hipSuccess;
    const EdgePlane e = edge_plane_carve(const_cast<void*>(edge_base), W, H);
    // the unoccluded-tile exit needs finite positive weights (x * rcp(x) of a finite positive total) and the SSAO pass's map
    const uint32_t frameStamp = stamp;
    if (!blur_weights_positive(cb)) stamp = 0u;
    const dim3 grid = blur_tile_grid(W, row0, rows);
#define CRY_LAUNCH_PAIR(K) hipLaunchKernelGGL(K, grid, dim3(512), 0, stream, cb, e, in, out, W, H, row0, row0 + rows, stamp, onesMargin, \
                                              (int)ssaoRow0, (int)(ssaoRow0 + ssaoRows), frameStamp)
    if (record) CRY_LAUNCH_PAIR(blur_pair_kernel<true>); else CRY_LAUNCH_PAIR(blur_pair_kernel<false>);
#undef CRY_LAUNCH_PAIR
    return hipGetLastError();
}

hipError_t launch_blur_replay(const crychic_ssao_constants& cb, const void* edge_base, const uint16_t* in, uint16_t* out, uint32_t W, uint32_t H,
                              uint32_t row0, uint32_t rows, uint32_t stamp, hipStream_t stream)
{
    if (rows == 0) return hipSuccess;
    const EdgePlane e = edge_plane_carve(const_cast<void*>(edge_base), W, H);
    const int ones = blur_weights_positive(cb) ? 1 : 0;
    if (!ones) stamp = 0u;             // the pair launch settled no tile either
    hipLaunchKernelGGL(blur_replay_kernel, blur_tile_grid(W, row0, rows), dim3(512), 0, stream, cb, e, in, out, W, H, row0, row0 + rows, stamp, ones);
    return hipGetLastError();
}

hipError_t launch_blur_replay_chain(const crychic_ssao_constants& cb, const void* edge_base, uint16_t* plane0, uint16_t* plane1, uint32_t W, uint32_t H,
                                    int blurCount, uint32_t row0, uint32_t rows, uint32_t stamp, hipStream_t stream)
{
    const int n = blurCount - 1;                       // replay iterations 1 .. blurCount - 1
    if (n <= 0 || rows == 0) return hipSuccess;
    if (n > 8) return hipErrorInvalidValue;
    const EdgePlane e = edge_plane_carve(const_cast<void*>(edge_base), W, H);
    BlurChainPlan plan;
    plan.plane[0] = plane0;
    plan.plane[1] = plane1;
    uint32_t lo = 0xFFFFFFFFu, hi = 0u;
    for (int i = 1; i <= n; ++i) {
        const BlurStep st = blur_chain_step(blurCount, row0, rows, H / 2u, i);
        plan.row0[i - 1] = st.row0;
        plan.row1[i - 1] = st.row0 + st.rows;
        plan.in[i - 1] = (uint32_t)st.in;
        if (st.rows) { lo = st.row0 < lo ? st.row0 : lo; hi = st.row0 + st.rows > hi ? st.row0 + st.rows : hi; }
    }
    for (int i = n; i < 8; ++i) { plan.row0[i] = plan.row1[i] = 0u; plan.in[i] = 0u; }
    if (hi <= lo) return hipSuccess;
    const dim3 g = blur_tile_grid(W, lo, hi - lo);
    plan.tileRow0 = lo / (uint32_t)kBlurTileH;
    const int ones = blur_weights_positive(cb) ? 1 : 0;
    hipLaunchKernelGGL(blur_replay_chain_kernel, dim3(g.x, g.y, (unsigned)n), dim3(512), 0, stream, cb, e, plan, W, H, stamp, ones ? stamp : 0u, ones);
    return hipGetLastError();
}

hipError_t launch_light(const LightParams& P, const float* g0, const float* g1, const float* g2,
                        const uint32_t* depth, const uint16_t* ambient, const uint8_t* cube, uint8_t* out,
                        float* radiance, uint32_t row0, uint32_t rows, hipStream_t stream)
{
    if (rows == 0) return hipSuccess;
    const dim3 grid = grid_for(P.W, rows);
    const bool mips = P.cubeLevels > 1u;          // the chain: quads inside wavefronts (light_tile_pixel), so the rows must start a quad
    if (mips && (row0 & 1u)) return hipErrorInvalidValue;
    if (P.numPointLights) {
#define CRY_LAUNCH_POINTS(K) hipLaunchKernelGGL(K, grid, dim3(256), 0, stream, P, (const f4a*)g0, (const f4a*)g1, (const f4a*)g2, depth, ambient, \
                                                (const uint32_t*)cube, (uint32_t*)out, (f4a*)radiance, row0, row0 + rows)
        if (P.pcfSearchRadius == 0.0f) { if (mips) CRY_LAUNCH_POINTS((light_points_kernel<true, true>)); else CRY_LAUNCH_POINTS((light_points_kernel<true, false>)); }
        else { if (mips) CRY_LAUNCH_POINTS((light_points_kernel<false, true>)); else CRY_LAUNCH_POINTS((light_points_kernel<false, false>)); }
#undef CRY_LAUNCH_POINTS
        return hipGetLastError();
    }
    const bool fix = (P.flags & (CRYCHIC_FIX_Q1 | CRYCHIC_FIX_Q3 | CRYCHIC_FIX_Q4)) != 0;
#define CRY_LAUNCH_LIGHT(K) hipLaunchKernelGGL(K, grid, dim3(256), 0, stream, P, (const f4a*)g0, (const f4a*)g1, (const f4a*)g2, depth, ambient, \
                                               (const uint32_t*)cube, (uint32_t*)out, (f4a*)radiance, row0, row0 + rows)
    if (mips) {                                   // FIX compiled in: the chain is not the benchmark's instantiation
        if (P.pcfSearchRadius == 0.0f) CRY_LAUNCH_LIGHT((light_kernel<true, true, true>)); else CRY_LAUNCH_LIGHT((light_kernel<false, true, true>));
    }
    else if (P.pcfSearchRadius == 0.0f) { if (fix) CRY_LAUNCH_LIGHT((light_kernel<true, true>)); else CRY_LAUNCH_LIGHT((light_kernel<true, false>)); }
    else { if (fix) CRY_LAUNCH_LIGHT((light_kernel<false, true>)); else CRY_LAUNCH_LIGHT((light_kernel<false, false>)); }
#undef CRY_LAUNCH_LIGHT
    return hipGetLastError();
}

}  // namespace cry
