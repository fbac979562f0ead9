// ssao_core.hpp -- per-pixel bodies of the SSAO and bilateral-blur kernels (Shaders/Ssao.hlsl:117-199,
// Shaders/SsaoBlur.hlsl:85-146).  Host+device inline so the same text is checked on the CPU against the
// oracle (tests/hostsim) before it ever runs on a GPU.
#pragma once
#include "devmath.hpp"
#include "crychic_hip.h"

namespace cry {

// ---- edge workspace ------------------------------------------------------------------------------------
// Everything a blur tap needs besides the ambient value is a function of the half-res pixel only
// (SsaoBlur.hlsl:109-111,121-123): the point-sampled normal and the linearised bilinear depth.  The SSAO
// kernel already computes both for its own pixel, so it stores them once per frame:
//   nrm  [h2][w2]  8 B  (the fp16 texel (2x+1, 2y+1) verbatim)   -- plus per-direction tap masks, below
//   vz   [h2][w2]  4 B  (view-space depth, fp32)
//   gcol [h2]      8 B  normal texel (0, 2y+1): CLAMP target of every horizontal tap with x + i < 0
//   grow [w2]      8 B  normal texel (2x+1, 0): CLAMP target of every vertical tap with y + i < 0
struct EdgePlane {
    u2* nrm;
    float* vz;
    u2* gcol;
    u2* grow;
    // The accept/reject decision of every blur tap (SsaoBlur.hlsl:131-132) depends on geometry only, so it is the
    // same in all blurCount iterations.  The first sweep of each direction records the 11 decisions of a pixel as a
    // bit mask; later sweeps of that direction replay them (blur_pixel_replay) -- the totalWeight follows from the mask.
    uint32_t* masks;   // horizontal decisions | vertical decisions << 16
    const void* pairs; // DepthPairs plane (below), depth_pairs_bytes(W, H)
    uint32_t* geo;     // coarse geometry map (below "sky shortcut"), geo_map_bytes(W, H)
    uint32_t* ones;    // unoccluded-wavefront map (below "unoccluded tiles"), ones_map_bytes(W, H)
    float* zcull;      // nearest-depth map per 9 x 9 texel cell (below "tap culling"): what a tap looks up, zmin_map_bytes(W, H)
    uint32_t* tiles;   // unoccluded-tile flags of the blur launches (blur_tiles.hpp), blur_tile_map_bytes(W, H)
    unsigned long long* progress;      // per tile: (frame stamp << 8) | replay iterations completed (kernels.hip blur_replay_chain_kernel), 2 x blur_tile_map_bytes
};

// ---- depth pairs -----------------------------------------------------------------------------------------------------
// The 14 + 1 bilinear depth fetches of a pixel (gsamDepthMap: linear, BORDER 1.0) are the SSAO pass's gather.  On the raw
// D24 plane a footprint costs two 8-byte loads in two rows, four integer -> float decodes and four BORDER selects: 40 of the
// ~100 VALU instructions of a tap.  depth_pairs_kernel therefore re-lays the plane once per frame as decoded floats in
// entries {d(x, y), d(x, y+1)} with a guard band that holds the BORDER value: entry (x, y) for x in [-2, W+1], y in [-2, H].
// A footprint with top-left texel (i0, j0), indices clamped to [-2, W] x [-2, H] (anything further out addresses only BORDER
// texels either way), is then ONE 16-byte load of entries (i0, j0) and (i0+1, j0) = {t00, t01, t10, t11}, ready to filter.
// Decoding is the same exact d24_to_float, so every output bit is unchanged.
CRY_HD uint32_t depth_pairs_pitch(uint32_t W) { return W + 4u; }                       // entries per row
CRY_HD size_t depth_pairs_bytes(uint32_t W, uint32_t H) { return (size_t)(W + 4u) * (H + 3u) * 8u; }
CRY_HD size_t edge_plane_pairs_offset(uint32_t W, uint32_t H)
{
    size_t w2 = W / 2, h2 = H / 2;
    return (w2 * h2 * 16 + (w2 + h2) * 8 + 15) & ~(size_t)15;
}
// Coarse geometry map: one word per cell of 128 x 32 depth texels, holding the frame stamp of the last frame in which the
// cell contained a texel below the clear depth.  Cell cx covers texel columns [128 cx - 2, 128 cx + 126) -- the footprint of
// one wave of depth_pairs_kernel -- and cell cy rows [32 cy, 32 cy + 32).
CRY_HD uint32_t geo_map_cols(uint32_t W) { return (W + 2u) / 128u + 1u; }
CRY_HD uint32_t geo_map_rows(uint32_t H) { return (H + 31u) / 32u; }
CRY_HD size_t geo_map_bytes(uint32_t W, uint32_t H) { return (size_t)geo_map_cols(W) * geo_map_rows(H) * 4u; }
CRY_HD size_t edge_plane_geo_offset(uint32_t W, uint32_t H) { return (edge_plane_pairs_offset(W, H) + depth_pairs_bytes(W, H) + 15) & ~(size_t)15; }
// Unoccluded-wavefront map: one word per 64 half-res pixels of a row (= one wavefront of the SSAO pass), holding the frame
// stamp when all 64 ambient values the wavefront wrote are 65535.
CRY_HD uint32_t ones_map_cols(uint32_t W) { return (W / 2u + 63u) / 64u; }
CRY_HD size_t ones_map_bytes(uint32_t W, uint32_t H) { return (size_t)ones_map_cols(W) * (H / 2u) * 4u; }
CRY_HD size_t edge_plane_ones_offset(uint32_t W, uint32_t H) { return (edge_plane_geo_offset(W, H) + geo_map_bytes(W, H) + 15) & ~(size_t)15; }
// Coarse nearest-depth map ("tap culling" below): one float per cell of 9 x 9 texels of the padded depth plane, cells 8 texels
// apart (padded texel (ex, ey) = texel (ex - 2, ey - 2): the coordinates of the pairs plane, BORDER band included): cell (cx, cy)
// covers padded texels [8 cx, 8 cx + 8] x [8 cy, 8 cy + 8], so the 2 x 2 footprint whose top-left padded texel is (ex, ey) lies
// inside cell (ex >> 3, ey >> 3).
CRY_HD uint32_t zmin_map_cols(uint32_t W) { return (W + 4u + 7u) / 8u; }
CRY_HD uint32_t zmin_map_rows(uint32_t H) { return (H + 4u + 7u) / 8u; }
CRY_HD size_t zmin_map_bytes(uint32_t W, uint32_t H) { return (size_t)zmin_map_cols(W) * zmin_map_rows(H) * 4u; }
CRY_HD size_t edge_plane_zcull_offset(uint32_t W, uint32_t H) { return (edge_plane_ones_offset(W, H) + ones_map_bytes(W, H) + 15) & ~(size_t)15; }
// Unoccluded-tile flags of the blur launches: one word per 64 x 16 half-res tile of an absolute grid (blur_tiles.hpp).
CRY_HD uint32_t blur_tiles_x(uint32_t W) { return (W / 2u + 63u) / 64u; }
CRY_HD uint32_t blur_tiles_y(uint32_t H) { return (H / 2u + 15u) / 16u; }
CRY_HD size_t blur_tile_map_bytes(uint32_t W, uint32_t H) { return (size_t)blur_tiles_x(W) * blur_tiles_y(H) * 4u; }
CRY_HD size_t edge_plane_tiles_offset(uint32_t W, uint32_t H) { return (edge_plane_zcull_offset(W, H) + zmin_map_bytes(W, H) + 15) & ~(size_t)15; }
CRY_HD size_t edge_plane_progress_offset(uint32_t W, uint32_t H) { return (edge_plane_tiles_offset(W, H) + blur_tile_map_bytes(W, H) + 15) & ~(size_t)15; }
CRY_HD size_t edge_plane_bytes(uint32_t W, uint32_t H) { return edge_plane_progress_offset(W, H) + 2u * blur_tile_map_bytes(W, H) + 16u; }      // + the chain's error word
// Entries (x, y) and (x + 1, y) of the pairs plane from the D24 plane; x even (so the two texels of a row are one 8-byte load).
CRY_HD f4a depth_pairs_entry2(const uint32_t* __restrict__ depth, uint32_t W, uint32_t H, int x, int y)
{
    const bool inx = (uint32_t)x < W, r0 = (uint32_t)y < H, r1 = (uint32_t)(y + 1) < H;     // W even, x even: x + 1 < W as well
    const uint32_t cx = inx ? (uint32_t)x : 0u;
    const RawPair border{ 0x00FFFFFFu, 0x00FFFFFFu };
    const RawPair a = (inx && r0) ? load_at<RawPair>(depth, (mul24((uint32_t)y, W) + cx) * 4u) : border;
    const RawPair b = (inx && r1) ? load_at<RawPair>(depth, (mul24((uint32_t)(y + 1), W) + cx) * 4u) : border;
    return f4a{ d24_to_float(a.lo), d24_to_float(b.lo), d24_to_float(a.hi), d24_to_float(b.hi) };
}
CRY_HD EdgePlane edge_plane_carve(void* base, uint32_t W, uint32_t H)
{
    size_t w2 = W / 2, h2 = H / 2, n = w2 * h2;
    char* b = (char*)base;
    EdgePlane e;
    e.nrm = (u2*)b;                       // n * 8
    e.vz = (float*)(b + n * 8);           // n * 4
    e.masks = (uint32_t*)(b + n * 12);    // n * 4
    e.gcol = (u2*)(b + n * 16);           // h2 * 8
    e.grow = e.gcol + h2;                 // w2 * 8
    e.pairs = b + edge_plane_pairs_offset(W, H);
    e.geo = (uint32_t*)(b + edge_plane_geo_offset(W, H));
    e.ones = (uint32_t*)(b + edge_plane_ones_offset(W, H));
    e.zcull = (float*)(b + edge_plane_zcull_offset(W, H));
    e.tiles = (uint32_t*)(b + edge_plane_tiles_offset(W, H));
    e.progress = (unsigned long long*)(b + edge_plane_progress_offset(W, H));
    return e;
}

CRY_HD f3 unpack_normal(u2 t)
{
    return f3{ half_to_float((uint16_t)(t.x & 0xFFFFu)), half_to_float((uint16_t)(t.x >> 16)),
               half_to_float((uint16_t)(t.y & 0xFFFFu)) };
}

// Ssao.hlsl:110-115; HLSL gProj[r][c] is mem[4c + r] in the transposed cbuffer layout.
CRY_HD float ndc_to_view(const crychic_ssao_constants& cb, float z_ndc)
{
    return divf(cb.Proj[4 * 2 + 3], z_ndc - cb.Proj[4 * 2 + 2]);
}

// gsamDepthMap: linear filter, BORDER (1.0) addressing  (CRYCHIC.cpp:1057-1066)
CRY_HD float depth_texel(const uint32_t* __restrict__ depth, uint32_t W, uint32_t H, int x, int y)
{
    // Always fetch (from the clamped address) and select afterwards: the four texel loads of a bilinear footprint
    // issue back to back instead of sitting in four dependent exec-masked branches.
    const bool in = ((uint32_t)x < W) && ((uint32_t)y < H);
    const uint32_t cx = (uint32_t)clampi(x, 0, (int)W - 1), cy = (uint32_t)clampi(y, 0, (int)H - 1);
    const uint32_t t = load_at<uint32_t>(depth, (mul24(cy, W) + cx) * 4u);
    return d24_to_float(in ? t : 0x00FFFFFFu);
}
CRY_HD float depth_linear_border(const uint32_t* __restrict__ depth, uint32_t W, uint32_t H, float u, float v)
{
    const Bilin b = bilinear_setup(u, v, W, H);
    // rows j0, j0+1: one 8-byte load each (texels i0, i0+1), fetched from clamped addresses, border selected after
    const uint32_t r0 = (uint32_t)clampi(b.j0, 0, (int)H - 1), r1 = (uint32_t)clampi(b.j0 + 1, 0, (int)H - 1);
    const TexelPair p0 = pair_at(depth, r0, W, b.i0);
    const TexelPair p1 = pair_at(depth, r1, W, b.i0);
    const bool xa = (uint32_t)b.i0 < W, xb = (uint32_t)(b.i0 + 1) < W;
    const bool y0 = (uint32_t)b.j0 < H, y1 = (uint32_t)(b.j0 + 1) < H;
    // the BORDER colour 1.0 is exactly D24 0xFFFFFF: select on the integer texel, then decode unconditionally
    const float t00 = d24_to_float((xa && y0) ? p0.a : 0x00FFFFFFu);
    const float t10 = d24_to_float((xb && y0) ? p0.b : 0x00FFFFFFu);
    const float t01 = d24_to_float((xa && y1) ? p1.a : 0x00FFFFFFu);
    const float t11 = d24_to_float((xb && y1) ? p1.b : 0x00FFFFFFu);
    return bilerp(t00, t10, t01, t11, b.fx, b.fy);
}
// The two depth sources of the SSAO pass.  footprint(): the four decoded texels of the bilinear footprint whose top-left
// texel is (i0, j0), both already clamped to [-2, dim], BORDER texels as 1.0.
struct DepthD24 {
    const uint32_t* __restrict__ plane; uint32_t W, H;
    CRY_HD void footprint(int i0, int j0, float& t00, float& t10, float& t01, float& t11) const
    {
        // rows j0, j0+1: one 8-byte load each (texels i0, i0+1), fetched from clamped addresses, border selected after;
        // the BORDER colour 1.0 is exactly D24 0xFFFFFF: select on the integer texel, then decode unconditionally
        const uint32_t r0 = (uint32_t)clampi(j0, 0, (int)H - 1), r1 = (uint32_t)clampi(j0 + 1, 0, (int)H - 1);
        const TexelPair a0 = pair_at(plane, r0, W, i0), a1 = pair_at(plane, r1, W, i0);
        const bool xa = (uint32_t)i0 < W, xb = (uint32_t)(i0 + 1) < W, y0 = (uint32_t)j0 < H, y1 = (uint32_t)(j0 + 1) < H;
        t00 = d24_to_float((xa && y0) ? a0.a : 0x00FFFFFFu); t10 = d24_to_float((xb && y0) ? a0.b : 0x00FFFFFFu);
        t01 = d24_to_float((xa && y1) ? a1.a : 0x00FFFFFFu); t11 = d24_to_float((xb && y1) ? a1.b : 0x00FFFFFFu);
    }
};
struct DepthPairs {
    const void* __restrict__ base; uint32_t pitch;      // depth_pairs_pitch(W)
    CRY_HD void footprint(int i0, int j0, float& t00, float& t10, float& t01, float& t11) const
    {
        const f4a v = load_at<f4a>(base, (mul24((uint32_t)(j0 + 2), pitch) + (uint32_t)(i0 + 2)) * 8u);
        t00 = v.x; t01 = v.y; t10 = v.z; t11 = v.w;
    }
};
// A tap of the SSAO loop: the footprint of a lane whose tap was culled is never looked at, so such a lane fetches the plane's
// first footprint instead of its own (every culled lane of the wavefront the same line: the gather it is spared stays spared)
// and the loads of a tap pair need no branch around them.
template <class Depth>
CRY_HD void depth_tap(const Depth& d, bool culled, int i0, int j0, float& t00, float& t10, float& t01, float& t11)
{
    d.footprint(culled ? -2 : i0, culled ? -2 : j0, t00, t10, t01, t11);
}
// Row-limited pairs plane (a rank's strip of a multi-GPU frame): the depth pass prepared the pairs plane and the
// coarse maps only for footprints whose top row j0 lies in [j0lo, j0lo + nj) -- the strip's own rows and a margin.  The taps of
// the strip's pixels may land anywhere in the frame; one that leaves the prepared rows gathers from the raw D24 plane instead
// (and is never culled: ZminMapRows).  Same decode, same filter: every output bit is unchanged whatever the margin is, which is
// therefore a pure performance knob.
struct DepthPairsRows {
    DepthPairs pairs; DepthD24 raw; int j0lo; uint32_t nj;
    CRY_HD bool prepared(int j0) const { return (uint32_t)(j0 - j0lo) < nj; }
};
CRY_HD void depth_tap(const DepthPairsRows& d, bool culled, int i0, int j0, float& t00, float& t10, float& t01, float& t11)
{
    const bool ok = d.prepared(j0), dummy = culled | !ok;
    d.pairs.footprint(dummy ? -2 : i0, dummy ? d.j0lo : j0, t00, t10, t01, t11);        // (-2, j0lo): a prepared, always valid entry
    const bool slow = !ok & !culled;
#if defined(__HIP_DEVICE_COMPILE__)
    if (__builtin_amdgcn_ballot_w64(slow) == 0) return;       // the whole wavefront stayed inside the prepared rows (nearly always)
#endif
    if (slow) d.raw.footprint(i0, j0, t00, t10, t01, t11);
}

// The same sampler at the centre of half-res pixel (xi, yi) (even W, H): texels 2xi..2xi+1 x 2yi..2yi+1 with
// weights 1/2; a pixel outside the half-res map only ever addresses border texels.
CRY_HD void depth_at_half_pixel(const uint32_t* __restrict__ depth, uint32_t W, uint32_t H, int xi, int yi, float& t00, float& t10, float& t01,
                                float& t11)
{
    const bool in = ((uint32_t)(2 * xi) < W) & ((uint32_t)(2 * yi) < H);   // even sizes: all four in or all four out
    const uint32_t cx = (uint32_t)clampi(2 * xi, 0, (int)W - 2), cy = (uint32_t)clampi(2 * yi, 0, (int)H - 2);
    const uint32_t t0 = mul24(cy, W) + cx;
    const RawPair p0 = load_pair(depth, t0);
    const RawPair p1 = load_pair(depth, t0 + W);
    t00 = d24_to_float(in ? p0.lo : 0x00FFFFFFu); t10 = d24_to_float(in ? p0.hi : 0x00FFFFFFu);
    t01 = d24_to_float(in ? p1.lo : 0x00FFFFFFu); t11 = d24_to_float(in ? p1.hi : 0x00FFFFFFu);
}
CRY_HD float depth_at_half_pixel(const uint32_t* __restrict__ depth, uint32_t W, uint32_t H, int xi, int yi)
{
    float t00, t10, t01, t11;
    depth_at_half_pixel(depth, W, H, xi, yi, t00, t10, t01, t11);
    return bilerp(t00, t10, t01, t11, 0.5f, 0.5f);
}

// gsamPointClamp on the full-res normal map at the centre of half-res pixel (xi, yi): texel
// clamp(2xi+1), clamp(2yi+1) (the texel-edge tie is resolved as floor).
CRY_HD u2 normal_texel_bits(const u2* __restrict__ normal, uint32_t W, uint32_t H, int xi, int yi)
{
    int tx = clampi(2 * xi + 1, 0, (int)W - 1);
    int ty = clampi(2 * yi + 1, 0, (int)H - 1);
    return load_at<u2>(normal, (mul24((uint32_t)ty, W) + (uint32_t)tx) * 8u);
}

// gsamLinearWrap on the 256x256 RGBA8 random-vector map  (CRYCHIC.cpp:1068-1073)
CRY_HD f3 randvec_linear_wrap(const uint32_t* __restrict__ rv, float u, float v)
{
    float uw = u - __builtin_floorf(u), vw = v - __builtin_floorf(v);
    Bilin b = bilinear_setup(uw, vw, 256, 256);
    uint32_t x0 = (uint32_t)b.i0 & 255u, x1 = (uint32_t)(b.i0 + 1) & 255u;
    uint32_t y0 = (uint32_t)b.j0 & 255u, y1 = (uint32_t)(b.j0 + 1) & 255u;
    const uint32_t t00 = load_at<uint32_t>(rv, (y0 * 256u + x0) * 4u), t10 = load_at<uint32_t>(rv, (y0 * 256u + x1) * 4u);
    const uint32_t t01 = load_at<uint32_t>(rv, (y1 * 256u + x0) * 4u), t11 = load_at<uint32_t>(rv, (y1 * 256u + x1) * 4u);
    f3 o;
    o.x = bilerp(unorm8_to_float(t00 & 255u), unorm8_to_float(t10 & 255u), unorm8_to_float(t01 & 255u),
                 unorm8_to_float(t11 & 255u), b.fx, b.fy);
    o.y = bilerp(unorm8_to_float((t00 >> 8) & 255u), unorm8_to_float((t10 >> 8) & 255u),
                 unorm8_to_float((t01 >> 8) & 255u), unorm8_to_float((t11 >> 8) & 255u), b.fx, b.fy);
    o.z = bilerp(unorm8_to_float((t00 >> 16) & 255u), unorm8_to_float((t10 >> 16) & 255u),
                 unorm8_to_float((t01 >> 16) & 255u), unorm8_to_float((t11 >> 16) & 255u), b.fx, b.fy);
    return o;
}

// gProjTex = Proj * T (CRYCHIC.cpp:828-834,918) has seven structural zeros and a one for every perspective projection:
//   x' = q.x PT[0] + q.z PT[2],  y' = q.y PT[5] + q.z PT[6],  w' = q.z.
// Skipping the products with exact zeros changes no output bit as long as q is finite (x * 0 = +-0, and a +-0 term can
// only change the sign of a zero sum; a zero x', y' or w' yields the same texel / BORDER decision under either sign, see
// DESIGN.md) -- so the short form is used when the constants have that pattern and are bounded (checked once on the
// host) and the pixel's p is bounded (checked per pixel; a wave with any other lane takes the general loop).
template <bool V> struct SparseTag { static constexpr bool value = V; };
CRY_HD bool ssao_projtex_is_sparse(const crychic_ssao_constants& cb)
{
    const float* PT = cb.ProjTex;
    bool ok = PT[1] == 0.0f && PT[3] == 0.0f && PT[4] == 0.0f && PT[7] == 0.0f && PT[12] == 0.0f && PT[13] == 0.0f && PT[14] == 1.0f &&
              PT[15] == 0.0f;
    const float big = 1.0e12f;
    ok = ok && __builtin_fabsf(PT[0]) < big && __builtin_fabsf(PT[2]) < big && __builtin_fabsf(PT[5]) < big && __builtin_fabsf(PT[6]) < big;
    ok = ok && __builtin_fabsf(cb.OcclusionRadius) < big;
    for (int i = 0; i < 14; ++i)
        for (int k = 0; k < 3; ++k) ok = ok && __builtin_fabsf(cb.OffsetVectors[i][k]) < big;   // false for NaN too
    return ok;
}

// ---- sky shortcut -------------------------------------------------------------------------------------------------------
// A pixel whose own footprint and every tap footprint read only texels at the clear depth has ambient access exactly 1:
// every tap then samples z_ndc = 1, so rz equals the pixel's own pz bit for bit, distZ = p.z - r.z is a few ulps of the far
// distance -- below SurfaceEpsilon -- the occlusion term is 0 and a finite dp times 0 leaves the sum at +0.  Where the taps
// of such a pixel can land is bounded on the host from the constants (SkyReach); whether that neighbourhood is free of geometry
// is answered by the coarse geometry map the depth-pairs pass fills.  A wavefront of such pixels writes 65535 and skips its
// 14 taps (arithmetic and gather); in the benchmark frame that is about 45 % of all wavefronts.  Exact by construction and by
// the parity / fuzz tests, which run the kernel bodies with the shortcut against an oracle that has none.
struct SkyReach {
    int enabled;          // 0: the constants do not allow the argument (see ssao_sky_reach)
    uint32_t rx, ry;      // a sky pixel's taps stay within +-rx / +-ry depth texels of its own 2 x 2 footprint
    int y0, y1;           // depth texel rows [y0, y1) whose cells of the geometry map this frame's depth pass wrote (launcher)
};
CRY_HD SkyReach ssao_sky_reach(const crychic_ssao_constants& cb, uint32_t W, uint32_t H)
{
    SkyReach r{ 0, 0u, 0u, 0, (int)H };
    if (!ssao_projtex_is_sparse(cb)) return r;
    const double A = cb.Proj[4 * 2 + 2], B = cb.Proj[4 * 2 + 3];
    const double farZ = B / (1.0 - A);                            // view depth of z_ndc = 1 (Ssao.hlsl:110-115)
    const double eps = cb.SurfaceEpsilon, R = __builtin_fabs((double)cb.OcclusionRadius);
    if (!(farZ > 0.0) || !(farZ < 1.0e6) || !(eps > farZ * 1.52587890625e-5)) return r;      // |distZ| <= a few ulp(far) << 2^-16 far < eps
    double omax = 0.0;
    for (int i = 0; i < 14; ++i) {
        const double o = __builtin_sqrt((double)cb.OffsetVectors[i][0] * cb.OffsetVectors[i][0] + (double)cb.OffsetVectors[i][1] * cb.OffsetVectors[i][1] +
                                        (double)cb.OffsetVectors[i][2] * cb.OffsetVectors[i][2]);
        omax = o > omax ? o : omax;
    }
    // |reflect(o, rv)| <= |o| (1 + 2 |rv|^2) with rv in [-1, 1]^3; the tap moves q by at most D in view space
    const double D = R * omax * 7.0 * 1.001;
    if (!(D < 0.5 * farZ)) return r;
    // view-ray slopes |PosV.x / PosV.z|, |PosV.y / PosV.z| at the screen corners (VS :58-72)
    double tx = 0.0, ty = 0.0;
    for (int c = 0; c < 4; ++c) {
        const double hx = (c & 1) ? 1.0 : -1.0, hy = (c & 2) ? 1.0 : -1.0;
        double ph[4];
        for (int j = 0; j < 4; ++j) ph[j] = hx * cb.InvProj[4 * j + 0] + hy * cb.InvProj[4 * j + 1] + cb.InvProj[4 * j + 3];
        if (!(__builtin_fabs(ph[2]) > 1.0e-12)) return r;
        const double sx = __builtin_fabs(ph[0] / ph[2]), sy = __builtin_fabs(ph[1] / ph[2]);
        tx = sx > tx ? sx : tx; ty = sy > ty ? sy : ty;
    }
    // u = PT0 q.x / q.z + PT2 (sparse gProjTex): |q.x / q.z - p.x / p.z| <= D (1 + t) / (far - D), in texels times W |PT0|;
    // + 2 for the bilinear footprint, + 6 of slack for the binary32 evaluation of both sides
    const double dx = (double)W * __builtin_fabs((double)cb.ProjTex[0]) * D * (1.0 + tx) / (farZ - D);
    const double dy = (double)H * __builtin_fabs((double)cb.ProjTex[5]) * D * (1.0 + ty) / (farZ - D);
    if (!(dx < 4096.0) || !(dy < 4096.0)) return r;
    r.enabled = 1;
    r.rx = (uint32_t)dx + 8u;
    r.ry = (uint32_t)dy + 8u;
    return r;
}
// Lane predicate: the pixel's own footprint is four texels at the clear depth (so pz is the far distance exactly) and its
// normal is finite (an infinite normal would turn 0 * dp into NaN).
CRY_HD bool ssao_sky_lane(float t00, float t10, float t01, float t11, u2 nrmBits)
{
    const f3 n = unpack_normal(nrmBits);
    const bool finite = (__builtin_fabsf(n.x) < 3.0e38f) & (__builtin_fabsf(n.y) < 3.0e38f) & (__builtin_fabsf(n.z) < 3.0e38f);
    return (t00 == 1.0f) & (t10 == 1.0f) & (t01 == 1.0f) & (t11 == 1.0f) & finite;
}
// The cells of the coarse geometry map that the taps of half-res pixels [x0, x0 + n) of half-res row y can reach.
struct GeoCells { uint32_t cx0, cx1, cy0, cy1; bool known; };
CRY_HD GeoCells ssao_sky_cells(const SkyReach& r, uint32_t W, uint32_t H, uint32_t x0, uint32_t n, uint32_t y)
{
    const int xa = 2 * (int)x0 - (int)r.rx, xb = 2 * (int)(x0 + n) - 1 + (int)r.rx;      // texel columns, inclusive
    const int ya = clampi(2 * (int)y - (int)r.ry, 0, (int)H - 1), yb = clampi(2 * (int)y + 1 + (int)r.ry, 0, (int)H - 1);
    GeoCells g;                                       // texels outside the plane are BORDER = the clear depth: nothing to look up
    g.cx0 = (uint32_t)((clampi(xa, 0, (int)W - 1) + 2) / 128);
    g.cx1 = (uint32_t)((clampi(xb, 0, (int)W - 1) + 2) / 128);
    g.cy0 = (uint32_t)(ya / 32);
    g.cy1 = (uint32_t)(yb / 32);
    // a cell the depth pass of this frame did not visit says nothing (a row-limited pass, DepthPairsRows): no shortcut then
    const int cellsEnd = 32 * (int)g.cy1 + 32 < (int)H ? 32 * (int)g.cy1 + 32 : (int)H;      // texel rows [32 cy0, cellsEnd) belong to the cells
    g.known = 32 * (int)g.cy0 >= r.y0 && cellsEnd <= r.y1;
    return g;
}

// Cell rows of the nearest-depth map [*c0, *c0 + *cn) that the depth pass visits for an SSAO pass over half-res rows
// [row0, row0 + rows): the rows' own texels and a margin of H / 11 texel rows on either side (4K: 200; the sky shortcut's reach
// is 170 there), which keeps nearly every tap of the rows inside -- what leaves it takes DepthPairsRows' detour, so the margin is
// a performance knob only (`margin` < 0: the default).  Footprint rows j0 with 8 c0 <= j0 + 2 < 8 (c0 + cn) are prepared.
CRY_HD void depth_pass_cell_rows(uint32_t H, uint32_t row0, uint32_t rows, uint32_t* c0, uint32_t* cn, int margin = -1)
{
    if (margin < 0) margin = (int)((H / 11u + 7u) & ~7u) < 64 ? 64 : (int)((H / 11u + 7u) & ~7u);
    int a = (2 * (int)row0 - margin + 2) / 8, b = (2 * (int)(row0 + rows) + margin + 2 + 7) / 8;
    a = a < 0 ? 0 : a;
    b = b > (int)zmin_map_rows(H) ? (int)zmin_map_rows(H) : b;
    *c0 = (uint32_t)a;
    *cn = (uint32_t)(b > a ? b - a : 0);
}

// ---- tap culling ---------------------------------------------------------------------------------------------------------
// A tap adds dp * occlusion to the sum, and occlusion is 0 whenever distZ = p.z - r.z <= SurfaceEpsilon (Ssao.hlsl:76-108): the
// surface under the tap is not in front of the pixel by more than epsilon.  On open ground that is nine taps out of ten, and the
// depth gather -- the whole cost of the pass -- only confirms it.  depth_pairs_kernel therefore also leaves, per cell of 9 x 9
// padded depth texels (cells 8 texels apart, so that they overlap by one row and one column), a LOWER BOUND of the view depth
// any footprint inside the cell can return:
//     cell = ndc_to_view(min over the cell of the decoded texels - 2^-21) * 0.999998.
// A footprint with top-left padded texel (ex, ey) lies inside cell (ex >> 3, ey >> 3); if that cell is >= p.z - epsilon the tap
// is skipped: gather, reconstruction, normalisation and all.  Why that is exact:
//   * bilinear filtering returns >= min(texels) - 2^-22 in binary32 (three lerps, each a mad on values in [0, 1]);
//   * ndc_to_view is B * rcp(z - A) with B < 0 < 1 < A (checked on the host: ssao_cull_params): every step is monotone, so the
//     cell bounds the tap's rz from below; r.z = (rz * rcp(q.z)) * q.z is rz within 3 ulp -- covered by the factor 0.999998;
//   * hence p.z - r.z <= epsilon in real numbers, and rounding is monotone: distZ <= epsilon, occlusion = 0;
//   * dp is finite -- the pixel's normal and p are finite and |p| < 1e15 (checked per pixel), q.z >= 1e-3 (checked per tap) -- so
//     the term is +0 and the sum does not change.
// The lookup is one 4-byte load from a 0.5 MB map (L1 / L2 resident) in place of a 16-byte gather that misses L1; the parity
// and fuzz tests run the kernel bodies with culling against an oracle that has none.
//
// Clear cells.  A cell whose 9 x 9 texels all hold the clear depth (decoded: exactly 1.0) gives every footprint inside it four
// texels of 1.0, whatever the fractions: zndc = 1.0 exactly, rz = ndc_to_view(1.0) = the far distance F, r.z = F within 3 ulp.  The
// pixel's own p.z is ndc_to_view of a filtered value <= 1.0, so p.z <= F within 3 ulp and distZ <= 2^-20 F -- below SurfaceEpsilon
// when SurfaceEpsilon > 2^-16 F (the sky shortcut's guard, checked on the host: CullParams::clear).  Such a cell is stored as +inf:
// every tap of a pixel that may cull at all is culled there, a tap of a pixel that may not (non-finite normal, huge p) reads the
// footprint as four times 1.0 without looking at memory (ssao_pixel) -- so NOTHING reads the pairs entries that only clear cells
// can reach, and depth_pairs_kernel does not write them: on the benchmark frame that is 35 of the plane's 67 MB.
struct CullParams { int enabled; float A, B; int clear; };
CRY_HD CullParams ssao_cull_params(const crychic_ssao_constants& cb)
{
    CullParams c{ 0, cb.Proj[4 * 2 + 2], cb.Proj[4 * 2 + 3], 0 };
    const float eps = cb.SurfaceEpsilon;
    // z - A < 0 for every filtered z <= 1 + 2^-21, B < 0: view depth positive and increasing in z; far plane finite
    const bool ok = c.A > 1.000002f && c.A < 1.0e6f && c.B < 0.0f && c.B > -1.0e12f && eps == eps && __builtin_fabsf(eps) < 1.0e30f;
    c.enabled = ok ? 1 : 0;
    const double farZ = (double)c.B / (1.0 - (double)c.A);
    c.clear = (ok && farZ > 0.0 && farZ < 1.0e6 && (double)eps > farZ * 1.52587890625e-5) ? 1 : 0;
    return c;
}
CRY_HD float zmin_cell_value(const CullParams& c, float blockMinNdc)
{
    if (c.clear && blockMinNdc == 1.0f) return u2f(0x7F800000u);                      // a clear cell
    return divf(c.B, (blockMinNdc - 4.76837158203125e-7f) - c.A) * 0.999998f;         // ndc_to_view(min - 2^-21), scaled down
}
CRY_HD bool zmin_cell_is_clear(float cell) { return cell == u2f(0x7F800000u); }
// per pixel: p.z - epsilon when the pixel may cull at all, NaN (no comparison succeeds) otherwise
CRY_HD float ssao_cull_threshold(f3 nRaw, f3 p, float eps)
{
    const float big = 1.0e15f;
    const bool ok = (__builtin_fabsf(nRaw.x) < 3.0e38f) & (__builtin_fabsf(nRaw.y) < 3.0e38f) & (__builtin_fabsf(nRaw.z) < 3.0e38f) &
                    (__builtin_fabsf(p.x) < big) & (__builtin_fabsf(p.y) < big) & (__builtin_fabsf(p.z) < big);
    return ok ? p.z - eps : u2f(0x7FC00000u);
}
struct NoCull {
    static constexpr bool active = false;
    CRY_HD float cell(int, int) const { return 0.0f; }
};
struct ZminMap {
    static constexpr bool active = true;
    const float* cells; uint32_t pitch;                            // EdgePlane::zcull, zmin_map_cols(W)
    // the cell of the footprint whose top-left texel is (i0, j0), both already in [-2, dim]: inside the map for every clamped index
    CRY_HD float cell(int i0, int j0) const
    {
        const uint32_t cx = (uint32_t)(i0 + 2) >> 3, cy = (uint32_t)(j0 + 2) >> 3;
        return load_at<float>(cells, (mul24(cy, pitch) + cx) * 4u);
    }
};

// The same over a row-limited depth pass (DepthPairsRows): a footprint outside the prepared rows has no cell -- NaN, which no
// comparison culls.
struct ZminMapRows {
    static constexpr bool active = true;
    ZminMap map; int j0lo; uint32_t nj;
    CRY_HD float cell(int i0, int j0) const
    {
        const bool ok = (uint32_t)(j0 - j0lo) < nj;
        const float c = map.cell(i0, ok ? j0 : j0lo);
        return ok ? c : u2f(0x7FC00000u);
    }
};

struct SsaoCentre {
    u2 nrm_bits;  // raw fp16 normal texel
    float vz;        // linear view depth at the pixel centre
    bool sky;        // ssao_sky_lane(): candidate for the sky shortcut (pairs path only)
};

CRY_HD SsaoCentre ssao_centre(const crychic_ssao_constants& cb, const u2* __restrict__ normal,
                              const uint32_t* __restrict__ depth, uint32_t W, uint32_t H, int x, int y)
{
    SsaoCentre c;
    c.nrm_bits = normal_texel_bits(normal, W, H, x, y);
    float t00, t10, t01, t11;
    depth_at_half_pixel(depth, W, H, x, y, t00, t10, t01, t11);
    c.vz = ndc_to_view(cb, bilerp(t00, t10, t01, t11, 0.5f, 0.5f));
    c.sky = ssao_sky_lane(t00, t10, t01, t11, c.nrm_bits);
    return c;
}
// Ssao.hlsl:117-199 for half-res pixel (x, y); returns the R16_UNORM ambient value.  `sparseProjTex` = ssao_projtex_is_sparse(cb).
// `cull`: NoCull, or the ZminMap of the tap culling.  `culledTaps` (host builds only, may be null): three words -- [0] += the number
// of taps culled, [1] |= bit i for every culled tap i, [2] += the taps that were evaluated on a clear cell's constant footprint --
// for the tests that check that the culling bites.
// The pixel centre's uv = (x + 0.5) / (W / 2), (y + 0.5) / (H / 2) is a * rcp(b) with a wave-uniform b: the reciprocals are taken
// once on the host (the same correctly rounded rcp, devmath.hpp) and handed to the kernel.
struct HalfResScale { float rw2, rh2; };
CRY_HD HalfResScale half_res_scale(uint32_t W, uint32_t H) { return HalfResScale{ rcp((float)(W / 2)), rcp((float)(H / 2)) }; }

template <class Depth, class Cull = NoCull>
CRY_HD uint32_t ssao_pixel(const crychic_ssao_constants& cb, const SsaoCentre& c,
                           const Depth depth, const uint32_t* __restrict__ randvec, uint32_t W,
                           uint32_t H, uint32_t x, uint32_t y, const HalfResScale hs, bool sparseProjTex, const Cull cull = Cull(),
                           uint32_t* culledTaps = nullptr)
{
    const float u = ((float)x + 0.5f) * hs.rw2;
    const float v = ((float)y + 0.5f) * hs.rh2;

    // VS :58-72 evaluated at the pixel centre
    const float hx = fma(2.0f, u, -1.0f), hy = fma(-2.0f, v, 1.0f);
    const float phx = mulcol(hx, hy, 0.0f, 1.0f, cb.InvProj + 0);
    const float phy = mulcol(hx, hy, 0.0f, 1.0f, cb.InvProj + 4);
    const float phz = mulcol(hx, hy, 0.0f, 1.0f, cb.InvProj + 8);
    const float phw = mulcol(hx, hy, 0.0f, 1.0f, cb.InvProj + 12);
    const float rphw = rcp(phw);
    const f3 PosV{ phx * rphw, phy * rphw, phz * rphw };

    const f3 nRaw = unpack_normal(c.nrm_bits);
    const f3 n = normalize3(nRaw);                        // :125
    const float pz = c.vz;                                // :126-127
    const float t = divf(pz, PosV.z);                     // :135
    const f3 p{ t * PosV.x, t * PosV.y, t * PosV.z };
    const float pzEps = Cull::active ? ssao_cull_threshold(nRaw, p, cb.SurfaceEpsilon) : 0.0f;   // "tap culling"

    const f3 rv = randvec_linear_wrap(randvec, 4.0f * u, 4.0f * v);  // :138
    const f3 randVec{ fma(2.0f, rv.x, -1.0f), fma(2.0f, rv.y, -1.0f), fma(2.0f, rv.z, -1.0f) };

    const float eps = cb.SurfaceEpsilon, fadeEnd = cb.OcclusionFadeEnd;
    const float rFadeLength = rcp(cb.OcclusionFadeEnd - cb.OcclusionFadeStart);  // :100,104: one reciprocal for all taps

    // Taps are evaluated two at a time in packed fp32 (devmath.hpp "two-wide packed fp32"); lane .x is tap i, lane .y tap
    // i+1, and the occlusion terms are added to the sum in tap order, so every bit equals the one-tap-at-a-time loop.
    const f3x2 n2 = splat3(n), p2 = splat3(p), rv2 = splat3(randVec);
    const float A = cb.Proj[4 * 2 + 2], B = cb.Proj[4 * 2 + 3];   // Ssao.hlsl:110-115
    const float* PT = cb.ProjTex;
    // |p| < 2^100 and bounded constants keep every q finite (|offset| <= 3e12 * 5, |fr| <= 1e12)
    const float pmax = 1.2676506e30f;
    bool sparse = sparseProjTex && __builtin_fabsf(p.x) < pmax && __builtin_fabsf(p.y) < pmax && __builtin_fabsf(p.z) < pmax;
#if defined(__HIP_DEVICE_COMPILE__)
    sparse = __builtin_amdgcn_ballot_w64(!sparse) == 0;     // wave-uniform: one loop per wave
#endif
    float occlusionSum = 0.0f;
    auto taps = [&](auto sparseTag) {
    constexpr bool SPARSE = decltype(sparseTag)::value;
    // the offset vectors of a pair are fetched (scalar loads from the constants) one iteration ahead of their use
    f3x2 oNext{ v2f{ cb.OffsetVectors[0][0], cb.OffsetVectors[1][0] }, v2f{ cb.OffsetVectors[0][1], cb.OffsetVectors[1][1] },
                v2f{ cb.OffsetVectors[0][2], cb.OffsetVectors[1][2] } };
#pragma unroll 1
    for (int i = 0; i < 14; i += 2) {
        const f3x2 o = oNext;
        const int in = i + 2 < 14 ? i + 2 : 0;
        oNext = f3x2{ v2f{ cb.OffsetVectors[in][0], cb.OffsetVectors[in + 1][0] }, v2f{ cb.OffsetVectors[in][1], cb.OffsetVectors[in + 1][1] },
                      v2f{ cb.OffsetVectors[in][2], cb.OffsetVectors[in + 1][2] } };
        const v2f d2 = 2.0f * dot3x2(rv2, o);                                     // reflect(o, randVec)  :148
        const v2f nd2 = -d2;
        const f3x2 offset{ fma2(nd2, rv2.x, o.x), fma2(nd2, rv2.y, o.y), fma2(nd2, rv2.z, o.z) };
        const v2f fr = sign2(dot3x2(offset, n2)) * cb.OcclusionRadius;            // :151,154
        const f3x2 q{ fma2(fr, offset.x, p2.x), fma2(fr, offset.y, p2.y), fma2(fr, offset.z, p2.z) };
        v2f pqx, pqy, pqw;                                                        // mul(float4(q,1), gProjTex)  :157
        if (SPARSE) {
            pqx = fma2(q.z, PT[2], q.x * PT[0]);
            pqy = fma2(q.z, PT[6], q.y * PT[5]);
            pqw = q.z;
        } else {
            pqx = fma2(q.z, PT[2], fma2(q.y, PT[1], q.x * PT[0])) + PT[3];
            pqy = fma2(q.z, PT[6], fma2(q.y, PT[5], q.x * PT[4])) + PT[7];
            pqw = fma2(q.z, PT[14], fma2(q.y, PT[13], q.x * PT[12])) + PT[15];
        }
        const v2f rqz = rcp2(q.z);                                                // 1 / q.z: used at :171, and at :158 when w' = q.z
        const v2f rpw = SPARSE ? rqz : rcp2(pqw);
        const v2f tu = pqx * rpw, tv = pqy * rpw;                                 // :158

        // gsamDepthMap, both taps: bilinear setup in packed form, the 2 x 2 footprints through the paired loads
        const v2f tx = fma2(tu, (float)W, -0.5f), ty = fma2(tv, (float)H, -0.5f);
        const v2f flx = floor2(tx), fly = floor2(ty);
        // Fractions: in [0, 1) or NaN (non-finite coordinate); fmax(NaN, 0) = 0.  Texel indices: clamp(floor, -2, dim)
        // sends +-inf out of range and NaN to -2, so a non-finite coordinate addresses only BORDER texels on its own axis --
        // which makes all four texels of the footprint the border value, the same result as the scalar sampler's joint
        // "bad => (-2, -2)" rule (the CLAMP samplers, where the two rules differ, keep the scalar bilinear_setup).
        const v2f fx = v2f{ __builtin_fmaxf(tx.x - flx.x, 0.0f), __builtin_fmaxf(tx.y - flx.y, 0.0f) };
        const v2f fy = v2f{ __builtin_fmaxf(ty.x - fly.x, 0.0f), __builtin_fmaxf(ty.y - fly.y, 0.0f) };
        const int i0a = texel_index(flx.x, W), j0a = texel_index(fly.x, H);
        const int i0b = texel_index(flx.y, W), j0b = texel_index(fly.y, H);
        // tap culling: a tap whose footprint cannot return a surface in front of the pixel adds exactly +0.  Both lookups are
        // issued unconditionally and together (one round trip; a NaN on either side of the comparison: not culled)
        const float cellA = cull.cell(i0a, j0a), cellB = cull.cell(i0b, j0b);
        const bool ca = Cull::active & (q.z.x >= 1.0e-3f) & (cellA >= pzEps);          // & not &&: neither load may hide behind a branch
        const bool cb2 = Cull::active & (q.z.y >= 1.0e-3f) & (cellB >= pzEps);
        if (Cull::active) {
#if defined(__HIP_DEVICE_COMPILE__)
            if (__builtin_amdgcn_ballot_w64(!(ca & cb2)) == 0) continue;      // the whole wavefront skips both taps
#else
            if (culledTaps) { culledTaps[0] += (ca ? 1u : 0u) + (cb2 ? 1u : 0u); culledTaps[1] |= ((ca ? 1u : 0u) << i) | ((cb2 ? 1u : 0u) << (i + 1)); }
            if (ca & cb2) continue;
#endif
        }
        // Both footprints in one round trip as well: a culled lane fetches the plane's first footprint instead of its own (every
        // culled lane of the wavefront the same line, so the gather it is spared stays spared) and never looks at it.
        float a00, a10, a01, a11, b00, b10, b01, b11;
        depth_tap(depth, ca, i0a, j0a, a00, a10, a01, a11);
        depth_tap(depth, cb2, i0b, j0b, b00, b10, b01, b11);
        if (Cull::active) {      // a footprint inside a clear cell is four times 1.0; its pairs entries may not exist ("clear cells")
            const bool sa = zmin_cell_is_clear(cellA), sb = zmin_cell_is_clear(cellB);
            a00 = sa ? 1.0f : a00; a10 = sa ? 1.0f : a10; a01 = sa ? 1.0f : a01; a11 = sa ? 1.0f : a11;
            b00 = sb ? 1.0f : b00; b10 = sb ? 1.0f : b10; b01 = sb ? 1.0f : b01; b11 = sb ? 1.0f : b11;
#if !defined(__HIP_DEVICE_COMPILE__)
            if (culledTaps) culledTaps[2] += (!ca && sa ? 1u : 0u) + (!cb2 && sb ? 1u : 0u);
#endif
        }
        const v2f t00{ a00, b00 }, t10{ a10, b10 }, t01{ a01, b01 }, t11{ a11, b11 };
        const v2f zndc = lerp2(lerp2(t00, t10, fx), lerp2(t01, t11, fx), fy);
        // :164-165.  With the tap culling on, the host has checked 1.000002 < A < 1e6 (ssao_cull_params) and zndc is a filtered
        // decoded D24 value in [0, 1 + 2^-21]: zndc - A is a normal number with a normal reciprocal, where rcp_normal IS rcp
        const v2f rz = B * (Cull::active ? rcp_normal2(zndc - A) : rcp2(zndc - A));
        const v2f sc = rz * rqz;                                                  // :171
        const f3x2 r{ sc * q.x, sc * q.y, sc * q.z };
        const v2f distZ = p2.z - r.z;                                             // :185
        const f3x2 dv{ r.x - p2.x, r.y - p2.y, r.z - p2.z };
        const v2f inv = inv_len_from_sq2(dot3x2(dv, dv));                         // normalize(r - p)
        const f3x2 dn{ dv.x * inv, dv.y * inv, dv.z * inv };
        const v2f dp = max0_2(dot3x2(n2, dn));                                    // :186
        const v2f fade = saturate2((fadeEnd - distZ) * rFadeLength);              // :76-108
        const v2f occ = select2(distZ > eps, fade, splat(0.0f));
        // a culled tap of a lane whose wavefront went on: its term is the +0 the argument above promises (the placeholder
        // footprint is never looked at)
        occlusionSum = fma(ca ? 0.0f : dp.x, ca ? 0.0f : occ.x, occlusionSum);      // :188-190, tap i then tap i+1
        occlusionSum = fma(cb2 ? 0.0f : dp.y, cb2 ? 0.0f : occ.y, occlusionSum);
    }
    };
    if (sparse) taps(SparseTag<true>{}); else taps(SparseTag<false>{});
    occlusionSum = occlusionSum * (1.0f / 14.0f);                                // :193: a / 14 = a * rcp(14)
    const float access = 1.0f - occlusionSum;                                    // :195
    const float a2 = access * access, a4 = a2 * a2;                              // :198 pow(access, 6)
    return float_to_unorm16(a4 * a2);
}

// ---- unoccluded tiles --------------------------------------------------------------------------------------------------
// A blur output whose 11-tap window holds only 1.0 is 1.0 whatever the edge tests decide (the accepted weights are summed into
// colour and total alike, and x * rcp(x) quantises to 65535).  So if every ambient value within `margin` = 5 pixels per
// iteration of a tile is 65535 when the first sweep starts, the tile stays 65535 through all sweeps of the frame: the first blur
// launch settles it (writes 65535, a centre-only decision and the tile's flag -- blur_tiles.hpp) and the later launch skips it.
// Whether the neighbourhood is all ones is answered by the unoccluded-wavefront map the SSAO pass fills: one word per wavefront
// that emitted ambient values, the frame's stamp if all 64 are 65535 and 0 otherwise.  A tile whose neighbourhood leaves the rows
// the SSAO pass computed this frame is never settled, so no word is looked at that this frame did not write.
// CLAMP addressing maps taps beyond the map to its edge texels, so the neighbourhood is clamped to the map, not extended.
struct OnesRegion { uint32_t c0, c1, r0, r1; };      // inclusive cell columns (64-pixel segments) and half-res rows
CRY_HD OnesRegion blur_ones_region(uint32_t w2, uint32_t h2, int x0, int y0, int bw, int bh, int margin)
{
    OnesRegion g;
    g.c0 = (uint32_t)clampi(x0 - margin, 0, (int)w2 - 1) / 64u;
    g.c1 = (uint32_t)clampi(x0 + bw - 1 + margin, 0, (int)w2 - 1) / 64u;
    g.r0 = (uint32_t)clampi(y0 - margin, 0, (int)h2 - 1);
    g.r1 = (uint32_t)clampi(y0 + bh - 1 + margin, 0, (int)h2 - 1);
    return g;
}

// ---- bilateral blur (SsaoBlur.hlsl:85-146) ----------------------------------------------------------------
struct BlurTap {
    f3 n;     // raw (un-normalised) normal
    float z;  // linear view depth
    float a;  // ambient value
};

// Fetch the edge data + ambient for half-res position (xi, yi), which may lie outside the map along the
// sweep axis: normal CLAMPs in full-res texel space, depth takes the BORDER value, ambient CLAMPs.
// In two halves, so that a caller can issue the loads of many positions before it decodes the first (blur_tiles.hpp).
struct BlurTapRaw { u2 nrm; float vz; uint32_t amb; bool inside; };
CRY_HD BlurTapRaw blur_fetch_raw(const EdgePlane& e, const uint16_t* __restrict__ amb, int w2, int h2, int xi, int yi)
{
    const int cx = clampi(xi, 0, w2 - 1), cy = clampi(yi, 0, h2 - 1);
    const uint32_t idx = mul24((uint32_t)cy, (uint32_t)w2) + (uint32_t)cx;
    const u2* src = e.nrm + idx;                       // one load through a selected address (no divergent branches)
    src = (yi < 0) ? e.grow + cx : src;
    src = (xi < 0) ? e.gcol + cy : src;
    BlurTapRaw r;
    r.nrm = *src;
    r.vz = e.vz[idx];
    r.amb = amb[idx];
    r.inside = ((uint32_t)xi < (uint32_t)w2) & ((uint32_t)yi < (uint32_t)h2);
    return r;
}
CRY_HD BlurTap blur_fetch_decode(const BlurTapRaw& r, float borderZ)
{
    BlurTap t;
    t.n = unpack_normal(r.nrm);
    t.z = r.inside ? r.vz : borderZ;
    t.a = unorm16_to_float(r.amb);
    return t;
}
CRY_HD BlurTap blur_fetch(const EdgePlane& e, const uint16_t* __restrict__ amb, float borderZ, int w2, int h2,
                          int xi, int yi)
{
    return blur_fetch_decode(blur_fetch_raw(e, amb, w2, h2, xi, yi), borderZ);
}

// One output pixel from its 11 taps; fetch(i) returns tap i (i = 5 is the centre).  Accumulation order is the
// shader's loop order (SsaoBlur.hlsl:113).  Also reports which taps passed the edge test and the final totalWeight.
struct BlurOut {
    uint32_t value;  // R16_UNORM
    uint32_t mask;   // bit i set <=> tap i accepted (bit 5, the centre, always set)
    float total;
};
template <class Fetch>
CRY_HD BlurOut blur_pixel_full(const float* __restrict__ w, Fetch fetch)
{
    const BlurTap c = fetch(5);
    float color = w[5] * c.a;           // :106
    float total = w[5];                 // :107
    uint32_t mask = 1u << 5;
#pragma unroll
    for (int i = 0; i < 11; ++i) {      // :113
        if (i == 5) continue;
        const BlurTap t = fetch(i);
        const bool ok = (dot3(t.n, c.n) >= 0.8f) & (__builtin_fabsf(t.z - c.z) <= 0.2f);  // :131-132
        // A rejected tap adds weight 0: color + 0 * a and total + 0 leave both sums bit-unchanged (a is a decoded UNORM,
        // finite and >= 0, so neither sum is ever -0) -- one select instead of two, and the wave never diverges here.
        const float ws = ok ? w[i] : 0.0f;
        color = fma(ws, t.a, color);
        total = total + ws;
        mask |= ok ? (1u << i) : 0u;
    }
    return BlurOut{ float_to_unorm16(divf(color, total)), mask, total };  // :145
}
template <class Fetch>
CRY_HD uint32_t blur_pixel(const float* __restrict__ w, Fetch fetch) { return blur_pixel_full(w, fetch).value; }

// The same pixel with the decisions replayed from a previous sweep of the same direction: identical float operations in
// identical order, minus the normal / depth tests.  amb(i) returns the ambient value of tap i.  The totalWeight is rebuilt from
// the mask by the very additions blur_pixel_full made (+ w[i] for an accepted tap, + 0 for a rejected one, in loop order).
template <class Amb>
CRY_HD uint32_t blur_pixel_replay(const float* __restrict__ w, uint32_t mask, Amb amb)
{
    float color = w[5] * amb(5);
    float total = w[5];
#pragma unroll
    for (int i = 0; i < 11; ++i) {
        if (i == 5) continue;
        // weight or +0.0 by AND-ing the weight's bits with the sign-extended mask bit (v_bfe_i32 + v_and_b32); adding 0 * a
        // leaves the colour sum bit-unchanged, see blur_pixel_full
        const uint32_t keep = (uint32_t)(((int32_t)(mask << (31 - i))) >> 31);
        const float wk = u2f(f2u(w[i]) & keep);
        color = fma(wk, amb(i), color);
        total = total + wk;
    }
    return float_to_unorm16(divf(color, total));
}

}  // namespace cry
