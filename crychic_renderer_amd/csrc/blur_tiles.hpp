// blur_tiles.hpp -- workgroup-level bodies of the two blur launches of Ssao::ComputeSsao (Ssao.cpp:231-293 issues
// 2 * blurCount draws of Shaders/SsaoBlur.hlsl:85-146; here the whole chain is two launches):
//
//   blur_pair_tile          iteration 0: the horizontal AND the vertical sweep of one 64 x 16 tile.  The horizontal results of
//                           the tile's rows plus a 5-row apron stay in LDS (quantised to R16_UNORM and decoded again exactly as
//                           the round trip through the ambient map does) and feed the vertical sweep; both sweeps record their
//                           tap decisions (ssao_core.hpp EdgePlane::mask_* / total_*).
//   blur_replay_fused_tile  iterations 1 .. k (k <= 3) of one tile in one go: the tile plus an apron of 5 k pixels is staged
//                           once, every iteration replays the recorded decisions on a region that shrinks by 5 pixels per side,
//                           and only the tile itself is written back.  Redundant apron work buys the absence of any exchange
//                           between tiles, hence of launches between iterations.
//
// Tiles whose whole neighbourhood came out of the SSAO pass as 65535 ("unoccluded tiles", ssao_core.hpp) are settled by the first
// launch (it writes 65535 and a per-tile flag) and cost the second launch one scalar load.  In the benchmark frame that is three
// tiles out of four.
//
// The bodies are host+device text over a `Block` (thread id, block size, barrier, block-wide vote): the kernels instantiate
// them with the real workgroup (kernels.hip BlockDev), tests/hostsim runs the very same text with one sequential "thread"
// (BlockSeq) against the oracle's sweep-by-sweep chain.  Every output bit equals the per-sweep kernels': same per-pixel
// functions (blur_pixel_full / blur_pixel_replay), same operands, same order.
#pragma once
#include "ssao_core.hpp"

namespace cry {

constexpr int kBlurTileW = 64, kBlurTileH = 16, kBlurRadius = 5;
constexpr int kBlurMaxFused = 3;                                        // replay iterations per launch (LDS: 2 x 94 x 46 floats)
constexpr int kBlurPairSW = kBlurTileW + 2 * kBlurRadius;               // staged width of the pair launch: 74
constexpr int kBlurPairSH = kBlurTileH + 2 * kBlurRadius;               // staged height: 26
constexpr int kBlurFusedMaxW = kBlurTileW + 2 * kBlurRadius * kBlurMaxFused;    // 94
constexpr int kBlurFusedMaxH = kBlurTileH + 2 * kBlurRadius * kBlurMaxFused;    // 46

// ---- the launch plan of Ssao::ComputeSsao's blur chain (shared by api.cpp and tests/hostsim) -----------------------------------
// Each launch reads one ambient plane and writes the other (never in place: a tile's apron is its neighbours' output):
// iteration 0 as one H + V launch that records the tap decisions, then the remaining iterations in launches of up to
// kBlurMaxFused, replayed.  The final map has to be ambient0 (Ssao.cpp:75-78), so the planes alternate backwards from there:
// with an odd number of launches the SSAO pass itself writes ambient1.  Tiles the first launch settles as unoccluded hold 65535
// in BOTH planes from then on (the SSAO pass wrote one, the first launch the other), so later launches neither read nor write them.
// Rows: the caller is owed half-res rows [row0, row0 + rows) of the final map; a vertical sweep reaches 5 rows (gBlurRadius,
// SsaoBlur.hlsl:48), so every earlier stage is computed on 5 more rows per remaining iteration (a halo recomputed, not exchanged).
CRY_HD void clamp_rows(uint32_t limit, int64_t lo, int64_t hi, uint32_t* row0, uint32_t* rows)
{
    if (lo < 0) lo = 0;
    if (hi > (int64_t)limit) hi = limit;
    if (hi < lo) hi = lo;
    *row0 = (uint32_t)lo;
    *rows = (uint32_t)(hi - lo);
}
CRY_HD int blur_chain_launches(int blurCount)
{
    return blurCount > 0 ? 1 + (blurCount - 1 + kBlurMaxFused - 1) / kBlurMaxFused : 0;
}
CRY_HD int blur_chain_ssao_plane(int blurCount) { return blur_chain_launches(blurCount) & 1; }      // 0 = ambient0, 1 = ambient1
struct BlurStep {
    int iterations;            // 0: the pair launch (iteration 0); k > 0: k replayed iterations
    int in, out;               // plane indices
    uint32_t row0, rows;       // rows of `out` the launch owes
};
CRY_HD BlurStep blur_chain_step(int blurCount, uint32_t row0, uint32_t rows, uint32_t h2, int i)
{
    BlurStep s;
    const int in = (blur_chain_ssao_plane(blurCount) + i) & 1;
    s.in = in;
    s.out = in ^ 1;
    int itersAfter = blurCount - 1;               // blur iterations still to run after this launch
    s.iterations = 0;
    int left = blurCount - 1;
    for (int l = 1; l <= i; ++l) {
        const int launchesLeft = (left + kBlurMaxFused - 1) / kBlurMaxFused;
        s.iterations = (left + launchesLeft - 1) / launchesLeft;        // spread evenly: 4 = 2 + 2, not 3 + 1 (apron work grows with k)
        left -= s.iterations;
        itersAfter = left;
    }
    clamp_rows(h2, (int64_t)row0 - 5 * itersAfter, (int64_t)row0 + rows + 5 * itersAfter, &s.row0, &s.rows);
    return s;
}
// rows the SSAO pass has to produce for the chain
CRY_HD void blur_chain_ssao_rows(int blurCount, uint32_t row0, uint32_t rows, uint32_t h2, uint32_t* r0, uint32_t* rn)
{
    clamp_rows(h2, (int64_t)row0 - 5 * blurCount, (int64_t)row0 + rows + 5 * blurCount, r0, rn);
}

// All 11 blur weights finite and in (0, 1e30): every total a recording sweep can produce is then finite and positive, which is
// what the all-ones arguments below (x * rcp(x) quantises to 65535) need.  Checked by the launchers; false switches the exits off.
CRY_HD bool blur_weights_positive(const crychic_ssao_constants& cb)
{
    const float* w = &cb.BlurWeights[0][0];
    for (int i = 0; i < 11; ++i)
        if (!(w[i] > 0.0f && w[i] < 1.0e30f)) return false;
    return true;
}

// One sequential "thread" standing in for the workgroup (host builds: tests/hostsim).
struct BlockSeq {
    CRY_HD int tid() const { return 0; }
    CRY_HD int size() const { return 1; }
    CRY_HD void sync() const {}
    CRY_HD bool all(bool p) const { return p; }
    CRY_HD bool any(bool p) const { return p; }
};

struct BlurTileArgs {
    const float* w;                 // the 11 weights, SsaoConstants::BlurWeights flattened (SsaoBlur.hlsl:66-72)
    EdgePlane e;
    const uint16_t* in;
    uint16_t* out;                  // != in
    int w2, h2;                     // half-res map size
    int x0, y0;                     // tile origin on the absolute 64 x 16 grid
    int row0, row1;                 // half-res rows of `out` this launch owes: [row0, row1)
    float borderZ;                  // view depth of the BORDER colour of gsamDepthMap
    uint32_t tileIndex;             // (y0 / 16) * blur_tiles_x + x0 / 64
};

// i / d and i % d for 0 <= i < 2^16, 0 < d < 2^12 without an integer division (rd = 1.0f / d)
CRY_HD void divmod_small(int i, int d, float rd, int& q, int& r)
{
    q = (int)((float)i * rd);
    r = i - q * d;
    if (r >= d) { ++q; r -= d; }
    if (r < 0) { --q; r += d; }
}

CRY_HD void blur_tile_fill_ones(int tid, int n, const BlurTileArgs& a)
{
    for (int k = tid; k < kBlurTileW * kBlurTileH; k += n) {
        const int x = a.x0 + (k & 63), y = a.y0 + (k >> 6);
        if (x < a.w2 && y >= a.row0 && y < a.row1) a.out[(uint32_t)y * (uint32_t)a.w2 + (uint32_t)x] = (uint16_t)0xFFFFu;
    }
}

// Iteration 0 of the blur for one tile: H sweep of rows y0 - 5 .. y0 + 20 into LDS, V sweep of the tile from it.
// RECORD: also store both sweeps' tap decisions and totals, and the tile's flag.  stamp != 0: the unoccluded-tile exit may be
// taken -- the SSAO pass of THIS frame wrote the unoccluded-wavefront map for half-res rows [ssaoRow0, ssaoRow1) with that stamp
// (a word of the map is only ever looked at inside those rows, where it was written this frame: stale contents cannot matter).
// s_nz: kBlurPairSW * kBlurPairSH entries; s_a: the same count; s_mid: kBlurTileW * kBlurPairSH.
template <bool RECORD, class Block>
CRY_HD void blur_pair_tile(const Block& blk, const BlurTileArgs& a, uint32_t stamp, int onesMargin, int ssaoRow0, int ssaoRow1,
                           f4a* s_nz, float* s_a, float* s_mid)
{
    constexpr int SW = kBlurPairSW, SH = kBlurPairSH, R = kBlurRadius;
    const int tid = blk.tid(), n = blk.size();
    bool settled = false;
    if (stamp != 0u) {
        const OnesRegion g = blur_ones_region((uint32_t)a.w2, (uint32_t)a.h2, a.x0, a.y0, kBlurTileW, kBlurTileH, onesMargin);
        if ((int)g.r0 >= ssaoRow0 && (int)g.r1 < ssaoRow1) {
            const int ncol = (int)(g.c1 - g.c0) + 1, ncell = ncol * ((int)(g.r1 - g.r0) + 1);
            const float rncol = 1.0f / (float)ncol;
            const uint32_t pitch = (uint32_t)((a.w2 + 63) / 64);
            bool occluded = false;
            for (int k = tid; k < ncell; k += n) {
                int r, c;
                divmod_small(k, ncol, rncol, r, c);
                occluded |= a.e.ones[(g.r0 + (uint32_t)r) * pitch + g.c0 + (uint32_t)c] != stamp;
            }
            settled = !blk.any(occluded);
        }
    }
    if (RECORD && tid == 0) a.e.tiles[a.tileIndex] = settled ? stamp : 0u;      // always written: a flag is never stale
    if (settled) {
        // Every ambient value within reach of this frame's sweeps is 65535, so the tile is 65535 after each of them.  The recorded
        // decision is "centre tap only": a neighbouring tile that replays these pixels in its apron gets w5 * 1 * rcp(w5) -> 65535,
        // the value every other decision would give too.
        for (int k = tid; k < kBlurTileW * kBlurTileH; k += n) {
            const int x = a.x0 + (k & 63), y = a.y0 + (k >> 6);
            if (x < a.w2 && y >= a.row0 && y < a.row1) {
                const uint32_t p = (uint32_t)y * (uint32_t)a.w2 + (uint32_t)x;
                a.out[p] = (uint16_t)0xFFFFu;
                if (RECORD) {
                    a.e.mask_h[p] = (uint16_t)(1u << 5); a.e.total_h[p] = a.w[5];
                    a.e.mask_v[p] = (uint16_t)(1u << 5); a.e.total_v[p] = a.w[5];
                }
            }
        }
        return;
    }
    // stage normal + depth + ambient of columns x0 - 5 .. x0 + 68, rows y0 - 5 .. y0 + 20 (rows CLAMPed into the map: a vertical
    // tap above / below the map reads the edge row's horizontal result)
    for (int k = tid; k < SW * SH; k += n) {
        int ly, lx;
        divmod_small(k, SW, 1.0f / (float)SW, ly, lx);
        const int cy = clampi(a.y0 - R + ly, 0, a.h2 - 1);
        const BlurTap t = blur_fetch(a.e, a.in, a.borderZ, a.w2, a.h2, a.x0 - R + lx, cy);
        s_nz[k] = f4a{ t.n.x, t.n.y, t.n.z, t.z };
        s_a[k] = t.a;
    }
    blk.sync();
    // horizontal sweep (gHorizontalBlur = 1, Ssao.cpp:240) of all staged rows
    for (int k = tid; k < kBlurTileW * SH; k += n) {
        const int ly = k >> 6, lx = k & 63, x = a.x0 + lx;
        if (x < a.w2) {
            const BlurOut o = blur_pixel_full(a.w, [&](int i) {
                const int idx = ly * SW + lx + i;
                const f4a q = s_nz[idx];
                return BlurTap{ f3{ q.x, q.y, q.z }, q.w, s_a[idx] };
            });
            s_mid[k] = unorm16_to_float(o.value);
            const int y = a.y0 - R + ly;
            if (RECORD && ly >= R && ly < R + kBlurTileH && y >= a.row0 && y < a.row1) {
                const uint32_t p = (uint32_t)y * (uint32_t)a.w2 + (uint32_t)x;
                a.e.mask_h[p] = (uint16_t)o.mask;
                a.e.total_h[p] = o.total;
            }
        }
    }
    blk.sync();
    // A vertical tap outside the map tests the normal texel CLAMP gives it and the BORDER depth (blur_fetch), not the edge row's:
    // re-stage those entries of the centre columns (the horizontal sweep is done with them).
    if (a.y0 - R < 0 || a.y0 + kBlurTileH + R > a.h2) {
        for (int k = tid; k < kBlurTileW * SH; k += n) {
            const int ly = k >> 6, lx = k & 63, x = a.x0 + lx, y = a.y0 - R + ly;
            if (x < a.w2 && (y < 0 || y >= a.h2)) {
                const BlurTap t = blur_fetch(a.e, a.in, a.borderZ, a.w2, a.h2, x, y);
                s_nz[ly * SW + lx + R] = f4a{ t.n.x, t.n.y, t.n.z, t.z };
            }
        }
        blk.sync();
    }
    // vertical sweep (gHorizontalBlur = 0, Ssao.cpp:241) of the tile
    for (int k = tid; k < kBlurTileW * kBlurTileH; k += n) {
        const int ty = k >> 6, lx = k & 63, x = a.x0 + lx, y = a.y0 + ty;
        if (x < a.w2 && y >= a.row0 && y < a.row1) {
            const BlurOut o = blur_pixel_full(a.w, [&](int i) {
                const f4a q = s_nz[(ty + i) * SW + lx + R];
                return BlurTap{ f3{ q.x, q.y, q.z }, q.w, s_mid[(ty + i) * kBlurTileW + lx] };
            });
            const uint32_t p = (uint32_t)y * (uint32_t)a.w2 + (uint32_t)x;
            a.out[p] = (uint16_t)o.value;
            if (RECORD) { a.e.mask_v[p] = (uint16_t)o.mask; a.e.total_v[p] = o.total; }
        }
    }
}

// Entries of rows [r0, r1) x columns [c0, c1) of a staged region that lie outside the map take the value of the map position
// CLAMP addressing gives them (which lies inside the same ranges: the ranges always contain the tile).
CRY_HD void blur_region_replicate(int tid, int n, float* buf, int RW, int xb, int yb, int w2, int h2, int c0, int c1, int r0, int r1)
{
    const int cw = c1 - c0, total = cw * (r1 - r0);
    const float rcw = 1.0f / (float)cw;
    for (int i = tid; i < total; i += n) {
        int ry, rx;
        divmod_small(i, cw, rcw, ry, rx);
        const int lx = c0 + rx, ly = r0 + ry, xi = xb + lx, yi = yb + ly;
        if ((uint32_t)xi >= (uint32_t)w2 || (uint32_t)yi >= (uint32_t)h2)
            buf[ly * RW + lx] = buf[(clampi(yi, 0, h2 - 1) - yb) * RW + clampi(xi, 0, w2 - 1) - xb];
    }
}

// Iterations 1 .. k of the blur (k <= kBlurMaxFused) for one tile, replaying the decisions blur_pair_tile<true> recorded.
// stamp != 0: tiles flagged with it by that launch are settled (65535 in `in` and in `out` already) and return at once.
// s0, s1: (64 + 10 k) * (16 + 10 k) floats each.
template <class Block>
CRY_HD void blur_replay_fused_tile(const Block& blk, const BlurTileArgs& a, int k, uint32_t stamp, bool onesShortcut, float* s0, float* s1)
{
    if (stamp != 0u && a.e.tiles[a.tileIndex] == stamp) return;
    const int tid = blk.tid(), n = blk.size();
    const int A = kBlurRadius * k, RW = kBlurTileW + 2 * A, RH = kBlurTileH + 2 * A;
    const int xb = a.x0 - A, yb = a.y0 - A;
    const float rRW = 1.0f / (float)RW;
    bool allOne = onesShortcut;
    for (int i = tid; i < RW * RH; i += n) {
        int ly, lx;
        divmod_small(i, RW, rRW, ly, lx);
        const int cx = clampi(xb + lx, 0, a.w2 - 1), cy = clampi(yb + ly, 0, a.h2 - 1);     // ambient: point / CLAMP
        const uint32_t raw = a.in[(uint32_t)cy * (uint32_t)a.w2 + (uint32_t)cx];
        allOne = allOne && raw == 0xFFFFu;
        s0[i] = unorm16_to_float(raw);
    }
    // A window whose ambient values are all 1.0 blurs to exactly 1.0 whatever the recorded decisions are: the colour sum adds the
    // very weights the recorded total was built from, in the same order, so colour == total bit for bit and x * rcp(x) quantises
    // to 65535 for the finite positive totals that finite positive weights give (onesShortcut, checked by the launcher).
    if (blk.all(allOne)) {
        blur_tile_fill_ones(tid, n, a);
        return;
    }
    const bool edge = xb < 0 || xb + RW > a.w2 || yb < 0 || yb + RH > a.h2;
    float* cur = s0;
    float* nxt = s1;
    for (int j = 1; j <= k; ++j) {
        const int ih = kBlurRadius * (j - 1), iv = kBlurRadius * j;      // insets before / after this iteration
        const int cw = RW - 2 * iv;
        const float rcw = 1.0f / (float)cw;
        // horizontal sweep: columns [iv, RW - iv), rows [ih, RH - ih)     cur -> nxt
        for (int i = tid; i < cw * (RH - 2 * ih); i += n) {
            int ry, rx;
            divmod_small(i, cw, rcw, ry, rx);
            const int lx = iv + rx, ly = ih + ry, xi = xb + lx, yi = yb + ly;
            if ((uint32_t)xi < (uint32_t)a.w2 && (uint32_t)yi < (uint32_t)a.h2) {
                const uint32_t p = (uint32_t)yi * (uint32_t)a.w2 + (uint32_t)xi;
                const float* row = cur + ly * RW + lx - kBlurRadius;
                const uint32_t v = blur_pixel_replay(a.w, a.e.mask_h[p], a.e.total_h[p], [&](int t) { return row[t]; });
                nxt[ly * RW + lx] = unorm16_to_float(v);
            }
        }
        blk.sync();
        if (edge) {
            blur_region_replicate(tid, n, nxt, RW, xb, yb, a.w2, a.h2, iv, RW - iv, ih, RH - ih);
            blk.sync();
        }
        // vertical sweep: columns [iv, RW - iv), rows [iv, RH - iv)       nxt -> cur, or -> out for the last iteration (the tile)
        for (int i = tid; i < cw * (RH - 2 * iv); i += n) {
            int ry, rx;
            divmod_small(i, cw, rcw, ry, rx);
            const int lx = iv + rx, ly = iv + ry, xi = xb + lx, yi = yb + ly;
            if ((uint32_t)xi < (uint32_t)a.w2 && (uint32_t)yi < (uint32_t)a.h2) {
                const uint32_t p = (uint32_t)yi * (uint32_t)a.w2 + (uint32_t)xi;
                const float* col = nxt + (ly - kBlurRadius) * RW + lx;
                const uint32_t v = blur_pixel_replay(a.w, a.e.mask_v[p], a.e.total_v[p], [&](int t) { return col[t * RW]; });
                if (j == k) {
                    if (yi >= a.row0 && yi < a.row1) a.out[p] = (uint16_t)v;
                } else {
                    cur[ly * RW + lx] = unorm16_to_float(v);
                }
            }
        }
        if (j < k) {
            blk.sync();
            if (edge) {
                blur_region_replicate(tid, n, cur, RW, xb, yb, a.w2, a.h2, iv, RW - iv, iv, RH - iv);
                blk.sync();
            }
        }
    }
}

}  // namespace cry
