// blur_tiles.hpp -- workgroup-level bodies of the blur launches of Ssao::ComputeSsao (Ssao.cpp:231-293 issues
// 2 * blurCount draws of Shaders/SsaoBlur.hlsl:85-146; here one launch per iteration):
//
//   blur_pair_tile    iteration 0: the horizontal AND the vertical sweep of one 64 x 16 tile.  The horizontal results of the
//                     tile's rows plus a 5-row apron stay in LDS (quantised to R16_UNORM and decoded again exactly as the round
//                     trip through the ambient map does) and feed the vertical sweep; both sweeps record their tap decisions
//                     (ssao_core.hpp EdgePlane::mask_*).
//   blur_replay_tile  a later iteration of one tile, both sweeps, replaying the recorded decisions: ambient values and masks of
//                     the tile + apron are staged in one burst of loads, the sweeps touch only LDS, and rows whose whole window
//                     is 1.0 skip their taps.
//
// Tiles whose whole neighbourhood came out of the SSAO pass as 65535 ("unoccluded tiles", ssao_core.hpp) are settled by the first
// launch (it writes 65535 and a per-tile flag) and cost the later launches one scalar load.
//
// The bodies are host+device text over a `Block` (thread id, block size, barrier, block-wide vote, wavefronts): the kernels
// instantiate them with the real workgroup (kernels.hip BlockDev), tests/hostsim runs the very same text with one sequential
// "thread" (BlockSeq) against the oracle's sweep-by-sweep chain.  Every output bit equals the per-sweep kernels': same per-pixel
// functions (blur_pixel_full / blur_pixel_replay), same operands, same order.
#pragma once
#include "ssao_core.hpp"

namespace cry {

constexpr int kBlurTileW = 64, kBlurTileH = 16, kBlurRadius = 5;
constexpr int kBlurPairSW = kBlurTileW + 2 * kBlurRadius;               // staged width: 74
constexpr int kBlurPairSH = kBlurTileH + 2 * kBlurRadius;               // staged height: 26

// ---- the launch plan of Ssao::ComputeSsao's blur chain (shared by api.cpp and tests/hostsim) -----------------------------------
// One launch per iteration, each reading one ambient plane and writing the other (never in place: a tile's apron is its
// neighbours' output): iteration 0 records the tap decisions, the others replay them.  The final map has to be ambient0
// (Ssao.cpp:75-78), so the planes alternate backwards from there: with an odd blurCount the SSAO pass itself writes ambient1.
// Tiles the first launch settles as unoccluded hold 65535 in BOTH planes from then on (the SSAO pass wrote one, the first launch
// the other), so later launches neither read nor write them.
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
CRY_HD int blur_chain_launches(int blurCount) { return blurCount > 0 ? blurCount : 0; }
CRY_HD int blur_chain_ssao_plane(int blurCount) { return blur_chain_launches(blurCount) & 1; }      // 0 = ambient0, 1 = ambient1
struct BlurStep {
    int in, out;               // plane indices
    uint32_t row0, rows;       // rows of `out` the launch owes
};
CRY_HD BlurStep blur_chain_step(int blurCount, uint32_t row0, uint32_t rows, uint32_t h2, int i)      // i = iteration, 0 .. blurCount - 1
{
    BlurStep s;
    s.in = (blur_chain_ssao_plane(blurCount) + i) & 1;
    s.out = s.in ^ 1;
    const int itersAfter = blurCount - 1 - i;
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

// One sequential "thread" standing in for the workgroup (host builds: tests/hostsim).  Besides the flat view (tid / size) a
// block is a set of WAVEFRONTS that each take whole rows of a tile: lanes(f) runs f(l) for the wavefront's lanes l = 0 .. 63
// (the host loops over them, accumulating into whatever f captures; a device lane runs its own), wave_all(p) is the AND of p
// over the wavefront's lanes.
struct BlockSeq {
    static constexpr int kLanes = 64;         // lanes whose private values one host "thread" has to hold
    CRY_HD int tid() const { return 0; }
    CRY_HD int size() const { return 1; }
    CRY_HD void sync() const {}
    CRY_HD bool all(bool p) const { return p; }
    CRY_HD bool any(bool p) const { return p; }
    CRY_HD int wave() const { return 0; }
    CRY_HD int waves() const { return 1; }
    template <class F> CRY_HD void lanes(F f) const { for (int l = 0; l < 64; ++l) f(l); }
    CRY_HD bool wave_all(bool p) const { return p; }
    CRY_HD bool wave_leader() const { return true; }
    CRY_HD void wave_sync() const {}         // device: orders a wavefront's LDS writes before its other lanes' reads of them
    // accesses to the ambient planes of a replay iteration (kernels.hip: device-coherent in the single-launch chain)
    CRY_HD uint32_t amb_load(const uint16_t* p) const { return *p; }
    CRY_HD void amb_store(uint16_t* p, uint32_t v) const { *p = (uint16_t)v; }
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
// RECORD: also store both sweeps' tap decisions and the tile's flag.  stamp != 0: the unoccluded-tile exit may be
// taken -- the SSAO pass of THIS frame wrote the unoccluded-wavefront map for half-res rows [ssaoRow0, ssaoRow1) with that stamp
// (a word of the map is only ever looked at inside those rows, where it was written this frame: stale contents cannot matter).
// s_nz: kBlurPairSW * kBlurPairSH entries; s_a: the same count; s_mid: kBlurTileW * kBlurPairSH; s_hmask: kBlurTileW * kBlurTileH.
template <bool RECORD, class Block>
CRY_HD void blur_pair_tile(const Block& blk, const BlurTileArgs& a, uint32_t stamp, int onesMargin, int ssaoRow0, int ssaoRow1,
                           f4a* s_nz, float* s_a, float* s_mid, uint16_t* s_hmask)
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
                if (RECORD) a.e.masks[p] = (1u << 5) | (1u << 21);
            }
        }
        return;
    }
    // stage normal + depth + ambient of columns x0 - 5 .. x0 + 68, rows y0 - 5 .. y0 + 20 (rows CLAMPed into the map: a vertical
    // tap above / below the map reads the edge row's horizontal result)
    // -- in batches whose loads are all issued (from CLAMPed, always valid positions) before the first is decoded
    constexpr int U = 4;
    for (int base = tid; base < SW * SH; base += n * U) {
        BlurTapRaw raw[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = base + u * n < SW * SH ? base + u * n : SW * SH - 1;
            int ly, lx;
            divmod_small(k, SW, 1.0f / (float)SW, ly, lx);
            raw[u] = blur_fetch_raw(a.e, a.in, a.w2, a.h2, a.x0 - R + lx, clampi(a.y0 - R + ly, 0, a.h2 - 1));
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = base + u * n;
            if (k < SW * SH) {
                const BlurTap t = blur_fetch_decode(raw[u], a.borderZ);
                s_nz[k] = f4a{ t.n.x, t.n.y, t.n.z, t.z };
                s_a[k] = t.a;
            }
        }
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
            if (RECORD && ly >= R && ly < R + kBlurTileH) s_hmask[k - R * kBlurTileW] = (uint16_t)o.mask;      // joins the vertical mask below
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
            if (RECORD) a.e.masks[p] = (uint32_t)s_hmask[k] | (o.mask << 16);
        }
    }
}

// A later iteration of the blur for one tile (both sweeps), replaying the decisions blur_pair_tile<true> recorded.
// stamp != 0: tiles flagged with it by that launch are settled (65535 in `in` and in `out` already) and return early.
// A wavefront owns rows of the staged region (rows CLAMPed into the map, like blur_pair_tile): it stages them -- ambient values
// (s_in) and both sweeps' decision masks (s_mask: horizontal | vertical << 16, as recorded) of columns x0 - 5 .. x0 + 68, every load of its
// rows issued before the first is used -- runs their horizontal sweep into s_mid, and after the barrier the vertical sweep of its
// share of the tile's rows.  Rows whose whole window holds 1.0 skip their taps -- exact: a window of ones blurs to exactly 65535
// whatever the recorded decisions are (the colour sum adds the very weights the recorded total was built from, in the same order,
// so colour == total bit for bit and x * rcp(x) quantises to 65535 for the finite positive totals that finite positive weights
// give: onesShortcut, checked by the launcher; false: nothing skips).
// s_in, s_mask: kBlurPairSW * kBlurPairSH words; s_mid: kBlurTileW * kBlurPairSH; s_rows: 8 words.
constexpr int kBlurMaxWaves = 8;
template <class Block>
CRY_HD void blur_replay_tile(const Block& blk, const BlurTileArgs& a, uint32_t stamp, bool onesShortcut, float* s_in, uint32_t* s_mask,
                             float* s_mid, uint32_t* s_rows)
{
    constexpr int SW = kBlurPairSW, SH = kBlurPairSH, R = kBlurRadius, MAXR = 4;       // up to 4 staged rows per wavefront (8 wavefronts)
    const int wv = blk.wave(), nw = blk.waves();
    if (stamp != 0u && a.e.tiles[a.tileIndex] == stamp) return;      // (fetching speculatively past this test was measured: 1.4x slower)
    uint32_t onesRows = 0u;                 // bit ly: the horizontal result of staged row ly is all ones (this wavefront's rows)
    for (int base = wv; base < SH; base += nw * MAXR) {
        // stage up to MAXR rows: lanes 0 .. 63 take columns 0 .. 63, lanes 0 .. 9 also columns 64 .. 73
        bool ones[MAXR] = { true, true, true, true };
        uint32_t raw[MAXR][2][Block::kLanes], mk[MAXR][2][Block::kLanes];         // per lane (kLanes = 1 on the device)
        blk.lanes([&](int l) {
            const int s = l % Block::kLanes;
#pragma unroll
            for (int m = 0; m < MAXR; ++m) {
                const int ly = base + m * nw < SH ? base + m * nw : SH - 1;
                const int cy = clampi(a.y0 - R + ly, 0, a.h2 - 1);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int lx = h ? 64 + (l < 10 ? l : 9) : l;
                    const uint32_t p = (uint32_t)cy * (uint32_t)a.w2 + (uint32_t)clampi(a.x0 - R + lx, 0, a.w2 - 1);   // ambient: point / CLAMP
                    raw[m][h][s] = blk.amb_load(a.in + p);
                    mk[m][h][s] = a.e.masks[p];
                }
            }
        });
        blk.lanes([&](int l) {
            const int s = l % Block::kLanes;
#pragma unroll
            for (int m = 0; m < MAXR; ++m) {
                const int ly = base + m * nw;
                if (ly < SH) {
                    s_in[ly * SW + l] = unorm16_to_float(raw[m][0][s]);
                    s_mask[ly * SW + l] = mk[m][0][s];
                    bool o = raw[m][0][s] == 0xFFFFu;
                    if (l < 10) {
                        s_in[ly * SW + 64 + l] = unorm16_to_float(raw[m][1][s]);
                        s_mask[ly * SW + 64 + l] = mk[m][1][s];
                        o = o && raw[m][1][s] == 0xFFFFu;
                    }
                    ones[m] = ones[m] && o;
                }
            }
        });
        // horizontal sweep (gHorizontalBlur = 1, Ssao.cpp:240) of the same rows: the wavefront reads what it wrote itself
        blk.wave_sync();
#pragma unroll
        for (int m = 0; m < MAXR; ++m) {
            const int ly = base + m * nw;
            if (ly >= SH) continue;
            const bool skip = onesShortcut && blk.wave_all(ones[m]);
            bool outOnes = true;
            blk.lanes([&](int l) {
                if (a.x0 + l < a.w2) {
                    float v = 1.0f;
                    if (!skip) {
                        const uint32_t mh = s_mask[ly * SW + l + R] & 0xFFFFu;
                        const float* row = s_in + ly * SW + l;
                        v = unorm16_to_float(blur_pixel_replay(a.w, mh, [&](int i) { return row[i]; }));
                    }
                    s_mid[ly * kBlurTileW + l] = v;
                    outOnes = outOnes && v == 1.0f;
                }
            });
            if (skip || (onesShortcut && blk.wave_all(outOnes))) onesRows |= 1u << ly;
        }
    }
    if (blk.wave_leader() && wv < kBlurMaxWaves) s_rows[wv] = onesRows;
    blk.sync();
    uint32_t hOnes = 0u;
    for (int w = 0; w < nw && w < kBlurMaxWaves; ++w) hOnes |= s_rows[w];
    // vertical sweep (gHorizontalBlur = 0, Ssao.cpp:241) of the tile
    for (int ty = wv; ty < kBlurTileH; ty += nw) {
        const int y = a.y0 + ty;
        if (y < a.row0 || y >= a.row1) continue;
        const bool skip = ((hOnes >> ty) & 0x7FFu) == 0x7FFu;          // staged rows ty .. ty + 10 = taps y - 5 .. y + 5
        blk.lanes([&](int l) {
            const int x = a.x0 + l;
            if (x < a.w2) {
                uint32_t v = 0xFFFFu;
                if (!skip) {
                    const uint32_t mv = s_mask[(ty + R) * SW + l + R] >> 16;
                    const float* col = s_mid + ty * kBlurTileW + l;
                    v = blur_pixel_replay(a.w, mv, [&](int i) { return col[i * kBlurTileW]; });
                }
                blk.amb_store(a.out + ((uint32_t)y * (uint32_t)a.w2 + (uint32_t)x), v);
            }
        });
    }
}

}  // namespace cry
