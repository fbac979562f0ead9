// hostsim.cpp -- TEST HARNESS ONLY.  Compiles the product's per-pixel kernel bodies (csrc/*_core.hpp, the very
// text the gfx950 kernels inline) for the host CPU so that the CPU-only test tier can compare them bit for bit
// with the oracle before any GPU time is spent.  Nothing in the product loads this library.
#include <cstring>
#include "ssao_core.hpp"
#include "blur_tiles.hpp"
#include "light_core.hpp"
#include "raster_core.hpp"
#include <vector>

using namespace cry;

extern "C" {

float hs_d24_to_float(uint32_t v) { return d24_to_float(v); }
float hs_unorm16_to_float(uint32_t v) { return unorm16_to_float(v); }
float hs_unorm8_to_float(uint32_t v) { return unorm8_to_float(v); }
float hs_half_to_float(uint16_t v) { return half_to_float(v); }
float hs_det_sin(float x) { return det_sin(x); }
float hs_det_cos(float x) { return det_cos(x); }
float hs_det_log2(float x) { return det_log2(x); }
float hs_det_exp2(float x) { return det_exp2(x); }
float hs_det_pow(float x, float y) { return det_pow(x, y); }
float hs_nrand(float u, float v) { return nrand(u, v); }

// d24_threshold against its definition, for every D24 value t and the references decode(t), the float below and the float above:
// returns the number of mismatches (0 expected).  Also probes the out-of-range references.
uint32_t hs_check_d24_threshold(void)
{
    uint32_t bad = 0;
    auto expect = [](float ref, uint32_t near) {
        uint32_t lo = near > 8u ? near - 8u : 0u;
        uint32_t T = lo;
        while (T <= 0x00FFFFFFu && d24_to_float(T) < ref) ++T;          // monotone: the first value that does not decode below ref
        return T;
    };
    for (uint32_t t = 0; t <= 0x00FFFFFFu; ++t) {
        const float d = d24_to_float(t);
        const float refs[3] = { d, __builtin_nextafterf(d, 2.0f), __builtin_nextafterf(d, -1.0f) };
        for (float r : refs) bad += d24_threshold(r) != expect(r, t) ? 1u : 0u;
    }
    const float special[] = { -1.0f, -0.0f, 0.0f, 1.0f, 1.0000001f, 2.0f, 1.0e30f, -1.0e30f, __builtin_inff(), -__builtin_inff() };
    for (float r : special) {
        uint32_t T = 0;
        while (T <= 0x00FFFFFFu && d24_to_float(T) < r) { if (r > 1.5f) { T = 0x01000000u; break; } ++T; }
        bad += d24_threshold(r) != T ? 1u : 0u;
    }
    bad += d24_threshold(__builtin_nanf("")) != 0x01000000u ? 1u : 0u;
    return bad;
}

// kind: 0 sin, 1 cos, 2 log2, 3 exp2, 4 pow, 5 nrand, 6 d24, 7 unorm16, 8 unorm8, 9 half, 10 pow(., 1/2.2) (same numbering as or_eval_array);
// 11: pow(., 1/2.2) through the packed form
void hs_eval_array(int kind, size_t n, const float* in, const float* in2, float* out)
{
    const uint32_t* bits = (const uint32_t*)in;
    for (size_t i = 0; i < n; ++i) {
        switch (kind) {
        case 0: out[i] = det_sin(in[i]); break;
        case 1: out[i] = det_cos(in[i]); break;
        case 2: out[i] = det_log2(in[i]); break;
        case 3: out[i] = det_exp2(in[i]); break;
        case 4: out[i] = det_pow(in[i], in2[i]); break;
        case 5: out[i] = nrand(in[i], in2[i]); break;
        case 6: out[i] = d24_to_float(bits[i]); break;
        case 7: out[i] = unorm16_to_float(bits[i]); break;
        case 8: out[i] = unorm8_to_float(bits[i]); break;
        case 9: out[i] = half_to_float((uint16_t)bits[i]); break;
        case 10: out[i] = pow_inv_gamma(in[i]); break;
        case 11: out[i] = pow_inv_gamma2(v2f{ in2[i], in[i] }).y; break;      // the packed form's second lane
        default: out[i] = 0.0f;
        }
    }
}

static uint32_t g_sky_waves = 0;     // wavefronts the last hs_ssao_path call resolved through the sky shortcut
uint32_t hs_last_sky_waves(void) { return g_sky_waves; }
static uint32_t g_culled_taps = 0;   // taps the last hs_ssao_path call skipped through the nearest-depth map
uint32_t hs_last_culled_taps(void) { return g_culled_taps; }
static uint32_t g_clear_cell_taps = 0;   // taps of that call that were NOT culled and landed in a clear cell (footprint = 1.0 by rule)
uint32_t hs_last_clear_cell_taps(void) { return g_clear_cell_taps; }
static uint16_t* g_cull_masks = nullptr;      // optional: per half-res pixel, bit i = tap i culled (analysis only)
void hs_set_cull_mask_plane(uint16_t* plane) { g_cull_masks = plane; }
// The frame stamp the next hs_ssao_path / hs_blur_chain calls run with (api.cpp draws a fresh one per frame from a process-wide
// counter).  Like the device, nothing here clears the workspace: whatever the caller left in it is what a recycled allocation holds.
static uint32_t g_stamp = 1u;
void hs_set_stamp(uint32_t stamp) { g_stamp = stamp; }

// margin of the depth pass around the rows of the call, in texel rows (launch_depth_pairs / depth_pass_cell_rows); < 0: the
// product's default.  Everything the pass does not visit is poisoned, so a tap that wrongly trusted it would show.
static int g_prep_margin = -1;
void hs_set_prep_margin(int margin) { g_prep_margin = margin; }
static uint32_t g_slow_taps = 0;
uint32_t hs_last_unprepared_rows(void) { return g_slow_taps; }

// use_pairs != 0: the product's default path -- build the decoded depth-pairs plane in the edge workspace (the body of
// depth_pairs_kernel) and let the taps gather from it; 0: taps on the raw D24 plane (what runs without a workspace).
void hs_ssao_path(const crychic_ssao_constants* cb, const void* normal, const uint32_t* depth, const uint8_t* randvec,
                  uint16_t* ambient, void* edge_base, uint32_t W, uint32_t H, uint32_t row0, uint32_t rows, int use_pairs)
{
    const uint32_t w2 = W / 2;
    EdgePlane e{};
    if (edge_base) e = edge_plane_carve(edge_base, W, H);
    const u2* nrm = (const u2*)normal;
    const bool pairs = use_pairs && edge_base;
    // the cell rows the depth pass visits for these rows
    uint32_t c0 = 0, cn = zmin_map_rows(H);
    if (pairs) depth_pass_cell_rows(H, row0, rows, &c0, &cn, g_prep_margin);
    const bool limited = pairs && (c0 > 0u || c0 + cn < zmin_map_rows(H));
    const int j0lo = 8 * (int)c0 - 2;
    const uint32_t nj = 8u * cn;
    g_slow_taps = limited ? 8u * (zmin_map_rows(H) - cn) : 0u;
    // the nearest-depth map as depth_pairs_kernel fills it: per cell of 9 x 9 padded texels, cells 8 apart (padded (ex, ey) = texel
    // (ex - 2, ey - 2), anything outside the plane reads as the clear depth)
    const CullParams cp = ssao_cull_params(*cb);
    const bool culling = pairs && cp.enabled && use_pairs != 2;         // use_pairs == 2: pairs plane without tap culling
    if (pairs) {
        for (uint32_t cy = 0; cy < zmin_map_rows(H); ++cy)
            for (uint32_t cx = 0; cx < zmin_map_cols(W); ++cx) {
                float m = 1.0f;
                for (uint32_t ey = 8u * cy; ey <= 8u * cy + 8u; ++ey)
                    for (uint32_t ex = 8u * cx; ex <= 8u * cx + 8u; ++ex) {
                        const int tx = (int)ex - 2, ty = (int)ey - 2;
                        if ((uint32_t)tx < W && (uint32_t)ty < H) m = __builtin_fminf(m, d24_to_float(depth[(uint32_t)ty * W + (uint32_t)tx]));
                    }
                const bool visited = cy >= c0 && cy < c0 + cn;
                e.zcull[cy * zmin_map_cols(W) + cx] = visited ? zmin_cell_value(cp, m) : 1.0e30f;      // poison: "cull everything"
            }
    }
    // The pairs plane.  Entries the pass does not visit are poisoned, and so is every entry that only clear cells can reach
    // (ssao_core.hpp "clear cells": the kernel may leave those unwritten -- here ALL of them are, the kernel keeps a few)
    if (pairs) {
        f4a* out = (f4a*)const_cast<void*>(e.pairs);
        const uint32_t halfPitch = depth_pairs_pitch(W) / 2u;
        auto clear = [&](int cx, uint32_t cy) {
            return cx < 0 || ((uint32_t)cx < zmin_map_cols(W) && cy < zmin_map_rows(H) && zmin_cell_is_clear(e.zcull[cy * zmin_map_cols(W) + (uint32_t)cx]));
        };
        for (uint32_t py = 0; py < H + 3u; ++py)
            for (uint32_t px2 = 0; px2 < halfPitch; ++px2) {
                const bool visited = py >= 8u * c0 && py < 8u * (c0 + cn);
                const bool unread = culling && visited && clear((2 * (int)px2 - 1) >> 3, py >> 3) && clear((int)(px2 >> 2), py >> 3);
                out[py * halfPitch + px2] = visited && !unread ? depth_pairs_entry2(depth, W, H, 2 * (int)px2 - 2, (int)py - 2) : f4a{ -7.0f, -7.0f, -7.0f, -7.0f };
            }
    }
    const DepthPairs dp{ e.pairs, depth_pairs_pitch(W) };
    const DepthD24 dd{ depth, W, H };
    const DepthPairsRows dr{ dp, dd, j0lo, nj };
    const bool sparse = ssao_projtex_is_sparse(*cb);
    // the coarse geometry map as depth_pairs_kernel fills it (cells with geometry get the stamp, the others are left alone), and
    // the sky shortcut per "wavefront" of 64 consecutive pixels of a row, exactly as ssao_kernel takes it
    SkyReach sky = ssao_sky_reach(*cb, W, H);
    if (!pairs) sky.enabled = 0;
    if (limited) {
        sky.y0 = j0lo < 0 ? 0 : j0lo;
        sky.y1 = 8 * (int)(c0 + cn) - 2 > (int)H ? (int)H : 8 * (int)(c0 + cn) - 2;
    }
    const uint32_t stamp = g_stamp, gpitch = geo_map_cols(W);
    if (pairs) {
        const uint32_t halfPitch = depth_pairs_pitch(W) / 2u;
        for (uint32_t py = 2; py < H + 2u; ++py) {
            if (py < 8u * c0 || py >= 8u * (c0 + cn)) continue;
            for (uint32_t px2 = 0; px2 < halfPitch; ++px2) {
                const f4a v = depth_pairs_entry2(depth, W, H, 2 * (int)px2 - 2, (int)py - 2);
                if (v.x != 1.0f || v.z != 1.0f) e.geo[((py - 2u) >> 5) * gpitch + px2 / 64u] = stamp;
            }
        }
    }
    const ZminMap win{ e.zcull, zmin_map_cols(W) };
    const ZminMapRows winRows{ win, j0lo, nj };
    std::vector<SsaoCentre> row(w2);
    g_sky_waves = 0;
    g_culled_taps = 0;
    g_clear_cell_taps = 0;
    const HalfResScale hs = half_res_scale(W, H);
    auto pixel = [&](uint32_t x, uint32_t y, uint32_t* acc) -> uint32_t {
        if (limited) return culling ? ssao_pixel(*cb, row[x], dr, (const uint32_t*)randvec, W, H, x, y, hs, sparse, winRows, acc)
                                    : ssao_pixel(*cb, row[x], dr, (const uint32_t*)randvec, W, H, x, y, hs, sparse);
        if (culling) return ssao_pixel(*cb, row[x], dp, (const uint32_t*)randvec, W, H, x, y, hs, sparse, win, acc);
        return pairs ? ssao_pixel(*cb, row[x], dp, (const uint32_t*)randvec, W, H, x, y, hs, sparse)
                     : ssao_pixel(*cb, row[x], dd, (const uint32_t*)randvec, W, H, x, y, hs, sparse);
    };
    for (uint32_t y = row0; y < row0 + rows; ++y) {
        for (uint32_t x = 0; x < w2; ++x) {
            const SsaoCentre c = ssao_centre(*cb, nrm, depth, W, H, (int)x, (int)y);
            row[x] = c;
            if (e.nrm) {
                e.nrm[y * w2 + x] = c.nrm_bits;
                e.vz[y * w2 + x] = c.vz;
                if (x == 0) e.gcol[y] = nrm[(2u * y + 1u) * W];
                if (y == row0) e.grow[x] = nrm[2u * x + 1u];
            }
        }
        if (!ambient) continue;
        for (uint32_t x0 = 0; x0 < w2; x0 += 64u) {
            const uint32_t n = (w2 - x0) < 64u ? (w2 - x0) : 64u;
            bool skip = sky.enabled != 0;
            for (uint32_t k = 0; k < n && skip; ++k) skip = row[x0 + k].sky;
            if (skip) {
                const GeoCells g = ssao_sky_cells(sky, W, H, x0, n, y);
                skip = g.known;
                for (uint32_t cy = g.cy0; cy <= g.cy1 && skip; ++cy)
                    for (uint32_t cx = g.cx0; cx <= g.cx1 && skip; ++cx) skip = e.geo[cy * gpitch + cx] != stamp;
            }
            g_sky_waves += skip ? 1u : 0u;
            bool allOnes = true;
            for (uint32_t x = x0; x < x0 + n; ++x) {
                uint32_t acc[3] = { 0u, 0u, 0u };
                ambient[y * w2 + x] = skip ? (uint16_t)0xFFFFu : (uint16_t)pixel(x, y, acc);
                g_culled_taps += acc[0];
                g_clear_cell_taps += acc[2];
                if (!skip && culling && g_cull_masks) g_cull_masks[y * w2 + x] = (uint16_t)(acc[1] | 0x8000u);
                allOnes = allOnes && ambient[y * w2 + x] == 0xFFFFu;
            }
            if (pairs) e.ones[y * ones_map_cols(W) + x0 / 64u] = allOnes ? stamp : 0u;      // the unoccluded-wavefront map, as ssao_kernel writes it
        }
    }
}
void hs_ssao(const crychic_ssao_constants* cb, const void* normal, const uint32_t* depth, const uint8_t* randvec,
             uint16_t* ambient, void* edge_base, uint32_t W, uint32_t H, uint32_t row0, uint32_t rows)
{
    hs_ssao_path(cb, normal, depth, randvec, ambient, edge_base, W, H, row0, rows, 1);
}

void hs_blur(const crychic_ssao_constants* cb, void* edge_base, const uint16_t* in, uint16_t* out, uint32_t W,
             uint32_t H, int horizontal, uint32_t row0, uint32_t rows)
{
    const int w2 = (int)(W / 2), h2 = (int)(H / 2);
    const EdgePlane e = edge_plane_carve(edge_base, W, H);
    const float borderZ = ndc_to_view(*cb, 1.0f);
    for (int y = (int)row0; y < (int)(row0 + rows); ++y)
        for (int x = 0; x < w2; ++x) {
            out[y * w2 + x] = (uint16_t)blur_pixel(&cb->BlurWeights[0][0], [&](int i) {
                return blur_fetch(e, in, borderZ, w2, h2, horizontal ? x + i - 5 : x, horizontal ? y : y + i - 5);
            });
        }
}

// The blur chain of Ssao::ComputeSsao as api.cpp issues it: the launch plan of blur_tiles.hpp, every launch's tiles run one after
// the other through the very tile bodies the kernels instantiate (one sequential "thread" per tile).  planes[0] / planes[1] =
// ambient0 / ambient1; the SSAO output is expected in planes[blur_chain_ssao_plane(blurCount)] for rows blur_chain_ssao_rows(..)
// (hs_ssao_path with the same stamp wrote the unoccluded-wavefront map); the result is in planes[0], rows [row0, row0 + rows).
// use_exit == 0: no unoccluded-tile exit (stamp 0), every tile takes the general path.
static uint32_t g_settled_tiles = 0;          // tiles the last hs_blur_chain settled through the unoccluded-tile exit
uint32_t hs_last_settled_tiles(void) { return g_settled_tiles; }
void hs_blur_chain(const crychic_ssao_constants* cb, void* edge_base, uint16_t* plane0, uint16_t* plane1, uint32_t W, uint32_t H, int blurCount,
                   uint32_t row0, uint32_t rows, int use_exit, int onesMargin)
{
    const uint32_t h2 = H / 2;
    const EdgePlane e = edge_plane_carve(edge_base, W, H);
    uint16_t* planes[2] = { plane0, plane1 };
    uint32_t sr0, srn;
    blur_chain_ssao_rows(blurCount, row0, rows, h2, &sr0, &srn);
    const bool positive = blur_weights_positive(*cb);        // as the launchers decide
    const uint32_t stamp = (use_exit && positive) ? g_stamp : 0u;
    std::vector<f4a> s_nz(kBlurPairSW * kBlurPairSH);
    std::vector<float> s_a(kBlurPairSW * kBlurPairSH), s_mid(kBlurTileW * kBlurPairSH);
    std::vector<uint32_t> s_mask(kBlurPairSW * kBlurPairSH), s_rows(kBlurMaxWaves);
    std::vector<uint16_t> s_hmask(kBlurTileW * kBlurTileH);
    g_settled_tiles = 0;
    for (int i = 0; i < blur_chain_launches(blurCount); ++i) {
        const BlurStep st = blur_chain_step(blurCount, row0, rows, h2, i);
        if (st.rows == 0) continue;
        const uint32_t t0 = st.row0 / (uint32_t)kBlurTileH, t1 = (st.row0 + st.rows - 1u) / (uint32_t)kBlurTileH;
        for (uint32_t ty = t0; ty <= t1; ++ty)
            for (uint32_t tx = 0; tx < blur_tiles_x(W); ++tx) {
                BlurTileArgs a;
                a.w = &cb->BlurWeights[0][0]; a.e = e; a.in = planes[st.in]; a.out = planes[st.out];
                a.w2 = (int)(W / 2); a.h2 = (int)h2; a.x0 = (int)tx * kBlurTileW; a.y0 = (int)ty * kBlurTileH;
                a.row0 = (int)st.row0; a.row1 = (int)(st.row0 + st.rows);
                a.borderZ = ndc_to_view(*cb, 1.0f);
                a.tileIndex = ty * blur_tiles_x(W) + tx;
                if (i == 0) {
                    if (blurCount > 1) blur_pair_tile<true>(BlockSeq{}, a, stamp, onesMargin, (int)sr0, (int)(sr0 + srn), s_nz.data(), s_a.data(), s_mid.data(), s_hmask.data());
                    else blur_pair_tile<false>(BlockSeq{}, a, stamp, onesMargin, (int)sr0, (int)(sr0 + srn), s_nz.data(), s_a.data(), s_mid.data(), s_hmask.data());
                    if (blurCount > 1 && stamp != 0u && e.tiles[a.tileIndex] == stamp) ++g_settled_tiles;
                } else {
                    blur_replay_tile(BlockSeq{}, a, stamp, positive, s_a.data(), s_mask.data(), s_mid.data(), s_rows.data());
                }
            }
    }
}
int hs_blur_chain_ssao_plane(int blurCount) { return blur_chain_ssao_plane(blurCount); }
void hs_blur_chain_ssao_rows(int blurCount, uint32_t row0, uint32_t rows, uint32_t h2, uint32_t* r0, uint32_t* rn) { blur_chain_ssao_rows(blurCount, row0, rows, h2, r0, rn); }

void hs_light(const crychic_pass_constants* cb, const float* g0, const float* g1, const float* g2,
              const uint32_t* depth, const uint16_t* ambient, const uint32_t* const shadow[4], uint32_t shadowDim,
              const uint8_t* cube, uint32_t cubeDim, uint8_t* out, float* radiance, uint32_t W, uint32_t H,
              uint32_t row0, uint32_t rows, int numDirLights, float pcfSearchRadius, uint32_t flags, const crychic_light* pointLights,
              uint32_t numPointLights)
{
    LightParams P;
    std::memcpy(P.ViewProjTex, cb->ViewProjTex, sizeof P.ViewProjTex);
    std::memcpy(P.ShadowTransforms, cb->ShadowTransforms, sizeof P.ShadowTransforms);
    std::memcpy(P.InvProj, cb->InvProj, sizeof P.InvProj);
    std::memcpy(P.InvView, cb->InvView, sizeof P.InvView);
    std::memcpy(P.EyePosW, cb->EyePosW, sizeof P.EyePosW);
    P.pcfSearchRadius = pcfSearchRadius;
    std::memcpy(P.AmbientLight, cb->AmbientLight, sizeof P.AmbientLight);
    std::memcpy(P.Lights, cb->Lights, sizeof P.Lights);
    for (int i = 0; i < 4; ++i) P.shadow[i] = shadow[i];
    P.shadowDim = shadowDim; P.cubeDim = cubeDim; P.W = W; P.H = H; P.numDirLights = numDirLights; P.flags = flags;
    P.pointLights = pointLights; P.numPointLights = numPointLights;
    P.shadowWIsOne = light_shadow_w_is_one(P.ShadowTransforms) ? 1u : 0u;
    P.darkLights = light_dark_mask(P.Lights, numDirLights);
    P.unitLights = light_dark_lengths_ok(P.Lights, numDirLights) ? 1u : 0u;
    P.rcpW = rcp((float)W); P.rcpH = rcp((float)H);
    P.cubeLevels = (flags >> 16) & 15u;                    // CRYCHIC_LIGHT_CUBE_LEVELS
    light_params_derive(P);
    const bool chain = P.cubeLevels > 1u;
    const AllPointLights pl{ pointLights, numPointLights };
    const f4a* G0 = (const f4a*)g0; const f4a* G1 = (const f4a*)g1; const f4a* G2 = (const f4a*)g2;
    for (uint32_t y = row0; y < row0 + rows; ++y)
        for (uint32_t x = 0; x < W; ++x) {
            const uint32_t idx = y * W + x;
            f4 lit;
            const bool fix = (flags & (CRYCHIC_FIX_Q1 | CRYCHIC_FIX_Q3 | CRYCHIC_FIX_Q4)) != 0;     // as launch_light picks the instantiation
            if (chain && (depth[idx] & 0x00FFFFFFu) < 0x00FFFFFFu) {
                // light_kernel<.., MIPS>: the neighbours' reflection vectors arrive by lane exchange there, by recomputation here
                auto shaded = [&](uint32_t xx, uint32_t yy) { return xx < W && yy < row0 + rows && (depth[yy * W + xx] & 0x00FFFFFFu) < 0x00FFFFFFu; };
                const f3 r = reflection_dir(P, G0[idx], G2[idx]);
                f3 ddx{ 0.0f, 0.0f, 0.0f }, ddy{ 0.0f, 0.0f, 0.0f };
                if (shaded(x ^ 1u, y)) { const f3 n = reflection_dir(P, G0[y * W + (x ^ 1u)], G2[y * W + (x ^ 1u)]); ddx = (x & 1u) ? f3{ r.x - n.x, r.y - n.y, r.z - n.z } : f3{ n.x - r.x, n.y - r.y, n.z - r.z }; }
                if (shaded(x, y ^ 1u)) { const f3 n = reflection_dir(P, G0[(y ^ 1u) * W + x], G2[(y ^ 1u) * W + x]); ddy = (y & 1u) ? f3{ r.x - n.x, r.y - n.y, r.z - n.z } : f3{ n.x - r.x, n.y - r.y, n.z - r.z }; }
                const float lod = cube_lod(P.cubeDim, P.cubeLevels, r, ddx, ddy);
                const CubeChain cc{ lod, cube_chain_flat(lod) };
                if (pcfSearchRadius == 0.0f) lit = light_pixel<true, AllPointLights, true, CubeChain>(P, G0[idx], G1[idx], G2[idx], ambient, (const uint32_t*)cube, pl, cc);
                else lit = light_pixel<false, AllPointLights, true, CubeChain>(P, G0[idx], G1[idx], G2[idx], ambient, (const uint32_t*)cube, pl, cc);
            }
            else if ((depth[idx] & 0x00FFFFFFu) < 0x00FFFFFFu) {
                if (pcfSearchRadius == 0.0f) lit = fix ? light_pixel<true, AllPointLights, true>(P, G0[idx], G1[idx], G2[idx], ambient, (const uint32_t*)cube, pl)
                                                       : light_pixel<true, AllPointLights, false>(P, G0[idx], G1[idx], G2[idx], ambient, (const uint32_t*)cube, pl);
                else lit = fix ? light_pixel<false, AllPointLights, true>(P, G0[idx], G1[idx], G2[idx], ambient, (const uint32_t*)cube, pl)
                               : light_pixel<false, AllPointLights, false>(P, G0[idx], G1[idx], G2[idx], ambient, (const uint32_t*)cube, pl);
            }
            else if (flags & CRYCHIC_LIGHT_SKY) lit = chain ? sky_pixel_chain(P, (const uint32_t*)cube, x, y) : sky_pixel(P, (const uint32_t*)cube, x, y);
            else lit = f4{ 0.690196097f, 0.768627524f, 0.870588303f, 1.0f };
            if (radiance) { radiance[4 * idx] = lit.x; radiance[4 * idx + 1] = lit.y; radiance[4 * idx + 2] = lit.z; radiance[4 * idx + 3] = lit.w; }
            ((uint32_t*)out)[idx] = pack_rgba8(lit);
        }
}

// The producer passes executed sequentially on the host with the kernels' own bodies (raster_core.hpp): setup in draw
// order into the same slot numbering as setup_kernel, coverage by min() on the 64-bit key, then the resolve stage.
int hs_rasterize(int mode, const float* view, const float* viewProj, const crychic_draw_item* items, uint32_t nItems,
                 const crychic_material_data* materials, uint32_t nMaterials, const crychic_texture* textures, uint32_t nTextures,
                 uint32_t W, uint32_t H, int depthBias, float slopeBias, uint32_t* depth, uint16_t* normal, float* g0, float* g1, float* g2)
{
    std::vector<SetupTri> tris;
    bool overflow = false;
    for (uint32_t it = 0; it < nItems; ++it) {
        const crychic_draw_item& d = items[it];
        const uint32_t ntri = d.indexCount / 3u;
        for (uint32_t inst = 0; inst < d.instanceCount; ++inst)
            for (uint32_t tri = 0; tri < ntri; ++tri) {
                const crychic_instance_data& I = d.instances_dev[inst];
                const crychic_material_data* M = (materials && I.MaterialIndex < nMaterials) ? &materials[I.MaterialIndex] : nullptr;
                const size_t slot0 = tris.size();
                tris.resize(slot0 + kSlotsPerTriangle);
                for (int c = 0; c < kSlotsPerTriangle; ++c) tris[slot0 + (size_t)c].A2 = 0;
                VsOut poly[kMaxPolyVerts], tmp[kMaxPolyVerts];
                for (int c = 0; c < 3; ++c) {
                    const int64_t vi = (int64_t)d.indices_dev[d.startIndexLocation + tri * 3u + c] + d.baseVertexLocation;
                    if (vi < 0 || vi >= (int64_t)d.vertexCount) return -1;
                    poly[c] = vertex_shader(d.vertices_dev[vi], I, M, viewProj);
                }
                const int n = clip_triangle(poly, tmp, W, H);
                for (int c = 1; c + 1 < n; ++c) {
                    SetupTri s;
                    if (setup_triangle(poly[0], poly[c], poly[c + 1], I.MaterialIndex, W, H, s, &overflow)) tris[slot0 + (size_t)(c - 1)] = s;
                }
            }
    }
    if (overflow) return -2;
    std::vector<uint64_t> vis((size_t)W * H, kVisClear);
    int live = 0;
    for (size_t slot = 0; slot < tris.size(); ++slot) {
        const SetupTri& t = tris[slot];
        if (t.A2 <= 0) continue;
        ++live;
        const PixelBox b = triangle_box(t, W, H);
        const EdgeFlags e = triangle_edge_flags(t);
        const double bias = mode == 0 ? triangle_depth_bias(t, depthBias, slopeBias) : 0.0;
        for (int y = b.y0; y <= b.y1; ++y)
            for (int x = b.x0; x <= b.x1; ++x) {
                const uint64_t key = fragment_key(t, e, bias, x, y, (uint32_t)slot + 1u);
                uint64_t& v = vis[(size_t)y * W + x];
                if (key < v) v = key;
            }
    }
    const Texture* tex = reinterpret_cast<const Texture*>(textures);
    for (uint32_t y = 0; y < H; ++y)
        for (uint32_t x = 0; x < W; ++x) {
            const size_t idx = (size_t)y * W + x;
            const uint64_t key = vis[idx];
            const uint32_t serial = (uint32_t)(key & 0xFFFFFFFFull);
            depth[idx] = (uint32_t)(key >> 32);
            if (mode == 0) continue;
            if (serial == 0) {
                if (mode == 1) { normal[idx * 4] = 0; normal[idx * 4 + 1] = 0; normal[idx * 4 + 2] = 0x3C00; normal[idx * 4 + 3] = 0; }
                else for (int c = 0; c < 4; ++c) { g0[idx * 4 + c] = 0; g1[idx * 4 + c] = 0; g2[idx * 4 + c] = 0; }
                continue;
            }
            const ResolveOut r = resolve_pixel(mode, tris[serial - 1u], (int)x, (int)y, view, materials, nMaterials, tex, nTextures);
            if (mode == 1) {
                normal[idx * 4] = float_to_half(r.normalV.x); normal[idx * 4 + 1] = float_to_half(r.normalV.y);
                normal[idx * 4 + 2] = float_to_half(r.normalV.z); normal[idx * 4 + 3] = 0;
            } else {
                const float a[12] = { r.g0.x, r.g0.y, r.g0.z, r.g0.w, r.g1.x, r.g1.y, r.g1.z, r.g1.w, r.g2.x, r.g2.y, r.g2.z, r.g2.w };
                for (int c = 0; c < 4; ++c) { g0[idx * 4 + c] = a[c]; g1[idx * 4 + c] = a[4 + c]; g2[idx * 4 + c] = a[8 + c]; }
            }
        }
    return live;
}

}  // extern "C"
