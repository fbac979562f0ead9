// api.cpp -- the extern "C" surface of libcrychic_hip.so (include/crychic_hip.h): argument validation, the
// stream-ordered pass sequencing of Ssao::ComputeSsao (Ssao.cpp:185-243) and of the hot part of
// CRYCHIC::Draw (CRYCHIC.cpp:220-221,238-279), and per-pass HIP-event timing.  No CPU compute path exists
// here: every compute entry point needs a context bound to a HIP device.
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#include <cstring>
#include <utility>
#include <new>
#include "crychic_hip.h"
#include "kernels.hpp"
#include "light_core.hpp"
#include "ssao_core.hpp"
#include "blur_tiles.hpp"
#include "raster_core.hpp"
#include "internal.hpp"

namespace {
thread_local char g_err[512] = "";
}

namespace cry {
// shared with comm.cpp (internal.hpp): records the thread-local message behind crychic_last_error()
int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
}  // namespace cry

namespace {
using cry::fail;

#define CRY_HIP(expr)                                                                                  \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) return fail(CRYCHIC_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

int check_dims(uint32_t W, uint32_t H)
{
    if (W == 0 || H == 0 || (W & 1u) || (H & 1u))
        return fail(CRYCHIC_E_INVALID_ARG, "frame size %ux%u must be non-zero and even (half-res maps are W/2 x H/2)", W, H);
    // 2^28 pixels: 32-bit plane offsets; 4 * 65535 rows: one workgroup row of the lighting pass per 4 pixel rows, grid.y <= 65535
    if ((uint64_t)W * H > (1ull << 28) || W >= (1u << 20) || H > 4u * 65535u)
        return fail(CRYCHIC_E_UNSUPPORTED, "frame %ux%u exceeds 2^28 pixels, 2^20 columns or 262140 rows", W, H);
    return 0;
}

}  // namespace


namespace {

int bind(crychic_ctx* ctx)
{
    if (!ctx) return fail(CRYCHIC_E_INVALID_ARG, "null context");
    CRY_HIP(hipSetDevice(ctx->device));
    return 0;
}

// Frame stamps mark what the SSAO pass of one frame wrote into the (caller-owned) edge workspace.  They come from one
// process-wide counter, so two contexts -- or a context re-created over recycled memory -- never issue the same value.
std::atomic<uint32_t> g_frameStamp{ 0x5EED0000u };
uint32_t next_stamp()
{
    uint32_t s;
    do { s = g_frameStamp.fetch_add(1u, std::memory_order_relaxed) + 1u; } while (s == 0u);      // never 0
    return s;
}

using cry::clamp_rows;

int ssao_compute_impl(const crychic_ssao_constants* cb, const void* normal, const uint32_t* depth,
                      const uint8_t* randvec, uint16_t* ambient0, uint16_t* ambient1, void* edge, uint32_t W,
                      uint32_t H, int blurCount, uint32_t row0, uint32_t rows, hipStream_t stream, hipEvent_t afterSsao, crychic_ctx* ctx)
{
    const uint32_t h2 = H / 2;
    if (row0 > h2 || rows > h2 - row0) return fail(CRYCHIC_E_INVALID_ARG, "rows [%u,+%u) outside the %u-row ambient map", row0, rows, h2);
    if (blurCount < 0) return fail(CRYCHIC_E_INVALID_ARG, "blurCount %d < 0", blurCount);
    if (blurCount > 0 && (!ambient1 || !edge)) return fail(CRYCHIC_E_INVALID_ARG, "blurCount > 0 needs ambient1 and the edge workspace");
    // rows, planes and launches: blur_tiles.hpp "the launch plan"
    uint32_t r0, rn;
    cry::blur_chain_ssao_rows(blurCount, row0, rows, h2, &r0, &rn);
    uint16_t* planes[2] = { ambient0, ambient1 };
    // With a workspace at hand the depth plane is re-laid once per frame as decoded pairs -- for the rows of this call and a margin:
    // the taps of a row reach far up and down the frame, and the few that leave the margin take the raw plane -- and
    // the taps gather from that; its cost is part of the SSAO pass.
    const uint32_t stamp = next_stamp();
    // (gathering from the raw D24 plane with the coarse maps alone was measured too: the depth pass drops from 20 to 7 us and the
    // SSAO kernel gains 15: the pairs plane stays)
    // A small strip of a multi-GPU frame (few wavefronts: its SSAO pass is a chain of round trips, not arithmetic) skips the pairs plane:
    // the depth pass writes the coarse maps only and the taps gather from the raw plane -- the same bits, 1.6 us less for a 1/8 strip of
    // the 4K frame, where the whole frame gains 8 us from the plane (profiles/r04_experiments.txt).  CRYCHIC_STRIP_PAIRS=1: always the plane.
    static const bool stripPairs = getenv("CRYCHIC_STRIP_PAIRS") != nullptr;
    const bool smallStrip = rn < h2 && ((W / 2u + 63u) / 64u) * rn < 4096u;
    const bool usePairs = edge != nullptr && (stripPairs || !smallStrip);
    if (edge) CRY_HIP(cry::launch_depth_pairs(*cb, depth, edge, W, H, stamp, usePairs, r0, rn, stream));
    CRY_HIP(cry::launch_ssao(*cb, normal, depth, randvec, planes[cry::blur_chain_ssao_plane(blurCount)], edge, W, H, r0, rn, true, usePairs,
                             edge ? stamp : 0u, stream));
    if (afterSsao) CRY_HIP(hipEventRecord(afterSsao, stream));
    static const bool perIteration = getenv("CRYCHIC_BLUR_PER_ITERATION") != nullptr;      // the round-3 launch plan, kept for A / B runs
    for (int i = 0; i < cry::blur_chain_launches(blurCount); ++i) {
        const cry::BlurStep s = cry::blur_chain_step(blurCount, row0, rows, h2, i);
        if (i == 0)      // a pixel's value after the frame's sweeps depends on inputs within 5 pixels per iteration
            CRY_HIP(cry::launch_blur_pair(*cb, edge, planes[s.in], planes[s.out], W, H, s.row0, s.rows, blurCount > 1, stamp, 5 * blurCount, r0, rn, stream));
        else if (perIteration || blurCount - 1 > 8)
            CRY_HIP(cry::launch_blur_replay(*cb, edge, planes[s.in], planes[s.out], W, H, s.row0, s.rows, stamp, stream));
        else {           // iterations 1 .. blurCount - 1 as one launch with per-tile dependencies
            CRY_HIP(cry::launch_blur_replay_chain(*cb, edge, planes[0], planes[1], W, H, blurCount, row0, rows, stamp, stream));
            if (ctx) {
                ctx->chainStatus = cry::edge_plane_carve(edge, W, H).progress + (size_t)cry::blur_tiles_x(W) * cry::blur_tiles_y(H);
                ctx->chainTag = ((unsigned long long)stamp << 8) | 255ull;
            }
            break;
        }
    }
    return 0;
}

int fill_light_params(cry::LightParams& P, const crychic_pass_constants* cb, const uint32_t* const shadow[4],
                      uint32_t shadowDim, uint32_t cubeDim, uint32_t W, uint32_t H, int numDirLights,
                      float pcfSearchRadius, uint32_t flags)
{
    if (numDirLights < 0 || numDirLights > CRYCHIC_MAX_LIGHTS)
        return fail(CRYCHIC_E_INVALID_ARG, "numDirLights %d outside [0,%d]", numDirLights, CRYCHIC_MAX_LIGHTS);
    if (shadowDim < 2 || cubeDim < 2 || shadowDim > 16384 || cubeDim > 8192)
        return fail(CRYCHIC_E_INVALID_ARG, "shadowDim %u / cubeDim %u outside [2, 16384] / [2, 8192]", shadowDim, cubeDim);
    if (!(pcfSearchRadius >= 0.0f)) return fail(CRYCHIC_E_INVALID_ARG, "pcfSearchRadius must be >= 0");
    memcpy(P.ViewProjTex, cb->ViewProjTex, sizeof P.ViewProjTex);
    memcpy(P.ShadowTransforms, cb->ShadowTransforms, sizeof P.ShadowTransforms);  // cascades 0..3
    memcpy(P.InvProj, cb->InvProj, sizeof P.InvProj);
    memcpy(P.InvView, cb->InvView, sizeof P.InvView);
    memcpy(P.EyePosW, cb->EyePosW, sizeof P.EyePosW);
    P.pcfSearchRadius = pcfSearchRadius;
    memcpy(P.AmbientLight, cb->AmbientLight, sizeof P.AmbientLight);
    memcpy(P.Lights, cb->Lights, sizeof P.Lights);
    for (int i = 0; i < 4; ++i) {
        if (!shadow[i]) return fail(CRYCHIC_E_INVALID_ARG, "shadow cascade %d is null", i);
        P.shadow[i] = shadow[i];
    }
    P.shadowDim = shadowDim;
    P.cubeDim = cubeDim;
    P.W = W;
    P.H = H;
    P.numDirLights = numDirLights;
    P.flags = flags;
    // CRYCHIC_LIGHT_CUBE_LEVELS: the cube map's mip chain (0 / 1 = level 0 alone); a chain ends at 1 x 1 at the latest
    P.cubeLevels = (flags >> 16) & 15u;
    { uint32_t full = 1; for (uint32_t m = cubeDim; m > 1u; m >>= 1) ++full;
      if (P.cubeLevels > full) return fail(CRYCHIC_E_INVALID_ARG, "%u cube map levels, a %u-texel face has at most %u", P.cubeLevels, cubeDim, full); }
    P.pointLights = nullptr;
    P.numPointLights = 0;
    P.shadowWIsOne = cry::light_shadow_w_is_one(P.ShadowTransforms) ? 1u : 0u;
    P.darkLights = cry::light_dark_mask(P.Lights, numDirLights);
    P.unitLights = cry::light_dark_lengths_ok(P.Lights, numDirLights) ? 1u : 0u;
    P.rcpW = cry::rcp((float)W);          // sky_pixel's pixel-centre uv: (x + 0.5) * rcp(W), the reciprocal taken once
    P.rcpH = cry::rcp((float)H);
    cry::light_params_derive(P);
    return 0;
}

// With a mip chain the level of detail comes from 2 x 2 pixel quads: a call's rows have to be whole quad rows.
int check_chain_rows(const cry::LightParams& P, uint32_t row0, uint32_t rows, uint32_t H)
{
    if (P.cubeLevels > 1u && ((row0 & 1u) || ((rows & 1u) && row0 + rows != H)))
        return fail(CRYCHIC_E_INVALID_ARG, "rows [%u,+%u): with a cube map mip chain a call covers whole pixel quads (even row0; even rows unless they end the frame)", row0, rows);
    return 0;
}

}  // namespace

extern "C" {

const char* crychic_last_error(void) { return g_err; }
const char* crychic_version(void) { return "crychic-hip 0.1 (gfx950)"; }

int crychic_ctx_create(int device_ordinal, crychic_ctx** out)
{
    if (!out) return fail(CRYCHIC_E_INVALID_ARG, "out is null");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return fail(CRYCHIC_E_NO_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
    if (device_ordinal < 0 || device_ordinal >= count)
        return fail(CRYCHIC_E_NO_DEVICE, "device ordinal %d outside [0,%d)", device_ordinal, count);
    CRY_HIP(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    CRY_HIP(hipGetDeviceProperties(&prop, device_ordinal));
    crychic_ctx* ctx = new (std::nothrow) crychic_ctx();
    if (!ctx) return fail(CRYCHIC_E_HIP, "out of host memory");
    ctx->device = device_ordinal;
    snprintf(ctx->name, sizeof ctx->name, "%s %s", prop.gcnArchName, prop.name);
    ctx->profiling = false;
    ctx->times_valid = false;
    ctx->rasterStatus = nullptr;
    ctx->chainStatus = nullptr;
    ctx->chainTag = 0ull;
    for (auto& ev : ctx->ev) {
        hipError_t ee = hipEventCreate(&ev);
        if (ee != hipSuccess) { delete ctx; return fail(CRYCHIC_E_HIP, "hipEventCreate failed: %s", hipGetErrorString(ee)); }
    }
    *out = ctx;
    return 0;
}

void crychic_ctx_destroy(crychic_ctx* ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    for (auto& ev : ctx->ev) (void)hipEventDestroy(ev);
    delete ctx;
}

const char* crychic_ctx_device_name(crychic_ctx* ctx) { return ctx ? ctx->name : nullptr; }

size_t crychic_edge_plane_bytes(uint32_t W, uint32_t H) { return cry::edge_plane_bytes(W, H); }

int crychic_ssao(crychic_ctx* ctx, const crychic_ssao_constants* cb, const void* normal_dev, const uint32_t* depth_dev,
                 const uint8_t* randvec_dev, uint16_t* ambient_out_dev, void* edge_dev, uint32_t W, uint32_t H,
                 uint32_t row0, uint32_t rows, void* stream)
{
    if (int rc = bind(ctx)) return rc;
    if (int rc = check_dims(W, H)) return rc;
    if (!cb || !normal_dev || !depth_dev || !randvec_dev || !ambient_out_dev) return fail(CRYCHIC_E_INVALID_ARG, "null argument");
    if (row0 > H / 2 || rows > H / 2 - row0) return fail(CRYCHIC_E_INVALID_ARG, "rows [%u,+%u) outside the %u-row ambient map", row0, rows, H / 2);
    const uint32_t stamp = next_stamp();
    const bool usePairs = edge_dev != nullptr;
    if (edge_dev) CRY_HIP(cry::launch_depth_pairs(*cb, depth_dev, edge_dev, W, H, stamp, usePairs, row0, rows, (hipStream_t)stream));
    CRY_HIP(cry::launch_ssao(*cb, normal_dev, depth_dev, randvec_dev, ambient_out_dev, edge_dev, W, H, row0, rows, true, usePairs,
                             edge_dev ? stamp : 0u, (hipStream_t)stream));
    return 0;
}

int crychic_ssao_edges(crychic_ctx* ctx, const crychic_ssao_constants* cb, const void* normal_dev,
                       const uint32_t* depth_dev, void* edge_dev, uint32_t W, uint32_t H, uint32_t row0, uint32_t rows,
                       void* stream)
{
    if (int rc = bind(ctx)) return rc;
    if (int rc = check_dims(W, H)) return rc;
    if (!cb || !normal_dev || !depth_dev || !edge_dev) return fail(CRYCHIC_E_INVALID_ARG, "null argument");
    if (row0 > H / 2 || rows > H / 2 - row0) return fail(CRYCHIC_E_INVALID_ARG, "rows [%u,+%u) outside the %u-row ambient map", row0, rows, H / 2);
    CRY_HIP(cry::launch_ssao(*cb, normal_dev, depth_dev, nullptr, nullptr, edge_dev, W, H, row0, rows, false, false, 0u, (hipStream_t)stream));
    return 0;
}

int crychic_ssao_blur(crychic_ctx* ctx, const crychic_ssao_constants* cb, const void* edge_dev,
                      const uint16_t* ambient_in_dev, uint16_t* ambient_out_dev, uint32_t W, uint32_t H, int horizontal,
                      uint32_t row0, uint32_t rows, void* stream)
{
    if (int rc = bind(ctx)) return rc;
    if (int rc = check_dims(W, H)) return rc;
    if (!cb || !edge_dev || !ambient_in_dev || !ambient_out_dev) return fail(CRYCHIC_E_INVALID_ARG, "null argument");
    if (ambient_in_dev == ambient_out_dev) return fail(CRYCHIC_E_INVALID_ARG, "blur cannot run in place (the reference ping-pongs, Ssao.cpp:253-266)");
    if (row0 > H / 2 || rows > H / 2 - row0) return fail(CRYCHIC_E_INVALID_ARG, "rows [%u,+%u) outside the %u-row ambient map", row0, rows, H / 2);
    CRY_HIP(cry::launch_blur(*cb, edge_dev, ambient_in_dev, ambient_out_dev, W, H, horizontal != 0, row0, rows, (hipStream_t)stream));
    return 0;
}

int crychic_ssao_compute(crychic_ctx* ctx, const crychic_ssao_constants* cb, const void* normal_dev,
                         const uint32_t* depth_dev, const uint8_t* randvec_dev, uint16_t* ambient0_dev,
                         uint16_t* ambient1_dev, void* edge_dev, uint32_t W, uint32_t H, int blurCount, uint32_t row0,
                         uint32_t rows, void* stream)
{
    if (int rc = bind(ctx)) return rc;
    if (int rc = check_dims(W, H)) return rc;
    if (!cb || !normal_dev || !depth_dev || !randvec_dev || !ambient0_dev) return fail(CRYCHIC_E_INVALID_ARG, "null argument");
    return ssao_compute_impl(cb, normal_dev, depth_dev, randvec_dev, ambient0_dev, ambient1_dev, edge_dev, W, H,
                             blurCount, row0, rows, (hipStream_t)stream, nullptr, ctx);
}

int crychic_deferred_light(crychic_ctx* ctx, const crychic_pass_constants* cb, const float* g0_dev, const float* g1_dev,
                           const float* g2_dev, const uint32_t* depth_dev, const uint16_t* ambient_dev,
                           const uint32_t* const shadow_dev[4], uint32_t shadowDim, const uint8_t* cube_dev,
                           uint32_t cubeDim, uint8_t* out_rgba8_dev, float* radiance_out_dev, uint32_t W, uint32_t H,
                           uint32_t row0, uint32_t rows, int numDirLights, float pcfSearchRadius, uint32_t flags,
                           void* stream)
{
    if (int rc = bind(ctx)) return rc;
    if (int rc = check_dims(W, H)) return rc;
    if (!cb || !g0_dev || !g1_dev || !g2_dev || !depth_dev || !shadow_dev || !cube_dev || !out_rgba8_dev)
        return fail(CRYCHIC_E_INVALID_ARG, "null argument");
    if (row0 > H || rows > H - row0) return fail(CRYCHIC_E_INVALID_ARG, "rows [%u,+%u) outside the %u-row frame", row0, rows, H);
    cry::LightParams P;
    if (int rc = fill_light_params(P, cb, shadow_dev, shadowDim, cubeDim, W, H, numDirLights, pcfSearchRadius, flags)) return rc;
    if (int rc = check_chain_rows(P, row0, rows, H)) return rc;
    CRY_HIP(cry::launch_light(P, g0_dev, g1_dev, g2_dev, depth_dev, ambient_dev, cube_dev, out_rgba8_dev, radiance_out_dev,
                              row0, rows, (hipStream_t)stream));
    return 0;
}

int crychic_deferred_light_points(crychic_ctx* ctx, const crychic_pass_constants* cb, const float* g0_dev, const float* g1_dev,
                                  const float* g2_dev, const uint32_t* depth_dev, const uint16_t* ambient_dev,
                                  const uint32_t* const shadow_dev[4], uint32_t shadowDim, const uint8_t* cube_dev,
                                  uint32_t cubeDim, uint8_t* out_rgba8_dev, float* radiance_out_dev, uint32_t W, uint32_t H,
                                  uint32_t row0, uint32_t rows, int numDirLights, float pcfSearchRadius, uint32_t flags,
                                  const crychic_light* point_lights_dev, uint32_t numPointLights, void* stream)
{
    if (int rc = bind(ctx)) return rc;
    if (int rc = check_dims(W, H)) return rc;
    if (!cb || !g0_dev || !g1_dev || !g2_dev || !depth_dev || !shadow_dev || !cube_dev || !out_rgba8_dev)
        return fail(CRYCHIC_E_INVALID_ARG, "null argument");
    if (row0 > H || rows > H - row0) return fail(CRYCHIC_E_INVALID_ARG, "rows [%u,+%u) outside the %u-row frame", row0, rows, H);
    if (numPointLights > cry::kMaxPointLights || (numPointLights && !point_lights_dev))
        return fail(CRYCHIC_E_INVALID_ARG, "numPointLights %u (max %u) / null light buffer", numPointLights, cry::kMaxPointLights);
    cry::LightParams P;
    if (int rc = fill_light_params(P, cb, shadow_dev, shadowDim, cubeDim, W, H, numDirLights, pcfSearchRadius, flags)) return rc;
    P.pointLights = point_lights_dev;
    P.numPointLights = numPointLights;
    if (int rc = check_chain_rows(P, row0, rows, H)) return rc;
    CRY_HIP(cry::launch_light(P, g0_dev, g1_dev, g2_dev, depth_dev, ambient_dev, cube_dev, out_rgba8_dev, radiance_out_dev,
                              row0, rows, (hipStream_t)stream));
    return 0;
}

}  // extern "C"

// crychic_draw_hot_path with the lighting pass issued as `nparts` consecutive row ranges of the strip (even-aligned, the
// crychic_strip_rows cut of the strip's rows); after(user, part, row0, rows) runs behind part's launch -- comm.cpp hangs that
// part's exchange there (crychic_draw_hot_path_shared).  nparts == 1, after == nullptr is crychic_draw_hot_path itself.
int cry::hot_path_parts(crychic_ctx* ctx, const crychic_ssao_constants* ssaoCB, const crychic_pass_constants* passCB,
                        const crychic_frame_desc* f, hipStream_t stream, uint32_t nparts, cry::PartHook after, void* user)
{
    if (int rc = bind(ctx)) return rc;
    if (!ssaoCB || !passCB || !f) return fail(CRYCHIC_E_INVALID_ARG, "null argument");
    if (int rc = check_dims(f->W, f->H)) return rc;
    const uint32_t W = f->W, H = f->H;
    if (f->row0 > H || f->rows > H - f->row0 || (f->row0 & 1u) || ((f->rows & 1u) && f->row0 + f->rows != H))
        return fail(CRYCHIC_E_INVALID_ARG, "strip [%u,+%u) must lie inside the frame and start on an even row", f->row0, f->rows);
    if (!f->depth_dev || !f->g0_dev || !f->g1_dev || !f->g2_dev || !f->cube_dev || !f->out_rgba8_dev)
        return fail(CRYCHIC_E_INVALID_ARG, "null plane in frame descriptor");
    const bool ssaoOn = f->blurCount >= 0;
    if (ssaoOn && (!f->normal_dev || !f->randvec_dev || !f->ambient0_dev)) return fail(CRYCHIC_E_INVALID_ARG, "SSAO planes missing");
    if (nparts == 0 || nparts > CRYCHIC_MAX_EXCHANGE_PARTS) return fail(CRYCHIC_E_INVALID_ARG, "nparts %u (1 .. %d)", nparts, CRYCHIC_MAX_EXCHANGE_PARTS);
    cry::LightParams P;
    if (int rc = fill_light_params(P, passCB, f->shadow_dev, f->shadowDim, f->cubeDim, W, H, f->numDirLights,
                                   f->pcfSearchRadius, f->flags)) return rc;
    if (f->numPointLights > cry::kMaxPointLights || (f->numPointLights && !f->point_lights_dev))
        return fail(CRYCHIC_E_INVALID_ARG, "numPointLights %u (max %u) / null light buffer", f->numPointLights, cry::kMaxPointLights);
    P.pointLights = f->point_lights_dev;
    P.numPointLights = f->numPointLights;
    const bool prof = ctx->profiling;
    if (prof) { ctx->times_valid = false; CRY_HIP(hipEventRecord(ctx->ev[0], stream)); }
    if (ssaoOn) {
        // The lighting pass filters the half-res AO map bilinearly at (about) its own pixel centre
        // (DeferredShading.hlsl:40-42): rows row0/2 - 1 .. (row0+rows)/2; keep one more row of slack.
        uint32_t a0, an;
        clamp_rows(H / 2, (int64_t)(f->row0 / 2) - 2, (int64_t)((f->row0 + f->rows + 1) / 2) + 2, &a0, &an);
        if (int rc = ssao_compute_impl(ssaoCB, f->normal_dev, f->depth_dev, f->randvec_dev, f->ambient0_dev,
                                       f->ambient1_dev, f->edge_dev, W, H, f->blurCount, a0, an, stream,
                                       prof ? ctx->ev[1] : nullptr, ctx)) return rc;
    } else if (prof) {
        CRY_HIP(hipEventRecord(ctx->ev[1], stream));
    }
    if (prof) CRY_HIP(hipEventRecord(ctx->ev[2], stream));
    // parts: whole half-res rows each, the last one takes the remainder (and an odd last row of the frame)
    const uint32_t pairs = f->rows / 2u, per = pairs / nparts;
    for (uint32_t p = 0; p < nparts; ++p) {
        const uint32_t r0 = f->row0 + 2u * per * p;
        const uint32_t r1 = (p + 1u == nparts) ? f->row0 + f->rows : r0 + 2u * per;
        if (r1 > r0)
            CRY_HIP(cry::launch_light(P, f->g0_dev, f->g1_dev, f->g2_dev, f->depth_dev, ssaoOn ? f->ambient0_dev : nullptr,
                                      f->cube_dev, f->out_rgba8_dev, nullptr, r0, r1 - r0, stream));
        if (prof && p + 1u == nparts) { CRY_HIP(hipEventRecord(ctx->ev[3], stream)); ctx->times_valid = true; }
        if (after)
            if (int rc = after(user, p, r0, r1 - r0)) return rc;
    }
    return 0;
}

extern "C" {

int crychic_draw_hot_path(crychic_ctx* ctx, const crychic_ssao_constants* ssaoCB, const crychic_pass_constants* passCB,
                          const crychic_frame_desc* f, void* stream)
{
    return cry::hot_path_parts(ctx, ssaoCB, passCB, f, (hipStream_t)stream, 1u, nullptr, nullptr);
}

int crychic_ctx_set_profiling(crychic_ctx* ctx, int enabled)
{
    if (!ctx) return fail(CRYCHIC_E_INVALID_ARG, "null context");
    ctx->profiling = enabled != 0;
    ctx->times_valid = false;
    return 0;
}

int crychic_ctx_last_pass_times(crychic_ctx* ctx, crychic_pass_times* out)
{
    if (int rc = bind(ctx)) return rc;
    if (!out) return fail(CRYCHIC_E_INVALID_ARG, "out is null");
    if (!ctx->times_valid) return fail(CRYCHIC_E_INVALID_ARG, "no profiled frame recorded (enable profiling, then draw)");
    CRY_HIP(hipEventSynchronize(ctx->ev[3]));
    CRY_HIP(hipEventElapsedTime(&out->ssao_ms, ctx->ev[0], ctx->ev[1]));
    CRY_HIP(hipEventElapsedTime(&out->blur_ms, ctx->ev[1], ctx->ev[2]));
    CRY_HIP(hipEventElapsedTime(&out->light_ms, ctx->ev[2], ctx->ev[3]));
    CRY_HIP(hipEventElapsedTime(&out->total_ms, ctx->ev[0], ctx->ev[3]));
    return 0;
}

size_t crychic_raster_workspace_bytes(uint64_t triangles, uint32_t W, uint32_t H) { return cry::raster_workspace_bytes(triangles, W, H); }

static int raster_common(crychic_ctx* ctx, cry::RasterPass& p, const crychic_pass_constants* passCB, const crychic_draw_item* items,
                         uint32_t nItems, void* stream)
{
    if (int rc = bind(ctx)) return rc;
    if (!passCB || (!items && nItems) || (!p.depth && p.nTargets < 2u) || !p.workspace) return fail(CRYCHIC_E_INVALID_ARG, "null argument");
    if (p.W == 0 || p.H == 0) return fail(CRYCHIC_E_INVALID_ARG, "bad target size %ux%u", p.W, p.H);
    // the same bounds as the full-screen passes: 32-bit plane offsets, and one workgroup row of the resolve pass per 4 pixel rows
    if ((uint64_t)p.W * p.H > (1ull << 28) || p.W >= (1u << 20) || p.H > 4u * 65535u)
        return fail(CRYCHIC_E_UNSUPPORTED, "target %ux%u exceeds 2^28 pixels, 2^20 columns or 262140 rows", p.W, p.H);
    for (uint32_t i = 0; i < nItems; ++i) {
        const crychic_draw_item& d = items[i];
        if (d.indexCount % 3u) return fail(CRYCHIC_E_INVALID_ARG, "item %u: indexCount %u is not a triangle list", i, d.indexCount);
        if (d.indexCount && d.instanceCount && (!d.vertices_dev || !d.indices_dev || !d.instances_dev))
            return fail(CRYCHIC_E_INVALID_ARG, "item %u: null buffer", i);
    }
    p.view = passCB->View;
    p.viewProj = passCB->ViewProj;
    p.items = items;
    p.nItems = nItems;
    hipError_t e = cry::launch_raster_pass(p, (hipStream_t)stream);
    if (e == hipSuccess) ctx->rasterStatus = p.statusWord;
    if (e == hipErrorInvalidValue) return fail(CRYCHIC_E_INVALID_ARG, "raster workspace too small (need crychic_raster_workspace_bytes) or too many textures");
    if (e != hipSuccess) return fail(CRYCHIC_E_HIP, "raster pass failed: %s", hipGetErrorString(e));
    return 0;
}

int crychic_raster_status(crychic_ctx* ctx, void* stream, uint32_t* flags)
{
    if (int rc = bind(ctx)) return rc;
    if (!flags) return fail(CRYCHIC_E_INVALID_ARG, "flags is null");
    *flags = 0;
    if (!ctx->rasterStatus) return fail(CRYCHIC_E_INVALID_ARG, "no producer pass has been issued on this context");
    CRY_HIP(hipMemcpyAsync(flags, ctx->rasterStatus, sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
    CRY_HIP(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}

int crychic_blur_chain_status(crychic_ctx* ctx, void* stream, uint32_t* timed_out)
{
    if (int rc = bind(ctx)) return rc;
    if (!timed_out) return fail(CRYCHIC_E_INVALID_ARG, "timed_out is null");
    *timed_out = 0;
    if (!ctx->chainStatus) return 0;             // no single-launch chain has run on this context
    unsigned long long word = 0;
    CRY_HIP(hipMemcpyAsync(&word, ctx->chainStatus, sizeof word, hipMemcpyDeviceToHost, (hipStream_t)stream));
    CRY_HIP(hipStreamSynchronize((hipStream_t)stream));
    *timed_out = word == ctx->chainTag ? 1u : 0u;
    return 0;
}

int crychic_draw_scene_to_shadow_map(crychic_ctx* ctx, const crychic_pass_constants* passCB, const crychic_draw_item* items,
                                     uint32_t nItems, uint32_t* shadow_dev, uint32_t shadowDim, int depthBias,
                                     float slopeScaledDepthBias, void* workspace_dev, size_t workspaceBytes, void* stream)
{
    cry::RasterPass p = {};
    p.mode = 0; p.W = shadowDim; p.H = shadowDim; p.depthBias = depthBias; p.slopeScaledDepthBias = slopeScaledDepthBias;
    p.depth = shadow_dev; p.workspace = workspace_dev; p.workspaceBytes = workspaceBytes;
    return raster_common(ctx, p, passCB, items, nItems, stream);
}

int crychic_draw_scene_to_shadow_maps(crychic_ctx* ctx, const crychic_pass_constants* passCBs, uint32_t nCascades, const crychic_draw_item* items,
                                      uint32_t nItems, uint32_t* const* shadow_dev, uint32_t shadowDim, int depthBias, float slopeScaledDepthBias,
                                      void* workspace_dev, size_t workspaceBytes, void* stream)
{
    if (!passCBs || !shadow_dev || nCascades < 1u || nCascades > 4u) return fail(CRYCHIC_E_INVALID_ARG, "1..4 cascades with their pass constants and targets");
    if (nCascades == 1u)
        return crychic_draw_scene_to_shadow_map(ctx, passCBs, items, nItems, shadow_dev[0], shadowDim, depthBias, slopeScaledDepthBias, workspace_dev,
                                                workspaceBytes, stream);
    cry::RasterPass p = {};
    p.mode = 0; p.W = shadowDim; p.H = shadowDim; p.depthBias = depthBias; p.slopeScaledDepthBias = slopeScaledDepthBias;
    p.workspace = workspace_dev; p.workspaceBytes = workspaceBytes;
    p.nTargets = nCascades;
    for (uint32_t c = 0; c < nCascades; ++c) {
        if (!shadow_dev[c]) return fail(CRYCHIC_E_INVALID_ARG, "shadow target %u is null", c);
        p.viewProjN[c] = passCBs[c].ViewProj;
        p.depthN[c] = shadow_dev[c];
    }
    return raster_common(ctx, p, passCBs, items, nItems, stream);
}

int crychic_draw_normals_and_depth(crychic_ctx* ctx, const crychic_pass_constants* passCB, const crychic_draw_item* items,
                                   uint32_t nItems, void* normal_dev, uint32_t* depth_dev, uint32_t W, uint32_t H,
                                   void* workspace_dev, size_t workspaceBytes, void* stream)
{
    if (!normal_dev) return fail(CRYCHIC_E_INVALID_ARG, "null normal target");
    cry::RasterPass p = {};
    p.mode = 1; p.W = W; p.H = H; p.depth = depth_dev; p.normal = normal_dev; p.workspace = workspace_dev; p.workspaceBytes = workspaceBytes;
    return raster_common(ctx, p, passCB, items, nItems, stream);
}

int crychic_draw_gbuffer(crychic_ctx* ctx, const crychic_pass_constants* passCB, const crychic_draw_item* items, uint32_t nItems,
                         const crychic_material_data* materials_dev, uint32_t nMaterials, const crychic_texture* textures,
                         uint32_t nTextures, float* g0_dev, float* g1_dev, float* g2_dev, uint32_t* depth_dev, uint32_t W, uint32_t H,
                         void* workspace_dev, size_t workspaceBytes, void* stream)
{
    if (!g0_dev || !g1_dev || !g2_dev) return fail(CRYCHIC_E_INVALID_ARG, "null G-buffer target");
    if (nTextures && !textures) return fail(CRYCHIC_E_INVALID_ARG, "null texture table");
    cry::RasterPass p = {};
    p.mode = 2; p.W = W; p.H = H; p.depth = depth_dev; p.g0 = g0_dev; p.g1 = g1_dev; p.g2 = g2_dev;
    p.materials = materials_dev; p.nMaterials = nMaterials;
    p.textures = reinterpret_cast<const cry::Texture*>(textures); p.nTextures = nTextures;
    p.workspace = workspace_dev; p.workspaceBytes = workspaceBytes;
    return raster_common(ctx, p, passCB, items, nItems, stream);
}

int crychic_draw_normals_depth_and_gbuffer(crychic_ctx* ctx, const crychic_pass_constants* passCB, const crychic_draw_item* items, uint32_t nItems,
                                           const crychic_material_data* materials_dev, uint32_t nMaterials, const crychic_texture* textures,
                                           uint32_t nTextures, void* normal_dev, float* g0_dev, float* g1_dev, float* g2_dev, uint32_t* depth_dev,
                                           uint32_t W, uint32_t H, void* workspace_dev, size_t workspaceBytes, void* stream)
{
    if (!normal_dev || !g0_dev || !g1_dev || !g2_dev) return fail(CRYCHIC_E_INVALID_ARG, "null normal / G-buffer target");
    if (nTextures && !textures) return fail(CRYCHIC_E_INVALID_ARG, "null texture table");
    cry::RasterPass p = {};
    p.mode = 3; p.W = W; p.H = H; p.depth = depth_dev; p.normal = normal_dev; p.g0 = g0_dev; p.g1 = g1_dev; p.g2 = g2_dev;
    p.materials = materials_dev; p.nMaterials = nMaterials;
    p.textures = reinterpret_cast<const cry::Texture*>(textures); p.nTextures = nTextures;
    p.workspace = workspace_dev; p.workspaceBytes = workspaceBytes;
    return raster_common(ctx, p, passCB, items, nItems, stream);
}

int crychic_draw_gbuffer_rows(crychic_ctx* ctx, const crychic_pass_constants* passCB, const crychic_draw_item* items, uint32_t nItems,
                              const crychic_material_data* materials_dev, uint32_t nMaterials, const crychic_texture* textures,
                              uint32_t nTextures, float* g0_dev, float* g1_dev, float* g2_dev, uint32_t* depth_dev, uint32_t W, uint32_t H,
                              uint32_t gRow0, uint32_t gRows, void* workspace_dev, size_t workspaceBytes, void* stream)
{
    if (!g0_dev || !g1_dev || !g2_dev) return fail(CRYCHIC_E_INVALID_ARG, "null G-buffer target");
    if (nTextures && !textures) return fail(CRYCHIC_E_INVALID_ARG, "null texture table");
    if (gRows == 0 || gRow0 > H || gRows > H - gRow0) return fail(CRYCHIC_E_INVALID_ARG, "G-buffer rows [%u,+%u) outside the %u-row target", gRow0, gRows, H);
    cry::RasterPass p = {};
    p.mode = 2; p.W = W; p.H = H; p.depth = depth_dev; p.g0 = g0_dev; p.g1 = g1_dev; p.g2 = g2_dev;
    p.materials = materials_dev; p.nMaterials = nMaterials;
    p.textures = reinterpret_cast<const cry::Texture*>(textures); p.nTextures = nTextures;
    p.workspace = workspace_dev; p.workspaceBytes = workspaceBytes;
    p.gRow0 = gRow0; p.gRows = gRows;
    return raster_common(ctx, p, passCB, items, nItems, stream);
}

int crychic_draw_normals_depth_and_gbuffer_rows(crychic_ctx* ctx, const crychic_pass_constants* passCB, const crychic_draw_item* items, uint32_t nItems,
                                                const crychic_material_data* materials_dev, uint32_t nMaterials, const crychic_texture* textures,
                                                uint32_t nTextures, void* normal_dev, float* g0_dev, float* g1_dev, float* g2_dev, uint32_t* depth_dev,
                                                uint32_t W, uint32_t H, uint32_t gRow0, uint32_t gRows, void* workspace_dev, size_t workspaceBytes,
                                                void* stream)
{
    if (!normal_dev || !g0_dev || !g1_dev || !g2_dev) return fail(CRYCHIC_E_INVALID_ARG, "null normal / G-buffer target");
    if (nTextures && !textures) return fail(CRYCHIC_E_INVALID_ARG, "null texture table");
    if (gRows == 0 || gRow0 > H || gRows > H - gRow0) return fail(CRYCHIC_E_INVALID_ARG, "G-buffer rows [%u,+%u) outside the %u-row target", gRow0, gRows, H);
    cry::RasterPass p = {};
    p.mode = 3; p.W = W; p.H = H; p.depth = depth_dev; p.normal = normal_dev; p.g0 = g0_dev; p.g1 = g1_dev; p.g2 = g2_dev;
    p.materials = materials_dev; p.nMaterials = nMaterials;
    p.textures = reinterpret_cast<const cry::Texture*>(textures); p.nTextures = nTextures;
    p.workspace = workspace_dev; p.workspaceBytes = workspaceBytes;
    p.gRow0 = gRow0; p.gRows = gRows;
    return raster_common(ctx, p, passCB, items, nItems, stream);
}

int crychic_strip_rows(uint32_t H, int nranks, int rank, uint32_t* row0, uint32_t* rows)
{
    if (!row0 || !rows || nranks <= 0 || rank < 0 || rank >= nranks || (H & 1u))
        return fail(CRYCHIC_E_INVALID_ARG, "bad strip request (H=%u nranks=%d rank=%d)", H, nranks, rank);
    const uint32_t pairs = H / 2;                    // strips are whole half-res rows
    const uint32_t per = pairs / (uint32_t)nranks;   // last rank takes the remainder
    const uint32_t p0 = per * (uint32_t)rank;
    const uint32_t p1 = (rank == nranks - 1) ? pairs : p0 + per;
    *row0 = 2 * p0;
    *rows = 2 * (p1 - p0);
    return 0;
}

}  // extern "C"
