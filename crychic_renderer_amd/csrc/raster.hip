// raster.hip -- gfx950 kernels of the producer passes: a D3D-rules rasteriser that fills the shadow
// cascades, the view-normal + depth target and the G-buffer from indexed, instanced triangle lists.
//
// Structure (all launches stream-ordered, no host synchronisation):
//   clear_vis      visibility buffer := (depth 1.0, no primitive)
//   setup_kernel   one lane per (instance, triangle): vertex shader x3, clip to 0 <= z <= w, viewport + 1/256 snap,
//                  cull; up to 7 setup triangles (z and guard-band clipping) land in slots fixed by draw order (slot = serial, so equal depths
//                  resolve to the earlier primitive exactly like in-order LESS testing); live slots are appended to
//                  a list with one atomic each
//   raster_kernel  persistent workgroups walk the live list, one wavefront per small triangle (a workgroup per large one); a 16 x 4 lane footprint sweeps
//                  the triangle's pixel box with 64-bit integer edge functions and atomicMin()s the 64-bit key
//                  (d24 << 32 | serial); the shadow pass atomicMin()s the D24 value straight into the depth plane
//   resolve_kernel one lane per pixel: depth plane from the key, perspective-correct attributes of the winning
//                  primitive, then the pass's pixel shader (DrawNormals.hlsl / GeometryPass.hlsl)
#include <hip/hip_runtime.h>
#include "kernels.hpp"
#include "raster_core.hpp"

namespace cry {

constexpr int kLargeBox = 2048;   // pixel-box area above which a triangle gets a whole workgroup
struct RasterCounters { uint32_t nlive; uint32_t overflow; uint32_t nlarge; uint32_t pad; };

// Clears: 16-byte stores (two visibility keys / four depth texels per lane); the planes come from hipMalloc / torch and are
// 16-byte aligned, odd tails are finished with scalar stores by the last lanes.
__global__ __launch_bounds__(256) void clear_vis_kernel(uint64_t* __restrict__ vis, uint32_t n, RasterCounters* c)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t pairs = n / 2u;
    if (i < pairs) reinterpret_cast<ulonglong2*>(vis)[i] = ulonglong2{ kVisClear, kVisClear };
    if (i == pairs && (n & 1u)) vis[n - 1u] = kVisClear;
    if (i == 0) { c->nlive = 0; c->overflow = 0; c->nlarge = 0; }
}

struct ViewProjSet { crychic_pass_constants_viewproj m[4]; };     // one per target of a fused shadow pass
struct DepthTargets { uint32_t* p[4]; };

// Up to kBatchItems render items of a pass share one setup launch (a pass of the reference scene has two: boxes + grid).
constexpr uint32_t kBatchItems = 8;
struct ItemBatch {
    crychic_draw_item item[kBatchItems];
    uint64_t first[kBatchItems + 1];      // prefix sums of instanceCount * triangles; first[n] = total of the batch
    uint32_t n;
};

__global__ __launch_bounds__(128) void setup_kernel(ItemBatch batch, const crychic_material_data* __restrict__ materials,
                                                    uint32_t nMaterials, ViewProjSet vps, uint32_t W, uint32_t H,
                                                    SetupTri* __restrict__ tris, uint32_t slotBase, uint32_t slotsPerTarget,
                                                    uint32_t* __restrict__ live, uint32_t liveCapacity, RasterCounters* __restrict__ counters,
                                                    uint32_t yLo, uint32_t yHi)
{
    const uint32_t target = blockIdx.y;                                // shadow cascades of one fused pass; 0 otherwise
    const crychic_pass_constants_viewproj& vp = vps.m[target];
    // small triangles are listed from the front of `live`, large ones (pixel box > kLargeBox) from its back
    auto append = [&](const SetupTri& s, uint32_t slot) {
        const PixelBox b = triangle_box(s, W, yLo, yHi);
        if (b.x0 > b.x1 || b.y0 > b.y1) return;                       // covers no pixel centre inside the target (or its scissor rows)
        if ((b.x1 - b.x0 + 1) * (b.y1 - b.y0 + 1) > kLargeBox) live[liveCapacity - 1u - atomicAdd(&counters->nlarge, 1u)] = slot;
        else live[atomicAdd(&counters->nlive, 1u)] = slot;
    };
    const uint64_t bid = (uint64_t)blockIdx.x * 128u + threadIdx.x;     // triangle of the batch; draw order = item, instance, triangle
    if (bid >= batch.first[batch.n]) return;
    uint32_t k = 0;
    while (k + 1u < batch.n && bid >= batch.first[k + 1u]) ++k;
    const crychic_draw_item& item = batch.item[k];
    const uint32_t ntri = item.indexCount / 3u;
    const uint64_t gid = bid - batch.first[k];
    const uint32_t inst = (uint32_t)(gid / ntri), tri = (uint32_t)(gid - (uint64_t)inst * ntri);
    const crychic_instance_data I = item.instances_dev[inst];
    const crychic_material_data* M = (materials && I.MaterialIndex < nMaterials) ? &materials[I.MaterialIndex] : nullptr;
    const uint32_t slot0 = target * slotsPerTarget + slotBase + (uint32_t)bid * (uint32_t)kSlotsPerTriangle;
    // slots that receive no triangle are never listed in `live`, and only listed slots are read: nothing to clear here

    VsOut v[3];
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int64_t vi = (int64_t)item.indices_dev[item.startIndexLocation + tri * 3u + c] + item.baseVertexLocation;
        if (vi < 0 || vi >= (int64_t)item.vertexCount) { bad = true; break; }
        v[c] = vertex_shader(item.vertices_dev[vi], I, M, vp.m);
    }
    if (bad) { atomicOr(&counters->overflow, 2u); return; }

    bool overflow = false;
    const float gx = guard_band(W), gy = guard_band(H);
    bool inside = true;                    // nearly every triangle: inside all six planes, no clipping
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int plane = 0; plane < kClipPlanes; ++plane) inside = inside & (clip_distance(v[c], plane, gx, gy) >= 0.0f);
    if (inside) {
        SetupTri s;
        if (setup_triangle(v[0], v[1], v[2], I.MaterialIndex, W, H, s, &overflow)) {
            s.pad = target;
            tris[slot0] = s;
            append(s, slot0);
        }
    } else {
        VsOut poly[kMaxPolyVerts], tmp[kMaxPolyVerts];
        poly[0] = v[0]; poly[1] = v[1]; poly[2] = v[2];
        const int n = clip_triangle(poly, tmp, W, H);
        for (int c = 1; c + 1 < n; ++c) {
            SetupTri s;
            if (setup_triangle(poly[0], poly[c], poly[c + 1], I.MaterialIndex, W, H, s, &overflow)) {
                s.pad = target;
                tris[slot0 + (uint32_t)(c - 1)] = s;
                append(s, slot0 + (uint32_t)(c - 1));
            }
        }
    }
    if (overflow) atomicOr(&counters->overflow, 1u);
}

__global__ __launch_bounds__(256) void clear_depth_kernel(uint32_t* __restrict__ depth, uint32_t n, RasterCounters* c)
{
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    const uint32_t head = (uint32_t)((16u - ((uintptr_t)depth & 15u)) & 15u) / 4u;   // texels before the first 16-byte boundary
    if (head) {                                                                        // caller's plane is only 4-byte aligned
        if (i < head && i < n) depth[i] = 0x00FFFFFFu;
        depth += head < n ? head : n;
        n -= head < n ? head : n;
    }
    const uint32_t quads = n / 4u;
    if (i < quads) reinterpret_cast<uint4*>(depth)[i] = uint4{ 0x00FFFFFFu, 0x00FFFFFFu, 0x00FFFFFFu, 0x00FFFFFFu };
    if (i == quads) for (uint32_t k = quads * 4u; k < n; ++k) depth[k] = 0x00FFFFFFu;
    if (i == 0) { c->nlive = 0; c->overflow = 0; c->nlarge = 0; }
}

// Small triangles (pixel box <= kLargeBox): one wavefront per triangle with a 2^FWL2-wide lane footprint (16 x 4), four
// triangles per workgroup step.  Large triangles: one workgroup per triangle, its four waves sweeping interleaved
// footprint-high stripes.  Which lane tests which pixel does not matter: the result is an atomicMin over
// order-independent keys.  SHADOW: depth-only pass -- the minimum of the D24 values is all the pass needs (equal depths
// carry the same value whichever primitive wins), so the atomics go straight to the 32-bit depth plane and neither the
// 64-bit visibility buffer nor the resolve pass exist.
template <bool SHADOW, int FWL2>
__global__ __launch_bounds__(256) void raster_kernel(const SetupTri* __restrict__ tris, const uint32_t* __restrict__ live, uint32_t liveCapacity,
                                                     const RasterCounters* __restrict__ counters, unsigned long long* __restrict__ vis,
                                                     DepthTargets depthTargets, uint32_t W, uint32_t H, int depthBias, float slopeBias,
                                                     uint32_t yLo, uint32_t yHi)
{
    const uint32_t nsmall = counters->nlive, nlarge = counters->nlarge;   // written by the setup kernels that precede this launch
    const int lane = (int)(threadIdx.x & 63u), wave = (int)(threadIdx.x >> 6);
    constexpr int FW = 1 << FWL2, FH = 64 >> FWL2;     // footprint of one wavefront
    const int lx = lane & (FW - 1), ly = lane >> FWL2;
    auto sweep = [&](uint32_t slot, int yfirst, int ystep) {
        const SetupTri t = tris[slot];
        uint32_t* __restrict__ depth = depthTargets.p[SHADOW ? (t.pad & 3u) : 0u];
        const PixelBox b = triangle_box(t, W, yLo, yHi);
        const EdgeFlags e = triangle_edge_flags(t);
        const double bias = SHADOW ? triangle_depth_bias(t, depthBias, slopeBias) : 0.0;
        for (int y = b.y0 + ly + yfirst; y <= b.y1; y += ystep)
            for (int x = b.x0 + lx; x <= b.x1; x += FW) {
                const uint64_t key = fragment_key(t, e, bias, x, y, slot + 1u);
                if (key == ~0ull) continue;
                if (SHADOW) atomicMin(&depth[(uint32_t)y * W + (uint32_t)x], (uint32_t)(key >> 32));
                else atomicMin(&vis[(uint32_t)y * W + (uint32_t)x], (unsigned long long)key);
            }
    };
    // every workgroup / wave reaches the same exit conditions (the counters are fixed for the whole launch)
    for (uint32_t k = blockIdx.x; k < nlarge; k += gridDim.x) sweep(live[liveCapacity - 1u - k], FH * wave, 4 * FH);
    for (uint32_t k = blockIdx.x * 4u + (uint32_t)wave; k < nsmall; k += gridDim.x * 4u)
        sweep((uint32_t)__builtin_amdgcn_readfirstlane((int)live[k]), 0, FH);       // wave-uniform: the triangle record comes in by scalar loads
}

__global__ __launch_bounds__(256) void resolve_kernel(int mode, const unsigned long long* __restrict__ vis, const SetupTri* __restrict__ tris,
                                                      crychic_pass_constants_viewproj view, const crychic_material_data* __restrict__ materials,
                                                      uint32_t nMaterials, const Texture* __restrict__ textures, uint32_t nTextures,
                                                      uint32_t W, uint32_t H, uint32_t* __restrict__ depth, u2* __restrict__ normal,
                                                      f4a* __restrict__ g0, f4a* __restrict__ g1, f4a* __restrict__ g2, uint32_t gLo, uint32_t gHi)
{
    const uint32_t x = blockIdx.x * 64u + (threadIdx.x & 63u), y = blockIdx.y * 4u + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    // G-buffer rows [gLo, gHi) only (a rank's strip): outside them the G-buffer stage of this pixel does not run and its
    // texels are left untouched; a G-buffer-only pass leaves the whole pixel (its depth target too) untouched there.
    if (y < gLo || y >= gHi) { mode &= ~2; if (mode == 0) return; }
    const uint32_t idx = y * W + x;
    const uint64_t key = vis[idx];
    const uint32_t serial = (uint32_t)(key & 0xFFFFFFFFull);
    depth[idx] = (uint32_t)(key >> 32);
    if (serial == 0) {                       // nothing drawn here: the pass's clear values
        if (mode & 1) normal[idx] = u2{ 0u, 0x00003C00u };                      // (0, 0, 1, 0) in fp16, Ssao.cpp:317
        if (mode & 2) { g0[idx] = f4a{ 0, 0, 0, 0 }; g1[idx] = f4a{ 0, 0, 0, 0 }; g2[idx] = f4a{ 0, 0, 0, 0 }; }   // CRYCHIC.cpp:2554
        return;
    }
    const ResolveOut r = resolve_pixel(mode, tris[serial - 1u], (int)x, (int)y, view.m, materials, nMaterials, textures, nTextures);
    if (mode & 1)
        normal[idx] = u2{ (uint32_t)float_to_half(r.normalV.x) | ((uint32_t)float_to_half(r.normalV.y) << 16), (uint32_t)float_to_half(r.normalV.z) };
    if (mode & 2) {
        g0[idx] = f4a{ r.g0.x, r.g0.y, r.g0.z, r.g0.w };
        g1[idx] = f4a{ r.g1.x, r.g1.y, r.g1.z, r.g1.w };
        g2[idx] = f4a{ r.g2.x, r.g2.y, r.g2.z, r.g2.w };
    }
}

size_t raster_workspace_bytes(uint64_t triangles, uint32_t W, uint32_t H)
{
    const uint64_t slots = triangles * (uint64_t)kSlotsPerTriangle;
    return (size_t)((uint64_t)W * H * 8u + slots * sizeof(SetupTri) + slots * 4u + 64u * sizeof(Texture) + sizeof(RasterCounters) + 1024u);
}

hipError_t launch_raster_pass(RasterPass& p, hipStream_t stream)
{
    // carve the workspace: vis | tris | live | textures | counters (all 16-byte aligned)
    uint64_t total = 0;
    for (uint32_t i = 0; i < p.nItems; ++i) total += (uint64_t)(p.items[i].indexCount / 3u) * p.items[i].instanceCount;
    const bool shadow = p.mode == 0;
    const uint32_t nT = (shadow && p.nTargets > 1u) ? p.nTargets : 1u;     // cascades rasterised by one fused shadow pass
    if (nT > 4u) return hipErrorInvalidValue;
    const uint64_t slotsPerTarget = total * (uint64_t)kSlotsPerTriangle, slots = slotsPerTarget * nT;
    if (slots >= 0xFFFFFFF0ull) return hipErrorInvalidValue;
    if (raster_workspace_bytes(total * nT, p.W, p.H) > p.workspaceBytes || p.nTextures > 64u) return hipErrorInvalidValue;
    char* base = (char*)p.workspace;
    unsigned long long* vis = (unsigned long long*)base;
    size_t off = (size_t)p.W * p.H * 8u;
    SetupTri* tris = (SetupTri*)(base + off); off += (size_t)slots * sizeof(SetupTri);
    uint32_t* live = (uint32_t*)(base + off); off += ((size_t)slots * 4u + 15u) & ~(size_t)15u;
    Texture* texDev = (Texture*)(base + off); off += 64u * sizeof(Texture);
    RasterCounters* counters = (RasterCounters*)(base + off);
    p.statusWord = &counters->overflow;

    const uint32_t npx = p.W * p.H;
    ViewProjSet vps;
    DepthTargets targets;
    for (uint32_t t = 0; t < 4u; ++t) {
        const float* m = (nT > 1u) ? p.viewProjN[t < nT ? t : 0u] : p.viewProj;
        for (int i = 0; i < 16; ++i) vps.m[t].m[i] = m[i];
        targets.p[t] = (nT > 1u) ? p.depthN[t < nT ? t : 0u] : p.depth;
    }
    if (shadow) for (uint32_t t = 0; t < nT; ++t)
        hipLaunchKernelGGL(clear_depth_kernel, dim3((npx / 4u + 256u) / 256u), dim3(256), 0, stream, targets.p[t], npx, counters);
    else hipLaunchKernelGGL(clear_vis_kernel, dim3((npx / 2u + 256u) / 256u), dim3(256), 0, stream, (uint64_t*)vis, npx, counters);
    if (p.nTextures) {
        hipError_t e = hipMemcpyAsync(texDev, p.textures, p.nTextures * sizeof(Texture), hipMemcpyHostToDevice, stream);
        if (e != hipSuccess) return e;
    }
    crychic_pass_constants_viewproj view;
    for (int i = 0; i < 16; ++i) view.m[i] = p.view[i];
    // Scissor rows (a rank's strip): the G-buffer stage of the resolve runs on rows [gLo, gHi) only; a pass that renders the
    // G-buffer alone (mode 2) needs no visibility outside them either, so its rasterisation is limited to the same rows.
    const uint32_t gLo = p.gRows ? p.gRow0 : 0u, gHi = p.gRows ? p.gRow0 + p.gRows : p.H;
    if (gLo > p.H || gHi > p.H) return hipErrorInvalidValue;
    const uint32_t yLo = p.mode == 2 ? gLo : 0u, yHi = p.mode == 2 ? gHi : p.H;
    uint32_t slotBase = 0;
    for (uint32_t i0 = 0; i0 < p.nItems;) {
        ItemBatch b;
        b.n = 0;
        b.first[0] = 0;
        for (; i0 < p.nItems && b.n < kBatchItems; ++i0) {
            const uint64_t n = (uint64_t)(p.items[i0].indexCount / 3u) * p.items[i0].instanceCount;
            if (n == 0) continue;
            b.item[b.n] = p.items[i0];
            b.first[b.n + 1u] = b.first[b.n] + n;
            ++b.n;
        }
        if (b.n == 0) continue;
        const uint64_t n = b.first[b.n];
        hipLaunchKernelGGL(setup_kernel, dim3((uint32_t)((n + 127u) / 128u), nT), dim3(128), 0, stream, b, p.materials, p.nMaterials,
                           vps, p.W, p.H, tris, slotBase, (uint32_t)slotsPerTarget, live, (uint32_t)slots, counters, yLo, yHi);
        slotBase += (uint32_t)(n * (uint64_t)kSlotsPerTriangle);
    }
    if (slots) {
#define CRY_RASTER(S, F) hipLaunchKernelGGL((raster_kernel<S, F>), dim3(256u * 8u), dim3(256), 0, stream, tris, live, (uint32_t)slots, counters, vis, targets, p.W, p.H, p.depthBias, p.slopeScaledDepthBias, yLo, yHi)
        if (shadow) CRY_RASTER(true, 4); else CRY_RASTER(false, 4);     // 16 x 4 footprint: measured best of 8x8 / 16x4 / 64x1
#undef CRY_RASTER
    }
    if (shadow) return hipGetLastError();
    hipLaunchKernelGGL(resolve_kernel, dim3((p.W + 63u) / 64u, (p.H + 3u) / 4u), dim3(256), 0, stream, p.mode, vis, tris, view,
                       p.materials, p.nMaterials, p.nTextures ? texDev : nullptr, p.nTextures, p.W, p.H, p.depth, (u2*)p.normal,
                       (f4a*)p.g0, (f4a*)p.g1, (f4a*)p.g2, gLo, gHi);
    return hipGetLastError();
}

}  // namespace cry
