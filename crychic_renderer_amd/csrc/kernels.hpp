// kernels.hpp -- launcher interface between the C ABI (api.cpp) and the gfx950 kernels (kernels.hip).
#pragma once
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include "crychic_hip.h"

namespace cry {

struct LightParams;

// D24 depth plane -> the decoded, BORDER-padded pairs plane inside the edge workspace (ssao_core.hpp "depth pairs") and the coarse
// maps of the SSAO shortcuts, for the SSAO pass over half-res rows [row0, row0 + rows): those rows' texels and a margin (the whole
// plane when the rows are the whole map; launch_ssao called with the same rows knows what was prepared).
// `stamp` (non-zero, different from the previous frame's) marks the cells of the coarse geometry map written by this call.
hipError_t launch_depth_pairs(const crychic_ssao_constants& cb, const uint32_t* depth, void* edge_base, uint32_t W, uint32_t H, uint32_t stamp,
                              bool writePairs, uint32_t row0, uint32_t rows, hipStream_t stream);
// use_pairs: the taps read the pairs plane (launch_depth_pairs earlier on the same stream) instead of the raw D24 plane.
hipError_t launch_ssao(const crychic_ssao_constants& cb, const void* normal, const uint32_t* depth,
                       const uint8_t* randvec, uint16_t* ambient, void* edge_base, uint32_t W, uint32_t H,
                       uint32_t row0, uint32_t rows, bool emit_ao, bool use_pairs, uint32_t stamp, hipStream_t stream);

// One self-contained sweep of SsaoBlur.hlsl (Ssao::BlurAmbientMap(cmdList, bool), Ssao.cpp:245-293) over half-res rows [row0, row0 + rows).
hipError_t launch_blur(const crychic_ssao_constants& cb, const void* edge_base, const uint16_t* in, uint16_t* out,
                       uint32_t W, uint32_t H, bool horizontal, uint32_t row0, uint32_t rows, hipStream_t stream);
// Iteration 0 of the blur chain: horizontal + vertical sweep in one launch (in -> out, out != in); the rows are those of the
// vertical sweep's output.  record: store both sweeps' tap decisions and the per-tile flags for launch_blur_replay.
// stamp (0 = no exit) / onesMargin / ssaoRow0, ssaoRows: the unoccluded-tile exit (blur_tiles.hpp) -- the SSAO pass of this frame
// wrote the unoccluded-wavefront map with `stamp` for half-res rows [ssaoRow0, ssaoRow0 + ssaoRows).
hipError_t launch_blur_pair(const crychic_ssao_constants& cb, const void* edge_base, const uint16_t* in, uint16_t* out, uint32_t W, uint32_t H,
                            uint32_t row0, uint32_t rows, bool record, uint32_t stamp, int onesMargin, uint32_t ssaoRow0, uint32_t ssaoRows,
                            hipStream_t stream);
// A later iteration of the chain (both sweeps), replaying what launch_blur_pair(record) stored (in -> out, out != in); the rows
// are those owed after it.  stamp: the one the pair launch ran with (its settled tiles return at once).
hipError_t launch_blur_replay(const crychic_ssao_constants& cb, const void* edge_base, const uint16_t* in, uint16_t* out, uint32_t W, uint32_t H,
                              uint32_t row0, uint32_t rows, uint32_t stamp, hipStream_t stream);

// Iterations 1 .. blurCount - 1 of the chain in one launch (kernels.hip blur_replay_chain_kernel: per-tile dependencies instead of
// kernel boundaries); plane0 / plane1 = ambient0 / ambient1, rows = what the caller is owed of the final map, stamp = the frame's
// (the one launch_blur_pair ran with).  Same pixels as blurCount - 1 calls of launch_blur_replay along blur_chain_step.
hipError_t launch_blur_replay_chain(const crychic_ssao_constants& cb, const void* edge_base, uint16_t* plane0, uint16_t* plane1, uint32_t W, uint32_t H,
                                    int blurCount, uint32_t row0, uint32_t rows, uint32_t stamp, hipStream_t stream);

hipError_t launch_light(const LightParams& P, const float* g0, const float* g1, const float* g2,
                        const uint32_t* depth, const uint16_t* ambient, const uint8_t* cube, uint8_t* out,
                        float* radiance, uint32_t row0, uint32_t rows, hipStream_t stream);


// ---- producer passes (raster.hip) ----
struct crychic_pass_constants_viewproj { float m[16]; };   // one transposed 4x4 passed by value in the kernarg segment
struct Texture;
struct RasterPass {
    int mode;                               // 0 shadow depth, 1 view normals + depth, 2 G-buffer + depth
    const float* view;                      // PassConstants.View (transposed storage)
    const float* viewProj;                  // PassConstants.ViewProj
    const crychic_draw_item* items; uint32_t nItems;     // host array, device pointers inside
    const crychic_material_data* materials; uint32_t nMaterials;   // device
    const Texture* textures; uint32_t nTextures;         // host array of {device pointer, w, h}
    uint32_t W, H;
    int depthBias; float slopeScaledDepthBias;
    uint32_t* depth; void* normal; float* g0; float* g1; float* g2;
    void* workspace; size_t workspaceBytes;
    uint32_t gRow0, gRows;                  // G-buffer rows to render (a rank's strip); gRows == 0 = the whole target
    const uint32_t* statusWord;             // OUT: device address of the pass's status word (bit 0: a vertex left the +-2^22 px
                                            // range and its triangle was dropped; bit 1: a vertex index outside the vertex buffer)
    // fused shadow pass (mode 0, nTargets 2..4): one ViewProj and one depth target per cascade, same items
    uint32_t nTargets; const float* viewProjN[4]; uint32_t* depthN[4];
};
size_t raster_workspace_bytes(uint64_t triangles, uint32_t W, uint32_t H);
hipError_t launch_raster_pass(RasterPass& p, hipStream_t stream);

}  // namespace cry
