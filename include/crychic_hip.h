/*
 * crychic_hip.h -- C ABI of libcrychic_hip.so, the MI355X (gfx950) backend for CRYCHIC's per-pixel
 * hot path: G-buffer -> 14-tap SSAO -> bilateral blur -> deferred PBR lighting with cascaded-shadow PCF.
 *
 * The reference (UnlimitedRainWorks/CRYCHIC-RENDERER) has no FFI: the seam is the public surface of its
 * pass objects (Ssao.h:10-125, DeferredShading.h:4-45, ShadowMap.h:4-47), FrameResource.h and
 * CRYCHIC::Draw (CRYCHIC.cpp:172-306).  Every entry point below cites the reference code it replaces.
 * The C++ veneer in include/crychic/ keeps the reference's class and method names on top of this ABI.
 *
 * Conventions
 *   - plain pointers and sizes only; every `*_dev` pointer is DEVICE memory (hipMalloc or equivalent),
 *     every constant-buffer pointer is HOST memory and is copied at call time (the caller may reuse
 *     it immediately, like an UploadBuffer slot guarded by the frame fence, CRYCHIC.cpp:135-146);
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); all work is
 *     stream-ordered and asynchronous, there is no host synchronisation inside any entry point;
 *   - every function returns 0 on success or a negative crychic_status; crychic_last_error()
 *     returns a thread-local description (the reference throws DxException, Common/d3dUtil.h:132-144);
 *   - there is NO CPU fallback: without a usable HIP device every compute entry point fails.
 *
 * Plane layouts (row-major, pitch = width * texel size, no tiling)
 *   depth    uint32  D24 in bits 0..23 (stencil bits ignored)           W x H
 *   normal   4 x fp16 view-space normal (DrawNormals.hlsl:93)            W x H
 *   ambient  uint16 R16_UNORM (Ssao.h:21)                                (W/2) x (H/2)
 *   randvec  4 x uint8 R8G8B8A8_UNORM (Ssao.cpp:362)                     256 x 256
 *   g0,g1,g2 4 x fp32 (GBuffer.hlsl:22-31)                               W x H
 *   shadow   uint32 D24 per cascade (ShadowMap.cpp:83,94)                shadowDim x shadowDim
 *   cube     4 x uint8 RGBA, faces +X,-X,+Y,-Y,+Z,-Z                     6 x cubeDim x cubeDim
 *   out      4 x uint8 R8G8B8A8_UNORM back buffer (Common/d3dApp.h:124)  W x H
 *   edge     opaque half-res workspace, crychic_edge_plane_bytes(W,H) bytes (see crychic_ssao)
 * W and H must be even (the half-res maps are W/2 x H/2, Ssao.cpp:22-30); the smallest frame is 2 x 2, the largest 2^28 pixels
 * with W < 2^20 and H <= 262140 (CRYCHIC_E_UNSUPPORTED beyond).
 */
#ifndef CRYCHIC_HIP_H
#define CRYCHIC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CRYCHIC_MAX_LIGHTS 16 /* Common/d3dUtil.h:226 */

typedef enum crychic_status {
    CRYCHIC_OK = 0,
    CRYCHIC_E_INVALID_ARG = -1,   /* null pointer, odd/zero size, bad range */
    CRYCHIC_E_NO_DEVICE = -2,     /* no HIP device / wrong ordinal */
    CRYCHIC_E_HIP = -3,           /* a HIP runtime call failed (see crychic_last_error) */
    CRYCHIC_E_UNSUPPORTED = -4,
    CRYCHIC_E_COMM = -5           /* RCCL failure */
} crychic_status;

/* Common/d3dUtil.h:216-224 (48 B) */
typedef struct crychic_light {
    float Strength[3];
    float FalloffStart;
    float Direction[3];
    float FalloffEnd;
    float Position[3];
    float SpotPower;
} crychic_light;

/* FrameResource.h:29-51 == cbPass Shaders/Common.hlsl:82-107 (2048 B).  Matrices are stored as the
 * reference stores them: XMMatrixTranspose of the row-vector matrix (CRYCHIC.cpp:843-849). */
typedef struct crychic_pass_constants {
    float View[16];
    float InvView[16];
    float Proj[16];
    float InvProj[16];
    float ViewProj[16];
    float InvViewProj[16];
    float ViewProjTex[16];
    float ShadowTransforms[12][16];
    float EyePosW[3];
    float cbPerObjectPad1;
    float RenderTargetSize[2];
    float InvRenderTargetSize[2];
    float NearZ;
    float FarZ;
    float TotalTime;
    float DeltaTime;
    float AmbientLight[4];
    crychic_light Lights[CRYCHIC_MAX_LIGHTS];
} crychic_pass_constants;

/* FrameResource.h:53-67 == cbSsao Shaders/Ssao.hlsl:5-22 (496 B) */
typedef struct crychic_ssao_constants {
    float Proj[16];
    float InvProj[16];
    float ProjTex[16];
    float OffsetVectors[14][4];
    float BlurWeights[3][4];
    float RenderTargetSize[2];
    float InvRenderTargetSize[2];
    float OcclusionRadius;
    float OcclusionFadeStart;
    float OcclusionFadeEnd;
    float SurfaceEpsilon;
} crychic_ssao_constants;

/* Camera state consumed by the constant builders (Common/Camera.h:20-97). */
typedef struct crychic_camera {
    float pos[3];
    float look[3];
    float up[3];
    float fovY, aspect, nearZ, farZ;
} crychic_camera;

typedef struct crychic_ctx crychic_ctx;

/* ---- context / errors ---------------------------------------------------------------------------- */
/* Replaces D3DApp::InitDirect3D's device creation (Common/d3dApp.cpp:415-479): binds the context to one
 * GPU.  One context per GPU, single-threaded, like the reference's one queue / one list. */
int crychic_ctx_create(int device_ordinal, crychic_ctx** out);
void crychic_ctx_destroy(crychic_ctx* ctx);
const char* crychic_last_error(void);
const char* crychic_version(void);
/* Name of the device the context is bound to ("gfx950..."); NULL on error. */
const char* crychic_ctx_device_name(crychic_ctx* ctx);

/* ---- host-side constant builders (no GPU needed) --------------------------------------------------- */
/* Ssao::CalcGaussWeights  Ssao.cpp:37-68.  Returns the number of weights written (2*ceil(2*sigma)+1) or a
 * negative status if capacity is too small / radius > Ssao::MaxBlurRadius (the reference asserts). */
int crychic_calc_gauss_weights(float sigma, float* weights, int capacity);
/* MathHelper::RandF over the MSVC CRT rand() LCG the reference links against (Common/MathHelper.h:17-20). */
int crychic_msvc_rand(uint32_t* state);
/* Ssao::BuildOffsetVectors  Ssao.cpp:423-462 */
void crychic_build_offset_vectors(uint32_t* rand_state, float offsets[14][4]);
/* Ssao::BuildRandomVectorTexture  Ssao.cpp:392-402 (texel bytes as the shader samples them). */
void crychic_build_random_vector_texture(uint32_t* rand_state, int args_right_to_left, uint8_t* rgba8_256x256);
/* CRYCHIC::UpdateCascadeShadowTransform  CRYCHIC.cpp:634-815.  Outputs are the untransposed row-vector
 * matrices (mLightViews / mLightProjs / mShadowTransforms). */
int crychic_update_cascade_shadow_transform(const crychic_camera* cam, const float lightDir[3],
                                            uint32_t shadowMapWidth, float lightView[4][16],
                                            float lightProj[4][16], float shadowTransform[4][16]);
/* CRYCHIC::UpdateMainPassCB  CRYCHIC.cpp:817-868 */
int crychic_update_main_pass_cb(const crychic_camera* cam, uint32_t W, uint32_t H,
                                const float shadowTransform[4][16], const float lightDirs[3][3],
                                crychic_pass_constants* out);
/* CRYCHIC::UpdateSsaoCB  CRYCHIC.cpp:903-937 */
int crychic_update_ssao_cb(const crychic_camera* cam, uint32_t W, uint32_t H, const float offsets[14][4],
                           crychic_ssao_constants* out);
/* Common.hlsl:305 `5 / width / 2.0f`.  literal != 0 keeps the shader's unsigned integer division (radius 0
 * for every width > 5: what the reference computes); literal == 0 gives the float division (2.5 texels). */
float crychic_pcf_search_radius(uint32_t shadowMapWidth, int literal);

/* ---- SSAO (Ssao.h / Ssao.cpp / Ssao.hlsl / SsaoBlur.hlsl) --------------------------------------------- */
/* Bytes of the edge workspace for a W x H frame: per-pixel centre normals / linear depths and recorded blur decisions (half
 * res), the decoded depth-pairs plane (full res) and the coarse maps of the exact shortcuts (csrc/ssao_core.hpp).
 * Contract: the workspace is caller-owned scratch with NO initialisation requirement and no meaning between frames -- it may
 * be freshly allocated, recycled from another context, or shared by consecutive frames of one stream.  Every word a frame
 * reads was written earlier in that same frame (the maps are stamped from one process-wide counter and the passes write the
 * stamp or a non-stamp into every word they will look at), so stale contents can never be mistaken for this frame's.  Parts
 * of it are deliberately left unwritten (depth-pairs entries under clear sky, which nothing reads): do not read it back. */
size_t crychic_edge_plane_bytes(uint32_t W, uint32_t H);

/* Ssao.hlsl:PS (Shaders/Ssao.hlsl:117-199) over half-res rows [row0, row0+rows): writes ambient_out and,
 * when edge_dev is not NULL, the per-pixel centre normal / linear depth that every later blur sweep of
 * this frame re-uses (SsaoBlur.hlsl:109-111,121-123 recompute them per tap). */
int crychic_ssao(crychic_ctx* ctx, const crychic_ssao_constants* cb, const void* normal_dev,
                 const uint32_t* depth_dev, const uint8_t* randvec_dev, uint16_t* ambient_out_dev,
                 void* edge_dev, uint32_t W, uint32_t H, uint32_t row0, uint32_t rows, void* stream);

/* Builds only the edge workspace (for callers that blur an ambient map they produced elsewhere). */
int crychic_ssao_edges(crychic_ctx* ctx, const crychic_ssao_constants* cb, const void* normal_dev,
                       const uint32_t* depth_dev, void* edge_dev, uint32_t W, uint32_t H, uint32_t row0,
                       uint32_t rows, void* stream);

/* One sweep of SsaoBlur.hlsl:PS == Ssao::BlurAmbientMap(cmdList, bool horzBlur) (Ssao.cpp:245-293) over
 * half-res rows [row0, row0+rows).  `horizontal` is the gHorizontalBlur root constant. */
int crychic_ssao_blur(crychic_ctx* ctx, const crychic_ssao_constants* cb, const void* edge_dev,
                      const uint16_t* ambient_in_dev, uint16_t* ambient_out_dev, uint32_t W, uint32_t H,
                      int horizontal, uint32_t row0, uint32_t rows, void* stream);

/* Ssao::ComputeSsao(cmdList, currFrame, blurCount)  Ssao.cpp:185-243: SSAO into ambient0, then blurCount x
 * (H: 0->1, V: 1->0); the final AO is in ambient0 (Ssao.cpp:75-78).  Rows [row0, row0+rows) of the final
 * map are guaranteed valid; the implementation recomputes a halo of 5 rows per remaining vertical sweep
 * (multi-GPU strips need no exchange).  row0 = 0, rows = H/2 is the whole map. */
int crychic_ssao_compute(crychic_ctx* ctx, const crychic_ssao_constants* cb, const void* normal_dev,
                         const uint32_t* depth_dev, const uint8_t* randvec_dev, uint16_t* ambient0_dev,
                         uint16_t* ambient1_dev, void* edge_dev, uint32_t W, uint32_t H, int blurCount,
                         uint32_t row0, uint32_t rows, void* stream);

/* ---- deferred lighting (DeferredShading.hlsl:PS, Shaders/DeferredShading.hlsl:23-101) ------------------ */
#define CRYCHIC_LIGHT_SKY 1u /* fill uncovered pixels from the cubemap (sky.hlsl:21-47) instead of the clear colour */
/* The reference's shader quirks are reproduced by default (SURVEY.md quirk checklist); each switch selects the evidently
 * intended form instead, for callers that want the fixed maths (parity is then against this repo's oracle with the same flag):
 *   Q1  DeferredShading.hlsl:60 `abs(distance - radius[j] < 5.0f)` (abs of a bool: always blends cascades j and j+1)
 *       -> `abs(distance - radius[j]) < 5.0f`, the form of the forward shader (Default.hlsl:131)
 *   Q3  PBR.hlsl:58,66 the specular denominator uses hDotv where nDotv is meant -> nDotl * nDotv
 *   Q4  PBR.hlsl:61-68 `ks * fs` with fs already holding F (Fresnel applied twice) -> kd * fd + fs */
/* The cube map's mip chain (the reference binds the whole chain, CRYCHIC.cpp:1148-1151, and samples it MIN_MAG_MIP_LINEAR,
 * :2617-2622): CRYCHIC_LIGHT_CUBE_LEVELS(n) in `flags` says that cube_dev holds n levels (crychic_load_dds_cube_rgba8_mips' layout:
 * level after level, each six faces of max(cubeDim >> level, 1)^2 RGBA8 texels); the reflection lookup (DeferredShading.hlsl:95)
 * and the sky (sky.hlsl:46) are then trilinear, the level of detail taken from the direction's differences inside the pixel's
 * 2 x 2 quad -- a BUILD DEFINITION (D3D leaves the arithmetic to the hardware; DESIGN.md section 3 states it), parity against
 * this repo's CPU checker.  n = 0 or 1: level 0 alone, the lookup of every earlier release.  With n > 1 a call's rows must be whole
 * quad rows (even row0; even rows unless they end the frame). */
#define CRYCHIC_LIGHT_CUBE_LEVELS(n) (((uint32_t)(n) & 15u) << 16)
#define CRYCHIC_FIX_Q1 0x100u
#define CRYCHIC_FIX_Q3 0x200u
#define CRYCHIC_FIX_Q4 0x400u

/* Full-screen replacement of the geometry re-draw at CRYCHIC.cpp:238-273 over full-res rows
 * [row0, row0+rows): pixels with depth < 1.0 are lit, the others get Colors::LightSteelBlue
 * (CRYCHIC.cpp:247) or the sky.  ambient_dev may be NULL (SSAO off, ambientAccess = 1); radiance_out_dev
 * (optional, 4 floats per pixel) receives litColor before UNORM8 quantisation.  numDirLights is the
 * shader's NUM_DIR_LIGHTS (1 in the reference build, Common.hlsl:6-8). */
int crychic_deferred_light(crychic_ctx* ctx, const crychic_pass_constants* cb, const float* g0_dev,
                           const float* g1_dev, const float* g2_dev, const uint32_t* depth_dev,
                           const uint16_t* ambient_dev, const uint32_t* const shadow_dev[4], uint32_t shadowDim,
                           const uint8_t* cube_dev, uint32_t cubeDim, uint8_t* out_rgba8_dev,
                           float* radiance_out_dev, uint32_t W, uint32_t H, uint32_t row0, uint32_t rows,
                           int numDirLights, float pcfSearchRadius, uint32_t flags, void* stream);

/* crychic_deferred_light plus `numPointLights` (<= 1024) point lights read from a device array of crychic_light
 * (Strength, FalloffStart, FalloffEnd, Position).  BUILD-DEFINED EXTENSION: the reference's NUM_POINT_LIGHTS branch
 * (PBR.hlsl:109-124) is dead code; it is enabled here as evidently intended (range test, l /= d, linear attenuation,
 * shadow factor 1) with per-tile light culling in LDS.  Parity is against this repo's oracle only. */
int crychic_deferred_light_points(crychic_ctx* ctx, const crychic_pass_constants* cb, const float* g0_dev,
                                  const float* g1_dev, const float* g2_dev, const uint32_t* depth_dev,
                                  const uint16_t* ambient_dev, const uint32_t* const shadow_dev[4], uint32_t shadowDim,
                                  const uint8_t* cube_dev, uint32_t cubeDim, uint8_t* out_rgba8_dev,
                                  float* radiance_out_dev, uint32_t W, uint32_t H, uint32_t row0, uint32_t rows,
                                  int numDirLights, float pcfSearchRadius, uint32_t flags,
                                  const crychic_light* point_lights_dev, uint32_t numPointLights, void* stream);

/* ---- whole hot path of CRYCHIC::Draw (CRYCHIC.cpp:220-221 + 238-279) -------------------------------------- */
typedef struct crychic_frame_desc {
    uint32_t W, H;
    int blurCount;              /* CRYCHIC.cpp:221 passes 3; <0 = SSAO off */
    int numDirLights;
    float pcfSearchRadius;
    uint32_t flags;             /* CRYCHIC_LIGHT_* */
    uint32_t row0, rows;        /* full-res output rows owned by this GPU (0, H = whole frame) */
    const void* normal_dev;
    const uint32_t* depth_dev;
    const uint8_t* randvec_dev;
    const float* g0_dev;
    const float* g1_dev;
    const float* g2_dev;
    const uint32_t* shadow_dev[4];
    uint32_t shadowDim;
    const uint8_t* cube_dev;
    uint32_t cubeDim;
    uint16_t* ambient0_dev;
    uint16_t* ambient1_dev;
    void* edge_dev;
    uint8_t* out_rgba8_dev;
    /* Extension (BASELINE configs[4], parity vs this repo's oracle only): point lights evaluated after the directional
     * ones, with tiled light culling; NULL / 0 = the reference configuration. */
    const crychic_light* point_lights_dev;
    uint32_t numPointLights;
} crychic_frame_desc;

int crychic_draw_hot_path(crychic_ctx* ctx, const crychic_ssao_constants* ssaoCB,
                          const crychic_pass_constants* passCB, const crychic_frame_desc* frame, void* stream);

/* Per-kernel timing of the last crychic_draw_hot_path issued with profiling enabled (HIP events recorded
 * on the caller's stream around each pass).  Times are milliseconds; blocks until the events complete. */
typedef struct crychic_pass_times {
    float ssao_ms;
    float blur_ms;   /* all 2*blurCount sweeps */
    float light_ms;
    float total_ms;
} crychic_pass_times;
int crychic_ctx_set_profiling(crychic_ctx* ctx, int enabled);
int crychic_ctx_last_pass_times(crychic_ctx* ctx, crychic_pass_times* out);


/* ---- producer passes: the draws that fill the hot path's input planes ------------------------- */
/* FrameResource.h:69-75 (44 B; input layout CRYCHIC.cpp:1241-1247) */
typedef struct crychic_vertex { float Pos[3]; float Normal[3]; float TexC[2]; float TangentU[3]; } crychic_vertex;
/* FrameResource.h:7-15 (144 B); matrices stored transposed as UpdateInstanceData writes them (CRYCHIC.cpp:546-547) */
typedef struct crychic_instance_data { float World[16]; float TexTransform[16]; uint32_t MaterialIndex; uint32_t pad[3]; } crychic_instance_data;
/* FrameResource.h:17-27 (112 B); MatTransform stored transposed (CRYCHIC.cpp:582) */
typedef struct crychic_material_data {
    float DiffuseAlbedo[4]; float FresnelR0[3]; float Roughness; float MatTransform[16];
    uint32_t DiffuseMapIndex; uint32_t NormalMapIndex; float Metalness; uint32_t pad;
} crychic_material_data;
/* One DrawIndexedInstanced(IndexCount, InstanceCount, StartIndexLocation, BaseVertexLocation, 0) with its bound vertex,
 * index (R32_UINT) and instance buffers (CRYCHIC::DrawRenderItems, CRYCHIC.cpp:2438-2475).  All pointers are device. */
typedef struct crychic_draw_item {
    const crychic_vertex* vertices_dev; uint32_t vertexCount;
    const uint32_t* indices_dev; uint32_t indexCount; uint32_t startIndexLocation; int32_t baseVertexLocation;
    const crychic_instance_data* instances_dev; uint32_t instanceCount;
} crychic_draw_item;
/* An R8G8B8A8_UNORM texture of gTextureMaps[] (Common.hlsl:52); rgba8_dev may be NULL (white / flat normal).  mipLevels > 1:
 * the levels follow one another in memory, level k being max(1, width >> k) x max(1, height >> k) texels (what
 * crychic_load_dds_rgba8_mips returns), and the G-buffer pass samples it as gsamAnisotropicWrap does (anisotropic 8, trilinear;
 * the kernel D3D leaves open is defined in csrc/raster_core.hpp).  mipLevels 0 or 1: level 0 only, bilinear. */
typedef struct crychic_texture { const uint8_t* rgba8_dev; uint32_t width, height, mipLevels; } crychic_texture;

/* GeometryGenerator::CreateBox / CreateGrid (Common/GeometryGenerator.cpp:10-101, 551-614) and the Models/<name>.txt loader
 * of CRYCHIC::BuildSkullGeometry (CRYCHIC.cpp:1447-1557), host memory.  Call with NULL buffers to get the counts.
 * Return the vertex count, or a negative status. */
int crychic_create_box(float width, float height, float depth, uint32_t numSubdivisions, crychic_vertex* vertices,
                       uint32_t vertexCapacity, uint32_t* indices, uint32_t indexCapacity, uint32_t* indexCount);
int crychic_create_grid(float width, float depth, uint32_t m, uint32_t n, crychic_vertex* vertices, uint32_t vertexCapacity,
                        uint32_t* indices, uint32_t indexCapacity, uint32_t* indexCount);
int crychic_load_mesh_text(const char* path, crychic_vertex* vertices, uint32_t vertexCapacity, uint32_t* indices,
                           uint32_t indexCapacity, uint32_t* vertexCount, uint32_t* indexCount);

/* CRYCHIC::UpdateInstanceData's frustum culling (CRYCHIC.cpp:515-564; mFrustumCullingEnabled defaults to true,
 * CRYCHIC.h:188): visible[i] = 1 when the render item's local-space bounding box (center, extents) under world matrix
 * worlds[16 i ..] (row-major, row-vector convention, untransposed) is not DISJOINT from the camera frustum.  Only visible
 * instances are copied to the frame's instance buffer, so a culled instance is also missing from the shadow pass.
 * Returns the number of visible instances (>= 0) or a negative status. */
int crychic_frustum_cull(const crychic_camera* cam, const float boundsCenter[3], const float boundsExtents[3],
                         const float* worlds, uint32_t count, uint8_t* visible);

/* Material textures: a DDS file holding DXT1, DXT5 or 32-bit-mask pixels (the formats of the six
 * textures CRYCHIC::LoadTextures opens, CRYCHIC.cpp:939-973) decoded on the host to the R8G8B8A8 mip-0 image that
 * crychic_draw_gbuffer samples.  NULL buffer: only *width / *height are written.  Replaces the subset of
 * Common/DDSTextureLoader.cpp + GPU block decompression the path depends on. */
int crychic_load_dds_rgba8(const char* path, uint8_t* rgba8, size_t capacityBytes, uint32_t* width, uint32_t* height);
/* The same with the file's whole mip chain (Common/DDSTextureLoader.cpp uploads the stored levels, it generates none): the
 * levels are written back to back, level k = max(1, w >> k) x max(1, h >> k); *mipLevels = the number of levels decoded.
 * NULL buffer: only the sizes are written; the bytes needed are the sum over the levels of 4 * w_k * h_k. */
int crychic_load_dds_rgba8_mips(const char* path, uint8_t* rgba8, size_t capacityBytes, uint32_t* width, uint32_t* height,
                                uint32_t* mipLevels);

/* The sky cube map CRYCHIC::LoadTextures opens (CRYCHIC.cpp:960,968: snowcube1024.dds through CreateDDSTextureFromFile12; bound
 * as a TextureCube, CRYCHIC.cpp:1148-1151): a DDS cube map -- DDSCAPS2_CUBEMAP with all six faces, or a DX10 header with
 * DDS_RESOURCE_MISC_TEXTURECUBE -- holding DXT1 / DXT5 / 32-bit pixels, decoded to the 6 x dim x dim R8G8B8A8 plane (faces +X, -X,
 * +Y, -Y, +Z, -Z stacked) that crychic_deferred_light / crychic_draw_hot_path take as the cube map.  Level 0 of every face: the
 * lighting and sky passes filter level 0.  NULL buffer: only *dim is written.  The 2-D loaders above refuse
 * a cube file and this one refuses a 2-D file (CRYCHIC_E_UNSUPPORTED). */
int crychic_load_dds_cube_rgba8(const char* path, uint8_t* rgba8, size_t capacityBytes, uint32_t* dim);
/* The same with the mip chain the file stores (CRYCHIC.cpp:1148-1151 binds every level): level after level, each level the six
 * faces of max(dim >> level, 1)^2 texels -- what CRYCHIC_LIGHT_CUBE_LEVELS(*mipLevels) announces to the lighting pass.  rgba8 = NULL
 * queries *dim and *mipLevels (capacity: sum over the levels of 6 * d * d * 4 bytes). */
int crychic_load_dds_cube_rgba8_mips(const char* path, uint8_t* rgba8, size_t capacityBytes, uint32_t* dim, uint32_t* mipLevels);

/* Present stand-in (row f3; the reference calls IDXGISwapChain::Present, CRYCHIC.cpp:294-297): writes a HOST R8G8B8A8
 * image as binary PPM (alpha dropped). */
int crychic_save_ppm(const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height);

/* Device workspace for one rasterised pass over `triangles` input triangles (sum over items of instanceCount *
 * indexCount / 3) into a W x H target: 8 bytes per target pixel plus SEVEN set-up slots (192 B + a 4-byte live word each)
 * per input triangle -- a triangle clipped against the six guard-band / depth planes becomes a fan of up to seven, and slots
 * are fixed by draw order so that depth ties resolve as D3D12 resolves them.  That is ~1.4 KB per triangle: ~450 MB for the
 * four fused 4096^2 cascades of the 81 k-triangle benchmark scene (4 x 81 k triangle-targets).  Only the slots a triangle
 * lists are written or read.  A workspace sized by a build older than the guard-band clipping (three slots per triangle) is
 * refused with CRYCHIC_E_INVALID_ARG, never overrun: always size it with this function of the library you link. */
size_t crychic_raster_workspace_bytes(uint64_t triangles, uint32_t W, uint32_t H);

/* The three producer passes.  Rasteriser state = CD3DX12_RASTERIZER_DESC(D3D12_DEFAULT): solid, cull back, clockwise
 * front, depth clip and guard-band clip (Common/d3dx12.h:203-216); depth LESS + write against a target cleared to 1.0 (:120-132);
 * top-left rule, pixel centres at +0.5, 1/256-pixel vertex snap.  `passCB` supplies View / ViewProj.  Every pass clears its
 * targets first, like the reference (CRYCHIC.cpp:2489-2492, 2526-2527, 2554-2556).
 *   crychic_draw_scene_to_shadow_map : one cascade of CRYCHIC::DrawSceneToShadowMap (CRYCHIC.cpp:2477-2510) with the
 *       shadow PSO's DepthBias / SlopeScaledDepthBias (CRYCHIC.cpp:1601-1603: 10000, 2.0)
 *   crychic_draw_normals_and_depth   : CRYCHIC::DrawNormalsAndDepth (CRYCHIC.cpp:2512-2543), DrawNormals.hlsl
 *   crychic_draw_gbuffer             : CRYCHIC::DrawGBuffer (CRYCHIC.cpp:2545-2571), GeometryPass.hlsl; depth_dev is
 *       the depth target of this pass (may alias the normals pass's: same geometry, same values) */
int crychic_draw_scene_to_shadow_map(crychic_ctx* ctx, const crychic_pass_constants* passCB, const crychic_draw_item* items,
                                     uint32_t nItems, uint32_t* shadow_dev, uint32_t shadowDim, int depthBias,
                                     float slopeScaledDepthBias, void* workspace_dev, size_t workspaceBytes, void* stream);
int crychic_draw_normals_and_depth(crychic_ctx* ctx, const crychic_pass_constants* passCB, const crychic_draw_item* items,
                                   uint32_t nItems, void* normal_dev, uint32_t* depth_dev, uint32_t W, uint32_t H,
                                   void* workspace_dev, size_t workspaceBytes, void* stream);
int crychic_draw_gbuffer(crychic_ctx* ctx, const crychic_pass_constants* passCB, const crychic_draw_item* items,
                         uint32_t nItems, const crychic_material_data* materials_dev, uint32_t nMaterials,
                         const crychic_texture* textures, uint32_t nTextures, float* g0_dev, float* g1_dev, float* g2_dev,
                         uint32_t* depth_dev, uint32_t W, uint32_t H, void* workspace_dev, size_t workspaceBytes, void* stream);

/* What the most recent producer pass on this context could not draw (its workspace must still be alive): waits for `stream`, then
 * *flags = CRYCHIC_RASTER_* bits, 0 when every triangle was rasterised.  Triangles with vertices far outside the viewport ARE
 * drawn: like the D3D12 runtime the rasteriser clips to 0 <= z <= w and to a guard band (+-2^21 pixels around the viewport centre;
 * its fixed-point edge functions are defined up to +-2^22).  What remains undrawable is a vertex whose position is not finite. */
#define CRYCHIC_RASTER_COORD_OVERFLOW 1u   /* a post-clip vertex outside the +-2^22 pixel range (a non-finite position): triangle dropped */
#define CRYCHIC_RASTER_BAD_INDEX 2u        /* an index pointed outside the item's vertex buffer: triangle dropped */
int crychic_raster_status(crychic_ctx* ctx, void* stream, uint32_t* flags);

/* Ssao::ComputeSsao's blur iterations 1 .. blurCount - 1 run as ONE launch whose tiles wait for their neighbours' previous iteration
 * through per-tile counters in the edge workspace (Ssao.cpp:231-266 issues one draw per sweep and relies on the barrier between
 * draws).  The wait is bounded: a workgroup that gives up -- which would mean the device did not dispatch workgroups in grid order,
 * the assumption forward progress rests on -- records it and goes on.  This call synchronises `stream` and reports whether that
 * happened in the most recent chain issued on `ctx` (*timed_out = 1: that frame's ambient map is wrong).  Tests and bench.py
 * check it; CRYCHIC_BLUR_PER_ITERATION=1 in the environment selects one launch per iteration instead (the same pixels). */
int crychic_blur_chain_status(crychic_ctx* ctx, void* stream, uint32_t* timed_out);

/* All cascades of CRYCHIC::DrawSceneToShadowMap (CRYCHIC.cpp:2477-2510 loops over four) in one pass: passCBs[c].ViewProj and
 * shadow_dev[c] per cascade, the same items for all.  Bit-identical to nCascades calls of crychic_draw_scene_to_shadow_map;
 * the workspace must hold crychic_raster_workspace_bytes(nCascades * triangles, shadowDim, shadowDim). */
int crychic_draw_scene_to_shadow_maps(crychic_ctx* ctx, const crychic_pass_constants* passCBs, uint32_t nCascades,
                                      const crychic_draw_item* items, uint32_t nItems, uint32_t* const* shadow_dev, uint32_t shadowDim,
                                      int depthBias, float slopeScaledDepthBias, void* workspace_dev, size_t workspaceBytes, void* stream);

/* DrawNormalsAndDepth + DrawGBuffer in one rasterisation: the two passes draw the same items with the same ViewProj into
 * depth targets cleared to 1.0, so their visibility is identical; this entry rasterises once and runs both pixel shaders
 * on the winning primitive.  Every plane is bit-identical to calling the two passes one after the other. */
int crychic_draw_normals_depth_and_gbuffer(crychic_ctx* ctx, const crychic_pass_constants* passCB, const crychic_draw_item* items,
                                           uint32_t nItems, const crychic_material_data* materials_dev, uint32_t nMaterials,
                                           const crychic_texture* textures, uint32_t nTextures, void* normal_dev, float* g0_dev,
                                           float* g1_dev, float* g2_dev, uint32_t* depth_dev, uint32_t W, uint32_t H,
                                           void* workspace_dev, size_t workspaceBytes, void* stream);

/* Strip-limited producers (SURVEY.md 8e "G-buffer for own strip only"; the reference scissors its passes to the whole client
 * area, CRYCHIC.cpp:2547-2548 -- with N GPUs each rank scissors the G-buffer pass to the rows it lights).  As the two entry
 * points above, but G0..G2 are produced for rows [gRow0, gRow0 + gRows) only; G-buffer texels outside those rows are left
 * untouched.  crychic_draw_gbuffer_rows also limits rasterisation and its depth target to the rows;
 * crychic_draw_normals_depth_and_gbuffer_rows still renders depth and view normals for the WHOLE frame (the SSAO taps of a
 * strip reach far outside it) and only skips the G-buffer stage elsewhere.  Inside the rows every plane is bit-identical to
 * the unscissored pass. */
int crychic_draw_gbuffer_rows(crychic_ctx* ctx, const crychic_pass_constants* passCB, const crychic_draw_item* items,
                              uint32_t nItems, const crychic_material_data* materials_dev, uint32_t nMaterials,
                              const crychic_texture* textures, uint32_t nTextures, float* g0_dev, float* g1_dev, float* g2_dev,
                              uint32_t* depth_dev, uint32_t W, uint32_t H, uint32_t gRow0, uint32_t gRows, void* workspace_dev,
                              size_t workspaceBytes, void* stream);
int crychic_draw_normals_depth_and_gbuffer_rows(crychic_ctx* ctx, const crychic_pass_constants* passCB, const crychic_draw_item* items,
                                                uint32_t nItems, const crychic_material_data* materials_dev, uint32_t nMaterials,
                                                const crychic_texture* textures, uint32_t nTextures, void* normal_dev, float* g0_dev,
                                                float* g1_dev, float* g2_dev, uint32_t* depth_dev, uint32_t W, uint32_t H,
                                                uint32_t gRow0, uint32_t gRows, void* workspace_dev, size_t workspaceBytes, void* stream);

/* ---- multi-GPU strip plan (pure host arithmetic) ---------------------------------------------- */
/* Full-res rows [*row0, *row0 + *rows) owned by `rank` of `nranks` for an H-row frame: strips are multiples
 * of 2 rows (half-res alignment); the last rank takes the remainder. */
int crychic_strip_rows(uint32_t H, int nranks, int rank, uint32_t* row0, uint32_t* rows);

/* ---- multi-GPU exchange (SURVEY.md 8b `crychic_allgather_frame`, 8e): RCCL over xGMI ----------------------------- */
/* The reference is single-GPU (NodeMask 0, CRYCHIC.cpp:96,105); these entry points complete the back buffer that
 * CRYCHIC::Draw presents (CRYCHIC.cpp:282-297) when N GPUs each render a row strip of it.  One communicator per GPU.
 * Every rank renders its strip IN PLACE into a full W x H RGBA8 frame buffer; the exchange fills in the peers' strips,
 * stream-ordered behind the strip's lighting pass, with no host synchronisation.  Any RCCL failure returns
 * CRYCHIC_E_COMM (crychic_last_error() carries RCCL's own message). */
typedef struct crychic_comm crychic_comm;
#define CRYCHIC_COMM_ID_BYTES 128

/* One process per GPU: rank 0 obtains a rendezvous id (ncclGetUniqueId), hands it to its peers out of band (file, socket,
 * key-value store), then every rank calls crychic_comm_create (ncclCommInitRank: blocks until all nranks have joined). */
int crychic_comm_unique_id(uint8_t id[CRYCHIC_COMM_ID_BYTES]);
int crychic_comm_create(crychic_ctx* ctx, int nranks, int rank, const uint8_t id[CRYCHIC_COMM_ID_BYTES], crychic_comm** out);
/* One process driving all GPUs (the reference's own shape: one WinMain, one thread): communicators for ctxs[0..nranks),
 * rank k on ctxs[k]'s device (ncclCommInitAll). */
int crychic_comm_create_all(crychic_ctx* const* ctxs, int nranks, crychic_comm** out);
void crychic_comm_destroy(crychic_comm* comm);
/* Tears the communicator down without waiting for outstanding operations (ncclCommAbort): the way out of an exchange a
 * peer never joined.  The handle stays valid only for crychic_comm_destroy. */
int crychic_comm_abort(crychic_comm* comm);
int crychic_comm_rank(const crychic_comm* comm);
int crychic_comm_size(const crychic_comm* comm);
/* 0 while no asynchronous RCCL error is pending on the communicator, else CRYCHIC_E_COMM. */
int crychic_comm_async_error(crychic_comm* comm);

/* All-gather of the composed strips.  bounds = NULL: the crychic_strip_rows plan; otherwise nranks x (row0, rows) pairs that
 * tile the frame in rank order (any heights: cost-balanced strips).  Equal strips run as one in-place ncclAllGather, ragged
 * ones as one group of in-place ncclBroadcasts (one per strip).  Must be called by every rank, in the same order. */
int crychic_allgather_frame(crychic_comm* comm, uint8_t* frame_rgba8_dev, uint32_t W, uint32_t H, const uint32_t* bounds,
                            void* stream);
/* The same for a single host thread that owns every rank: frames_rgba8_dev[k] / streams[k] belong to comms[k]. */
int crychic_allgather_frame_all(crychic_comm* const* comms, int nranks, uint8_t* const* frames_rgba8_dev, uint32_t W, uint32_t H,
                                const uint32_t* bounds, void* const* streams);
/* SURVEY.md 8e "overlap by chunking the strip and gathering on a side stream": crychic_draw_hot_path for this rank's strip
 * (frame->row0 / rows must be this rank's entry of the plan, frame->out_rgba8_dev the full W x H frame) AND the exchange, with the
 * lighting pass issued in `nparts` (1 .. CRYCHIC_MAX_EXCHANGE_PARTS) consecutive row ranges: as soon as range p is lit, range p of
 * every rank's strip travels (one group of in-place ncclBroadcasts on a stream the communicator owns) while the caller's stream
 * lights range p + 1.  Ranges are whole half-res rows -- the crychic_strip_rows cut of each strip's rows -- so the image is the
 * one crychic_draw_hot_path + crychic_allgather_frame produce, bit for bit.  On return the caller's stream is ordered behind the last
 * range's exchange (no host wait).  nparts == 1 is exactly those two calls on the caller's stream.  Every rank must call it with
 * the same bounds and nparts, in the same order relative to the communicator's other collectives.  One process per GPU only. */
#define CRYCHIC_MAX_EXCHANGE_PARTS 8
int crychic_draw_hot_path_shared(crychic_comm* comm, const crychic_ssao_constants* ssaoCB, const crychic_pass_constants* passCB,
                                 const crychic_frame_desc* frame, const uint32_t* bounds, uint32_t nparts, void* stream);
/* Stream-ordered rendezvous of all ranks (a one-word ncclAllReduce): brackets timed regions; no host wait inside. */
int crychic_comm_barrier(crychic_comm* comm, void* stream);

#ifdef __cplusplus
}
#endif
#endif
