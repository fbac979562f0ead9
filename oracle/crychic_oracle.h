/*
 * crychic_oracle.h -- CPU oracle for the CRYCHIC deferred-shading hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the shipped product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
 * checker (or as the timed CPU baseline), never as the thing measured or shipped.
 *
 * PARITY UNPINNED.  The reference (UnlimitedRainWorks/CRYCHIC-RENDERER) is a Windows/D3D12 demo with no
 * tests, golden vectors or CPU path, and it cannot be compiled here (needs windows.h, d3d12.h,
 * DirectXMath.h and the HLSL compiler).  This file set is a literal restatement of the reference's
 * HLSL/C++ *source text*; each function cites the file:line it follows.  Where HLSL/D3D leaves the
 * arithmetic implementation-defined (rcp/rsqrt/pow/sin precision, mul() summation order, sampler
 * weight precision, texel-edge point sampling) the oracle DEFINES one IEEE-754 binary32 evaluation
 * order, listed in DESIGN.md section "Oracle definitions".  The only pinned answers are the derived
 * known-answer values of SURVEY.md Appendix C (tests/test_oracle_kat.py).
 *
 * All matrices in constant structs are stored exactly as the reference stores them: the transpose of
 * the row-vector matrix, row-major floats (CRYCHIC.cpp:843-849,918), so HLSL `mul(v, M)[j]` is
 * sum_i v[i] * mem[4*j + i] and HLSL `M[r][c]` is mem[4*c + r].
 */
#ifndef CRYCHIC_ORACLE_H
#define CRYCHIC_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OR_MAX_LIGHTS 16 /* Common/d3dUtil.h:226, Shaders/LightingUtil.hlsl:7 */

/* Common/d3dUtil.h:216-224 */
typedef struct or_light {
    float Strength[3];
    float FalloffStart;
    float Direction[3];
    float FalloffEnd;
    float Position[3];
    float SpotPower;
} or_light; /* 48 B */

/* FrameResource.h:29-51 == cbPass, Shaders/Common.hlsl:82-107 */
typedef struct or_pass_constants {
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
    or_light Lights[OR_MAX_LIGHTS];
} or_pass_constants; /* 2048 B */

/* FrameResource.h:53-67 == cbSsao, Shaders/Ssao.hlsl:5-22 */
typedef struct or_ssao_constants {
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
} or_ssao_constants; /* 496 B */

/* ---- host-side constant builders -------------------------------------------------------------- */

/* MSVC CRT rand(): x = x*214013 + 2531011; return (x >> 16) & 0x7FFF (RAND_MAX 0x7FFF).  The reference
 * never seeds (Common/MathHelper.h:17-20), so the state starts at 1. */
int   or_msvc_rand(uint32_t* state);
/* MathHelper::RandF()  Common/MathHelper.h:17-20 */
float or_randf(uint32_t* state);
/* MathHelper::RandF(a,b)  Common/MathHelper.h:23-26 */
float or_randf_range(uint32_t* state, float a, float b);

/* Ssao::CalcGaussWeights  Ssao.cpp:37-68.  Writes 2*ceil(2*sigma)+1 weights, returns the count. */
int or_calc_gauss_weights(float sigma, float* weights, int capacity);

/* Ssao::BuildOffsetVectors  Ssao.cpp:423-462 (14 float4, w = 0). */
void or_build_offset_vectors(uint32_t* rand_state, float offsets[14][4]);

/* Ssao::BuildRandomVectorTexture  Ssao.cpp:392-402: 256x256 texels as the shader sees them
 * (R8G8B8A8 byte order).  XMCOLOR(v.x,v.y,v.z,0) packs A8R8G8B8, i.e. bytes B,G,R,A in memory, so the
 * shader's .r is v.z.  `args_right_to_left` selects the (unspecified) evaluation order of the three
 * RandF() constructor arguments at Ssao.cpp:398 (MSVC x64: right to left = 1). */
void or_build_random_vector_texture(uint32_t* rand_state, int args_right_to_left, uint8_t* rgba8_256x256);

/* LH matrices as DirectXMath defines them (row-vector convention, row-major storage; NOT transposed). */
void or_mat_perspective_fov_lh(float fovY, float aspect, float zn, float zf, float out[16]);
void or_mat_look_at_lh(const float eye[3], const float at[3], const float up[3], float out[16]);
void or_mat_ortho_off_center_lh(float l, float r, float b, float t, float zn, float zf, float out[16]);
void or_mat_mul(const float a[16], const float b[16], float out[16]);
int  or_mat_inverse(const float m[16], float out[16]);
void or_mat_transpose(const float m[16], float out[16]);

/* Camera description used by the constant builders (Common/Camera.cpp:116-128,226-273). */
typedef struct or_camera {
    float pos[3];
    float look[3];
    float up[3];
    float fovY, aspect, nearZ, farZ;
} or_camera;

/* CRYCHIC::UpdateCascadeShadowTransform  CRYCHIC.cpp:634-815.  Produces (untransposed) light view, light
 * projection and shadow transform for the 4 cascades. */
void or_cascade_shadow_transforms(const or_camera* cam, const float lightDir[3], uint32_t shadowMapWidth,
                                  float lightView[4][16], float lightProj[4][16], float shadowTransform[4][16]);

/* CRYCHIC::UpdateMainPassCB  CRYCHIC.cpp:817-868 (ShadowTransforms[4..11] are zero-filled here; the
 * reference copies uninitialised members, CRYCHIC.cpp:837-841). */
int or_frustum_cull(const or_camera* cam, const float center[3], const float extents[3], const float* worlds, uint32_t count,
                    uint8_t* visible, double* margin);
void or_build_pass_constants(const or_camera* cam, uint32_t W, uint32_t H, const float shadowTransform[4][16],
                             const float lightDirs[3][3], or_pass_constants* out);

/* CRYCHIC::UpdateSsaoCB  CRYCHIC.cpp:903-937 */
void or_build_ssao_constants(const or_camera* cam, uint32_t W, uint32_t H, const float offsets[14][4],
                             or_ssao_constants* out);

/* ---- per-pixel passes -------------------------------------------------------------------------- */
/* Plane layouts (row-major, pitch = width * bytes-per-texel):
 *   depth   uint32, D24 in bits 0..23            W x H
 *   normal  4 x fp16 (view-space normal, w)      W x H
 *   ambient uint16 UNORM                          (W/2) x (H/2)
 *   randvec 4 x uint8 UNORM                       256 x 256
 *   G0..G2  4 x fp32                              W x H
 *   shadow  uint32 D24                            4 x shadowDim x shadowDim (one pointer per cascade)
 *   cube    4 x uint8 UNORM                       6 x cubeDim x cubeDim (+X,-X,+Y,-Y,+Z,-Z)
 *   out     4 x uint8 UNORM RGBA                  W x H
 * Row ranges [row0, row0+rows) are in units of the pass's own output rows. */

/* Shaders/Ssao.hlsl:117-199 (VS :58-72).  Output rows are half-res rows. */
void or_ssao(const or_ssao_constants* cb, const uint16_t* normal, const uint32_t* depth, const uint8_t* randvec,
             uint32_t W, uint32_t H, uint16_t* ambient_out, uint32_t row0, uint32_t rows);

/* Shaders/SsaoBlur.hlsl:85-146.  horizontal = gHorizontalBlur root constant. */
void or_ssao_blur(const or_ssao_constants* cb, const uint16_t* normal, const uint32_t* depth,
                  const uint16_t* ambient_in, uint16_t* ambient_out, uint32_t W, uint32_t H, int horizontal,
                  uint32_t row0, uint32_t rows);

/* Ssao::ComputeSsao  Ssao.cpp:185-243: SSAO into ambient0 then blurCount x (H: 0->1, V: 1->0). */
void or_compute_ssao(const or_ssao_constants* cb, const uint16_t* normal, const uint32_t* depth,
                     const uint8_t* randvec, uint32_t W, uint32_t H, uint16_t* ambient0, uint16_t* ambient1,
                     int blurCount);

/* Common.hlsl:305 `5 / width / 2.0f`: literal = 1 keeps the uint division (0 for width > 5), literal = 0
 * is the evidently intended float division (2.5 texels). */
float or_pcf_search_radius(uint32_t shadowWidth, int literal);

/* Shaders/DeferredShading.hlsl:23-101 as a full-screen pass masked by depth < 1.
 * ambient may be NULL (SSAO off: ambientAccess = 1).  radiance_out (optional) receives litColor before
 * UNORM8 quantisation as 4 floats per pixel.  `sky` is a flag word: bit 0 = uncovered pixels get the sky cubemap along
 * the view ray (Shaders/sky.hlsl:21-47) instead of the clear colour (CRYCHIC.cpp:247); 0x100 / 0x200 / 0x400 select the
 * evidently intended forms of quirks Q1 / Q3 / Q4 (same bits as CRYCHIC_FIX_* of the product's ABI); 0 = as written;
 * bits 16..19 = the number of mip levels `cube` holds (CRYCHIC_LIGHT_CUBE_LEVELS; 0 / 1 = level 0 alone): with more than one the
 * reflection and sky lookups are trilinear with quad derivatives (or_samplers.h "TextureCube.Sample with a mip chain"). */
void or_deferred_light(const or_pass_constants* cb, const float* g0, const float* g1, const float* g2,
                       const uint32_t* depth, const uint16_t* ambient, const uint32_t* const shadow[4],
                       uint32_t shadowDim, const uint8_t* cube, uint32_t cubeDim, uint8_t* out_rgba8,
                       float* radiance_out, uint32_t W, uint32_t H, uint32_t row0, uint32_t rows,
                       int numDirLights, float pcfSearchRadius, int sky);


/* ---- producer passes: D3D-rules software rasteriser ---------------------------------------- */
/* FrameResource.h:69-75 */
typedef struct or_vertex { float Pos[3]; float Normal[3]; float TexC[2]; float TangentU[3]; } or_vertex;          /* 44 B */
/* FrameResource.h:7-15, matrices stored transposed (CRYCHIC.cpp:546-547) */
typedef struct or_instance_data { float World[16]; float TexTransform[16]; uint32_t MaterialIndex; uint32_t pad[3]; } or_instance_data; /* 144 B */
/* FrameResource.h:17-27, MatTransform stored transposed (CRYCHIC.cpp:582) */
typedef struct or_material_data {
    float DiffuseAlbedo[4]; float FresnelR0[3]; float Roughness; float MatTransform[16];
    uint32_t DiffuseMapIndex; uint32_t NormalMapIndex; float Metalness; uint32_t pad;
} or_material_data; /* 112 B */
/* One DrawIndexedInstanced (CRYCHIC.cpp:2473). */
typedef struct or_draw_item {
    const or_vertex* vertices; uint32_t vertexCount;
    const uint32_t* indices; uint32_t indexCount; uint32_t startIndexLocation; int32_t baseVertexLocation;
    const or_instance_data* instances; uint32_t instanceCount;
} or_draw_item;
/* mipLevels > 1: levels back to back, level k = max(1, w >> k) x max(1, h >> k); sampled with the anisotropic kernel defined in
 * or_raster.c (sample_texture).  0 or 1: level 0 only, bilinear. */
typedef struct or_texture { const uint8_t* rgba8; uint32_t width, height, mipLevels; } or_texture;

/* Rasterises the items with the default rasteriser state (solid, cull back, clockwise = front, depth clip;
 * Common/d3dx12.h:203-216), depth LESS + write against depth cleared to 1.0, top-left rule, pixel centres at +0.5,
 * vertex positions snapped to 1/256 pixel.  Returns the number of setup triangles (after clipping/culling) or -1.
 * mode 0: Shadows.hlsl -- depth only, with depthBias / slopeScaledDepthBias (CRYCHIC.cpp:1601-1603)
 * mode 1: DrawNormals.hlsl -- view-space normal (4 x fp16, w = 0) + depth; clear normal (0,0,1,0) (Ssao.cpp:317)
 * mode 2: GeometryPass.hlsl -- G0..G2 (clear 0, CRYCHIC.cpp:2554) + depth
 * view / viewProj are the transposed cbuffer matrices (PassConstants.View / .ViewProj). */
int or_rasterize(int mode, const float view[16], const float viewProj[16], const or_draw_item* items, uint32_t nItems,
                 const or_material_data* materials, uint32_t nMaterials, const or_texture* textures, uint32_t nTextures,
                 uint32_t W, uint32_t H, int depthBias, float slopeScaledDepthBias,
                 uint32_t* depth_out, uint16_t* normal_out, float* g0, float* g1, float* g2);

/* GeometryGenerator::CreateBox / CreateGrid (Common/GeometryGenerator.cpp:10-101, 214-300, 551-614).  Return the
 * vertex count and write the index count; -1 if the capacities are too small. */
int or_create_box(float width, float height, float depth, uint32_t numSubdivisions, or_vertex* v, uint32_t vcap,
                  uint32_t* idx, uint32_t icap, uint32_t* nIdx);
int or_create_grid(float width, float depth, uint32_t m, uint32_t n, or_vertex* v, uint32_t vcap, uint32_t* idx,
                   uint32_t icap, uint32_t* nIdx);
/* The "pos normal" + triangle-list text format of Models/skull.txt as CRYCHIC::BuildSkullGeometry parses it
 * (CRYCHIC.cpp:1447-1557), tangent generation included.  Pass NULL buffers to query the counts. */
int or_load_mesh_text(const char* path, or_vertex* v, uint32_t vcap, uint32_t* idx, uint32_t icap, uint32_t* nVerts,
                      uint32_t* nIdx);
uint16_t or_float_to_half(float f);
/* DDS (DXT1 / DXT5 / 32-bit masks) -> R8G8B8A8 mip 0 (row f4); NULL buffer queries the size.  0 on success. */
int or_load_dds_rgba8(const char* path, uint8_t* rgba8, size_t capacity, uint32_t* width, uint32_t* height);
/* The same with the file's mip chain (levels back to back); *mips = number of levels decoded. */
int or_load_dds_rgba8_mips(const char* path, uint8_t* rgba8, size_t capacity, uint32_t* width, uint32_t* height, uint32_t* mips);
/* A DDS cube map (legacy caps2 or DX10 header): level 0 of the faces +X, -X, +Y, -Y, +Z, -Z as one 6 x dim x dim R8G8B8A8 plane. */
int or_load_dds_cube_rgba8(const char* path, uint8_t* rgba8, size_t capacity, uint32_t* dim);
/* The same with the chain the file stores, level after level (each six faces of max(dim >> level, 1)^2 texels). */
int or_load_dds_cube_rgba8_mips(const char* path, uint8_t* rgba8, size_t capacity, uint32_t* dim, uint32_t* mips);

/* or_deferred_light plus NUM_POINT_LIGHTS point lights from a separate buffer: BUILD-DEFINED EXTENSION for BASELINE
 * configs[4] (the reference's point-light branch, PBR.hlsl:109-124, is dead code); see or_light.c. */
void or_deferred_light_points(const or_pass_constants* cb, const float* g0, const float* g1, const float* g2,
                              const uint32_t* depth, const uint16_t* ambient, const uint32_t* const shadow[4],
                              uint32_t shadowDim, const uint8_t* cube, uint32_t cubeDim, uint8_t* out_rgba8,
                              float* radiance_out, uint32_t W, uint32_t H, uint32_t row0, uint32_t rows,
                              int numDirLights, float pcfSearchRadius, int sky, const or_light* pointLights, uint32_t numPointLights);

/* Exposed pieces (unit-tested individually). */
void  or_eval_array(int kind, size_t n, const float* in, const float* in2, float* out);
float or_det_sinf(float x);
float or_det_cosf(float x);
float or_det_log2f(float x);
float or_det_exp2f(float x);
float or_det_powf(float x, float y);
float or_half_to_float(uint16_t h);
float or_d24_to_float(uint32_t d24);
float or_ndc_depth_to_view_depth(const or_ssao_constants* cb, float z_ndc);
float or_nrand(float u, float v);
float or_sample_depth_linear_border(const uint32_t* depth, uint32_t W, uint32_t H, float u, float v);
float or_sample_shadow_cmp(const uint32_t* shadow, uint32_t dim, float u, float v, float ref);
float or_pcf_poisson(const uint32_t* shadow, uint32_t dim, const float shadowPosH[4], float searchRadius);
void  or_sample_cube(const uint8_t* cube, uint32_t dim, const float dir[3], float rgb[3]);
float or_sample_cube_lod(uint32_t dim, uint32_t levels, const float dir[3], const float ddx[3], const float ddy[3]);
void  or_sample_cube_level(const uint8_t* chain, uint32_t dim, uint32_t levels, const float dir[3], float lod, float rgb[3]);
void  or_sample_randvec(const uint8_t* randvec, float u, float v, float rgb[3]);
float or_sample_ambient_linear_clamp(const uint16_t* ambient, uint32_t w2, uint32_t h2, float u, float v);

int or_num_threads(void);
void or_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
