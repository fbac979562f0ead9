// CRYCHIC.h -- headless counterpart of the reference application class (CRYCHIC.h:56-190, CRYCHIC.cpp) for the
// deferred hot path: same member names (mSsao, mDeferred, mShadowMap, mMainPassCB, mFrameResources ...), same
// Initialize / OnResize / Update / Draw call sequence, D3D12 replaced by libcrychic_hip.so.
//
// What Draw() covers: the whole deferred branch of CRYCHIC::Draw -- DrawSceneToShadowMap, DrawNormalsAndDepth
// (CRYCHIC.cpp:208,214), ComputeSsao (:220-221), DrawGBuffer (:236) and the deferred lighting + sky (:238-279) -- on
// the reference's live scene (100 instanced boxes + grid, CRYCHIC.cpp:2274-2436).  Set mRunProducerPasses = false to
// feed externally produced planes through the Resource accessors instead.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>
#include "DeferredShading.h"
#include "FrameResource.h"
#include "ShadowMap.h"
#include "Ssao.h"

const int gNumFrameResources = 3;  // CRYCHIC.h:20

class GameTimer {  // Common/GameTimer.h: only what Update/Draw read
public:
    float TotalTime() const { return mTotal; }
    float DeltaTime() const { return mDelta; }
    void Tick(float dt) { mDelta = dt; mTotal += dt; }
private:
    float mTotal = 0.0f, mDelta = 0.0f;
};

class Camera {  // Common/Camera.h:20-97: position / look / up state; the matrices are rebuilt from it by the C ABI's builders
public:
    void SetPosition(float x, float y, float z) { mCam.pos[0] = x; mCam.pos[1] = y; mCam.pos[2] = z; }
    void SetLens(float fovY, float aspect, float zn, float zf) { mCam.fovY = fovY; mCam.aspect = aspect; mCam.nearZ = zn; mCam.farZ = zf; }
    void LookTo(float lx, float ly, float lz, float ux, float uy, float uz) { mCam.look[0] = lx; mCam.look[1] = ly; mCam.look[2] = lz; mCam.up[0] = ux; mCam.up[1] = uy; mCam.up[2] = uz; }
    void LookAt(const DirectX::XMFLOAT3& pos, const DirectX::XMFLOAT3& target, const DirectX::XMFLOAT3& up)   // Camera.cpp:131-154
    {
        SetPosition(pos.x, pos.y, pos.z);
        LookTo(target.x - pos.x, target.y - pos.y, target.z - pos.z, up.x, up.y, up.z);
        UpdateViewMatrix();
    }
    DirectX::XMFLOAT3 GetPosition3f() const { return { mCam.pos[0], mCam.pos[1], mCam.pos[2] }; }
    DirectX::XMFLOAT3 GetLook3f() const { return { mCam.look[0], mCam.look[1], mCam.look[2] }; }
    DirectX::XMFLOAT3 GetUp3f() const { return { mCam.up[0], mCam.up[1], mCam.up[2] }; }
    DirectX::XMFLOAT3 GetRight3f() const { float r[3]; Cross(mCam.up, mCam.look, r); Normalize(r); return { r[0], r[1], r[2] }; }
    void Walk(float d) { for (int i = 0; i < 3; ++i) mCam.pos[i] += d * mCam.look[i]; }                    // Camera.cpp:190-199
    void Strafe(float d) { float r[3]; Cross(mCam.up, mCam.look, r); Normalize(r); for (int i = 0; i < 3; ++i) mCam.pos[i] += d * r[i]; }  // :179-188
    void Pitch(float angle)                                                                                // :201-211, about the right vector
    {
        float r[3]; Cross(mCam.up, mCam.look, r); Normalize(r);
        Rotate(mCam.up, r, angle); Rotate(mCam.look, r, angle);
    }
    void RotateY(float angle)                                                                              // :213-224, about the world y axis
    {
        const float y[3] = { 0.0f, 1.0f, 0.0f };
        Rotate(mCam.up, y, angle); Rotate(mCam.look, y, angle);
    }
    void UpdateViewMatrix()                                                                                // :226-273: re-orthonormalise
    {
        float r[3], u[3];
        Normalize(mCam.look);
        Cross(mCam.up, mCam.look, r); Normalize(r);
        Cross(mCam.look, r, u); Normalize(u);
        for (int i = 0; i < 3; ++i) mCam.up[i] = u[i];
    }
    float GetNearZ() const { return mCam.nearZ; }
    float GetFarZ() const { return mCam.farZ; }
    float GetFovY() const { return mCam.fovY; }
    float GetAspect() const { return mCam.aspect; }
    const crychic_camera& Raw() const { return mCam; }
private:
    static void Cross(const float a[3], const float b[3], float o[3]) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0]; }
    static void Normalize(float v[3]) { const float l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); v[0] /= l; v[1] /= l; v[2] /= l; }
    static void Rotate(float v[3], const float axis[3], float angle)   // Rodrigues; left-handed like XMMatrixRotationAxis (clockwise looking along the axis towards the origin)
    {
        const float c = std::cos(angle), s = std::sin(angle);
        float k[3]; Cross(axis, v, k);
        const float d = (axis[0] * v[0] + axis[1] * v[1] + axis[2] * v[2]) * (1.0f - c);
        for (int i = 0; i < 3; ++i) v[i] = v[i] * c + k[i] * s + axis[i] * d;
    }
    crychic_camera mCam = { { 0, 0, 0 }, { 0, 0, 1 }, { 0, 1, 0 }, 0.7853981634f, 1.0f, 1.0f, 1000.0f };
};

// Common/d3dUtil.h:163-214 (the fields the draws consume)
struct BoundingBox { DirectX::XMFLOAT3 Center = { 0, 0, 0 }, Extents = { 1, 1, 1 }; };   // DirectXCollision's, as far as the culling needs it
struct SubmeshGeometry {
    UINT IndexCount = 0;
    UINT StartIndexLocation = 0;
    int BaseVertexLocation = 0;
    BoundingBox Bounds;
};
struct MeshGeometry {
    std::string Name;
    std::unique_ptr<ID3D12Resource> VertexBufferGPU, IndexBufferGPU;
    UINT VertexCount = 0, IndexCount = 0;
    std::unordered_map<std::string, SubmeshGeometry> DrawArgs;
};
// Common/d3dUtil.h:240-265
struct Material {
    std::string Name;
    int MatCBIndex = -1;
    int DiffuseSrvHeapIndex = -1;
    int NormalSrvHeapIndex = -1;
    int NumFramesDirty = gNumFrameResources;
    DirectX::XMFLOAT4 DiffuseAlbedo = { 1.0f, 1.0f, 1.0f, 1.0f };
    DirectX::XMFLOAT3 FresnelR0 = { 0.01f, 0.01f, 0.01f };
    float Roughness = .25f;
    DirectX::XMFLOAT4X4 MatTransform = crychic_detail::Identity4x4();
};
// CRYCHIC.h:23-42
struct RenderItem {
    RenderItem() = default;
    RenderItem(const RenderItem& rhs) = delete;
    Material* Mat = nullptr;
    MeshGeometry* Geo = nullptr;
    UINT IndexCount = 0;
    UINT StartIndexLocation = 0;
    int BaseVertexLocation = 0;
    UINT InstanceCount = 0;
    std::vector<InstanceData> Instances;
    UINT itemIndex = 0;
    BoundingBox Bounds;
};
enum class RenderLayer : int { Opaque = 0, OpaqueShadow, Count };  // CRYCHIC.h:44-54 (the layers the deferred path draws)

class CRYCHIC {
public:
    CRYCHIC(int deviceOrdinal, UINT width, UINT height) : mClientWidth(width), mClientHeight(height)
    {
        md3dDevice = std::make_unique<ID3D12Device>(deviceOrdinal);
        mCommandList = std::make_unique<ID3D12GraphicsCommandList>();
    }
    CRYCHIC(const CRYCHIC& rhs) = delete;
    CRYCHIC& operator=(const CRYCHIC& rhs) = delete;
    ~CRYCHIC()
    {
        if (mCommandList) { try { mCommandList->Flush(); } catch (...) {} }
        if (mComm) crychic_comm_destroy(mComm);
        for (auto& fr : mFrameResources) if (fr && fr->FenceEvent) (void)hipEventDestroy(fr->FenceEvent);
    }

    bool Initialize()  // CRYCHIC.cpp:38-86
    {
        mCamera.SetPosition(0.0f, 2.0f, -15.0f);                                                   // :46
        mShadowMap = std::make_unique<ShadowMap>(md3dDevice.get(), mShadowMapSize, mShadowMapSize); // :48-49
        mSsao = std::make_unique<Ssao>(md3dDevice.get(), mCommandList.get(), mClientWidth, mClientHeight);  // :51-54
        mDeferred = std::make_unique<DeferredShading>(md3dDevice.get(), mClientWidth, mClientHeight, DXGI_FORMAT_R32G32B32A32_FLOAT);  // :56-58
        BuildShapeGeometry();                                                                       // :65
        BuildMaterials();                                                                           // :67
        BuildCascadeShadowRenderItems();                                                            // :70
        BuildCascadeShadowRenderItemsWithShadow();                                                  // :71
        BuildFrameResources();                                                                      // :72
        OnResize();
        mCommandList->Flush();                                                                      // :83
        return true;
    }

    void OnResize()  // CRYCHIC.cpp:110-128 (+ D3DApp::OnResize's back/depth buffer re-creation)
    {
        const size_t n = (size_t)mClientWidth * mClientHeight;
        mDepthStencilBuffer = std::make_unique<ID3D12Resource>(n * 4, ID3D12Resource::DEFAULT_HEAP);
        CrychicHipThrowIfFailed(hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(mDepthStencilBuffer->Data()), 0x00FFFFFF, n));
        mBackBuffer = std::make_unique<ID3D12Resource>(n * 4, ID3D12Resource::DEFAULT_HEAP);
        mCamera.SetLens(0.25f * 3.1415926535f, AspectRatio(), 1.0f, 100.0f);                       // :114
        if (mSsao) { mSsao->OnResize(mClientWidth, mClientHeight); mSsao->RebuildDescriptors(mDepthStencilBuffer.get()); }
        if (mDeferred) { mDeferred->OnResize(mClientWidth, mClientHeight); mDeferred->BuildDescriptors(); }
    }
    void Resize(UINT w, UINT h) { mClientWidth = w; mClientHeight = h; OnResize(); }

    void Update(const GameTimer& gt)  // CRYCHIC.cpp:130-170
    {
        mCurrFrameResourceIndex = (mCurrFrameResourceIndex + 1) % gNumFrameResources;
        mCurrFrameResource = mFrameResources[mCurrFrameResourceIndex].get();
        // :138-146: wait until the GPU has finished the frame that last used this resource
        if (mCurrFrameResource->Fence != 0) CrychicHipThrowIfFailed(hipEventSynchronize(mCurrFrameResource->FenceEvent));
        mLightRotationAngle += 0.0f * gt.DeltaTime();                                               // :152
        const float c = std::cos(mLightRotationAngle), s = std::sin(mLightRotationAngle);
        for (int i = 0; i < 3; ++i) {                                                               // :154-160 (XMMatrixRotationY)
            const DirectX::XMFLOAT3 d = mBaseLightDirections[i];
            mRotatedLightDirections[i] = { d.x * c + d.z * s, d.y, -d.x * s + d.z * c };
        }
        UpdateInstanceData(gt);                                                                     // :164
        UpdateMaterialBuffer(gt);                                                                   // :165
        UpdateCascadeShadowTransform(gt);
        UpdateMainPassCB(gt);
        UpdateShadowPassCB(gt);
        UpdateSsaoCB(gt);
    }

    void Draw(const GameTimer&)  // CRYCHIC.cpp:172-306, deferred branch
    {
        if (mRunProducerPasses) {
            DrawSceneToShadowMap();                                                                 // :208
            if (mFuseCameraPasses) DrawNormalsDepthAndGBuffer();                                    // :214 + :236 on one rasterisation
            else { DrawNormalsAndDepth(); DrawGBuffer(); }
        }
        crychic_frame_desc f = {};
        f.W = mClientWidth; f.H = mClientHeight;
        f.blurCount = mBlurCount;                                                                   // :221
        f.numDirLights = mNumDirLights;
        f.pcfSearchRadius = crychic_pcf_search_radius(mShadowMap->Width(), mPcfLiteral ? 1 : 0);
        f.flags = (mSkyEnabled ? CRYCHIC_LIGHT_SKY : 0u) | (mCubeMapLevels > 1 ? CRYCHIC_LIGHT_CUBE_LEVELS(mCubeMapLevels) : 0u);   // :278-279, :1148-1151
        f.row0 = mStripRow0; f.rows = mWholeFrame ? mClientHeight : mStripRows;                      // whole frame unless SetStrip / JoinNode
        f.normal_dev = mSsao->NormalMap()->Data();
        f.depth_dev = static_cast<const uint32_t*>(mDepthStencilBuffer->Data());
        f.randvec_dev = static_cast<const uint8_t*>(mSsao->RandomVectorMap()->Data());
        f.g0_dev = static_cast<const float*>(mDeferred->Resource(0)->Data());
        f.g1_dev = static_cast<const float*>(mDeferred->Resource(1)->Data());
        f.g2_dev = static_cast<const float*>(mDeferred->Resource(2)->Data());
        for (int i = 0; i < 4; ++i) f.shadow_dev[i] = static_cast<const uint32_t*>(mShadowMap->Resource(i)->Data());
        f.shadowDim = mShadowMap->Width();
        if (!mCubeMap) throw CrychicException(CRYCHIC_E_INVALID_ARG, "CRYCHIC::Draw (no cube map set)", __FILE__, __LINE__);
        f.cube_dev = static_cast<const uint8_t*>(mCubeMap->Data());
        f.cubeDim = mCubeMapSize;
        f.ambient0_dev = static_cast<uint16_t*>(mSsao->AmbientMap()->Data());
        f.ambient1_dev = static_cast<uint16_t*>(mSsao->AmbientMap1()->Data());
        f.edge_dev = mSsao->EdgePlane()->Data();
        f.out_rgba8_dev = static_cast<uint8_t*>(mBackBuffer->Data());
        const SsaoConstants& scb = mCurrFrameResource->SsaoCB->Element(0);
        const PassConstants& pcb = mCurrFrameResource->PassCB->Element(0);
        // Several GPUs, one frame: the peers' strips arrive in place behind this strip's lighting pass (RCCL over xGMI), so the
        // back buffer that Present sees is complete on every GPU.  The reference has one GPU (NodeMask 0, CRYCHIC.cpp:96,105).
        // With SetExchangeParts(n > 1) the lighting pass runs in n row ranges and each range travels while the next is lit.
        if (mComm)
            CrychicThrowIfFailed(crychic_draw_hot_path_shared(mComm, reinterpret_cast<const crychic_ssao_constants*>(&scb),
                                                              reinterpret_cast<const crychic_pass_constants*>(&pcb), &f,
                                                              mStripBounds.empty() ? nullptr : mStripBounds.data(), mExchangeParts,
                                                              mCommandList->Stream()));
        else
            CrychicThrowIfFailed(crychic_draw_hot_path(md3dDevice->Ctx(), reinterpret_cast<const crychic_ssao_constants*>(&scb),
                                                       reinterpret_cast<const crychic_pass_constants*>(&pcb), &f, mCommandList->Stream()));
        // :300-305: advance the fence and signal it behind this frame's commands
        mCurrFrameResource->Fence = ++mCurrentFence;
        CrychicHipThrowIfFailed(hipEventRecord(mCurrFrameResource->FenceEvent, mCommandList->Stream()));
    }

    // ---- one frame on several GPUs (no counterpart in the single-GPU reference) ----------------------------------
    // Rows [row0, row0 + rows) are this GPU's share of the hot path (row0 even).  An empty share is refused: a rank without rows
    // would have nothing to contribute to the gather (and "0 rows" must never come to mean "the whole frame").
    void SetStrip(UINT row0, UINT rows)
    {
        if (rows == 0 || row0 >= mClientHeight || rows > mClientHeight - row0)
            throw CrychicException(CRYCHIC_E_INVALID_ARG, "CRYCHIC::SetStrip (empty strip or rows outside the frame)", __FILE__, __LINE__);
        mStripRow0 = row0; mStripRows = rows; mWholeFrame = false;
    }
    void SetWholeFrame() { mStripRow0 = 0; mStripRows = 0; mWholeFrame = true; }
    // One process per GPU: `id` is the rendezvous id rank 0 obtained from crychic_comm_unique_id and handed to its peers.
    // Takes the crychic_strip_rows plan unless `bounds` (nranks x {row0, rows}) gives another tiling.
    void JoinNode(int nranks, int rank, const uint8_t id[CRYCHIC_COMM_ID_BYTES], const std::vector<uint32_t>& bounds = {})
    {
        LeaveNode();
        CrychicThrowIfFailed(crychic_comm_create(md3dDevice->Ctx(), nranks, rank, id, &mComm));
        mStripBounds = bounds;
        uint32_t r0 = 0, rn = 0;
        if (bounds.empty()) CrychicThrowIfFailed(crychic_strip_rows(mClientHeight, nranks, rank, &r0, &rn));
        else { r0 = bounds.at(2 * (size_t)rank); rn = bounds.at(2 * (size_t)rank + 1); }
        SetStrip(r0, rn);
    }
    // 1 (default): the strip, then one exchange.  n > 1: crychic_draw_hot_path_shared's overlapped exchange in n parts.
    void SetExchangeParts(UINT n)
    {
        if (n == 0 || n > CRYCHIC_MAX_EXCHANGE_PARTS) throw CrychicException(CRYCHIC_E_INVALID_ARG, "CRYCHIC::SetExchangeParts", __FILE__, __LINE__);
        mExchangeParts = n;
    }
    void LeaveNode()
    {
        if (mComm) { mCommandList->Flush(); crychic_comm_destroy(mComm); mComm = nullptr; }
        mStripBounds.clear();
        SetWholeFrame();
    }

    // mSwapChain->Present (CRYCHIC.cpp:294-297) for a headless build: read the back buffer back and write it as PPM.
    void Present(const std::string& path)
    {
        std::vector<uint8_t> host((size_t)mClientWidth * mClientHeight * 4);
        mBackBuffer->Download(host.data(), host.size(), mCommandList->Stream());
        mCommandList->Flush();
        CrychicThrowIfFailed(crychic_save_ppm(path.c_str(), host.data(), mClientWidth, mClientHeight));
    }

    // ---- planes the producer passes would fill (row f1) + the result ----
    ID3D12Resource* DepthStencilBuffer() { return mDepthStencilBuffer.get(); }
    ID3D12Resource* CurrentBackBuffer() { return mBackBuffer.get(); }
    // `levels` > 1: the resource holds the mip chain in crychic_load_dds_cube_rgba8_mips' layout (the reference binds every level,
    // CRYCHIC.cpp:1148-1151) and Draw announces it with CRYCHIC_LIGHT_CUBE_LEVELS
    void SetCubeMap(std::unique_ptr<ID3D12Resource> cube, UINT dim, UINT levels = 1) { mCubeMap = std::move(cube); mCubeMapSize = dim; mCubeMapLevels = levels ? levels : 1; }

    // CRYCHIC::LoadTextures (CRYCHIC.cpp:939-973): the six material textures in heap order (= gTextureMaps indices, :954-959) with the
    // mip chains their files store, and the sky cube map, from `dir` (the reference opens "Textures/...").  The DDS decoding that
    // CreateDDSTextureFromFile12 + the texture units do is done on the host (crychic_load_dds_*); a missing cube file keeps the
    // cube map set through SetCubeMap (snowcube1024.dds is not part of the reference checkout).
    void LoadTextures(const std::string& dir, bool requireCubeMap = false)
    {
        static const char* const files[6] = { "bricks2.dds", "bricks2_nmap.dds", "tile.dds", "tile_nmap.dds", "white1x1.dds", "default_nmap.dds" };
        hipStream_t s = mCommandList->Stream();
        std::vector<crychic_texture> tex;
        std::vector<std::unique_ptr<ID3D12Resource>> planes;
        std::vector<uint8_t> host;
        for (const char* name : files) {
            const std::string path = dir + "/" + name;
            uint32_t w = 0, h = 0, levels = 0;
            CrychicThrowIfFailed(crychic_load_dds_rgba8_mips(path.c_str(), nullptr, 0, &w, &h, &levels));
            size_t bytes = 0;
            for (uint32_t k = 0, lw = w, lh = h; k < levels; ++k) { bytes += (size_t)lw * lh * 4; lw = lw > 1 ? lw >> 1 : 1; lh = lh > 1 ? lh >> 1 : 1; }
            host.resize(bytes);
            CrychicThrowIfFailed(crychic_load_dds_rgba8_mips(path.c_str(), host.data(), host.size(), &w, &h, &levels));
            planes.push_back(std::make_unique<ID3D12Resource>(bytes, ID3D12Resource::DEFAULT_HEAP));
            planes.back()->Upload(host.data(), bytes, s);
            mCommandList->Flush();                                      // `host` is reused for the next file
            tex.push_back(crychic_texture{ static_cast<const uint8_t*>(planes.back()->Data()), w, h, levels });
        }
        const std::string cubePath = dir + "/snowcube1024.dds";
        uint32_t dim = 0, cubeLevels = 0;
        const int rc = crychic_load_dds_cube_rgba8_mips(cubePath.c_str(), nullptr, 0, &dim, &cubeLevels);      // :1148-1151: the whole chain
        if (rc == 0) {
            size_t bytes = 0;
            for (uint32_t k = 0; k < cubeLevels; ++k) { const size_t d = (dim >> k) ? (dim >> k) : 1; bytes += 6 * d * d * 4; }
            host.resize(bytes);
            CrychicThrowIfFailed(crychic_load_dds_cube_rgba8_mips(cubePath.c_str(), host.data(), host.size(), &dim, &cubeLevels));
            auto cube = std::make_unique<ID3D12Resource>(host.size(), ID3D12Resource::DEFAULT_HEAP);
            cube->Upload(host.data(), host.size(), s);
            mCommandList->Flush();
            SetCubeMap(std::move(cube), dim, cubeLevels);
        } else if (requireCubeMap) {
            CrychicThrowIfFailed(rc);
        }
        mTextures = std::move(tex);
        mTexturePlanes = std::move(planes);
    }
    ID3D12GraphicsCommandList* CommandList() { return mCommandList.get(); }
    ID3D12Device* Device() { return md3dDevice.get(); }
    float AspectRatio() const { return (float)mClientWidth / (float)mClientHeight; }

    bool mRunProducerPasses = true;   // false: the caller fills the input planes itself (tests, external producers)
    std::unique_ptr<ShadowMap> mShadowMap;
    std::unique_ptr<Ssao> mSsao;
    std::unique_ptr<DeferredShading> mDeferred;
    Camera mCamera;
    PassConstants mMainPassCB;  // index 0 of pass cbuffer (CRYCHIC.h:149)
    int mBlurCount = 3;         // CRYCHIC.cpp:221
    int mNumDirLights = 1;      // NUM_DIR_LIGHTS of the deferred shader (Common.hlsl:6-8)
    bool mPcfLiteral = true;    // Common.hlsl:305 evaluated as written
    bool mSkyEnabled = true;
    bool mFrustumCullingEnabled = true;   // CRYCHIC.h:188
    bool mFuseCameraPasses = true;        // false: DrawNormalsAndDepth and DrawGBuffer rasterise separately, as the reference records them
    UINT mShadowMapSize = 4096; // CRYCHIC.cpp:48-49
    DirectX::XMFLOAT4X4 mLightViews[MaxLights], mLightProjs[MaxLights], mShadowTransforms[MaxLights];  // CRYCHIC.h:166-170
    FrameResource* mCurrFrameResource = nullptr;

private:
    // ---- scene construction ---------------------------------------------------------------------------------------
    void BuildShapeGeometry()  // CRYCHIC.cpp:1250-1445: box + grid concatenated into one vertex / index buffer
    {
        uint32_t nbI = 0, ngI = 0;
        const int nbV = crychic_create_box(1.0f, 1.0f, 1.0f, 3, nullptr, 0, nullptr, 0, &nbI);         // :1253
        const int ngV = crychic_create_grid(20.0f, 30.0f, 60, 40, nullptr, 0, nullptr, 0, &ngI);       // :1254
        std::vector<crychic_vertex> v((size_t)nbV + ngV);
        std::vector<uint32_t> idx((size_t)nbI + ngI);
        CrychicThrowIfFailed(crychic_create_box(1.0f, 1.0f, 1.0f, 3, v.data(), nbV, idx.data(), nbI, &nbI));
        CrychicThrowIfFailed(crychic_create_grid(20.0f, 30.0f, 60, 40, v.data() + nbV, ngV, idx.data() + nbI, ngI, &ngI));
        auto geo = std::make_unique<MeshGeometry>();
        geo->Name = "shapeGeo";
        geo->VertexCount = (UINT)v.size(); geo->IndexCount = (UINT)idx.size();
        geo->VertexBufferGPU = std::make_unique<ID3D12Resource>(v.size() * sizeof(crychic_vertex), ID3D12Resource::DEFAULT_HEAP);
        geo->IndexBufferGPU = std::make_unique<ID3D12Resource>(idx.size() * 4, ID3D12Resource::DEFAULT_HEAP);
        geo->VertexBufferGPU->Upload(v.data(), v.size() * sizeof(crychic_vertex), mCommandList->Stream());
        geo->IndexBufferGPU->Upload(idx.data(), idx.size() * 4, mCommandList->Stream());
        mCommandList->Flush();
        auto bounds = [&](size_t first, size_t count) {                                                 // :1318-1337: min / max of the positions
            float lo[3] = { 3.4e38f, 3.4e38f, 3.4e38f }, hi[3] = { -3.4e38f, -3.4e38f, -3.4e38f };
            for (size_t k = first; k < first + count; ++k)
                for (int c = 0; c < 3; ++c) { lo[c] = std::min(lo[c], v[k].Pos[c]); hi[c] = std::max(hi[c], v[k].Pos[c]); }
            BoundingBox b;
            b.Center = { 0.5f * (lo[0] + hi[0]), 0.5f * (lo[1] + hi[1]), 0.5f * (lo[2] + hi[2]) };
            b.Extents = { 0.5f * (hi[0] - lo[0]), 0.5f * (hi[1] - lo[1]), 0.5f * (hi[2] - lo[2]) };
            return b;
        };
        geo->DrawArgs["box"] = SubmeshGeometry{ nbI, 0, 0, bounds(0, (size_t)nbV) };                   // :1271-1301
        geo->DrawArgs["grid"] = SubmeshGeometry{ ngI, nbI, nbV, bounds((size_t)nbV, (size_t)ngV) };
        mGeometries[geo->Name] = std::move(geo);
    }
    void BuildMaterials()  // CRYCHIC.cpp:1768-1821
    {
        auto add = [&](const char* name, int cb, int d, int n, DirectX::XMFLOAT4 alb, DirectX::XMFLOAT3 r0, float rough) {
            auto m = std::make_unique<Material>();
            m->Name = name; m->MatCBIndex = cb; m->DiffuseSrvHeapIndex = d; m->NormalSrvHeapIndex = n;
            m->DiffuseAlbedo = alb; m->FresnelR0 = r0; m->Roughness = rough;
            mMaterials[name] = std::move(m);
        };
        add("bricks0", 0, 0, 1, { 1.0f, 1.0f, 1.0f, 1.0f }, { 0.1f, 0.1f, 0.1f }, 0.3f);
        add("tile0", 1, 2, 3, { 0.9f, 0.9f, 0.9f, 1.0f }, { 0.2f, 0.2f, 0.2f }, 0.7f);
        add("mirror0", 2, 4, 5, { 0.0f, 0.0f, 0.0f, 1.0f }, { 0.98f, 0.97f, 0.95f }, 0.1f);
        add("skullMat", 3, 4, 5, { 1.0f, 1.0f, 1.0f, 1.0f }, { 0.6f, 0.6f, 0.6f }, 0.8f);
        add("sky", 4, 6, 7, { 1.0f, 1.0f, 1.0f, 1.0f }, { 0.1f, 0.1f, 0.1f }, 1.0f);
    }
    static DirectX::XMFLOAT4X4 ScaleTranslate(float s, float tx, float ty, float tz)   // XMMatrixScaling * XMMatrixTranslation
    {
        return DirectX::XMFLOAT4X4{ { { s, 0, 0, 0 }, { 0, s, 0, 0 }, { 0, 0, s, 0 }, { tx, ty, tz, 1 } } };
    }
    void AddBoxAndGrid(RenderLayer layer, int boxMaterialModulo, UINT gridMaterial)
    {
        MeshGeometry* geo = mGeometries["shapeGeo"].get();
        auto box = std::make_unique<RenderItem>();
        box->itemIndex = mItemIndex++;
        box->Geo = geo;
        box->IndexCount = geo->DrawArgs["box"].IndexCount;
        box->StartIndexLocation = geo->DrawArgs["box"].StartIndexLocation;
        box->BaseVertexLocation = geo->DrawArgs["box"].BaseVertexLocation;
        box->Bounds = geo->DrawArgs["box"].Bounds;                                                   // :1882
        box->Instances.resize(100);
        box->InstanceCount = 100;
        for (int i = 0; i < 10; ++i)
            for (int j = 0; j < 10; ++j) {                                                            // :2338-2346, 2398-2406
                box->Instances[i * 10 + j].World = ScaleTranslate(1.6f, (-5 + i) * 5.0f, 0.8f, (-5 + j) * 5.0f);
                box->Instances[i * 10 + j].MaterialIndex = (UINT)(i % boxMaterialModulo);
            }
        mInstanceCounts.push_back(100);
        mRitemLayer[(int)layer].push_back(box.get());
        mAllRitems.push_back(std::move(box));
        auto grid = std::make_unique<RenderItem>();
        grid->itemIndex = mItemIndex++;
        grid->Geo = geo;
        grid->IndexCount = geo->DrawArgs["grid"].IndexCount;
        grid->StartIndexLocation = geo->DrawArgs["grid"].StartIndexLocation;
        grid->BaseVertexLocation = geo->DrawArgs["grid"].BaseVertexLocation;
        grid->Bounds = geo->DrawArgs["grid"].Bounds;                                                 // :1930
        grid->Instances.resize(1);
        grid->InstanceCount = 1;
        grid->Instances[0].World = ScaleTranslate(3.0f, 0.0f, 0.0f, 0.0f);                            // :2371, 2431
        grid->Instances[0].MaterialIndex = gridMaterial;
        mInstanceCounts.push_back(1);
        mRitemLayer[(int)layer].push_back(grid.get());
        mAllRitems.push_back(std::move(grid));
    }
    void BuildCascadeShadowRenderItems() { AddBoxAndGrid(RenderLayer::Opaque, 2, 3); }               // CRYCHIC.cpp:2322-2375 (sky / debug quad are not drawn by the deferred path's producers)
    void BuildCascadeShadowRenderItemsWithShadow() { AddBoxAndGrid(RenderLayer::OpaqueShadow, 3, 1); }  // :2380-2435

    void BuildFrameResources()  // CRYCHIC.cpp:1759-1766: 1 main + 12 shadow pass slots
    {
        for (int i = 0; i < gNumFrameResources; ++i) {
            mFrameResources.push_back(std::make_unique<FrameResource>(md3dDevice.get(), 1 + 12, mInstanceCounts, (UINT)mAllRitems.size(),
                                                                      (UINT)mMaterials.size()));
            CrychicHipThrowIfFailed(hipEventCreateWithFlags(&mFrameResources.back()->FenceEvent, hipEventDisableTiming));
        }
    }
    static DirectX::XMFLOAT4X4 Transposed(const DirectX::XMFLOAT4X4& a)
    {
        DirectX::XMFLOAT4X4 t;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) t.m[i][j] = a.m[j][i];
        return t;
    }
    void UpdateInstanceData(const GameTimer&)  // CRYCHIC.cpp:515-564: only instances that pass the frustum test reach the instance buffer
    {
        std::vector<uint8_t> visible;
        for (auto& ri : mAllRitems) {
            auto* buf = mCurrFrameResource->InstanceBuffers[ri->itemIndex].get();
            visible.assign(ri->Instances.size(), 1);
            if (mFrustumCullingEnabled && !ri->Instances.empty()) {                                   // :543-544
                std::vector<float> worlds(ri->Instances.size() * 16);
                for (size_t k = 0; k < ri->Instances.size(); ++k) std::memcpy(&worlds[16 * k], ri->Instances[k].World.m, 64);
                const float c[3] = { ri->Bounds.Center.x, ri->Bounds.Center.y, ri->Bounds.Center.z };
                const float e[3] = { ri->Bounds.Extents.x, ri->Bounds.Extents.y, ri->Bounds.Extents.z };
                CrychicThrowIfFailed(std::min(0, crychic_frustum_cull(&mCamera.Raw(), c, e, worlds.data(), (uint32_t)ri->Instances.size(), visible.data())));
            }
            int n = 0;
            for (size_t k = 0; k < ri->Instances.size(); ++k) {
                if (!visible[k]) continue;
                const InstanceData& src = ri->Instances[k];
                InstanceData data;
                data.World = Transposed(src.World);                                                   // :546
                data.TexTransform = Transposed(src.TexTransform);                                     // :547
                data.MaterialIndex = src.MaterialIndex;
                buf->CopyData(n++, data);
            }
            ri->InstanceCount = (UINT)n;
        }
    }
    void UpdateMaterialBuffer(const GameTimer&)  // CRYCHIC.cpp:566-592
    {
        auto* buf = mCurrFrameResource->MaterialBuffer.get();
        for (auto& e : mMaterials) {
            Material* mat = e.second.get();
            if (mat->NumFramesDirty > 0) {
                MaterialData d;
                d.DiffuseAlbedo = mat->DiffuseAlbedo; d.FresnelR0 = mat->FresnelR0; d.Roughness = mat->Roughness;
                d.MatTransform = Transposed(mat->MatTransform);
                d.DiffuseMapIndex = (UINT)mat->DiffuseSrvHeapIndex; d.NormalMapIndex = (UINT)mat->NormalSrvHeapIndex;
                buf->CopyData(mat->MatCBIndex, d);                                                    // Metalness keeps its default 0.5 (Q5)
                mat->NumFramesDirty--;
            }
        }
    }
    void UpdateShadowPassCB(const GameTimer&)  // CRYCHIC.cpp:870-901: slots 1..12, of which cascades 0..3 are meaningful
    {
        for (int i = 0; i < 4; ++i) {
            PassConstants cb;
            float vp[4][4];
            for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) {
                float acc = 0.0f;
                for (int k = 0; k < 4; ++k) acc += mLightViews[i].m[r][k] * mLightProjs[i].m[k][c];
                vp[r][c] = acc;
            }
            for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { cb.ViewProj.m[c][r] = vp[r][c]; cb.View.m[c][r] = mLightViews[i].m[r][c]; cb.Proj.m[c][r] = mLightProjs[i].m[r][c]; }
            cb.RenderTargetSize = { (float)mShadowMap->Width(), (float)mShadowMap->Height() };
            cb.InvRenderTargetSize = { 1.0f / mShadowMap->Width(), 1.0f / mShadowMap->Height() };
            mCurrFrameResource->PassCB->CopyData(1 + i, cb);                                          // :897-898
        }
    }

    // ---- producer passes --------------------------------------------------------------------------------------------
    std::vector<crychic_draw_item> DrawItems(const std::vector<RenderItem*>& ritems)  // CRYCHIC::DrawRenderItems, CRYCHIC.cpp:2438-2475
    {
        std::vector<crychic_draw_item> out;
        mSceneTriangles = 0;
        for (RenderItem* ri : ritems) {
            crychic_draw_item d = {};
            d.vertices_dev = static_cast<const crychic_vertex*>(ri->Geo->VertexBufferGPU->Data());
            d.vertexCount = ri->Geo->VertexCount;
            d.indices_dev = static_cast<const uint32_t*>(ri->Geo->IndexBufferGPU->Data());
            d.indexCount = ri->IndexCount; d.startIndexLocation = ri->StartIndexLocation; d.baseVertexLocation = ri->BaseVertexLocation;
            d.instances_dev = reinterpret_cast<const crychic_instance_data*>(mCurrFrameResource->InstanceBuffers[ri->itemIndex]->Resource()->Data());
            d.instanceCount = ri->InstanceCount;
            out.push_back(d);
            mSceneTriangles += (uint64_t)(ri->IndexCount / 3) * ri->InstanceCount;
        }
        return out;
    }
    void* RasterWorkspace(uint64_t triangles, UINT W, UINT H, size_t* bytes)
    {
        *bytes = crychic_raster_workspace_bytes(triangles, W, H);
        if (!mRasterWorkspace || mRasterWorkspace->Bytes() < *bytes) mRasterWorkspace = std::make_unique<ID3D12Resource>(*bytes, ID3D12Resource::DEFAULT_HEAP);
        return mRasterWorkspace->Data();
    }
    void DrawSceneToShadowMap()  // CRYCHIC.cpp:2477-2510 (the reference loops 6 times; cascades 0..3 carry data, Q8)
    {
        auto items = DrawItems(mRitemLayer[(int)RenderLayer::OpaqueShadow]);
        size_t bytes;
        void* ws = RasterWorkspace(4 * mSceneTriangles, mShadowMap->Width(), mShadowMap->Height(), &bytes);
        // the four cascades' pass constants sit in consecutive 256-byte-aligned slots of the upload buffer; gather them
        crychic_pass_constants cbs[4];
        uint32_t* targets[4];
        for (int i = 0; i < 4; ++i) {
            std::memcpy(&cbs[i], &mCurrFrameResource->PassCB->Element(1 + i), sizeof cbs[i]);
            targets[i] = static_cast<uint32_t*>(mShadowMap->Resource(i)->Data());
        }
        CrychicThrowIfFailed(crychic_draw_scene_to_shadow_maps(md3dDevice->Ctx(), cbs, 4, items.data(), (uint32_t)items.size(), targets,
                                                               mShadowMap->Width(), 10000, 2.0f, ws, bytes, mCommandList->Stream()));  // bias: :1601-1603
    }
    void DrawNormalsAndDepth()  // CRYCHIC.cpp:2512-2543
    {
        auto items = DrawItems(mRitemLayer[(int)RenderLayer::Opaque]);
        size_t bytes;
        void* ws = RasterWorkspace(mSceneTriangles, mClientWidth, mClientHeight, &bytes);
        const PassConstants& cb = mCurrFrameResource->PassCB->Element(0);
        CrychicThrowIfFailed(crychic_draw_normals_and_depth(md3dDevice->Ctx(), reinterpret_cast<const crychic_pass_constants*>(&cb), items.data(),
                                                            (uint32_t)items.size(), mSsao->NormalMap()->Data(),
                                                            static_cast<uint32_t*>(mDepthStencilBuffer->Data()), mClientWidth, mClientHeight, ws, bytes,
                                                            mCommandList->Stream()));
    }
    void DrawGBuffer()  // CRYCHIC.cpp:2545-2571
    {
        auto items = DrawItems(mRitemLayer[(int)RenderLayer::Opaque]);
        size_t bytes;
        void* ws = RasterWorkspace(mSceneTriangles, mClientWidth, mClientHeight, &bytes);
        const PassConstants& cb = mCurrFrameResource->PassCB->Element(0);
        // the scissor rectangle of this pass (CRYCHIC.cpp:2547-2548) is this GPU's strip when the frame is shared (SetStrip / JoinNode)
        CrychicThrowIfFailed(crychic_draw_gbuffer_rows(
            md3dDevice->Ctx(), reinterpret_cast<const crychic_pass_constants*>(&cb), items.data(), (uint32_t)items.size(),
            reinterpret_cast<const crychic_material_data*>(mCurrFrameResource->MaterialBuffer->Resource()->Data()), (uint32_t)mMaterials.size(),
            mTextures.empty() ? nullptr : mTextures.data(), (uint32_t)mTextures.size(), static_cast<float*>(mDeferred->Resource(0)->Data()),
            static_cast<float*>(mDeferred->Resource(1)->Data()), static_cast<float*>(mDeferred->Resource(2)->Data()),
            static_cast<uint32_t*>(mDepthStencilBuffer->Data()), mClientWidth, mClientHeight, mStripRow0,
            mWholeFrame ? mClientHeight : mStripRows, ws, bytes, mCommandList->Stream()));
    }
    void DrawNormalsDepthAndGBuffer()  // DrawNormalsAndDepth + DrawGBuffer: same items, same ViewProj, same visibility -> one rasterisation
    {
        auto items = DrawItems(mRitemLayer[(int)RenderLayer::Opaque]);
        size_t bytes;
        void* ws = RasterWorkspace(mSceneTriangles, mClientWidth, mClientHeight, &bytes);
        const PassConstants& cb = mCurrFrameResource->PassCB->Element(0);
        CrychicThrowIfFailed(crychic_draw_normals_depth_and_gbuffer_rows(
            md3dDevice->Ctx(), reinterpret_cast<const crychic_pass_constants*>(&cb), items.data(), (uint32_t)items.size(),
            reinterpret_cast<const crychic_material_data*>(mCurrFrameResource->MaterialBuffer->Resource()->Data()), (uint32_t)mMaterials.size(),
            mTextures.empty() ? nullptr : mTextures.data(), (uint32_t)mTextures.size(), mSsao->NormalMap()->Data(),
            static_cast<float*>(mDeferred->Resource(0)->Data()), static_cast<float*>(mDeferred->Resource(1)->Data()),
            static_cast<float*>(mDeferred->Resource(2)->Data()), static_cast<uint32_t*>(mDepthStencilBuffer->Data()), mClientWidth, mClientHeight,
            mStripRow0, mWholeFrame ? mClientHeight : mStripRows, ws, bytes, mCommandList->Stream()));
    }
    void UpdateCascadeShadowTransform(const GameTimer&)  // CRYCHIC.cpp:634-815
    {
        const float ld[3] = { mBaseLightDirections[0].x, mBaseLightDirections[0].y, mBaseLightDirections[0].z };  // :726
        float lv[4][16], lp[4][16], st[4][16];
        CrychicThrowIfFailed(crychic_update_cascade_shadow_transform(&mCamera.Raw(), ld, mShadowMap->Width(), lv, lp, st));
        for (int i = 0; i < 4; ++i) {
            std::memcpy(&mLightViews[i], lv[i], 64); std::memcpy(&mLightProjs[i], lp[i], 64); std::memcpy(&mShadowTransforms[i], st[i], 64);
        }
    }
    void UpdateMainPassCB(const GameTimer& gt)  // CRYCHIC.cpp:817-868
    {
        float st[4][16], dirs[3][3];
        for (int i = 0; i < 4; ++i) std::memcpy(st[i], &mShadowTransforms[i], 64);
        for (int i = 0; i < 3; ++i) { dirs[i][0] = mRotatedLightDirections[i].x; dirs[i][1] = mRotatedLightDirections[i].y; dirs[i][2] = mRotatedLightDirections[i].z; }
        CrychicThrowIfFailed(crychic_update_main_pass_cb(&mCamera.Raw(), mClientWidth, mClientHeight, st, dirs,
                                                         reinterpret_cast<crychic_pass_constants*>(&mMainPassCB)));
        mMainPassCB.TotalTime = gt.TotalTime();
        mMainPassCB.DeltaTime = gt.DeltaTime();
        mCurrFrameResource->PassCB->CopyData(0, mMainPassCB);                                       // :866-867
    }
    void UpdateSsaoCB(const GameTimer&)  // CRYCHIC.cpp:903-937
    {
        SsaoConstants ssaoCB;
        DirectX::XMFLOAT4 off[14];
        mSsao->GetOffsetVectors(off);                                                               // :920
        float o[14][4];
        for (int i = 0; i < 14; ++i) { o[i][0] = off[i].x; o[i][1] = off[i].y; o[i][2] = off[i].z; o[i][3] = off[i].w; }
        CrychicThrowIfFailed(crychic_update_ssao_cb(&mCamera.Raw(), mClientWidth, mClientHeight, o, reinterpret_cast<crychic_ssao_constants*>(&ssaoCB)));
        mCurrFrameResource->SsaoCB->CopyData(0, ssaoCB);                                            // :935-936
    }

    std::unordered_map<std::string, std::unique_ptr<MeshGeometry>> mGeometries;      // CRYCHIC.h:123-125
    std::unordered_map<std::string, std::unique_ptr<Material>> mMaterials;
    std::vector<crychic_texture> mTextures;                                           // gTextureMaps (row f4 loads the DDS files; empty = white / flat)
    std::vector<std::unique_ptr<ID3D12Resource>> mTexturePlanes;                      // the device copies mTextures points into (LoadTextures)
    std::vector<std::unique_ptr<RenderItem>> mAllRitems;                              // CRYCHIC.h:132-135
    std::vector<RenderItem*> mRitemLayer[(int)RenderLayer::Count];
    std::vector<int> mInstanceCounts;                                                 // CRYCHIC.h:183
    UINT mItemIndex = 0;
    uint64_t mSceneTriangles = 0;
    std::unique_ptr<ID3D12Resource> mRasterWorkspace;
    std::unique_ptr<ID3D12Device> md3dDevice;
    std::unique_ptr<ID3D12GraphicsCommandList> mCommandList;
    std::vector<std::unique_ptr<FrameResource>> mFrameResources;
    int mCurrFrameResourceIndex = 0;
    UINT64 mCurrentFence = 0;
    crychic_comm* mComm = nullptr;            // set by JoinNode: this GPU renders a strip and gathers the others'
    std::vector<uint32_t> mStripBounds;       // nranks x {row0, rows}; empty = crychic_strip_rows
    UINT mStripRow0 = 0, mStripRows = 0;      // this GPU's rows when the frame is shared
    UINT mExchangeParts = 1;                  // SetExchangeParts
    bool mWholeFrame = true;
    std::unique_ptr<ID3D12Resource> mDepthStencilBuffer, mBackBuffer, mCubeMap;
    UINT mCubeMapSize = 0, mCubeMapLevels = 1;
    UINT mClientWidth, mClientHeight;
    float mLightRotationAngle = 0.0f;
    DirectX::XMFLOAT3 mBaseLightDirections[3] = { { 0.57735f, -0.57735f, 0.57735f }, { -0.57735f, -0.57735f, 0.57735f }, { 0.0f, -0.707f, -0.707f } };  // CRYCHIC.h:173-177
    DirectX::XMFLOAT3 mRotatedLightDirections[3];
};
