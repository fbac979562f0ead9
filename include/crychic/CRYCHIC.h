// CRYCHIC.h -- headless counterpart of the reference application class (CRYCHIC.h:56-190, CRYCHIC.cpp) for the
// deferred hot path: same member names (mSsao, mDeferred, mShadowMap, mMainPassCB, mFrameResources ...), same
// Initialize / OnResize / Update / Draw call sequence, D3D12 replaced by libcrychic_hip.so.
//
// What Draw() covers: CRYCHIC.cpp:220-221 (ComputeSsao) and :238-279 (deferred lighting + sky) on the GPU.  The
// producer passes that fill the shadow maps, the normal/depth target and the G-buffer through the D3D rasteriser
// (DrawSceneToShadowMap / DrawNormalsAndDepth / DrawGBuffer, CRYCHIC.cpp:208,214,236) are SURVEY.md row f1; until they
// exist the caller writes those planes directly (Resource accessors below).
#pragma once
#include <cmath>
#include <memory>
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

class Camera {  // Common/Camera.h:20-97 (the subset the constant builders consume)
public:
    void SetPosition(float x, float y, float z) { mCam.pos[0] = x; mCam.pos[1] = y; mCam.pos[2] = z; }
    void SetLens(float fovY, float aspect, float zn, float zf) { mCam.fovY = fovY; mCam.aspect = aspect; mCam.nearZ = zn; mCam.farZ = zf; }
    void LookTo(float lx, float ly, float lz, float ux, float uy, float uz) { mCam.look[0] = lx; mCam.look[1] = ly; mCam.look[2] = lz; mCam.up[0] = ux; mCam.up[1] = uy; mCam.up[2] = uz; }
    float GetNearZ() const { return mCam.nearZ; }
    float GetFarZ() const { return mCam.farZ; }
    float GetFovY() const { return mCam.fovY; }
    float GetAspect() const { return mCam.aspect; }
    const crychic_camera& Raw() const { return mCam; }
private:
    crychic_camera mCam = { { 0, 0, 0 }, { 0, 0, 1 }, { 0, 1, 0 }, 0.7853981634f, 1.0f, 1.0f, 1000.0f };
};

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
        for (auto& fr : mFrameResources) if (fr && fr->FenceEvent) (void)hipEventDestroy(fr->FenceEvent);
    }

    bool Initialize()  // CRYCHIC.cpp:38-86
    {
        mCamera.SetPosition(0.0f, 2.0f, -15.0f);                                                   // :46
        mShadowMap = std::make_unique<ShadowMap>(md3dDevice.get(), mShadowMapSize, mShadowMapSize); // :48-49
        mSsao = std::make_unique<Ssao>(md3dDevice.get(), mCommandList.get(), mClientWidth, mClientHeight);  // :51-54
        mDeferred = std::make_unique<DeferredShading>(md3dDevice.get(), mClientWidth, mClientHeight, DXGI_FORMAT_R32G32B32A32_FLOAT);  // :56-58
        BuildFrameResources();
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
        UpdateCascadeShadowTransform(gt);
        UpdateMainPassCB(gt);
        UpdateSsaoCB(gt);
    }

    void Draw(const GameTimer&)  // CRYCHIC.cpp:172-306, deferred branch
    {
        crychic_frame_desc f = {};
        f.W = mClientWidth; f.H = mClientHeight;
        f.blurCount = mBlurCount;                                                                   // :221
        f.numDirLights = mNumDirLights;
        f.pcfSearchRadius = crychic_pcf_search_radius(mShadowMap->Width(), mPcfLiteral ? 1 : 0);
        f.flags = mSkyEnabled ? CRYCHIC_LIGHT_SKY : 0u;                                             // :278-279
        f.row0 = 0; f.rows = mClientHeight;
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
        CrychicThrowIfFailed(crychic_draw_hot_path(md3dDevice->Ctx(), reinterpret_cast<const crychic_ssao_constants*>(&scb),
                                                   reinterpret_cast<const crychic_pass_constants*>(&pcb), &f, mCommandList->Stream()));
        // :300-305: advance the fence and signal it behind this frame's commands
        mCurrFrameResource->Fence = ++mCurrentFence;
        CrychicHipThrowIfFailed(hipEventRecord(mCurrFrameResource->FenceEvent, mCommandList->Stream()));
    }

    // ---- planes the producer passes would fill (row f1) + the result ----
    ID3D12Resource* DepthStencilBuffer() { return mDepthStencilBuffer.get(); }
    ID3D12Resource* CurrentBackBuffer() { return mBackBuffer.get(); }
    void SetCubeMap(std::unique_ptr<ID3D12Resource> cube, UINT dim) { mCubeMap = std::move(cube); mCubeMapSize = dim; }
    ID3D12GraphicsCommandList* CommandList() { return mCommandList.get(); }
    ID3D12Device* Device() { return md3dDevice.get(); }
    float AspectRatio() const { return (float)mClientWidth / (float)mClientHeight; }

    std::unique_ptr<ShadowMap> mShadowMap;
    std::unique_ptr<Ssao> mSsao;
    std::unique_ptr<DeferredShading> mDeferred;
    Camera mCamera;
    PassConstants mMainPassCB;  // index 0 of pass cbuffer (CRYCHIC.h:149)
    int mBlurCount = 3;         // CRYCHIC.cpp:221
    int mNumDirLights = 1;      // NUM_DIR_LIGHTS of the deferred shader (Common.hlsl:6-8)
    bool mPcfLiteral = true;    // Common.hlsl:305 evaluated as written
    bool mSkyEnabled = true;
    UINT mShadowMapSize = 4096; // CRYCHIC.cpp:48-49
    DirectX::XMFLOAT4X4 mLightViews[MaxLights], mLightProjs[MaxLights], mShadowTransforms[MaxLights];  // CRYCHIC.h:166-170
    FrameResource* mCurrFrameResource = nullptr;

private:
    void BuildFrameResources()  // CRYCHIC.cpp:1759-1766: 1 main + 12 shadow pass slots
    {
        std::vector<int> instanceCounts;
        for (int i = 0; i < gNumFrameResources; ++i) {
            mFrameResources.push_back(std::make_unique<FrameResource>(md3dDevice.get(), 1 + 12, instanceCounts, 0, 5));
            CrychicHipThrowIfFailed(hipEventCreateWithFlags(&mFrameResources.back()->FenceEvent, hipEventDisableTiming));
        }
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

    std::unique_ptr<ID3D12Device> md3dDevice;
    std::unique_ptr<ID3D12GraphicsCommandList> mCommandList;
    std::vector<std::unique_ptr<FrameResource>> mFrameResources;
    int mCurrFrameResourceIndex = 0;
    UINT64 mCurrentFence = 0;
    std::unique_ptr<ID3D12Resource> mDepthStencilBuffer, mBackBuffer, mCubeMap;
    UINT mCubeMapSize = 0;
    UINT mClientWidth, mClientHeight;
    float mLightRotationAngle = 0.0f;
    DirectX::XMFLOAT3 mBaseLightDirections[3] = { { 0.57735f, -0.57735f, 0.57735f }, { -0.57735f, -0.57735f, 0.57735f }, { 0.0f, -0.707f, -0.707f } };  // CRYCHIC.h:173-177
    DirectX::XMFLOAT3 mRotatedLightDirections[3];
};
