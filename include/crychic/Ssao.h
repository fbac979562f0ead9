// Ssao.h -- the reference's Ssao pass object (Ssao.h:10-125, Ssao.cpp) over libcrychic_hip.so.  Same public names,
// arity and ownership: the object owns the view-normal map, two half-res ambient maps (ping-pong) and the
// random-vector map; ComputeSsao issues SSAO + blurCount x (H, V) blur on the command list's stream.
#pragma once
#include <memory>
#include <vector>
#include "FrameResource.h"

class Ssao {
public:
    Ssao(ID3D12Device* device, ID3D12GraphicsCommandList* cmdList, UINT width, UINT height)  // Ssao.cpp:8-20
    {
        md3dDevice = device;
        OnResize(width, height);
        BuildOffsetVectors();
        BuildRandomVectorTexture(cmdList);
    }
    Ssao(const Ssao& rhs) = delete;
    Ssao& operator=(const Ssao& rhs) = delete;
    ~Ssao() = default;

    static const DXGI_FORMAT AmbientMapFormat = DXGI_FORMAT_R16_UNORM;          // Ssao.h:21
    static const DXGI_FORMAT NormalMapFormat = DXGI_FORMAT_R16G16B16A16_FLOAT;  // Ssao.h:22
    static const int MaxBlurRadius = 5;                                         // Ssao.h:24

    UINT SsaoMapWidth() const { return mRenderTargetWidth / 2; }    // Ssao.cpp:22-25
    UINT SsaoMapHeight() const { return mRenderTargetHeight / 2; }  // Ssao.cpp:27-30

    void GetOffsetVectors(DirectX::XMFLOAT4 offsets[14])            // Ssao.cpp:32-35
    {
        for (int i = 0; i < 14; ++i) offsets[i] = mOffsets[i];
    }
    std::vector<float> CalcGaussWeights(float sigma)                // Ssao.cpp:37-68
    {
        float w[2 * MaxBlurRadius + 1];
        int n = crychic_calc_gauss_weights(sigma, w, 2 * MaxBlurRadius + 1);
        if (n < 0) throw CrychicException(n, "crychic_calc_gauss_weights (blurRadius <= MaxBlurRadius)", __FILE__, __LINE__);
        return std::vector<float>(w, w + n);
    }

    ID3D12Resource* NormalMap() { return mNormalMap.get(); }        // Ssao.cpp:70-73
    ID3D12Resource* AmbientMap() { return mAmbientMap0.get(); }     // Ssao.cpp:75-78

    // Descriptor plumbing has no HIP meaning: the handles are inert, only the depth buffer binding is kept.
    CD3DX12_CPU_DESCRIPTOR_HANDLE NormalMapRtv() const { return {}; }
    CD3DX12_GPU_DESCRIPTOR_HANDLE NormalMapSrv() const { return {}; }
    CD3DX12_GPU_DESCRIPTOR_HANDLE AmbientMapSrv() const { return {}; }
    void BuildDescriptors(ID3D12Resource* depthStencilBuffer, CD3DX12_CPU_DESCRIPTOR_HANDLE, CD3DX12_GPU_DESCRIPTOR_HANDLE,
                          CD3DX12_CPU_DESCRIPTOR_HANDLE, UINT, UINT) { RebuildDescriptors(depthStencilBuffer); }
    void RebuildDescriptors(ID3D12Resource* depthStencilBuffer) { mDepthStencilBuffer = depthStencilBuffer; }  // Ssao.cpp:125-162
    void SetPSOs(ID3D12PipelineState*, ID3D12PipelineState*) {}     // Ssao.cpp:119-123

    void OnResize(UINT newWidth, UINT newHeight)                    // Ssao.cpp:164-183
    {
        if (mRenderTargetWidth != newWidth || mRenderTargetHeight != newHeight) {
            if ((newWidth & 1u) || (newHeight & 1u)) throw CrychicException(CRYCHIC_E_INVALID_ARG, "Ssao::OnResize (even size required)", __FILE__, __LINE__);
            mRenderTargetWidth = newWidth;
            mRenderTargetHeight = newHeight;
            mViewport = { 0.0f, 0.0f, mRenderTargetWidth / 2.0f, mRenderTargetHeight / 2.0f, 0.0f, 1.0f };
            mScissorRect = { 0, 0, (int)mRenderTargetWidth / 2, (int)mRenderTargetHeight / 2 };
            BuildResources();
        }
    }

    // Ssao.cpp:185-229.  The depth buffer is the one bound by (Re)BuildDescriptors.
    void ComputeSsao(ID3D12GraphicsCommandList* cmdList, FrameResource* currFrame, int blurCount)
    {
        const SsaoConstants& cb = currFrame->SsaoCB->Element(0);
        if (mLiteralSequence) {
            // the reference's own recording order: the SSAO draw (Ssao.cpp:198-222), then BlurAmbientMap (:228), sweep by sweep
            CrychicThrowIfFailed(crychic_ssao(
                md3dDevice->Ctx(), reinterpret_cast<const crychic_ssao_constants*>(&cb), mNormalMap->Data(),
                static_cast<const uint32_t*>(mDepthStencilBuffer->Data()), static_cast<const uint8_t*>(mRandomVectorMap->Data()),
                static_cast<uint16_t*>(mAmbientMap0->Data()), mEdgePlane->Data(), mRenderTargetWidth, mRenderTargetHeight, 0,
                mRenderTargetHeight / 2, cmdList->Stream()));
            BlurAmbientMap(cmdList, currFrame, blurCount);
            return;
        }
        CrychicThrowIfFailed(crychic_ssao_compute(
            md3dDevice->Ctx(), reinterpret_cast<const crychic_ssao_constants*>(&cb), mNormalMap->Data(),
            static_cast<const uint32_t*>(mDepthStencilBuffer->Data()), static_cast<const uint8_t*>(mRandomVectorMap->Data()),
            static_cast<uint16_t*>(mAmbientMap0->Data()), static_cast<uint16_t*>(mAmbientMap1->Data()), mEdgePlane->Data(),
            mRenderTargetWidth, mRenderTargetHeight, blurCount, 0, mRenderTargetHeight / 2, cmdList->Stream()));
    }

    // false (default): ComputeSsao is one crychic_ssao_compute call (depth pass + SSAO pass + one blur launch per iteration: blur_pair_kernel records
    // iteration 0, blur_replay_kernel replays iterations 1 .. blurCount - 1; exits as in DESIGN.md 4);
    // true: the reference's literal sequence of one SSAO pass and 2 * blurCount single sweeps.  Same bits either way
    // (tests/cpp/veneer_driver.cpp renders both).
    bool mLiteralSequence = false;

    ID3D12Resource* AmbientMap1() { return mAmbientMap1.get(); }
    ID3D12Resource* RandomVectorMap() { return mRandomVectorMap.get(); }
    ID3D12Resource* EdgePlane() { return mEdgePlane.get(); }

private:
    // Ssao.cpp:231-243 (Ssao.h:76)
    void BlurAmbientMap(ID3D12GraphicsCommandList* cmdList, FrameResource* currFrame, int blurCount)
    {
        mBoundFrame = currFrame;            // the reference binds currFrame's SsaoCB on the command list here (Ssao.cpp:234-236)
        for (int i = 0; i < blurCount; ++i) {
            BlurAmbientMap(cmdList, true);
            BlurAmbientMap(cmdList, false);
        }
    }
    // Ssao.cpp:245-293 (Ssao.h:77): the constants come from the constant buffer bound by the caller
    void BlurAmbientMap(ID3D12GraphicsCommandList* cmdList, bool horzBlur)
    {
        const SsaoConstants& cb = mBoundFrame->SsaoCB->Element(0);
        ID3D12Resource* in = horzBlur ? mAmbientMap0.get() : mAmbientMap1.get();
        ID3D12Resource* out = horzBlur ? mAmbientMap1.get() : mAmbientMap0.get();
        CrychicThrowIfFailed(crychic_ssao_blur(md3dDevice->Ctx(), reinterpret_cast<const crychic_ssao_constants*>(&cb), mEdgePlane->Data(),
                                               static_cast<const uint16_t*>(in->Data()), static_cast<uint16_t*>(out->Data()),
                                               mRenderTargetWidth, mRenderTargetHeight, horzBlur ? 1 : 0, 0, mRenderTargetHeight / 2,
                                               cmdList->Stream()));
    }

    void BuildResources()  // Ssao.cpp:295-350
    {
        const size_t n = (size_t)mRenderTargetWidth * mRenderTargetHeight, n2 = n / 4;
        mNormalMap = std::make_unique<ID3D12Resource>(n * 8, ID3D12Resource::DEFAULT_HEAP);
        mAmbientMap0 = std::make_unique<ID3D12Resource>(n2 * 2, ID3D12Resource::DEFAULT_HEAP);
        mAmbientMap1 = std::make_unique<ID3D12Resource>(n2 * 2, ID3D12Resource::DEFAULT_HEAP);
        mEdgePlane = std::make_unique<ID3D12Resource>(crychic_edge_plane_bytes(mRenderTargetWidth, mRenderTargetHeight), ID3D12Resource::DEFAULT_HEAP);
        CrychicHipThrowIfFailed(hipMemset(mAmbientMap0->Data(), 0xFF, n2 * 2));  // clear value 1.0 (Ssao.cpp:333)
        CrychicHipThrowIfFailed(hipMemset(mAmbientMap1->Data(), 0xFF, n2 * 2));
        // not required (crychic_hip.h "Contract" of the edge workspace): hygiene for tools that look at the workspace
        CrychicHipThrowIfFailed(hipMemset(mEdgePlane->Data(), 0, crychic_edge_plane_bytes(mRenderTargetWidth, mRenderTargetHeight)));
    }
    void BuildRandomVectorTexture(ID3D12GraphicsCommandList* cmdList)  // Ssao.cpp:352-421
    {
        mRandomVectorMap = std::make_unique<ID3D12Resource>(256 * 256 * 4, ID3D12Resource::DEFAULT_HEAP);
        std::vector<uint8_t> init(256 * 256 * 4);
        crychic_build_random_vector_texture(&RandState(), 1, init.data());
        mRandomVectorMap->Upload(init.data(), init.size(), cmdList->Stream());
        cmdList->Flush();  // the staging vector dies here; the reference keeps an upload heap alive instead
    }
    void BuildOffsetVectors()  // Ssao.cpp:423-462
    {
        float o[14][4];
        crychic_build_offset_vectors(&RandState(), o);
        for (int i = 0; i < 14; ++i) mOffsets[i] = { o[i][0], o[i][1], o[i][2], o[i][3] };
    }
    // The process-wide CRT rand() state the reference draws from (MathHelper::RandF, never seeded).
    static uint32_t& RandState() { static uint32_t s = 1; return s; }

private:
    ID3D12Device* md3dDevice = nullptr;
    ID3D12Resource* mDepthStencilBuffer = nullptr;
    FrameResource* mBoundFrame = nullptr;
    std::unique_ptr<ID3D12Resource> mRandomVectorMap, mNormalMap, mAmbientMap0, mAmbientMap1, mEdgePlane;
    UINT mRenderTargetWidth = 0, mRenderTargetHeight = 0;
    DirectX::XMFLOAT4 mOffsets[14];
    D3D12_VIEWPORT mViewport;
    D3D12_RECT mScissorRect;
};
