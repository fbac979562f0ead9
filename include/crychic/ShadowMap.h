// ShadowMap.h -- ShadowMap.h:4-47 / ShadowMap.cpp: the cascade depth maps (R24G8_TYPELESS, sampled as R24_UNORM_X8).
// The reference creates 12 textures of 12 slices each (.cpp:112); 4 cascades carry data (CRYCHIC.cpp:644), the
// other slots alias cascade storage lazily here (allocated on first Resource(i >= 4) use).
#pragma once
#include <memory>
#include "d3d_shim.h"

class ShadowMap {
public:
    ShadowMap(ID3D12Device* device, UINT width, UINT height)
    {
        md3dDevice = device;
        mWidth = width; mHeight = height;
        if (width != height) throw CrychicException(CRYCHIC_E_UNSUPPORTED, "ShadowMap (square maps only)", __FILE__, __LINE__);
        mViewport = { 0.0f, 0.0f, (float)width, (float)height, 0.0f, 1.0f };
        mScissorRect = { 0, 0, (int)width, (int)height };
        BuildResource();
    }
    ShadowMap(const ShadowMap& rhs) = delete;
    ShadowMap& operator=(const ShadowMap& rhs) = delete;
    ~ShadowMap() = default;

    UINT Width() const { return mWidth; }
    UINT Height() const { return mHeight; }
    ID3D12Resource* Resource(int index)
    {
        if (!mShadowMap[index]) Alloc(index);
        return mShadowMap[index].get();
    }
    CD3DX12_GPU_DESCRIPTOR_HANDLE Srv(int) const { return {}; }
    CD3DX12_CPU_DESCRIPTOR_HANDLE Dsv(int) const { return {}; }
    D3D12_VIEWPORT Viewport() const { return mViewport; }
    D3D12_RECT ScissorRect() const { return mScissorRect; }
    void BuildDescriptors(CD3DX12_CPU_DESCRIPTOR_HANDLE, CD3DX12_GPU_DESCRIPTOR_HANDLE, CD3DX12_CPU_DESCRIPTOR_HANDLE) {}
    void OnResize(UINT newWidth, UINT newHeight)  // ShadowMap.cpp:67-76
    {
        if (mWidth != newWidth || mHeight != newHeight) {
            mWidth = newWidth; mHeight = newHeight;
            BuildResource();
        }
    }

private:
    void Alloc(int i)
    {
        const size_t n = (size_t)mWidth * mHeight;
        mShadowMap[i] = std::make_unique<ID3D12Resource>(n * 4, ID3D12Resource::DEFAULT_HEAP);
        // optimized clear value depth 1.0 / stencil 0 (ShadowMap.cpp:119-122): D24 = 0x00FFFFFF
        CrychicHipThrowIfFailed(hipMemsetD32(reinterpret_cast<hipDeviceptr_t>(mShadowMap[i]->Data()), 0x00FFFFFF, n));
    }
    void BuildResource()
    {
        for (auto& s : mShadowMap) s = nullptr;
        for (int i = 0; i < 4; ++i) Alloc(i);
    }
    ID3D12Device* md3dDevice = nullptr;
    D3D12_VIEWPORT mViewport;
    D3D12_RECT mScissorRect;
    UINT mWidth = 0, mHeight = 0;
    DXGI_FORMAT mFormat = DXGI_FORMAT_R24G8_TYPELESS;
    std::unique_ptr<ID3D12Resource> mShadowMap[12];
};
