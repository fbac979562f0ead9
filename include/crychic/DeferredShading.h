// DeferredShading.h -- DeferredShading.h:4-45 / DeferredShading.cpp: owns the G-buffer planes
// (R32G32B32A32_FLOAT, CRYCHIC.cpp:56-58).  Index 3 (GBuffer3, constant 0: GBuffer.hlsl:29) is kept addressable but is
// never read by the lighting kernel; the reference additionally allocates each plane as a 4-slice array (.cpp:120).
#pragma once
#include <memory>
#include "d3d_shim.h"

class DeferredShading {
public:
    DeferredShading(ID3D12Device* device, UINT width, UINT height, DXGI_FORMAT format)
    {
        md3dDevice = device;
        mFormat = format;
        if (format != DXGI_FORMAT_R32G32B32A32_FLOAT) throw CrychicException(CRYCHIC_E_UNSUPPORTED, "DeferredShading (R32G32B32A32_FLOAT only)", __FILE__, __LINE__);
        mWidth = width; mHeight = height;
        mViewport = { 0.0f, 0.0f, (float)width, (float)height, 0.0f, 1.0f };
        mScissorRect = { 0, 0, (int)width, (int)height };
        BuildResource();
    }
    DeferredShading(const DeferredShading& rhs) = delete;
    DeferredShading& operator=(const DeferredShading& rhs) = delete;
    virtual ~DeferredShading() = default;

    UINT Width() const { return mWidth; }
    UINT Height() const { return mHeight; }
    DXGI_FORMAT Format() const { return mFormat; }
    ID3D12Resource* Resource(int index) { return mGBuffer[index].get(); }
    CD3DX12_GPU_DESCRIPTOR_HANDLE Srv(int) const { return {}; }
    CD3DX12_CPU_DESCRIPTOR_HANDLE Rtv(int) const { return {}; }
    D3D12_VIEWPORT Viewport() const { return mViewport; }
    D3D12_RECT ScissorRect() const { return mScissorRect; }
    void BuildDescriptors(CD3DX12_CPU_DESCRIPTOR_HANDLE, CD3DX12_GPU_DESCRIPTOR_HANDLE, CD3DX12_CPU_DESCRIPTOR_HANDLE) {}
    void BuildDescriptors() {}
    void OnResize(UINT newWidth, UINT newHeight)  // DeferredShading.cpp:79-93
    {
        if (mWidth != newWidth || mHeight != newHeight) {
            mWidth = newWidth; mHeight = newHeight;
            mViewport = { 0.0f, 0.0f, (float)newWidth, (float)newHeight, 0.0f, 1.0f };
            mScissorRect = { 0, 0, (int)newWidth, (int)newHeight };
            BuildResource();
        }
    }

private:
    void BuildResource()  // DeferredShading.cpp:108-142
    {
        const size_t bytes = (size_t)mWidth * mHeight * 16;
        for (int i = 0; i < 3; ++i) {
            mGBuffer[i] = std::make_unique<ID3D12Resource>(bytes, ID3D12Resource::DEFAULT_HEAP);
            CrychicHipThrowIfFailed(hipMemset(mGBuffer[i]->Data(), 0, bytes));  // cleared to black, CRYCHIC.cpp:2554
        }
        mGBuffer[3] = nullptr;
    }
    ID3D12Device* md3dDevice = nullptr;
    D3D12_VIEWPORT mViewport;
    D3D12_RECT mScissorRect;
    UINT mWidth = 0, mHeight = 0;
    DXGI_FORMAT mFormat = DXGI_FORMAT_R32G32B32A32_FLOAT;
    std::unique_ptr<ID3D12Resource> mGBuffer[4];
};
