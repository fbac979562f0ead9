// d3d_shim.h -- the handful of D3D12 / DirectXMath names the reference's pass classes mention in their public
// signatures (Ssao.h:14-66, DeferredShading.h:7-30, ShadowMap.h:7-26, FrameResource.h:80-81, UploadBuffer.h:9-56),
// re-declared as thin HIP-backed types so reference call sites keep compiling against include/crychic/*.h.
// Nothing here emulates D3D12: a "device" is a crychic_ctx bound to one GPU, a "command list" is a HIP stream,
// a "resource" is a linear HBM (or pinned host) allocation.
#pragma once
#include <hip/hip_runtime_api.h>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include "crychic_hip.h"

typedef uint32_t UINT;
typedef uint8_t BYTE;
typedef uint64_t UINT64;

enum DXGI_FORMAT {
    DXGI_FORMAT_UNKNOWN = 0,
    DXGI_FORMAT_R32G32B32A32_FLOAT = 2,
    DXGI_FORMAT_R16G16B16A16_FLOAT = 10,
    DXGI_FORMAT_R8G8B8A8_UNORM = 28,
    DXGI_FORMAT_R24G8_TYPELESS = 44,
    DXGI_FORMAT_D24_UNORM_S8_UINT = 45,
    DXGI_FORMAT_R16_UNORM = 56,
};

namespace DirectX {
struct XMFLOAT2 { float x, y; };
struct XMFLOAT3 { float x, y, z; };
struct XMFLOAT4 { float x, y, z, w; };
struct XMFLOAT4X4 { float m[4][4]; };
}  // namespace DirectX

// The reference's DxException (Common/d3dUtil.h:132-144): thrown by the veneer when a C-ABI call fails.
class CrychicException : public std::runtime_error {
public:
    CrychicException(int status, const std::string& call, const char* file, int line)
        : std::runtime_error(call + " failed (" + std::to_string(status) + "): " + crychic_last_error() + " at " + file + ":" +
                             std::to_string(line)),
          Status(status) {}
    int Status;
};
#define CrychicThrowIfFailed(x)                                               \
    do {                                                                      \
        int st__ = (x);                                                       \
        if (st__ < 0) throw CrychicException(st__, #x, __FILE__, __LINE__);   \
    } while (0)
#define CrychicHipThrowIfFailed(x)                                                                         \
    do {                                                                                                   \
        hipError_t e__ = (x);                                                                              \
        if (e__ != hipSuccess) throw std::runtime_error(std::string(#x) + ": " + hipGetErrorString(e__));  \
    } while (0)

// ID3D12Device: owns the crychic_ctx of one GPU (Common/d3dApp.cpp:429-444 creates exactly one device).
class ID3D12Device {
public:
    explicit ID3D12Device(int ordinal = 0) { CrychicThrowIfFailed(crychic_ctx_create(ordinal, &mCtx)); }
    ~ID3D12Device() { crychic_ctx_destroy(mCtx); }
    ID3D12Device(const ID3D12Device&) = delete;
    ID3D12Device& operator=(const ID3D12Device&) = delete;
    crychic_ctx* Ctx() const { return mCtx; }
private:
    crychic_ctx* mCtx = nullptr;
};

// ID3D12GraphicsCommandList: the stream the frame's kernels are issued on (one direct queue + one list in the
// reference, Common/d3dApp.cpp:481-503).
class ID3D12GraphicsCommandList {
public:
    ID3D12GraphicsCommandList() { CrychicHipThrowIfFailed(hipStreamCreate(&mStream)); }
    ~ID3D12GraphicsCommandList() { (void)hipStreamDestroy(mStream); }
    ID3D12GraphicsCommandList(const ID3D12GraphicsCommandList&) = delete;
    ID3D12GraphicsCommandList& operator=(const ID3D12GraphicsCommandList&) = delete;
    hipStream_t Stream() const { return mStream; }
    void Flush() { CrychicHipThrowIfFailed(hipStreamSynchronize(mStream)); }  // D3DApp::FlushCommandQueue
private:
    hipStream_t mStream = nullptr;
};

// ID3D12Resource: a linear allocation.  Default heap = hipMalloc (HBM); upload heap = pinned, device-visible host memory.
class ID3D12Resource {
public:
    enum Heap { DEFAULT_HEAP, UPLOAD_HEAP };
    ID3D12Resource(size_t bytes, Heap heap) : mBytes(bytes), mHeap(heap)
    {
        if (heap == DEFAULT_HEAP) CrychicHipThrowIfFailed(hipMalloc(&mPtr, bytes));
        else CrychicHipThrowIfFailed(hipHostMalloc(&mPtr, bytes, hipHostMallocDefault));
    }
    ~ID3D12Resource() { if (mHeap == DEFAULT_HEAP) (void)hipFree(mPtr); else (void)hipHostFree(mPtr); }
    ID3D12Resource(const ID3D12Resource&) = delete;
    ID3D12Resource& operator=(const ID3D12Resource&) = delete;
    void* Data() const { return mPtr; }
    size_t Bytes() const { return mBytes; }
    UINT64 GetGPUVirtualAddress() const { return (UINT64)(uintptr_t)mPtr; }
    void Upload(const void* src, size_t bytes, hipStream_t s) { CrychicHipThrowIfFailed(hipMemcpyAsync(mPtr, src, bytes, hipMemcpyHostToDevice, s)); }
    void Download(void* dst, size_t bytes, hipStream_t s) const { CrychicHipThrowIfFailed(hipMemcpyAsync(dst, mPtr, bytes, hipMemcpyDeviceToHost, s)); }
private:
    void* mPtr = nullptr;
    size_t mBytes = 0;
    Heap mHeap;
};

// Descriptor handles have no HIP meaning; they exist so reference call sites that pass them around still compile.
struct CD3DX12_CPU_DESCRIPTOR_HANDLE { size_t ptr = 0; };
struct CD3DX12_GPU_DESCRIPTOR_HANDLE { UINT64 ptr = 0; };
struct ID3D12PipelineState {};
struct D3D12_VIEWPORT { float TopLeftX, TopLeftY, Width, Height, MinDepth, MaxDepth; };
struct D3D12_RECT { int left, top, right, bottom; };
