// UploadBuffer<T> -- Common/UploadBuffer.h:5-63 over pinned, persistently mapped host memory.  Same constructor,
// same Resource()/CopyData(), same 256-byte rounding of constant-buffer elements (:22, d3dUtil::CalcConstantBufferByteSize).
#pragma once
#include <cstring>
#include <memory>
#include "d3d_shim.h"

template <typename T>
class UploadBuffer {
public:
    UploadBuffer(ID3D12Device* /*device*/, UINT elementCount, bool isConstantBuffer) : mIsConstantBuffer(isConstantBuffer)
    {
        mElementByteSize = sizeof(T);
        if (isConstantBuffer) mElementByteSize = (sizeof(T) + 255) & ~255u;
        mUploadBuffer = std::make_unique<ID3D12Resource>((size_t)mElementByteSize * elementCount, ID3D12Resource::UPLOAD_HEAP);
        mMappedData = static_cast<BYTE*>(mUploadBuffer->Data());  // persistently mapped, like Map(0, nullptr, ...) at :32
    }
    UploadBuffer(const UploadBuffer&) = delete;
    UploadBuffer& operator=(const UploadBuffer&) = delete;

    ID3D12Resource* Resource() const { return mUploadBuffer.get(); }
    void CopyData(int elementIndex, const T& data) { std::memcpy(&mMappedData[(size_t)elementIndex * mElementByteSize], &data, sizeof(T)); }
    // Read side used by the HIP backend: element i as the pass objects consume it.
    const T& Element(int elementIndex) const { return *reinterpret_cast<const T*>(&mMappedData[(size_t)elementIndex * mElementByteSize]); }
    UINT ElementByteSize() const { return mElementByteSize; }

private:
    std::unique_ptr<ID3D12Resource> mUploadBuffer;
    BYTE* mMappedData = nullptr;
    UINT mElementByteSize = 0;
    bool mIsConstantBuffer = false;
};
