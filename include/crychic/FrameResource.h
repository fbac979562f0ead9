// FrameResource.h -- FrameResource.h:7-96 with byte-identical constant / structured-buffer layouts.
#pragma once
#include <cstddef>
#include <memory>
#include <vector>
#include "UploadBuffer.h"

#define MaxLights 16  // Common/d3dUtil.h:226

struct Light {  // Common/d3dUtil.h:216-224
    DirectX::XMFLOAT3 Strength = { 0.5f, 0.5f, 0.5f };
    float FalloffStart = 1.0f;
    DirectX::XMFLOAT3 Direction = { 0.0f, -1.0f, 0.0f };
    float FalloffEnd = 10.0f;
    DirectX::XMFLOAT3 Position = { 0.0f, 0.0f, 0.0f };
    float SpotPower = 64.0f;
};

namespace crychic_detail {
inline DirectX::XMFLOAT4X4 Identity4x4() { return DirectX::XMFLOAT4X4{ { { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, 1, 0 }, { 0, 0, 0, 1 } } }; }
}

struct InstanceData {  // FrameResource.h:7-15
    DirectX::XMFLOAT4X4 World = crychic_detail::Identity4x4();
    DirectX::XMFLOAT4X4 TexTransform = crychic_detail::Identity4x4();
    UINT MaterialIndex = 0;
    UINT ObjPad0 = 0, ObjPad1 = 0, ObjPad2 = 0;
};

struct MaterialData {  // FrameResource.h:17-27
    DirectX::XMFLOAT4 DiffuseAlbedo = { 1.0f, 1.0f, 1.0f, 1.0f };
    DirectX::XMFLOAT3 FresnelR0 = { 0.01f, 0.01f, 0.01f };
    float Roughness = 0.5f;
    DirectX::XMFLOAT4X4 MatTransform = crychic_detail::Identity4x4();
    UINT DiffuseMapIndex = 0;
    UINT NormalMapIndex = 0;
    float Metalness = 0.5f;
    UINT MaterialPad0 = 0;
};

struct PassConstants {  // FrameResource.h:29-51
    DirectX::XMFLOAT4X4 View = crychic_detail::Identity4x4();
    DirectX::XMFLOAT4X4 InvView = crychic_detail::Identity4x4();
    DirectX::XMFLOAT4X4 Proj = crychic_detail::Identity4x4();
    DirectX::XMFLOAT4X4 InvProj = crychic_detail::Identity4x4();
    DirectX::XMFLOAT4X4 ViewProj = crychic_detail::Identity4x4();
    DirectX::XMFLOAT4X4 InvViewProj = crychic_detail::Identity4x4();
    DirectX::XMFLOAT4X4 ViewProjTex = crychic_detail::Identity4x4();
    DirectX::XMFLOAT4X4 ShadowTransforms[12];
    DirectX::XMFLOAT3 EyePosW = { 0.0f, 0.0f, 0.0f };
    float cbPerObjectPad1 = 0.0f;
    DirectX::XMFLOAT2 RenderTargetSize = { 0.0f, 0.0f };
    DirectX::XMFLOAT2 InvRenderTargetSize = { 0.0f, 0.0f };
    float NearZ = 0.0f;
    float FarZ = 0.0f;
    float TotalTime = 0.0f;
    float DeltaTime = 0.0f;
    DirectX::XMFLOAT4 AmbientLight = { 0.0f, 0.0f, 0.0f, 1.0f };
    Light Lights[MaxLights];
};

struct SsaoConstants {  // FrameResource.h:53-67
    DirectX::XMFLOAT4X4 Proj;
    DirectX::XMFLOAT4X4 InvProj;
    DirectX::XMFLOAT4X4 ProjTex;
    DirectX::XMFLOAT4 OffsetVectors[14];
    DirectX::XMFLOAT4 BlurWeights[3];
    DirectX::XMFLOAT2 RenderTargetSize = { 0.0f, 0.0f };
    DirectX::XMFLOAT2 InvRenderTargetSize = { 0.0f, 0.0f };
    float OcclusionRadius = 0.5f;
    float OcclusionFadeStart = 0.2f;
    float OcclusionFadeEnd = 2.0f;
    float SurfaceEpsilon = 0.05f;
};

struct Vertex {  // FrameResource.h:69-75
    DirectX::XMFLOAT3 Pos;
    DirectX::XMFLOAT3 Normal;
    DirectX::XMFLOAT2 TexC;
    DirectX::XMFLOAT3 TangentU;
};

static_assert(sizeof(Light) == 48 && sizeof(Light) == sizeof(crychic_light), "Light ABI");
static_assert(sizeof(PassConstants) == 2048 && sizeof(PassConstants) == sizeof(crychic_pass_constants), "PassConstants ABI");
static_assert(sizeof(SsaoConstants) == 496 && sizeof(SsaoConstants) == sizeof(crychic_ssao_constants), "SsaoConstants ABI");
static_assert(offsetof(PassConstants, ShadowTransforms) == 448 && offsetof(PassConstants, EyePosW) == 1216 &&
              offsetof(PassConstants, AmbientLight) == 1264 && offsetof(PassConstants, Lights) == 1280, "cbPass offsets");
static_assert(offsetof(SsaoConstants, OffsetVectors) == 192 && offsetof(SsaoConstants, BlurWeights) == 416 &&
              offsetof(SsaoConstants, OcclusionRadius) == 480, "cbSsao offsets");
static_assert(sizeof(MaterialData) == 112 && sizeof(InstanceData) == 144 && sizeof(Vertex) == 44, "structured-buffer ABI");

struct FrameResource {  // FrameResource.h:77-96, FrameResource.cpp:3-20
public:
    FrameResource(ID3D12Device* device, UINT passCount, std::vector<int>& InstanceCounts, UINT itemCount, UINT materialCount)
    {
        PassCB = std::make_unique<UploadBuffer<PassConstants>>(device, passCount, true);
        SsaoCB = std::make_unique<UploadBuffer<SsaoConstants>>(device, 1, true);
        MaterialBuffer = std::make_unique<UploadBuffer<MaterialData>>(device, materialCount, false);
        for (UINT i = 0; i < itemCount; ++i)
            InstanceBuffers.push_back(std::make_unique<UploadBuffer<InstanceData>>(device, (UINT)InstanceCounts[i], false));
    }
    FrameResource(const FrameResource&) = delete;
    FrameResource& operator=(const FrameResource&) = delete;

    std::unique_ptr<UploadBuffer<PassConstants>> PassCB = nullptr;
    std::unique_ptr<UploadBuffer<MaterialData>> MaterialBuffer = nullptr;
    std::unique_ptr<UploadBuffer<SsaoConstants>> SsaoCB = nullptr;
    std::vector<std::unique_ptr<UploadBuffer<InstanceData>>> InstanceBuffers;
    // Fence value marking commands up to this point (CRYCHIC.cpp:300-305); here: a HIP event recorded after Draw.
    UINT64 Fence = 0;
    hipEvent_t FenceEvent = nullptr;
};
