// FrameResource.h -- FrameResource.h:7-96 with byte-identical constant / structured-buffer layouts.
#pragma once
#include <cstddef>
#include <memory>
#include <vector>
#include "UploadBuffer.h"

#define MaxLights 16  // Common/d3dUtil.h:226

namespace crychic_detail {
inline DirectX::XMFLOAT4X4 Identity4x4() { return DirectX::XMFLOAT4X4{ { { 1, 0, 0, 0 }, { 0, 1, 0, 0 }, { 0, 0, 1, 0 }, { 0, 0, 0, 1 } } }; }
}

// The records below are the data ABI of the path: what the reference's upload buffers hold byte for byte, hence what the
// kernels read.  Field names follow the reference so its call sites compile; the numbers are byte offsets (checked by the
// static_asserts underneath).  Defaults are set by the constructors.

struct Light {                              // Common/d3dUtil.h:216-224, 48 B
    DirectX::XMFLOAT3 Strength;             // +0
    float FalloffStart;                     // +12   point / spot only
    DirectX::XMFLOAT3 Direction;            // +16   directional / spot only
    float FalloffEnd;                       // +28   point / spot only
    DirectX::XMFLOAT3 Position;             // +32   point / spot only
    float SpotPower;                        // +44   spot only
    Light() : Strength{ 0.5f, 0.5f, 0.5f }, FalloffStart(1.0f), Direction{ 0.0f, -1.0f, 0.0f }, FalloffEnd(10.0f), Position{ 0.0f, 0.0f, 0.0f }, SpotPower(64.0f) {}
};

struct InstanceData {                       // FrameResource.h:7-15, 144 B, one per instance in a structured buffer
    DirectX::XMFLOAT4X4 World;              // +0    stored transposed by UpdateInstanceData (CRYCHIC.cpp:546)
    DirectX::XMFLOAT4X4 TexTransform;       // +64
    UINT MaterialIndex;                     // +128
    UINT ObjPad0, ObjPad1, ObjPad2;         // +132
    InstanceData() : World(crychic_detail::Identity4x4()), TexTransform(crychic_detail::Identity4x4()), MaterialIndex(0), ObjPad0(0), ObjPad1(0), ObjPad2(0) {}
};

struct MaterialData {                       // FrameResource.h:17-27, 112 B
    DirectX::XMFLOAT4 DiffuseAlbedo;        // +0
    DirectX::XMFLOAT3 FresnelR0;            // +16
    float Roughness;                        // +28
    DirectX::XMFLOAT4X4 MatTransform;       // +32
    UINT DiffuseMapIndex;                   // +96
    UINT NormalMapIndex;                    // +100
    float Metalness;                        // +104  never assigned by UpdateMaterialBuffer: stays 0.5 (quirk Q5)
    UINT MaterialPad0;                      // +108
    MaterialData() : DiffuseAlbedo{ 1.0f, 1.0f, 1.0f, 1.0f }, FresnelR0{ 0.01f, 0.01f, 0.01f }, Roughness(0.5f), MatTransform(crychic_detail::Identity4x4()),
                     DiffuseMapIndex(0), NormalMapIndex(0), Metalness(0.5f), MaterialPad0(0) {}
};

struct PassConstants {                      // cbPass, FrameResource.h:29-51, 2048 B (= crychic_pass_constants)
    DirectX::XMFLOAT4X4 View;               // +0     every matrix: transpose of the row-vector matrix (CRYCHIC.cpp:843-849)
    DirectX::XMFLOAT4X4 InvView;            // +64
    DirectX::XMFLOAT4X4 Proj;               // +128
    DirectX::XMFLOAT4X4 InvProj;            // +192
    DirectX::XMFLOAT4X4 ViewProj;           // +256
    DirectX::XMFLOAT4X4 InvViewProj;        // +320
    DirectX::XMFLOAT4X4 ViewProjTex;        // +384   lighting pass: world -> ambient-map uv
    DirectX::XMFLOAT4X4 ShadowTransforms[12];  // +448   [0..3] = the cascades; [4..11] never valid in the reference either
    DirectX::XMFLOAT3 EyePosW;              // +1216
    float cbPerObjectPad1;                  // +1228
    DirectX::XMFLOAT2 RenderTargetSize;     // +1232
    DirectX::XMFLOAT2 InvRenderTargetSize;  // +1240
    float NearZ, FarZ;                      // +1248, +1252
    float TotalTime, DeltaTime;             // +1256, +1260
    DirectX::XMFLOAT4 AmbientLight;         // +1264
    Light Lights[MaxLights];                // +1280  [0, NUM_DIR_LIGHTS) directional
    PassConstants()
        : View(crychic_detail::Identity4x4()), InvView(View), Proj(View), InvProj(View), ViewProj(View), InvViewProj(View), ViewProjTex(View),
          ShadowTransforms{}, EyePosW{ 0.0f, 0.0f, 0.0f }, cbPerObjectPad1(0.0f), RenderTargetSize{ 0.0f, 0.0f }, InvRenderTargetSize{ 0.0f, 0.0f },
          NearZ(0.0f), FarZ(0.0f), TotalTime(0.0f), DeltaTime(0.0f), AmbientLight{ 0.0f, 0.0f, 0.0f, 1.0f } {}
};

struct SsaoConstants {                      // cbSsao, FrameResource.h:53-67, 496 B (= crychic_ssao_constants)
    DirectX::XMFLOAT4X4 Proj;               // +0
    DirectX::XMFLOAT4X4 InvProj;            // +64
    DirectX::XMFLOAT4X4 ProjTex;            // +128   view -> depth-map uv
    DirectX::XMFLOAT4 OffsetVectors[14];    // +192
    DirectX::XMFLOAT4 BlurWeights[3];       // +416   11 weights, 12th float unused
    DirectX::XMFLOAT2 RenderTargetSize;     // +464   left at (0, 0) by the reference
    DirectX::XMFLOAT2 InvRenderTargetSize;  // +472   1 / half-res size
    float OcclusionRadius;                  // +480
    float OcclusionFadeStart;               // +484
    float OcclusionFadeEnd;                 // +488
    float SurfaceEpsilon;                   // +492
    SsaoConstants() : Proj{}, InvProj{}, ProjTex{}, OffsetVectors{}, BlurWeights{}, RenderTargetSize{ 0.0f, 0.0f }, InvRenderTargetSize{ 0.0f, 0.0f },
                      OcclusionRadius(0.5f), OcclusionFadeStart(0.2f), OcclusionFadeEnd(2.0f), SurfaceEpsilon(0.05f) {}
};

struct Vertex {                             // FrameResource.h:69-75, 44 B (= crychic_vertex)
    DirectX::XMFLOAT3 Pos;                  // +0
    DirectX::XMFLOAT3 Normal;               // +12
    DirectX::XMFLOAT2 TexC;                 // +24
    DirectX::XMFLOAT3 TangentU;             // +32
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
