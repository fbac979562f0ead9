// host_constants.cpp -- host-side producers of the constant buffers the hot path consumes: the product's
// counterparts of Ssao::CalcGaussWeights / BuildOffsetVectors / BuildRandomVectorTexture (Ssao.cpp:37-68,
// 392-402, 423-462) and CRYCHIC::UpdateCascadeShadowTransform / UpdateMainPassCB / UpdateSsaoCB
// (CRYCHIC.cpp:634-937).  DirectXMath is replaced by a small row-vector Mat4 (LH, row-major) with the same
// constructor definitions; values are stored transposed exactly where the reference calls XMMatrixTranspose.
#include <algorithm>
#include <cmath>
#include <cstring>
#include "crychic_hip.h"

namespace {

struct V3 {
    float x, y, z;
};
inline V3 operator-(V3 a, V3 b) { return { a.x - b.x, a.y - b.y, a.z - b.z }; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return { a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }
inline V3 normalize(V3 a)
{
    float l = std::sqrt(dot(a, a));
    return { a.x / l, a.y / l, a.z / l };
}

struct Mat4 {
    float m[4][4];
    static Mat4 zero()
    {
        Mat4 r;
        std::memset(r.m, 0, sizeof r.m);
        return r;
    }
    Mat4 operator*(const Mat4& b) const
    {
        Mat4 r;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                float s = 0.0f;
                for (int k = 0; k < 4; ++k) s += m[i][k] * b.m[k][j];
                r.m[i][j] = s;
            }
        return r;
    }
    void store(float out[16]) const { std::memcpy(out, m, sizeof m); }
    void store_transposed(float out[16]) const
    {
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) out[4 * j + i] = m[i][j];
    }
};

// XMMatrixInverse: Gauss-Jordan with partial pivoting in double precision.
bool inverse(const Mat4& a, Mat4& out)
{
    double w[4][8];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            w[i][j] = a.m[i][j];
            w[i][4 + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < 4; ++c) {
        int piv = c;
        for (int r = c + 1; r < 4; ++r)
            if (std::fabs(w[r][c]) > std::fabs(w[piv][c])) piv = r;
        if (w[piv][c] == 0.0) return false;
        if (piv != c)
            for (int j = 0; j < 8; ++j) std::swap(w[piv][j], w[c][j]);
        const double d = w[c][c];
        for (int j = 0; j < 8; ++j) w[c][j] /= d;
        for (int r = 0; r < 4; ++r) {
            if (r == c) continue;
            const double f = w[r][c];
            if (f != 0.0)
                for (int j = 0; j < 8; ++j) w[r][j] -= f * w[c][j];
        }
    }
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) out.m[i][j] = (float)w[i][4 + j];
    return true;
}

// XMMatrixPerspectiveFovLH (Common/Camera.cpp:127)
Mat4 perspective_fov_lh(float fovY, float aspect, float zn, float zf)
{
    const float h = std::cos(0.5f * fovY) / std::sin(0.5f * fovY);
    const float range = zf / (zf - zn);
    Mat4 r = Mat4::zero();
    r.m[0][0] = h / aspect;
    r.m[1][1] = h;
    r.m[2][2] = range;
    r.m[2][3] = 1.0f;
    r.m[3][2] = -range * zn;
    return r;
}
// XMMatrixLookAtLH (CRYCHIC.cpp:734)
Mat4 look_at_lh(V3 eye, V3 at, V3 up)
{
    const V3 zaxis = normalize(at - eye);
    const V3 xaxis = normalize(cross(up, zaxis));
    const V3 yaxis = cross(zaxis, xaxis);
    Mat4 r = Mat4::zero();
    r.m[0][0] = xaxis.x; r.m[0][1] = yaxis.x; r.m[0][2] = zaxis.x;
    r.m[1][0] = xaxis.y; r.m[1][1] = yaxis.y; r.m[1][2] = zaxis.y;
    r.m[2][0] = xaxis.z; r.m[2][1] = yaxis.z; r.m[2][2] = zaxis.z;
    r.m[3][0] = -dot(xaxis, eye); r.m[3][1] = -dot(yaxis, eye); r.m[3][2] = -dot(zaxis, eye);
    r.m[3][3] = 1.0f;
    return r;
}
// XMMatrixOrthographicOffCenterLH (CRYCHIC.cpp:804)
Mat4 ortho_off_center_lh(float l, float r_, float b, float t, float zn, float zf)
{
    Mat4 r = Mat4::zero();
    r.m[0][0] = 2.0f / (r_ - l);
    r.m[1][1] = 2.0f / (t - b);
    r.m[2][2] = 1.0f / (zf - zn);
    r.m[3][0] = (l + r_) / (l - r_);
    r.m[3][1] = (t + b) / (b - t);
    r.m[3][2] = zn / (zn - zf);
    r.m[3][3] = 1.0f;
    return r;
}
// Camera::UpdateViewMatrix (Common/Camera.cpp:226-273)
Mat4 camera_view(const crychic_camera& c)
{
    const V3 pos{ c.pos[0], c.pos[1], c.pos[2] };
    const V3 L = normalize(V3{ c.look[0], c.look[1], c.look[2] });
    V3 R = normalize(cross(V3{ c.up[0], c.up[1], c.up[2] }, L));
    const V3 U = normalize(cross(L, R));
    R = cross(U, L);
    Mat4 v = Mat4::zero();
    v.m[0][0] = R.x; v.m[0][1] = U.x; v.m[0][2] = L.x;
    v.m[1][0] = R.y; v.m[1][1] = U.y; v.m[1][2] = L.y;
    v.m[2][0] = R.z; v.m[2][1] = U.z; v.m[2][2] = L.z;
    v.m[3][0] = -dot(pos, R); v.m[3][1] = -dot(pos, U); v.m[3][2] = -dot(pos, L);
    v.m[3][3] = 1.0f;
    return v;
}
// NDC [-1,1]^2 -> texture [0,1]^2 (CRYCHIC.cpp:805-809, 828-832, 910-914)
Mat4 tex_matrix()
{
    Mat4 t = Mat4::zero();
    t.m[0][0] = 0.5f; t.m[1][1] = -0.5f; t.m[2][2] = 1.0f;
    t.m[3][0] = 0.5f; t.m[3][1] = 0.5f; t.m[3][3] = 1.0f;
    return t;
}
// XMVector3Transform: (v, 1) * M, all four components
void transform_point(V3 v, const Mat4& m, float out[4])
{
    for (int j = 0; j < 4; ++j) out[j] = v.x * m.m[0][j] + v.y * m.m[1][j] + v.z * m.m[2][j] + m.m[3][j];
}

float randf(uint32_t* s) { return (float)crychic_msvc_rand(s) / 32767.0f; }  // MathHelper::RandF, RAND_MAX 0x7FFF

}  // namespace

extern "C" {

int crychic_msvc_rand(uint32_t* state)
{
    *state = *state * 214013u + 2531011u;
    return (int)((*state >> 16) & 0x7FFFu);
}

int crychic_calc_gauss_weights(float sigma, float* weights, int capacity)
{
    if (!weights || !(sigma > 0.0f)) return CRYCHIC_E_INVALID_ARG;
    const float twoSigma2 = 2.0f * sigma * sigma;
    const int radius = (int)std::ceil(2.0f * sigma);
    if (radius > 5) return CRYCHIC_E_INVALID_ARG;  // assert(blurRadius <= MaxBlurRadius), Ssao.cpp:45
    const int n = 2 * radius + 1;
    if (n > capacity) return CRYCHIC_E_INVALID_ARG;
    float sum = 0.0f;
    for (int i = -radius; i <= radius; ++i) {
        const float x = (float)i;
        weights[i + radius] = expf(-x * x / twoSigma2);
        sum += weights[i + radius];
    }
    for (int i = 0; i < n; ++i) weights[i] /= sum;
    return n;
}

void crychic_build_offset_vectors(uint32_t* rand_state, float offsets[14][4])
{
    // 8 cube corners then 6 face centres, opposite pairs adjacent (Ssao.cpp:431-451)
    int k = 0;
    const float corner[4][3] = { { 1, 1, 1 }, { -1, 1, 1 }, { 1, 1, -1 }, { -1, 1, -1 } };
    for (int c = 0; c < 4; ++c) {
        for (int s = 0; s < 2; ++s) {
            const float sg = s ? -1.0f : 1.0f;
            offsets[k][0] = sg * corner[c][0]; offsets[k][1] = sg * corner[c][1]; offsets[k][2] = sg * corner[c][2];
            offsets[k][3] = 0.0f;
            ++k;
        }
    }
    for (int axis = 0; axis < 3; ++axis) {
        for (int s = 0; s < 2; ++s) {
            offsets[k][0] = offsets[k][1] = offsets[k][2] = offsets[k][3] = 0.0f;
            offsets[k][axis] = s ? 1.0f : -1.0f;
            ++k;
        }
    }
    for (int i = 0; i < 14; ++i) {
        const float s = 0.25f + randf(rand_state) * (1.0f - 0.25f);  // RandF(0.25f, 1.0f), Ssao.cpp:456
        const float len = std::sqrt(offsets[i][0] * offsets[i][0] + offsets[i][1] * offsets[i][1] +
                                    offsets[i][2] * offsets[i][2] + offsets[i][3] * offsets[i][3]);
        for (int c = 0; c < 4; ++c) offsets[i][c] = s * (offsets[i][c] / len);  // s * XMVector4Normalize(v)
    }
}

void crychic_build_random_vector_texture(uint32_t* rand_state, int args_right_to_left, uint8_t* out)
{
    for (int t = 0; t < 256 * 256; ++t) {
        float v[3];
        if (args_right_to_left) { v[2] = randf(rand_state); v[1] = randf(rand_state); v[0] = randf(rand_state); }
        else { v[0] = randf(rand_state); v[1] = randf(rand_state); v[2] = randf(rand_state); }
        // XMCOLOR(v.x, v.y, v.z, 0) = A8R8G8B8: memory bytes B, G, R, A; sampled as R8G8B8A8 the shader's .r is v.z
        out[4 * t + 0] = (uint8_t)std::nearbyint(std::fmin(std::fmax(v[2], 0.0f), 1.0f) * 255.0f);
        out[4 * t + 1] = (uint8_t)std::nearbyint(std::fmin(std::fmax(v[1], 0.0f), 1.0f) * 255.0f);
        out[4 * t + 2] = (uint8_t)std::nearbyint(std::fmin(std::fmax(v[0], 0.0f), 1.0f) * 255.0f);
        out[4 * t + 3] = 0;
    }
}

int crychic_update_cascade_shadow_transform(const crychic_camera* cam, const float lightDir[3], uint32_t shadowMapWidth,
                                            float lightViewOut[4][16], float lightProjOut[4][16],
                                            float shadowTransformOut[4][16])
{
    if (!cam || !lightDir || !lightViewOut || !lightProjOut || !shadowTransformOut || shadowMapWidth == 0)
        return CRYCHIC_E_INVALID_ARG;
    const Mat4 view = camera_view(*cam);
    const float zNear[4] = { cam->nearZ, 30.0f, 50.0f, 80.0f };
    const float zFar[4] = { 30.0f, 50.0f, 80.0f, cam->farZ };
    const V3 ld{ lightDir[0], lightDir[1], lightDir[2] };
    for (int i = 0; i < 4; ++i) {
        Mat4 invViewProj;
        if (!inverse(view * perspective_fov_lh(cam->fovY, cam->aspect, zNear[i], zFar[i]), invViewProj))
            return CRYCHIC_E_INVALID_ARG;
        // NDC corners: near plane 0..3 (CW from top-left), far plane 4..7  (CRYCHIC.cpp:656-669)
        V3 corners[8];
        for (int j = 0; j < 8; ++j) {
            const V3 ndc{ (j & 3) == 1 || (j & 3) == 2 ? 1.0f : -1.0f, (j & 3) < 2 ? 1.0f : -1.0f, j < 4 ? 0.0f : 1.0f };
            float w[4];
            transform_point(ndc, invViewProj, w);
            corners[j] = { w[0] / w[3], w[1] / w[3], w[2] / w[3] };
        }
        const V3 dFar = corners[7] - corners[5], dDiag = corners[3] - corners[5];
        const float crossFar = std::sqrt(dFar.x * dFar.x + dFar.y * dFar.y + dFar.z * dFar.z);
        const float crossNear2Far = std::sqrt(dDiag.x * dDiag.x + dDiag.y * dDiag.y + dDiag.z * dDiag.z);
        const float boxLen = crossFar > crossNear2Far ? crossFar : crossNear2Far;      // :714
        const V3 target{ 0.5f * (corners[3].x + corners[5].x), 0.5f * (corners[3].y + corners[5].y),
                         0.5f * (corners[3].z + corners[5].z) };                       // :716-720
        const V3 lightPos{ -boxLen * ld.x + target.x, -boxLen * ld.y + target.y, -boxLen * ld.z + target.z };
        const Mat4 lightView = look_at_lh(lightPos, target, V3{ 0.0f, 1.0f, 0.0f });  // :734
        float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (int j = 0; j < 8; ++j) {                                                  // :738-753
            float c[4];
            transform_point(corners[j], lightView, c);
            for (int k = 0; k < 3; ++k) { lo[k] = std::fmin(lo[k], c[k]); hi[k] = std::fmax(hi[k], c[k]); }
        }
        const float unitsPerTexel = boxLen / (float)shadowMapWidth;                    // :758
        float centre[3];
        for (int k = 0; k < 3; ++k) {                                                  // :759-772
            centre[k] = 0.5f * (lo[k] + hi[k]);
            centre[k] /= unitsPerTexel;
            centre[k] = std::floor(centre[k]);
            centre[k] *= unitsPerTexel;
        }
        const double half = 0.5 * boxLen;                                              // :789-794
        const Mat4 lightProj = ortho_off_center_lh((float)(centre[0] - half), (float)(centre[0] + half),
                                                   (float)(centre[1] - half), (float)(centre[1] + half),
                                                   (float)(centre[2] - half), (float)(centre[2] + half));  // :804
        (lightView * lightProj * tex_matrix()).store(shadowTransformOut[i]);          // :810
        lightView.store(lightViewOut[i]);
        lightProj.store(lightProjOut[i]);
    }
    return 0;
}

int crychic_update_main_pass_cb(const crychic_camera* cam, uint32_t W, uint32_t H, const float shadowTransform[4][16],
                                const float lightDirs[3][3], crychic_pass_constants* out)
{
    if (!cam || !shadowTransform || !lightDirs || !out || W == 0 || H == 0) return CRYCHIC_E_INVALID_ARG;
    const Mat4 view = camera_view(*cam);
    const Mat4 proj = perspective_fov_lh(cam->fovY, cam->aspect, cam->nearZ, cam->farZ);
    const Mat4 viewProj = view * proj;
    Mat4 invView, invProj, invViewProj;
    if (!inverse(view, invView) || !inverse(proj, invProj) || !inverse(viewProj, invViewProj)) return CRYCHIC_E_INVALID_ARG;
    std::memset(out, 0, sizeof *out);
    for (int i = 0; i < 4; ++i) {  // cascades 0..3; slots 4..11 are uninitialised in the reference (:837-841), zero here
        Mat4 s;
        std::memcpy(s.m, shadowTransform[i], sizeof s.m);
        s.store_transposed(out->ShadowTransforms[i]);
    }
    view.store_transposed(out->View);
    invView.store_transposed(out->InvView);
    proj.store_transposed(out->Proj);
    invProj.store_transposed(out->InvProj);
    viewProj.store_transposed(out->ViewProj);
    invViewProj.store_transposed(out->InvViewProj);
    (viewProj * tex_matrix()).store_transposed(out->ViewProjTex);
    std::memcpy(out->EyePosW, cam->pos, sizeof out->EyePosW);
    out->RenderTargetSize[0] = (float)W; out->RenderTargetSize[1] = (float)H;
    out->InvRenderTargetSize[0] = 1.0f / W; out->InvRenderTargetSize[1] = 1.0f / H;
    out->NearZ = 1.0f;
    out->FarZ = 1000.0f;  // sic, CRYCHIC.cpp:855
    out->AmbientLight[0] = 0.4f; out->AmbientLight[1] = 0.4f; out->AmbientLight[2] = 0.6f; out->AmbientLight[3] = 1.0f;
    for (auto& L : out->Lights) {  // Light defaults, Common/d3dUtil.h:216-224
        L.Strength[0] = L.Strength[1] = L.Strength[2] = 0.5f;
        L.FalloffStart = 1.0f;
        L.Direction[0] = 0.0f; L.Direction[1] = -1.0f; L.Direction[2] = 0.0f;
        L.FalloffEnd = 10.0f;
        L.Position[0] = L.Position[1] = L.Position[2] = 0.0f;
        L.SpotPower = 64.0f;
    }
    const float strength[3][3] = { { 2.4f, 2.4f, 2.5f }, { 0.1f, 0.1f, 0.1f }, { 0.0f, 0.0f, 0.0f } };  // :859-864
    for (int i = 0; i < 3; ++i) {
        std::memcpy(out->Lights[i].Direction, lightDirs[i], 12);
        std::memcpy(out->Lights[i].Strength, strength[i], 12);
    }
    return 0;
}

int crychic_update_ssao_cb(const crychic_camera* cam, uint32_t W, uint32_t H, const float offsets[14][4],
                           crychic_ssao_constants* out)
{
    if (!cam || !offsets || !out || W < 2 || H < 2) return CRYCHIC_E_INVALID_ARG;
    const Mat4 proj = perspective_fov_lh(cam->fovY, cam->aspect, cam->nearZ, cam->farZ);
    Mat4 invProj;
    if (!inverse(proj, invProj)) return CRYCHIC_E_INVALID_ARG;
    std::memset(out, 0, sizeof *out);
    proj.store_transposed(out->Proj);
    invProj.store_transposed(out->InvProj);
    (proj * tex_matrix()).store_transposed(out->ProjTex);
    std::memcpy(out->OffsetVectors, offsets, sizeof out->OffsetVectors);
    float w[12] = { 0 };  // the 12th float read at CRYCHIC.cpp:925 is out of bounds in the reference; zero here
    crychic_calc_gauss_weights(2.5f, w, 11);
    std::memcpy(out->BlurWeights, w, sizeof out->BlurWeights);
    out->InvRenderTargetSize[0] = 1.0f / (float)(W / 2);  // :927; RenderTargetSize stays (0,0) as in the reference
    out->InvRenderTargetSize[1] = 1.0f / (float)(H / 2);
    out->OcclusionRadius = 0.5f;
    out->OcclusionFadeStart = 0.2f;
    out->OcclusionFadeEnd = 1.0f;
    out->SurfaceEpsilon = 0.05f;
    return 0;
}

// CRYCHIC::UpdateInstanceData's visibility test, CRYCHIC.cpp:515-564, with the frustum of CRYCHIC.cpp:115.
// DirectXCollision is not part of the reference checkout; this restates its published behaviour:
//   BoundingFrustum::CreateFromMatrix(proj): origin 0, identity orientation, slopes +-1/P00 and +-1/P11, near / far;
//   Transform(viewToLocal): rotation = the matrix's normalised rows, origin = its translation, near / far scaled by the
//     largest row length (uniform scale assumed), slopes kept;
//   Contains(BoundingBox) != DISJOINT: six outward planes, each normalised; the box is outside a plane when
//     dot(center, n) + d > dot(extents, |n|).
int crychic_frustum_cull(const crychic_camera* cam, const float boundsCenter[3], const float boundsExtents[3],
                         const float* worlds, uint32_t count, uint8_t* visible)
{
    if (!cam || !boundsCenter || !boundsExtents || (count && (!worlds || !visible))) return CRYCHIC_E_INVALID_ARG;
    const Mat4 view = camera_view(*cam);
    const Mat4 proj = perspective_fov_lh(cam->fovY, cam->aspect, cam->nearZ, cam->farZ);
    Mat4 invView;
    if (!inverse(view, invView)) return CRYCHIC_E_INVALID_ARG;
    const float rs = 1.0f / proj.m[0][0], ts = 1.0f / proj.m[1][1];
    int nvis = 0;
    for (uint32_t i = 0; i < count; ++i) {
        Mat4 world, invWorld;
        std::memcpy(world.m, worlds + 16 * (size_t)i, sizeof world.m);
        if (!inverse(world, invWorld)) { visible[i] = 1; ++nvis; continue; }   // degenerate instance: never culled
        const Mat4 m = invView * invWorld;                                     // view space -> the instance's local space
        V3 r[3];
        float scale2 = 0.0f;
        for (int k = 0; k < 3; ++k) {
            const V3 row{ m.m[k][0], m.m[k][1], m.m[k][2] };
            scale2 = std::max(scale2, dot(row, row));
            r[k] = normalize(row);
        }
        const float scale = std::sqrt(scale2);
        const V3 origin{ m.m[3][0], m.m[3][1], m.m[3][2] };
        const float zn = cam->nearZ * scale, zf = cam->farZ * scale;
        // frustum-space planes (n, d): near, far, right, left, top, bottom
        const float pl[6][4] = { { 0, 0, -1, zn }, { 0, 0, 1, -zf }, { 1, 0, -rs, 0 }, { -1, 0, -rs, 0 }, { 0, 1, -ts, 0 }, { 0, -1, -ts, 0 } };
        bool outside = false;
        for (int p = 0; p < 6 && !outside; ++p) {
            V3 n{ pl[p][0] * r[0].x + pl[p][1] * r[1].x + pl[p][2] * r[2].x, pl[p][0] * r[0].y + pl[p][1] * r[1].y + pl[p][2] * r[2].y,
                  pl[p][0] * r[0].z + pl[p][1] * r[1].z + pl[p][2] * r[2].z };
            float d = pl[p][3] - dot(n, origin);
            const float len = std::sqrt(dot(n, n));
            n = V3{ n.x / len, n.y / len, n.z / len };
            d /= len;
            const float dist = n.x * boundsCenter[0] + n.y * boundsCenter[1] + n.z * boundsCenter[2] + d;
            const float radius = std::fabs(n.x) * boundsExtents[0] + std::fabs(n.y) * boundsExtents[1] + std::fabs(n.z) * boundsExtents[2];
            outside = dist > radius;
        }
        visible[i] = outside ? 0 : 1;
        nvis += outside ? 0 : 1;
    }
    return nvis;
}

float crychic_pcf_search_radius(uint32_t width, int literal)
{
    if (width == 0) return 0.0f;
    if (literal) return (float)(5u / width) / 2.0f;  // `5 / width / 2.0f` with uint width: integer division first
    return 5.0f / (float)width / 2.0f;
}

}  // extern "C"

static_assert(sizeof(crychic_light) == 48, "Light ABI (d3dUtil.h:216-224)");
static_assert(sizeof(crychic_pass_constants) == 2048, "PassConstants ABI (FrameResource.h:29-51)");
static_assert(sizeof(crychic_ssao_constants) == 496, "SsaoConstants ABI (FrameResource.h:53-67)");
