/*
 * or_constants.c -- oracle restatement of the host-side constant builders that feed the hot path
 * (TEST INFRASTRUCTURE, parity unpinned: see crychic_oracle.h).  DirectXMath is not available here;
 * its LH matrix constructors are restated from their documented definitions.
 */
#include "crychic_oracle.h"
#include "or_math.h"
#ifdef _OPENMP
#include <omp.h>
#endif

int or_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;   /* the sanitizer build runs single-threaded */
#endif
}
void or_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* MSVC CRT rand() (ucrt rand.cpp): state = state * 214013 + 2531011; return (state >> 16) & 0x7fff. */
int or_msvc_rand(uint32_t* state)
{
    *state = *state * 214013u + 2531011u;
    return (int)((*state >> 16) & 0x7FFFu);
}
/* Common/MathHelper.h:17-20  (float)(rand()) / (float)RAND_MAX */
float or_randf(uint32_t* state) { return (float)or_msvc_rand(state) / (float)0x7FFF; }
/* Common/MathHelper.h:23-26  a + RandF()*(b-a) */
float or_randf_range(uint32_t* state, float a, float b) { return a + or_randf(state) * (b - a); }

/* Ssao.cpp:37-68 */
int or_calc_gauss_weights(float sigma, float* weights, int capacity)
{
    float twoSigma2 = 2.0f * sigma * sigma;
    int blurRadius = (int)ceil(2.0f * sigma);
    int n = 2 * blurRadius + 1;
    if (n > capacity) return -1;
    float weightSum = 0.0f;
    for (int i = -blurRadius; i <= blurRadius; ++i) {
        float x = (float)i;
        weights[i + blurRadius] = expf(-x * x / twoSigma2);
        weightSum += weights[i + blurRadius];
    }
    for (int i = 0; i < n; ++i) weights[i] /= weightSum;
    return n;
}

/* Ssao.cpp:423-462 */
void or_build_offset_vectors(uint32_t* rand_state, float offsets[14][4])
{
    static const float base[14][4] = {
        { +1, +1, +1, 0 }, { -1, -1, -1, 0 }, { -1, +1, +1, 0 }, { +1, -1, -1, 0 },
        { +1, +1, -1, 0 }, { -1, -1, +1, 0 }, { -1, +1, -1, 0 }, { +1, -1, +1, 0 },
        { -1, 0, 0, 0 }, { +1, 0, 0, 0 }, { 0, -1, 0, 0 }, { 0, +1, 0, 0 }, { 0, 0, -1, 0 }, { 0, 0, +1, 0 }
    };
    for (int i = 0; i < 14; ++i) {
        float s = or_randf_range(rand_state, 0.25f, 1.0f);
        const float* b = base[i];
        float len = sqrtf(((b[0] * b[0] + b[1] * b[1]) + b[2] * b[2]) + b[3] * b[3]); /* XMVector4Normalize */
        for (int c = 0; c < 4; ++c) offsets[i][c] = s * (b[c] / len);
    }
}

/* Ssao.cpp:392-402.  XMCOLOR(r,g,b,a) stores round(sat(c)*255) packed as A8R8G8B8 (bytes B,G,R,A). */
void or_build_random_vector_texture(uint32_t* rand_state, int args_right_to_left, uint8_t* out)
{
    for (int i = 0; i < 256; ++i) {
        for (int j = 0; j < 256; ++j) {
            float vx, vy, vz;
            if (args_right_to_left) { vz = or_randf(rand_state); vy = or_randf(rand_state); vx = or_randf(rand_state); }
            else { vx = or_randf(rand_state); vy = or_randf(rand_state); vz = or_randf(rand_state); }
            uint8_t* t = out + ((size_t)i * 256 + j) * 4;
            t[0] = (uint8_t)nearbyintf(or_saturate(vz) * 255.0f); /* B byte -> shader .r */
            t[1] = (uint8_t)nearbyintf(or_saturate(vy) * 255.0f); /* G */
            t[2] = (uint8_t)nearbyintf(or_saturate(vx) * 255.0f); /* R byte -> shader .b */
            t[3] = 0;                                             /* A */
        }
    }
}

/* ---- LH matrix helpers (row-vector convention: v' = v * M) ------------------------------------- */
void or_mat_mul(const float a[16], const float b[16], float out[16])
{
    float r[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r[4 * i + j] = ((a[4 * i + 0] * b[0 + j] + a[4 * i + 1] * b[4 + j]) + a[4 * i + 2] * b[8 + j]) + a[4 * i + 3] * b[12 + j];
    memcpy(out, r, sizeof r);
}
void or_mat_transpose(const float m[16], float out[16])
{
    float r[16];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r[4 * j + i] = m[4 * i + j];
    memcpy(out, r, sizeof r);
}
/* XMMatrixInverse: general 4x4 inverse (cofactor expansion, evaluated in double then rounded). */
int or_mat_inverse(const float m[16], float out[16])
{
    double a[16], inv[16];
    for (int i = 0; i < 16; ++i) a[i] = m[i];
    inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
    inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
    inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
    inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
    inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
    inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
    inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
    inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
    inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
    inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
    inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
    inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
    inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
    inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
    inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
    inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
    double det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
    if (det == 0.0) return -1;
    for (int i = 0; i < 16; ++i) out[i] = (float)(inv[i] / det);
    return 0;
}
/* XMMatrixPerspectiveFovLH */
void or_mat_perspective_fov_lh(float fovY, float aspect, float zn, float zf, float out[16])
{
    float s = sinf(0.5f * fovY), c = cosf(0.5f * fovY);
    float h = c / s, w = h / aspect, fRange = zf / (zf - zn);
    memset(out, 0, 16 * sizeof(float));
    out[0] = w; out[5] = h; out[10] = fRange; out[11] = 1.0f; out[14] = -fRange * zn;
}
static void cross3(const float a[3], const float b[3], float o[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}
static void norm3(const float v[3], float o[3])
{
    float l = sqrtf(or_dot3_host(v, v));
    o[0] = v[0] / l; o[1] = v[1] / l; o[2] = v[2] / l;
}
/* XMMatrixLookAtLH = XMMatrixLookToLH(eye, at - eye, up) */
void or_mat_look_at_lh(const float eye[3], const float at[3], const float up[3], float out[16])
{
    float dir[3] = { at[0] - eye[0], at[1] - eye[1], at[2] - eye[2] }, r0[3], r1[3], r2[3], t[3];
    norm3(dir, r2);
    cross3(up, r2, t); norm3(t, r0);
    cross3(r2, r0, r1);
    float neg[3] = { -eye[0], -eye[1], -eye[2] };
    float d0 = or_dot3_host(r0, neg), d1 = or_dot3_host(r1, neg), d2 = or_dot3_host(r2, neg);
    float m[16] = { r0[0], r1[0], r2[0], 0, r0[1], r1[1], r2[1], 0, r0[2], r1[2], r2[2], 0, d0, d1, d2, 1 };
    memcpy(out, m, sizeof m);
}
/* XMMatrixOrthographicOffCenterLH */
void or_mat_ortho_off_center_lh(float l, float r, float b, float t, float zn, float zf, float out[16])
{
    float rw = 1.0f / (r - l), rh = 1.0f / (t - b), fRange = 1.0f / (zf - zn);
    memset(out, 0, 16 * sizeof(float));
    out[0] = rw + rw; out[5] = rh + rh; out[10] = fRange;
    out[12] = -(l + r) * rw; out[13] = -(t + b) * rh; out[14] = -fRange * zn; out[15] = 1.0f;
}

/* Camera::UpdateViewMatrix  Common/Camera.cpp:226-273 */
static void camera_view(const or_camera* cam, float view[16])
{
    float L[3], U[3], R[3], up[3] = { cam->up[0], cam->up[1], cam->up[2] };
    norm3(cam->look, L);
    float r0[3];
    cross3(up, L, r0); norm3(r0, R);          /* right = up x look (LH) */
    float t[3];
    cross3(L, R, t); norm3(t, U);             /* U = normalize(L x R) */
    cross3(U, L, R);                          /* R = U x L */
    float x = -or_dot3_host(cam->pos, R), y = -or_dot3_host(cam->pos, U), z = -or_dot3_host(cam->pos, L);
    float m[16] = { R[0], U[0], L[0], 0, R[1], U[1], L[1], 0, R[2], U[2], L[2], 0, x, y, z, 1 };
    memcpy(view, m, sizeof m);
}

/* XMVector3Transform: (x,y,z,1) * M, all four components. */
static void xform3(const float v[3], const float m[16], float o[4])
{
    for (int j = 0; j < 4; ++j) o[j] = ((v[0] * m[0 + j] + v[1] * m[4 + j]) + v[2] * m[8 + j]) + m[12 + j];
}

static const float TEX_T[16] = { 0.5f, 0, 0, 0, 0, -0.5f, 0, 0, 0, 0, 1, 0, 0.5f, 0.5f, 0, 1 }; /* CRYCHIC.cpp:828-832 */

/* CRYCHIC.cpp:634-815 */
void or_cascade_shadow_transforms(const or_camera* cam, const float lightDir[3], uint32_t shadowMapWidth,
                                  float lightViewOut[4][16], float lightProjOut[4][16], float shadowTransformOut[4][16])
{
    float view[16];
    camera_view(cam, view);
    float zNear[4] = { cam->nearZ, 30.0f, 50.0f, 80.0f };            /* :640 */
    float zFar[4] = { 30.0f, 50.0f, 80.0f, cam->farZ };              /* :641 */
    for (int i = 0; i < 4; ++i) {
        float proj[16], vp[16], invVP[16];
        or_mat_perspective_fov_lh(cam->fovY, cam->aspect, zNear[i], zFar[i], proj); /* :646 */
        or_mat_mul(view, proj, vp);
        or_mat_inverse(vp, invVP);                                    /* :650 */
        float corners[8][4] = {                                       /* :656-669 */
            { -1, +1, 0, 1 }, { +1, +1, 0, 1 }, { +1, -1, 0, 1 }, { -1, -1, 0, 1 },
            { -1, +1, 1, 1 }, { +1, +1, 1, 1 }, { +1, -1, 1, 1 }, { -1, -1, 1, 1 }
        };
        for (int j = 0; j < 8; ++j) {                                 /* :687-697 */
            float w[4];
            xform3(corners[j], invVP, w);
            corners[j][0] = w[0] / w[3]; corners[j][1] = w[1] / w[3]; corners[j][2] = w[2] / w[3]; corners[j][3] = w[3];
        }
        float dxf = corners[7][0] - corners[5][0], dyf = corners[7][1] - corners[5][1], dzf = corners[7][2] - corners[5][2];
        float crossFar = sqrtf(dxf * dxf + dyf * dyf + dzf * dzf);   /* :708-710 */
        float dxn = corners[3][0] - corners[5][0], dyn = corners[3][1] - corners[5][1], dzn = corners[3][2] - corners[5][2];
        float crossNear2Far = sqrtf(dxn * dxn + dyn * dyn + dzn * dzn); /* :711-713 */
        float boundingBoxLength = crossFar > crossNear2Far ? crossFar : crossNear2Far; /* :714 */
        float target[3] = { 0.5f * (corners[3][0] + corners[5][0]), 0.5f * (corners[3][1] + corners[5][1]),
                            0.5f * (corners[3][2] + corners[5][2]) }; /* :716-720 */
        float distance = boundingBoxLength;                           /* :725 */
        float lightPos[3] = { -distance * lightDir[0] + target[0], -distance * lightDir[1] + target[1],
                              -distance * lightDir[2] + target[2] }; /* :727-731 */
        float up[3] = { 0, 1, 0 };
        float lightView[16];
        or_mat_look_at_lh(lightPos, target, up, lightView);           /* :734 */
        float vmin[3] = { INFINITY, INFINITY, INFINITY }, vmax[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (int j = 0; j < 8; ++j) {                                 /* :738-753 */
            float c4[4];
            xform3(corners[j], lightView, c4);
            for (int k = 0; k < 3; ++k) { if (c4[k] < vmin[k]) vmin[k] = c4[k]; if (c4[k] > vmax[k]) vmax[k] = c4[k]; }
        }
        float fWorldUnitsPerTexel = boundingBoxLength / (float)shadowMapWidth; /* :758 */
        float fCenter[3];
        for (int k = 0; k < 3; ++k) {                                 /* :759-772 */
            fCenter[k] = 0.5f * (vmin[k] + vmax[k]);
            fCenter[k] /= fWorldUnitsPerTexel;
            fCenter[k] = floorf(fCenter[k]);
            fCenter[k] *= fWorldUnitsPerTexel;
        }
        float l = (float)(fCenter[0] - 0.5 * boundingBoxLength);      /* :789-794 (double 0.5 literal) */
        float b = (float)(fCenter[1] - 0.5 * boundingBoxLength);
        float n = (float)(fCenter[2] - 0.5 * boundingBoxLength);
        float r = (float)(fCenter[0] + 0.5 * boundingBoxLength);
        float t = (float)(fCenter[1] + 0.5 * boundingBoxLength);
        float f = (float)(fCenter[2] + 0.5 * boundingBoxLength);
        float lightProj[16], tmp[16];
        or_mat_ortho_off_center_lh(l, r, b, t, n, f, lightProj);      /* :804 */
        or_mat_mul(lightView, lightProj, tmp);
        or_mat_mul(tmp, TEX_T, shadowTransformOut[i]);                /* :810 */
        memcpy(lightViewOut[i], lightView, sizeof lightView);
        memcpy(lightProjOut[i], lightProj, sizeof lightProj);
    }
}

static void default_light(or_light* L) /* Common/d3dUtil.h:216-224 */
{
    L->Strength[0] = L->Strength[1] = L->Strength[2] = 0.5f; L->FalloffStart = 1.0f;
    L->Direction[0] = 0.0f; L->Direction[1] = -1.0f; L->Direction[2] = 0.0f; L->FalloffEnd = 10.0f;
    L->Position[0] = L->Position[1] = L->Position[2] = 0.0f; L->SpotPower = 64.0f;
}

/* CRYCHIC.cpp:817-868 */
void or_build_pass_constants(const or_camera* cam, uint32_t W, uint32_t H, const float shadowTransform[4][16],
                             const float lightDirs[3][3], or_pass_constants* out)
{
    float view[16], proj[16], viewProj[16], invView[16], invProj[16], invViewProj[16], viewProjTex[16];
    camera_view(cam, view);
    or_mat_perspective_fov_lh(cam->fovY, cam->aspect, cam->nearZ, cam->farZ, proj);
    or_mat_mul(view, proj, viewProj);
    or_mat_inverse(view, invView);
    or_mat_inverse(proj, invProj);
    or_mat_inverse(viewProj, invViewProj);
    or_mat_mul(viewProj, TEX_T, viewProjTex);                         /* :834 */
    memset(out, 0, sizeof *out);
    for (int i = 0; i < 4; ++i) or_mat_transpose(shadowTransform[i], out->ShadowTransforms[i]); /* :837-841 */
    or_mat_transpose(view, out->View);                                /* :843-849 */
    or_mat_transpose(invView, out->InvView);
    or_mat_transpose(proj, out->Proj);
    or_mat_transpose(invProj, out->InvProj);
    or_mat_transpose(viewProj, out->ViewProj);
    or_mat_transpose(invViewProj, out->InvViewProj);
    or_mat_transpose(viewProjTex, out->ViewProjTex);
    memcpy(out->EyePosW, cam->pos, 12);                               /* :851 */
    out->RenderTargetSize[0] = (float)W; out->RenderTargetSize[1] = (float)H;
    out->InvRenderTargetSize[0] = 1.0f / (float)W; out->InvRenderTargetSize[1] = 1.0f / (float)H;
    out->NearZ = 1.0f; out->FarZ = 1000.0f;                           /* :854-855 (Q10) */
    out->AmbientLight[0] = 0.4f; out->AmbientLight[1] = 0.4f; out->AmbientLight[2] = 0.6f; out->AmbientLight[3] = 1.0f;
    for (int i = 0; i < OR_MAX_LIGHTS; ++i) default_light(&out->Lights[i]);
    static const float strength[3][3] = { { 2.4f, 2.4f, 2.5f }, { 0.1f, 0.1f, 0.1f }, { 0.0f, 0.0f, 0.0f } }; /* :859-864 */
    for (int i = 0; i < 3; ++i) {
        memcpy(out->Lights[i].Direction, lightDirs[i], 12);
        memcpy(out->Lights[i].Strength, strength[i], 12);
    }
}

/* CRYCHIC.cpp:903-937 */
void or_build_ssao_constants(const or_camera* cam, uint32_t W, uint32_t H, const float offsets[14][4],
                             or_ssao_constants* out)
{
    float proj[16], invProj[16], projTex[16];
    or_mat_perspective_fov_lh(cam->fovY, cam->aspect, cam->nearZ, cam->farZ, proj);
    or_mat_inverse(proj, invProj);
    or_mat_mul(proj, TEX_T, projTex);
    memset(out, 0, sizeof *out);
    or_mat_transpose(proj, out->Proj);                                /* :916 */
    or_mat_transpose(invProj, out->InvProj);                          /* :917 */
    or_mat_transpose(projTex, out->ProjTex);                          /* :918 */
    memcpy(out->OffsetVectors, offsets, sizeof out->OffsetVectors);   /* :920 */
    float w[12] = { 0 };
    or_calc_gauss_weights(2.5f, w, 11);                               /* :922; w[11] is the OOB read (Q11), zero here */
    memcpy(out->BlurWeights, w, sizeof out->BlurWeights);             /* :923-925 */
    out->InvRenderTargetSize[0] = 1.0f / (float)(W / 2);              /* :927 */
    out->InvRenderTargetSize[1] = 1.0f / (float)(H / 2);
    out->OcclusionRadius = 0.5f;                                      /* :930-933 */
    out->OcclusionFadeStart = 0.2f;
    out->OcclusionFadeEnd = 1.0f;
    out->SurfaceEpsilon = 0.05f;
}

_Static_assert(sizeof(or_light) == 48, "Light ABI");
_Static_assert(sizeof(or_pass_constants) == 2048, "PassConstants ABI");
_Static_assert(sizeof(or_ssao_constants) == 496, "SsaoConstants ABI");
_Static_assert(offsetof(or_pass_constants, ShadowTransforms) == 448, "ShadowTransforms@448");
_Static_assert(offsetof(or_pass_constants, EyePosW) == 1216, "EyePosW@1216");
_Static_assert(offsetof(or_pass_constants, AmbientLight) == 1264, "AmbientLight@1264");
_Static_assert(offsetof(or_pass_constants, Lights) == 1280, "Lights@1280");
_Static_assert(offsetof(or_ssao_constants, OffsetVectors) == 192, "OffsetVectors@192");
_Static_assert(offsetof(or_ssao_constants, BlurWeights) == 416, "BlurWeights@416");
_Static_assert(offsetof(or_ssao_constants, OcclusionRadius) == 480, "OcclusionRadius@480");

/* CRYCHIC::UpdateInstanceData visibility (CRYCHIC.cpp:515-564), stated geometrically and in double precision: the
 * instance's bounding box is DISJOINT from the frustum exactly when all eight corners lie outside one of the six clip
 * planes of world * view * proj.  margin[i] (optional) = min over planes of the largest corner "insideness"
 * (>= 0 <=> visible); the product's float plane test may only disagree where |margin| is at rounding level. */
int or_frustum_cull(const or_camera* cam, const float center[3], const float extents[3], const float* worlds, uint32_t count,
                    uint8_t* visible, double* margin)
{
    float view[16], proj[16];
    camera_view(cam, view);
    or_mat_perspective_fov_lh(cam->fovY, cam->aspect, cam->nearZ, cam->farZ, proj);
    int nvis = 0;
    for (uint32_t i = 0; i < count; ++i) {
        const float* w = worlds + 16 * (size_t)i;
        double best[6];
        for (int p = 0; p < 6; ++p) best[p] = -1e300;
        for (int c = 0; c < 8; ++c) {
            double l[4] = { center[0] + ((c & 1) ? extents[0] : -extents[0]), center[1] + ((c & 2) ? extents[1] : -extents[1]),
                            center[2] + ((c & 4) ? extents[2] : -extents[2]), 1.0 };
            double a[4], b[4], h[4];
            for (int j = 0; j < 4; ++j) a[j] = l[0] * w[j] + l[1] * w[4 + j] + l[2] * w[8 + j] + l[3] * w[12 + j];
            for (int j = 0; j < 4; ++j) b[j] = a[0] * view[j] + a[1] * view[4 + j] + a[2] * view[8 + j] + a[3] * view[12 + j];
            for (int j = 0; j < 4; ++j) h[j] = b[0] * proj[j] + b[1] * proj[4 + j] + b[2] * proj[8 + j] + b[3] * proj[12 + j];
            const double s[6] = { h[2], h[3] - h[2], h[3] - h[0], h[3] + h[0], h[3] - h[1], h[3] + h[1] };
            for (int p = 0; p < 6; ++p) if (s[p] > best[p]) best[p] = s[p];
        }
        double m = best[0];
        for (int p = 1; p < 6; ++p) if (best[p] < m) m = best[p];
        if (margin) margin[i] = m;
        visible[i] = m >= 0.0 ? 1 : 0;
        nvis += visible[i];
    }
    return nvis;
}
