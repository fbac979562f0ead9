// light_budget.hip -- per-section instruction budget of light_kernel<true, false> (the benchmark's instantiation of
// Shaders/DeferredShading.hlsl:23-101).  Each section of light_core.hpp's light_pixel is instantiated as a kernel of its own,
// fed from memory and stored to memory so that nothing is folded away; tools/light_budget.py compiles this file for gfx950,
// counts the VALU / memory instructions of every kernel's hot path (the wave-uniform fast paths the benchmark frame takes) and
// subtracts the `frame` kernel (loads + stores only).  Not part of the product.
#include <hip/hip_runtime.h>
#include "light_core.hpp"

using namespace cry;

#define SECTION(name) extern "C" __global__ __launch_bounds__(256) void sec_##name(LightParams P, const f4a* __restrict__ in, f4a* __restrict__ out, \
                                                                                 const uint16_t* __restrict__ ambient, const uint32_t* __restrict__ cube)
#define IDX const uint32_t idx = blockIdx.x * 256u + threadIdx.x

SECTION(frame)          // three 16-byte loads, one 16-byte store: the cost every section below carries as well
{
    IDX;
    const f4a a = in[idx], b = in[idx + 1000000u], c = in[idx + 2000000u];
    out[idx] = f4a{ a.x + b.x, a.y + b.y + c.x, a.z + c.y, a.w + c.z };
}
SECTION(decode)         // GBuffer.hlsl:37-41 + DeferredShading.hlsl:32-35: normalW, toEye, view, R0, distance
{
    IDX;
    const f4a G0 = in[idx], G1 = in[idx + 1000000u], G2 = in[idx + 2000000u];
    const f3 n = normalize3(f3{ G2.x, G2.y, G2.z });
    const f3 toEye{ P.EyePosW[0] - G0.x, P.EyePosW[1] - G0.y, P.EyePosW[2] - G0.z };
    const f3 view = normalize3(toEye);
    const float distance = len_from_sq(dot3(toEye, toEye));
    const f3 R0{ lerpf(0.04f, G1.x, G0.w), lerpf(0.04f, G1.y, G0.w), lerpf(0.04f, G1.z, G0.w) };
    out[idx] = f4a{ n.x + view.x + R0.x, n.y + view.y + R0.y, n.z + view.z + R0.z, distance };
}
SECTION(ambient)        // :40-42: projection to the ambient map, four texel loads, the all-ones resolve
{
    IDX;
    const f4a G0 = in[idx], b = in[idx + 1000000u], c = in[idx + 2000000u];
    const AmbientPairs af = ambient_fetch_projected(P, ambient, true, (const uint16_t*)cube, f3{ G0.x, G0.y, G0.z });
    const float a = ambient_resolve(P, ambient, af);
    out[idx] = f4a{ a * P.AmbientLight[0] * b.x, a * P.AmbientLight[1] * b.y, a * P.AmbientLight[2] * b.z, c.x };
}
SECTION(cube_fetch)     // :94-95 reflect + face selection + footprint addresses + two 8-byte loads
{
    IDX;
    const f4a v = in[idx], n = in[idx + 1000000u], c = in[idx + 2000000u];
    const f3 r = reflect3(f3{ -v.x, -v.y, -v.z }, f3{ n.x, n.y, n.z });
    const CubeRows cf = cube_fetch(cube, P.cubeDim, r);
    out[idx] = f4a{ u2f(cf.r0.lo), u2f(cf.r0.hi ^ cf.r1.lo), u2f(cf.r1.hi), cf.fx + cf.fy + c.x + (float)cf.i0 };
}
SECTION(cube_resolve)   // :95-97 decode + filter of the footprint, Schlick fresnel, the specular mad
{
    IDX;
    const f4a t = in[idx], n = in[idx + 1000000u], r = in[idx + 2000000u];
    CubeRows cf;
    cf.r0 = RawPair{ f2u(t.x), f2u(t.y) }; cf.r1 = RawPair{ f2u(t.z), f2u(t.w) }; cf.fx = n.w; cf.fy = r.w; cf.i0 = (int)threadIdx.x;
    const f4 refl = cube_resolve<false>(cube_pick(cf, P.cubeDim));
    const float f0 = 1.0f - saturate(dot3(f3{ n.x, n.y, n.z }, f3{ r.x, r.y, r.z }));
    const float f5 = f0 * f0 * f0 * f0 * f0;
    const float sh = 1.0f - t.w;
    out[idx] = f4a{ fma(sh * fma(1.0f - n.x, f5, n.x), refl.x, r.x), fma(sh * fma(1.0f - n.y, f5, n.y), refl.y, r.y),
                    fma(sh * fma(1.0f - n.z, f5, n.z), refl.z, r.z), 1.0f };
}
SECTION(cascade)        // :53-76 on the wave-uniform path: two cascade lookups fetched together, compared, filtered, blended
{
    IDX;
    const f4a G0 = in[idx], b = in[idx + 1000000u], c = in[idx + 2000000u];
    float s = 1.0f;
    CascadeTexels ct;
    int J;
    if (cascade_uniform_test<true>(P, f3{ G0.x, G0.y, G0.z }, b.x, false, J)) { cascade_uniform_fetch(P, f3{ G0.x, G0.y, G0.z }, J, ct); s = cascade_uniform_resolve(P, ct); }
    out[idx] = f4a{ s, b.y, c.x, c.y };
}
SECTION(guard)          // "dark lights": the input bounds of the wavefront
{
    IDX;
    const f4a G0 = in[idx], G1 = in[idx + 1000000u], G2 = in[idx + 2000000u];
    bool bounded = light_dark_guard(G0, G1, G2);
    bounded = __builtin_amdgcn_ballot_w64(!bounded) == 0;
    out[idx] = f4a{ bounded ? G0.x : G1.x, G0.y, G2.x, G1.y };
}
SECTION(one_light)      // PBR.hlsl:99-106 for one directional light (short reciprocals), incl. what the loop hoists
{
    IDX;
    const f4a G0 = in[idx], G1 = in[idx + 1000000u], G2 = in[idx + 2000000u];
    f3 direct{ G0.w, G1.w * 0.5f, G2.w };
    pbr_dir_light<true>(P.Lights[0], f3{ G1.x, G1.y, G1.z }, G1.w, G0.w, f3{ G2.x, G2.y, G2.z }, f3{ G0.x, G0.y, G0.z }, G2.w, direct);
    out[idx] = f4a{ direct.x, direct.y, direct.z, 0.0f };
}
SECTION(two_lights)     // the same for two lights: the difference to one_light is the marginal cost of a light
{
    IDX;
    const f4a G0 = in[idx], G1 = in[idx + 1000000u], G2 = in[idx + 2000000u];
    f3 direct{ G0.w, G1.w * 0.5f, G2.w };
    pbr_dir_light<true>(P.Lights[0], f3{ G1.x, G1.y, G1.z }, G1.w, G0.w, f3{ G2.x, G2.y, G2.z }, f3{ G0.x, G0.y, G0.z }, G2.w, direct);
    pbr_dir_light<true>(P.Lights[1], f3{ G1.x, G1.y, G1.z }, G1.w, G0.w, f3{ G2.x, G2.y, G2.z }, f3{ G0.x, G0.y, G0.z }, 1.0f, direct);
    out[idx] = f4a{ direct.x, direct.y, direct.z, 0.0f };
}
SECTION(tone_map)       // :89-92: x / (x + 1), pow(., 1 / 2.2), three channels
{
    IDX;
    const f4a d = in[idx], a = in[idx + 1000000u], c = in[idx + 2000000u];
    const v2f d2{ d.x, d.y };
    const v2f tm = pow_inv_gamma2(d2 * rcp2(d2 + 1.0f)) + v2f{ a.x, a.y };
    const float z = pow_inv_gamma(divf(d.z, d.z + 1.0f)) + a.z;
    out[idx] = f4a{ tm.x, tm.y, z, c.x };
}
SECTION(pack)           // RGBA8 quantisation of the result
{
    IDX;
    const f4a a = in[idx], b = in[idx + 1000000u], c = in[idx + 2000000u];
    out[idx] = f4a{ u2f(pack_rgba8(f4{ a.x, a.y, a.z, a.w })), b.x, c.x, 0.0f };
}
