// host_geometry.cpp -- host-side mesh producers for the producer passes: GeometryGenerator::CreateBox / CreateGrid
// (Common/GeometryGenerator.cpp:10-101, 214-305, 551-614) and the "pos normal / triangle list" text loader of
// CRYCHIC::BuildSkullGeometry (CRYCHIC.cpp:1447-1557, format: Models/skull.txt:1-4).
#include <cmath>
#include <fstream>
#include <string>
#include <vector>
#include "crychic_hip.h"

namespace {

using V = crychic_vertex;

V make_vertex(float px, float py, float pz, float nx, float ny, float nz, float tx, float ty, float tz, float u, float v)
{
    V r;
    r.Pos[0] = px; r.Pos[1] = py; r.Pos[2] = pz;
    r.Normal[0] = nx; r.Normal[1] = ny; r.Normal[2] = nz;
    r.TangentU[0] = tx; r.TangentU[1] = ty; r.TangentU[2] = tz;
    r.TexC[0] = u; r.TexC[1] = v;
    return r;
}
void normalize3(const float in[3], float out[3])
{
    const float len = std::sqrt((in[0] * in[0] + in[1] * in[1]) + in[2] * in[2]);
    for (int c = 0; c < 3; ++c) out[c] = in[c] / len;
}
// GeometryGenerator::MidPoint (:277-305): positions/texcoords averaged, normal/tangent averaged then renormalised
V mid_point(const V& a, const V& b)
{
    V m;
    float n[3], t[3];
    for (int c = 0; c < 3; ++c) {
        m.Pos[c] = 0.5f * (a.Pos[c] + b.Pos[c]);
        n[c] = 0.5f * (a.Normal[c] + b.Normal[c]);
        t[c] = 0.5f * (a.TangentU[c] + b.TangentU[c]);
    }
    normalize3(n, m.Normal);
    normalize3(t, m.TangentU);
    m.TexC[0] = 0.5f * (a.TexC[0] + b.TexC[0]);
    m.TexC[1] = 0.5f * (a.TexC[1] + b.TexC[1]);
    return m;
}
// GeometryGenerator::Subdivide (:214-275): every triangle becomes four, vertices are not shared
void subdivide(std::vector<V>& verts, std::vector<uint32_t>& idx)
{
    const std::vector<V> inV = verts;
    const std::vector<uint32_t> inI = idx;
    verts.clear(); idx.clear();
    const uint32_t numTris = (uint32_t)inI.size() / 3;
    verts.reserve((size_t)numTris * 6); idx.reserve((size_t)numTris * 12);
    for (uint32_t i = 0; i < numTris; ++i) {
        const V v0 = inV[inI[i * 3 + 0]], v1 = inV[inI[i * 3 + 1]], v2 = inV[inI[i * 3 + 2]];
        const V m0 = mid_point(v0, v1), m1 = mid_point(v1, v2), m2 = mid_point(v0, v2);
        for (const V& v : { v0, v1, v2, m0, m1, m2 }) verts.push_back(v);
        for (uint32_t k : { 0u, 3u, 5u, 3u, 4u, 5u, 5u, 4u, 2u, 3u, 1u, 4u }) idx.push_back(i * 6 + k);
    }
}
int deliver(const std::vector<V>& verts, const std::vector<uint32_t>& idx, V* vout, uint32_t vcap, uint32_t* iout, uint32_t icap, uint32_t* nIdx)
{
    if (nIdx) *nIdx = (uint32_t)idx.size();
    if (vout && iout) {
        if (verts.size() > vcap || idx.size() > icap) return CRYCHIC_E_INVALID_ARG;
        for (size_t i = 0; i < verts.size(); ++i) vout[i] = verts[i];
        for (size_t i = 0; i < idx.size(); ++i) iout[i] = idx[i];
    }
    return (int)verts.size();
}

}  // namespace

extern "C" {

int crychic_create_box(float width, float height, float depth, uint32_t numSubdivisions, crychic_vertex* vout, uint32_t vcap,
                       uint32_t* iout, uint32_t icap, uint32_t* nIdx)
{
    const float w2 = 0.5f * width, h2 = 0.5f * height, d2 = 0.5f * depth;
    // six faces: (normal, tangent, four corners in the order of GeometryGenerator.cpp:24-57 with their texcoords)
    std::vector<V> verts = {
        make_vertex(-w2, -h2, -d2, 0, 0, -1, 1, 0, 0, 0, 1), make_vertex(-w2, +h2, -d2, 0, 0, -1, 1, 0, 0, 0, 0),
        make_vertex(+w2, +h2, -d2, 0, 0, -1, 1, 0, 0, 1, 0), make_vertex(+w2, -h2, -d2, 0, 0, -1, 1, 0, 0, 1, 1),          // front
        make_vertex(-w2, -h2, +d2, 0, 0, 1, -1, 0, 0, 1, 1), make_vertex(+w2, -h2, +d2, 0, 0, 1, -1, 0, 0, 0, 1),
        make_vertex(+w2, +h2, +d2, 0, 0, 1, -1, 0, 0, 0, 0), make_vertex(-w2, +h2, +d2, 0, 0, 1, -1, 0, 0, 1, 0),          // back
        make_vertex(-w2, +h2, -d2, 0, 1, 0, 1, 0, 0, 0, 1), make_vertex(-w2, +h2, +d2, 0, 1, 0, 1, 0, 0, 0, 0),
        make_vertex(+w2, +h2, +d2, 0, 1, 0, 1, 0, 0, 1, 0), make_vertex(+w2, +h2, -d2, 0, 1, 0, 1, 0, 0, 1, 1),            // top
        make_vertex(-w2, -h2, -d2, 0, -1, 0, -1, 0, 0, 1, 1), make_vertex(+w2, -h2, -d2, 0, -1, 0, -1, 0, 0, 0, 1),
        make_vertex(+w2, -h2, +d2, 0, -1, 0, -1, 0, 0, 0, 0), make_vertex(-w2, -h2, +d2, 0, -1, 0, -1, 0, 0, 1, 0),        // bottom
        make_vertex(-w2, -h2, +d2, -1, 0, 0, 0, 0, -1, 0, 1), make_vertex(-w2, +h2, +d2, -1, 0, 0, 0, 0, -1, 0, 0),
        make_vertex(-w2, +h2, -d2, -1, 0, 0, 0, 0, -1, 1, 0), make_vertex(-w2, -h2, -d2, -1, 0, 0, 0, 0, -1, 1, 1),        // left
        make_vertex(+w2, -h2, -d2, 1, 0, 0, 0, 0, 1, 0, 1), make_vertex(+w2, +h2, -d2, 1, 0, 0, 0, 0, 1, 0, 0),
        make_vertex(+w2, +h2, +d2, 1, 0, 0, 0, 0, 1, 1, 0), make_vertex(+w2, -h2, +d2, 1, 0, 0, 0, 0, 1, 1, 1),            // right
    };
    std::vector<uint32_t> idx;
    for (uint32_t f = 0; f < 6; ++f)
        for (uint32_t k : { 0u, 1u, 2u, 0u, 2u, 3u }) idx.push_back(4 * f + k);      // :67-91
    if (numSubdivisions > 6u) numSubdivisions = 6u;                                   // :95
    for (uint32_t s = 0; s < numSubdivisions; ++s) subdivide(verts, idx);
    return deliver(verts, idx, vout, vcap, iout, icap, nIdx);
}

int crychic_create_grid(float width, float depth, uint32_t m, uint32_t n, crychic_vertex* vout, uint32_t vcap, uint32_t* iout,
                        uint32_t icap, uint32_t* nIdx)
{
    if (m < 2 || n < 2) return CRYCHIC_E_INVALID_ARG;
    std::vector<V> verts((size_t)m * n);
    const float halfWidth = 0.5f * width, halfDepth = 0.5f * depth;
    const float dx = width / (n - 1), dz = depth / (m - 1), du = 1.0f / (n - 1), dv = 1.0f / (m - 1);
    for (uint32_t i = 0; i < m; ++i) {
        const float z = halfDepth - i * dz;
        for (uint32_t j = 0; j < n; ++j)
            verts[(size_t)i * n + j] = make_vertex(-halfWidth + j * dx, 0.0f, z, 0, 1, 0, 1, 0, 0, j * du, i * dv);
    }
    std::vector<uint32_t> idx;
    idx.reserve((size_t)(m - 1) * (n - 1) * 6);
    for (uint32_t i = 0; i + 1 < m; ++i)
        for (uint32_t j = 0; j + 1 < n; ++j)
            for (uint32_t k : { i * n + j, i * n + j + 1, (i + 1) * n + j, (i + 1) * n + j, i * n + j + 1, (i + 1) * n + j + 1 }) idx.push_back(k);
    return deliver(verts, idx, vout, vcap, iout, icap, nIdx);
}

int crychic_load_mesh_text(const char* path, crychic_vertex* vout, uint32_t vcap, uint32_t* iout, uint32_t icap, uint32_t* nVerts,
                           uint32_t* nIdx)
{
    if (!path) return CRYCHIC_E_INVALID_ARG;
    std::ifstream fin(path);
    if (!fin) return CRYCHIC_E_INVALID_ARG;
    uint32_t vcount = 0, tcount = 0;
    std::string ignore;
    fin >> ignore >> vcount;                              // "VertexCount: N"
    fin >> ignore >> tcount;                              // "TriangleCount: M"
    fin >> ignore >> ignore >> ignore >> ignore;          // "VertexList (pos, normal) {"
    if (!fin) return CRYCHIC_E_INVALID_ARG;
    if (nVerts) *nVerts = vcount;
    if (nIdx) *nIdx = 3 * tcount;
    if (!vout || !iout) return (int)vcount;
    if (vcount > vcap || 3 * tcount > icap) return CRYCHIC_E_INVALID_ARG;
    for (uint32_t i = 0; i < vcount; ++i) {
        V& p = vout[i];
        fin >> p.Pos[0] >> p.Pos[1] >> p.Pos[2] >> p.Normal[0] >> p.Normal[1] >> p.Normal[2];
        p.TexC[0] = p.TexC[1] = 0.0f;
        // any tangent orthogonal to the normal (CRYCHIC.cpp:1484-1499): up x N, or N x (0,0,1) when N is (anti)parallel to up
        const float* N = p.Normal;
        float c[3];
        const float ndotup = (N[0] * 0.0f + N[1] * 1.0f) + N[2] * 0.0f;
        if (std::fabs(ndotup) < 1.0f - 0.001f) { c[0] = 1.0f * N[2] - 0.0f * N[1]; c[1] = 0.0f * N[0] - 0.0f * N[2]; c[2] = 0.0f * N[1] - 1.0f * N[0]; }
        else { c[0] = N[1] * 1.0f - N[2] * 0.0f; c[1] = N[2] * 0.0f - N[0] * 1.0f; c[2] = N[0] * 0.0f - N[1] * 0.0f; }
        normalize3(c, p.TangentU);
    }
    fin >> ignore >> ignore >> ignore;                    // "} TriangleList {"
    for (uint32_t i = 0; i < 3 * tcount; ++i) fin >> iout[i];
    if (!fin) return CRYCHIC_E_INVALID_ARG;
    return (int)vcount;
}

}  // extern "C"

static_assert(sizeof(crychic_vertex) == 44 && sizeof(crychic_instance_data) == 144 && sizeof(crychic_material_data) == 112,
              "structured-buffer ABI (FrameResource.h:7-27,69-75)");
