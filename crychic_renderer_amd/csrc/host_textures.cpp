// host_textures.cpp -- material-texture loader: the subset of Common/DDSTextureLoader.cpp the reference's
// six material textures need (CRYCHIC::LoadTextures, CRYCHIC.cpp:939-973): DDS containers holding DXT1 (BC1), DXT5 (BC3)
// or uncompressed 32-bit masks, decoded on the host to the R8G8B8A8 mip-0 image the G-buffer pass samples.
// Block decompression follows the D3D10 BC1/BC3 definition with 5:6:5 -> 8:8:8 bit replication and round-to-nearest
// integer interpolation ((2a + b + 1) / 3, (wa*a + wb*b + 3) / 7); D3D leaves the low bits of these to the hardware.
#include <cstdint>
#include <cstring>
#include <fstream>
#include <vector>
#include "crychic_hip.h"

namespace {

struct DdsInfo {
    uint32_t width = 0, height = 0;
    enum Kind { BC1, BC3, RGBA_MASKS } kind = BC1;
    uint32_t bitCount = 0, rMask = 0, gMask = 0, bMask = 0, aMask = 0;
    size_t dataOffset = 128;
    bool cube = false;             // six faces (+X, -X, +Y, -Y, +Z, -Z), each with its own mip chain, one after the other
    uint32_t fileLevels = 1;       // levels the file stores per image (dwMipMapCount when DDSD_MIPMAPCOUNT is set)
};

uint32_t rd32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

bool set_dxgi_format(uint32_t fmt, DdsInfo& d)
{
    switch (fmt) {
    case 71: case 72: d.kind = DdsInfo::BC1; return true;                    // BC1_UNORM(_SRGB)
    case 77: case 78: d.kind = DdsInfo::BC3; return true;                    // BC3_UNORM(_SRGB)
    case 28: case 29: d.kind = DdsInfo::RGBA_MASKS; d.bitCount = 32;         // R8G8B8A8_UNORM(_SRGB)
        d.rMask = 0xFFu; d.gMask = 0xFF00u; d.bMask = 0xFF0000u; d.aMask = 0xFF000000u; return true;
    case 87: case 91: d.kind = DdsInfo::RGBA_MASKS; d.bitCount = 32;         // B8G8R8A8_UNORM(_SRGB)
        d.rMask = 0xFF0000u; d.gMask = 0xFF00u; d.bMask = 0xFFu; d.aMask = 0xFF000000u; return true;
    case 88: case 93: d.kind = DdsInfo::RGBA_MASKS; d.bitCount = 32;         // B8G8R8X8_UNORM(_SRGB)
        d.rMask = 0xFF0000u; d.gMask = 0xFF00u; d.bMask = 0xFFu; d.aMask = 0u; return true;
    default: return false;
    }
}

// DDS_HEADER (Common/DDSTextureLoader.cpp:94-110) and, behind the FourCC "DX10", DDS_HEADER_DXT10 (:112-119, read at :1355-1400)
bool parse_header(const std::vector<uint8_t>& f, DdsInfo& d)
{
    if (f.size() < 128 || std::memcmp(f.data(), "DDS ", 4) != 0 || rd32(&f[4]) != 124) return false;
    d.height = rd32(&f[12]);
    d.width = rd32(&f[16]);
    d.fileLevels = (rd32(&f[8]) & 0x20000u) ? rd32(&f[28]) : 1u;          // DDSD_MIPMAPCOUNT
    if (d.fileLevels == 0) d.fileLevels = 1;
    const uint32_t pfFlags = rd32(&f[80]), caps2 = rd32(&f[112]);
    if (caps2 & 0x200u) {                       // DDSCAPS2_CUBEMAP: all six faces or nothing (DDSTextureLoader.cpp:1774-1779)
        if ((caps2 & 0xFC00u) != 0xFC00u) return false;
        d.cube = true;
    }
    if (pfFlags & 0x4u) {                       // DDPF_FOURCC
        if (std::memcmp(&f[84], "DXT1", 4) == 0) d.kind = DdsInfo::BC1;
        else if (std::memcmp(&f[84], "DXT5", 4) == 0) d.kind = DdsInfo::BC3;
        else if (std::memcmp(&f[84], "DX10", 4) == 0) {
            if (f.size() < 148) return false;
            if (!set_dxgi_format(rd32(&f[128]), d)) return false;
            if (rd32(&f[132]) != 3u) return false;                       // D3D10_RESOURCE_DIMENSION_TEXTURE2D
            if (rd32(&f[136]) & 0x4u) d.cube = true;                      // DDS_RESOURCE_MISC_TEXTURECUBE (:1729-1732)
            if (rd32(&f[140]) != 1u) return false;                        // arrays (of textures or of cubes) are not used by the reference
            d.dataOffset = 148;
        }
        else return false;                      // the other FourCCs are not used by the reference's textures
    } else if (pfFlags & 0x40u) {               // DDPF_RGB
        d.kind = DdsInfo::RGBA_MASKS;
        d.bitCount = rd32(&f[88]);
        d.rMask = rd32(&f[92]); d.gMask = rd32(&f[96]); d.bMask = rd32(&f[100]); d.aMask = (pfFlags & 0x1u) ? rd32(&f[104]) : 0;
        if (d.bitCount != 32) return false;
    } else {
        return false;
    }
    return d.width > 0 && d.height > 0;
}

inline void expand565(uint16_t c, int rgb[3])
{
    const int r = (c >> 11) & 31, g = (c >> 5) & 63, b = c & 31;
    rgb[0] = (r << 3) | (r >> 2); rgb[1] = (g << 2) | (g >> 4); rgb[2] = (b << 3) | (b >> 2);
}

// 8-byte colour block -> 16 RGBA texels (alpha 255, or 0 for BC1's transparent index)
void decode_color_block(const uint8_t* blk, bool bc1, uint8_t out[16][4])
{
    const uint16_t c0 = (uint16_t)(blk[0] | (blk[1] << 8)), c1 = (uint16_t)(blk[2] | (blk[3] << 8));
    int pal[4][4];
    expand565(c0, pal[0]); expand565(c1, pal[1]);
    pal[0][3] = pal[1][3] = 255;
    if (!bc1 || c0 > c1) {
        for (int k = 0; k < 3; ++k) { pal[2][k] = (2 * pal[0][k] + pal[1][k] + 1) / 3; pal[3][k] = (pal[0][k] + 2 * pal[1][k] + 1) / 3; }
        pal[2][3] = pal[3][3] = 255;
    } else {
        for (int k = 0; k < 3; ++k) { pal[2][k] = (pal[0][k] + pal[1][k] + 1) / 2; pal[3][k] = 0; }
        pal[2][3] = 255; pal[3][3] = 0;
    }
    const uint32_t bits = rd32(blk + 4);
    for (int t = 0; t < 16; ++t) {
        const int* p = pal[(bits >> (2 * t)) & 3u];
        for (int k = 0; k < 4; ++k) out[t][k] = (uint8_t)p[k];
    }
}

void decode_alpha_block(const uint8_t* blk, uint8_t out[16])
{
    const int a0 = blk[0], a1 = blk[1];
    int pal[8] = { a0, a1 };
    if (a0 > a1) for (int k = 1; k < 7; ++k) pal[k + 1] = ((7 - k) * a0 + k * a1 + 3) / 7;
    else { for (int k = 1; k < 5; ++k) pal[k + 1] = ((5 - k) * a0 + k * a1 + 2) / 5; pal[6] = 0; pal[7] = 255; }
    uint64_t bits = 0;
    for (int k = 0; k < 6; ++k) bits |= (uint64_t)blk[2 + k] << (8 * k);
    for (int t = 0; t < 16; ++t) out[t] = (uint8_t)pal[(bits >> (3 * t)) & 7u];
}

int channel_shift(uint32_t mask) { int s = 0; while (mask && !(mask & 1u)) { mask >>= 1; ++s; } return s; }

}  // namespace

namespace {

// Bytes one level of `w` x `h` texels occupies in the file.
size_t level_file_bytes(const DdsInfo& d, uint32_t w, uint32_t h)
{
    if (d.kind == DdsInfo::RGBA_MASKS) return (size_t)w * h * 4;
    return (size_t)((w + 3) / 4) * ((h + 3) / 4) * (d.kind == DdsInfo::BC1 ? 8 : 16);
}

// Decodes one level (w x h texels at `src`) to tightly packed RGBA8.
void decode_level(const DdsInfo& d, const uint8_t* src, uint32_t w, uint32_t h, uint8_t* rgba8)
{
    if (d.kind == DdsInfo::RGBA_MASKS) {
        const int rs = channel_shift(d.rMask), gs = channel_shift(d.gMask), bs = channel_shift(d.bMask), as = channel_shift(d.aMask);
        for (size_t i = 0; i < (size_t)w * h; ++i) {
            const uint32_t px = rd32(src + 4 * i);
            rgba8[4 * i + 0] = (uint8_t)((px & d.rMask) >> rs);
            rgba8[4 * i + 1] = (uint8_t)((px & d.gMask) >> gs);
            rgba8[4 * i + 2] = (uint8_t)((px & d.bMask) >> bs);
            rgba8[4 * i + 3] = d.aMask ? (uint8_t)((px & d.aMask) >> as) : 255;
        }
        return;
    }
    const uint32_t bw = (w + 3) / 4, bh = (h + 3) / 4;
    const size_t blockBytes = d.kind == DdsInfo::BC1 ? 8 : 16;
    for (uint32_t by = 0; by < bh; ++by)
        for (uint32_t bx = 0; bx < bw; ++bx) {
            const uint8_t* blk = src + ((size_t)by * bw + bx) * blockBytes;
            uint8_t texels[16][4], alpha[16];
            if (d.kind == DdsInfo::BC3) { decode_alpha_block(blk, alpha); decode_color_block(blk + 8, false, texels); }
            else decode_color_block(blk, true, texels);
            for (int t = 0; t < 16; ++t) {
                const uint32_t x = bx * 4 + (uint32_t)(t & 3), y = by * 4 + (uint32_t)(t >> 2);
                if (x >= w || y >= h) continue;
                uint8_t* o = rgba8 + ((size_t)y * w + x) * 4;
                o[0] = texels[t][0]; o[1] = texels[t][1]; o[2] = texels[t][2];
                o[3] = d.kind == DdsInfo::BC3 ? alpha[t] : texels[t][3];
            }
        }
}

// maxLevels == 1: level 0 only.  Otherwise every level the header announces (DDSD_MIPMAPCOUNT, dwMipMapCount).
int load_dds(const char* path, uint8_t* rgba8, size_t capacityBytes, uint32_t* width, uint32_t* height, uint32_t* mipLevels, bool wantMips)
{
    if (!path) return CRYCHIC_E_INVALID_ARG;
    std::ifstream in(path, std::ios::binary);
    if (!in) return CRYCHIC_E_INVALID_ARG;
    std::vector<uint8_t> f((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    DdsInfo d;
    if (!parse_header(f, d)) return CRYCHIC_E_UNSUPPORTED;
    // D3D12's own limit (D3D12_REQ_TEXTURE2D_U_OR_V_DIMENSION): also keeps every size computed below far from wrapping
    if (d.width == 0 || d.height == 0 || d.width > 16384u || d.height > 16384u) return CRYCHIC_E_UNSUPPORTED;
    if (d.cube) return CRYCHIC_E_UNSUPPORTED;                   // a cube map: crychic_load_dds_cube_rgba8
    uint32_t levels = 1;
    if (wantMips) {
        levels = d.fileLevels;
        uint32_t full = 1;
        for (uint32_t m = d.width > d.height ? d.width : d.height; m > 1; m >>= 1) ++full;
        if (levels > full) return CRYCHIC_E_UNSUPPORTED;        // more levels than a 1 x 1 tail allows: not a file this loader trusts
    }
    if (width) *width = d.width;
    if (height) *height = d.height;
    if (mipLevels) *mipLevels = levels;
    if (!rgba8) return 0;
    size_t need = 0, fileNeed = 0;
    { uint32_t w = d.width, h = d.height;
      for (uint32_t k = 0; k < levels; ++k) { need += (size_t)w * h * 4; fileNeed += level_file_bytes(d, w, h); w = w > 1 ? w >> 1 : 1; h = h > 1 ? h >> 1 : 1; } }
    if (capacityBytes < need) return CRYCHIC_E_INVALID_ARG;
    if (f.size() - d.dataOffset < fileNeed) return CRYCHIC_E_INVALID_ARG;      // truncated payload
    const uint8_t* src = f.data() + d.dataOffset;
    uint32_t w = d.width, h = d.height;
    for (uint32_t k = 0; k < levels; ++k) {
        decode_level(d, src, w, h, rgba8);
        src += level_file_bytes(d, w, h);
        rgba8 += (size_t)w * h * 4;
        w = w > 1 ? w >> 1 : 1; h = h > 1 ? h >> 1 : 1;
    }
    return 0;
}

// The sky cube map: the six faces in the file's (= D3D's) order.  wantMips false: level 0 of each face, stacked.  Otherwise every
// level the file stores, re-ordered level after level (each level six faces) -- the layout light_core.hpp's cube_trilinear reads.
int load_dds_cube(const char* path, uint8_t* rgba8, size_t capacityBytes, uint32_t* dim, uint32_t* mipLevels, bool wantMips)
{
    if (!path) return CRYCHIC_E_INVALID_ARG;
    std::ifstream in(path, std::ios::binary);
    if (!in) return CRYCHIC_E_INVALID_ARG;
    std::vector<uint8_t> f((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    DdsInfo d;
    if (!parse_header(f, d) || !d.cube) return CRYCHIC_E_UNSUPPORTED;
    if (d.width != d.height || d.width == 0 || d.width > 16384u) return CRYCHIC_E_UNSUPPORTED;
    uint32_t full = 1;
    for (uint32_t m = d.width; m > 1; m >>= 1) ++full;
    if (d.fileLevels > full) return CRYCHIC_E_UNSUPPORTED;
    const uint32_t outLevels = wantMips ? d.fileLevels : 1u;
    if (dim) *dim = d.width;
    if (mipLevels) *mipLevels = outLevels;
    if (!rgba8) return 0;
    size_t faceFile = 0, need = 0;                                // a face's whole chain in the file; the output's size
    { uint32_t w = d.width;
      for (uint32_t k = 0; k < d.fileLevels; ++k) {
          faceFile += level_file_bytes(d, w, w);
          if (k < outLevels) need += (size_t)6 * w * w * 4;
          w = w > 1 ? w >> 1 : 1;
      } }
    if (capacityBytes < need) return CRYCHIC_E_INVALID_ARG;
    if (f.size() - d.dataOffset < 6 * faceFile) return CRYCHIC_E_INVALID_ARG;      // truncated payload
    for (uint32_t face = 0; face < 6; ++face) {
        const uint8_t* src = f.data() + d.dataOffset + face * faceFile;
        uint8_t* levelOut = rgba8;
        uint32_t w = d.width;
        for (uint32_t k = 0; k < outLevels; ++k) {
            decode_level(d, src, w, w, levelOut + (size_t)face * w * w * 4);
            src += level_file_bytes(d, w, w);
            levelOut += (size_t)6 * w * w * 4;
            w = w > 1 ? w >> 1 : 1;
        }
    }
    return 0;
}

}  // namespace

extern "C" int crychic_load_dds_cube_rgba8(const char* path, uint8_t* rgba8, size_t capacityBytes, uint32_t* dim)
{
    if (!dim) return CRYCHIC_E_INVALID_ARG;
    return load_dds_cube(path, rgba8, capacityBytes, dim, nullptr, false);
}

extern "C" int crychic_load_dds_cube_rgba8_mips(const char* path, uint8_t* rgba8, size_t capacityBytes, uint32_t* dim, uint32_t* mipLevels)
{
    if (!dim || !mipLevels) return CRYCHIC_E_INVALID_ARG;
    return load_dds_cube(path, rgba8, capacityBytes, dim, mipLevels, true);
}

extern "C" int crychic_load_dds_rgba8(const char* path, uint8_t* rgba8, size_t capacityBytes, uint32_t* width, uint32_t* height)
{
    return load_dds(path, rgba8, capacityBytes, width, height, nullptr, false);
}

extern "C" int crychic_load_dds_rgba8_mips(const char* path, uint8_t* rgba8, size_t capacityBytes, uint32_t* width, uint32_t* height,
                                            uint32_t* mipLevels)
{
    if (!mipLevels) return CRYCHIC_E_INVALID_ARG;
    return load_dds(path, rgba8, capacityBytes, width, height, mipLevels, true);
}

// Present stand-in: the reference hands the back buffer to the swap chain (CRYCHIC.cpp:294-297); a
// headless build writes it out instead.  Binary PPM (P6), alpha dropped.
extern "C" int crychic_save_ppm(const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height)
{
    if (!path || !rgba8 || !width || !height) return CRYCHIC_E_INVALID_ARG;
    std::ofstream out(path, std::ios::binary);
    if (!out) return CRYCHIC_E_INVALID_ARG;
    out << "P6\n" << width << " " << height << "\n255\n";
    std::vector<uint8_t> row((size_t)width * 3);
    for (uint32_t y = 0; y < height; ++y) {
        for (uint32_t x = 0; x < width; ++x)
            for (int c = 0; c < 3; ++c) row[(size_t)x * 3 + c] = rgba8[((size_t)y * width + x) * 4 + c];
        out.write(reinterpret_cast<const char*>(row.data()), (std::streamsize)row.size());
    }
    return out ? 0 : CRYCHIC_E_INVALID_ARG;
}
