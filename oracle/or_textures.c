/*
 * or_textures.c -- oracle restatement of the material-texture path: DDS (DXT1 / DXT5 / 32-bit masks) ->
 * R8G8B8A8 mip 0, for the files CRYCHIC::LoadTextures opens (CRYCHIC.cpp:939-973).  TEST INFRASTRUCTURE, parity unpinned:
 * the reference decodes block-compressed textures in GPU hardware; the interpolation rounding used here
 * (bit-replicated 5:6:5, (2a+b+1)/3, (wa*a+wb*b+3)/7) is the oracle's definition.
 */
#include "crychic_oracle.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint32_t le32(const unsigned char* p) { return p[0] | (p[1] << 8) | (p[2] << 16) | ((uint32_t)p[3] << 24); }

static void color565(unsigned c, int* r, int* g, int* b)
{
    int r5 = (c >> 11) & 31, g6 = (c >> 5) & 63, b5 = c & 31;
    *r = (r5 << 3) | (r5 >> 2); *g = (g6 << 2) | (g6 >> 4); *b = (b5 << 3) | (b5 >> 2);
}

static void bc_color(const unsigned char* blk, int is_bc1, int x, int y, unsigned char out[4])
{
    unsigned c0 = blk[0] | (blk[1] << 8), c1 = blk[2] | (blk[3] << 8);
    unsigned sel = (le32(blk + 4) >> (2 * (4 * y + x))) & 3u;
    int r0, g0, b0, r1, g1, b1;
    color565(c0, &r0, &g0, &b0); color565(c1, &r1, &g1, &b1);
    int four = !is_bc1 || c0 > c1;
    int r, g, b, a = 255;
    switch (sel) {
    case 0: r = r0; g = g0; b = b0; break;
    case 1: r = r1; g = g1; b = b1; break;
    case 2:
        if (four) { r = (2 * r0 + r1 + 1) / 3; g = (2 * g0 + g1 + 1) / 3; b = (2 * b0 + b1 + 1) / 3; }
        else { r = (r0 + r1 + 1) / 2; g = (g0 + g1 + 1) / 2; b = (b0 + b1 + 1) / 2; }
        break;
    default:
        if (four) { r = (r0 + 2 * r1 + 1) / 3; g = (g0 + 2 * g1 + 1) / 3; b = (b0 + 2 * b1 + 1) / 3; }
        else { r = g = b = 0; a = 0; }
    }
    out[0] = (unsigned char)r; out[1] = (unsigned char)g; out[2] = (unsigned char)b; out[3] = (unsigned char)a;
}

static unsigned char bc3_alpha(const unsigned char* blk, int x, int y)
{
    int a0 = blk[0], a1 = blk[1];
    unsigned long long bits = 0;
    for (int k = 0; k < 6; ++k) bits |= (unsigned long long)blk[2 + k] << (8 * k);
    unsigned sel = (unsigned)((bits >> (3 * (4 * y + x))) & 7u);
    if (sel == 0) return (unsigned char)a0;
    if (sel == 1) return (unsigned char)a1;
    if (a0 > a1) return (unsigned char)(((8 - sel) * a0 + (sel - 1) * a1 + 3) / 7);
    if (sel == 6) return 0;
    if (sel == 7) return 255;
    return (unsigned char)(((6 - sel) * a0 + (sel - 1) * a1 + 2) / 5);
}

/* One level of w x h texels starting at `src` (at most `avail` bytes): returns the bytes consumed, 0 if the level does not fit.
 * kind: 1 = DXT1, 5 = DXT5, 0 = 32-bit masks m[0..3] (0 = channel absent: 255). */
static size_t decode_level(int kind, const uint32_t m[4], const unsigned char* src, size_t avail, uint32_t w, uint32_t h, uint8_t* rgba8)
{
    if (kind == 0) {
        size_t need = (size_t)w * h * 4;
        if (avail < need) return 0;
        for (size_t i = 0; i < (size_t)w * h; ++i) {
            uint32_t px = le32(src + 4 * i);
            for (int c = 0; c < 4; ++c) {
                if (!m[c]) { rgba8[4 * i + c] = 255; continue; }
                uint32_t v = px & m[c], mm = m[c];
                while (!(mm & 1u)) { mm >>= 1; v >>= 1; }
                rgba8[4 * i + c] = (uint8_t)v;
            }
        }
        return need;
    }
    int bc1 = kind == 1;
    size_t bsz = bc1 ? 8 : 16;
    uint32_t bw = (w + 3) / 4, bh = (h + 3) / 4;
    size_t need = (size_t)bw * bh * bsz;
    if (avail < need) return 0;
    for (uint32_t y = 0; y < h; ++y)
        for (uint32_t x = 0; x < w; ++x) {
            const unsigned char* blk = src + ((size_t)(y / 4) * bw + x / 4) * bsz;
            unsigned char* o = rgba8 + ((size_t)y * w + x) * 4;
            bc_color(bc1 ? blk : blk + 8, bc1, (int)(x & 3), (int)(y & 3), o);
            if (!bc1) o[3] = bc3_alpha(blk, (int)(x & 3), (int)(y & 3));
        }
    return need;
}

/* Pixel format of a DDS file whose magic and header size have been checked: 1 = DXT1, 5 = DXT5, 0 = 32-bit masks in m[], -1 = not
 * one of the formats the path uses.  *off = where the pixel data starts (behind the DX10 header when there is one), *cube = the
 * file is a cube map with all six faces (Common/DDSTextureLoader.cpp:1729-1732, 1774-1779). */
static int pixel_kind(const unsigned char* d, size_t n, uint32_t m[4], size_t* off, int* cube)
{
    uint32_t pf = le32(d + 80), caps2 = le32(d + 112);
    *off = 128;
    *cube = 0;
    if (caps2 & 0x200u) {
        if ((caps2 & 0xFC00u) != 0xFC00u) return -1;
        *cube = 1;
    }
    if ((pf & 4u) && memcmp(d + 84, "DXT1", 4) == 0) return 1;
    if ((pf & 4u) && memcmp(d + 84, "DXT5", 4) == 0) return 5;
    if ((pf & 4u) && memcmp(d + 84, "DX10", 4) == 0) {
        if (n < 148 || le32(d + 132) != 3u || le32(d + 140) != 1u) return -1;
        if (le32(d + 136) & 4u) *cube = 1;
        *off = 148;
        switch (le32(d + 128)) {
        case 71: case 72: return 1;
        case 77: case 78: return 5;
        case 28: case 29: m[0] = 0xFFu; m[1] = 0xFF00u; m[2] = 0xFF0000u; m[3] = 0xFF000000u; return 0;
        case 87: case 91: m[0] = 0xFF0000u; m[1] = 0xFF00u; m[2] = 0xFFu; m[3] = 0xFF000000u; return 0;
        case 88: case 93: m[0] = 0xFF0000u; m[1] = 0xFF00u; m[2] = 0xFFu; m[3] = 0u; return 0;
        default: return -1;
        }
    }
    if (!(pf & 4u) && (pf & 0x40u) && le32(d + 88) == 32) {
        m[0] = le32(d + 92); m[1] = le32(d + 96); m[2] = le32(d + 100); m[3] = (pf & 1u) ? le32(d + 104) : 0;
        return 0;
    }
    return -1;
}

/* want_mips = 0: level 0 only.  Levels are written back to back, level k = max(1, w >> k) x max(1, h >> k)
 * (Common/DDSTextureLoader.cpp uploads the levels the file stores; it generates none). */
static int load_dds(const char* path, uint8_t* rgba8, size_t capacity, uint32_t* width, uint32_t* height, uint32_t* mips, int want_mips)
{
    FILE* f = fopen(path, "rb");
    if (!f) return -1;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char* d = (unsigned char*)malloc((size_t)(n > 0 ? n : 1));
    if (fread(d, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(d); return -1; }
    fclose(f);
    int rc = -4;
    if (n >= 128 && memcmp(d, "DDS ", 4) == 0 && le32(d + 4) == 124) {
        uint32_t h = le32(d + 12), w = le32(d + 16);
        uint32_t m[4] = { 0, 0, 0, 0 };
        size_t off = 128;
        int cube = 0;
        int kind = pixel_kind(d, (size_t)n, m, &off, &cube);
        if (cube) kind = -1;      /* a cube map: or_load_dds_cube_rgba8 */
        uint32_t levels = 1;
        if (want_mips && (le32(d + 8) & 0x20000u)) {
            levels = le32(d + 28);
            if (levels == 0) levels = 1;
            uint32_t full = 1;
            for (uint32_t mm = w > h ? w : h; mm > 1; mm >>= 1) ++full;
            if (levels > full) kind = -1;
        }
        if (kind >= 0 && w > 0 && h > 0) {
            if (width) *width = w;
            if (height) *height = h;
            if (mips) *mips = levels;
            size_t need = 0;
            { uint32_t lw = w, lh = h; for (uint32_t k = 0; k < levels; ++k) { need += (size_t)lw * lh * 4; lw = lw > 1 ? lw >> 1 : 1; lh = lh > 1 ? lh >> 1 : 1; } }
            if (!rgba8) rc = 0;
            else if (capacity < need) rc = -1;
            else {
                const unsigned char* src = d + off;
                size_t avail = (size_t)n - off;
                uint32_t lw = w, lh = h;
                rc = 0;
                /* refuse a truncated file before writing anything */
                { const unsigned char* s2 = src; size_t a2 = avail; uint32_t w2 = lw, h2 = lh;
                  for (uint32_t k = 0; k < levels; ++k) {
                      size_t b = kind == 0 ? (size_t)w2 * h2 * 4 : (size_t)((w2 + 3) / 4) * ((h2 + 3) / 4) * (kind == 1 ? 8 : 16);
                      if (a2 < b) { rc = -1; break; }
                      s2 += b; a2 -= b; w2 = w2 > 1 ? w2 >> 1 : 1; h2 = h2 > 1 ? h2 >> 1 : 1;
                  } }
                for (uint32_t k = 0; k < levels && rc == 0; ++k) {
                    size_t used = decode_level(kind, m, src, avail, lw, lh, rgba8);
                    src += used; avail -= used;
                    rgba8 += (size_t)lw * lh * 4;
                    lw = lw > 1 ? lw >> 1 : 1; lh = lh > 1 ? lh >> 1 : 1;
                }
            }
        }
    }
    free(d);
    return rc;
}

int or_load_dds_rgba8(const char* path, uint8_t* rgba8, size_t capacity, uint32_t* width, uint32_t* height)
{
    return load_dds(path, rgba8, capacity, width, height, NULL, 0);
}
int or_load_dds_rgba8_mips(const char* path, uint8_t* rgba8, size_t capacity, uint32_t* width, uint32_t* height, uint32_t* mips)
{
    if (!mips) return -1;
    return load_dds(path, rgba8, capacity, width, height, mips, 1);
}

/* The sky cube map (CRYCHIC.cpp:960,968,1148-1151): the six faces, +X -X +Y -Y +Z -Z, each face followed in the file by the rest of
 * its mip chain.  want_mips == 0: level 0 of each face, stacked.  Otherwise the whole chain the file stores, re-ordered level after
 * level (each level: six faces of max(dim >> level, 1)^2 texels) -- the layout or_cube_trilinear reads. */
static int load_dds_cube(const char* path, uint8_t* rgba8, size_t capacity, uint32_t* dim, uint32_t* mips, int want_mips)
{
    FILE* f = fopen(path, "rb");
    if (!f) return -1;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char* d = (unsigned char*)malloc((size_t)(n > 0 ? n : 1));
    if (fread(d, 1, (size_t)n, f) != (size_t)n) { fclose(f); free(d); return -1; }
    fclose(f);
    int rc = -4;
    if (n >= 128 && memcmp(d, "DDS ", 4) == 0 && le32(d + 4) == 124) {
        uint32_t h = le32(d + 12), w = le32(d + 16), m[4] = { 0, 0, 0, 0 };
        uint32_t levels = (le32(d + 8) & 0x20000u) ? le32(d + 28) : 1u;
        size_t off = 128;
        int cube = 0;
        int kind = pixel_kind(d, (size_t)n, m, &off, &cube);
        uint32_t full = 1;
        for (uint32_t mm = w; mm > 1; mm >>= 1) ++full;
        if (levels == 0) levels = 1;
        if (kind >= 0 && cube && w == h && w > 0 && w <= 16384u && levels <= full) {
            uint32_t out_levels = want_mips ? levels : 1u;
            if (dim) *dim = w;
            if (mips) *mips = out_levels;
            size_t face_file = 0, need = 0;
            { uint32_t lw = w; for (uint32_t k = 0; k < levels; ++k) {
                  face_file += kind == 0 ? (size_t)lw * lw * 4 : (size_t)((lw + 3) / 4) * ((lw + 3) / 4) * (kind == 1 ? 8 : 16);
                  if (k < out_levels) need += (size_t)6 * lw * lw * 4;
                  lw = lw > 1 ? lw >> 1 : 1; } }
            if (!rgba8) rc = 0;
            else if (capacity < need || (size_t)n - off < 6 * face_file) rc = -1;
            else {
                rc = 0;
                for (uint32_t face = 0; face < 6; ++face) {
                    const unsigned char* src = d + off + face * face_file;
                    size_t avail = (size_t)n - off - face * face_file;
                    uint8_t* level_out = rgba8;
                    uint32_t lw = w;
                    for (uint32_t k = 0; k < out_levels && rc == 0; ++k) {
                        size_t used = decode_level(kind, m, src, avail, lw, lw, level_out + (size_t)face * lw * lw * 4);
                        if (!used) rc = -1;
                        src += used; avail -= used;
                        level_out += (size_t)6 * lw * lw * 4;
                        lw = lw > 1 ? lw >> 1 : 1;
                    }
                }
            }
        }
    }
    free(d);
    return rc;
}
int or_load_dds_cube_rgba8(const char* path, uint8_t* rgba8, size_t capacity, uint32_t* dim) { return load_dds_cube(path, rgba8, capacity, dim, NULL, 0); }
int or_load_dds_cube_rgba8_mips(const char* path, uint8_t* rgba8, size_t capacity, uint32_t* dim, uint32_t* mips)
{
    return load_dds_cube(path, rgba8, capacity, dim, mips, 1);
}
