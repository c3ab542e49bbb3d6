// image_io.h -- image files for the host program: the reference loads its target through prlib/stb_image
// (main.cpp:253-259) and shows image0 in a window (:794, :839-840); headless, that becomes file I/O.
// Readers return tightly packed RGB8.  Formats: .s2di (the repo's raw fixtures), binary PPM (P6), PNG
// (8-bit grey / RGB / palette / with or without alpha, non-interlaced; inflate/deflate by zlib), JPEG (read only,
// jpeg_decode.h).  Host-side
// C++ only: nothing here is on the training path.
#pragma once

#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "jpeg_decode.h"

namespace s2dio {

struct Image8 {
    int w = 0, h = 0;
    std::vector<uint8_t> rgb; // w*h*3
};

inline bool ends_with(const std::string& s, const char* suf)
{
    const size_t n = std::strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

inline bool read_file(const std::string& path, std::vector<uint8_t>* out)
{
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::fseek(f, 0, SEEK_END);
    const long n = std::ftell(f);
    std::fseek(f, 0, SEEK_SET);
    out->resize(n > 0 ? (size_t)n : 0);
    const bool ok = n >= 0 && std::fread(out->data(), 1, out->size(), f) == out->size();
    std::fclose(f);
    return ok;
}

inline bool load_s2di(const std::vector<uint8_t>& d, Image8* im)
{
    if (d.size() < 16 || std::memcmp(d.data(), "S2DI", 4) != 0) return false;
    uint32_t hdr[3];
    std::memcpy(hdr, d.data() + 4, 12);
    if (hdr[2] != 3 || d.size() != 16 + (size_t)hdr[0] * hdr[1] * 3) return false;
    im->w = (int)hdr[0];
    im->h = (int)hdr[1];
    im->rgb.assign(d.begin() + 16, d.end());
    return true;
}

inline bool load_ppm(const std::vector<uint8_t>& d, Image8* im)
{
    size_t p = 0;
    auto token = [&](std::string* t) {
        t->clear();
        while (p < d.size()) {
            if (d[p] == '#') { while (p < d.size() && d[p] != '\n') p++; }
            else if (d[p] == ' ' || d[p] == '\n' || d[p] == '\r' || d[p] == '\t') p++;
            else break;
        }
        while (p < d.size() && !(d[p] == ' ' || d[p] == '\n' || d[p] == '\r' || d[p] == '\t')) t->push_back((char)d[p++]);
        return !t->empty();
    };
    std::string t;
    if (!token(&t) || t != "P6") return false;
    int v[3];
    for (int k = 0; k < 3; k++) {
        if (!token(&t)) return false;
        v[k] = std::atoi(t.c_str());
    }
    p++; // the single whitespace byte after maxval
    if (v[0] <= 0 || v[1] <= 0 || v[2] != 255 || d.size() < p + (size_t)v[0] * v[1] * 3) return false;
    im->w = v[0];
    im->h = v[1];
    im->rgb.assign(d.begin() + p, d.begin() + p + (size_t)v[0] * v[1] * 3);
    return true;
}

inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

inline bool load_png(const std::vector<uint8_t>& d, Image8* im)
{
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    if (d.size() < 8 || std::memcmp(d.data(), sig, 8) != 0) return false;
    size_t p = 8;
    int w = 0, h = 0, depth = 0, ctype = -1, interlace = 0;
    std::vector<uint8_t> idat, plte;
    while (p + 12 <= d.size()) {
        const uint32_t len = be32(&d[p]);
        const uint8_t* type = &d[p + 4];
        if (p + 12 + (size_t)len > d.size()) return false;
        const uint8_t* body = &d[p + 8];
        uLong crc = crc32(0L, type, 4);
        if (len) crc = crc32(crc, body, len);
        if ((uint32_t)crc != be32(body + len)) return false;
        if (!std::memcmp(type, "IHDR", 4) && len == 13) {
            w = (int)be32(body); h = (int)be32(body + 4);
            depth = body[8]; ctype = body[9]; interlace = body[12];
        } else if (!std::memcmp(type, "PLTE", 4)) plte.assign(body, body + len);
        else if (!std::memcmp(type, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
        else if (!std::memcmp(type, "IEND", 4)) break;
        p += 12 + (size_t)len;
    }
    int ch = ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : ctype == 6 ? 4 : 0;
    if (w <= 0 || h <= 0 || depth != 8 || ch == 0 || interlace != 0 || (ctype == 3 && plte.size() < 3)) return false;
    const size_t stride = (size_t)w * ch;
    std::vector<uint8_t> raw((stride + 1) * (size_t)h);
    uLongf raw_len = (uLongf)raw.size();
    if (uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size()) != Z_OK || raw_len != raw.size()) return false;
    std::vector<uint8_t> px(stride * (size_t)h);
    for (int y = 0; y < h; y++) { // undo the per-scanline filters (PNG spec 9.2)
        const uint8_t ft = raw[(stride + 1) * (size_t)y];
        const uint8_t* in = &raw[(stride + 1) * (size_t)y + 1];
        uint8_t* out = &px[stride * (size_t)y];
        const uint8_t* up = y ? out - stride : nullptr;
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= (size_t)ch ? out[i - ch] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)ch) ? up[i - ch] : 0;
            int pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = b;
            else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) {
                const int pa = std::abs(b - c), pb = std::abs(a - c), pc = std::abs(a + b - 2 * c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            } else if (ft != 0) return false;
            out[i] = (uint8_t)(in[i] + pred);
        }
    }
    im->w = w;
    im->h = h;
    im->rgb.resize((size_t)w * h * 3);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        const uint8_t* s = &px[i * ch];
        uint8_t* o = &im->rgb[i * 3];
        if (ctype == 0 || ctype == 4) o[0] = o[1] = o[2] = s[0];
        else if (ctype == 3) {
            if ((size_t)s[0] * 3 + 2 >= plte.size()) return false;
            std::memcpy(o, &plte[(size_t)s[0] * 3], 3);
        } else std::memcpy(o, s, 3); // alpha, where present, is dropped: imageRef's .w is never read
    }
    return true;
}

inline bool load_image(const std::string& path, Image8* im)
{
    std::vector<uint8_t> d;
    if (!read_file(path, &d)) return false;
    if (d.size() >= 8 && d[0] == 0x89 && d[1] == 'P') return load_png(d, im);
    if (d.size() >= 4 && d[0] == 0xFF && d[1] == 0xD8) return load_jpeg(d, &im->w, &im->h, &im->rgb);
    if (d.size() >= 4 && !std::memcmp(d.data(), "S2DI", 4)) return load_s2di(d, im);
    if (d.size() >= 2 && d[0] == 'P' && d[1] == '6') return load_ppm(d, im);
    return false;
}

inline bool save_ppm(const std::string& path, const Image8& im)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    std::fprintf(f, "P6\n%d %d\n255\n", im.w, im.h);
    const bool ok = std::fwrite(im.rgb.data(), 1, im.rgb.size(), f) == im.rgb.size();
    std::fclose(f);
    return ok;
}

inline bool save_s2di(const std::string& path, const Image8& im)
{
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const uint32_t hdr[3] = {(uint32_t)im.w, (uint32_t)im.h, 3u};
    bool ok = std::fwrite("S2DI", 1, 4, f) == 4 && std::fwrite(hdr, 4, 3, f) == 3 &&
              std::fwrite(im.rgb.data(), 1, im.rgb.size(), f) == im.rgb.size();
    std::fclose(f);
    return ok;
}

inline bool save_png(const std::string& path, const Image8& im)
{
    const size_t stride = (size_t)im.w * 3;
    std::vector<uint8_t> raw((stride + 1) * (size_t)im.h);
    for (int y = 0; y < im.h; y++) {
        raw[(stride + 1) * (size_t)y] = 0; // filter type 0 (None)
        std::memcpy(&raw[(stride + 1) * (size_t)y + 1], &im.rgb[stride * (size_t)y], stride);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return false;
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    auto put32 = [](uint8_t* p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; };
    auto chunk = [&](const char* type, const uint8_t* body, uint32_t len) {
        uint8_t hdr[8], crc[4];
        put32(hdr, len);
        std::memcpy(hdr + 4, type, 4);
        uLong c = crc32(0L, (const Bytef*)type, 4);
        if (len) c = crc32(c, body, len); // crc32(c, NULL, 0) would return the initial value, not c
        put32(crc, (uint32_t)c);
        return std::fwrite(hdr, 1, 8, f) == 8 && (len == 0 || std::fwrite(body, 1, len, f) == len) && std::fwrite(crc, 1, 4, f) == 4;
    };
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    uint8_t ihdr[13];
    put32(ihdr, (uint32_t)im.w);
    put32(ihdr + 4, (uint32_t)im.h);
    ihdr[8] = 8; ihdr[9] = 2; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    const bool ok = std::fwrite(sig, 1, 8, f) == 8 && chunk("IHDR", ihdr, 13) && chunk("IDAT", z.data(), (uint32_t)zlen) &&
                    chunk("IEND", nullptr, 0);
    std::fclose(f);
    return ok;
}

inline bool save_image(const std::string& path, const Image8& im)
{
    if (ends_with(path, ".png")) return save_png(path, im);
    if (ends_with(path, ".s2di")) return save_s2di(path, im);
    return save_ppm(path, im);
}

// image0 (RGBA32F, main.cpp:310) -> RGB8, the way a texture upload would quantise it
inline Image8 quantise(const std::vector<float>& rgba, int w, int h)
{
    Image8 im;
    im.w = w;
    im.h = h;
    im.rgb.resize((size_t)w * h * 3);
    for (size_t p = 0; p < (size_t)w * h; p++)
        for (int c = 0; c < 3; c++) {
            const float v = rgba[p * 4 + c] * 255.0f + 0.5f;
            im.rgb[p * 3 + c] = (uint8_t)(v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v));
        }
    return im;
}

} // namespace s2dio
