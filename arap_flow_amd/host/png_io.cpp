// png_io.cpp -- see png_io.h
#include "png_io.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace arapio {

static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

static bool read_file(const std::string& path, std::vector<uint8_t>& buf)
{
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    buf.resize(n > 0 ? (size_t)n : 0);
    size_t got = n > 0 ? fread(buf.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    return got == buf.size();
}

static int paeth(int a, int b, int c)
{
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

bool read_png_rgb(const std::string& path, Image& out, std::string& err)
{
    std::vector<uint8_t> file;
    if (!read_file(path, file)) { err = "cannot read " + path; return false; }
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (file.size() < 8 || memcmp(file.data(), sig, 8) != 0) { err = path + ": not a PNG file"; return false; }
    size_t pos = 8;
    int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
    bool have_ihdr = false;
    while (pos + 12 <= file.size()) {
        const uint32_t len = be32(&file[pos]);
        const uint8_t* type = &file[pos + 4];
        if (pos + 12 + (size_t)len > file.size()) { err = path + ": truncated chunk"; return false; }
        const uint8_t* data = &file[pos + 8];
        if (memcmp(type, "IHDR", 4) == 0 && len >= 13) {
            w = (int)be32(data); h = (int)be32(data + 4);
            depth = data[8]; ctype = data[9]; interlace = data[12];
            have_ihdr = true;
        } else if (memcmp(type, "PLTE", 4) == 0) {
            plte.assign(data, data + len);
        } else if (memcmp(type, "IDAT", 4) == 0) {
            idat.insert(idat.end(), data, data + len);
        } else if (memcmp(type, "IEND", 4) == 0) {
            break;
        }
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || w <= 0 || h <= 0) { err = path + ": missing IHDR"; return false; }
    if (interlace) { err = path + ": interlaced PNG is not supported"; return false; }
    int channels;
    switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: err = path + ": bad colour type"; return false;
    }
    if (!(depth == 1 || depth == 2 || depth == 4 || depth == 8 || depth == 16) ||
        ((ctype == 2 || ctype == 4 || ctype == 6) && depth < 8) || (ctype == 3 && depth > 8)) {
        err = path + ": unsupported bit depth";
        return false;
    }
    const int bits = channels * depth;
    const size_t stride = ((size_t)w * bits + 7) / 8;
    const int bpp = bits >= 8 ? bits / 8 : 1;
    std::vector<uint8_t> raw((stride + 1) * (size_t)h);
    uLongf dlen = (uLongf)raw.size();
    int zr = uncompress(raw.data(), &dlen, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || dlen != raw.size()) { err = path + ": zlib inflate failed"; return false; }
    // undo the scanline filters in place
    std::vector<uint8_t> zero(stride, 0);
    for (int y = 0; y < h; ++y) {
        uint8_t* row = &raw[(stride + 1) * (size_t)y];
        const int ft = row[0];
        uint8_t* cur = row + 1;
        const uint8_t* up = y ? &raw[(stride + 1) * (size_t)(y - 1) + 1] : zero.data();
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = up[i], c = i >= (size_t)bpp ? up[i - bpp] : 0;
            int v = cur[i];
            switch (ft) {
            case 0: break;
            case 1: v += a; break;
            case 2: v += b; break;
            case 3: v += (a + b) / 2; break;
            case 4: v += paeth(a, b, c); break;
            default: err = path + ": bad filter type"; return false;
            }
            cur[i] = (uint8_t)v;
        }
    }
    out.w = w; out.h = h;
    out.rgb.assign((size_t)w * h * 3, 0);
    const int maxv = (1 << (depth > 8 ? 8 : depth)) - 1;
    for (int y = 0; y < h; ++y) {
        const uint8_t* cur = &raw[(stride + 1) * (size_t)y + 1];
        for (int x = 0; x < w; ++x) {
            auto sample = [&](int ch) -> int {            // channel value reduced to 8 bits (or the palette index)
                if (depth == 16) return cur[((size_t)x * channels + ch) * 2];
                if (depth == 8) return cur[(size_t)x * channels + ch];
                const size_t bit = (size_t)x * depth;     // only 1-channel images have depth < 8
                return (cur[bit / 8] >> (8 - depth - (bit % 8))) & maxv;
            };
            uint8_t* o = &out.rgb[((size_t)y * w + x) * 3];
            if (ctype == 3) {
                const size_t idx = (size_t)sample(0);
                if (3 * idx + 2 < plte.size()) { o[0] = plte[3 * idx]; o[1] = plte[3 * idx + 1]; o[2] = plte[3 * idx + 2]; }
            } else if (ctype == 0 || ctype == 4) {
                int v = sample(0);
                if (depth < 8) v = v * 255 / maxv;
                o[0] = o[1] = o[2] = (uint8_t)v;
            } else {
                o[0] = (uint8_t)sample(0); o[1] = (uint8_t)sample(1); o[2] = (uint8_t)sample(2);
            }
        }
    }
    return true;
}

static void put_chunk(std::vector<uint8_t>& f, const char* type, const uint8_t* data, size_t len)
{
    uint8_t hdr[8] = {(uint8_t)(len >> 24), (uint8_t)(len >> 16), (uint8_t)(len >> 8), (uint8_t)len,
                      (uint8_t)type[0], (uint8_t)type[1], (uint8_t)type[2], (uint8_t)type[3]};
    f.insert(f.end(), hdr, hdr + 8);
    if (len) f.insert(f.end(), data, data + len);
    uLong crc = crc32(0L, (const Bytef*)type, 4);
    if (len) crc = crc32(crc, data, (uInt)len);
    uint8_t c[4] = {(uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc};
    f.insert(f.end(), c, c + 4);
}

static bool write_png(const std::string& path, int w, int h, int depth, int ctype, const std::vector<uint8_t>& rows,
                      std::string& err)
{
    std::vector<uint8_t> f = {137, 80, 78, 71, 13, 10, 26, 10};
    uint8_t ihdr[13] = {(uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w,
                        (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h,
                        (uint8_t)depth, (uint8_t)ctype, 0, 0, 0};
    put_chunk(f, "IHDR", ihdr, 13);
    uLongf clen = compressBound((uLong)rows.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, rows.data(), (uLong)rows.size(), 6) != Z_OK) { err = "zlib deflate failed"; return false; }
    put_chunk(f, "IDAT", comp.data(), clen);
    put_chunk(f, "IEND", nullptr, 0);
    FILE* o = fopen(path.c_str(), "wb");
    if (!o) { err = "cannot write " + path; return false; }
    const bool ok = fwrite(f.data(), 1, f.size(), o) == f.size();
    fclose(o);
    if (!ok) err = "short write " + path;
    return ok;
}

bool write_png_rgb(const std::string& path, int w, int h, const uint8_t* rgb, std::string& err)
{
    std::vector<uint8_t> rows(((size_t)w * 3 + 1) * h);
    for (int y = 0; y < h; ++y) {
        uint8_t* r = &rows[((size_t)w * 3 + 1) * y];
        r[0] = 0;                                        // filter type None
        memcpy(r + 1, rgb + (size_t)y * w * 3, (size_t)w * 3);
    }
    return write_png(path, w, h, 8, 2, rows, err);
}

bool write_png_mask1(const std::string& path, int w, int h, const uint8_t* mask, std::string& err)
{
    const size_t stride = ((size_t)w + 7) / 8;
    std::vector<uint8_t> rows((stride + 1) * h, 0);
    for (int y = 0; y < h; ++y) {
        uint8_t* r = &rows[(stride + 1) * y] + 1;
        for (int x = 0; x < w; ++x)
            if (mask[(size_t)y * w + x]) r[x / 8] |= (uint8_t)(0x80 >> (x % 8));
    }
    return write_png(path, w, h, 1, 0, rows, err);
}

}  // namespace arapio
