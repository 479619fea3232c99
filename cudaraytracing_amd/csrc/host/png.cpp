// PNG read (any non-interlaced colour type / bit depth -> RGB8) and write
// (RGB8) on top of zlib.  Conversion to RGB8 follows what the reference gets
// from lodepng::decode(..., LCT_RGB) (lib/xmlload.cuh:76): alpha is dropped,
// grey is replicated, palettes are expanded, 16-bit samples keep their high
// byte and sub-byte grey samples are scaled to the full range.
#include "host.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace crt {

namespace {

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}

int paeth(int a, int b, int c) {
    const int p = a + b - c;
    const int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    if (pb <= pc) return b;
    return c;
}

}  // namespace

bool png_read_rgb8(const char* path, std::vector<uint8_t>& rgb, uint32_t& width, uint32_t& height, std::string& err) {
    FILE* fp = fopen(path, "rb");
    if (!fp) { err = std::string("cannot open ") + path; return false; }
    std::vector<uint8_t> file;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, fp)) > 0) file.insert(file.end(), buf, buf + n);
    fclose(fp);

    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (file.size() < 8 + 25 || memcmp(file.data(), sig, 8) != 0) { err = "not a PNG file"; return false; }

    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, palette;
    size_t pos = 8;
    bool have_ihdr = false, done = false;
    while (!done && pos + 12 <= file.size()) {
        const uint32_t len = be32(&file[pos]);
        const char* type = (const char*)&file[pos + 4];
        if (pos + 12 + (size_t)len > file.size()) { err = "truncated PNG chunk"; return false; }
        const uint8_t* data = &file[pos + 8];
        if (!memcmp(type, "IHDR", 4)) {
            if (len < 13) { err = "bad IHDR"; return false; }
            w = be32(data); h = be32(data + 4);
            depth = data[8]; ctype = data[9]; interlace = data[12];
            have_ihdr = true;
        } else if (!memcmp(type, "PLTE", 4)) {
            palette.assign(data, data + len);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), data, data + len);
        } else if (!memcmp(type, "IEND", 4)) {
            done = true;
        }
        pos += 12 + (size_t)len;
    }
    if (!have_ihdr || w == 0 || h == 0) { err = "PNG without IHDR"; return false; }
    if (interlace != 0) { err = "interlaced PNG is not supported"; return false; }
    int channels;
    switch (ctype) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 3: channels = 1; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: err = "bad PNG colour type"; return false;
    }
    if (!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) {
        err = "unsupported PNG bit depth"; return false;
    }
    const size_t bpp_bits = (size_t)channels * depth;
    const size_t stride = ((size_t)w * bpp_bits + 7) / 8;
    const size_t bpp = bpp_bits >= 8 ? bpp_bits / 8 : 1;   // filter unit
    std::vector<uint8_t> raw((stride + 1) * (size_t)h);
    uLongf rawlen = (uLongf)raw.size();
    const int zr = uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size());
    if (zr != Z_OK || rawlen != raw.size()) { err = "PNG inflate failed"; return false; }

    // undo the scanline filters in place
    std::vector<uint8_t> img(stride * (size_t)h);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t ft = raw[(stride + 1) * y];
        const uint8_t* src = &raw[(stride + 1) * y + 1];
        uint8_t* dst = &img[stride * y];
        const uint8_t* up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t x = 0; x < stride; x++) {
            const int a = x >= bpp ? dst[x - bpp] : 0;
            const int b = up ? up[x] : 0;
            const int c = (up && x >= bpp) ? up[x - bpp] : 0;
            int v = src[x];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: err = "bad PNG filter type"; return false;
            }
            dst[x] = (uint8_t)v;
        }
    }

    rgb.resize((size_t)w * h * 3);
    for (uint32_t y = 0; y < h; y++) {
        const uint8_t* row = &img[stride * y];
        for (uint32_t x = 0; x < w; x++) {
            uint8_t* o = &rgb[((size_t)y * w + x) * 3];
            if (depth == 8 || depth == 16) {
                const size_t bs = depth / 8;                 // bytes per sample; high byte first
                const uint8_t* p = row + (size_t)x * channels * bs;
                switch (ctype) {
                    case 0: case 4: o[0] = o[1] = o[2] = p[0]; break;
                    case 2: case 6: o[0] = p[0]; o[1] = p[bs]; o[2] = p[2 * bs]; break;
                    case 3: {
                        const size_t idx = p[0];
                        if (idx * 3 + 2 < palette.size()) { o[0] = palette[idx * 3]; o[1] = palette[idx * 3 + 1]; o[2] = palette[idx * 3 + 2]; }
                        else { o[0] = o[1] = o[2] = 0; }
                        break;
                    }
                }
            } else {
                const size_t bit = (size_t)x * depth;
                const unsigned s = (row[bit >> 3] >> (8 - depth - (bit & 7))) & ((1u << depth) - 1);
                if (ctype == 3) {
                    if ((size_t)s * 3 + 2 < palette.size()) { o[0] = palette[s * 3]; o[1] = palette[s * 3 + 1]; o[2] = palette[s * 3 + 2]; }
                    else { o[0] = o[1] = o[2] = 0; }
                } else {
                    const unsigned maxv = (1u << depth) - 1;
                    o[0] = o[1] = o[2] = (uint8_t)((s * 255u) / maxv);
                }
            }
        }
    }
    width = w; height = h;
    return true;
}

bool png_write_rgb8(const char* path, const uint8_t* rgb, uint32_t width, uint32_t height, std::string& err) {
    if (!rgb || width == 0 || height == 0) { err = "empty image"; return false; }
    const size_t stride = (size_t)width * 3;
    std::vector<uint8_t> raw((stride + 1) * (size_t)height);
    for (uint32_t y = 0; y < height; y++) {
        raw[(stride + 1) * y] = 0;   // filter: none
        memcpy(&raw[(stride + 1) * y + 1], rgb + stride * y, stride);
    }
    uLongf clen = compressBound((uLong)raw.size());
    std::vector<uint8_t> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), (uLong)raw.size(), 6) != Z_OK) { err = "deflate failed"; return false; }
    comp.resize(clen);

    std::vector<uint8_t> out = {137, 80, 78, 71, 13, 10, 26, 10};
    auto chunk = [&](const char* type, const std::vector<uint8_t>& data) {
        put_be32(out, (uint32_t)data.size());
        const size_t c0 = out.size();
        out.insert(out.end(), type, type + 4);
        out.insert(out.end(), data.begin(), data.end());
        const uint32_t crc = (uint32_t)crc32(0L, &out[c0], (uInt)(out.size() - c0));
        put_be32(out, crc);
    };
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, width); put_be32(ihdr, height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk("IHDR", ihdr);
    chunk("IDAT", comp);
    chunk("IEND", {});
    FILE* fp = fopen(path, "wb");
    if (!fp) { err = std::string("cannot write ") + path; return false; }
    const bool ok = fwrite(out.data(), 1, out.size(), fp) == out.size();
    fclose(fp);
    if (!ok) err = std::string("short write to ") + path;
    return ok;
}

}  // namespace crt
