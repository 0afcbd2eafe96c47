// image.cpp — Image struct + a small PNG codec on zlib (role of reference src/image.cpp, which
// wraps the vendored stb_image).  Decoder: 8-bit gray / gray+alpha / RGB / RGBA / palette,
// non-interlaced, all five scanline filters.  Encoder: filter 0 + one zlib stream.
#include "image.h"

#include <zlib.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <vector>

namespace {

uint32_t be32(const byte* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }
void put32(std::vector<byte>& v, uint32_t x) { v.push_back(x >> 24); v.push_back(x >> 16); v.push_back(x >> 8); v.push_back(x); }

std::vector<byte> readFile(const std::string& name)
{
    FILE* f = fopen(name.c_str(), "rb");
    if (!f) throw std::runtime_error("Failed to load image '" + name + "': cannot open file");
    std::vector<byte> buf;
    byte tmp[65536];
    size_t n;
    while ((n = fread(tmp, 1, sizeof(tmp), f)) > 0) buf.insert(buf.end(), tmp, tmp + n);
    fclose(f);
    return buf;
}

int paeth(int a, int b, int c)
{
    const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// Decodes to tightly packed `want` channels.
byte* decodePng(const std::string& name, int want, int& width, int& height)
{
    const std::vector<byte> file = readFile(name);
    static const byte sig[8] = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    if (file.size() < 33 || memcmp(file.data(), sig, 8) != 0) throw std::runtime_error("Failed to load image '" + name + "': not a PNG");
    size_t pos = 8;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<byte> idat, palette;
    width = height = 0;
    while (pos + 12 <= file.size()) {
        const uint32_t len = be32(&file[pos]);
        const char* type = (const char*)&file[pos + 4];
        const byte* body = &file[pos + 8];
        if (pos + 12 + len > file.size()) throw std::runtime_error("Failed to load image '" + name + "': truncated chunk");
        if (!memcmp(type, "IHDR", 4)) {
            width = (int)be32(body); height = (int)be32(body + 4);
            depth = body[8]; ctype = body[9]; interlace = body[12];
        } else if (!memcmp(type, "PLTE", 4)) {
            palette.assign(body, body + len);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + len;
    }
    if (width <= 0 || height <= 0) throw std::runtime_error("Failed to load image '" + name + "': missing IHDR");
    if (depth != 8 || interlace != 0) throw std::runtime_error("Failed to load image '" + name + "': only 8-bit non-interlaced PNGs are supported");
    int src;
    switch (ctype) {
        case 0: src = 1; break;
        case 2: src = 3; break;
        case 3: src = 1; break;
        case 4: src = 2; break;
        case 6: src = 4; break;
        default: throw std::runtime_error("Failed to load image '" + name + "': bad colour type");
    }
    const size_t stride = (size_t)width * src;
    std::vector<byte> raw((stride + 1) * height);
    uLongf rawLen = raw.size();
    if (uncompress(raw.data(), &rawLen, idat.data(), idat.size()) != Z_OK || rawLen != raw.size())
        throw std::runtime_error("Failed to load image '" + name + "': zlib stream is corrupt");
    std::vector<byte> pix(stride * height);
    for (int y = 0; y < height; ++y) {
        const byte ft = raw[(stride + 1) * y];
        const byte* in = &raw[(stride + 1) * y + 1];
        byte* out = &pix[stride * y];
        const byte* up = y ? &pix[stride * (y - 1)] : nullptr;
        for (size_t i = 0; i < stride; ++i) {
            const int a = i >= (size_t)src ? out[i - src] : 0, b = up ? up[i] : 0, c = (up && i >= (size_t)src) ? up[i - src] : 0;
            int v = in[i];
            switch (ft) {
                case 0: break;
                case 1: v += a; break;
                case 2: v += b; break;
                case 3: v += (a + b) >> 1; break;
                case 4: v += paeth(a, b, c); break;
                default: throw std::runtime_error("Failed to load image '" + name + "': bad filter type");
            }
            out[i] = (byte)v;
        }
    }
    byte* data = (byte*)malloc((size_t)width * height * want);
    if (!data) throw std::runtime_error("Failed to load image '" + name + "': out of memory");
    for (size_t i = 0; i < (size_t)width * height; ++i) {
        byte r, g, b, a = 255;
        const byte* s = &pix[i * src];
        if (ctype == 3) {
            if ((size_t)s[0] * 3 + 2 >= palette.size()) { free(data); throw std::runtime_error("Failed to load image '" + name + "': palette index out of range"); }
            r = palette[s[0] * 3]; g = palette[s[0] * 3 + 1]; b = palette[s[0] * 3 + 2];
        } else if (src <= 2) { r = g = b = s[0]; if (src == 2) a = s[1]; }
        else { r = s[0]; g = s[1]; b = s[2]; if (src == 4) a = s[3]; }
        byte* d = &data[i * want];
        switch (want) {   // same conversions as stb's req_comp
            case 1: d[0] = (byte)((r * 77 + g * 150 + b * 29) >> 8); break;
            case 2: d[0] = (byte)((r * 77 + g * 150 + b * 29) >> 8); d[1] = a; break;
            case 3: d[0] = r; d[1] = g; d[2] = b; break;
            default: d[0] = r; d[1] = g; d[2] = b; d[3] = a; break;
        }
    }
    return data;
}

void chunk(std::vector<byte>& out, const char* type, const std::vector<byte>& body)
{
    put32(out, (uint32_t)body.size());
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), body.begin(), body.end());
    put32(out, (uint32_t)crc32(0L, &out[start], (uInt)(out.size() - start)));
}

void encodePng(const std::string& name, const byte* data, int3 shape)
{
    if (!data || shape.x <= 0 || shape.y <= 0 || shape.z < 1 || shape.z > 4)
        throw std::runtime_error("Failed to save image '" + name + "': bad shape");
    static const byte ctype[5] = { 0, 0, 4, 2, 6 };
    std::vector<byte> out = { 0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a };
    std::vector<byte> ihdr;
    put32(ihdr, shape.x); put32(ihdr, shape.y);
    ihdr.push_back(8); ihdr.push_back(ctype[shape.z]); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    chunk(out, "IHDR", ihdr);
    const size_t stride = (size_t)shape.x * shape.z;
    std::vector<byte> raw((stride + 1) * shape.y);
    for (int y = 0; y < shape.y; ++y) {
        raw[(stride + 1) * y] = 0;
        memcpy(&raw[(stride + 1) * y + 1], data + stride * y, stride);
    }
    uLongf clen = compressBound(raw.size());
    std::vector<byte> comp(clen);
    if (compress2(comp.data(), &clen, raw.data(), raw.size(), 6) != Z_OK) throw std::runtime_error("Failed to save image '" + name + "': zlib error");
    comp.resize(clen);
    chunk(out, "IDAT", comp);
    chunk(out, "IEND", {});
    FILE* f = fopen(name.c_str(), "wb");
    if (!f) throw std::runtime_error("Failed to save image '" + name + "': cannot open file");
    const size_t n = fwrite(out.data(), 1, out.size(), f);
    fclose(f);
    if (n != out.size()) throw std::runtime_error("Failed to save image '" + name + "': short write");
}

}  // namespace

Image::Image() : shape{ 0, 0, 0 }, data(nullptr) {}

Image::Image(int3 s) : shape(s), data((byte*)calloc((size_t)s.x * s.y * s.z, 1)) {}

Image::Image(byte* src, int3 s) : shape(s), data((byte*)malloc((size_t)s.x * s.y * s.z))
{
    if (data && src) memcpy(data, src, (size_t)s.x * s.y * s.z);
}

Image::Image(std::string filename, int channels) : shape{ 0, 0, channels }, data(nullptr)
{
    if (channels < 1 || channels > 4) throw std::runtime_error("Failed to load image '" + filename + "': channels must be 1..4");
    data = decodePng(filename, channels, shape.x, shape.y);
}

Image::Image(Image&& o) noexcept : shape(o.shape), data(o.data) { o.data = nullptr; o.shape = { 0, 0, 0 }; }

Image& Image::operator=(Image&& o) noexcept
{
    if (this != &o) { free(data); shape = o.shape; data = o.data; o.data = nullptr; o.shape = { 0, 0, 0 }; }
    return *this;
}

Image::~Image() { free(data); }

void Image::save(std::string filename) { encodePng(filename, data, shape); }

void Image::save(std::string filename, byte* d, int3 s) { encodePng(filename, d, s); }
