// Minimal PNG codec on zlib: decode to RGBA8 (what the reference gets from
// lodepng::decode, cudaTracer/CudaMesh.h:85-104) and encode RGBA8 (what
// lodepng::encode writes for cudaTracer/PicDumper.h:39).  Decodes every
// standard colour type and bit depth (grey 1-16, RGB, palette 1-8, grey+alpha,
// RGBA), tRNS transparency and Adam7 interlacing; 16-bit samples keep their
// high byte.
#ifndef RT_HOST_PNG_H
#define RT_HOST_PNG_H

#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace rt {
namespace png {

inline uint32_t be32(const unsigned char* p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }

inline bool decodeMemory(const std::vector<unsigned char>& file, std::vector<unsigned char>& rgba, unsigned& width,
                         unsigned& height, std::string* err) {
    auto fail = [&](const char* m) {
        if (err) *err = m;
        return false;
    };
    static const unsigned char sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (file.size() < 8 || std::memcmp(file.data(), sig, 8) != 0) return fail("not a PNG");
    unsigned bitDepth = 0, colorType = 0, interlace = 0;
    std::vector<unsigned char> idat, palette, trns;
    size_t pos = 8;
    bool haveHeader = false, done = false;
    while (!done && pos + 12 <= file.size()) {
        uint32_t len = be32(&file[pos]);
        const unsigned char* type = &file[pos + 4];
        const unsigned char* body = &file[pos + 8];
        if (pos + 12 + (size_t)len > file.size()) return fail("truncated chunk");
        if (!std::memcmp(type, "IHDR", 4)) {
            if (len < 13) return fail("short IHDR");
            width = be32(body);
            height = be32(body + 4);
            bitDepth = body[8];
            colorType = body[9];
            interlace = body[12];
            haveHeader = true;
        } else if (!std::memcmp(type, "PLTE", 4)) {
            palette.assign(body, body + len);
        } else if (!std::memcmp(type, "tRNS", 4)) {
            trns.assign(body, body + len);
        } else if (!std::memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!std::memcmp(type, "IEND", 4)) {
            done = true;
        }
        pos += 12 + (size_t)len;
    }
    if (!haveHeader || width == 0 || height == 0) return fail("missing IHDR");
    if (interlace > 1) return fail("bad interlace method");
    unsigned channels;
    switch (colorType) {
        case 0: channels = 1; break;
        case 2: channels = 3; break;
        case 3: channels = 1; break;
        case 4: channels = 2; break;
        case 6: channels = 4; break;
        default: return fail("bad colour type");
    }
    const bool depthOk = (colorType == 0) ? (bitDepth == 1 || bitDepth == 2 || bitDepth == 4 || bitDepth == 8 || bitDepth == 16)
                       : (colorType == 3) ? (bitDepth == 1 || bitDepth == 2 || bitDepth == 4 || bitDepth == 8)
                                          : (bitDepth == 8 || bitDepth == 16);
    if (!depthOk) return fail("bit depth not allowed for this colour type");
    const size_t bitsPerPixel = (size_t)channels * bitDepth;
    const size_t bpp = bitsPerPixel >= 8 ? bitsPerPixel / 8 : 1;  // filter distance in bytes

    // passes: the whole image, or the seven Adam7 sub-images
    struct Pass { unsigned x0, y0, dx, dy, w, h; };
    std::vector<Pass> passes;
    if (interlace == 0) {
        passes.push_back({0, 0, 1, 1, width, height});
    } else {
        static const unsigned X0[7] = {0, 4, 0, 2, 0, 1, 0}, Y0[7] = {0, 0, 4, 0, 2, 0, 1};
        static const unsigned DX[7] = {8, 8, 4, 4, 2, 2, 1}, DY[7] = {8, 8, 8, 4, 4, 2, 2};
        for (int i = 0; i < 7; ++i) {
            const unsigned w = (width + DX[i] - 1 - X0[i]) / DX[i], h = (height + DY[i] - 1 - Y0[i]) / DY[i];
            if (width > X0[i] && height > Y0[i] && w && h) passes.push_back({X0[i], Y0[i], DX[i], DY[i], w, h});
        }
    }
    size_t rawSize = 0;
    for (const Pass& ps : passes) rawSize += (((size_t)ps.w * bitsPerPixel + 7) / 8 + 1) * ps.h;
    std::vector<unsigned char> raw(rawSize);
    uLongf rawLen = (uLongf)raw.size();
    if (idat.empty() || uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size())
        return fail("inflate failed");

    rgba.assign((size_t)width * height * 4, 0);
    const unsigned maxVal = (1u << (bitDepth < 8 ? bitDepth : 8)) - 1u;
    size_t at = 0;
    for (const Pass& ps : passes) {
        const size_t stride = ((size_t)ps.w * bitsPerPixel + 7) / 8;
        std::vector<unsigned char> prev(stride, 0);
        for (unsigned y = 0; y < ps.h; ++y) {
            unsigned char* row = &raw[at];
            at += stride + 1;
            const unsigned filter = row[0];
            unsigned char* cur = row + 1;
            for (size_t i = 0; i < stride; ++i) {
                const int a = i >= bpp ? cur[i - bpp] : 0;
                const int b = prev[i];
                const int c = i >= bpp ? prev[i - bpp] : 0;
                int pred = 0;
                switch (filter) {
                    case 0: pred = 0; break;
                    case 1: pred = a; break;
                    case 2: pred = b; break;
                    case 3: pred = (a + b) / 2; break;
                    case 4: {
                        const int p = a + b - c;
                        const int pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
                        pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                        break;
                    }
                    default: return fail("bad filter");
                }
                cur[i] = (unsigned char)(cur[i] + pred);
            }
            std::memcpy(prev.data(), cur, stride);
            for (unsigned x = 0; x < ps.w; ++x) {
                unsigned char* d = &rgba[((size_t)(ps.y0 + y * ps.dy) * width + (ps.x0 + x * ps.dx)) * 4];
                if (bitDepth < 8) {  // grey or palette, packed most significant bit first
                    const size_t bit = (size_t)x * bitDepth;
                    const unsigned v = (cur[bit >> 3] >> (8 - bitDepth - (bit & 7))) & maxVal;
                    if (colorType == 3) {
                        if ((v + 1) * 3 > palette.size()) return fail("palette index");
                        d[0] = palette[v * 3]; d[1] = palette[v * 3 + 1]; d[2] = palette[v * 3 + 2];
                        d[3] = v < trns.size() ? trns[v] : 255;
                    } else {
                        d[0] = d[1] = d[2] = (unsigned char)(v * 255u / maxVal);
                        d[3] = (trns.size() >= 2 && v == ((unsigned)trns[0] << 8 | trns[1])) ? 0 : 255;
                    }
                    continue;
                }
                const size_t step = bitDepth / 8;  // 16-bit samples: keep the high byte
                const unsigned char* sp = cur + x * bpp;
                switch (colorType) {
                    case 0:
                        d[0] = d[1] = d[2] = sp[0];
                        d[3] = 255;
                        if (trns.size() >= 2 && (step == 2 ? (sp[0] == trns[0] && sp[1] == trns[1]) : (trns[0] == 0 && sp[0] == trns[1]))) d[3] = 0;
                        break;
                    case 2: {
                        d[0] = sp[0]; d[1] = sp[step]; d[2] = sp[2 * step]; d[3] = 255;
                        if (trns.size() >= 6) {
                            bool key = true;
                            for (int k = 0; k < 3 && key; ++k)
                                key = step == 2 ? (sp[2 * k] == trns[2 * k] && sp[2 * k + 1] == trns[2 * k + 1])
                                                : (trns[2 * k] == 0 && sp[k] == trns[2 * k + 1]);
                            if (key) d[3] = 0;
                        }
                        break;
                    }
                    case 3: {
                        const unsigned idx = sp[0];
                        if ((idx + 1) * 3 > palette.size()) return fail("palette index");
                        d[0] = palette[idx * 3]; d[1] = palette[idx * 3 + 1]; d[2] = palette[idx * 3 + 2];
                        d[3] = idx < trns.size() ? trns[idx] : 255;
                        break;
                    }
                    case 4: d[0] = d[1] = d[2] = sp[0]; d[3] = sp[step]; break;
                    case 6: d[0] = sp[0]; d[1] = sp[step]; d[2] = sp[2 * step]; d[3] = sp[3 * step]; break;
                }
            }
        }
    }
    return true;
}

inline bool decodeFile(const std::string& path, std::vector<unsigned char>& rgba, unsigned& width, unsigned& height,
                       std::string* err) {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) {
        if (err) *err = "cannot open file";
        return false;
    }
    std::vector<unsigned char> bytes;
    unsigned char buf[65536];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) bytes.insert(bytes.end(), buf, buf + n);
    std::fclose(f);
    return decodeMemory(bytes, rgba, width, height, err);
}

inline void putChunk(std::vector<unsigned char>& out, const char* type, const unsigned char* body, size_t len) {
    auto put32 = [&](uint32_t v) {
        for (int s = 24; s >= 0; s -= 8) out.push_back((unsigned char)(v >> s));
    };
    put32((uint32_t)len);
    const size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    if (len) out.insert(out.end(), body, body + len);
    put32((uint32_t)crc32(0L, &out[start], (uInt)(out.size() - start)));
}

inline std::vector<unsigned char> encodeMemory(const unsigned char* rgba, unsigned width, unsigned height) {
    std::vector<unsigned char> out = {137, 80, 78, 71, 13, 10, 26, 10};
    unsigned char ihdr[13] = {0};
    for (int k = 0; k < 4; ++k) {
        ihdr[k] = (unsigned char)(width >> (24 - 8 * k));
        ihdr[4 + k] = (unsigned char)(height >> (24 - 8 * k));
    }
    ihdr[8] = 8;  // bit depth
    ihdr[9] = 6;  // RGBA
    putChunk(out, "IHDR", ihdr, 13);
    const size_t stride = (size_t)width * 4;
    std::vector<unsigned char> raw((stride + 1) * height);
    for (unsigned y = 0; y < height; ++y) {
        raw[y * (stride + 1)] = 0;  // filter: none
        std::memcpy(&raw[y * (stride + 1) + 1], rgba + y * stride, stride);
    }
    uLongf zLen = compressBound((uLong)raw.size());
    std::vector<unsigned char> z(zLen);
    compress2(z.data(), &zLen, raw.data(), (uLong)raw.size(), 6);
    putChunk(out, "IDAT", z.data(), zLen);
    putChunk(out, "IEND", nullptr, 0);
    return out;
}

inline bool encodeFile(const std::string& path, const unsigned char* rgba, unsigned width, unsigned height) {
    const std::vector<unsigned char> bytes = encodeMemory(rgba, width, height);
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    const bool ok = std::fwrite(bytes.data(), 1, bytes.size(), f) == bytes.size();
    std::fclose(f);
    return ok;
}

}  // namespace png
}  // namespace rt
#endif
