// image_decode.cc -- SURVEY 8f-3, the step before the path: encoded image bytes -> RgbImage.
//
// Reference: load_image_from_memory / load_image (oar-ocr-core/src/utils/image.rs:65-92) = image 0.25.6 `load_from_memory` +
// `DynamicImage::to_rgb8`.  This file decodes PNG (every colour type, bit depth and interlace mode of the format) and produces
// exactly the bytes that chain produces:
//   * png 0.17 with Transformations::EXPAND (what image's PngDecoder sets): palette -> RGB, 1 / 2 / 4-bit grey scaled to 8 bits by
//     255 / (2^depth - 1), tRNS turned into an alpha channel; 16-bit samples stay 16-bit;
//   * to_rgb8: grey replicated into the three channels, alpha dropped (no pre-multiplication), 16-bit -> 8-bit as (v + 128) / 257
//     (image's FromPrimitive<u16> for u8).
// PNG is lossless and its decoding is fully specified, so "the same bytes" is a property of the format, not of an
// implementation: pinned against PIL and against an independent zlib + numpy oracle (tests/test_image_decode_cpu.py).
// The DEFLATE stream is inflated by the system zlib (the reference uses a Rust inflate; any conforming inflate yields the same
// bytes).  JPEG is jpeg_decode.cc, BMP / PNM / TIFF / GIF image_misc_decode.cc; the other formats the image crate knows (WebP, ...) are reported as
// OAR_UNSUPPORTED_OP -- never guessed at.
// Host code by nature (a DEFLATE stream and PNG's left / up filters are serial per image); images decode in parallel across
// caller threads (the entry point holds no lock).
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "common.h"

namespace oar {
namespace img {

namespace {
inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

struct Header {
    uint32_t w = 0, h = 0;
    int depth = 0, color = 0, interlace = 0;
    int channels() const { return color == 0 ? 1 : color == 2 ? 3 : color == 3 ? 1 : color == 4 ? 2 : 4; }
    size_t row_bytes(uint32_t width) const { return ((size_t)width * channels() * depth + 7) / 8; }
    int bpp() const { return std::max(1, channels() * depth / 8); }   // filter unit: bytes per complete pixel, at least 1
};

inline int paeth(int a, int b, int c) {
    const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// undo the filter of one scanline in place; prev = the previous unfiltered scanline of the same pass (nullptr: all zero)
void unfilter(int type, uint8_t* cur, const uint8_t* prev, size_t n, int bpp) {
    switch (type) {
        case 0: break;
        case 1: for (size_t i = bpp; i < n; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]); break;
        case 2: if (prev) for (size_t i = 0; i < n; ++i) cur[i] = (uint8_t)(cur[i] + prev[i]); break;
        case 3:
            for (size_t i = 0; i < n; ++i) {
                const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev ? prev[i] : 0;
                cur[i] = (uint8_t)(cur[i] + ((a + b) >> 1));
            }
            break;
        case 4:
            for (size_t i = 0; i < n; ++i) {
                const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, b = prev ? prev[i] : 0, c = (prev && i >= (size_t)bpp) ? prev[i - bpp] : 0;
                cur[i] = (uint8_t)(cur[i] + paeth(a, b, c));
            }
            break;
        default: fail(OAR_INVALID_INPUT, "image load: PNG scanline with filter type " + std::to_string(type));
    }
}

// sample k (0-based, over all channels) of an unfiltered scanline, as stored: depth 1 / 2 / 4 / 8 -> the raw value, 16 -> big endian
inline uint32_t sample(const uint8_t* row, size_t k, int depth) {
    switch (depth) {
        case 8: return row[k];
        case 16: return ((uint32_t)row[k * 2] << 8) | row[k * 2 + 1];
        case 4: return (row[k >> 1] >> ((1 - (k & 1)) * 4)) & 15u;
        case 2: return (row[k >> 2] >> ((3 - (k & 3)) * 2)) & 3u;
        default: return (row[k >> 3] >> (7 - (k & 7))) & 1u;
    }
}
inline uint8_t to8(uint32_t v16) { return (uint8_t)((v16 + 128u) / 257u); }   // image: FromPrimitive<u16> for u8

// one unfiltered scanline of `width` pixels -> RGB8 at dst (3 * width bytes)
void row_to_rgb(const Header& hd, const uint8_t* row, uint32_t width, const std::vector<uint8_t>& plte, uint8_t* dst, size_t dst_stride_px) {
    const int d = hd.depth;
    for (uint32_t x = 0; x < width; ++x) {
        uint8_t* o = dst + (size_t)x * dst_stride_px * 3;
        switch (hd.color) {
            case 0: {   // grey
                const uint32_t v = sample(row, x, d);
                const uint8_t g = d == 16 ? to8(v) : d == 8 ? (uint8_t)v : (uint8_t)(v * (255u / ((1u << d) - 1u)));
                o[0] = o[1] = o[2] = g;
                break;
            }
            case 4: {   // grey + alpha (8 / 16): alpha dropped
                const uint32_t v = sample(row, (size_t)x * 2, d);
                const uint8_t g = d == 16 ? to8(v) : (uint8_t)v;
                o[0] = o[1] = o[2] = g;
                break;
            }
            case 2: case 6: {   // RGB / RGBA (8 / 16)
                const size_t k = (size_t)x * (hd.color == 2 ? 3 : 4);
                for (int c = 0; c < 3; ++c) { const uint32_t v = sample(row, k + c, d); o[c] = d == 16 ? to8(v) : (uint8_t)v; }
                break;
            }
            default: {  // palette
                const uint32_t i = sample(row, x, d);
                OAR_CHECK((size_t)i * 3 + 2 < plte.size(), OAR_INVALID_INPUT, "image load: PNG palette index out of range");
                o[0] = plte[i * 3]; o[1] = plte[i * 3 + 1]; o[2] = plte[i * 3 + 2];
            }
        }
    }
}
}  // namespace

bool is_png(const uint8_t* b, size_t n) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    return n >= 8 && std::memcmp(b, sig, 8) == 0;
}

// what image::guess_format would have called the bytes (for the error message of an undecoded format)
const char* sniff(const uint8_t* b, size_t n) {
    if (is_png(b, n)) return "PNG";
    if (n >= 3 && b[0] == 0xFF && b[1] == 0xD8 && b[2] == 0xFF) return "JPEG";
    if (n >= 6 && (!std::memcmp(b, "GIF87a", 6) || !std::memcmp(b, "GIF89a", 6))) return "GIF";
    if (n >= 12 && !std::memcmp(b, "RIFF", 4) && !std::memcmp(b + 8, "WEBP", 4)) return "WebP";
    if (n >= 2 && b[0] == 'B' && b[1] == 'M') return "BMP";
    if (n >= 4 && (!std::memcmp(b, "II*\0", 4) || !std::memcmp(b, "MM\0*", 4))) return "TIFF";
    if (n >= 2 && b[0] == 'P' && b[1] >= '1' && b[1] <= '7') return "PNM";
    return nullptr;
}

void decode_png(const uint8_t* b, size_t n, std::vector<uint8_t>& rgb, uint32_t& width, uint32_t& height) {
    OAR_CHECK(is_png(b, n), OAR_INVALID_INPUT, "image load: not a PNG signature");
    Header hd;
    std::vector<uint8_t> plte, idat;
    bool have_ihdr = false, have_iend = false;
    size_t pos = 8;
    while (pos + 12 <= n && !have_iend) {
        const uint32_t len = be32(b + pos);
        OAR_CHECK(len <= 0x7fffffffu && pos + 12 + (size_t)len <= n, OAR_INVALID_INPUT, "image load: truncated PNG chunk");
        const uint8_t* type = b + pos + 4;
        const uint8_t* data = b + pos + 8;
        const uint32_t crc = be32(data + len);
        OAR_CHECK((uint32_t)crc32(crc32(0L, Z_NULL, 0), type, 4 + len) == crc, OAR_INVALID_INPUT, "image load: PNG chunk CRC mismatch");
        if (!std::memcmp(type, "IHDR", 4)) {
            OAR_CHECK(!have_ihdr && len == 13, OAR_INVALID_INPUT, "image load: bad PNG IHDR");
            hd.w = be32(data); hd.h = be32(data + 4); hd.depth = data[8]; hd.color = data[9]; hd.interlace = data[12];
            OAR_CHECK(hd.w > 0 && hd.h > 0 && hd.w <= 0x7fffffffu && hd.h <= 0x7fffffffu && data[10] == 0 && data[11] == 0 && hd.interlace <= 1, OAR_INVALID_INPUT,
                      "image load: unsupported PNG IHDR fields");
            const int d = hd.depth;
            const bool ok = (hd.color == 0 && (d == 1 || d == 2 || d == 4 || d == 8 || d == 16)) || ((hd.color == 2 || hd.color == 4 || hd.color == 6) && (d == 8 || d == 16)) ||
                            (hd.color == 3 && (d == 1 || d == 2 || d == 4 || d == 8));
            OAR_CHECK(ok, OAR_INVALID_INPUT, "image load: invalid PNG colour type / bit depth combination");
            OAR_CHECK((uint64_t)hd.w * hd.h <= (uint64_t)1 << 30, OAR_INVALID_INPUT, "image load: PNG dimensions exceed the decoder's limit");
            have_ihdr = true;
        } else {
            OAR_CHECK(have_ihdr, OAR_INVALID_INPUT, "image load: PNG chunk before IHDR");
            if (!std::memcmp(type, "PLTE", 4)) {
                OAR_CHECK(len % 3 == 0 && len <= 768, OAR_INVALID_INPUT, "image load: bad PNG PLTE");
                plte.assign(data, data + len);
            } else if (!std::memcmp(type, "IDAT", 4)) {
                idat.insert(idat.end(), data, data + len);
            } else if (!std::memcmp(type, "IEND", 4)) {
                have_iend = true;
            } else {
                OAR_CHECK(type[0] & 0x20, OAR_INVALID_INPUT, "image load: unknown critical PNG chunk");   // ancillary chunks (tRNS, gAMA, ...) change nothing in RGB8
            }
        }
        pos += 12 + (size_t)len;
    }
    OAR_CHECK(have_ihdr && have_iend && !idat.empty(), OAR_INVALID_INPUT, "image load: PNG without IHDR / IDAT / IEND");
    OAR_CHECK(hd.color != 3 || !plte.empty(), OAR_INVALID_INPUT, "image load: palette PNG without PLTE");

    // size of the filtered stream: per pass, (1 filter byte + row bytes) per non-empty scanline
    static const int x0[7] = {0, 4, 0, 2, 0, 1, 0}, y0[7] = {0, 0, 4, 0, 2, 0, 1}, dx[7] = {8, 8, 4, 4, 2, 2, 1}, dy[7] = {8, 8, 8, 4, 4, 2, 2};
    const int passes = hd.interlace ? 7 : 1;
    auto pass_w = [&](int p) { return hd.interlace ? (hd.w > (uint32_t)x0[p] ? (hd.w - x0[p] + dx[p] - 1) / dx[p] : 0u) : hd.w; };
    auto pass_h = [&](int p) { return hd.interlace ? (hd.h > (uint32_t)y0[p] ? (hd.h - y0[p] + dy[p] - 1) / dy[p] : 0u) : hd.h; };
    size_t raw_len = 0;
    for (int p = 0; p < passes; ++p) if (pass_w(p) && pass_h(p)) raw_len += (size_t)pass_h(p) * (1 + hd.row_bytes(pass_w(p)));
    // Allocation budget of image::Limits::default() (max_alloc = 512 MiB, image 0.25 ImageReader): a forged IHDR must not make the
    // library reserve gigabytes before a single IDAT byte has been looked at.  Distinct error, as the crate's LimitError is.
    constexpr uint64_t kMaxAlloc = 512ull << 20;
    OAR_CHECK((uint64_t)raw_len + (uint64_t)hd.w * hd.h * 3 <= kMaxAlloc, OAR_INVALID_INPUT,
              "image load: PNG needs more than the 512 MiB allocation limit (image::Limits::default)");
    // inflate in 32-bit-safe slices, and grow `raw` with the bytes actually produced: the allocation follows the data, not the header
    std::vector<uint8_t> raw;
    {
        z_stream zs;
        std::memset(&zs, 0, sizeof zs);
        OAR_CHECK(inflateInit(&zs) == Z_OK, OAR_INTERNAL, "image load: zlib inflateInit failed");
        size_t in_at = 0;
        int rc = Z_OK;
        while (rc != Z_STREAM_END) {
            if (zs.avail_in == 0 && in_at < idat.size()) {
                const size_t take = std::min<size_t>(idat.size() - in_at, 1u << 30);
                zs.next_in = idat.data() + in_at; zs.avail_in = (uInt)take; in_at += take;
            }
            const size_t have = raw.size();
            if (have == raw_len) break;   // the image is complete: whatever follows in the stream (end marker, adler32, excess IDAT bytes) is not
                                          // looked at -- the png crate behind image 0.25 stops at the last scanline too
            const size_t grow = std::min<size_t>(raw_len - have, std::max<size_t>(have, 1u << 20));   // doubling, at least 1 MiB
            raw.resize(have + grow);
            zs.next_out = raw.data() + have; zs.avail_out = (uInt)grow;
            rc = inflate(&zs, Z_NO_FLUSH);
            raw.resize(have + (grow - zs.avail_out));
            if (rc == Z_STREAM_END) break;
            if (rc != Z_OK || (zs.avail_out != 0 && zs.avail_in == 0 && in_at >= idat.size())) {   // error, or input exhausted before the stream ended
                inflateEnd(&zs);
                oar::fail(OAR_INVALID_INPUT, "image load: corrupt or truncated PNG image data");
            }
        }
        inflateEnd(&zs);
        OAR_CHECK(raw.size() == raw_len, OAR_INVALID_INPUT, "image load: corrupt or truncated PNG image data");
    }
    width = hd.w; height = hd.h;
    rgb.assign((size_t)hd.w * hd.h * 3, 0);
    const int bpp = hd.bpp();
    size_t at = 0;
    for (int p = 0; p < passes; ++p) {
        const uint32_t pw = pass_w(p), ph = pass_h(p);
        if (!pw || !ph) continue;
        const size_t rb = hd.row_bytes(pw);
        uint8_t* prev = nullptr;
        for (uint32_t r = 0; r < ph; ++r) {
            const int ft = raw[at];
            uint8_t* cur = raw.data() + at + 1;
            unfilter(ft, cur, prev, rb, bpp);
            const uint32_t y = hd.interlace ? (uint32_t)y0[p] + r * dy[p] : r;
            uint8_t* dst = rgb.data() + ((size_t)y * hd.w + (hd.interlace ? x0[p] : 0)) * 3;
            row_to_rgb(hd, cur, pw, plte, dst, hd.interlace ? (size_t)dx[p] : 1);
            prev = cur;
            at += 1 + rb;
        }
    }
}

}  // namespace img
}  // namespace oar
