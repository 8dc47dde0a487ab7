// jpeg_decode.h -- JPEG: entropy decoding on the host, the pixel half on the host or on the GPU (jpeg_decode.cc, jpeg.hip).
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace oar {
namespace img {

#if defined(__HIPCC__)
#define OAR_JPEG_HD __host__ __device__
#else
#define OAR_JPEG_HD
#endif
// The inverse DCT's 32-bit arithmetic.  On a valid stream no intermediate leaves int32 (jidctint.c's own range analysis); coefficients
// of a corrupt one can, and there libjpeg's arithmetic wraps.  An int that overflows is undefined in C++, so the sums and products are
// carried in uint32 (defined wrap-around, the same bits) and only the shifts look at the value as signed.  Host renderer and device
// kernel share the type, so they also agree on garbage.
struct W32 {
    uint32_t u;
    W32() = default;
    OAR_JPEG_HD constexpr W32(int v) : u((uint32_t)v) {}
    OAR_JPEG_HD static constexpr W32 raw(uint32_t x) { W32 r(0); r.u = x; return r; }
    OAR_JPEG_HD constexpr int s() const { return (int)u; }
    OAR_JPEG_HD friend constexpr W32 operator+(W32 a, W32 b) { return raw(a.u + b.u); }
    OAR_JPEG_HD friend constexpr W32 operator-(W32 a, W32 b) { return raw(a.u - b.u); }
    OAR_JPEG_HD friend constexpr W32 operator*(W32 a, W32 b) { return raw(a.u * b.u); }
    OAR_JPEG_HD W32& operator+=(W32 b) { u += b.u; return *this; }
    OAR_JPEG_HD W32& operator*=(W32 b) { u *= b.u; return *this; }
    OAR_JPEG_HD constexpr W32 shl(int n) const { return raw(u << n); }
    OAR_JPEG_HD constexpr int sra(int n) const { return (int)u >> n; }   // arithmetic shift of the two's-complement value
};

struct JpegComp {
    int h = 1, v = 1;            // sampling factors
    int bw = 0, bh = 0;          // blocks per row / column of the coefficient plane (padded to whole MCUs)
    int dw = 0, dh = 0;          // real extent of the down-sampled component in samples: ceil(W h / hmax), ceil(H v / vmax)
    uint16_t q[64];              // quantisation table, natural (row-major) order
    std::vector<int16_t> coef;   // bw * bh blocks of 64 quantised coefficients, natural order
};
struct JpegImage {
    uint32_t w = 0, h = 0;
    int ncomp = 0, hmax = 1, vmax = 1, mcux = 0, mcuy = 0;
    int color = 1;               // 0 grey, 1 YCbCr, 2 RGB (libjpeg's default_decompress_parms rule)
    JpegComp comp[3];
};

bool is_jpeg(const uint8_t* b, size_t n);
// markers + Huffman (baseline / extended sequential / progressive): coefficient planes, tables, geometry.  Throws oar::Error
// (OAR_INVALID_INPUT = corrupt / truncated, OAR_UNSUPPORTED_OP = a JPEG process this library does not decode).
void jpeg_entropy_decode(const uint8_t* b, size_t n, JpegImage& out);
// dequantise + IDCT (jidctint islow) + upsample (jdsample fancy) + colour (jdcolor) on the host: w * h * 3 bytes
void jpeg_render_host(const JpegImage& im, std::vector<uint8_t>& rgb);

// the three arithmetic pieces, shared with nothing else on the host but stated once (jpeg.hip carries the same statements for the device)
void jpeg_idct_block(const int16_t* coef, const uint16_t* q, uint8_t* out, int stride);
uint8_t jpeg_upsampled(const uint8_t* plane, int stride, int dw, int dh, int hs, int vs, int x, int y);
void jpeg_ycc_to_rgb(int y, int cb, int cr, uint8_t* rgb);

}  // namespace img
}  // namespace oar
