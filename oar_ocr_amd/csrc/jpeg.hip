// jpeg.hip -- the pixel half of JPEG decoding on the GPU: dequantise + IDCT + chroma upsampling + YCbCr -> RGB.
// The same integer statements as jpeg_decode.cc (libjpeg's jidctint islow, jdsample fancy filters, jdcolor tables), so the device image is
// bit-identical to the host one (tests/test_gpu_jpeg.py).  The Huffman stream itself is serial and stays on the host; what the GPU takes
// over is the arithmetic: per 960 x 960 4:2:0 page 21 600 IDCTs and 0.9 M pixel conversions, and the decoded page is born in HBM --
// oar_ocr_predict_device reads it in place.
#include <hip/hip_runtime.h>

#include "common.h"
#include "jpeg_dev.h"
#include "jpeg_decode.h"

namespace oar {
namespace pp {

namespace {
using img::W32;
inline unsigned grid_for(long work, int block = 256, long cap = 256L * 32) {
    long g = (work + block - 1) / block;
    return (unsigned)(g < 1 ? 1 : g > cap ? cap : g);
}

// one thread = one 8x8 block: 64 coefficients in, 64 samples out (jidctint.c jpeg_idct_islow; CONST_BITS 13, PASS1_BITS 2)
__global__ __launch_bounds__(64) void jpeg_idct_kernel(JpegDevPlan plan) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= plan.total_blocks) return;
    int c = 0;
    long b = gid;
    while (c + 1 < plan.ncomp && b >= (long)plan.comp[c].bw * plan.comp[c].bh) { b -= (long)plan.comp[c].bw * plan.comp[c].bh; ++c; }
    const JpegDevComp& k = plan.comp[c];
    const int by = (int)(b / k.bw), bx = (int)(b - (long)by * k.bw);
    const int16_t* cf = k.coef + b * 64;
    const uint16_t* q = plan.q + c * 64;
    constexpr int F0298 = 2446, F0390 = 3196, F0541 = 4433, F0765 = 6270, F0899 = 7373, F1175 = 9633, F1501 = 12299, F1847 = 15137, F1961 = 16069, F2053 = 16819,
                  F2562 = 20995, F3072 = 25172;
    W32 ws[64];
#pragma unroll
    for (int x = 0; x < 8; ++x) {
        const W32 i0 = cf[x] * q[x], i1 = cf[8 + x] * q[8 + x], i2 = cf[16 + x] * q[16 + x], i3 = cf[24 + x] * q[24 + x], i4 = cf[32 + x] * q[32 + x], i5 = cf[40 + x] * q[40 + x],
                  i6 = cf[48 + x] * q[48 + x], i7 = cf[56 + x] * q[56 + x];   // |int16 x uint16| < 2^31
        W32 z1 = (i2 + i6) * F0541;
        const W32 t2 = z1 + i6 * (-F1847), t3 = z1 + i2 * F0765;
        const W32 t0 = (i0 + i4).shl(13), t1 = (i0 - i4).shl(13);
        const W32 t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
        W32 o0 = i7, o1 = i5, o2 = i3, o3 = i1;
        z1 = o0 + o3; W32 z2 = o1 + o2, z3 = o0 + o2, z4 = o1 + o3;
        const W32 z5 = (z3 + z4) * F1175;
        o0 *= F0298; o1 *= F2053; o2 *= F3072; o3 *= F1501;
        z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
        z3 += z5; z4 += z5;
        o0 += z1 + z3; o1 += z2 + z4; o2 += z2 + z3; o3 += z1 + z4;
        ws[x] = (t10 + o3 + 1024).sra(11); ws[56 + x] = (t10 - o3 + 1024).sra(11); ws[8 + x] = (t11 + o2 + 1024).sra(11); ws[48 + x] = (t11 - o2 + 1024).sra(11);
        ws[16 + x] = (t12 + o1 + 1024).sra(11); ws[40 + x] = (t12 - o1 + 1024).sra(11); ws[24 + x] = (t13 + o0 + 1024).sra(11); ws[32 + x] = (t13 - o0 + 1024).sra(11);
    }
    uint8_t* out = k.plane + ((long)by * 8 * k.bw + bx) * 8;
    const int stride = k.bw * 8;
#pragma unroll
    for (int y = 0; y < 8; ++y) {
        const W32* w = ws + y * 8;
        W32 z1 = (w[2] + w[6]) * F0541;
        const W32 t2 = z1 + w[6] * (-F1847), t3 = z1 + w[2] * F0765;
        const W32 t0 = (w[0] + w[4]).shl(13), t1 = (w[0] - w[4]).shl(13);
        const W32 t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
        W32 o0 = w[7], o1 = w[5], o2 = w[3], o3 = w[1];
        z1 = o0 + o3; W32 z2 = o1 + o2, z3 = o0 + o2, z4 = o1 + o3;
        const W32 z5 = (z3 + z4) * F1175;
        o0 *= F0298; o1 *= F2053; o2 *= F3072; o3 *= F1501;
        z1 *= -F0899; z2 *= -F2562; z3 *= -F1961; z4 *= -F0390;
        z3 += z5; z4 += z5;
        o0 += z1 + z3; o1 += z2 + z4; o2 += z2 + z3; o3 += z1 + z4;
        auto rl = [](W32 x) { const int v = (x + (1 << 17)).sra(18) + 128; return (unsigned)(v < 0 ? 0 : v > 255 ? 255 : v); };
        const unsigned lo = rl(t10 + o3) | (rl(t11 + o2) << 8) | (rl(t12 + o1) << 16) | (rl(t13 + o0) << 24);
        const unsigned hi = rl(t13 - o0) | (rl(t12 - o1) << 8) | (rl(t11 - o2) << 16) | (rl(t10 - o3) << 24);
        *reinterpret_cast<uint2*>(out + (long)y * stride) = make_uint2(lo, hi);
    }
}

// jdsample.c: one full-resolution sample of a component plane
__device__ __forceinline__ int upsampled(const JpegDevComp& k, int hs, int vs, int x, int y) {
    const int stride = k.bw * 8;
    const int cx = x / hs, cy = y / vs;
    if (hs == 1 && vs == 1) return k.plane[(long)cy * stride + cx];
    const bool fancy_h = hs == 2 && k.dw > 2;
    if (hs == 2 && vs == 1 && fancy_h) {
        const uint8_t* r = k.plane + (long)cy * stride;
        const int v = r[cx] * 3;
        if ((x & 1) == 0) return cx == 0 ? r[0] : (v + r[cx - 1] + 1) >> 2;
        return cx == k.dw - 1 ? r[cx] : (v + r[cx + 1] + 2) >> 2;
    }
    if (hs == 1 && vs == 2) {
        const int ny = (y & 1) == 0 ? max(cy - 1, 0) : min(cy + 1, k.dh - 1);
        return (k.plane[(long)cy * stride + cx] * 3 + k.plane[(long)ny * stride + cx] + ((y & 1) == 0 ? 1 : 2)) >> 2;
    }
    if (hs == 2 && vs == 2 && fancy_h) {
        const int ny = (y & 1) == 0 ? max(cy - 1, 0) : min(cy + 1, k.dh - 1);
        const uint8_t* r0 = k.plane + (long)cy * stride;
        const uint8_t* r1 = k.plane + (long)ny * stride;
        const int cur = r0[cx] * 3 + r1[cx];
        if ((x & 1) == 0) return cx == 0 ? (cur * 4 + 8) >> 4 : (cur * 3 + (r0[cx - 1] * 3 + r1[cx - 1]) + 8) >> 4;
        return cx == k.dw - 1 ? (cur * 4 + 7) >> 4 : (cur * 3 + (r0[cx + 1] * 3 + r1[cx + 1]) + 7) >> 4;
    }
    return k.plane[(long)min(cy, k.dh - 1) * stride + min(cx, k.dw - 1)];
}

__global__ __launch_bounds__(256) void jpeg_color_kernel(JpegDevPlan plan, uint8_t* __restrict__ rgb) {
    const long total = (long)plan.w * plan.h;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / plan.w), x = (int)(i - (long)y * plan.w);
        int s[3] = {0, 0, 0};
        for (int c = 0; c < plan.ncomp; ++c) s[c] = upsampled(plan.comp[c], plan.hmax / plan.comp[c].h, plan.vmax / plan.comp[c].v, x, y);
        int r, g, b;
        if (plan.color == 0) { r = g = b = s[0]; }
        else if (plan.color == 2) { r = s[0]; g = s[1]; b = s[2]; }
        else {   // jdcolor.c ycc_rgb_convert (SCALEBITS 16)
            const int xb = s[1] - 128, xr = s[2] - 128;
            r = s[0] + ((91881 * xr + 32768) >> 16);
            g = s[0] + ((-22554 * xb + 32768 + (-46802) * xr) >> 16);
            b = s[0] + ((116130 * xb + 32768) >> 16);
            r = r < 0 ? 0 : r > 255 ? 255 : r; g = g < 0 ? 0 : g > 255 ? 255 : g; b = b < 0 ? 0 : b > 255 ? 255 : b;
        }
        rgb[i * 3] = (uint8_t)r; rgb[i * 3 + 1] = (uint8_t)g; rgb[i * 3 + 2] = (uint8_t)b;
    }
}
}  // namespace

void jpeg_render(hipStream_t s, const JpegDevPlan& plan, uint8_t* rgb) {
    {
        ProfScope ps(s, "jpeg_idct", 3.0 * 64 * (double)plan.total_blocks, 0.0);
        hipLaunchKernelGGL(jpeg_idct_kernel, dim3((unsigned)((plan.total_blocks + 63) / 64)), dim3(64), 0, s, plan);
    }
    {
        ProfScope ps(s, "jpeg_color", 4.5 * (double)plan.w * plan.h, 0.0);
        hipLaunchKernelGGL(jpeg_color_kernel, dim3(grid_for((long)plan.w * plan.h)), dim3(256), 0, s, plan, rgb);
    }
}

}  // namespace pp
}  // namespace oar
