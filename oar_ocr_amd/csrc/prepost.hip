// prepost.hip -- gfx950 kernels for the byte/index stages around the two networks.
//
// These stages are HBM-bound integer/byte work; they are NOT reshaped into GEMMs.  Each kernel restates the
// reference's CPU loop with the same f32 operation order (compiled with -ffp-contract=off: the reference
// never fuses mul+add, processors/simd.rs:11-14), so results are bit-identical, not merely close.
#include "prepost.h"

#include <hip/hip_runtime.h>

#include "common.h"

namespace oar {
namespace pp {

static inline unsigned grid_for(long work, int block = 256, long cap = 256L * 32) {
    long g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

// ------------------------------------------------------------------------------------------ a4 normalize
// One thread = 4 pixels = 12 source bytes (three aligned 32-bit loads) -> 12 floats.
struct NormP {
    int src[3];
    float alpha[3], beta[3];
};
struct NormSrcs { const uint8_t* p[32]; };   // separate page buffers of one launch (null table: images are contiguous in rgb)
__global__ __launch_bounds__(256) void normalize_kernel(const uint8_t* __restrict__ rgb, NormSrcs srcs, int use_srcs, float* __restrict__ out, long n_images,
                                                        long plane, NormP p, int layout) {
    const long quads_per_img = (plane + 3) >> 2;
    const long total = n_images * quads_per_img;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        long img = i / quads_per_img, q = i - img * quads_per_img;
        long p0 = q * 4;
        const uint8_t* s = (use_srcs ? srcs.p[img] + p0 * 3 : rgb + (img * plane + p0) * 3);
        int np = (int)min(4L, plane - p0);
        uint8_t b[12];
        if (np == 4 && ((reinterpret_cast<uintptr_t>(s) & 3) == 0)) {
            const uint32_t* s4 = reinterpret_cast<const uint32_t*>(s);
            uint32_t w0 = s4[0], w1 = s4[1], w2 = s4[2];
#pragma unroll
            for (int k = 0; k < 4; ++k) { b[k] = (w0 >> (8 * k)) & 0xFF; b[4 + k] = (w1 >> (8 * k)) & 0xFF; b[8 + k] = (w2 >> (8 * k)) & 0xFF; }
        } else {
            for (int k = 0; k < np * 3; ++k) b[k] = s[k];
        }
        if (layout == 1) {
            float* o = out + (img * plane + p0) * 3;
            float v[12];
#pragma unroll
            for (int px = 0; px < 4; ++px)
#pragma unroll
                for (int c = 0; c < 3; ++c) { float t = (float)b[px * 3 + p.src[c]] * p.alpha[c]; v[px * 3 + c] = t + p.beta[c]; }
            if (np == 4 && ((reinterpret_cast<uintptr_t>(o) & 15) == 0)) {
                float4* o4 = reinterpret_cast<float4*>(o);
                o4[0] = make_float4(v[0], v[1], v[2], v[3]);
                o4[1] = make_float4(v[4], v[5], v[6], v[7]);
                o4[2] = make_float4(v[8], v[9], v[10], v[11]);
            } else {
                for (int k = 0; k < np * 3; ++k) o[k] = v[k];
            }
        } else {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float* o = out + (img * 3 + c) * plane + p0;
                for (int px = 0; px < np; ++px) { float t = (float)b[px * 3 + p.src[c]] * p.alpha[c]; o[px] = t + p.beta[c]; }
            }
        }
    }
}
void normalize(hipStream_t s, const uint8_t* rgb, float* out, int64_t n_images, int64_t plane, const int src[3],
               const float alpha[3], const float beta[3], int layout) {
    if (n_images * plane == 0) return;
    NormP p;
    for (int i = 0; i < 3; ++i) { p.src[i] = src[i]; p.alpha[i] = alpha[i]; p.beta[i] = beta[i]; }
    ProfScope ps(s, "normalize", 15.0 * (double)n_images * plane, 6.0 * (double)n_images * plane);
    NormSrcs none{};
    hipLaunchKernelGGL(normalize_kernel, dim3(grid_for(n_images * ((plane + 3) / 4))), dim3(256), 0, s, rgb, none, 0, out, (long)n_images, (long)plane, p, layout);
}
void normalize_pages(hipStream_t s, const uint8_t* const* d_pages, int n_pages, float* out, int64_t plane, const int src[3], const float alpha[3],
                     const float beta[3], int layout) {
    if (n_pages * plane == 0) return;
    OAR_CHECK(n_pages <= 32, OAR_INTERNAL, "normalize_pages: at most 32 pages per launch");
    NormP p;
    for (int i = 0; i < 3; ++i) { p.src[i] = src[i]; p.alpha[i] = alpha[i]; p.beta[i] = beta[i]; }
    NormSrcs t{};
    for (int i = 0; i < n_pages; ++i) t.p[i] = d_pages[i];
    ProfScope ps(s, "normalize", 15.0 * (double)n_pages * plane, 6.0 * (double)n_pages * plane);
    hipLaunchKernelGGL(normalize_kernel, dim3(grid_for(n_pages * ((plane + 3) / 4))), dim3(256), 0, s, nullptr, t, 1, out, (long)n_pages, (long)plane, p, layout);
}

// ------------------------------------------------------------------------------------------ Triangle resize helpers
// image 0.25.6 imageops::sample: per output coordinate o along an axis of input length `in`:
//   ratio = in/out; sratio = max(ratio,1); support = sratio; c = (o+0.5)*ratio;
//   left = clamp(floor(c-support), 0, in-1); right = clamp(ceil(c+support), left+1, in);
//   w_i = tri((i - (c-0.5))/sratio) for i in [left,right), normalised by their (sequential) sum.
struct Taps {
    int left, right;
    float center, sratio, sum;
};
__device__ __forceinline__ float tri(float x) { float a = fabsf(x); return a < 1.0f ? 1.0f - a : 0.0f; }
__device__ __forceinline__ Taps make_taps(int o, int in, int out) {
    Taps t;
    float ratio = (float)in / (float)out;
    t.sratio = ratio < 1.0f ? 1.0f : ratio;
    float support = 1.0f * t.sratio;
    float c = ((float)o + 0.5f) * ratio;
    long l = (long)floorf(c - support);
    l = l < 0 ? 0 : (l > (long)in - 1 ? (long)in - 1 : l);
    long r = (long)ceilf(c + support);
    r = r < l + 1 ? l + 1 : (r > (long)in ? (long)in : r);
    t.left = (int)l; t.right = (int)r;
    t.center = c - 0.5f;
    float sum = 0.0f;
    for (int i = t.left; i < t.right; ++i) sum += tri(((float)i - t.center) / t.sratio);
    t.sum = sum;
    return t;
}
__device__ __forceinline__ float tap_w(const Taps& t, int i) { return tri(((float)i - t.center) / t.sratio) / t.sum; }

// Vertical pass value (f32, unrounded) of source column x for output row described by tv.
__device__ __forceinline__ void vertical_sum(const uint8_t* src, int w, int x, const Taps& tv, float& r, float& g, float& b, long sgn = 1) {
    float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
    for (int j = tv.left; j < tv.right; ++j) {
        const uint8_t* p = src + sgn * ((long)j * w + x) * 3;
        float wv = tap_w(tv, j);
        t0 += (float)p[0] * wv; t1 += (float)p[1] * wv; t2 += (float)p[2] * wv;
    }
    r = t0; g = t1; b = t2;
}
__device__ __forceinline__ uint8_t to_u8_round(float v) { return (uint8_t)roundf(fminf(fmaxf(v, 0.0f), 255.0f)); }

// flip != 0: the image read is imageops::rotate180 of (src, w, h) -- pixel (x, y) of the rotated image is pixel
// (w - 1 - x, h - 1 - y) of the source, i.e. the source walked backwards from its last pixel: same taps, same order, same
// values as resizing a materialised rotated copy (src/oarocr/ocr.rs:785-788 followed by crnn.rs:104-109).
__device__ __forceinline__ void resize_pixel(const uint8_t* src, int w, int h, int nw, int nh, int ox, int oy, uint8_t out[3], int flip = 0) {
    const long sgn = flip ? -1 : 1;
    if (flip) src += ((long)w * h - 1) * 3;
    if (nw == w && nh == h) {
        const uint8_t* p = src + sgn * ((long)oy * w + ox) * 3;
        out[0] = p[0]; out[1] = p[1]; out[2] = p[2];
        return;
    }
    Taps tv = make_taps(oy, h, nh), th = make_taps(ox, w, nw);
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
    constexpr int MAXT = 8;
    const int nv = tv.right - tv.left;
    if (nv <= MAXT) {
        // the vertical weights do not depend on the column: evaluate them once (same values, same summation order as
        // vertical_sum -- only the redundant divisions go away)
        float wv[MAXT];
#pragma unroll
        for (int j = 0; j < MAXT; ++j) wv[j] = j < nv ? tap_w(tv, tv.left + j) : 0.0f;
        for (int i = th.left; i < th.right; ++i) {
            float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
            const uint8_t* p = src + sgn * ((long)tv.left * w + i) * 3;
#pragma unroll
            for (int j = 0; j < MAXT; ++j) {
                if (j < nv) { t0 += (float)p[0] * wv[j]; t1 += (float)p[1] * wv[j]; t2 += (float)p[2] * wv[j]; p += sgn * (long)w * 3; }
            }
            float wh = tap_w(th, i);
            a0 += t0 * wh; a1 += t1 * wh; a2 += t2 * wh;
        }
    } else {
        for (int i = th.left; i < th.right; ++i) {
            float r, g, b;
            vertical_sum(src, w, i, tv, r, g, b, sgn);
            float wh = tap_w(th, i);
            a0 += r * wh; a1 += g * wh; a2 += b * wh;
        }
    }
    out[0] = to_u8_round(a0); out[1] = to_u8_round(a1); out[2] = to_u8_round(a2);
}

__global__ __launch_bounds__(256) void resize_triangle_kernel(const uint8_t* src, int w, int h, uint8_t* dst, int nw, int nh) {
    long total = (long)nw * nh;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int ox = (int)(i % nw), oy = (int)(i / nw);
        uint8_t o[3];
        resize_pixel(src, w, h, nw, nh, ox, oy, o);
        dst[i * 3] = o[0]; dst[i * 3 + 1] = o[1]; dst[i * 3 + 2] = o[2];
    }
}
void resize_triangle(hipStream_t s, const uint8_t* src, int w, int h, uint8_t* dst, int nw, int nh) {
    if ((long)nw * nh == 0) return;
    ProfScope ps(s, "resize_triangle", 3.0 * ((double)w * h + (double)nw * nh), 0.0);
    hipLaunchKernelGGL(resize_triangle_kernel, dim3(grid_for((long)nw * nh)), dim3(256), 0, s, src, w, h, dst, nw, nh);
}

// ------------------------------------------------------------------------------------------ a16 recognizer input pack
__global__ __launch_bounds__(256) void rec_pack_kernel(const CropDesc* descs, int img_h, int Wt, float* out, int nchw) {
    const int n = blockIdx.y;
    const CropDesc d = descs[n];
    const long plane = (long)img_h * Wt;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (long)gridDim.x * blockDim.x) {
        int ox = (int)(i % Wt), oy = (int)(i / Wt);
        float v[3] = {0.0f, 0.0f, 0.0f};
        if (ox < d.rw) {
            uint8_t px[3];
            resize_pixel(d.src, d.w, d.h, d.rw, img_h, ox, oy, px, d.flip);
#pragma unroll
            for (int c = 0; c < 3; ++c) v[c] = ((float)px[2 - c] / 255.0f - 0.5f) / 0.5f;
        }
        if (nchw) {
            float* o = out + (long)n * 3 * plane + i;
            o[0] = v[0]; o[plane] = v[1]; o[2 * plane] = v[2];
        } else {
            float* o = out + ((long)n * plane + i) * 3;
            o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
        }
    }
}
void rec_pack(hipStream_t s, const CropDesc* d_descs, int n, int img_h, int Wt, float* out, int nchw) {
    if (n == 0 || Wt == 0) return;
    ProfScope ps(s, "rec_pack", 12.0 * (double)n * img_h * Wt, 0.0);
    long plane = (long)img_h * Wt;
    hipLaunchKernelGGL(rec_pack_kernel, dim3(grid_for(plane, 256, 64), n), dim3(256), 0, s, d_descs, img_h, Wt, out, nchw);
}

__global__ __launch_bounds__(256) void rec_resize_u8_kernel(const CropDesc* descs, const ResizedImg* dst, int img_h) {
    const int n = blockIdx.y;
    const CropDesc d = descs[n];
    uint8_t* out = dst[n].ptr;
    const long plane = (long)img_h * d.rw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % d.rw), oy = (int)(i / d.rw);
        uint8_t px[3];
        resize_pixel(d.src, d.w, d.h, d.rw, img_h, ox, oy, px, d.flip);
        out[i * 3] = px[0]; out[i * 3 + 1] = px[1]; out[i * 3 + 2] = px[2];
    }
}
void rec_resize_u8(hipStream_t s, const CropDesc* d_descs, const ResizedImg* d_dst, int n, int img_h, int max_rw) {
    if (n == 0 || max_rw == 0) return;
    ProfScope ps(s, "rec_pack", 3.0 * (double)n * img_h * max_rw, 0.0);
    hipLaunchKernelGGL(rec_resize_u8_kernel, dim3(grid_for((long)img_h * max_rw, 256, 64), n), dim3(256), 0, s, d_descs, d_dst, img_h);
}

// ------------------------------------------------------------------------------------------ a7 threshold
__global__ __launch_bounds__(256) void threshold_kernel(const float* __restrict__ pred, uint8_t* __restrict__ mask, long n, float thresh) {
    long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<const float4*>(pred)[i];
        uint32_t o = (v.x > thresh ? 0xFFu : 0u) | (v.y > thresh ? 0xFF00u : 0u) | (v.z > thresh ? 0xFF0000u : 0u) | (v.w > thresh ? 0xFF000000u : 0u);
        reinterpret_cast<uint32_t*>(mask)[i] = o;
    }
    for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) mask[i] = pred[i] > thresh ? 255 : 0;
}
void threshold(hipStream_t s, const float* pred, uint8_t* mask, int64_t n, float thresh) {
    if (n == 0) return;
    ProfScope ps(s, "threshold", 5.0 * (double)n, 0.0);
    hipLaunchKernelGGL(threshold_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, s, pred, mask, (long)n, thresh);
}

// ------------------------------------------------------------------------------------------ mask -> bits (what crosses PCIe)
// One thread packs 8 mask bytes of one row into one byte (pixel x -> bit x & 7); rows are padded to row_bytes = ceil(W / 8).
__global__ __launch_bounds__(256) void pack_mask_bits_kernel(const uint8_t* __restrict__ mask, uint8_t* __restrict__ bits, long rows, int W, int row_bytes) {
    const long total = rows * row_bytes;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / row_bytes;
        const int bx = (int)(i - r * row_bytes), x0 = bx * 8;
        const uint8_t* m = mask + r * W + x0;
        unsigned v = 0;
        if (x0 + 8 <= W && (((size_t)m) & 7) == 0) {
            const uint2 q = *reinterpret_cast<const uint2*>(m);
#pragma unroll
            for (int k = 0; k < 4; ++k) { v |= ((q.x >> (8 * k)) & 0xffu) ? (1u << k) : 0u; v |= ((q.y >> (8 * k)) & 0xffu) ? (1u << (k + 4)) : 0u; }
        } else {
            for (int k = 0; k < 8 && x0 + k < W; ++k) v |= m[k] ? (1u << k) : 0u;
        }
        bits[i] = (uint8_t)v;
    }
}
void pack_mask_bits(hipStream_t s, const uint8_t* mask, uint8_t* bits, int n_images, int height, int width) {
    const long rows = (long)n_images * height;
    if (rows == 0 || width == 0) return;
    const int row_bytes = (width + 7) / 8;
    ProfScope ps(s, "pack_mask", (double)rows * width + (double)rows * row_bytes, 0.0);
    hipLaunchKernelGGL(pack_mask_bits_kernel, dim3(grid_for(rows * row_bytes)), dim3(256), 0, s, mask, bits, rows, width, row_bytes);
}

// ------------------------------------------------------------------------------------------ a7 in one pass over the detector's output
// What a detector sub-batch needs of its network output when the host follows the borders (the default): channel 0 kept in `probs` (the box
// scores read it after the arena has been reused) and the thresholded mask as a bit plane for the read-back.  copy2d + threshold + pack_mask_bits
// did that in three launches over two streams; here one thread takes 8 pixels of a row: two float4 loads, two float4 stores, one byte of
// bits (bit k = pred[x0 + k] > thresh, the predicate threshold_kernel writes as 255 and pack_mask_bits reads as != 0).
__global__ __launch_bounds__(256) void db_keep_and_pack_kernel(const float* __restrict__ net_out, long img_stride, float* __restrict__ probs, uint8_t* __restrict__ bits,
                                                              long rows, int H, int W, int row_bytes, float thresh) {
    const long total = rows * row_bytes;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / row_bytes, img = r / H;
        const int bx = (int)(i - r * row_bytes), x0 = bx * 8;
        const float* src = net_out + img * img_stride + (r - img * H) * (long)W + x0;
        float* dst = probs + r * (long)W + x0;
        unsigned v = 0;
        if (x0 + 8 <= W && ((((size_t)src) | ((size_t)dst)) & 15) == 0) {
            const float4 a = reinterpret_cast<const float4*>(src)[0], b = reinterpret_cast<const float4*>(src)[1];
            reinterpret_cast<float4*>(dst)[0] = a; reinterpret_cast<float4*>(dst)[1] = b;
            v = (a.x > thresh ? 1u : 0u) | (a.y > thresh ? 2u : 0u) | (a.z > thresh ? 4u : 0u) | (a.w > thresh ? 8u : 0u) |
                (b.x > thresh ? 16u : 0u) | (b.y > thresh ? 32u : 0u) | (b.z > thresh ? 64u : 0u) | (b.w > thresh ? 128u : 0u);
        } else {
            for (int k = 0; k < 8 && x0 + k < W; ++k) { const float p = src[k]; dst[k] = p; v |= p > thresh ? (1u << k) : 0u; }
        }
        bits[i] = (uint8_t)v;
    }
}
void db_keep_and_pack(hipStream_t s, const float* net_out, int64_t img_stride, float* probs, uint8_t* bits, int n_images, int height, int width, float thresh) {
    const long rows = (long)n_images * height;
    if (rows == 0 || width == 0) return;
    const int row_bytes = (width + 7) / 8;
    ProfScope ps(s, "threshold", 8.0 * (double)rows * width + (double)rows * row_bytes, 0.0);
    hipLaunchKernelGGL(db_keep_and_pack_kernel, dim3(grid_for(rows * row_bytes)), dim3(256), 0, s, net_out, (long)img_stride, probs, bits, rows, height, width, row_bytes, thresh);
}

// ------------------------------------------------------------------------------------------ mask dilation (use_dilation)
__global__ __launch_bounds__(256) void dilate3x3_kernel(const uint8_t* __restrict__ mask, uint8_t* __restrict__ out, int height, int width) {
    const long plane = (long)height * width;
    const uint8_t* m = mask + (long)blockIdx.y * plane;
    uint8_t* o = out + (long)blockIdx.y * plane;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % width), y = (int)(i / width);
        unsigned any = 0;
        for (int dy = -1; dy <= 1; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= height) continue;
            const uint8_t* r = m + (long)yy * width;
            any |= r[x];
            if (x > 0) any |= r[x - 1];
            if (x + 1 < width) any |= r[x + 1];
        }
        o[i] = any ? 255 : 0;
    }
}
void dilate3x3(hipStream_t s, const uint8_t* mask, uint8_t* out, int n_images, int height, int width) {
    if (n_images == 0 || height == 0 || width == 0) return;
    ProfScope ps(s, "dilate", 2.0 * (double)n_images * height * width, 0.0);
    hipLaunchKernelGGL(dilate3x3_kernel, dim3(grid_for((long)height * width, 256, 64), n_images), dim3(256), 0, s, mask, out, height, width);
}

// ------------------------------------------------------------------------------------------ a18 CTC argmax
// One workgroup per (batch,time) row. Each lane keeps (max, LAST index attaining it) over its strided slice;
// the reduction prefers the larger value and, on equal values, the larger index => "last max index wins".
__device__ __forceinline__ void amax_merge(float& v, int& i, float ov, int oi) {
    if (ov > v || (ov == v && oi > i)) { v = ov; i = oi; }
}
__global__ __launch_bounds__(256) void ctc_argmax_kernel(const float* __restrict__ probs, int vocab, int64_t* idx, float* prob) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const long row = blockIdx.x;
    const float* r = probs + row * (long)vocab;
    float best = -INFINITY;
    int bi = 0;
    for (int i = threadIdx.x; i < vocab; i += 256) {
        float v = r[i];
        if (v >= best) { best = v; bi = i; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(best, o, 64);
        int oi = __shfl_xor(bi, o, 64);
        amax_merge(best, bi, ov, oi);
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) amax_merge(best, bi, sv[w], si[w]);
        idx[row] = bi;
        prob[row] = best;
    }
}
void ctc_argmax(hipStream_t s, const float* probs, int64_t rows, int vocab, int64_t* idx, float* prob) {
    if (rows == 0) return;
    OAR_CHECK(vocab > 0, OAR_INVALID_INPUT, "ctc_argmax: vocab == 0");
    ProfScope ps(s, "ctc_argmax", 4.0 * (double)rows * vocab, 0.0);
    hipLaunchKernelGGL(ctc_argmax_kernel, dim3((unsigned)rows), dim3(256), 0, s, probs, vocab, idx, prob);
}

// ------------------------------------------------------------------------------------------ a10 box score
// One workgroup per box. Rows are summed left-to-right by one lane each (the reference's sequential `+=`),
// then lane 0 adds the row sums in row order -- exactly the reference's summation tree (db_score.rs:86-132).
// The reference's sequential `+=` along a scanline span (db_score.rs:118-124): the ADDS keep that order, the LOADS do not have to.  A
// load-then-add loop pays one L2 round trip per pixel (78 us per launch in round 3; 16 single loads in flight: 33 us).  Here a lane keeps 128 pixels
// in flight as 32 four-pixel loads (rows are only 4-byte aligned: f4u), then 32, then a clamped tail.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
template <int NV>
__device__ __forceinline__ float span_batch(const float* p, float line) {
    f4u v[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = *reinterpret_cast<const f4u*>(p + i * 4);
#pragma unroll
    for (int i = 0; i < NV; ++i) { line += v[i].x; line += v[i].y; line += v[i].z; line += v[i].w; }
    return line;
}
__device__ __forceinline__ float span_sum(const float* row, unsigned x1, unsigned xe, float line) {
    unsigned x = x1;
    for (; x + 128 <= xe; x += 128) line = span_batch<32>(row + x, line);
    for (; x + 32 <= xe; x += 32) line = span_batch<8>(row + x, line);
    if (x < xe) {
        float v[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) v[k] = row[x + k < xe ? x + k : xe - 1];
#pragma unroll
        for (int k = 0; k < 32; ++k) if (x + k < xe) line += v[k];
    }
    return line;
}
__device__ __forceinline__ unsigned f2u(float v) { return v > 0.0f ? (v >= 4294967296.0f ? 0xFFFFFFFFu : (unsigned)v) : 0u; }
__global__ __launch_bounds__(256) void box_scores_kernel(const float* pred, int height, int width, const ScoreBox* boxes, float* scores) {
    __shared__ float rsum[256];
    __shared__ unsigned rcnt[256];
    const ScoreBox b = boxes[blockIdx.x];
    const float* map = pred + (long)b.image * height * width;
    float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float x = b.pts[i * 2], y = b.pts[i * 2 + 1];
        if (x < mnx) mnx = x; if (x > mxx) mxx = x;
        if (y < mny) mny = y; if (y > mxy) mxy = y;
    }
    float fx0 = fminf(fmaxf(floorf(mnx), 0.0f), (float)width - 1.0f), fx1 = fminf(fmaxf(ceilf(mxx), 0.0f), (float)width - 1.0f);
    float fy0 = fminf(fmaxf(floorf(mny), 0.0f), (float)height - 1.0f), fy1 = fminf(fmaxf(ceilf(mxy), 0.0f), (float)height - 1.0f);
    const unsigned start_y = f2u(fy0), end_y = f2u(fy1) + 1, start_x = f2u(fx0), end_x = f2u(fx1) + 1;
    float total = 0.0f;
    unsigned long long pixels = 0;
    for (unsigned y0 = start_y; y0 < end_y; y0 += 256) {
        unsigned yy = y0 + threadIdx.x;
        float line = 0.0f;
        unsigned lp = 0;
        if (yy < end_y) {
            float y = (float)yy + 0.5f;
            float xs[4];
            int ni = 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                int j = (i + 1) & 3;
                float p1x = b.pts[i * 2], p1y = b.pts[i * 2 + 1], p2x = b.pts[j * 2], p2y = b.pts[j * 2 + 1];
                if (((p1y <= y && y < p2y) || (p2y <= y && y < p1y)) && fabsf(p2y - p1y) > 1.1920929e-7f) {
                    float x = p1x + (y - p1y) * (p2x - p1x) / (p2y - p1y);
                    xs[ni++] = x;
                }
            }
            for (int i = 1; i < ni; ++i) { float k = xs[i]; int j = i - 1; while (j >= 0 && xs[j] > k) { xs[j + 1] = xs[j]; --j; } xs[j + 1] = k; }
            unsigned yi = f2u(y);
            if (yi < (unsigned)height) {
                const float* row = map + (long)yi * width;
                for (int c = 0; c + 1 < ni; c += 2) {
                    unsigned x1 = f2u(fmaxf(xs[c], (float)start_x)), x2 = f2u(fminf(xs[c + 1], (float)end_x));
                    if (x1 < x2 && x1 >= start_x && x2 <= end_x) {
                        unsigned xe = x2 < (unsigned)width ? x2 : (unsigned)width;
                        if (x1 < xe) {
                            line = span_sum(row, x1, xe, line);
                            lp += xe - x1;
                        }
                    }
                }
            }
        }
        rsum[threadIdx.x] = line; rcnt[threadIdx.x] = lp;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned lim = min(256u, end_y - y0);
            for (unsigned i = 0; i < lim; ++i) { total += rsum[i]; pixels += rcnt[i]; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) scores[blockIdx.x] = pixels > 0 ? total / (float)pixels : 0.0f;
}
void box_scores(hipStream_t s, const float* pred, int height, int width, const ScoreBox* d_boxes, int n_boxes, float* d_scores) {
    if (n_boxes == 0) return;
    ProfScope ps(s, "box_scores", 0.0, 0.0);
    hipLaunchKernelGGL(box_scores_kernel, dim3(n_boxes), dim3(256), 0, s, pred, height, width, d_boxes, d_scores);
}

// ------------------------------------------------------------------------------------------ a11 unclip
// host::unclip (db_host.cc) statement for statement: f64 area / perimeter / delta, the ring on the 1/100 px grid, per corner the arc
// that swings the previous edge's offset vector onto the next edge's (Clipper2 ClipperOffset::{BuildNormals, OffsetPoint, DoRound},
// arc tolerance radius / 500).  -ffp-contract=off: no multiply-add is fused, as on the host.
__global__ __launch_bounds__(64) void unclip_quads_kernel(const ScoreBox* __restrict__ boxes, int n, float ratio, UnclipOut* __restrict__ out) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n) return;
    constexpr double kGrid = 100.0, kPi = 3.14159265358979323846, kEpsD = 2.220446049250313e-16;
    UnclipOut& o = out[b];
    double qx[4], qy[4];
    for (int i = 0; i < 4; ++i) { qx[i] = (double)boxes[b].pts[i * 2]; qy[i] = (double)boxes[b].pts[i * 2 + 1]; }
    double shoelace = 0.0, perimeter = 0.0;
    for (int i = 0, p = 3; i < 4; p = i++) shoelace += (qy[p] + qy[i]) * (qx[p] - qx[i]);
    const double area = fabs(shoelace * 0.5);
    for (int i = 1; i < 4; ++i) perimeter += hypot(qx[i] - qx[i - 1], qy[i] - qy[i - 1]);
    perimeter += hypot(qx[0] - qx[3], qy[0] - qy[3]);
    if (area <= kEpsD || perimeter <= kEpsD) { o.n_pts = 0; return; }
    const double delta = area * (double)ratio / perimeter;
    if (fabs(delta) <= kEpsD) { o.n_pts = 0; return; }
    long long rx[4], ry[4];
    int rn = 0;
    for (int i = 0; i < 4; ++i) {
        const long long gx = (long long)round(qx[i] * kGrid), gy = (long long)round(qy[i] * kGrid);
        if (rn && rx[rn - 1] == gx && ry[rn - 1] == gy) continue;
        rx[rn] = gx; ry[rn] = gy; ++rn;
    }
    while (rn > 1 && rx[rn - 1] == rx[0] && ry[rn - 1] == ry[0]) --rn;
    if (rn < 3) { o.n_pts = 0; return; }
    int cnt = 0;
    bool overflow = false;
    auto emit = [&](double gx, double gy) {
        if (cnt < kUnclipMaxPts) {
            o.pts[cnt * 2] = (float)((double)(long long)round(gx) / kGrid);
            o.pts[cnt * 2 + 1] = (float)((double)(long long)round(gy) / kGrid);
        } else overflow = true;
        ++cnt;
    };
    const double grid_delta = delta * kGrid;
    if (fabs(grid_delta) < 0.5) {
        for (int i = 0; i < rn; ++i) emit((double)rx[i], (double)ry[i]);
    } else {
        double twice = 0.0;
        for (int i = 0, p = rn - 1; i < rn; p = i++) twice += (double)(ry[p] + ry[i]) * (double)(rx[p] - rx[i]);
        const double radius = twice * 0.5 < 0 ? -grid_delta : grid_delta;
        const double r = fabs(radius), tol = r * 0.002;
        const double per_turn = fmin(kPi / acos(1.0 - tol / r), r * kPi);
        double sn = sin(2.0 * kPi / per_turn);
        const double cs = cos(2.0 * kPi / per_turn);
        if (radius < 0.0) sn = -sn;
        const double per_rad = per_turn / (2.0 * kPi);
        double ux[4], uy[4];
        for (int e = 0; e < rn; ++e) {
            const int f = e + 1 == rn ? 0 : e + 1;
            double dx = (double)(rx[f] - rx[e]), dy = (double)(ry[f] - ry[e]);
            if (dx == 0.0 && dy == 0.0) { ux[e] = uy[e] = 0.0; continue; }
            const double inv_len = 1.0 / sqrt(dx * dx + dy * dy);
            dx *= inv_len; dy *= inv_len;
            ux[e] = dy; uy[e] = -dx;
        }
        for (int v = 0, in_e = rn - 1; v < rn; in_e = v++) {
            const double cx = (double)rx[v], cy = (double)ry[v];
            double turn_sin = uy[v] * ux[in_e] - uy[in_e] * ux[v];
            const double turn_cos = ux[v] * ux[in_e] + uy[v] * uy[in_e];
            turn_sin = turn_sin > 1.0 ? 1.0 : turn_sin < -1.0 ? -1.0 : turn_sin;
            double sx = ux[in_e] * radius, sy = uy[in_e] * radius;
            const double ex = cx + ux[v] * radius, ey = cy + uy[v] * radius;
            if (turn_cos > -0.999 && turn_sin * radius < 0) { o.n_pts = -1; return; }   // reflex corner: never for a mini box; the host handles it
            emit(cx + sx, cy + sy);
            const int hops = (int)ceil(per_rad * fabs(atan2(turn_sin, turn_cos)));
            for (int h = 1; h < hops; ++h) {
                const double nx = sx * cs - sn * sy, ny = sx * sn + sy * cs;
                sx = nx; sy = ny;
                emit(cx + sx, cy + sy);
            }
            emit(ex, ey);
        }
    }
    if (overflow) { o.n_pts = -1; return; }
    if (cnt > 1 && fabsf(o.pts[0] - o.pts[(cnt - 1) * 2]) < 1.1920929e-7f && fabsf(o.pts[1] - o.pts[(cnt - 1) * 2 + 1]) < 1.1920929e-7f) --cnt;
    o.n_pts = cnt < 3 ? 0 : cnt;
}
void unclip_quads(hipStream_t s, const ScoreBox* d_boxes, int n_boxes, float ratio, UnclipOut* d_out) {
    if (n_boxes == 0) return;
    ProfScope ps(s, "unclip", 0.0, 0.0);
    hipLaunchKernelGGL(unclip_quads_kernel, dim3((n_boxes + 63) / 64), dim3(64), 0, s, d_boxes, n_boxes, ratio, d_out);
}

// Polygon of any size.  One workgroup per polygon, one lane per scanline row.  The reference collects the row's edge
// crossings, sorts them and sums the spans pair by pair; here the crossings are produced in sorted order by repeated
// selection of the next smallest (x, edge) over the edge list -- the same multiset in the same order, with no per-row
// buffer whose size would cap the polygon.  Row sums and their reduction are those of box_scores_kernel.
__global__ __launch_bounds__(256) void poly_scores_kernel(const float* pred, int height, int width, const float* pts, const PolyDesc* polys, float* scores) {
    __shared__ float rsum[256];
    __shared__ unsigned rcnt[256];
    extern __shared__ float lp_pts[];     // the polygon's points when they fit (cap floats), else read from global
    const PolyDesc pd = polys[blockIdx.x];
    const float* map = pred + (long)pd.image * height * width;
    const float* gp = pts + (long)pd.first * 2;
    const int n = pd.count;
    const int cap = 8192;                 // floats = 4096 points
    const bool staged = n * 2 <= cap;
    if (staged) for (int i = threadIdx.x; i < n * 2; i += 256) lp_pts[i] = gp[i];
    __syncthreads();
    const float* P = staged ? lp_pts : gp;
    // aabb: the same single pass over the points, by lane 0-style min/max (order independent)
    float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    if (n == 0) { mnx = mny = mxx = mxy = 0.0f; }
    for (int i = 0; i < n; ++i) {
        const float x = P[i * 2], y = P[i * 2 + 1];
        if (x < mnx) mnx = x; if (x > mxx) mxx = x;
        if (y < mny) mny = y; if (y > mxy) mxy = y;
    }
    float fx0 = fminf(fmaxf(floorf(mnx), 0.0f), (float)width - 1.0f), fx1 = fminf(fmaxf(ceilf(mxx), 0.0f), (float)width - 1.0f);
    float fy0 = fminf(fmaxf(floorf(mny), 0.0f), (float)height - 1.0f), fy1 = fminf(fmaxf(ceilf(mxy), 0.0f), (float)height - 1.0f);
    const unsigned start_y = f2u(fy0), end_y = f2u(fy1) + 1, start_x = f2u(fx0), end_x = f2u(fx1) + 1;
    float total = 0.0f;
    unsigned long long pixels = 0;
    for (unsigned y0 = start_y; y0 < end_y; y0 += 256) {
        const unsigned yy = y0 + threadIdx.x;
        float line = 0.0f;
        unsigned lp = 0;
        if (yy < end_y && n > 0) {
            const float y = (float)yy + 0.5f;
            const unsigned yi = f2u(y);
            const float* row = map + (long)yi * width;
            float last_x = -INFINITY;
            int last_e = -1;
            bool have_open = false;
            float open_x = 0.0f;
            for (;;) {
                // next crossing in (x, edge index) order after (last_x, last_e)
                float bx = INFINITY;
                int be = -1;
                for (int i = 0; i < n; ++i) {
                    const int j = i + 1 == n ? 0 : i + 1;
                    const float p1y = P[i * 2 + 1], p2y = P[j * 2 + 1];
                    if (!(((p1y <= y && y < p2y) || (p2y <= y && y < p1y)) && fabsf(p2y - p1y) > 1.1920929e-7f)) continue;
                    const float p1x = P[i * 2], p2x = P[j * 2];
                    const float x = p1x + (y - p1y) * (p2x - p1x) / (p2y - p1y);
                    const bool after = x > last_x || (x == last_x && i > last_e);
                    if (after && (x < bx || (x == bx && i < be) || be < 0)) { bx = x; be = i; }
                }
                if (be < 0) break;
                last_x = bx; last_e = be;
                if (!have_open) { open_x = bx; have_open = true; continue; }
                have_open = false;
                if (yi < (unsigned)height) {
                    const unsigned x1 = f2u(fmaxf(open_x, (float)start_x)), x2 = f2u(fminf(bx, (float)end_x));
                    if (x1 < x2 && x1 >= start_x && x2 <= end_x) {
                        const unsigned xe = x2 < (unsigned)width ? x2 : (unsigned)width;
                        if (x1 < xe) {
                            line = span_sum(row, x1, xe, line);
                            lp += xe - x1;
                        }
                    }
                }
            }
        }
        rsum[threadIdx.x] = line; rcnt[threadIdx.x] = lp;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned lim = min(256u, end_y - y0);
            for (unsigned i = 0; i < lim; ++i) { total += rsum[i]; pixels += rcnt[i]; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) scores[blockIdx.x] = pixels > 0 ? total / (float)pixels : 0.0f;
}
void poly_scores(hipStream_t s, const float* pred, int height, int width, const float* d_pts_xy, const PolyDesc* d_polys, int n_polys, float* d_scores) {
    if (n_polys == 0) return;
    ProfScope ps(s, "poly_scores", 0.0, 0.0);
    hipLaunchKernelGGL(poly_scores_kernel, dim3(n_polys), dim3(256), 8192 * sizeof(float), s, pred, height, width, d_pts_xy, d_polys, d_scores);
}

// ------------------------------------------------------------------------------------------ a14 rotate-crop
__device__ __forceinline__ float cubic_kernel(float t) {
    const float A = -0.5f;
    float a = fabsf(t);
    if (a <= 1.0f) return (A + 2.0f) * a * a * a - (A + 3.0f) * a * a + 1.0f;
    else if (a < 2.0f) return A * a * a * a - 5.0f * A * a * a + 8.0f * A * a - 4.0f * A;
    return 0.0f;
}
__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__global__ __launch_bounds__(256) void rotate_crops_kernel(const WarpDesc* descs, uint8_t* pool) {
    const WarpDesc d = descs[blockIdx.y];
    const int out_w = d.rot ? d.oh : d.ow, out_h = d.rot ? d.ow : d.oh;
    const long total = (long)out_w * out_h;
    uint8_t* out = pool + d.out_off;
    const uint8_t* crop = d.page + ((long)d.top * d.page_w + d.left) * 3;  // AABB crop origin inside the page
    const long stride = (long)d.page_w * 3;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int xo = (int)(i % out_w), yo = (int)(i / out_w);
        // rotate270: out(x', y') = in(x = w-1-y', y = x')
        int dx = d.rot ? d.ow - 1 - yo : xo, dy = d.rot ? xo : yo;
        uint8_t r, g, b;
        if (d.mode == 1) {
            const uint8_t* p = crop + (long)dy * stride + (long)dx * 3;
            r = p[0]; g = p[1]; b = p[2];
        } else {
            float fx = (float)dx, fy = (float)dy;
            float px = d.inv[0] * fx; px = d.inv[1] * fy + px; px = d.inv[2] * 1.0f + px;
            float py = d.inv[3] * fx; py = d.inv[4] * fy + py; py = d.inv[5] * 1.0f + py;
            float pz = d.inv[6] * fx; pz = d.inv[7] * fy + pz; pz = d.inv[8] * 1.0f + pz;
            if (fabsf(pz) > 1.1920929e-7f) {
                float x = px / pz, y = py / pz;
                int xi = (int)floorf(x), yi = (int)floorf(y);
                float ddx = x - (float)xi, ddy = y - (float)yi;
                float wx[4] = {cubic_kernel(ddx + 1.0f), cubic_kernel(ddx), cubic_kernel(ddx - 1.0f), cubic_kernel(ddx - 2.0f)};
                float wy[4] = {cubic_kernel(ddy + 1.0f), cubic_kernel(ddy), cubic_kernel(ddy - 1.0f), cubic_kernel(ddy - 2.0f)};
                long cx[4], cy[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) { cx[k] = (long)clampi(xi - 1 + k, 0, d.cw - 1) * 3; cy[k] = (long)clampi(yi - 1 + k, 0, d.ch - 1) * stride; }
                float r0 = 0.0f, r1 = 0.0f, r2 = 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        float wt = wx[k] * wy[j];
                        const uint8_t* p = crop + cy[j] + cx[k];
                        r0 += wt * (float)p[0]; r1 += wt * (float)p[1]; r2 += wt * (float)p[2];
                    }
                r = (uint8_t)fminf(fmaxf(roundf(r0), 0.0f), 255.0f);
                g = (uint8_t)fminf(fmaxf(roundf(r1), 0.0f), 255.0f);
                b = (uint8_t)fminf(fmaxf(roundf(r2), 0.0f), 255.0f);
            } else {
                r = crop[0]; g = crop[1]; b = crop[2];
            }
        }
        out[i * 3] = r; out[i * 3 + 1] = g; out[i * 3 + 2] = b;
    }
}
void rotate_crops(hipStream_t s, const WarpDesc* d_descs, int n, uint8_t* out_pool, int max_out_pixels) {
    if (n == 0) return;
    ProfScope ps(s, "rotate_crops", 6.0 * (double)n * max_out_pixels, 0.0);
    hipLaunchKernelGGL(rotate_crops_kernel, dim3(grid_for(max_out_pixels, 256, 64), n), dim3(256), 0, s, d_descs, out_pool);
}

// ------------------------------------------------------------------------------------------ config 5: a22 / a23
__global__ __launch_bounds__(256) void cls_pack_kernel(const ClsDesc* descs, int crop_h, int crop_w, NormP np_, float* out, int nchw) {
    const int n = blockIdx.y;
    const ClsDesc d = descs[n];
    const long plane = (long)crop_h * crop_w;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += (long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % crop_w), oy = (int)(i / crop_w);
        uint8_t px[3];
        // the value of a resized pixel does not depend on the crop: sample the resize at (x1 + ox, y1 + oy) directly
        resize_pixel(d.src, d.w, d.h, d.nw, d.nh, d.x1 + ox, d.y1 + oy, px);
        float v[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { float t = (float)px[np_.src[c]] * np_.alpha[c]; v[c] = t + np_.beta[c]; }
        if (nchw) {
            float* o = out + (long)n * 3 * plane + i;
            o[0] = v[0]; o[plane] = v[1]; o[2 * plane] = v[2];
        } else {
            float* o = out + ((long)n * plane + i) * 3;
            o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
        }
    }
}
void cls_pack(hipStream_t s, const ClsDesc* d_descs, int n, int crop_h, int crop_w, const float alpha[3], const float beta[3], float* out, int nchw) {
    if (n == 0 || crop_h * crop_w == 0) return;
    NormP p;
    for (int i = 0; i < 3; ++i) { p.src[i] = i; p.alpha[i] = alpha[i]; p.beta[i] = beta[i]; }
    ProfScope ps(s, "cls_pack", 12.0 * (double)n * crop_h * crop_w, 0.0);
    hipLaunchKernelGGL(cls_pack_kernel, dim3(grid_for((long)crop_h * crop_w, 256, 64), n), dim3(256), 0, s, d_descs, crop_h, crop_w, p, out, nchw);
}

__global__ __launch_bounds__(256) void rotate_rgb_kernel(const uint8_t* __restrict__ src, int w, int h, int quarter, uint8_t* __restrict__ dst) {
    const long total = (long)w * h;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int x = (int)(i % w), y = (int)(i / w);
        long o;
        if (quarter == 1) o = (long)x * h + (h - 1 - y);
        else if (quarter == 2) o = (long)(h - 1 - y) * w + (w - 1 - x);
        else if (quarter == 3) o = (long)(w - 1 - x) * h + y;
        else o = i;
        const uint8_t* p = src + i * 3;
        dst[o * 3] = p[0]; dst[o * 3 + 1] = p[1]; dst[o * 3 + 2] = p[2];
    }
}
void rotate_rgb(hipStream_t s, const uint8_t* src, int w, int h, int quarter, uint8_t* dst) {
    if ((long)w * h == 0) return;
    ProfScope ps(s, "rotate_rgb", 6.0 * (double)w * h, 0.0);
    hipLaunchKernelGGL(rotate_rgb_kernel, dim3(grid_for((long)w * h)), dim3(256), 0, s, src, w, h, quarter, dst);
}

__global__ __launch_bounds__(256) void bgr_planes_to_rgb_kernel(const float* __restrict__ planes, long plane, float scale, uint8_t* __restrict__ out) {
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < plane; p += (long)gridDim.x * blockDim.x) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            float v = planes[(long)(2 - c) * plane + p] * scale;
            v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);       // f32::clamp; `NaN as u8` == 0
            out[p * 3 + c] = (v != v) ? (uint8_t)0 : (uint8_t)v;   // truncation
        }
    }
}
void bgr_planes_to_rgb(hipStream_t s, const float* planes, int64_t plane, float scale, uint8_t* out) {
    if (plane == 0) return;
    ProfScope ps(s, "bgr_planes_to_rgb", 15.0 * (double)plane, 0.0);
    hipLaunchKernelGGL(bgr_planes_to_rgb_kernel, dim3(grid_for(plane)), dim3(256), 0, s, planes, (long)plane, scale, out);
}

}  // namespace pp
}  // namespace oar
