// kernels.hip -- hand-written gfx950 (CDNA4) kernels for the detector / recognizer networks.
//
// Data layout: feature maps are NHWC f32 in HBM (channel innermost => every conv reads/writes fully
// coalesced 16-byte vectors).  Dense convs / Linear layers live in igemm*.hip (implicit GEMM on the matrix
// cores); this file holds the bandwidth kernels -- depthwise, stem conv, pooling, resize, elementwise, the
// small batched GEMM / layernorm / softmax of the SVTR neck and the CTC tail -- all with float4 accesses.
// Wave = 64 lanes everywhere.  Compiled with -ffp-contract=off; FMAs are written explicitly.
#include "kernels.h"

#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels_dev.h"

namespace oar {
namespace k {

// ------------------------------------------------------------------------------------------ depthwise conv
// Bandwidth kernel. One thread = 4 channels (one float4) x TW adjacent output pixels of a row: each input row of
// the window is loaded once ((TW-1)*S + K float4s) and reused by all TW outputs, cutting the load count per output
// from K*K to K*((TW-1)*S+K)/TW.  Lanes run along channels first, so a wave reads/writes whole 16-byte-per-lane
// contiguous NHWC segments.  Weights [kh][kw][C].
template <int K, int SH, int S, int TW, int TH, bool GAP = false>
__global__ __launch_bounds__(256) void conv_dw_tiled_kernel(ConvP p) {
    // One thread = 4 channels x TH output rows x TW output columns.  Every input row of the (TH-1)*SH+K row window is
    // loaded once ((TW-1)*S+K float4s) and feeds every output row it overlaps; the K*K*C weights sit in LDS (their own
    // 128 B/clk pipe), so per output float4 the vector-memory path carries ((TH-1)*SH+K)*((TW-1)*S+K)/(TH*TW) loads
    // (6 for 5x5 s1) instead of the 16.25 of a one-row tile with weights from L1 -- the one-row version ran at ~60 %
    // of the L1 peak (the practical ceiling measured on the igemm kernels) and 2.5 TB/s of HBM traffic.
    extern __shared__ float4 dw_w[];   // [K*K][C4]
    const int C4 = p.Cout >> 2;
    for (int i = threadIdx.x; i < K * K * C4; i += blockDim.x) dw_w[i] = reinterpret_cast<const float4*>(p.w)[i];
    __syncthreads();
    const int wtiles = (p.Wo + TW - 1) / TW, htiles = (p.Ho + TH - 1) / TH;
    const long total = (long)p.N * htiles * wtiles * C4;
    constexpr int NCOL = (TW - 1) * S + K, NROW = (TH - 1) * SH + K;
    // each XCD (workgroup id % 8) walks one contiguous band of output rows => the row-halo re-reads stay in its L2
    const long per_xcd = (total + 7) / 8;
    const int xcd = blockIdx.x & 7;
    const long lb = blockIdx.x >> 3, nlb = (gridDim.x + 7) >> 3;
    const long band_end = min(total, (long)(xcd + 1) * per_xcd);
    for (long i = (long)xcd * per_xcd + lb * blockDim.x + threadIdx.x; i < band_end; i += nlb * blockDim.x) {
        int c4 = (int)(i % C4); long t = i / C4;
        int wt = (int)(t % wtiles); t /= wtiles;
        int ht = (int)(t % htiles); long n = t / htiles;
        const int c = c4 * 4, ow0 = wt * TW, oh0 = ht * TH;
        float4 bias = p.bias ? *reinterpret_cast<const float4*>(p.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 acc[TH][TW];
#pragma unroll
        for (int r = 0; r < TH; ++r)
#pragma unroll
            for (int q = 0; q < TW; ++q) acc[r][q] = bias;
        const float* xb = p.x + n * (long)p.H * p.W * p.Cin + c;
        const int iw0 = ow0 * S - p.pl, ih0 = oh0 * SH - p.pt;
#pragma unroll
        for (int r = 0; r < NROW; ++r) {
            const int ih = ih0 + r;
            if (ih < 0 || ih >= p.H) continue;
            const float* xr = xb + (long)ih * p.W * p.Cin;
            float4 col[NCOL];
#pragma unroll
            for (int q = 0; q < NCOL; ++q) {
                int iw = iw0 + q;
                col[q] = (iw >= 0 && iw < p.W) ? *reinterpret_cast<const float4*>(xr + (long)iw * p.Cin) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int tr = 0; tr < TH; ++tr) {
                const int a = r - tr * SH;       // kernel row of this input row for output row tr (compile-time)
                if (a < 0 || a >= K) continue;
#pragma unroll
                for (int b = 0; b < K; ++b) {
                    const float4 wv = dw_w[(a * K + b) * C4 + c4];
#pragma unroll
                    for (int q = 0; q < TW; ++q) {
                        float4 xv = col[q * S + b];
                        acc[tr][q].x = fmaf(xv.x, wv.x, acc[tr][q].x); acc[tr][q].y = fmaf(xv.y, wv.y, acc[tr][q].y);
                        acc[tr][q].z = fmaf(xv.z, wv.z, acc[tr][q].z); acc[tr][q].w = fmaf(xv.w, wv.w, acc[tr][q].w);
                    }
                }
            }
        }
        float4 tsum = make_float4(0.f, 0.f, 0.f, 0.f);   // GAP: this tile's sum of the activated outputs (rows, then columns: fixed order)
#pragma unroll
        for (int tr = 0; tr < TH; ++tr) {
            if (oh0 + tr >= p.Ho) break;
            const long pix0 = (n * p.Ho + oh0 + tr) * (long)p.Wo + ow0;
#pragma unroll
            for (int q = 0; q < TW; ++q) {
                if (ow0 + q >= p.Wo) break;
                long o = (pix0 + q) * p.y_ld + c;
                float4 v = acc[tr][q];
                if (p.residual) { float4 rr = *reinterpret_cast<const float4*>(p.residual + o); v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w; }
                v.x = apply_act(v.x, p.act.kind, p.act.alpha, p.act.beta); v.y = apply_act(v.y, p.act.kind, p.act.alpha, p.act.beta);
                v.z = apply_act(v.z, p.act.kind, p.act.alpha, p.act.beta); v.w = apply_act(v.w, p.act.kind, p.act.alpha, p.act.beta);
                *reinterpret_cast<float4*>(p.y + o) = v;
                if (GAP) { tsum.x += v.x; tsum.y += v.y; tsum.z += v.z; tsum.w += v.w; }
            }
        }
        if (GAP) reinterpret_cast<float4*>(p.gap_part)[((n * htiles + ht) * (long)wtiles + wt) * C4 + c4] = tsum;
    }
}

// generic depthwise (any kernel / stride / dilation): one thread = 4 channels x 1 output pixel
__global__ __launch_bounds__(256) void conv_dw_kernel(ConvP p) {
    const int C4 = p.Cout >> 2;
    long total = (long)p.N * p.Ho * p.Wo * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c4 = (int)(i % C4); long pix = i / C4;
        int ow = (int)(pix % p.Wo); long t = pix / p.Wo;
        int oh = (int)(t % p.Ho); long n = t / p.Ho;
        const int c = c4 * 4;
        float4 acc = p.bias ? *reinterpret_cast<const float4*>(p.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float* xb = p.x + n * (long)p.H * p.W * p.Cin + c;
        for (int a = 0; a < p.kh; ++a) {
            int ih = oh * p.sh - p.pt + a * p.dh;
            if (ih < 0 || ih >= p.H) continue;
            for (int b = 0; b < p.kw; ++b) {
                int iw = ow * p.sw - p.pl + b * p.dw;
                if (iw < 0 || iw >= p.W) continue;
                float4 xv = *reinterpret_cast<const float4*>(xb + ((long)ih * p.W + iw) * p.Cin);
                float4 wv = *reinterpret_cast<const float4*>(p.w + (long)(a * p.kw + b) * p.Cout + c);
                acc.x = fmaf(xv.x, wv.x, acc.x); acc.y = fmaf(xv.y, wv.y, acc.y);
                acc.z = fmaf(xv.z, wv.z, acc.z); acc.w = fmaf(xv.w, wv.w, acc.w);
            }
        }
        long o = pix * p.y_ld + c;
        if (p.residual) { float4 r = *reinterpret_cast<const float4*>(p.residual + o); acc.x += r.x; acc.y += r.y; acc.z += r.z; acc.w += r.w; }
        acc.x = apply_act(acc.x, p.act.kind, p.act.alpha, p.act.beta);
        acc.y = apply_act(acc.y, p.act.kind, p.act.alpha, p.act.beta);
        acc.z = apply_act(acc.z, p.act.kind, p.act.alpha, p.act.beta);
        acc.w = apply_act(acc.w, p.act.kind, p.act.alpha, p.act.beta);
        *reinterpret_cast<float4*>(p.y + o) = acc;
    }
}

static inline unsigned grid_for(long work, int block = 256, long cap = 256L * 32) {
    long g = (work + block - 1) / block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

namespace {
constexpr int kDwTW = 4;
// output rows per thread: the tallest tile that still leaves >= 4 resident workgroups' worth of threads per CU
int dw_tile_rows(const ConvP& p) {
    auto threads_for = [&](int th) { return (long)p.N * ((p.Ho + th - 1) / th) * ((p.Wo + kDwTW - 1) / kDwTW) * (p.Cout / 4); };
    return threads_for(2) >= 256L * 256 * 4 ? 2 : 1;   // (4 rows per thread measured slower on the 5x5 layers: 134 vs 112 us)
}
bool dw_tiled_shape(const ConvP& p) {
    const bool sq = p.kh == p.kw && p.dh == 1 && p.dw == 1;
    return sq && (p.kh == 3 || p.kh == 5) && (p.sh == 1 || p.sh == 2) && (p.sw == 1 || p.sw == 2) && (size_t)p.kh * p.kw * p.Cout * sizeof(float) <= 96 * 1024;
}
}  // namespace

// the pooled variant is instantiated for the 5 x 5 kernels (where PP-LCNet puts its squeeze-excite blocks)
int conv_dw_gap_tiles(const ConvP& p) {
    if (!dw_tiled_shape(p) || p.kh != 5 || (p.Cout & 3) || p.N <= 0 || p.Ho <= 0 || p.Wo <= 0) return 0;
    const int th = dw_tile_rows(p);
    return ((p.Ho + th - 1) / th) * ((p.Wo + kDwTW - 1) / kDwTW);
}

void conv_dw(hipStream_t s, const ConvP& p) {
    long total = (long)p.N * p.Ho * p.Wo * (p.Cout / 4);
    if (total == 0) return;
    OAR_CHECK(!p.gap_part || conv_dw_gap_tiles(p) > 0, OAR_INTERNAL, "conv_dw: pooled output on a shape without that variant");
    double bytes = 4.0 * ((double)p.N * p.H * p.W * p.Cin + (double)p.N * p.Ho * p.Wo * p.Cout + (double)p.kh * p.kw * p.Cout);
    double flops = 2.0 * (double)p.N * p.Ho * p.Wo * p.Cout * p.kh * p.kw;
    char pname[96];
    const char* cls = "conv_dw";
    if (Profiler::get().detail) { snprintf(pname, sizeof pname, "conv_dw px=%ld C=%d k%d s%d", (long)p.N * p.Ho * p.Wo, p.Cout, p.kh, p.sh); cls = pname; }
    ProfScope ps(s, cls, bytes, flops, /*single_launch=*/true);
    const bool sq = p.kh == p.kw && p.dh == 1 && p.dw == 1;
    constexpr int TW = kDwTW;
    auto threads_for = [&](int th) { return (long)p.N * ((p.Ho + th - 1) / th) * ((p.Wo + TW - 1) / TW) * (p.Cout / 4); };
    const int TH = dw_tile_rows(p);
    long tiled = threads_for(TH);
    dim3 g((grid_for(tiled) + 7) / 8 * 8), b(256);
    const size_t lds = (size_t)p.kh * p.kw * p.Cout * sizeof(float);
    // weights of all channels in LDS: up to 96 KB (C = 768 at 5 x 5 needs 77 KB -- beyond the default 64 KB limit the kernel is
    // opted in per instantiation; two workgroups per CU still beat the generic kernel's 1 TB/s by 3x)
    const bool fits = lds <= 96 * 1024;
#define DW2(KV, SHV, SWV, THV)                                                                                                                              \
    do {                                                                                                                                                    \
        if (lds > 64 * 1024) {                                                                                                                              \
            OAR_MAX_LDS_ONCE((conv_dw_tiled_kernel<KV, SHV, SWV, TW, THV>), 96 * 1024);                    \
        }                                                                                                                                                   \
        hipExtLaunchKernelGGL((conv_dw_tiled_kernel<KV, SHV, SWV, TW, THV>), g, b, lds, s, ps.start(), ps.stop(), 0, p);                                    \
    } while (0)
#define DW2G(KV, SHV, SWV, THV)                                                                                                                             \
    do {                                                                                                                                                    \
        if (lds > 64 * 1024) {                                                                                                                              \
            OAR_MAX_LDS_ONCE((conv_dw_tiled_kernel<KV, SHV, SWV, TW, THV, true>), 96 * 1024);                    \
        }                                                                                                                                                   \
        hipExtLaunchKernelGGL((conv_dw_tiled_kernel<KV, SHV, SWV, TW, THV, true>), g, b, lds, s, ps.start(), ps.stop(), 0, p);                              \
    } while (0)
#define DWG(KV, SHV, SWV)                                  \
    do {                                                   \
        if (TH == 2) DW2G(KV, SHV, SWV, 2);                \
        else DW2G(KV, SHV, SWV, 1);                        \
    } while (0)
#define DW(KV, SHV, SWV)                                   \
    do {                                                   \
        if (TH == 2) DW2(KV, SHV, SWV, 2);                 \
        else DW2(KV, SHV, SWV, 1);                         \
    } while (0)
    if (p.gap_part) {   // (conv_dw_gap_tiles said yes: 5 x 5, strides 1 / 2)
        if (p.sh == 1 && p.sw == 1) DWG(5, 1, 1);
        else if (p.sh == 2 && p.sw == 2) DWG(5, 2, 2);
        else if (p.sh == 2 && p.sw == 1) DWG(5, 2, 1);
        else DWG(5, 1, 2);
    } else
    if (fits && sq && p.kh == 3 && p.sh == 1 && p.sw == 1) DW(3, 1, 1);
    else if (fits && sq && p.kh == 3 && p.sh == 2 && p.sw == 2) DW(3, 2, 2);
    else if (fits && sq && p.kh == 3 && p.sh == 2 && p.sw == 1) DW(3, 2, 1);
    else if (fits && sq && p.kh == 3 && p.sh == 1 && p.sw == 2) DW(3, 1, 2);
    else if (fits && sq && p.kh == 5 && p.sh == 1 && p.sw == 1) DW(5, 1, 1);
    else if (fits && sq && p.kh == 5 && p.sh == 2 && p.sw == 2) DW(5, 2, 2);
    else if (fits && sq && p.kh == 5 && p.sh == 2 && p.sw == 1) DW(5, 2, 1);
    else if (fits && sq && p.kh == 5 && p.sh == 1 && p.sw == 2) DW(5, 1, 2);
    else hipExtLaunchKernelGGL(conv_dw_kernel, dim3(grid_for(total)), b, 0, s, ps.start(), ps.stop(), 0, p);
#undef DW2
#undef DW
#undef DW2G
#undef DWG
}

// ------------------------------------------------------------------------------------------ small-Cin conv (network stems)
// Cin <= 4 (RGB stems): K = kh*kw*Cin <= 128 is too shallow for the matrix cores to matter, the layer is bound by
// the output write.  One thread = one output pixel x 16 output channels; the [K][16] weight slice sits in LDS
// (broadcast reads), the 16 results leave as four 16-byte stores.  w: [kh][kw][Cin][Cout] (direct layout).
__global__ __launch_bounds__(256) void conv_smallcin_kernel(ConvP p) {
    __shared__ float ws[128 * 16];
    const int K = p.kh * p.kw * p.Cin;
    const int co0 = blockIdx.y * 16;
    for (int i = threadIdx.x; i < K * 16; i += 256) {
        int k = i >> 4, c = i & 15;
        ws[i] = (co0 + c < p.Cout) ? p.w[(long)k * p.Cout + co0 + c] : 0.f;
    }
    __syncthreads();
    const long total = (long)p.N * p.Ho * p.Wo;
    for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix < total; pix += (long)gridDim.x * blockDim.x) {
        int ow = (int)(pix % p.Wo); long t = pix / p.Wo;
        int oh = (int)(t % p.Ho); long n = t / p.Ho;
        float acc[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = (p.bias && co0 + c < p.Cout) ? p.bias[co0 + c] : 0.f;
        const float* xb = p.x + n * (long)p.H * p.W * p.Cin;
        for (int a = 0; a < p.kh; ++a) {
            int ih = oh * p.sh - p.pt + a * p.dh;
            if (ih < 0 || ih >= p.H) continue;
            for (int b = 0; b < p.kw; ++b) {
                int iw = ow * p.sw - p.pl + b * p.dw;
                if (iw < 0 || iw >= p.W) continue;
                const float* xp = xb + ((long)ih * p.W + iw) * p.Cin;
                const float* wp = ws + (a * p.kw + b) * p.Cin * 16;
                for (int ci = 0; ci < p.Cin; ++ci) {
                    float xv = xp[ci];
#pragma unroll
                    for (int c = 0; c < 16; ++c) acc[c] = fmaf(xv, wp[ci * 16 + c], acc[c]);
                }
            }
        }
        float* o = p.y + pix * p.y_ld + co0;
        const bool vec = (co0 + 16 <= p.Cout) && ((p.y_ld & 3) == 0);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (p.residual && co0 + c < p.Cout) acc[c] += p.residual[pix * p.y_ld + co0 + c];
            acc[c] = apply_act(acc[c], p.act.kind, p.act.alpha, p.act.beta);
        }
        if (vec) {
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(o + q * 4) = make_float4(acc[q * 4], acc[q * 4 + 1], acc[q * 4 + 2], acc[q * 4 + 3]);
        } else {
            for (int c = 0; c < 16 && co0 + c < p.Cout; ++c) o[c] = acc[c];
        }
    }
}

// The same kernel fed by u8 pages (StemU8, kernels.h): tap value = (float)byte * alpha + beta, two separate f32 operations
// exactly as pp::normalize computes them; out-of-image taps contribute nothing (the conv zero-pads the NORMALISED tensor).
// SW: the 16 output channels of the group all exist, so a tap's 16 weights are 64 contiguous bytes at a wave-uniform address -- they
// are read through the scalar cache straight into SGPRs (the FMAs take them as their scalar operand) instead of 108 broadcast
// ds_read_b128 per pixel, which kept the CU's one LDS pipe busier than its four SIMDs' FMAs.  Same products, same order.
typedef const __attribute__((address_space(4))) float sfloat;
template <bool CRNN, bool K3, bool SW>
__global__ __launch_bounds__(256) void conv_smallcin_u8_kernel(ConvP p, StemU8 st) {
    __shared__ float ws[SW ? 1 : 128 * 16];
    __shared__ float lut[CRNN ? 256 : 1];
    __shared__ float ot[4 * 64 * 17];   // per-wave output transposition slabs (see the store below)
    const int K = p.kh * p.kw * 3;
    const int co0 = blockIdx.y * 16;
    if (!SW)
        for (int i = threadIdx.x; i < K * 16; i += 256) {
            int k = i >> 4, c = i & 15;
            ws[i] = (co0 + c < p.Cout) ? p.w[(long)k * p.Cout + co0 + c] : 0.f;
        }
    sfloat* wsc = (sfloat*)(uintptr_t)(p.w + co0);
    if (CRNN) lut[threadIdx.x] = ((float)threadIdx.x / 255.0f - 0.5f) / 0.5f;   // pp::rec_pack's expression, once per byte value
    __syncthreads();
    const long per_image = (long)p.Ho * p.Wo;
    const float a0 = st.alpha[0], a1 = st.alpha[1], a2 = st.alpha[2], b0 = st.beta[0], b1 = st.beta[1], b2 = st.beta[2];
    const int s0 = st.src[0], s1 = st.src[1], s2 = st.src[2];
    // grid.z = image: the page pointer is wave-uniform (a per-thread index into the kernel-argument table would go through scratch)
    const int n = blockIdx.z;
    const uint8_t* __restrict__ pg = CRNN ? st.dev[n].ptr : st.pages[n];
    const int img_w = CRNN ? st.dev[n].w : p.W;   // the image's own row length: taps past it are padding
    // (the loop condition is the WAVE's first pixel: a wave stays together for the LDS transposition of its stores; lanes past the
    // image's last pixel compute the last pixel again and store nothing)
    for (long pix = (long)blockIdx.x * blockDim.x + threadIdx.x; pix - (threadIdx.x & 63) < per_image; pix += (long)gridDim.x * blockDim.x) {
        const bool live = pix < per_image;
        const long pixc = live ? pix : per_image - 1;
        const int ow = (int)(pixc % p.Wo), oh = (int)(pixc / p.Wo);
        float acc[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = (p.bias && co0 + c < p.Cout) ? p.bias[co0 + c] : 0.f;
        auto tap = [&](int a, int b, int kw) __attribute__((always_inline)) {
            const int ih = oh * p.sh - p.pt + a * p.dh, iw = ow * p.sw - p.pl + b * p.dw;
            if (ih < 0 || ih >= p.H || iw < 0 || iw >= img_w) return;
            const uint8_t* xp = pg + ((long)ih * img_w + iw) * 3;
            float x0, x1, x2;
            if (CRNN) {
                x0 = lut[xp[s0]]; x1 = lut[xp[s1]]; x2 = lut[xp[s2]];
            } else {
                const float t0 = (float)xp[s0] * a0, t1 = (float)xp[s1] * a1, t2 = (float)xp[s2] * a2;
                x0 = t0 + b0; x1 = t1 + b1; x2 = t2 + b2;
            }
            if (SW) {
                sfloat* wp = wsc + (long)(a * kw + b) * 3 * p.Cout;
#pragma unroll
                for (int c = 0; c < 16; ++c) acc[c] = fmaf(x0, wp[c], acc[c]);
#pragma unroll
                for (int c = 0; c < 16; ++c) acc[c] = fmaf(x1, wp[p.Cout + c], acc[c]);
#pragma unroll
                for (int c = 0; c < 16; ++c) acc[c] = fmaf(x2, wp[2 * p.Cout + c], acc[c]);
            } else {
                const float* wp = ws + (a * kw + b) * 3 * 16;
#pragma unroll
                for (int c = 0; c < 16; ++c) acc[c] = fmaf(x0, wp[c], acc[c]);
#pragma unroll
                for (int c = 0; c < 16; ++c) acc[c] = fmaf(x1, wp[16 + c], acc[c]);
#pragma unroll
                for (int c = 0; c < 16; ++c) acc[c] = fmaf(x2, wp[32 + c], acc[c]);
            }
        };
        if (K3) {   // the 3 x 3 stems of both networks: compile-time trip counts (the nine taps' loads can all be in flight; same order of the FMAs)
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) tap(a, b, 3);
        } else {
            for (int a = 0; a < p.kh; ++a)
                for (int b = 0; b < p.kw; ++b) tap(a, b, p.kw);
        }
        const long opix = (long)n * per_image + pixc;
        float* o = p.y + opix * p.y_ld + co0;
        const bool vec = (co0 + 16 <= p.Cout) && ((p.y_ld & 3) == 0);
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            if (p.residual && co0 + c < p.Cout) acc[c] += p.residual[opix * p.y_ld + co0 + c];
            acc[c] = apply_act(acc[c], p.act.kind, p.act.alpha, p.act.beta);
        }
        if (vec) {
            // A lane's 16 channels are 64 contiguous bytes, but lane-by-lane that is four store instructions of 16 bytes at a 64-byte
            // stride: every one of them touches all 32 cache lines of the wave's 4 KB partially.  The wave's 64 pixels are consecutive,
            // so its 256 float4s go through LDS (a wave's own 64 x 17-float slab: no barrier, the LDS unit serves a wave's accesses in
            // order) and leave as four fully coalesced 1 KB stores.
            float* slab = ot + (threadIdx.x >> 6) * (64 * 17);
            const int lane = threadIdx.x & 63;
#pragma unroll
            for (int c = 0; c < 16; ++c) slab[lane * 17 + c] = acc[c];
            __builtin_amdgcn_wave_barrier();
            const long pix_w0 = pix - lane;                               // the wave's first pixel (its lanes hold pix_w0 .. pix_w0 + 63)
            float* ow_ = p.y + ((long)n * per_image + pix_w0) * p.y_ld + co0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = j * 64 + lane, px = e >> 2, qd = e & 3;
                if (pix_w0 + px < per_image) {
                    const float* sp = slab + px * 17 + qd * 4;
                    *reinterpret_cast<float4*>(ow_ + (long)px * p.y_ld + qd * 4) = make_float4(sp[0], sp[1], sp[2], sp[3]);
                }
            }
            __builtin_amdgcn_wave_barrier();
        } else if (live) {
            for (int c = 0; c < 16 && co0 + c < p.Cout; ++c) o[c] = acc[c];
        }
    }
}
void conv_smallcin_u8(hipStream_t s, const ConvP& p, const StemU8& st) {
    const long per_image = (long)p.Ho * p.Wo;
    if (per_image == 0 || p.N == 0) return;
    OAR_CHECK(p.groups == 1 && p.Cin == 3 && p.kh * p.kw * 3 <= 128 && (st.dev ? p.N <= 65535 : p.N <= 32), OAR_INTERNAL, "conv_smallcin_u8: not an RGB stem");
    const double total = (double)p.N * per_image * p.Cout;
    ProfScope ps(s, "conv_smallcin", 3.0 * (double)p.N * p.H * p.W + 4.0 * total, 2.0 * total * p.kh * p.kw * 3);
    const dim3 grid(grid_for(per_image, 256, 256L * 4), (p.Cout + 15) / 16, p.N);
    const bool k3 = p.kh == 3 && p.kw == 3;
    static const bool sw_off = [] { const char* e = getenv("OAR_STEM_SW"); return e && e[0] == '0'; }();
    const bool sw = k3 && p.Cout % 16 == 0 && !sw_off;   // (Cout % 16: every group is full and its 64 weight bytes are dword aligned)
    if (st.dev) {
        if (sw) hipLaunchKernelGGL((conv_smallcin_u8_kernel<true, true, true>), grid, dim3(256), 0, s, p, st);
        else if (k3) hipLaunchKernelGGL((conv_smallcin_u8_kernel<true, true, false>), grid, dim3(256), 0, s, p, st);
        else hipLaunchKernelGGL((conv_smallcin_u8_kernel<true, false, false>), grid, dim3(256), 0, s, p, st);
    } else {
        if (sw) hipLaunchKernelGGL((conv_smallcin_u8_kernel<false, true, true>), grid, dim3(256), 0, s, p, st);
        else if (k3) hipLaunchKernelGGL((conv_smallcin_u8_kernel<false, true, false>), grid, dim3(256), 0, s, p, st);
        else hipLaunchKernelGGL((conv_smallcin_u8_kernel<false, false, false>), grid, dim3(256), 0, s, p, st);
    }
}

// ------------------------------------------------------------------------------------------ direct conv (fallback)
// One thread = one output element. w: [kh][kw][Cin/g][Cout].
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvP p) {
    long total = (long)p.N * p.Ho * p.Wo * p.Cout;
    const int cpg = p.Cin / p.groups, opg = p.Cout / p.groups;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int co = (int)(i % p.Cout); long pix = i / p.Cout;
        int ow = (int)(pix % p.Wo); long t = pix / p.Wo;
        int oh = (int)(t % p.Ho); long n = t / p.Ho;
        int g = co / opg;
        float acc = p.bias ? p.bias[co] : 0.f;
        const float* xb = p.x + n * (long)p.H * p.W * p.Cin + (long)g * cpg;
        for (int a = 0; a < p.kh; ++a) {
            int ih = oh * p.sh - p.pt + a * p.dh;
            if (ih < 0 || ih >= p.H) continue;
            for (int b = 0; b < p.kw; ++b) {
                int iw = ow * p.sw - p.pl + b * p.dw;
                if (iw < 0 || iw >= p.W) continue;
                const float* xp = xb + ((long)ih * p.W + iw) * p.Cin;
                const float* wp = p.w + ((long)(a * p.kw + b) * cpg) * p.Cout + co;
                for (int ci = 0; ci < cpg; ++ci) acc = fmaf(xp[ci], wp[(long)ci * p.Cout], acc);
            }
        }
        long o = pix * p.y_ld + co;
        if (p.residual) acc += p.residual[o];
        p.y[o] = apply_act(acc, p.act.kind, p.act.alpha, p.act.beta);
    }
}
void conv_direct(hipStream_t s, const ConvP& p) {
    long total = (long)p.N * p.Ho * p.Wo * p.Cout;
    if (total == 0) return;
    double bytes = 4.0 * ((double)p.N * p.H * p.W * p.Cin + (double)total + (double)p.kh * p.kw * (p.Cin / p.groups) * p.Cout);
    double flops = 2.0 * (double)total * p.kh * p.kw * (p.Cin / p.groups);
    if (p.groups == 1 && p.Cin <= 4 && p.kh * p.kw * p.Cin <= 128) {
        ProfScope ps(s, "conv_smallcin", bytes, flops);
        long pixels = (long)p.N * p.Ho * p.Wo;
        hipLaunchKernelGGL(conv_smallcin_kernel, dim3(grid_for(pixels, 256, 256L * 16), (p.Cout + 15) / 16), dim3(256), 0, s, p);
        return;
    }
    ProfScope ps(s, "conv_direct", bytes, flops);
    hipLaunchKernelGGL(conv_direct_kernel, dim3(grid_for(total)), dim3(256), 0, s, p);
}

// ------------------------------------------------------------------------------------------ ConvTranspose 2x2 s2, twice (DB head tail)
// One thread = one input pixel x one position (a, b) of the first layer: C1 mid values in registers (bias, then ci ascending),
// activation, then the 2 x 2 block of the second layer for each of its C2 channels (bias, then cm ascending).  Both weight sets sit
// in LDS.  Consecutive lanes walk (b, then w): a wave writes whole contiguous runs of two output rows.
template <int C1>
__global__ __launch_bounds__(256) void convt2x2_pair_kernel(ConvT2Pair p) {
    extern __shared__ float4 cp_lds[];
    float4* w1 = cp_lds;                                                  // [C0][4][C1 / 4]
    float* w2 = reinterpret_cast<float*>(cp_lds + (long)p.C0 * C1);       // [C1][4][C2]
    for (int i = threadIdx.x; i < p.C0 * C1; i += blockDim.x) w1[i] = reinterpret_cast<const float4*>(p.w1)[i];
    for (int i = threadIdx.x; i < C1 * 4 * p.C2; i += blockDim.x) w2[i] = p.w2[i];
    __syncthreads();
    const long total = (long)p.N * p.H * p.W * 4;
    const int Wo = p.W * 4, Ho = p.H * 4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        // i = ((n * H + h) * 2 + a) * (2 W) + w * 2 + b : lanes sweep one row of first-layer outputs
        const int wb = (int)(i % (2 * p.W));
        const long t = i / (2 * p.W);
        const int a = (int)(t & 1);
        const long nh = t >> 1;
        const int h = (int)(nh % p.H);
        const long n = nh / p.H;
        const int w = wb >> 1, b = wb & 1, pos = a * 2 + b;
        const float* xp = p.x + ((n * p.H + h) * (long)p.W + w) * p.C0;
        float mid[C1];
#pragma unroll
        for (int c = 0; c < C1; ++c) mid[c] = p.b1 ? p.b1[c] : 0.f;
        for (int c4 = 0; c4 < p.C0; c4 += 4) {
            const float4 xv = *reinterpret_cast<const float4*>(xp + c4);
            const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float4* wr = w1 + ((long)(c4 + e) * 4 + pos) * (C1 / 4);
#pragma unroll
                for (int q = 0; q < C1 / 4; ++q) {
                    const float4 wv = wr[q];
                    mid[q * 4 + 0] = fmaf(xs[e], wv.x, mid[q * 4 + 0]); mid[q * 4 + 1] = fmaf(xs[e], wv.y, mid[q * 4 + 1]);
                    mid[q * 4 + 2] = fmaf(xs[e], wv.z, mid[q * 4 + 2]); mid[q * 4 + 3] = fmaf(xs[e], wv.w, mid[q * 4 + 3]);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < C1; ++c) mid[c] = apply_act(mid[c], p.act1.kind, p.act1.alpha, p.act1.beta);
        // second layer: this thread's first-layer pixel is (2h + a, 2w + b); its 2 x 2 block starts at (4h + 2a, 4w + 2b)
        const long orow = (n * Ho + 4 * h + 2 * a) * (long)Wo + 4 * w + 2 * b;
#pragma unroll
        for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
            for (int b2 = 0; b2 < 2; ++b2)
                for (int co = 0; co < p.C2; ++co) {
                    float acc = p.b2 ? p.b2[co] : 0.f;
#pragma unroll
                    for (int c = 0; c < C1; ++c) acc = fmaf(mid[c], w2[(c * 4 + a2 * 2 + b2) * p.C2 + co], acc);
                    p.y[(orow + (long)a2 * Wo + b2) * p.C2 + co] = apply_act(acc, p.act2.kind, p.act2.alpha, p.act2.beta);
                }
    }
}
bool convt2x2_pair_supported(int C0, int C1, int C2) {
    return C0 > 0 && (C0 & 3) == 0 && (C1 == 8 || C1 == 16 || C1 == 24 || C1 == 32) && C2 >= 1 && C2 <= 4 && (size_t)C0 * C1 * 16 + (size_t)C1 * 4 * C2 * 4 <= 60 * 1024;
}
void convt2x2_pair(hipStream_t s, const ConvT2Pair& p) {
    const long total = (long)p.N * p.H * p.W * 4;
    if (total == 0) return;
    OAR_CHECK(convt2x2_pair_supported(p.C0, p.C1, p.C2), OAR_INTERNAL, "convt2x2_pair: unsupported channel counts");
    const size_t lds = (size_t)p.C0 * p.C1 * 16 + (size_t)p.C1 * 4 * p.C2 * 4;
    const double px = (double)p.N * p.H * p.W;
    ProfScope ps(s, "convt_pair", 4.0 * px * (p.C0 + 16.0 * p.C2), 2.0 * px * 4.0 * ((double)p.C0 * p.C1 + 4.0 * p.C1 * p.C2));
    const dim3 grid(grid_for(total, 256, 256L * 8));
    switch (p.C1) {
        case 8: hipLaunchKernelGGL(convt2x2_pair_kernel<8>, grid, dim3(256), lds, s, p); break;
        case 16: hipLaunchKernelGGL(convt2x2_pair_kernel<16>, grid, dim3(256), lds, s, p); break;
        case 24: hipLaunchKernelGGL(convt2x2_pair_kernel<24>, grid, dim3(256), lds, s, p); break;
        default: hipLaunchKernelGGL(convt2x2_pair_kernel<32>, grid, dim3(256), lds, s, p); break;
    }
}

// General ConvTranspose, gather form: y[n,oh,ow,co] = sum_{a,b,ci} x[n,(oh+pt-a*dh)/sh,(ow+pl-b*dw)/sw,ci] * w[a][b][ci][co]
__global__ __launch_bounds__(256) void convt_direct_kernel(ConvP p) {
    long total = (long)p.N * p.Ho * p.Wo * p.Cout;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int co = (int)(i % p.Cout); long pix = i / p.Cout;
        int ow = (int)(pix % p.Wo); long t = pix / p.Wo;
        int oh = (int)(t % p.Ho); long n = t / p.Ho;
        float acc = p.bias ? p.bias[co] : 0.f;
        for (int a = 0; a < p.kh; ++a) {
            int th = oh + p.pt - a * p.dh;
            if (th < 0 || th % p.sh) continue;
            int ih = th / p.sh;
            if (ih >= p.H) continue;
            for (int b = 0; b < p.kw; ++b) {
                int tw = ow + p.pl - b * p.dw;
                if (tw < 0 || tw % p.sw) continue;
                int iw = tw / p.sw;
                if (iw >= p.W) continue;
                const float* xp = p.x + ((n * p.H + ih) * (long)p.W + iw) * p.Cin;
                const float* wp = p.w + ((long)(a * p.kw + b) * p.Cin) * p.Cout + co;
                for (int ci = 0; ci < p.Cin; ++ci) acc = fmaf(xp[ci], wp[(long)ci * p.Cout], acc);
            }
        }
        p.y[pix * p.y_ld + co] = apply_act(acc, p.act.kind, p.act.alpha, p.act.beta);
    }
}
void convt_direct(hipStream_t s, const ConvP& p) {
    long total = (long)p.N * p.Ho * p.Wo * p.Cout;
    if (total == 0) return;
    ProfScope ps(s, "convt_direct", 4.0 * ((double)p.N * p.H * p.W * p.Cin + (double)total), 2.0 * (double)p.N * p.H * p.W * p.Cin * p.Cout * p.kh * p.kw);
    hipLaunchKernelGGL(convt_direct_kernel, dim3(grid_for(total)), dim3(256), 0, s, p);
}

// ------------------------------------------------------------------------------------------ pooling
__global__ __launch_bounds__(256) void pool2d_kernel(PoolP p) {
    long total = (long)p.N * p.Ho * p.Wo * p.C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int c = (int)(i % p.C); long pix = i / p.C;
        int ow = (int)(pix % p.Wo); long t = pix / p.Wo;
        int oh = (int)(t % p.Ho); long n = t / p.Ho;
        float acc = p.is_max ? -3.402823466e38f : 0.f;
        int cnt = 0;
        for (int a = 0; a < p.kh; ++a) {
            int ih = oh * p.sh - p.pt + a;
            if (ih < 0 || ih >= p.H) continue;
            for (int b = 0; b < p.kw; ++b) {
                int iw = ow * p.sw - p.pl + b;
                if (iw < 0 || iw >= p.W) continue;
                float v = p.x[((n * p.H + ih) * (long)p.W + iw) * p.C + c];
                if (p.is_max) acc = fmaxf(acc, v); else acc += v;
                ++cnt;
            }
        }
        if (!p.is_max) {
            // count_include_pad: the window clipped to the PADDED extent (a ceil_mode window hanging over the padding does not count the overhang: ONNX AveragePool
            // since opset 19 / torch avg_pool2d; without ceil_mode every window lies inside the padded extent and this is kh * kw)
            const int ih0 = oh * p.sh - p.pt, iw0 = ow * p.sw - p.pl;
            const int full = (min(ih0 + p.kh, p.H + p.pb) - ih0) * (min(iw0 + p.kw, p.W + p.pr) - iw0);
            acc = acc / (float)(p.count_include_pad ? (full > 0 ? full : 1) : (cnt > 0 ? cnt : 1));
        }
        p.y[i] = acc;
    }
}
void pool2d(hipStream_t s, const PoolP& p) {
    long total = (long)p.N * p.Ho * p.Wo * p.C;
    if (total == 0) return;
    ProfScope ps(s, "pool2d", 4.0 * ((double)p.N * p.H * p.W * p.C + (double)total), 0.0);
    hipLaunchKernelGGL(pool2d_kernel, dim3(grid_for(total)), dim3(256), 0, s, p);
}

// One workgroup per (image, pixel split). Lanes run along channels (float4), `parts` thread groups stride over the split's
// pixels; the partial sums are combined through LDS in a fixed order, and -- when an image is cut into several splits so
// that a 9-image detector sub-batch still fills the GPU -- by a second launch of the same kernel over the [splits][C]
// partial sums (fixed order again: deterministic).  scale = 1 / HW on the final pass, 1 on the partial pass.
__global__ __launch_bounds__(256) void global_avgpool_kernel(const float* x, float* y, int HW, int C, int splits, float div) {
    extern __shared__ float4 red[];  // [parts][C4]
    const int C4 = C >> 2;
    const int parts = 256 / C4 > 0 ? 256 / C4 : 1;
    const int n = blockIdx.x / splits, sp = blockIdx.x - n * splits;
    const int per = (HW + splits - 1) / splits, p0 = sp * per, p1 = min(HW, p0 + per);
    const int c4 = threadIdx.x % C4, part = threadIdx.x / C4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (part < parts) {
        const float4* xb = reinterpret_cast<const float4*>(x + (long)n * HW * C) + c4;
        int i = p0 + part;
        for (; i + 3 * parts < p1; i += 4 * parts) {
            float4 a = xb[(long)i * C4], b = xb[(long)(i + parts) * C4], c = xb[(long)(i + 2 * parts) * C4], d = xb[(long)(i + 3 * parts) * C4];
            acc.x += (a.x + b.x) + (c.x + d.x); acc.y += (a.y + b.y) + (c.y + d.y);
            acc.z += (a.z + b.z) + (c.z + d.z); acc.w += (a.w + b.w) + (c.w + d.w);
        }
        for (; i < p1; i += parts) { float4 a = xb[(long)i * C4]; acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w; }
        red[part * C4 + c4] = acc;
    }
    __syncthreads();
    if (part == 0) {
        float4 t = red[c4];
        for (int q = 1; q < parts; ++q) { float4 u = red[q * C4 + c4]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        reinterpret_cast<float4*>(y + (long)blockIdx.x * C)[c4] = make_float4(t.x / div, t.y / div, t.z / div, t.w / div);
    }
}
__global__ __launch_bounds__(256) void global_avgpool_scalar_kernel(const float* x, float* y, int HW, int C) {
    const int n = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float* xb = x + (long)n * HW * C + c;
    float acc = 0.f;
    for (int i = 0; i < HW; ++i) acc += xb[(long)i * C];
    y[(long)n * C + c] = acc / (float)HW;
}
int global_avgpool_splits(int N, int HW, int C) {
    if ((C & 3) != 0 || C / 4 > 256 || N <= 0) return 1;
    // enough workgroups for the 256 CUs, at least 256 pixels per split
    int s = (1024 + N - 1) / N;
    s = std::min(s, std::max(HW / 256, 1));
    return std::max(1, std::min(s, 256));
}
void global_avgpool(hipStream_t s, const float* x, float* y, int N, int HW, int C, float* partial) {
    if (N == 0 || C == 0) return;
    char pname[64];
    const char* cls = "global_avgpool";
    if (Profiler::get().detail) { snprintf(pname, sizeof pname, "global_avgpool N=%d HW=%d C=%d", N, HW, C); cls = pname; }
    ProfScope ps(s, cls, 4.0 * (double)N * HW * C, 0.0);
    if ((C & 3) == 0 && C / 4 <= 256) {
        const int C4 = C / 4, parts = 256 / C4;
        const size_t lds = (size_t)parts * C4 * sizeof(float4);
        const int splits = partial ? global_avgpool_splits(N, HW, C) : 1;
        if (splits == 1) {
            hipLaunchKernelGGL(global_avgpool_kernel, dim3(N), dim3(256), lds, s, x, y, HW, C, 1, (float)HW);
        } else {
            hipLaunchKernelGGL(global_avgpool_kernel, dim3(N * splits), dim3(256), lds, s, x, partial, HW, C, splits, 1.0f);
            hipLaunchKernelGGL(global_avgpool_kernel, dim3(N), dim3(256), lds, s, (const float*)partial, y, splits, C, 1, (float)HW);
        }
    } else {
        hipLaunchKernelGGL(global_avgpool_scalar_kernel, dim3((C + 255) / 256, N), dim3(256), 0, s, x, y, HW, C);
    }
}

void global_avgpool_finish(hipStream_t s, const float* part, float* y, int N, int tiles, int C, int hw) {
    if (N == 0 || C == 0) return;
    OAR_CHECK((C & 3) == 0 && C <= 1024 && tiles > 0, OAR_INTERNAL, "global_avgpool_finish: channel count");
    ProfScope ps(s, "global_avgpool", 4.0 * ((double)N * tiles * C + (double)N * C), (double)N * tiles * C);
    const int C4 = C >> 2, parts = std::max(1, 256 / C4);
    hipLaunchKernelGGL(global_avgpool_kernel, dim3(N), dim3(256), (size_t)parts * C4 * sizeof(float4), s, part, y, tiles, C, 1, (float)hw);
}

// ------------------------------------------------------------------------------------------ resize
__host__ __device__ __forceinline__ float src_coord(int o, float scale, int in, int out, int ctm) {
    switch (ctm) {
        case 0: return (float)o / scale;
        case 2: return out > 1 ? (float)o * (float)(in - 1) / (float)(out - 1) : 0.f;
        case 3: return out > 1 ? ((float)o + 0.5f) / scale - 0.5f : 0.f;
        default: return ((float)o + 0.5f) / scale - 0.5f;
    }
}
// source index of output index o under mode = nearest (the planner evaluates the same function to recognise integer-factor maps)
__host__ __device__ __forceinline__ int nearest_src(int o, float scale, int in, int out, int ctm, int nm) {
    const float f = src_coord(o, scale, in, out, ctm);
    float r;
    switch (nm) {
        case 0: r = floorf(f); break;
        case 3: r = ceilf(f); break;
        case 2: r = floorf(f + 0.5f); break;
        default: r = ceilf(f - 0.5f); break;
    }
    const int i = (int)r;
    return i < 0 ? 0 : i > in - 1 ? in - 1 : i;
}
int resize_nearest_index(int o, float scale, int in, int out, int ctm, int nearest_mode) { return nearest_src(o, scale, in, out, ctm, nearest_mode); }
__global__ __launch_bounds__(256) void resize_kernel(const float* x, float* y, int N, int H, int W, int C, int Ho, int Wo,
                                                     float sh, float sw, int mode, int ctm, int nm, int y_ld) {
    const int C4 = C >> 2;  // launcher guarantees C % 4 == 0 for the vector path, else C4 == 0 and scalar path
    const bool vec = (C & 3) == 0 && (y_ld & 3) == 0;
    const int cw = vec ? C4 : C;
    long total = (long)N * Ho * Wo * cw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        int cc = (int)(i % cw); long pix = i / cw;
        int ow = (int)(pix % Wo); long t = pix / Wo;
        int oh = (int)(t % Ho); long n = t / Ho;
        float fy = src_coord(oh, sh, H, Ho, ctm), fx = src_coord(ow, sw, W, Wo, ctm);
        const float* xb = x + n * (long)H * W * C;
        if (mode == 0) {
            const int iy = nearest_src(oh, sh, H, Ho, ctm, nm), ix = nearest_src(ow, sw, W, Wo, ctm, nm);
            if (vec) *reinterpret_cast<float4*>(y + pix * y_ld + cc * 4) = *reinterpret_cast<const float4*>(xb + ((long)iy * W + ix) * C + cc * 4);
            else y[pix * y_ld + cc] = xb[((long)iy * W + ix) * C + cc];
        } else {
            fy = fminf(fmaxf(fy, 0.f), (float)(H - 1)); fx = fminf(fmaxf(fx, 0.f), (float)(W - 1));
            int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
            int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
            float wy = fy - (float)y0, wx = fx - (float)x0;
            int nc = vec ? 4 : 1;
            for (int q = 0; q < nc; ++q) {
                int c = vec ? cc * 4 + q : cc;
                float v00 = xb[((long)y0 * W + x0) * C + c], v01 = xb[((long)y0 * W + x1) * C + c];
                float v10 = xb[((long)y1 * W + x0) * C + c], v11 = xb[((long)y1 * W + x1) * C + c];
                float top = v00 + (v01 - v00) * wx, bot = v10 + (v11 - v10) * wx;
                y[pix * y_ld + c] = top + (bot - top) * wy;
            }
        }
    }
}
void resize(hipStream_t s, const float* x, float* y, int N, int H, int W, int C, int Ho, int Wo, float scale_h,
            float scale_w, int mode, int ctm, int nearest_mode, int y_ld) {
    long total = (long)N * Ho * Wo * C;
    if (total == 0) return;
    ProfScope ps(s, "resize", 4.0 * ((double)N * H * W * C + (double)total), 0.0);
    long work = ((C & 3) == 0 && (y_ld & 3) == 0) ? total / 4 : total;
    hipLaunchKernelGGL(resize_kernel, dim3(grid_for(work)), dim3(256), 0, s, x, y, N, H, W, C, Ho, Wo, scale_h, scale_w, mode, ctm, nearest_mode, y_ld);
}

__global__ __launch_bounds__(256) void concat_gather_kernel(ConcatGatherP p, float* __restrict__ y) {
    const int C4 = p.C >> 2;
    const long total = (long)p.N * p.Ho * p.Wo * C4;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const long pix = i / C4;
        const int ow = (int)(pix % p.Wo);
        const long t = pix / p.Wo;
        const int oh = (int)(t % p.Ho);
        const long n = t / p.Ho;
        int sidx = 0;
#pragma unroll
        for (int q = 1; q < 8; ++q) if (q < p.n_src && c4 * 4 >= p.off[q]) sidx = q;
        const int cs = p.c[sidx], fh = p.fh[sidx], fw = p.fw[sidx];
        const int Hs = p.Ho / fh, Ws = p.Wo / fw;
        const float* src = p.x[sidx] + ((n * Hs + oh / fh) * (long)Ws + ow / fw) * cs + (c4 * 4 - p.off[sidx]);
        reinterpret_cast<float4*>(y)[i] = *reinterpret_cast<const float4*>(src);
    }
}
void concat_gather(hipStream_t s, const ConcatGatherP& p, float* y) {
    const long total = (long)p.N * p.Ho * p.Wo * p.C;
    if (total == 0) return;
    OAR_CHECK(p.n_src >= 1 && p.n_src <= 8 && (p.C & 3) == 0, OAR_INTERNAL, "concat_gather: sources / channels");
    double rd = 0;
    for (int i = 0; i < p.n_src; ++i) rd += (double)p.N * (p.Ho / p.fh[i]) * (p.Wo / p.fw[i]) * p.c[i];
    ProfScope ps(s, "resize", 4.0 * (rd + (double)total), 0.0);
    hipLaunchKernelGGL(concat_gather_kernel, dim3(grid_for(total / 4)), dim3(256), 0, s, p, y);
}

// ------------------------------------------------------------------------------------------ elementwise
__global__ __launch_bounds__(256) void unary_kernel(const float* x, float* y, long n, int kind, float alpha, float beta) {
    long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<const float4*>(x)[i];
        v.x = apply_unary(v.x, kind, alpha, beta); v.y = apply_unary(v.y, kind, alpha, beta);
        v.z = apply_unary(v.z, kind, alpha, beta); v.w = apply_unary(v.w, kind, alpha, beta);
        reinterpret_cast<float4*>(y)[i] = v;
    }
    for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        y[i] = apply_unary(x[i], kind, alpha, beta);
}
void unary(hipStream_t s, const float* x, float* y, int64_t n, Act act) {
    if (n == 0) return;
    ProfScope ps(s, "unary", 8.0 * (double)n, 0.0);
    hipLaunchKernelGGL(unary_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, s, x, y, (long)n, act.kind, act.alpha, act.beta);
}

struct BinP {
    int rank, op;
    long dims[6], sa[6], sb[6];
    int akind;
    float alpha, beta;
};
__device__ __forceinline__ float bin_op(float a, float b, int op) {
    switch (op) {
        case 0: return a + b;
        case 1: return a - b;
        case 2: return a * b;
        case 3: return a / b;
        case 5: return a > 0.f ? a : a * b;   // PRelu(x, slope)
        case 6: return fmaxf(a, b);
        case 7: return fminf(a, b);
        case 8: return a == b ? 1.0f : 0.0f;
        case 9: return a < b ? 1.0f : 0.0f;
        case 10: return a > b ? 1.0f : 0.0f;
        case 11: return (a != 0.0f && b != 0.0f) ? 1.0f : 0.0f;
        case 12: return (a != 0.0f || b != 0.0f) ? 1.0f : 0.0f;
        default: return powf(a, b);
    }
}
__global__ __launch_bounds__(256) void binary_kernel(const float* a, const float* b, float* y, long n, BinP p) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        long r = i, oa = 0, ob = 0;
#pragma unroll
        for (int d = 5; d >= 0; --d) {
            if (d < p.rank) {
                long q = r / p.dims[d]; long idx = r - q * p.dims[d]; r = q;
                oa += idx * p.sa[d]; ob += idx * p.sb[d];
            }
        }
        y[i] = apply_act(bin_op(a[oa], b[ob], p.op), p.akind, p.alpha, p.beta);
    }
}
// fast paths: same shape (flat float4), and "inner broadcast" y[r][c] = a[r][c] op b[r / rep][c]-style handled by generic.
__global__ __launch_bounds__(256) void binary_flat_kernel(const float* a, const float* b, float* y, long n, int op, int akind, float alpha, float beta) {
    long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 u = reinterpret_cast<const float4*>(a)[i], v = reinterpret_cast<const float4*>(b)[i], o;
        o.x = apply_act(bin_op(u.x, v.x, op), akind, alpha, beta); o.y = apply_act(bin_op(u.y, v.y, op), akind, alpha, beta);
        o.z = apply_act(bin_op(u.z, v.z, op), akind, alpha, beta); o.w = apply_act(bin_op(u.w, v.w, op), akind, alpha, beta);
        reinterpret_cast<float4*>(y)[i] = o;
    }
    for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        y[i] = apply_act(bin_op(a[i], b[i], op), akind, alpha, beta);
}
// y[n][hw][c] = a[n][hw][c] op b[n or 0][c]  (per-channel / SE broadcast on NHWC); C % 4 == 0
__global__ __launch_bounds__(256) void binary_chan_kernel(const float* a, const float* b, float* y, long total4, int C4, long hw, long b_nstride4, int op, int akind, float alpha, float beta) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        int c4 = (int)(i % C4); long pix = i / C4; long n = pix / hw;
        float4 u = reinterpret_cast<const float4*>(a)[i], v = reinterpret_cast<const float4*>(b)[n * b_nstride4 + c4], o;
        o.x = apply_act(bin_op(u.x, v.x, op), akind, alpha, beta); o.y = apply_act(bin_op(u.y, v.y, op), akind, alpha, beta);
        o.z = apply_act(bin_op(u.z, v.z, op), akind, alpha, beta); o.w = apply_act(bin_op(u.w, v.w, op), akind, alpha, beta);
        reinterpret_cast<float4*>(y)[i] = o;
    }
}
void binary(hipStream_t s, const float* a, const float* b, float* y, int op, int rank, const int64_t* dims,
            const int64_t* sa, const int64_t* sb, Act post) {
    long n = 1;
    for (int i = 0; i < rank; ++i) n *= dims[i];
    if (n == 0) return;
    char pname[64];
    const char* cls = "binary";
    if (Profiler::get().detail) { snprintf(pname, sizeof pname, "binary op%d n=%ld last=%ld", op, n, (long)dims[rank - 1]); cls = pname; }
    ProfScope ps(s, cls, 12.0 * (double)n, (double)n);
    // contiguity analysis
    bool a_full = true, b_full = true;
    long exp = 1;
    for (int d = rank - 1; d >= 0; --d) {
        if (dims[d] != 1) {
            if (sa[d] != exp) a_full = false;
            if (sb[d] != exp) b_full = false;
        }
        exp *= dims[d];
    }
    if (a_full && b_full) {
        hipLaunchKernelGGL(binary_flat_kernel, dim3(grid_for(n / 4 + 1)), dim3(256), 0, s, a, b, y, n, op, post.kind, post.alpha, post.beta);
        return;
    }
    // channel broadcast: last dim C contiguous in both, b broadcast over all middle dims, optional leading batch
    if (a_full && rank >= 2 && (dims[rank - 1] & 3) == 0 && sb[rank - 1] == 1) {
        bool mid_bcast = true;
        for (int d = 1; d < rank - 1; ++d) if (dims[d] != 1 && sb[d] != 0) mid_bcast = false;
        if (mid_bcast && (sb[0] == 0 || sb[0] == dims[rank - 1] || dims[0] == 1)) {
            long C = dims[rank - 1], hw = 1;
            for (int d = 1; d < rank - 1; ++d) hw *= dims[d];
            long bn = (dims[0] == 1) ? 0 : sb[0];
            hipLaunchKernelGGL(binary_chan_kernel, dim3(grid_for(n / 4)), dim3(256), 0, s, a, b, y, n / 4, (int)(C / 4), hw, bn / 4, op, post.kind, post.alpha, post.beta);
            return;
        }
    }
    OAR_CHECK(rank <= 6, OAR_UNSUPPORTED_OP, "binary: rank > 6");
    BinP p;
    p.rank = rank; p.op = op; p.akind = post.kind; p.alpha = post.alpha; p.beta = post.beta;
    for (int i = 0; i < 6; ++i) { p.dims[i] = i < rank ? dims[i] : 1; p.sa[i] = i < rank ? sa[i] : 0; p.sb[i] = i < rank ? sb[i] : 0; }
    hipLaunchKernelGGL(binary_kernel, dim3(grid_for(n)), dim3(256), 0, s, a, b, y, n, p);
}

// y = a op nearest_upsample(b) on NHWC with integer factors: b is [N, Ho / fh, Wo / fw, C], read through the index map (FPN
// top-down sums: the upsampled tensor never exists); C % 4 == 0
__global__ __launch_bounds__(256) void binary_upsampled_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ y, long total4, int C4,
                                                                int Ho, int Wo, int fh, int fw, int op) {
    const int Wi = Wo / fw, Hi = Ho / fh;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        const long pix = i / C4;
        const int ow = (int)(pix % Wo);
        const long t = pix / Wo;
        const int oh = (int)(t % Ho);
        const long n = t / Ho;
        const float4 u = reinterpret_cast<const float4*>(a)[i];
        const float4 v = reinterpret_cast<const float4*>(b)[((n * Hi + oh / fh) * Wi + ow / fw) * C4 + c4];
        float4 o;
        o.x = bin_op(u.x, v.x, op); o.y = bin_op(u.y, v.y, op); o.z = bin_op(u.z, v.z, op); o.w = bin_op(u.w, v.w, op);
        reinterpret_cast<float4*>(y)[i] = o;
    }
}
void binary_upsampled(hipStream_t s, const float* a, const float* b, float* y, int N, int Ho, int Wo, int C, int fh, int fw, int op) {
    const long total4 = (long)N * Ho * Wo * (C / 4);
    if (total4 == 0) return;
    OAR_CHECK((C & 3) == 0 && fh > 0 && fw > 0 && Ho % fh == 0 && Wo % fw == 0, OAR_INTERNAL, "binary_upsampled: bad shape");
    ProfScope ps(s, "binary", 4.0 * (8.0 * (double)total4 + 4.0 * (double)total4 / (fh * fw)), 4.0 * (double)total4);
    hipLaunchKernelGGL(binary_upsampled_kernel, dim3(grid_for(total4)), dim3(256), 0, s, a, b, y, total4, C / 4, Ho, Wo, fh, fw, op);
}

__global__ __launch_bounds__(256) void copy2d_kernel(const float* x, float* y, long rows, int c, int x_ld, int y_ld, int vec) {
    if (vec) {
        int c4 = c >> 2; long total = rows * c4;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
            long r = i / c4; int q = (int)(i - r * c4);
            *reinterpret_cast<float4*>(y + r * y_ld + q * 4) = *reinterpret_cast<const float4*>(x + r * x_ld + q * 4);
        }
    } else {
        long total = rows * c;
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
            long r = i / c; int q = (int)(i - r * c);
            y[r * y_ld + q] = x[r * x_ld + q];
        }
    }
}
void copy2d(hipStream_t s, const float* x, float* y, int64_t rows, int c, int x_ld, int y_ld) {
    if (rows == 0 || c == 0) return;
    ProfScope ps(s, "copy2d", 8.0 * (double)rows * c, 0.0);
    int vec = ((c & 3) == 0 && (x_ld & 3) == 0 && (y_ld & 3) == 0 && (((uintptr_t)x | (uintptr_t)y) & 15) == 0) ? 1 : 0;
    hipLaunchKernelGGL(copy2d_kernel, dim3(grid_for(vec ? rows * c / 4 : rows * c)), dim3(256), 0, s, x, y, (long)rows, c, x_ld, y_ld, vec);
}

struct PermP {
    int rank;
    long dims[6], st[6];
};
__global__ __launch_bounds__(256) void permute_kernel(const float* x, float* y, long n, PermP p) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        long r = i, o = 0;
#pragma unroll
        for (int d = 5; d >= 0; --d) {
            if (d < p.rank) { long q = r / p.dims[d]; o += (r - q * p.dims[d]) * p.st[d]; r = q; }
        }
        y[i] = x[o];
    }
}
void permute(hipStream_t s, const float* x, float* y, int rank, const int64_t* out_dims, const int64_t* in_strides) {
    OAR_CHECK(rank <= 6, OAR_UNSUPPORTED_OP, "permute: rank > 6");
    PermP p; p.rank = rank; long n = 1;
    for (int i = 0; i < 6; ++i) { p.dims[i] = i < rank ? out_dims[i] : 1; p.st[i] = i < rank ? in_strides[i] : 0; if (i < rank) n *= out_dims[i]; }
    if (n == 0) return;
    ProfScope ps(s, "permute", 8.0 * (double)n, 0.0);
    hipLaunchKernelGGL(permute_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, y, n, p);
}

// ------------------------------------------------------------------------------------------ Pad (constant / reflect / edge)
struct PadP { int rank; long in_dims[6], out_dims[6], before[6], in_st[6]; int mode; float value; };
__global__ __launch_bounds__(256) void pad_kernel(const float* x, float* y, long n, PadP p) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        long r = i, o = 0;
        bool inside = true;
#pragma unroll
        for (int d = 5; d >= 0; --d) {
            if (d < p.rank) {
                const long q = r / p.out_dims[d];
                long c = (r - q * p.out_dims[d]) - p.before[d];
                r = q;
                const long D = p.in_dims[d];
                if (c < 0 || c >= D) {
                    if (p.mode == 0) inside = false;
                    else if (p.mode == 2) c = c < 0 ? 0 : D - 1;
                    else {   // reflect (no edge repeat); D == 1 degenerates to the only element
                        if (D == 1) c = 0;
                        else {
                            const long period = 2 * (D - 1);
                            c %= period; if (c < 0) c += period;
                            if (c >= D) c = period - c;
                        }
                    }
                }
                o += c * p.in_st[d];
            }
        }
        y[i] = inside ? x[o] : p.value;
    }
}
void pad_nd(hipStream_t s, const float* x, float* y, int rank, const int64_t* in_dims, const int64_t* out_dims, const int64_t* before, int mode, float value) {
    OAR_CHECK(rank >= 1 && rank <= 6, OAR_UNSUPPORTED_OP, "Pad: rank must be 1..6");
    PadP p; p.rank = rank; p.mode = mode; p.value = value;
    long n = 1, st = 1;
    for (int d = 5; d >= 0; --d) {
        p.in_dims[d] = d < rank ? in_dims[d] : 1; p.out_dims[d] = d < rank ? out_dims[d] : 1; p.before[d] = d < rank ? before[d] : 0;
    }
    for (int d = rank - 1; d >= 0; --d) { p.in_st[d] = st; st *= in_dims[d]; n *= out_dims[d]; }
    for (int d = rank; d < 6; ++d) p.in_st[d] = 0;
    if (n == 0) return;
    ProfScope ps(s, "pad", 8.0 * (double)n, 0.0);
    hipLaunchKernelGGL(pad_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, y, n, p);
}

// ------------------------------------------------------------------------------------------ batched GEMM (small, VALU)
// 64x64 C tile per workgroup, 16x16 threads, 4x4 micro-tile, K step 16 through LDS.
__global__ __launch_bounds__(256) void gemm_batched_kernel(GemmP p) {
    __shared__ float As[16][65];
    __shared__ float Bs[16][65];
    const int b = blockIdx.z;
    const float* A = p.A + (long)b * p.sA;
    const float* B = p.B + (long)b * p.sB;
    float* C = p.C + (long)b * p.sC;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < p.K; k0 += 16) {
        for (int i = threadIdx.x; i < 64 * 16; i += 256) {
            int r = i >> 4, kk = i & 15;  // A: rows along M, k fastest (coalesced along K)
            int m = m0 + r, k = k0 + kk;
            As[kk][r] = (m < p.M && k < p.K) ? A[(long)m * p.K + k] : 0.f;
        }
        if (p.transB) {
            for (int i = threadIdx.x; i < 64 * 16; i += 256) {
                int r = i >> 4, kk = i & 15;
                int n = n0 + r, k = k0 + kk;
                Bs[kk][r] = (n < p.N && k < p.K) ? B[(long)n * p.K + k] : 0.f;
            }
        } else {
            for (int i = threadIdx.x; i < 64 * 16; i += 256) {
                int kk = i >> 6, r = i & 63;
                int n = n0 + r, k = k0 + kk;
                Bs[kk][r] = (n < p.N && k < p.K) ? B[(long)k * p.N + n] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], bb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = As[kk][ty * 4 + i]; bb[i] = Bs[kk][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], bb[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int m = m0 + ty * 4 + i;
        if (m >= p.M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int n = n0 + tx * 4 + j;
            if (n >= p.N) continue;
            float v = acc[i][j] * p.alpha;
            if (p.bias) v += p.bias[n];
            if (p.residual) v += p.residual[(long)b * p.sC + (long)m * p.N + n];
            C[(long)m * p.N + n] = apply_act(v, p.act.kind, p.act.alpha, p.act.beta);
        }
    }
}
void gemm_batched(hipStream_t s, const GemmP& p) {
    if (p.batch == 0 || p.M == 0 || p.N == 0) return;
    ProfScope ps(s, "gemm_batched", 4.0 * (double)p.batch * ((double)p.M * p.K + (double)p.K * p.N + (double)p.M * p.N), 2.0 * (double)p.batch * p.M * p.N * p.K);
    dim3 grid((p.N + 63) / 64, (p.M + 63) / 64, p.batch);
    hipLaunchKernelGGL(gemm_batched_kernel, grid, dim3(256), 0, s, p);
}

// ------------------------------------------------------------------------------------------ layernorm / softmax
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// one wave per row
__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, const float* g, const float* b, float* y, long rows, int C, float eps) {
    long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + row * C;
    float s = 0.f;
    for (int i = lane; i < C; i += 64) s += xr[i];
    float mean = wave_sum(s) / (float)C;
    float v = 0.f;
    for (int i = lane; i < C; i += 64) { float d = xr[i] - mean; v += d * d; }
    float var = wave_sum(v) / (float)C;
    float inv = 1.0f / sqrtf(var + eps);
    for (int i = lane; i < C; i += 64) {
        float t = (xr[i] - mean) * inv;
        if (g) t *= g[i];
        if (b) t += b[i];
        y[row * C + i] = t;
    }
}
// Rows of 4 | C <= 1024 floats (SVTRv2: 128 / 256 / 384; the SVTR neck: 64 ... 192): half a wave per row, the row held in registers as float4s --
// one read and one write of HBM per element, 16-byte accesses, two rows per wave in flight.  The one-wave-per-row kernel above re-read the row
// three times through the cache with 4-byte loads and ran at 1.9 TB/s (76 launches, 7.4 % of a BASELINE C3 step).  Same arithmetic per element
// (mean, then centred squares, then (x - mean) * inv * g + b); the sums are taken in a different order (float4 lanes, then a 32-lane butterfly).
template <int NV>
__global__ __launch_bounds__(256) void layernorm_v4_kernel(const float4* __restrict__ x, const float4* __restrict__ g, const float4* __restrict__ b, float4* __restrict__ y, long rows, int C4, float invC, float eps) {
    const long row = (long)blockIdx.x * 8 + (threadIdx.x >> 5);
    if (row >= rows) return;
    const int l = threadIdx.x & 31;
    const float4* xr = x + row * C4;
    float4 v[NV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int j = l + 32 * i;
        v[i] = j < C4 ? xr[j] : make_float4(0.f, 0.f, 0.f, 0.f);
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) s += __shfl_xor(s, o, 32);
    const float mean = s * invC;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if (l + 32 * i < C4) {
            const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
            q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    }
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) q += __shfl_xor(q, o, 32);
    const float inv = 1.0f / sqrtf(q * invC + eps);
    float4* yr = y + row * C4;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int j = l + 32 * i;
        if (j >= C4) continue;
        float4 t = make_float4((v[i].x - mean) * inv, (v[i].y - mean) * inv, (v[i].z - mean) * inv, (v[i].w - mean) * inv);
        if (g) { const float4 gg = g[j]; t.x *= gg.x; t.y *= gg.y; t.z *= gg.z; t.w *= gg.w; }
        if (b) { const float4 bb = b[j]; t.x += bb.x; t.y += bb.y; t.z += bb.z; t.w += bb.w; }
        yr[j] = t;
    }
}
void layernorm(hipStream_t s, const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int C, float eps) {
    if (rows == 0) return;
    ProfScope ps(s, "layernorm", 8.0 * (double)rows * C, 8.0 * (double)rows * C);
    static const bool v4 = [] { const char* e = getenv("OAR_LN_V4"); return !e || atoi(e) != 0; }();
    const bool aligned = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(gamma) | reinterpret_cast<uintptr_t>(beta)) & 15) == 0;
    if (v4 && (C & 3) == 0 && C <= 1024 && aligned) {
        const int C4 = C / 4, nv = (C4 + 31) / 32;
        const dim3 grid((unsigned)((rows + 7) / 8));
        const float invC = 1.0f / (float)C;
        auto X = reinterpret_cast<const float4*>(x); auto G = reinterpret_cast<const float4*>(gamma); auto B = reinterpret_cast<const float4*>(beta); auto Y = reinterpret_cast<float4*>(y);
        if (nv <= 1) hipLaunchKernelGGL(layernorm_v4_kernel<1>, grid, dim3(256), 0, s, X, G, B, Y, (long)rows, C4, invC, eps);
        else if (nv <= 2) hipLaunchKernelGGL(layernorm_v4_kernel<2>, grid, dim3(256), 0, s, X, G, B, Y, (long)rows, C4, invC, eps);
        else if (nv <= 3) hipLaunchKernelGGL(layernorm_v4_kernel<3>, grid, dim3(256), 0, s, X, G, B, Y, (long)rows, C4, invC, eps);
        else if (nv <= 4) hipLaunchKernelGGL(layernorm_v4_kernel<4>, grid, dim3(256), 0, s, X, G, B, Y, (long)rows, C4, invC, eps);
        else hipLaunchKernelGGL(layernorm_v4_kernel<8>, grid, dim3(256), 0, s, X, G, B, Y, (long)rows, C4, invC, eps);
        return;
    }
    hipLaunchKernelGGL(layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, gamma, beta, y, (long)rows, C, eps);
}

// Softmax over the last dim. Small C: one wave per row; large C (CTC vocab): one workgroup per row.
__global__ __launch_bounds__(256) void softmax_wave_kernel(const float* x, float* y, long rows, int C) {
    long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    const float* xr = x + row * C;
    float m = -3.402823466e38f;
    for (int i = lane; i < C; i += 64) m = fmaxf(m, xr[i]);
    m = wave_max(m);
    float s = 0.f;
    for (int i = lane; i < C; i += 64) s += expf(xr[i] - m);
    s = wave_sum(s);
    for (int i = lane; i < C; i += 64) y[row * C + i] = expf(xr[i] - m) / s;
}
__global__ __launch_bounds__(256) void softmax_block_kernel(const float* x, float* y, int C) {
    // one workgroup per row; the row is staged in LDS (C * 4 bytes) so HBM is read exactly once
    extern __shared__ float rowbuf[];
    __shared__ float red[4];
    __shared__ float bcast;
    const long row = blockIdx.x;
    const float* xr = x + row * C;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float m = -3.402823466e38f;
    for (int i = tid; i < C; i += 256) { float v = xr[i]; rowbuf[i] = v; m = fmaxf(m, v); }
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    if (tid == 0) bcast = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    m = bcast;
    float s = 0.f;
    for (int i = tid; i < C; i += 256) { float e = expf(rowbuf[i] - m); rowbuf[i] = e; s += e; }
    s = wave_sum(s);
    __syncthreads();
    if (lane == 0) red[wave] = s;
    __syncthreads();
    if (tid == 0) bcast = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    s = bcast;
    for (int i = tid; i < C; i += 256) y[row * C + i] = rowbuf[i] / s;
}
// Fused CTC tail for the recognizer seam: softmax over the vocabulary + "last index of the row maximum" + its
// probability, without writing the [rows, C] probability tensor.  Bit-identical to softmax_block_kernel followed by
// pp::ctc_argmax_kernel: same expf, same partial-sum tree; p_max = expf(0)/s = 1/s and the tie set
// {i : p_i == p_max} is exactly {i : expf(x_i - m) == 1.0f} (division by the same s is monotone).
__global__ __launch_bounds__(256) void softmax_argmax_kernel(const float* x, int C, int ld, int64_t* idx, float* prob) {
    extern __shared__ float rowbuf[];
    __shared__ float red[4];
    __shared__ int redi[4];
    __shared__ float bcast;
    const long row = blockIdx.x;
    const float* xr = x + row * ld;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float m = -3.402823466e38f;
    for (int i = tid; i < C; i += 256) { float v = xr[i]; rowbuf[i] = v; m = fmaxf(m, v); }
    m = wave_max(m);
    if (lane == 0) red[wave] = m;
    __syncthreads();
    if (tid == 0) bcast = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    m = bcast;
    float s = 0.f;
    int last = -1;
    for (int i = tid; i < C; i += 256) { float e = expf(rowbuf[i] - m); s += e; if (e == 1.0f) last = i; }
    s = wave_sum(s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) last = max(last, __shfl_xor(last, o, 64));
    __syncthreads();
    if (lane == 0) { red[wave] = s; redi[wave] = last; }
    __syncthreads();
    if (tid == 0) {
        float tot = (red[0] + red[1]) + (red[2] + red[3]);
        int li = max(max(redi[0], redi[1]), max(redi[2], redi[3]));
        idx[row] = li < 0 ? 0 : li;
        prob[row] = 1.0f / tot;
    }
}
void softmax_argmax(hipStream_t s, const float* logits, int64_t rows, int C, int ld, int64_t* idx, float* prob) {
    if (rows == 0 || C == 0) return;
    OAR_CHECK((size_t)C * 4 <= 150 * 1024, OAR_UNSUPPORTED_OP, "softmax_argmax: row longer than the LDS staging buffer");
    ProfScope ps(s, "softmax_argmax", 4.0 * (double)rows * C, 4.0 * (double)rows * C);
    hipLaunchKernelGGL(softmax_argmax_kernel, dim3((unsigned)rows), dim3(256), (size_t)C * sizeof(float), s, logits, C, ld, idx, prob);
}

// merges the per-tile softmax partials of the CTC heads' epilogues: 16 lanes per row (round 5: one thread per row left 40 workgroups walking 54 tiles
// each, 21 us per batch; lane l takes tiles l, l + 16, ..., then a fixed xor tree: the result does not depend on the launch geometry)
__global__ __launch_bounds__(256) void ctc_combine_kernel(const float4* part, long rows, int tiles, int64_t* idx, float* prob) {
    const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const int l = threadIdx.x & 15;
    const bool live = row < rows;
    const float4* pr = part + (live ? row : 0) * tiles;
    float M = -3.402823466e38f;
    if (live) for (int t = l; t < tiles; t += 16) M = fmaxf(M, pr[t].x);
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) M = fmaxf(M, __shfl_xor(M, o, 16));
    float S = 0.f;
    int last = -1;
    if (live) for (int t = l; t < tiles; t += 16) {
        const float4 v = pr[t];
        const float e = expf(v.x - M);
        S += v.y * e;
        if (e == 1.0f && __float_as_int(v.z) >= 0) last = max(last, __float_as_int(v.z));   // columns ascend with the tile index: the largest hit is the last
    }
#pragma unroll
    for (int o = 8; o >= 1; o >>= 1) { S += __shfl_xor(S, o, 16); last = max(last, __shfl_xor(last, o, 16)); }
    if (live && l == 0) { idx[row] = last < 0 ? 0 : last; prob[row] = 1.0f / S; }
}
void ctc_combine(hipStream_t s, const float* part, int64_t rows, int tiles, int64_t* idx, float* prob) {
    if (rows == 0) return;
    ProfScope ps(s, "ctc_combine", 16.0 * (double)rows * tiles, 0.0);
    hipLaunchKernelGGL(ctc_combine_kernel, dim3((unsigned)((rows * 16 + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const float4*>(part), (long)rows, tiles, idx, prob);
}

void softmax_lastdim(hipStream_t s, const float* x, float* y, int64_t rows, int C) {
    if (rows == 0 || C == 0) return;
    ProfScope ps(s, "softmax", 8.0 * (double)rows * C, 4.0 * (double)rows * C);
    if (C <= 1024) hipLaunchKernelGGL(softmax_wave_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, y, (long)rows, C);
    else {
        OAR_CHECK((size_t)C * 4 <= 150 * 1024, OAR_UNSUPPORTED_OP, "softmax: row longer than the LDS staging buffer");
        hipLaunchKernelGGL(softmax_block_kernel, dim3((unsigned)rows), dim3(256), (size_t)C * sizeof(float), s, x, y, C);
    }
}

// ------------------------------------------------------------------------------------------ squeeze-excite gate
// GlobalAveragePool -> 1x1 conv -> act -> 1x1 conv -> act on [n, C, 1, 1] is two GEMVs per image: as implicit-GEMM launches
// they are two ~10 us kernels of a few workgroups each; here ONE launch does both, the pooled vector and the hidden vector
// staying in LDS.  The work is a few hundred kFLOP per image: what it costs is load round trips, so the kernel is shaped to make
// them few -- 16 waves per workgroup; FC1: every wave owns 4 hidden units per pass and keeps 16 weight loads in flight;
// FC2: the hidden axis is cut into P parts summed through LDS in a fixed order; with few images (a detector sub-batch) G
// workgroups share an image, each recomputing the cheap hidden vector and producing its slice of the outputs.
__global__ __launch_bounds__(1024) void se_fc_kernel(const float* __restrict__ x, const float* __restrict__ w1, const float* __restrict__ b1, Act a1,
                                                     const float* __restrict__ w2, const float* __restrict__ b2, Act a2, float* __restrict__ y, int C, int Cmid, int Cout,
                                                     int slice, int parts, int tiles, float hw) {
    extern __shared__ float se_lds[];   // pooled [C] | hidden [Cmid] | partial [parts][slice] (first: the tile sums' partial reduction [gparts][C])
    float* pooled = se_lds;
    float* hidden = se_lds + C;
    float* partial = hidden + Cmid;
    const long n = blockIdx.x;
    const int c_lo = blockIdx.y * slice, c_hi = min(c_lo + slice, Cout);
    if (tiles > 0) {
        // x = [N][tiles][C] channel sums of the producing depthwise conv's tiles: the reduction global_avgpool_kernel would do over them as a
        // launch of its own (HW = tiles, one split), term for term -- gparts groups stride over the tiles four at a time as (a + b) + (c + d),
        // the groups are added in order, the total is divided by the pixel count
        const int C4 = C >> 2, gparts = 256 / C4 > 0 ? 256 / C4 : 1;
        float* red = partial;   // [gparts][C]: gparts * C <= 1024 floats (the host sizes the LDS for it)
        const float* xb = x + n * (long)tiles * C;
        for (int t = threadIdx.x; t < gparts * C; t += blockDim.x) {
            const int c = t % C, part = t / C;
            float acc = 0.f;
            int i = part;
            for (; i + 3 * gparts < tiles; i += 4 * gparts)
                acc += (xb[(long)i * C + c] + xb[(long)(i + gparts) * C + c]) + (xb[(long)(i + 2 * gparts) * C + c] + xb[(long)(i + 3 * gparts) * C + c]);
            for (; i < tiles; i += gparts) acc += xb[(long)i * C + c];
            red[part * C + c] = acc;
        }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            float t = red[c];
            for (int q = 1; q < gparts; ++q) t += red[q * C + c];
            pooled[c] = t / hw;
        }
    } else {
        for (int c = threadIdx.x; c < C; c += blockDim.x) pooled[c] = x[n * C + c];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    for (int jb = wave * 4; jb < Cmid; jb += nwaves * 4) {   // four hidden units per wave per pass: their weight rows load together
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const float* r0 = w1 + (long)min(jb + 0, Cmid - 1) * C;
        const float* r1 = w1 + (long)min(jb + 1, Cmid - 1) * C;
        const float* r2 = w1 + (long)min(jb + 2, Cmid - 1) * C;
        const float* r3 = w1 + (long)min(jb + 3, Cmid - 1) * C;
        int c = lane;
        for (; c + 192 < C; c += 256) {   // 16 independent loads per round trip
            float w[16];
#pragma unroll
            for (int q = 0; q < 4; ++q) { w[q * 4 + 0] = r0[c + q * 64]; w[q * 4 + 1] = r1[c + q * 64]; w[q * 4 + 2] = r2[c + q * 64]; w[q * 4 + 3] = r3[c + q * 64]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float pv = pooled[c + q * 64];
                acc[0] = fmaf(w[q * 4 + 0], pv, acc[0]); acc[1] = fmaf(w[q * 4 + 1], pv, acc[1]);
                acc[2] = fmaf(w[q * 4 + 2], pv, acc[2]); acc[3] = fmaf(w[q * 4 + 3], pv, acc[3]);
            }
        }
        for (; c < C; c += 64) {
            const float pv = pooled[c];
            acc[0] = fmaf(r0[c], pv, acc[0]); acc[1] = fmaf(r1[c], pv, acc[1]); acc[2] = fmaf(r2[c], pv, acc[2]); acc[3] = fmaf(r3[c], pv, acc[3]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float v = wave_sum(acc[u]);
            if (lane == 0 && jb + u < Cmid) hidden[jb + u] = apply_act(v + (b1 ? b1[jb + u] : 0.f), a1.kind, a1.alpha, a1.beta);
        }
    }
    __syncthreads();
    // FC2: thread (part, k) sums hidden units [j0, j1) of output c_lo + k (W2 transposed: coalesced over the output channel)
    const int k = threadIdx.x % slice, part = threadIdx.x / slice;
    if (part < parts && c_lo + k < c_hi) {
        const int per = (Cmid + parts - 1) / parts, j0 = part * per, j1 = min(j0 + per, Cmid);
        const float* col = w2 + c_lo + k;
        float acc = 0.f;
        int j = j0;
        for (; j + 8 <= j1; j += 8) {   // eight independent loads in flight (a rolled loop pays one L2 round trip per term)
            float w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = col[(long)(j + u) * Cout];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = fmaf(w[u], hidden[j + u], acc);
        }
        for (; j < j1; ++j) acc = fmaf(col[(long)j * Cout], hidden[j], acc);
        partial[part * slice + k] = acc;
    }
    __syncthreads();
    if (threadIdx.x < slice && c_lo + (int)threadIdx.x < c_hi) {
        const int c = c_lo + threadIdx.x;
        float acc = b2 ? b2[c] : 0.f;
        for (int q = 0; q < parts; ++q) acc += partial[q * slice + threadIdx.x];
        y[n * Cout + c] = apply_act(acc, a2.kind, a2.alpha, a2.beta);
    }
}
void se_fc(hipStream_t s, const float* x, const float* w1, const float* b1, Act act1, const float* w2, const float* b2, Act act2, float* y, int N, int C,
           int Cmid, int Cout, int tiles, int hw) {
    if (N == 0) return;
    OAR_CHECK(tiles == 0 || ((C & 3) == 0 && C <= 1024 && hw > 0), OAR_INTERNAL, "se_fc: tile sums need C % 4 == 0, C <= 1024");
    // G workgroups per image when the images alone leave most of the chip idle; a slice is a multiple of 64 outputs, <= 1024
    int G = 1;
    if (N < 128) G = std::max(1, std::min(std::min(8, 256 / N), (Cout + 63) / 64));
    int slice = ((Cout + G - 1) / G + 63) / 64 * 64;
    while (slice > 1024) { ++G; slice = ((Cout + G - 1) / G + 63) / 64 * 64; }
    G = (Cout + slice - 1) / slice;
    const int parts = std::max(1, std::min(std::min(8, 1024 / slice), (Cmid + 7) / 8));
    const int gparts = tiles > 0 ? std::max(1, 256 / (C >> 2)) : 0;
    const size_t lds = (size_t)(C + Cmid + std::max(parts * slice, gparts * C)) * sizeof(float);
    OAR_CHECK(lds <= 64 * 1024, OAR_UNSUPPORTED_OP, "se_fc: vectors exceed LDS");
    ProfScope ps(s, "se_fc", 4.0 * ((double)N * ((double)C * std::max(tiles, 1) + Cout) + (double)Cmid * (C + Cout)), 2.0 * N * (double)Cmid * (C + Cout));
    hipLaunchKernelGGL(se_fc_kernel, dim3((unsigned)N, (unsigned)G), dim3(1024), lds, s, x, w1, b1, act1, w2, b2, act2, y, C, Cmid, Cout, slice, parts, tiles, (float)hw);
}

// ------------------------------------------------------------------------------------------ ReduceMean (last axis)
__device__ __forceinline__ float wave_reduce_mode(float v, int mode) {
    if (mode <= 1) return wave_sum(v);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_xor(v, off, 64);
        v = mode == 2 ? fmaxf(v, o) : mode == 3 ? fminf(v, o) : v * o;
    }
    return v;
}
__global__ __launch_bounds__(256) void reduce_lastdim_kernel(const float* __restrict__ x, float* __restrict__ y, long rows, int C, int mode) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float s = mode <= 1 ? 0.f : mode == 2 ? -INFINITY : mode == 3 ? INFINITY : 1.f;
    for (int i = lane; i < C; i += 64) {
        const float v = x[row * C + i];
        s = mode <= 1 ? s + v : mode == 2 ? fmaxf(s, v) : mode == 3 ? fminf(s, v) : s * v;
    }
    s = wave_reduce_mode(s, mode);
    if (lane == 0) y[row] = mode == 0 ? s / (float)C : s;
}
void reduce_lastdim(hipStream_t s, const float* x, float* y, int64_t rows, int C, int mode) {
    if (rows == 0 || C == 0) return;
    ProfScope ps(s, "reduce_mean", 4.0 * (double)rows * (C + 1), (double)rows * C);
    hipLaunchKernelGGL(reduce_lastdim_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, y, (long)rows, C, mode);
}

// ------------------------------------------------------------------------------------------ ArgMax / ArgMin (last axis)
// One wave per row; (value, index) pairs, ties to the smaller index (select_last_index = 0) or the larger one (= 1).  The
// index is written as f32 (exact below 2^24; the boundary converts to i64).
__global__ __launch_bounds__(256) void argreduce_lastdim_kernel(const float* __restrict__ x, float* __restrict__ y, long rows, int C, int is_min, int last) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float best = 0.f;
    int bi = -1;
    for (int i = lane; i < C; i += 64) {
        const float v = x[row * C + i];
        const bool better = bi < 0 || (is_min ? v < best : v > best) || (last && v == best);
        if (better) { best = v; bi = i; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off, 64);
        const int oi = __shfl_xor(bi, off, 64);
        const bool take = oi >= 0 && (bi < 0 || (is_min ? ov < best : ov > best) || (ov == best && (last ? oi > bi : oi < bi)));
        if (take) { best = ov; bi = oi; }
    }
    if (lane == 0) y[row] = (float)bi;
}
void argreduce_lastdim(hipStream_t s, const float* x, float* y, int64_t rows, int C, bool is_min, bool select_last) {
    if (rows == 0 || C == 0) return;
    ProfScope ps(s, "reduce_mean", 4.0 * (double)rows * (C + 1), (double)rows * C);
    hipLaunchKernelGGL(argreduce_lastdim_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, y, (long)rows, C, is_min ? 1 : 0, select_last ? 1 : 0);
}

// ------------------------------------------------------------------------------------------ Where
struct WhereP { int rank; long dims[6], sc[6], sa[6], sb[6]; };
__global__ __launch_bounds__(256) void where_kernel(const float* c, const float* a, const float* b, float* y, long n, WhereP p) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        long r = i, oc = 0, oa = 0, ob = 0;
#pragma unroll
        for (int d = 5; d >= 0; --d) {
            if (d < p.rank) {
                long q = r / p.dims[d]; long idx = r - q * p.dims[d]; r = q;
                oc += idx * p.sc[d]; oa += idx * p.sa[d]; ob += idx * p.sb[d];
            }
        }
        y[i] = c[oc] != 0.0f ? a[oa] : b[ob];
    }
}
void where(hipStream_t s, const float* cond, const float* a, const float* b, float* y, int rank, const int64_t* dims, const int64_t* sc, const int64_t* sa, const int64_t* sb) {
    OAR_CHECK(rank <= 6, OAR_UNSUPPORTED_OP, "Where: rank > 6");
    long n = 1;
    for (int i = 0; i < rank; ++i) n *= dims[i];
    if (n == 0) return;
    ProfScope ps(s, "where", 16.0 * (double)n, 0.0);
    WhereP p;
    p.rank = rank;
    for (int i = 0; i < 6; ++i) { p.dims[i] = i < rank ? dims[i] : 1; p.sc[i] = i < rank ? sc[i] : 0; p.sa[i] = i < rank ? sa[i] : 0; p.sb[i] = i < rank ? sb[i] : 0; }
    hipLaunchKernelGGL(where_kernel, dim3(grid_for(n)), dim3(256), 0, s, cond, a, b, y, n, p);
}

// ------------------------------------------------------------------------------------------ GridSample (UVDoc's un-warp)
// One thread per output pixel; channels are innermost (channels-last), so the four taps are contiguous C-float reads.
__device__ __forceinline__ float gs_reflect(float v, float lo, float hi) {   // reflect v into [lo, hi] (ONNX gs_reflect)
    const float range = hi - lo;
    if (range <= 0.f) return lo;
    if (v < lo) {
        const float dv = lo - v;
        const int n = (int)(dv / range);
        const float r = dv - n * range;
        return (n & 1) ? hi - r : lo + r;
    }
    if (v > hi) {
        const float dv = v - hi;
        const int n = (int)(dv / range);
        const float r = dv - n * range;
        return (n & 1) ? lo + r : hi - r;
    }
    return v;
}
__global__ __launch_bounds__(256) void grid_sample_kernel(const float* __restrict__ x, const float* __restrict__ grid, float* __restrict__ y, long total, int H, int W,
                                                          int C, int Ho, int Wo, int mode, int padding, int align) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long n = i / ((long)Ho * Wo);
        const float gx = grid[i * 2], gy = grid[i * 2 + 1];
        // [-1, 1] -> pixel coordinates
        float fx = align ? (gx + 1.f) * 0.5f * (float)(W - 1) : ((gx + 1.f) * (float)W - 1.f) * 0.5f;
        float fy = align ? (gy + 1.f) * 0.5f * (float)(H - 1) : ((gy + 1.f) * (float)H - 1.f) * 0.5f;
        if (padding == 1) { fx = fminf(fmaxf(fx, 0.f), (float)(W - 1)); fy = fminf(fmaxf(fy, 0.f), (float)(H - 1)); }
        else if (padding == 2) {
            if (align) { fx = gs_reflect(fx, 0.f, (float)(W - 1)); fy = gs_reflect(fy, 0.f, (float)(H - 1)); }
            else { fx = gs_reflect(fx, -0.5f, (float)W - 0.5f); fy = gs_reflect(fy, -0.5f, (float)H - 0.5f); }
            fx = fminf(fmaxf(fx, 0.f), (float)(W - 1)); fy = fminf(fmaxf(fy, 0.f), (float)(H - 1));
        }
        const float* xb = x + n * (long)H * W * C;
        float* o = y + i * C;
        auto tap = [&](int yy, int xx, int c) { return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? xb[((long)yy * W + xx) * C + c] : 0.f; };
        if (mode == 1) {
            const int xi = (int)nearbyintf(fx), yi = (int)nearbyintf(fy);
            for (int c = 0; c < C; ++c) o[c] = tap(yi, xi, c);
        } else {
            const float x0f = floorf(fx), y0f = floorf(fy);
            const int x0 = (int)x0f, y0 = (int)y0f;
            const float wx1 = fx - x0f, wy1 = fy - y0f, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
            for (int c = 0; c < C; ++c)
                o[c] = tap(y0, x0, c) * (wy0 * wx0) + tap(y0, x0 + 1, c) * (wy0 * wx1) + tap(y0 + 1, x0, c) * (wy1 * wx0) + tap(y0 + 1, x0 + 1, c) * (wy1 * wx1);
        }
    }
}
void grid_sample(hipStream_t s, const float* x, const float* grid, float* y, int N, int H, int W, int C, int Ho, int Wo, int mode, int padding, int align_corners) {
    const long total = (long)N * Ho * Wo;
    if (total == 0 || C == 0) return;
    ProfScope ps(s, "grid_sample", 4.0 * (double)total * (2 + 5 * C), 8.0 * (double)total * C);
    hipLaunchKernelGGL(grid_sample_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, grid, y, total, H, W, C, Ho, Wo, mode, padding, align_corners);
}

// ------------------------------------------------------------------------------------------ SVTR attention
// softmax(scale * q k^T) v for one (image, head) per workgroup: K and V of the head sit in LDS (rows padded to HD with
// zeros), one thread per query row.  Two passes over the keys (row maximum, then exp / sum / weighted V), all in f32
// with explicit FMAs in ascending key / channel order -- the arithmetic of the Mul -> MatMul -> Softmax -> MatMul chain
// it replaces, without the [n, heads, T, T] score tensor or the q / k / v transposes ever reaching HBM.
template <int HD>
__global__ __launch_bounds__(256) void attention_kernel(const float* __restrict__ qkv, float* __restrict__ out, int T, int heads, int hd, float scale) {
    extern __shared__ float4 att_lds[];   // K [T][HD] | V [T][HD]
    constexpr int H4 = HD / 4;
    float4* Ks = att_lds;
    float4* Vs = att_lds + (long)T * H4;
    const int n = blockIdx.x / heads, h = blockIdx.x - n * heads, dim = heads * hd;
    const float* base = qkv + (long)n * T * 3 * dim + h * hd;
    float* Kf = reinterpret_cast<float*>(Ks);
    float* Vf = reinterpret_cast<float*>(Vs);
    for (int i = threadIdx.x; i < T * HD; i += blockDim.x) {
        const int t = i / HD, d = i - t * HD;
        float kv = 0.f, vv = 0.f;
        if (d < hd) { kv = base[(long)t * 3 * dim + dim + d]; vv = base[(long)t * 3 * dim + 2 * dim + d]; }
        Kf[i] = kv; Vf[i] = vv;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        float q[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) q[d] = d < hd ? base[(long)t * 3 * dim + d] * scale : 0.f;
        auto score = [&](int j) {
            float a = 0.f;
#pragma unroll
            for (int d4 = 0; d4 < H4; ++d4) {
                const float4 kk = Ks[j * H4 + d4];
                a = fmaf(q[4 * d4], kk.x, a); a = fmaf(q[4 * d4 + 1], kk.y, a); a = fmaf(q[4 * d4 + 2], kk.z, a); a = fmaf(q[4 * d4 + 3], kk.w, a);
            }
            return a;
        };
        float m = -3.402823466e38f;
#pragma unroll 4
        for (int j = 0; j < T; ++j) m = fmaxf(m, score(j));
        float l = 0.f;
        float o[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] = 0.f;
#pragma unroll 4
        for (int j = 0; j < T; ++j) {
            const float p = expf(score(j) - m);
            l += p;
#pragma unroll
            for (int d4 = 0; d4 < H4; ++d4) {
                const float4 vv = Vs[j * H4 + d4];
                o[4 * d4] = fmaf(p, vv.x, o[4 * d4]); o[4 * d4 + 1] = fmaf(p, vv.y, o[4 * d4 + 1]);
                o[4 * d4 + 2] = fmaf(p, vv.z, o[4 * d4 + 2]); o[4 * d4 + 3] = fmaf(p, vv.w, o[4 * d4 + 3]);
            }
        }
        float* y = out + ((long)n * T + t) * dim + h * hd;
#pragma unroll
        for (int d = 0; d < HD; ++d)
            if (d < hd) y[d] = o[d] / l;
    }
}

template <int HD>
static void launch_attention(hipStream_t s, const float* qkv, float* out, int n, int T, int heads, int hd, float scale) {
    OAR_MAX_LDS_ONCE(attention_kernel<HD>, 160 * 1024);
    const size_t lds = (size_t)2 * T * HD * sizeof(float);
    OAR_CHECK(lds <= 150 * 1024, OAR_UNSUPPORTED_OP, "attention: K and V of one head exceed the LDS staging buffer");
    const int threads = std::min(256, (T + 63) / 64 * 64);
    hipLaunchKernelGGL((attention_kernel<HD>), dim3((unsigned)(n * heads)), dim3(threads), lds, s, qkv, out, T, heads, hd, scale);
}

// K and V of a head do not fit LDS and the head dim is not the streaming bf16x6 kernel's (32): the same one-thread-per-query arithmetic over key blocks of
// kAttnBlock rows staged in LDS, with the running maximum / sum / output rescaled once per block (online soft-max: exp(m_old - m_new) is exact 1 when the
// maximum did not move).  A fallback for shapes no PP-OCR graph has -- any head dim <= 64 at any T runs instead of being refused (tools/op_fuzz.py found the gap).
constexpr int kAttnBlock = 128;
template <int HD>
__global__ __launch_bounds__(256) void attention_stream_kernel(const float* __restrict__ qkv, float* __restrict__ out, int T, int heads, int hd, float scale) {
    extern __shared__ float4 att_lds[];   // K [kAttnBlock][HD] | V [kAttnBlock][HD]
    constexpr int H4 = HD / 4;
    float4* Ks = att_lds;
    float4* Vs = att_lds + kAttnBlock * H4;
    float* Kf = reinterpret_cast<float*>(Ks);
    float* Vf = reinterpret_cast<float*>(Vs);
    const int n = blockIdx.x / heads, h = blockIdx.x - n * heads, dim = heads * hd;
    const float* base = qkv + (long)n * T * 3 * dim + h * hd;
    const int t = (int)blockIdx.y * 256 + (int)threadIdx.x;
    const bool valid = t < T;
    float q[HD], o[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) { q[d] = (valid && d < hd) ? base[(long)t * 3 * dim + d] * scale : 0.f; o[d] = 0.f; }
    float m = -3.402823466e38f, l = 0.f;
    for (int j0 = 0; j0 < T; j0 += kAttnBlock) {
        const int nj = min(kAttnBlock, T - j0);
        __syncthreads();   // (the previous block's reads are over)
        for (int i = threadIdx.x; i < nj * HD; i += blockDim.x) {
            const int r = i / HD, d = i - r * HD;
            float kv = 0.f, vv = 0.f;
            if (d < hd) { kv = base[(long)(j0 + r) * 3 * dim + dim + d]; vv = base[(long)(j0 + r) * 3 * dim + 2 * dim + d]; }
            Kf[i] = kv; Vf[i] = vv;
        }
        __syncthreads();
        auto score = [&](int j) {
            float a = 0.f;
#pragma unroll
            for (int d4 = 0; d4 < H4; ++d4) {
                const float4 kk = Ks[j * H4 + d4];
                a = fmaf(q[4 * d4], kk.x, a); a = fmaf(q[4 * d4 + 1], kk.y, a); a = fmaf(q[4 * d4 + 2], kk.z, a); a = fmaf(q[4 * d4 + 3], kk.w, a);
            }
            return a;
        };
        float mb = m;
#pragma unroll 4
        for (int j = 0; j < nj; ++j) mb = fmaxf(mb, score(j));
        const float corr = expf(m - mb);   // first block: exp(-huge) = 0 on l = 0, o = 0
        m = mb;
        l *= corr;
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] *= corr;
#pragma unroll 4
        for (int j = 0; j < nj; ++j) {
            const float p = expf(score(j) - m);
            l += p;
#pragma unroll
            for (int d4 = 0; d4 < H4; ++d4) {
                const float4 vv = Vs[j * H4 + d4];
                o[4 * d4] = fmaf(p, vv.x, o[4 * d4]); o[4 * d4 + 1] = fmaf(p, vv.y, o[4 * d4 + 1]);
                o[4 * d4 + 2] = fmaf(p, vv.z, o[4 * d4 + 2]); o[4 * d4 + 3] = fmaf(p, vv.w, o[4 * d4 + 3]);
            }
        }
    }
    if (valid) {
        float* y = out + ((long)n * T + t) * dim + h * hd;
#pragma unroll
        for (int d = 0; d < HD; ++d)
            if (d < hd) y[d] = o[d] / l;
    }
}
template <int HD>
static void launch_attention_stream(hipStream_t s, const float* qkv, float* out, int n, int T, int heads, int hd, float scale) {
    OAR_MAX_LDS_ONCE(attention_stream_kernel<HD>, 160 * 1024);
    const size_t lds = (size_t)2 * kAttnBlock * HD * sizeof(float);
    hipLaunchKernelGGL((attention_stream_kernel<HD>), dim3((unsigned)(n * heads), (unsigned)((T + 255) / 256)), dim3(256), lds, s, qkv, out, T, heads, hd, scale);
}

static bool attention_x6_on() { static const bool on = [] { const char* e = getenv("OAR_ATTN_X6"); return !e || atoi(e) != 0; }(); return on; }
bool attention_fits(int T, int heads, int hd) {
    if (hd < 1 || hd > 64) return false;
    if (attention_x6_on() && attention_x6_supported(T, heads, hd)) return true;
    (void)T; (void)heads;
    return true;   // K and V in LDS when they fit, key blocks streamed through it when they do not (attention_stream_kernel)
}
static bool attention_whole_head_fits(int T, int hd) {
    const int HD = hd <= 16 ? 16 : hd <= 32 ? 32 : 64;
    return (size_t)2 * T * HD * sizeof(float) <= 150 * 1024;
}
void attention(hipStream_t s, const float* qkv, float* out, int n, int T, int heads, int hd, float scale) {
    if (n == 0 || T == 0) return;
    OAR_CHECK(hd >= 1 && hd <= 64, OAR_UNSUPPORTED_OP, "attention: head_dim must be in 1..=64");
    if (attention_x6_on() && attention_x6_supported(T, heads, hd) && T > 32) return attention_x6(s, qkv, out, n, T, heads, hd, scale);
    const double nh = (double)n * heads;
    ProfScope ps(s, "attention", 4.0 * nh * T * 4.0 * hd, 4.0 * nh * T * T * hd);
    if (!attention_whole_head_fits(T, hd)) {
        if (hd <= 16) launch_attention_stream<16>(s, qkv, out, n, T, heads, hd, scale);
        else if (hd <= 32) launch_attention_stream<32>(s, qkv, out, n, T, heads, hd, scale);
        else launch_attention_stream<64>(s, qkv, out, n, T, heads, hd, scale);
        return;
    }
    if (hd <= 16) launch_attention<16>(s, qkv, out, n, T, heads, hd, scale);
    else if (hd <= 32) launch_attention<32>(s, qkv, out, n, T, heads, hd, scale);
    else launch_attention<64>(s, qkv, out, n, T, heads, hd, scale);
}

}  // namespace k
}  // namespace oar
