// kernels.hip -- hand-written gfx950 (CDNA4) kernels for the detector / recognizer networks.
//
// Data layout: feature maps are NHWC f32 in HBM (channel innermost => every conv reads/writes fully
// coalesced 16-byte vectors).  Dense convs / Linear layers run as implicit GEMM on the f32-input matrix
// cores (v_mfma_f32_16x16x4_f32: exact f32 FMA chain, bit-reproducible) with LDS-staged operand tiles;
// depthwise / pooling / resize / elementwise are bandwidth kernels with float4 accesses.
// Wave = 64 lanes everywhere.  Compiled with -ffp-contract=off; FMAs are written explicitly.
#include "kernels.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace oar {
namespace k {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------ activations
__device__ __forceinline__ float apply_act(float v, int kind, float alpha, float beta) {
    switch (kind) {
        case ACT_NONE: return v;
        case ACT_RELU: return v > 0.f ? v : 0.f;
        case ACT_HSWISH: {
            float t = fminf(fmaxf(v * (1.0f / 6.0f) + 0.5f, 0.f), 1.f);
            return v * t;
        }
        case ACT_HSIGMOID: return fminf(fmaxf(v * alpha + beta, 0.f), 1.f);
        case ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        case ACT_SWISH: return v * (1.0f / (1.0f + expf(-v)));
        case ACT_LEAKY: return v > 0.f ? v : v * alpha;
        case ACT_CLIP: return fminf(fmaxf(v, alpha), beta);
        case ACT_TANH: return tanhf(v);
        case ACT_GELU_ERF: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
        default: return v;
    }
}

// ------------------------------------------------------------------------------------------ implicit-GEMM conv
// GEMM view: D[cout][pixel] = sum_k W[cout][k] * X[pixel][k];  k = (kh, kw, ci), ci innermost (NHWC).
// MFMA 16x16x4 f32 operand map (cdna_hip_programming.md section 3): A[i = lane&15][k = lane>>4],
// B[k = lane>>4][j = lane&15], D[row = (lane>>4)*4 + r][col = lane&15].  A = weights (rows = cout),
// B = pixels (cols = pixel), so a lane ends with 4 CONSECUTIVE output channels of one pixel = one 16-byte
// NHWC store.
//
// No LDS and no barriers: within a 16-deep K chunk lane (p = lane&15, g = lane>>4) loads ONE float4
// X[pixel p][16*kc + 4g .. +3] straight from HBM (16 pixels x 64 contiguous bytes per wave instruction) and
// feeds component j to MFMA step j, i.e. step j contracts k = {j, 4+j, 8+j, 12+j}.  The weights are stored
// host-side in the matching fragment order Wf[cout/16][kc][lane][4] so the A operand is one coalesced float4
// per lane (1 KiB per wave instruction, identical for every wave => L1/L2 resident).  The K permutation is the
// same on both operands, so the contraction is exact.
// Workgroup = 4 independent waves; a wave owns PF*16 pixels x NT*16 couts (accumulators NT*PF*4 VGPRs).
struct IgemmP {
    const float* x;
    const float* w;      // fragment order, see above; KC = ceil(K/16) chunks, rows padded to 64 couts
    const float* bias;
    const float* res;
    float* y;
    long M;              // GEMM columns: N*Ho*Wo pixels (convT: input pixels)
    int K, KC;
    int gemm_cout;       // GEMM rows (Cout, or 4*Cout for convT 2x2)
    int Cout;            // channel count of y
    int H, W, Cin, Ho, Wo, kh, kw, sh, sw, pt, pl, dh, dw;
    int y_ld;
    int act, convt;
    int ny;              // number of cout tiles (for the XCD-aware tile order)
    unsigned cin_magic, kw_magic;   // floor(2^32 / d) + 1: q = umulhi(n, magic) == n / d for n < 2^16
    long mx_per_xcd;     // pixel tiles per XCD band
    float alpha, beta;
};

__device__ __forceinline__ void igemm_store(const IgemmP& p, f32x4 v, bool valid, long obase, int c, bool vec_ok) {
    if (!valid || c >= p.gemm_cout) return;
    float o[4] = {v[0], v[1], v[2], v[3]};
    if (vec_ok) {
        int co = c; long opix = obase;
        if (p.convt) { int ab = c / p.Cout; co = c - ab * p.Cout; opix = obase + (long)(ab >> 1) * (2L * p.Wo) + (ab & 1); }
        if (p.bias) { float4 bv = *reinterpret_cast<const float4*>(p.bias + co); o[0] += bv.x; o[1] += bv.y; o[2] += bv.z; o[3] += bv.w; }
        float* dst = p.y + opix * p.y_ld + co;
        if (p.res) { float4 rv = *reinterpret_cast<const float4*>(p.res + opix * p.y_ld + co); o[0] += rv.x; o[1] += rv.y; o[2] += rv.z; o[3] += rv.w; }
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = apply_act(o[r], p.act, p.alpha, p.beta);
        *reinterpret_cast<float4*>(dst) = make_float4(o[0], o[1], o[2], o[3]);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            int cc = c + r;
            if (cc >= p.gemm_cout) continue;
            int co = cc; long opix = obase;
            if (p.convt) { int ab = cc / p.Cout; co = cc - ab * p.Cout; opix = obase + (long)(ab >> 1) * (2L * p.Wo) + (ab & 1); }
            float t = o[r];
            if (p.bias) t += p.bias[co];
            if (p.res) t += p.res[opix * p.y_ld + co];
            p.y[opix * p.y_ld + co] = apply_act(t, p.act, p.alpha, p.beta);
        }
    }
}

template <int NT, int PF, bool IS1X1>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(IgemmP p) {  // 2 waves/SIMD => full 256-VGPR budget, no spills
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pl_ = lane & 15, g = lane >> 4;
    // XCD-aware tile order: workgroup b runs on XCD b % 8 (observed dispatch), each XCD has a private 4 MiB L2.
    // Every XCD gets ONE CONTIGUOUS band of pixel tiles (so the kh x kw tap re-reads of neighbouring rows hit the
    // same L2 instead of HBM: measured 10x over-fetch on the 3x3 convs with an interleaved order), and the ny
    // cout-tiles that re-read one pixel tile are adjacent in that XCD's dispatch order.
    const long b = blockIdx.x;
    const int xcd = (int)(b & 7);
    const long slot = b >> 3;
    const int ntile = (int)(slot % p.ny);
    const long mtile = (long)xcd * p.mx_per_xcd + slot / p.ny;
    if (slot / p.ny >= p.mx_per_xcd) return;
    const long m0 = (mtile * 4 + wave) * (PF * 16);
    const int nf0 = ntile * NT;
    if (m0 >= p.M) return;

    long pix_base[PF];
    int ih0[PF], iw0[PF];
#pragma unroll
    for (int pf = 0; pf < PF; ++pf) {
        // rows past M are clamped to the last pixel: they compute garbage that igemm_store never writes.  Keeping
        // every load unconditional matters: a load under a bounds-check branch makes the compiler's s_waitcnt
        // bookkeeping conservative (vmcnt(0) right after the prefetch is issued), which serialises the pipeline.
        long m = min(m0 + pf * 16 + pl_, p.M - 1);
        if (IS1X1) {
            pix_base[pf] = m * (long)p.Cin; ih0[pf] = 0; iw0[pf] = 0;
        } else {
            long hw = (long)p.Ho * p.Wo;
            long n = m / hw; long r = m - n * hw;
            int oh = (int)(r / p.Wo), ow = (int)(r - (long)oh * p.Wo);
            pix_base[pf] = n * (long)p.H * p.W * p.Cin;
            ih0[pf] = oh * p.sh - p.pt; iw0[pf] = ow * p.sw - p.pl;
        }
    }
    f32x4 acc[NT][PF];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < PF; ++b) acc[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const float4* wf = reinterpret_cast<const float4*>(p.w) + ((long)nf0 * p.KC) * 64 + lane;

    // k of this lane's quad in chunk kc, clamped: k >= K only happens in the zero-padded tail of the last chunk (the
    // matching W entries are 0), where the last valid quad is re-read instead of branching
    auto load_x1 = [&](int kc, int pf, float4& out, bool& ok_out) {
        const int k = min(kc * 16 + 4 * g, p.K - 4);
        float4 v;
        if (IS1X1) {
            v = *reinterpret_cast<const float4*>(p.x + pix_base[pf] + k);
            ok_out = true;
        } else {
            // k -> (tap_h, tap_w, ci) with multiply-high divisions (exact for k < 2^16, checked at launch): no
            // loop-carried state and no branches between the MFMAs
            const int tap = (int)__umulhi((unsigned)k, p.cin_magic), ci = k - tap * p.Cin;
            const int tap_h = (int)__umulhi((unsigned)tap, p.kw_magic), tap_w = tap - tap_h * p.kw;
            const int ih = ih0[pf] + tap_h * p.dh, iw = iw0[pf] + tap_w * p.dw;
            const bool ok = ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            const int ihc = min(max(ih, 0), p.H - 1), iwc = min(max(iw, 0), p.W - 1);
            v = *reinterpret_cast<const float4*>(p.x + pix_base[pf] + ((long)ihc * p.W + iwc) * p.Cin + ci);
            ok_out = ok;   // the zero-padding select is applied when the chunk is consumed (a select here would put
                           // the s_waitcnt for this load right behind its issue)
        }
        out = v;
    };
    auto load_w1 = [&](int kc, int nf, float4& out) { out = wf[((long)nf * p.KC + kc) * 64]; };
    auto comp = [](const float4& v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; };

    // One chunk = NT*PF*4 MFMAs, cout-fragment-major (consecutive MFMAs alternate between the PF accumulators of one
    // cout fragment, so a dependent MFMA issues two slots after its producer).  The PF+NT loads of the NEXT chunk are
    // spread evenly between them, in the order the next chunk consumes them (x0.., w0..): the memory pipe sees a
    // steady trickle instead of one burst per chunk that stalls every wave of the CU in VMEM issue at the same time,
    // and the counted s_waitcnt in front of each cout fragment only covers loads issued >= 3/4 of a chunk earlier.
    auto chunk = [&](int kc, float4 (&xc)[PF], const bool (&okc)[PF], const float4 (&wc)[NT], float4 (&xn)[PF], bool (&okn)[PF], float4 (&wn)[NT]) {
        const int kn = min(kc + 1, p.KC - 1);   // the last chunk re-loads itself: unconditional loads keep the waits counted
        constexpr int NM = NT * PF * 4, NL = PF + NT;
        constexpr int GAP = NM / (NL + 1) > 0 ? NM / (NL + 1) : 1;
        int cnt = 0, li = 0;
        if (!IS1X1) {
#pragma clang loop unroll(full)
            for (int pf = 0; pf < PF; ++pf) {
                float4 v = xc[pf];
                v.x = okc[pf] ? v.x : 0.f; v.y = okc[pf] ? v.y : 0.f; v.z = okc[pf] ? v.z : 0.f; v.w = okc[pf] ? v.w : 0.f;
                xc[pf] = v;
            }
        }
#pragma clang loop unroll(full)
        for (int nf = 0; nf < NT; ++nf)
#pragma clang loop unroll(full)
            for (int j = 0; j < 4; ++j)
#pragma clang loop unroll(full)
                for (int pf = 0; pf < PF; ++pf) {
                    acc[nf][pf] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(wc[nf], j), comp(xc[pf], j), acc[nf][pf], 0, 0, 0);
                    ++cnt;
                    if (cnt % GAP == 0 && li < NL) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (li < PF) load_x1(kn, li, xn[li], okn[li]);
                        else load_w1(kn, li - PF, wn[li - PF]);
                        __builtin_amdgcn_sched_barrier(0);
                        ++li;
                    }
                }
    };

    float4 xa[PF], xb[PF], wa[NT], wb[NT];
    bool oka[PF], okb[PF];
#pragma clang loop unroll(full)
    for (int pf = 0; pf < PF; ++pf) load_x1(0, pf, xa[pf], oka[pf]);
#pragma clang loop unroll(full)
    for (int nf = 0; nf < NT; ++nf) load_w1(0, nf, wa[nf]);
    int kc = 0;
    for (; kc + 1 < p.KC; kc += 2) {
        chunk(kc, xa, oka, wa, xb, okb, wb);
        chunk(kc + 1, xb, okb, wb, xa, oka, wa);
    }
    if (kc < p.KC) chunk(kc, xa, oka, wa, xb, okb, wb);

    // epilogue: bias + residual + activation, 16-byte NHWC stores.  unroll(full): the accumulators must stay in
    // registers (a partially unrolled loop indexes acc[][] dynamically and the compiler moves it to scratch).
    const bool vec_ok = ((p.Cout & 3) == 0) && ((p.y_ld & 3) == 0);
#pragma clang loop unroll(full)
    for (int pf = 0; pf < PF; ++pf) {
        const long m = m0 + pf * 16 + pl_;
        long obase = m;
        if (p.convt && m < p.M) {
            long hw = (long)p.Ho * p.Wo;  // here Ho/Wo are the INPUT spatial dims of the convT
            long n = m / hw; long r = m - n * hw;
            int h = (int)(r / p.Wo), w = (int)(r - (long)h * p.Wo);
            obase = (n * (2L * p.Ho) + 2L * h) * (2L * p.Wo) + 2L * w;  // pixel index of (2h, 2w)
        }
#pragma clang loop unroll(full)
        for (int nf = 0; nf < NT; ++nf) {
            const f32x4 v = acc[nf][pf];
            igemm_store(p, v, m < p.M, obase, (nf0 + nf) * 16 + g * 4, vec_ok);
        }
    }
}

// ------------------------------------------------------------------------------------------ igemm k32 (f32)
// Same math as conv_igemm_kernel, 32-deep K chunks: lane (p, g) owns the 8 consecutive k = 32*kc + 8*g + 4*h + j
// (h = half, j = float4 component), i.e. the four g-lanes of a pixel read one whole 128-byte line per chunk instead
// of half a line per 16-chunk (the other half was re-fetched from L2 a chunk later: L1 does not hold 16 waves x 32
// lines).  Weights: Wk[cout/16][K/32][h][lane][4].  Pipeline per chunk: X(kc+1) is requested after the first k-step
// and consumed a whole chunk (8 k-steps) later; the W halves ping-pong, each requested one half (4 k-steps) ahead.
template <int NT, int PF, bool IS1X1>
__global__ __launch_bounds__(256, PF <= 2 ? 4 : 2) void conv_igemm_k32_kernel(IgemmP p) {   // PF <= 2: cap at 128 VGPRs (4 waves/SIMD)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pl_ = lane & 15, g = lane >> 4;
    const long b = blockIdx.x;
    const int xcd = (int)(b & 7);
    const long slot = b >> 3;
    const int ntile = (int)(slot % p.ny);
    const long mtile = (long)xcd * p.mx_per_xcd + slot / p.ny;
    if (slot / p.ny >= p.mx_per_xcd) return;
    const long m0 = (mtile * 4 + wave) * (PF * 16);
    const int nf0 = ntile * NT;
    if (m0 >= p.M) return;

    long pix_base[PF];
    int ih0[PF], iw0[PF];
#pragma unroll
    for (int pf = 0; pf < PF; ++pf) {
        long m = min(m0 + pf * 16 + pl_, p.M - 1);   // clamped rows compute garbage that is never stored
        if (IS1X1) {
            pix_base[pf] = m * (long)p.Cin; ih0[pf] = 0; iw0[pf] = 0;
        } else {
            long hw = (long)p.Ho * p.Wo;
            long n = m / hw; long r = m - n * hw;
            int oh = (int)(r / p.Wo), ow = (int)(r - (long)oh * p.Wo);
            pix_base[pf] = n * (long)p.H * p.W * p.Cin;
            ih0[pf] = oh * p.sh - p.pt; iw0[pf] = ow * p.sw - p.pl;
        }
    }
    f32x4 acc[NT][PF];
#pragma clang loop unroll(full)
    for (int a = 0; a < NT; ++a)
#pragma clang loop unroll(full)
        for (int q = 0; q < PF; ++q) acc[a][q] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // (tap_h, tap_w, ci) of this lane's two quads, tracked incrementally (general case only)
    int ci[2] = {8 * g, 8 * g + 4}, tap_h[2] = {0, 0}, tap_w[2] = {0, 0};
    if (!IS1X1) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
            while (ci[h] >= p.Cin) { ci[h] -= p.Cin; if (++tap_w[h] == p.kw) { tap_w[h] = 0; ++tap_h[h]; } }
    }
    const float4* wf = reinterpret_cast<const float4*>(p.w) + ((long)nf0 * p.KC) * 128 + lane;

    auto load_x = [&](int kc, float4 (&xv)[PF][2]) {
#pragma clang loop unroll(full)
        for (int h = 0; h < 2; ++h) {
            const int k = min(kc * 32 + 8 * g + 4 * h, p.K - 4);   // zero-padded tail of W: re-read a valid quad
#pragma clang loop unroll(full)
            for (int pf = 0; pf < PF; ++pf) {
                float4 v;
                if (IS1X1) {
                    v = *reinterpret_cast<const float4*>(p.x + pix_base[pf] + k);
                } else {
                    const int ih = ih0[pf] + tap_h[h] * p.dh, iw = iw0[pf] + tap_w[h] * p.dw;
                    const bool ok = ih >= 0 && ih < p.H && iw >= 0 && iw < p.W && tap_h[h] < p.kh;
                    const int ihc = min(max(ih, 0), p.H - 1), iwc = min(max(iw, 0), p.W - 1);
                    v = *reinterpret_cast<const float4*>(p.x + pix_base[pf] + ((long)ihc * p.W + iwc) * p.Cin + min(ci[h], p.Cin - 4));
                    v.x = ok ? v.x : 0.f; v.y = ok ? v.y : 0.f; v.z = ok ? v.z : 0.f; v.w = ok ? v.w : 0.f;
                }
                xv[pf][h] = v;
            }
            if (!IS1X1) {
                ci[h] += 32;
                while (ci[h] >= p.Cin) { ci[h] -= p.Cin; if (++tap_w[h] == p.kw) { tap_w[h] = 0; ++tap_h[h]; } }
            }
        }
    };
    auto load_w = [&](int kc, int h, float4 (&wv)[NT]) {
#pragma clang loop unroll(full)
        for (int nf = 0; nf < NT; ++nf)
            wv[nf] = wf[(((long)nf * p.KC + kc) * 2 + h) * 64];
    };
    auto mma_step = [&](const float4 (&wv)[NT], const float4 (&xv)[PF][2], auto H, auto comp) {
#pragma clang loop unroll(full)
        for (int nf = 0; nf < NT; ++nf)
#pragma clang loop unroll(full)
            for (int pf = 0; pf < PF; ++pf)
                acc[nf][pf] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(wv[nf]), comp(xv[pf][decltype(H)::value]), acc[nf][pf], 0, 0, 0);
    };
    auto cx = [](const float4& v) { return v.x; };
    auto cy = [](const float4& v) { return v.y; };
    auto cz = [](const float4& v) { return v.z; };
    auto cw = [](const float4& v) { return v.w; };
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;

    float4 xa[PF][2], xb[PF][2], w0[NT], w1[NT];
    // one chunk: cur = X(kc) (resident), nxt <- X(kc+1); w0 = W(kc, 0) resident, w1 = W(kc, 1) in flight
    auto chunk = [&](int kc, float4 (&cur)[PF][2], float4 (&nxt)[PF][2]) {
        const bool more = kc + 1 < p.KC;
        mma_step(w0, cur, H0{}, cx);
        __builtin_amdgcn_sched_barrier(0);
        if (more) load_x(kc + 1, nxt);
        __builtin_amdgcn_sched_barrier(0);
        mma_step(w0, cur, H0{}, cy); mma_step(w0, cur, H0{}, cz); mma_step(w0, cur, H0{}, cw);
        __builtin_amdgcn_sched_barrier(0);
        if (more) load_w(kc + 1, 0, w0);
        __builtin_amdgcn_sched_barrier(0);
        mma_step(w1, cur, H1{}, cx); mma_step(w1, cur, H1{}, cy); mma_step(w1, cur, H1{}, cz); mma_step(w1, cur, H1{}, cw);
        __builtin_amdgcn_sched_barrier(0);
        if (more) load_w(kc + 1, 1, w1);
        __builtin_amdgcn_sched_barrier(0);
    };
    load_x(0, xa);
    load_w(0, 0, w0);
    load_w(0, 1, w1);
    int kc = 0;
    for (; kc + 1 < p.KC; kc += 2) { chunk(kc, xa, xb); chunk(kc + 1, xb, xa); }
    if (kc < p.KC) chunk(kc, xa, xb);

    const bool vec_ok = ((p.Cout & 3) == 0) && ((p.y_ld & 3) == 0);
#pragma clang loop unroll(full)
    for (int pf = 0; pf < PF; ++pf) {
        const long m = m0 + pf * 16 + pl_;
        long obase = m;
        if (p.convt && m < p.M) {
            long hw = (long)p.Ho * p.Wo;
            long n = m / hw; long r = m - n * hw;
            int h = (int)(r / p.Wo), w = (int)(r - (long)h * p.Wo);
            obase = (n * (2L * p.Ho) + 2L * h) * (2L * p.Wo) + 2L * w;
        }
#pragma clang loop unroll(full)
        for (int nf = 0; nf < NT; ++nf) {
            const f32x4 v = acc[nf][pf];
            igemm_store(p, v, m < p.M, obase, (nf0 + nf) * 16 + g * 4, vec_ok);
        }
    }
}

// ------------------------------------------------------------------------------------------ igemm x6: f32 via 6 bf16 MFMAs
// f32-equivalent GEMM on the bf16 matrix pipe.  Every f32 operand is split EXACTLY into three bf16 pieces by
// truncation (x = h + m + l, each piece = the next 8 significand bits: h = x & 0xFFFF0000, m = (x-h) & .., l = ...),
// and the product is accumulated in f32 from the six terms hh, hm, mh, hl, lh, mm.  The dropped terms (ml, lm, ll)
// are <= 2^-24 relative, i.e. the result carries the same ~1 ulp error class as a plain f32 FMA chain (measured
// 4e-8 vs 6e-8 relative on K = 192 dot products) -- but one 32-deep k-step costs 6 x 16 cycles of
// v_mfma_f32_16x16x32_bf16 instead of 8 x 32 cycles of v_mfma_f32_16x16x4_f32: 2.7x less matrix-pipe time.
// Operand map: lane (p = lane&15, g = lane>>4) supplies the 8 consecutive k = 32*kc + 8*g + e of its row/column for
// both A (weights, pre-split on the host into 3 planes of 8 bf16 per lane) and B (pixels: two float4 loads = 32
// contiguous bytes, split in registers).  C/D layout equals the f32 kernel's, so the epilogue is shared.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void split3_pack(const float4& a, const float4& b, uint4& h, uint4& m, uint4& l) {
    const float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    unsigned hh[8], mm[8], ll[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        unsigned u = __float_as_uint(f[e]);
        unsigned uh = u & 0xFFFF0000u;
        float r1 = f[e] - __uint_as_float(uh);
        unsigned um = __float_as_uint(r1) & 0xFFFF0000u;
        float r2 = r1 - __uint_as_float(um);
        unsigned ul = __float_as_uint(r2) & 0xFFFF0000u;
        hh[e] = uh; mm[e] = um; ll[e] = ul;
    }
    // bf16 element e = upper half of piece e; two per dword, element 2i in the low half
    h = make_uint4((hh[0] >> 16) | hh[1], (hh[2] >> 16) | hh[3], (hh[4] >> 16) | hh[5], (hh[6] >> 16) | hh[7]);
    m = make_uint4((mm[0] >> 16) | mm[1], (mm[2] >> 16) | mm[3], (mm[4] >> 16) | mm[5], (mm[6] >> 16) | mm[7]);
    l = make_uint4((ll[0] >> 16) | ll[1], (ll[2] >> 16) | ll[3], (ll[4] >> 16) | ll[5], (ll[6] >> 16) | ll[7]);
}

template <int NT, int PF, bool IS1X1>
__global__ __launch_bounds__(256, 2) void conv_igemm_x6_kernel(IgemmP p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pl_ = lane & 15, g = lane >> 4;
    const long b = blockIdx.x;
    const int xcd = (int)(b & 7);
    const long slot = b >> 3;
    const int ntile = (int)(slot % p.ny);
    const long mtile = (long)xcd * p.mx_per_xcd + slot / p.ny;
    if (slot / p.ny >= p.mx_per_xcd) return;
    const long m0 = (mtile * 4 + wave) * (PF * 16);
    const int nf0 = ntile * NT;
    if (m0 >= p.M) return;

    long pix_base[PF];
    int ih0[PF], iw0[PF];
#pragma unroll
    for (int pf = 0; pf < PF; ++pf) {
        // rows past M are clamped to the last pixel: they compute garbage that igemm_store never writes.  Keeping
        // every load unconditional matters: a load under a bounds-check branch makes the compiler's s_waitcnt
        // bookkeeping conservative (vmcnt(0) right after the prefetch is issued), which serialises the pipeline.
        long m = min(m0 + pf * 16 + pl_, p.M - 1);
        if (IS1X1) {
            pix_base[pf] = m * (long)p.Cin; ih0[pf] = 0; iw0[pf] = 0;
        } else {
            long hw = (long)p.Ho * p.Wo;
            long n = m / hw; long r = m - n * hw;
            int oh = (int)(r / p.Wo), ow = (int)(r - (long)oh * p.Wo);
            pix_base[pf] = n * (long)p.H * p.W * p.Cin;
            ih0[pf] = oh * p.sh - p.pt; iw0[pf] = ow * p.sw - p.pl;
        }
    }
    f32x4 acc[NT][PF];
#pragma clang loop unroll(full)
    for (int a = 0; a < NT; ++a)
#pragma clang loop unroll(full)
        for (int q = 0; q < PF; ++q) acc[a][q] = (f32x4){0.f, 0.f, 0.f, 0.f};

    int ci = 8 * g, tap_h = 0, tap_w = 0;   // this lane's k = 32*kc + 8*g
    if (!IS1X1) {
        while (ci >= p.Cin) { ci -= p.Cin; if (++tap_w == p.kw) { tap_w = 0; ++tap_h; } }
    }
    const uint4* wf = reinterpret_cast<const uint4*>(p.w) + ((long)nf0 * p.KC) * 3 * 64 + lane;

    auto load_x = [&](int kc, float4 (&xv)[PF][2]) {
        const int k = min(kc * 32 + 8 * g, p.K - 8);   // padded tail: W is 0 there, re-read the last valid group
#pragma clang loop unroll(full)
        for (int pf = 0; pf < PF; ++pf) {
            float4 v0, v1;
            if (IS1X1) {
                const float* src = p.x + pix_base[pf] + k;
                v0 = *reinterpret_cast<const float4*>(src); v1 = *reinterpret_cast<const float4*>(src + 4);
            } else {
                const int ih = ih0[pf] + tap_h * p.dh, iw = iw0[pf] + tap_w * p.dw;
                const bool ok = ih >= 0 && ih < p.H && iw >= 0 && iw < p.W && tap_h < p.kh;
                const int ihc = min(max(ih, 0), p.H - 1), iwc = min(max(iw, 0), p.W - 1);
                const float* src = p.x + pix_base[pf] + ((long)ihc * p.W + iwc) * p.Cin + min(ci, p.Cin - 8);
                v0 = *reinterpret_cast<const float4*>(src); v1 = *reinterpret_cast<const float4*>(src + 4);
                v0.x = ok ? v0.x : 0.f; v0.y = ok ? v0.y : 0.f; v0.z = ok ? v0.z : 0.f; v0.w = ok ? v0.w : 0.f;
                v1.x = ok ? v1.x : 0.f; v1.y = ok ? v1.y : 0.f; v1.z = ok ? v1.z : 0.f; v1.w = ok ? v1.w : 0.f;
            }
            xv[pf][0] = v0; xv[pf][1] = v1;
        }
        if (!IS1X1) {
            ci += 32;
            while (ci >= p.Cin) { ci -= p.Cin; if (++tap_w == p.kw) { tap_w = 0; ++tap_h; } }
        }
    };
    auto load_w = [&](int kc, uint4 (&wv)[NT][3]) {
#pragma clang loop unroll(full)
        for (int nf = 0; nf < NT; ++nf)
#pragma clang loop unroll(full)
            for (int pl = 0; pl < 3; ++pl) wv[nf][pl] = wf[(((long)nf * p.KC + kc) * 3 + pl) * 64];
    };
    // six terms per (cout frag, pixel frag), smallest first: (w plane, x plane) = mm, lh, hl, mh, hm, hh.  Term
    // outermost so that consecutive MFMAs target different accumulators.  Terms [T0, T1) of one chunk:
    auto split = [&](const float4 (&xv)[PF][2], uint4 (&xs)[PF][3]) {
#pragma clang loop unroll(full)
        for (int pf = 0; pf < PF; ++pf) split3_pack(xv[pf][0], xv[pf][1], xs[pf][0], xs[pf][1], xs[pf][2]);
    };
    auto mma_terms = [&](const uint4 (&wv)[NT][3], const uint4 (&xs)[PF][3], auto T0, auto T1) {
        constexpr int WP[6] = {1, 2, 0, 1, 0, 0};
        constexpr int XP[6] = {1, 0, 2, 0, 1, 0};
#pragma clang loop unroll(full)
        for (int t = decltype(T0)::value; t < decltype(T1)::value; ++t)
#pragma clang loop unroll(full)
            for (int nf = 0; nf < NT; ++nf)
#pragma clang loop unroll(full)
                for (int pf = 0; pf < PF; ++pf)
                    acc[nf][pf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv[nf][WP[t]]),
                                                                          __builtin_bit_cast(bf16x8, xs[pf][XP[t]]), acc[nf][pf], 0, 0, 0);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;
    using I6 = std::integral_constant<int, 6>;

    // same issue-order pinning as the f32 kernel: first term of the chunk, then the next chunk's loads, then the rest
    float4 xa[PF][2], xb[PF][2];
    uint4 wa[NT][3], wb[NT][3], xs[PF][3];
    load_x(0, xa);
    load_w(0, wa);
    int kc = 0;
    for (; kc + 1 < p.KC; kc += 2) {
        split(xa, xs);
        mma_terms(wa, xs, I0{}, I1{});
        __builtin_amdgcn_sched_barrier(0);
        load_x(kc + 1, xb);
        load_w(kc + 1, wb);
        __builtin_amdgcn_sched_barrier(0);
        mma_terms(wa, xs, I1{}, I6{});
        __builtin_amdgcn_sched_barrier(0);   // keep the split of the prefetched chunk (and its vmcnt wait) down here
        split(xb, xs);
        mma_terms(wb, xs, I0{}, I1{});
        __builtin_amdgcn_sched_barrier(0);
        if (kc + 2 < p.KC) { load_x(kc + 2, xa); load_w(kc + 2, wa); }
        __builtin_amdgcn_sched_barrier(0);
        mma_terms(wb, xs, I1{}, I6{});
        __builtin_amdgcn_sched_barrier(0);
    }
    if (kc < p.KC) { split(xa, xs); mma_terms(wa, xs, I0{}, I6{}); }

    const bool vec_ok = ((p.Cout & 3) == 0) && ((p.y_ld & 3) == 0);
#pragma clang loop unroll(full)
    for (int pf = 0; pf < PF; ++pf) {
        const long m = m0 + pf * 16 + pl_;
        long obase = m;
        if (p.convt && m < p.M) {
            long hw = (long)p.Ho * p.Wo;
            long n = m / hw; long r = m - n * hw;
            int h = (int)(r / p.Wo), w = (int)(r - (long)h * p.Wo);
            obase = (n * (2L * p.Ho) + 2L * h) * (2L * p.Wo) + 2L * w;
        }
#pragma clang loop unroll(full)
        for (int nf = 0; nf < NT; ++nf) {
            const f32x4 v = acc[nf][pf];
            igemm_store(p, v, m < p.M, obase, (nf0 + nf) * 16 + g * 4, vec_ok);
        }
    }
}

int igemm_weight_format(int K, int Cin, bool is1x1) {
    // OAR_IGEMM_FMT: -1 auto (default), 0 force k16, 1 x6 where eligible, 2 k32 everywhere
    static const int mode = [] { const char* e = getenv("OAR_IGEMM_FMT"); return e ? atoi(e) : -1; }();
    static const int k32_min = [] { const char* e = getenv("OAR_IGEMM_K32_MIN"); return e ? atoi(e) : (1 << 30); }();
    if (mode == 0) return IGEMM_W_K16;
    if (mode == 1) {
        // x6: every lane's 8-float group must be all-valid or all-padding, and must not straddle two taps
        if (K % 8 == 0 && (is1x1 || Cin % 8 == 0)) return IGEMM_W_X6;
        return IGEMM_W_K16;
    }
    if (mode == 2) return IGEMM_W_K32;
    return K >= k32_min ? IGEMM_W_K32 : IGEMM_W_K16;
}

void conv_igemm(hipStream_t s, const ConvP& c) {
    IgemmP p;
    p.x = c.x; p.w = c.w; p.bias = c.bias; p.res = c.residual; p.y = c.y;
    p.H = c.H; p.W = c.W; p.Cin = c.Cin; p.kh = c.kh; p.kw = c.kw; p.sh = c.sh; p.sw = c.sw;
    p.pt = c.pt; p.pl = c.pl; p.dh = c.dh; p.dw = c.dw; p.y_ld = c.y_ld;
    p.act = c.act.kind; p.alpha = c.act.alpha; p.beta = c.act.beta;
    p.convt = c.convt2x2; p.Cout = c.Cout;
    if (c.convt2x2) {
        p.Ho = c.H; p.Wo = c.W;  // GEMM columns are INPUT pixels
        p.M = (long)c.N * c.H * c.W; p.K = c.Cin; p.gemm_cout = 4 * c.Cout;
    } else {
        p.Ho = c.Ho; p.Wo = c.Wo;
        p.M = (long)c.N * c.Ho * c.Wo; p.K = c.kh * c.kw * c.Cin; p.gemm_cout = c.Cout;
    }
    const bool x6 = c.w_fmt == IGEMM_W_X6, k32 = c.w_fmt == IGEMM_W_K32;
    p.KC = (x6 || k32) ? (p.K + 31) / 32 : (p.K + 15) / 16;
    if (p.M == 0) return;
    const bool is1x1 = c.convt2x2 || (c.kh == 1 && c.kw == 1 && c.sh == 1 && c.sw == 1 && c.pt == 0 && c.pl == 0);
    int nfrag = (p.gemm_cout + 15) / 16;
    // cout fragments per wave: minimise padded (wasted) MFMA work, prefer the larger tile on ties
    int NT = 1;
    {
        int best = 1 << 30;
        for (int t = 4; t >= 1; --t) { int padded = (nfrag + t - 1) / t * t; if (padded < best) { best = padded; NT = t; } }
    }
    // pixel fragments per wave: fewer when the launch would otherwise leave most of the 256 CUs idle
    const long ny = (nfrag + NT - 1) / NT;
    static const int pf_max = [] { const char* e = getenv("OAR_IGEMM_PF"); int v = e ? atoi(e) : 2; return v == 1 || v == 4 ? v : 2; }();  // PF=2: 102 VGPRs => 4 waves/SIMD (measured 1.2x over PF=4)
    int PF = (x6 || k32) ? 2 : pf_max;
    while (PF > 1 && ((p.M + 4L * PF * 16 - 1) / (4L * PF * 16)) * ny < 1024) PF >>= 1;
    const long mx = (p.M + 4L * PF * 16 - 1) / (4L * PF * 16);
    p.ny = (int)ny;
    OAR_CHECK(p.K < 65536, OAR_UNSUPPORTED_OP, "conv_igemm: K = kh*kw*Cin must be < 65536");
    p.cin_magic = (unsigned)((1ull << 32) / (unsigned)c.Cin + 1);
    p.kw_magic = (unsigned)((1ull << 32) / (unsigned)c.kw + 1);
    p.mx_per_xcd = (mx + 7) / 8;
    dim3 grid((unsigned)(p.mx_per_xcd * 8 * ny));
    // (A variant with the cout tile of W resident in LDS and chunk-pair X prefetch was measured at parity with this
    // kernel -- 78-81 TFLOP/s on the K=192/256 shapes either way -- and dropped.)
    double flops = 2.0 * (double)p.M * p.K * p.gemm_cout;
    double bytes = 4.0 * ((double)c.N * c.H * c.W * c.Cin + (double)p.M * p.gemm_cout + (double)p.K * p.gemm_cout);
    char pname[96];
    const char* cls = "conv_igemm";
    if (Profiler::get().detail) {
        snprintf(pname, sizeof pname, "conv_igemm%s M=%ld K=%d N=%d k%dx%d s%d%s", x6 ? "_x6" : k32 ? "_k32" : "", p.M, p.K, p.gemm_cout, c.kh, c.kw, c.sh, c.convt2x2 ? " convT" : "");
        cls = pname;
    }
    ProfScope ps(s, cls, bytes, flops);
#define LAUNCH2(NTV, PFV)                                                                                          \
    do {                                                                                                          \
        if (x6) {                                                                                                 \
            if (is1x1) hipLaunchKernelGGL((conv_igemm_x6_kernel<NTV, PFV, true>), grid, dim3(256), 0, s, p);     \
            else hipLaunchKernelGGL((conv_igemm_x6_kernel<NTV, PFV, false>), grid, dim3(256), 0, s, p);          \
        } else if (k32) {                                                                                         \
            if (is1x1) hipLaunchKernelGGL((conv_igemm_k32_kernel<NTV, PFV, true>), grid, dim3(256), 0, s, p);    \
            else hipLaunchKernelGGL((conv_igemm_k32_kernel<NTV, PFV, false>), grid, dim3(256), 0, s, p);         \
        } else if (is1x1) hipLaunchKernelGGL((conv_igemm_kernel<NTV, PFV, true>), grid, dim3(256), 0, s, p);     \
        else hipLaunchKernelGGL((conv_igemm_kernel<NTV, PFV, false>), grid, dim3(256), 0, s, p);                 \
    } while (0)
#define LAUNCH(NTV)                                  \
    do {                                             \
        if (PF == 4) LAUNCH2(NTV, 4);                \
        else if (PF == 2) LAUNCH2(NTV, 2);           \
        else LAUNCH2(NTV, 1);                        \
    } while (0)
    if (NT == 4) LAUNCH(4);
    else if (NT == 3) LAUNCH(3);
    else if (NT == 2) LAUNCH(2);
    else LAUNCH(1);
#undef LAUNCH2
#undef LAUNCH
}

// ------------------------------------------------------------------------------------------ depthwise conv
// Bandwidth kernel. One thread = 4 channels (one float4) x TW adjacent output pixels of a row: each input row of
// the window is loaded once ((TW-1)*S + K float4s) and reused by all TW outputs, cutting the load count per output
// from K*K to K*((TW-1)*S+K)/TW.  Lanes run along channels first, so a wave reads/writes whole 16-byte-per-lane
// contiguous NHWC segments.  Weights [kh][kw][C].
template <int K, int SH, int S, int TW>
__global__ __launch_bounds__(256) void conv_dw_tiled_kernel(ConvP p) {
    const int C4 = p.Cout >> 2;
    const int wtiles = (p.Wo + TW - 1) / TW;
    const long total = (long)p.N * p.Ho * wtiles * C4;
    constexpr int NCOL = (TW - 1) * S + K;
    // each XCD (workgroup id % 8) walks one contiguous band of output rows => the K-row halo re-reads stay in its L2
    const long per_xcd = (total + 7) / 8;
    const int xcd = blockIdx.x & 7;
    const long lb = blockIdx.x >> 3, nlb = (gridDim.x + 7) >> 3;
    const long band_end = min(total, (long)(xcd + 1) * per_xcd);
    for (long i = (long)xcd * per_xcd + lb * blockDim.x + threadIdx.x; i < band_end; i += nlb * blockDim.x) {
        int c4 = (int)(i % C4); long t = i / C4;
        int wt = (int)(t % wtiles); t /= wtiles;
        int oh = (int)(t % p.Ho); long n = t / p.Ho;
        const int c = c4 * 4, ow0 = wt * TW;
        float4 bias = p.bias ? *reinterpret_cast<const float4*>(p.bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 acc[TW];
#pragma unroll
        for (int q = 0; q < TW; ++q) acc[q] = bias;
        const float* xb = p.x + n * (long)p.H * p.W * p.Cin + c;
        const int iw0 = ow0 * S - p.pl;
#pragma unroll
        for (int a = 0; a < K; ++a) {
            const int ih = oh * SH - p.pt + a;
            if (ih < 0 || ih >= p.H) continue;
            const float* xr = xb + (long)ih * p.W * p.Cin;
            float4 col[NCOL];
#pragma unroll
            for (int q = 0; q < NCOL; ++q) {
                int iw = iw0 + q;
                col[q] = (iw >= 0 && iw < p.W) ? *reinterpret_cast<const float4*>(xr + (long)iw * p.Cin) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int b = 0; b < K; ++b) {
                float4 wv = *reinterpret_cast<const float4*>(p.w + (long)(a * K + b) * p.Cout + c);
#pragma unroll
                for (int q = 0; q < TW; ++q) {
                    float4 xv = col[q * S + b];
                    acc[q].x = fmaf(xv.x, wv.x, acc[q].x); acc[q].y = fmaf(xv.y, wv.y, acc[q].y);
                    acc[q].z = fmaf(xv.z, wv.z, acc[q].z); acc[q].w = fmaf(xv.w, wv.w, acc[q].w);
                }
            }
        }
        const long pix0 = (n * p.Ho + oh) * (long)p.Wo + ow0;
#pragma unroll
        for (int q = 0; q < TW; ++q) {
            if (ow0 + q >= p.Wo) break;
            long o = (pix0 + q) * p.y_ld + c;
            float4 v = acc[q];
            if (p.residual) { float4 r = *reinterpret_cast<const float4*>(p.residual + o); v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w; }
            v.x = apply_act(v.x, p.act.kind, p.act.alpha, p.act.beta); v.y = apply_act(v.y, p.act.kind, p.act.alpha, p.act.beta);
            v.z = apply_act(v.z, p.act.kind, p.act.alpha, p.act.beta); v.w = apply_act(v.w, p.act.kind, p.act.alpha, p.act.beta);
            *reinterpret_cast<float4*>(p.y + o) = v;
        }
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

void conv_dw(hipStream_t s, const ConvP& p) {
    long total = (long)p.N * p.Ho * p.Wo * (p.Cout / 4);
    if (total == 0) return;
    double bytes = 4.0 * ((double)p.N * p.H * p.W * p.Cin + (double)p.N * p.Ho * p.Wo * p.Cout + (double)p.kh * p.kw * p.Cout);
    double flops = 2.0 * (double)p.N * p.Ho * p.Wo * p.Cout * p.kh * p.kw;
    char pname[96];
    const char* cls = "conv_dw";
    if (Profiler::get().detail) { snprintf(pname, sizeof pname, "conv_dw px=%ld C=%d k%d s%d", (long)p.N * p.Ho * p.Wo, p.Cout, p.kh, p.sh); cls = pname; }
    ProfScope ps(s, cls, bytes, flops);
    const bool sq = p.kh == p.kw && p.dh == 1 && p.dw == 1;
    constexpr int TW = 4;
    long tiled = (long)p.N * p.Ho * ((p.Wo + TW - 1) / TW) * (p.Cout / 4);
    dim3 g((grid_for(tiled) + 7) / 8 * 8), b(256);
#define DW(KV, SHV, SWV) hipLaunchKernelGGL((conv_dw_tiled_kernel<KV, SHV, SWV, TW>), g, b, 0, s, p)
    if (sq && p.kh == 3 && p.sh == 1 && p.sw == 1) DW(3, 1, 1);
    else if (sq && p.kh == 3 && p.sh == 2 && p.sw == 2) DW(3, 2, 2);
    else if (sq && p.kh == 3 && p.sh == 2 && p.sw == 1) DW(3, 2, 1);
    else if (sq && p.kh == 3 && p.sh == 1 && p.sw == 2) DW(3, 1, 2);
    else if (sq && p.kh == 5 && p.sh == 1 && p.sw == 1) DW(5, 1, 1);
    else if (sq && p.kh == 5 && p.sh == 2 && p.sw == 2) DW(5, 2, 2);
    else if (sq && p.kh == 5 && p.sh == 2 && p.sw == 1) DW(5, 2, 1);
    else if (sq && p.kh == 5 && p.sh == 1 && p.sw == 2) DW(5, 1, 2);
    else hipLaunchKernelGGL(conv_dw_kernel, dim3(grid_for(total)), b, 0, s, p);
#undef DW
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
        if (!p.is_max) acc = acc / (float)(p.count_include_pad ? p.kh * p.kw : (cnt > 0 ? cnt : 1));
        p.y[i] = acc;
    }
}
void pool2d(hipStream_t s, const PoolP& p) {
    long total = (long)p.N * p.Ho * p.Wo * p.C;
    if (total == 0) return;
    ProfScope ps(s, "pool2d", 4.0 * ((double)p.N * p.H * p.W * p.C + (double)total), 0.0);
    hipLaunchKernelGGL(pool2d_kernel, dim3(grid_for(total)), dim3(256), 0, s, p);
}

// One workgroup per image. Lanes run along channels (float4), `parts` thread groups split HW; the partial sums
// are combined through LDS in a fixed order (deterministic).
__global__ __launch_bounds__(256) void global_avgpool_kernel(const float* x, float* y, int HW, int C) {
    extern __shared__ float4 red[];  // [parts][C4]
    const int C4 = C >> 2;
    const int parts = 256 / C4 > 0 ? 256 / C4 : 1;
    const int n = blockIdx.x;
    const int c4 = threadIdx.x % C4, part = threadIdx.x / C4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (part < parts) {
        const float4* xb = reinterpret_cast<const float4*>(x + (long)n * HW * C) + c4;
        int i = part;
        for (; i + 3 * parts < HW; i += 4 * parts) {
            float4 a = xb[(long)i * C4], b = xb[(long)(i + parts) * C4], c = xb[(long)(i + 2 * parts) * C4], d = xb[(long)(i + 3 * parts) * C4];
            acc.x += (a.x + b.x) + (c.x + d.x); acc.y += (a.y + b.y) + (c.y + d.y);
            acc.z += (a.z + b.z) + (c.z + d.z); acc.w += (a.w + b.w) + (c.w + d.w);
        }
        for (; i < HW; i += parts) { float4 a = xb[(long)i * C4]; acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w; }
        red[part * C4 + c4] = acc;
    }
    __syncthreads();
    if (part == 0) {
        float4 t = red[c4];
        for (int q = 1; q < parts; ++q) { float4 u = red[q * C4 + c4]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
        float inv = (float)HW;
        reinterpret_cast<float4*>(y + (long)n * C)[c4] = make_float4(t.x / inv, t.y / inv, t.z / inv, t.w / inv);
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
void global_avgpool(hipStream_t s, const float* x, float* y, int N, int HW, int C) {
    if (N == 0 || C == 0) return;
    ProfScope ps(s, "global_avgpool", 4.0 * (double)N * HW * C, 0.0);
    if ((C & 3) == 0 && C / 4 <= 256) {
        int C4 = C / 4, parts = 256 / C4;
        hipLaunchKernelGGL(global_avgpool_kernel, dim3(N), dim3(256), (size_t)parts * C4 * sizeof(float4), s, x, y, HW, C);
    } else {
        hipLaunchKernelGGL(global_avgpool_scalar_kernel, dim3((C + 255) / 256, N), dim3(256), 0, s, x, y, HW, C);
    }
}

// ------------------------------------------------------------------------------------------ resize
__device__ __forceinline__ float src_coord(int o, float scale, int in, int out, int ctm) {
    switch (ctm) {
        case 0: return (float)o / scale;
        case 2: return out > 1 ? (float)o * (float)(in - 1) / (float)(out - 1) : 0.f;
        case 3: return out > 1 ? ((float)o + 0.5f) / scale - 0.5f : 0.f;
        default: return ((float)o + 0.5f) / scale - 0.5f;
    }
}
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
            float ry, rx;
            switch (nm) {
                case 0: ry = floorf(fy); rx = floorf(fx); break;
                case 3: ry = ceilf(fy); rx = ceilf(fx); break;
                case 2: ry = floorf(fy + 0.5f); rx = floorf(fx + 0.5f); break;
                default: ry = ceilf(fy - 0.5f); rx = ceilf(fx - 0.5f); break;
            }
            int iy = min(max((int)ry, 0), H - 1), ix = min(max((int)rx, 0), W - 1);
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

// ------------------------------------------------------------------------------------------ elementwise
__global__ __launch_bounds__(256) void unary_kernel(const float* x, float* y, long n, int kind, float alpha, float beta) {
    long n4 = n >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 v = reinterpret_cast<const float4*>(x)[i];
        v.x = apply_act(v.x, kind, alpha, beta); v.y = apply_act(v.y, kind, alpha, beta);
        v.z = apply_act(v.z, kind, alpha, beta); v.w = apply_act(v.w, kind, alpha, beta);
        reinterpret_cast<float4*>(y)[i] = v;
    }
    for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        y[i] = apply_act(x[i], kind, alpha, beta);
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
    ProfScope ps(s, "binary", 12.0 * (double)n, (double)n);
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
void layernorm(hipStream_t s, const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int C, float eps) {
    if (rows == 0) return;
    ProfScope ps(s, "layernorm", 8.0 * (double)rows * C, 8.0 * (double)rows * C);
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
__global__ __launch_bounds__(256) void softmax_argmax_kernel(const float* x, int C, int64_t* idx, float* prob) {
    extern __shared__ float rowbuf[];
    __shared__ float red[4];
    __shared__ int redi[4];
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
void softmax_argmax(hipStream_t s, const float* logits, int64_t rows, int C, int64_t* idx, float* prob) {
    if (rows == 0 || C == 0) return;
    OAR_CHECK((size_t)C * 4 <= 150 * 1024, OAR_UNSUPPORTED_OP, "softmax_argmax: row longer than the LDS staging buffer");
    ProfScope ps(s, "softmax_argmax", 4.0 * (double)rows * C, 4.0 * (double)rows * C);
    hipLaunchKernelGGL(softmax_argmax_kernel, dim3((unsigned)rows), dim3(256), (size_t)C * sizeof(float), s, logits, C, idx, prob);
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

}  // namespace k
}  // namespace oar
