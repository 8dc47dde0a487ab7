// igemm_dev.h -- shared pieces of the implicit-GEMM conv kernels (igemm.hip, igemm_ws_*.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "common.h"
#include "kernels_dev.h"

namespace oar {
namespace k {

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
    unsigned kw_one;                // kw == 1 (see igemm_tap_h)
    long mx_per_xcd;     // pixel tiles per XCD band
    float alpha, beta;
    float* ctc_part;     // != null (weight-stationary f32 kernel only): no logits are stored; per (row, cout tile) the
    int ctc_valid;       // softmax partials {max, sum exp(x - max), last arg max} over the tile's valid columns go here
    int res_up;          // > 1 (per-tile f32 kernels only): res is the low-resolution operand of an FPN sum, read at (h / res_up, w / res_up)
    const float* se;     // != null (bf16x6 weight-stationary kernel only): gate [image][K] multiplied into x on load
    int se_hw;           // pixels per image (se row of pixel m = m / se_hw)
    int n_msrc;          // output-stationary bf16x6 kernel, 1x1 only: > 0 => x is the never-materialised channel concat of these maps (kernels.h ConvP::msrc)
    const float* msrc[8];
    int msrc_c[8];
    int accum;           // row-streaming 3x3 kernel only (igemm_rs3_x6.hip): add to what y holds (a later pass over a channel slice of the input)
    int x_ld;            // output-stationary bf16x6 kernel and igemm_rs3_x6.hip: floats between two pixels of x (>= Cin; > Cin for one group of a grouped convolution,
                         // whose x points at the group's first channel).  Every other kernel reads x with stride Cin
};

// tap / kw through the host's magic number floor(2^32 / kw) + 1 (exact for tap < 2^16).  kw == 1 has no 32-bit magic (2^32 + 1 wraps to 1, umulhi returns 0): a k x 1
// kernel decoded every tap as row 0 until round 6 (tools/op_fuzz.py, "conv 7x1 ... g1": err ~1).  IgemmP::kw_one is 1 for kw == 1 (the quotient is then 0 + tap), else 0 --
// one multiply-add, no select (a select cost the two-fragment per-tile kernel a wave of occupancy: 98 registers).
__device__ __forceinline__ int igemm_tap_h(const IgemmP& p, int tap) { return (int)(__umulhi((unsigned)tap, p.kw_magic) + (unsigned)tap * p.kw_one); }

// element offset of the residual for output pixel `opix`, channel co: the same pixel, or -- res_up > 1 -- pixel (h / f, w / f) of the
// [N][Ho / f][Wo / f][Cout] low-resolution tensor (M < 2^31 checked by the host: 32-bit divisions)
__device__ __forceinline__ long igemm_res_off(const IgemmP& p, long opix, int co) {
    if (p.res_up <= 1) return opix * p.y_ld + co;
    const unsigned hw = (unsigned)(p.Ho * p.Wo), m = (unsigned)opix, f = (unsigned)p.res_up;
    const unsigned n = m / hw, r = m - n * hw, h = r / (unsigned)p.Wo, w = r - h * (unsigned)p.Wo;
    return (((long)n * (p.Ho / p.res_up) + h / f) * (long)(p.Wo / p.res_up) + w / f) * p.Cout + co;
}

__device__ __forceinline__ void igemm_store(const IgemmP& p, f32x4 v, bool valid, long obase, int c, bool vec_ok, bool add_bias = true) {
    if (!valid || c >= p.gemm_cout) return;
    float o[4] = {v[0], v[1], v[2], v[3]};
    if (vec_ok) {
        int co = c; long opix = obase;
        if (p.convt) { int ab = c / p.Cout; co = c - ab * p.Cout; opix = obase + (long)(ab >> 1) * (2L * p.Wo) + (ab & 1); }
        if (p.bias && add_bias) { float4 bv = *reinterpret_cast<const float4*>(p.bias + co); o[0] += bv.x; o[1] += bv.y; o[2] += bv.z; o[3] += bv.w; }
        float* dst = p.y + opix * p.y_ld + co;
        if (p.res) { float4 rv = *reinterpret_cast<const float4*>(p.res + igemm_res_off(p, opix, co)); o[0] += rv.x; o[1] += rv.y; o[2] += rv.z; o[3] += rv.w; }
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
            if (p.bias && add_bias) t += p.bias[co];
            if (p.res) t += p.res[igemm_res_off(p, opix, co)];
            p.y[opix * p.y_ld + co] = apply_act(t, p.act, p.alpha, p.beta);
        }
    }
}

// Epilogue of one wave tile: bias + residual + activation + NHWC stores.  Every load (bias, residual) is issued
// BEFORE the first store: with a load between two stores the compiler waits vmcnt(0) for it, which also waits for the
// previous store to be acknowledged -- NT serialised store round trips per tile (measured: as long as the whole K
// loop).  unroll(full) everywhere: the accumulators must stay in registers.
template <int NT, int PF, bool VEC_ONLY = false>
__device__ __forceinline__ void igemm_epilogue(const IgemmP& p, f32x4 (&acc)[NT][PF], long m0, int pl_, int g, int nf0, bool add_bias) {
    const bool vec_ok = ((p.Cout & 3) == 0) && ((p.y_ld & 3) == 0);
    long obase[PF];
    bool mvalid[PF];
#pragma clang loop unroll(full)
    for (int pf = 0; pf < PF; ++pf) {
        const long m = m0 + pf * 16 + pl_;
        mvalid[pf] = m < p.M;
        const long mc = mvalid[pf] ? m : p.M - 1;
        obase[pf] = mc;
        if (p.convt) {
            long hw = (long)p.Ho * p.Wo;  // here Ho/Wo are the INPUT spatial dims of the convT
            long n = mc / hw; long r = mc - n * hw;
            int h = (int)(r / p.Wo), w = (int)(r - (long)h * p.Wo);
            obase[pf] = (n * (2L * p.Ho) + 2L * h) * (2L * p.Wo) + 2L * w;  // pixel index of (2h, 2w)
        }
    }
    if (!VEC_ONLY && !vec_ok) {
#pragma clang loop unroll(full)
        for (int pf = 0; pf < PF; ++pf)
#pragma clang loop unroll(full)
            for (int nf = 0; nf < NT; ++nf) igemm_store(p, acc[nf][pf], mvalid[pf], obase[pf], (nf0 + nf) * 16 + g * 4, false, add_bias);
        return;
    }
    // this lane's element offset for (nf, pf): recomputed where it is needed instead of kept in NT*PF 64-bit registers
    auto coff = [&](int nf, int pf, bool& cvalid, int& co) -> long {
        const int c = (nf0 + nf) * 16 + g * 4;
        cvalid = c < p.gemm_cout;
        const int cc = cvalid ? c : 0;
        co = cc; long dpix = 0;
        if (p.convt) { int ab = cc / p.Cout; co = cc - ab * p.Cout; dpix = (long)(ab >> 1) * (2L * p.Wo) + (ab & 1); }
        return (obase[pf] + dpix) * p.y_ld + co;
    };
    if (p.bias && add_bias) {
        float4 bv[NT];
#pragma clang loop unroll(full)
        for (int nf = 0; nf < NT; ++nf) { bool cv; int co; (void)coff(nf, 0, cv, co); bv[nf] = *reinterpret_cast<const float4*>(p.bias + co); }
#pragma clang loop unroll(full)
        for (int nf = 0; nf < NT; ++nf)
#pragma clang loop unroll(full)
            for (int pf = 0; pf < PF; ++pf) { acc[nf][pf][0] += bv[nf].x; acc[nf][pf][1] += bv[nf].y; acc[nf][pf][2] += bv[nf].z; acc[nf][pf][3] += bv[nf].w; }
    }
    if (p.res) {
        float4 rv[NT][PF];
#pragma clang loop unroll(full)
        for (int nf = 0; nf < NT; ++nf)
#pragma clang loop unroll(full)
            for (int pf = 0; pf < PF; ++pf) {
                bool cv; int co;
                long ro = coff(nf, pf, cv, co);
                if constexpr (!VEC_ONLY) { if (p.res_up > 1) ro = igemm_res_off(p, obase[pf], co); }   // (only the per-tile kernels take an upsampled residual)
                rv[nf][pf] = *reinterpret_cast<const float4*>(p.res + ro);
            }
#pragma clang loop unroll(full)
        for (int nf = 0; nf < NT; ++nf)
#pragma clang loop unroll(full)
            for (int pf = 0; pf < PF; ++pf) { acc[nf][pf][0] += rv[nf][pf].x; acc[nf][pf][1] += rv[nf][pf].y; acc[nf][pf][2] += rv[nf][pf].z; acc[nf][pf][3] += rv[nf][pf].w; }
    }
    // one (uniform) switch around the whole tile instead of one per element
    auto act_all = [&](auto f) {
#pragma clang loop unroll(full)
        for (int nf = 0; nf < NT; ++nf)
#pragma clang loop unroll(full)
            for (int pf = 0; pf < PF; ++pf)
#pragma clang loop unroll(full)
                for (int r = 0; r < 4; ++r) acc[nf][pf][r] = f(acc[nf][pf][r]);
    };
    switch (p.act) {
        case ACT_NONE: break;
        case ACT_RELU: act_all([](float v) { return v > 0.f ? v : 0.f; }); break;
        case ACT_HSWISH: act_all([](float v) { float t = fminf(fmaxf(v * (1.0f / 6.0f) + 0.5f, 0.f), 1.f); return v * t; }); break;
        default: { const int kind = p.act; const float al = p.alpha, be = p.beta; act_all([=](float v) { return apply_act(v, kind, al, be); }); } break;
    }
#pragma clang loop unroll(full)
    for (int nf = 0; nf < NT; ++nf)
#pragma clang loop unroll(full)
        for (int pf = 0; pf < PF; ++pf) {
            bool cv; int co;
            const long o = coff(nf, pf, cv, co);
            if (mvalid[pf] && cv) *reinterpret_cast<float4*>(p.y + o) = make_float4(acc[nf][pf][0], acc[nf][pf][1], acc[nf][pf][2], acc[nf][pf][3]);
        }
}

// CTC-head epilogue of the weight-stationary kernel: the logits of this wave tile (16 rows x NT*16 columns) never reach
// HBM.  Per row, over the tile's columns c < ctc_valid: m = max, s = sum expf(x - m), i = the LAST column with
// expf(x - m) == 1.0f (the tie rule of the unfused tail: equal probabilities -> last index wins).  ctc_combine merges the
// ny tiles of a row.  Lane (p, g) holds columns (nf0+nf)*16 + g*4 + r of row p; the four g-lanes of a row are merged
// with two xor-shuffles.
template <int NT>
__device__ __forceinline__ void igemm_ctc_epilogue(const IgemmP& p, f32x4 (&acc)[NT][1], long m0, int pl_, int g, int nf0, int ntile, int ny) {
    float m = -3.402823466e38f;
#pragma clang loop unroll(full)
    for (int nf = 0; nf < NT; ++nf)
#pragma clang loop unroll(full)
        for (int r = 0; r < 4; ++r) {
            const int c = (nf0 + nf) * 16 + g * 4 + r;
            if (c < p.ctc_valid) m = fmaxf(m, acc[nf][0][r]);
        }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
    int last = -1;
#pragma clang loop unroll(full)
    for (int nf = 0; nf < NT; ++nf)
#pragma clang loop unroll(full)
        for (int r = 0; r < 4; ++r) {
            const int c = (nf0 + nf) * 16 + g * 4 + r;
            if (c < p.ctc_valid) {
                const float e = __expf(acc[nf][0][r] - m);   // v_exp_f32 path (~2 ulp): 32 of these per lane and tile
                sum += e;
                if (e == 1.0f) last = c;   // columns ascend with (nf, r): the last hit is the largest
            }
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    last = max(last, __shfl_xor(last, 16, 64));
    last = max(last, __shfl_xor(last, 32, 64));
    const long row = m0 + pl_;
    if (g == 0 && row < p.M) {
        float4* dst = reinterpret_cast<float4*>(p.ctc_part) + row * ny + ntile;
        *dst = make_float4(m, sum, __int_as_float(last), 0.f);
    }
}

// Accumulators start from the bias (row g*4+r of fragment nf = channel (nf0+nf)*16 + g*4 + r, i.e. exactly this lane's
// float4 of the bias vector), so the epilogue has no load in front of its stores.  Returns false (and zeroes) when the
// channel count does not allow the float4 path; the epilogue then adds the bias itself.
template <int NT, int PF>
__device__ __forceinline__ bool igemm_init_acc(const IgemmP& p, f32x4 (&acc)[NT][PF], int g, int nf0) {
    const bool fast = p.bias != nullptr && ((p.Cout & 3) == 0) && ((p.y_ld & 3) == 0);
#pragma clang loop unroll(full)
    for (int nf = 0; nf < NT; ++nf) {
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (fast) {
            const int c = (nf0 + nf) * 16 + g * 4;
            const int cc = c < p.gemm_cout ? c : 0;
            const int co = p.convt ? cc % p.Cout : cc;
            b = *reinterpret_cast<const float4*>(p.bias + co);
        }
#pragma clang loop unroll(full)
        for (int pf = 0; pf < PF; ++pf) acc[nf][pf] = (f32x4){b.x, b.y, b.z, b.w};
    }
    return fast;
}

// weight-stationary variant (igemm_ws.inc, instantiated in igemm_ws_1x1.hip / igemm_ws_gen.hip)
void conv_igemm_ws_1x1(hipStream_t s, const IgemmP& p, int ws_nt, int ny, size_t lds);
void conv_igemm_ws_gen(hipStream_t s, const IgemmP& p, int ws_nt, int ny, size_t lds);
// weight-stationary bf16x6 variant for 1x1 / Linear layers (igemm_ws_x6.hip); p.KC = ceil(K/32), p.w in x6 fragment order
void conv_igemm_ws_x6(hipStream_t s, const IgemmP& p, int ws_nt, int ny, size_t lds);
// output-stationary bf16x6 variant for long-K 1x1 layers and k x k convolutions (igemm_os_x6.hip); same weight format
int igemm_os_x6_tile(int nfrag);
void conv_igemm_os_x6(hipStream_t s, const IgemmP& p, int nfrag, bool is1x1);
// 3x3 / stride 1 / pad 1 variant with in-register horizontal tap reuse (igemm_ws3.hip); nfrag = ceil(cout / 16) <= 2
bool conv_igemm_ws3_eligible(const IgemmP& p, int nfrag);
void conv_igemm_ws3(hipStream_t s, const IgemmP& p, int nfrag);
// igemm_rs3_x6.hip: 3x3 same convolution, Cin 32 / 64, <= 16 output channels, bf16x6, row-streaming (weights IGEMM_W_X6)
bool conv3x3_n16_x6_eligible(long M, int Cin, int Cout, long img_px, int y_ld);
std::vector<int> conv3x3_n16_x6_slices(long M, int Cin, int Cout, long img_px, int y_ld);
// igemm_lk_x6.hip: k x k same convolution from an LDS-staged halo tile (k = 9), bf16x6, weights IGEMM_W_X6
void conv_lk_x6(hipStream_t s, const IgemmP& p, int n_images);
void conv3x3_n16_x6(hipStream_t s, const IgemmP& p, int n_images);

}  // namespace k
}  // namespace oar
