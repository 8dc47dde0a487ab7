// igemm.hip -- implicit-GEMM convolution on the f32 / bf16 matrix cores: per-wave-tile kernels + host dispatch.
#include "igemm_dev.h"

namespace oar {
namespace k {

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
    const bool bias_in_acc = igemm_init_acc<NT, PF>(p, acc, g, nf0);

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

    igemm_epilogue<NT, PF>(p, acc, m0, pl_, g, nf0, !bias_in_acc);
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
    const bool bias_in_acc = igemm_init_acc<NT, PF>(p, acc, g, nf0);

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

    igemm_epilogue<NT, PF>(p, acc, m0, pl_, g, nf0, !bias_in_acc);
}

int igemm_weight_format(int K, int Cin, bool is1x1) {
    // OAR_IGEMM_FMT: 0 f32 fragments (default), 1 bf16x6 fragments where eligible
    static const int mode = [] { const char* e = getenv("OAR_IGEMM_FMT"); return e ? atoi(e) : 0; }();
    // x6: every lane's 8-float group must be all-valid or all-padding, and must not straddle two taps
    if (mode == 1 && K % 8 == 0 && (is1x1 || Cin % 8 == 0)) return IGEMM_W_X6;
    return IGEMM_W_K16;
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
    const bool x6 = c.w_fmt == IGEMM_W_X6;
    p.KC = x6 ? (p.K + 31) / 32 : (p.K + 15) / 16;
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
    int PF = x6 ? 2 : pf_max;
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
    // weight-stationary variant (see conv_igemm_ws_kernel): OAR_IGEMM_WS = 0 off, 1 wherever it fits, default: N >= ws_min_n
    static const int ws_mode = [] { const char* e = getenv("OAR_IGEMM_WS"); return e ? atoi(e) : -1; }();
    static const int ws_min_n = [] { const char* e = getenv("OAR_IGEMM_WS_MIN_N"); return e ? atoi(e) : 96; }();
    int ws_nt = 0;
    const bool vec_ok = ((p.Cout & 3) == 0) && ((p.y_ld & 3) == 0);
    // auto rule (per-shape A/B on the bench graphs, profiles/r1): the persistent kernel needs enough wave tiles to fill its
    // 4096 resident waves and pays for staging W once per workgroup, so it wins on wide layers with many pixels and
    // on the long-K 3x3 convs; small-M / tiny-K layers stay on the per-tile kernel
    const long wave_tile_passes = ((p.M + 15) / 16) * ((nfrag + 7) / 8);   // (16-pixel tile, 128-cout tile) pairs
    const bool ws_auto = (p.gemm_cout >= ws_min_n && wave_tile_passes >= 4096) || (p.K >= 512 && p.M >= 262144);
    if (!x6 && vec_ok && p.KC >= 2 && ws_mode != 0 && (ws_mode == 1 || ws_auto)) {
        static const int max_nt = [] { const char* e = getenv("OAR_IGEMM_WS_MAXNT"); return e ? atoi(e) : 8; }();
        const int cand[6] = {8, 6, 4, 3, 2, 1};
        int best = 1 << 30;
        for (int t : cand) {
            if (t > max_nt || (size_t)t * p.KC * 1024 > 150 * 1024) continue;
            int padded = (nfrag + t - 1) / t * t;
            if (padded < best) { best = padded; ws_nt = t; }
        }
    }
    const bool ws = ws_nt > 0;
    double flops = 2.0 * (double)p.M * p.K * p.gemm_cout;
    double bytes = 4.0 * ((double)c.N * c.H * c.W * c.Cin + (double)p.M * p.gemm_cout + (double)p.K * p.gemm_cout);
    char pname[96];
    const char* cls = "conv_igemm";
    if (Profiler::get().detail) {
        snprintf(pname, sizeof pname, "conv_igemm%s M=%ld K=%d N=%d k%dx%d s%d%s", x6 ? "_x6" : ws ? "_ws" : "", p.M, p.K, p.gemm_cout, c.kh, c.kw, c.sh, c.convt2x2 ? " convT" : "");
        cls = pname;
    }
    ProfScope ps(s, cls, bytes, flops);
#define LAUNCH2(NTV, PFV)                                                                                          \
    do {                                                                                                          \
        if (x6) {                                                                                                 \
            if (is1x1) hipLaunchKernelGGL((conv_igemm_x6_kernel<NTV, PFV, true>), grid, dim3(256), 0, s, p);     \
            else hipLaunchKernelGGL((conv_igemm_x6_kernel<NTV, PFV, false>), grid, dim3(256), 0, s, p);          \
        } else if (is1x1) hipLaunchKernelGGL((conv_igemm_kernel<NTV, PFV, true>), grid, dim3(256), 0, s, p);     \
        else hipLaunchKernelGGL((conv_igemm_kernel<NTV, PFV, false>), grid, dim3(256), 0, s, p);                 \
    } while (0)
#define LAUNCH(NTV)                                  \
    do {                                             \
        if (PF == 4) LAUNCH2(NTV, 4);                \
        else if (PF == 2) LAUNCH2(NTV, 2);           \
        else LAUNCH2(NTV, 1);                        \
    } while (0)
    if (ws) {
        const int wny = (nfrag + ws_nt - 1) / ws_nt;
        const size_t lds = (size_t)ws_nt * p.KC * 1024 + (size_t)ws_nt * 64 + 16;
        if (is1x1) conv_igemm_ws_1x1(s, p, ws_nt, wny, lds);
        else conv_igemm_ws_gen(s, p, ws_nt, wny, lds);
    } else if (NT == 4) LAUNCH(4);
    else if (NT == 3) LAUNCH(3);
    else if (NT == 2) LAUNCH(2);
    else LAUNCH(1);
#undef LAUNCH2
#undef LAUNCH
}

}  // namespace k
}  // namespace oar
