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
            const int tap = (int)__umulhi((unsigned)k, p.cin_magic), ci = k - tap * p.Cin;   // (Cin >= 4 on these kernels: the magic number exists)
            const int tap_h = igemm_tap_h(p, tap), tap_w = tap - tap_h * p.kw;
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

// Latency variant for launches too small to hide memory latency behind other waves (SE / SVTR projections, the
// coarse FPN levels: a few hundred wave tiles).  The per-tile kernel above keeps one chunk in flight, so such a launch
// costs ~(KC + 1) load round trips (8 us at K = 64, 15 us at K = 256, whatever M is); here D = 4 chunks are in flight
// per wave (a register ring; D * (NT + 1) float4s), the arithmetic and the epilogue are the same.
// SE (1x1 only): p.se = squeeze-excite gate [image][K], multiplied into the pixel fragment before it is used -- the same v_mul_f32 the
// separate Mul launch did (round 5: the detector's two gated pointwise convs are too small for the bf16x6 kernel that folds the gate)
template <int NT, bool IS1X1, bool SE = false>
__global__ __launch_bounds__(256, 2) void conv_igemm_small_kernel(IgemmP p) {
    constexpr int D = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pl_ = lane & 15, g = lane >> 4;
    const long b = blockIdx.x;
    const int ntile = (int)(b % p.ny);
    const long m0 = ((b / p.ny) * 4 + wave) * 16;
    const int nf0 = ntile * NT;
    if (m0 >= p.M) return;
    long pix_base;
    int ih0 = 0, iw0 = 0;
    {
        const long m = min(m0 + pl_, p.M - 1);   // rows past M: clamped loads, never stored
        if (IS1X1) {
            pix_base = m * (long)p.Cin;
        } else {
            const long hw = (long)p.Ho * p.Wo;
            const long n = m / hw, r = m - n * hw;
            const int oh = (int)(r / p.Wo), ow = (int)(r - (long)oh * p.Wo);
            pix_base = n * (long)p.H * p.W * p.Cin;
            ih0 = oh * p.sh - p.pt; iw0 = ow * p.sw - p.pl;
        }
    }
    f32x4 acc[NT][1];
    const bool bias_in_acc = igemm_init_acc<NT, 1>(p, acc, g, nf0);
    const float4* wf = reinterpret_cast<const float4*>(p.w) + ((long)nf0 * p.KC) * 64 + lane;

    float4 xs[D], ws[D][NT], gs[SE ? D : 1];
    bool ok[D];
    const float* se_row = SE ? p.se + (min(m0 + pl_, p.M - 1) / p.se_hw) * (long)p.K : nullptr;
    auto request = [&](int kc_, int d) {
        const int kc = min(kc_, p.KC - 1);   // past the end: a harmless re-read keeps every load unconditional
        const int k = min(kc * 16 + 4 * g, p.K - 4);
        if (IS1X1) {
            xs[d] = *reinterpret_cast<const float4*>(p.x + pix_base + k);
            if (SE) gs[d] = *reinterpret_cast<const float4*>(se_row + k);
            ok[d] = true;
        } else {
            const int tap = (int)__umulhi((unsigned)k, p.cin_magic), ci = k - tap * p.Cin;   // (Cin >= 4 on these kernels: the magic number exists)
            const int tap_h = igemm_tap_h(p, tap), tap_w = tap - tap_h * p.kw;
            const int ih = ih0 + tap_h * p.dh, iw = iw0 + tap_w * p.dw;
            ok[d] = ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
            const int ihc = min(max(ih, 0), p.H - 1), iwc = min(max(iw, 0), p.W - 1);
            xs[d] = *reinterpret_cast<const float4*>(p.x + pix_base + ((long)ihc * p.W + iwc) * p.Cin + ci);
        }
#pragma clang loop unroll(full)
        for (int nf = 0; nf < NT; ++nf) ws[d][nf] = wf[((long)nf * p.KC + kc) * 64];
    };
    auto comp = [](const float4& v, int j) { return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w; };
    auto consume = [&](int d) {
        float4 v = xs[d];
        if (!IS1X1) { v.x = ok[d] ? v.x : 0.f; v.y = ok[d] ? v.y : 0.f; v.z = ok[d] ? v.z : 0.f; v.w = ok[d] ? v.w : 0.f; }
        if (SE) { const float4 gq = gs[d]; v.x *= gq.x; v.y *= gq.y; v.z *= gq.z; v.w *= gq.w; }
#pragma clang loop unroll(full)
        for (int j = 0; j < 4; ++j)
#pragma clang loop unroll(full)
            for (int nf = 0; nf < NT; ++nf)
                acc[nf][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(comp(ws[d][nf], j), comp(v, j), acc[nf][0], 0, 0, 0);
    };
#pragma clang loop unroll(full)
    for (int d = 0; d < D; ++d) request(d, d);
    int kc = 0;
    for (; kc + D <= p.KC; kc += D) {
#pragma clang loop unroll(full)
        for (int d = 0; d < D; ++d) {
            consume(d);
            request(kc + d + D, d);
        }
    }
    const int rem = p.KC - kc;   // stages 0 .. rem-1 hold the last chunks
#pragma clang loop unroll(full)
    for (int d = 0; d < D - 1; ++d)
        if (d < rem) consume(d);
    igemm_epilogue<NT, 1>(p, acc, m0, pl_, g, nf0, !bias_in_acc);
}

// (The bf16x6 arithmetic -- exact 3-way bf16 split of both operands, six MFMAs per product -- lives in igemm_ws_x6.hip.)

// cout fragments per LDS-resident tile of the x6 kernel for this K (0: does not fit)
static int ws_x6_tile(int K, int nfrag) {
    const int KC = (K + 31) / 32;
    const int cand[3] = {8, 6, 4};
    int best = 1 << 30, nt = 0;
    for (int t : cand) {
        if ((size_t)t * KC * 3072 > 150 * 1024) continue;
        int padded = (nfrag + t - 1) / t * t;
        if (padded < best) { best = padded; nt = t; }
    }
    return nt;
}

int ctc_tiles(int n_padded) { return ((n_padded + 15) / 16 + 7) / 8; }
bool ctc_partials_supported(int K) { return K % 4 == 0 && K >= 32 && (size_t)8 * ((K + 15) / 16) * 1024 <= 150 * 1024; }
// the same epilogue on the bf16x6 weight-stationary kernel: its 8-fragment tile must fit LDS
bool ctc_partials_supported_x6(int K) { return K % 8 == 0 && K >= 32 && (size_t)8 * ((K + 31) / 32) * 3072 <= 150 * 1024; }

// OAR_IGEMM_OS: 1 (default) = output-stationary bf16x6 kernel where the weights do not fit LDS (long-K 1x1, k x k), 0 = off,
// 2 = also wherever the weight-stationary x6 kernel would run (A/B)
static int os_mode() { static const int m = [] { const char* e = getenv("OAR_IGEMM_OS"); return e ? atoi(e) : 1; }(); return m; }
// layers the output-stationary x6 kernel takes: every lane's 8-float group inside one tap and all-valid or all-padding
// (Cin % 8), wide and long enough to be matrix-pipe work, enough 256-pixel tiles to fill the chip, float4 epilogue
static bool os_x6_eligible(long M, int K, int N, int Cin, bool grouped = false) {
    static const int min_k = [] { const char* e = getenv("OAR_IGEMM_OS_MINK"); return e ? atoi(e) : 256; }();
    // enough (256-pixel, 128-cout) tiles for the 512 workgroup slots of the chip: many pixels, or fewer pixels under a wide layer
    // (round 6: a group of a grouped convolution has no other matrix-pipe kernel -- any launch of >= 4096 pixels beats the direct kernel; a very long K
    // (the 9x9 LK-PAN convolutions, K = 20736) pays for half-filled CUs as well: 72 -> ~110 TFLOP/s at 200 tiles)
    const long tiles = ((M + 255) / 256) * ((N + 127) / 128);
    const bool fills = M >= 65536 || (M >= 8192 && tiles >= 256) || (grouped && M >= 4096) || (K >= 2048 && M >= 8192 && tiles >= 128);
    return Cin % 8 == 0 && K >= min_k && N >= (grouped ? 32 : 48) && (N & 3) == 0 && fills && K < 65536;   // (48: PP-HGNetV2's 48 -> 48 3x3 stage, three of a 4-fragment tile -- 98 TFLOP/s on the f32 kernel)
}

std::vector<int> conv3x3_n16_slices(long M, int Cin, int Cout, long img_px, int y_ld) { return conv3x3_n16_x6_slices(M, Cin, Cout, img_px, y_ld); }
// (the layer must also come out of igemm_weight_format as a bf16x6 layer -- OAR_IGEMM_X6=0 turns every layer into an f32 one --: the planner lays the weights out by that answer)
bool conv_msrc_ok(long M, int K, int N) { return os_mode() != 0 && (K & 7) == 0 && os_x6_eligible(M, K, N, K) && igemm_weight_format(M, K, N, true, K) == IGEMM_W_X6; }
bool conv_grouped_x6_ok(long M, int K, int N, int Cin) { return os_mode() != 0 && os_x6_eligible(M, K, N, Cin, true); }

int igemm_weight_format(long M, int K, int N, bool is1x1, int Cin, long same3x3_px, bool lk_ok) {
    // OAR_IGEMM_X6: 1 (default) = bf16x6 kernels on the wide layers, 0 = f32 MFMA everywhere
    static const int mode = [] { const char* e = getenv("OAR_IGEMM_X6"); return e ? atoi(e) : 1; }();
    if (!mode) return IGEMM_W_K16;
    // (decided before the output's leading dimension is final -- a Concat may place the layer in a wider buffer: the 2^31-byte bound of the kernel's
    // output descriptor is checked here for up to four equal branches; launch time re-checks with the real y_ld and fails loudly, never falls through)
    if (!is1x1 && same3x3_px > 0 && K == 9 * Cin && conv3x3_n16_x6_eligible(M, Cin, N, same3x3_px, 4 * N)) return IGEMM_W_X6;
    if (!is1x1 && lk_ok) return IGEMM_W_X6;   // large-kernel same convolution: LDS-tiled bf16x6 kernel, whatever the pixel count
    if (!is1x1) return (os_mode() && Cin > 0 && os_x6_eligible(M, K, N, Cin)) ? IGEMM_W_X6 : IGEMM_W_K16;
    // every lane's 8-float group must be all-valid or all-padding (K % 8); wide enough to be matrix-pipe bound
    // (N >= 96, K >= 96); enough (16-pixel tile, cout tile) pairs to fill the 4096 resident waves; float4 epilogue
    const int nfrag = (N + 15) / 16;
    const long passes = ((M + 15) / 16) * ((nfrag + 7) / 8);
    if (K % 8 == 0 && K >= 96 && N >= 96 && (N & 3) == 0 && passes >= 4096 && ws_x6_tile(K, nfrag) > 0) return IGEMM_W_X6;
    // (a K = 64 CTC head was measured on this kernel too: 157 us against 141 us on the f32 kernel -- two chunks of K do not
    // amortise the per-tile weight staging, it stays on f32)
    if (os_mode() && os_x6_eligible(M, K, N, K)) return IGEMM_W_X6;   // long K: the weights do not fit LDS
    return IGEMM_W_K16;
}

int ws_x6_se_rows(long M, int ny, int hw);   // igemm_ws_x6.hip
// the f32 kernel choice of conv_igemm for a plain 1x1 layer (no residual, no CTC partials): does it land on the latency variant?
static bool f32_1x1_goes_small(long M, int K, int N) {
    static const int ws_mode = [] { const char* e = getenv("OAR_IGEMM_WS"); return e ? atoi(e) : -1; }();
    static const int ws_min_n = [] { const char* e = getenv("OAR_IGEMM_WS_MIN_N"); return e ? atoi(e) : 96; }();
    static const long small_max = [] { const char* e = getenv("OAR_IGEMM_SMALL"); return e ? atol(e) : 4096L; }();
    const int nfrag = (N + 15) / 16, KC = (K + 15) / 16;
    int NT = 1, best = 1 << 30;
    for (int t = 4; t >= 1; --t) { int padded = (nfrag + t - 1) / t * t; if (padded < best) { best = padded; NT = t; } }
    const long ny = (nfrag + NT - 1) / NT;
    const long wave_tile_passes = ((M + 15) / 16) * ((nfrag + 7) / 8);
    const bool ws_auto = (N >= ws_min_n && wave_tile_passes >= 4096) || (K >= 512 && M >= 262144);
    const bool ws = (N & 3) == 0 && KC >= 2 && ws_mode != 0 && (ws_mode == 1 || ws_auto);
    return !ws && ((M + 15) / 16) * ny <= small_max;
}
bool conv_igemm_se_ok(long M, int K, int N, int hw) {
    static const bool on = [] { const char* e = getenv("OAR_FUSE_SE_SCALE"); return !e || atoi(e) != 0; }();
    if (!on || hw <= 0) return false;
    if (igemm_weight_format(M, K, N, true) == IGEMM_W_K16) {   // round 5: the latency variant of the f32 kernel multiplies the gate in as well
        const char* e = getenv("OAR_FUSE_SE_SCALE_SMALL");   // (plan-time question: read per call, a test flips it within one process)
        const bool small_se = !(e && e[0] == '0');
        return small_se && (K & 3) == 0 && (N & 3) == 0 && f32_1x1_goes_small(M, K, N);
    }
    if (igemm_weight_format(M, K, N, true) != IGEMM_W_X6) return false;
    const int nfrag = (N + 15) / 16, nt = ws_x6_tile(K, nfrag);
    if (nt == 0 || os_mode() == 2) return false;   // (the output-stationary kernel has no gate path)
    const int ny = (nfrag + nt - 1) / nt;
    const size_t lds = (size_t)nt * ((K + 31) / 32) * 3072 + (size_t)nt * 64 + 16 + (size_t)ws_x6_se_rows(M, ny, hw) * K * 4;
    return lds <= 160 * 1024;
}

void conv_igemm(hipStream_t s, const ConvP& c) {
    IgemmP p;
    p.x = c.x; p.w = c.w; p.bias = c.bias; p.res = c.residual; p.y = c.y;
    p.H = c.H; p.W = c.W; p.Cin = c.Cin; p.kh = c.kh; p.kw = c.kw; p.sh = c.sh; p.sw = c.sw;
    p.pt = c.pt; p.pl = c.pl; p.dh = c.dh; p.dw = c.dw; p.y_ld = c.y_ld;
    p.act = c.act.kind; p.alpha = c.act.alpha; p.beta = c.act.beta;
    p.convt = c.convt2x2; p.Cout = c.Cout;
    p.ctc_part = c.ctc_part; p.ctc_valid = c.ctc_valid;
    p.se = c.se; p.se_hw = c.Ho * c.Wo;
    p.x_ld = c.x_ld > 0 ? c.x_ld : c.Cin; p.accum = c.accum;
    p.n_msrc = c.n_msrc;
    for (int i = 0; i < 8; ++i) { p.msrc[i] = i < c.n_msrc ? c.msrc[i] : nullptr; p.msrc_c[i] = i < c.n_msrc ? c.msrc_c[i] : 0; }
    p.res_up = c.residual ? c.res_up : 0;
    if (c.convt2x2) {
        p.Ho = c.H; p.Wo = c.W;  // GEMM columns are INPUT pixels
        p.M = (long)c.N * c.H * c.W; p.K = c.Cin; p.gemm_cout = 4 * c.Cout;
    } else {
        p.Ho = c.Ho; p.Wo = c.Wo;
        p.M = (long)c.N * c.Ho * c.Wo; p.K = c.kh * c.kw * c.Cin; p.gemm_cout = c.Cout;
    }
    const bool x6 = c.w_fmt == IGEMM_W_X6;
    const bool grouped = p.x_ld != c.Cin;   // one group of a grouped convolution: output-stationary bf16x6 kernel only (the one that reads x with x_ld)
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
    int PF = pf_max;
    while (PF > 1 && ((p.M + 4L * PF * 16 - 1) / (4L * PF * 16)) * ny < 1024) PF >>= 1;
    const long mx = (p.M + 4L * PF * 16 - 1) / (4L * PF * 16);
    p.ny = (int)ny;
    OAR_CHECK(p.K < 65536, OAR_UNSUPPORTED_OP, "conv_igemm: K = kh*kw*Cin must be < 65536");
    p.cin_magic = (unsigned)((1ull << 32) / (unsigned)c.Cin + 1);
    p.kw_magic = (unsigned)((1ull << 32) / (unsigned)c.kw + 1);   // (kw == 1: wraps to 1, umulhi(tap, 1) == 0, and kw_one adds the tap itself)
    p.kw_one = c.kw == 1 ? 1u : 0u;
    p.mx_per_xcd = (mx + 7) / 8;
    dim3 grid((unsigned)(p.mx_per_xcd * 8 * ny));
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
    const bool res_up = p.res_up > 1;   // the upsampled residual lives in the per-tile kernels' epilogue only
    OAR_CHECK(!res_up || (!x6 && !c.convt2x2 && p.M < (1L << 31) && c.Ho % p.res_up == 0 && c.Wo % p.res_up == 0), OAR_INTERNAL, "conv_igemm: upsampled residual on an ineligible layer");
    if (!x6 && !res_up && vec_ok && p.KC >= 2 && ws_mode != 0 && (ws_mode == 1 || ws_auto)) {
        static const int max_nt = [] { const char* e = getenv("OAR_IGEMM_WS_MAXNT"); return e ? atoi(e) : 8; }();
        const int cand[6] = {8, 6, 4, 3, 2, 1};
        int best = 1 << 30;
        for (int t : cand) {
            if (t > max_nt || (size_t)t * p.KC * 1024 > 150 * 1024) continue;
            int padded = (nfrag + t - 1) / t * t;
            if (padded < best) { best = padded; ws_nt = t; }
        }
    }
    if (c.ctc_part && x6 && is1x1 && !c.residual && !c.se && c.act.kind == ACT_NONE && ctc_head_x6_supported(p.M, p.K, p.gemm_cout)) {
        ctc_head_x6(s, c.x, c.w, c.bias, c.ctc_part, p.M, p.K, p.gemm_cout, c.ctc_valid);   // short K: output-stationary, the weights streamed once per 384 rows
        return;
    }
    if (c.ctc_part) {   // the partial-softmax epilogue lives in the weight-stationary kernels (f32 and bf16x6), 8 fragments per tile
        OAR_CHECK(vec_ok && is1x1 && (x6 ? ctc_partials_supported_x6(p.K) : ctc_partials_supported(p.K)), OAR_INTERNAL, "conv_igemm: CTC partials on an ineligible layer");
        ws_nt = 8;
    }
    const bool ws3 = !x6 && !res_up && !c.ctc_part && conv_igemm_ws3_eligible(p, nfrag);
    const bool ws = ws_nt > 0 || ws3;
    // launches of at most `small_max` wave tiles are latency-bound: the deep-prefetch variant (OAR_IGEMM_SMALL=0 disables)
    static const long small_max = [] { const char* e = getenv("OAR_IGEMM_SMALL"); return e ? atol(e) : 4096L; }();
    const bool small = !x6 && !ws && ((p.M + 15) / 16) * ny <= small_max;
    double flops = 2.0 * (double)p.M * p.K * p.gemm_cout;
    double bytes = 4.0 * ((double)c.N * c.H * c.W * c.Cin + (double)p.M * p.gemm_cout + (double)p.K * p.gemm_cout);
    char pname[96];
    // wide 1x1 layers whose weight-stationary tile is down to 4 cout fragments (K >= 320: 64 couts x K x 96 B of LDS) make >= 8 passes over their pixels -- each pass
    // re-reading and re-splitting them; there the output-stationary kernel's 128-cout tiles win (SVTRv2's 384 -> 1152 projection: 130 -> 199 TFLOP/s).  OAR_IGEMM_OS_WIDE=0 off
    static const bool os_wide = [] { const char* e = getenv("OAR_IGEMM_OS_WIDE"); return !e || atoi(e) != 0; }();
    // (not behind an expensive activation: the output-stationary kernel's epilogue runs after its K loop with nothing to hide it -- the GELU layers 384 -> 1536 and
    // 256 -> 1024 measured 132 / 108 TFLOP/s there against 152 / 169 on the weight-stationary kernel, whose other waves multiply meanwhile)
    const bool cheap_act = c.act.kind == ACT_NONE || c.act.kind == ACT_RELU;
    const bool prefer_os = os_wide && os_mode() != 0 && x6 && is1x1 && !c.convt2x2 && !c.se && !c.ctc_part && cheap_act && p.K >= 320 && p.gemm_cout >= 512 && ws_x6_tile(p.K, nfrag) > 0 && ws_x6_tile(p.K, nfrag) <= 4 &&
                           os_x6_eligible(p.M, p.K, p.gemm_cout, c.Cin);
    const bool x6_os = x6 && !c.ctc_part && (!is1x1 || ws_x6_tile(p.K, nfrag) == 0 || prefer_os || c.n_msrc > 0 || (os_mode() == 2 && os_x6_eligible(p.M, p.K, p.gemm_cout, c.Cin)));
    const bool rs3_cls = x6 && !c.convt2x2 && c.kh == 3 && c.kw == 3 && c.sh == 1 && c.sw == 1 && c.pt == 1 && c.pl == 1 && c.dh == 1 && c.dw == 1 && c.Ho == c.H && c.Wo == c.W && !c.residual && !c.se &&
                         !c.ctc_part && conv3x3_n16_x6_eligible(p.M, c.Cin, c.Cout, (long)c.H * c.W, c.y_ld);
    const bool lk = x6 && !c.ctc_part && !c.se && !c.convt2x2 && !is1x1 &&
                    conv_lk_x6_eligible(c.kh, c.kw, c.sh, c.sw, c.pt, c.pl, c.dh, c.dw, c.H, c.W, c.Ho, c.Wo, c.Cin, c.Cout, c.y_ld, p.M);
    const char* cls = lk ? "conv_lk_x6" : rs3_cls ? "conv_rs3_x6" : x6_os ? "conv_igemm_os_x6" : x6 ? "conv_igemm_ws_x6" : ws ? "conv_igemm_ws" : "conv_igemm";   // one profiler class per kernel
    if (Profiler::get().detail) {
        snprintf(pname, sizeof pname, "conv_igemm%s M=%ld K=%d N=%d k%dx%d s%d%s", lk ? "_lk_x6" : rs3_cls ? "_rs3_x6" : x6_os ? "_os_x6" : x6 ? "_x6" : ws ? "_ws" : "", p.M, p.K, p.gemm_cout, c.kh, c.kw, c.sh, c.convt2x2 ? " convT" : "");
        cls = pname;
    }
    ProfScope ps(s, cls, bytes, flops);
#define LAUNCH2(NTV, PFV)                                                                                          \
    do {                                                                                                          \
        if (is1x1) hipLaunchKernelGGL((conv_igemm_kernel<NTV, PFV, true>), grid, dim3(256), 0, s, p);     \
        else hipLaunchKernelGGL((conv_igemm_kernel<NTV, PFV, false>), grid, dim3(256), 0, s, p);                 \
    } while (0)
#define LAUNCH(NTV)                                  \
    do {                                             \
        if (PF == 4) LAUNCH2(NTV, 4);                \
        else if (PF == 2) LAUNCH2(NTV, 2);           \
        else LAUNCH2(NTV, 1);                        \
    } while (0)
    OAR_CHECK(!c.se || x6 || (small && is1x1 && !c.residual && !c.convt2x2), OAR_INTERNAL, "conv_igemm: gate on a layer neither gated kernel takes (conv_igemm_se_ok should have said no)");
    const bool same3x3 = !c.convt2x2 && c.kh == 3 && c.kw == 3 && c.sh == 1 && c.sw == 1 && c.pt == 1 && c.pl == 1 && c.dh == 1 && c.dw == 1 && c.Ho == c.H && c.Wo == c.W;
    const bool rs3 = x6 && same3x3 && !c.residual && !c.se && !c.ctc_part && conv3x3_n16_x6_eligible(p.M, c.Cin, c.Cout, (long)c.H * c.W, c.y_ld);
    // a layer whose weights were laid out for the row-streaming 3x3 kernel (Cout <= 16: no other bf16x6 kernel takes it) must reach that kernel
    OAR_CHECK(!(x6 && same3x3 && c.Cout <= 16) || rs3, OAR_INTERNAL, "conv_igemm: 3x3 / Cout <= 16 bf16x6 weights but the row-streaming kernel's launch-time conditions do not hold (y_ld / output size / residual / gate changed after planning)");
    OAR_CHECK(!grouped || (x6 && !is1x1 && !c.ctc_part && !c.se && !c.convt2x2 && (rs3 || !(same3x3 && c.Cout <= 16))), OAR_INTERNAL, "conv_igemm: x_ld on a layer that runs on neither kernel that reads x with a stride of its own");
    OAR_CHECK(c.n_msrc == 0 || (x6 && is1x1 && !c.convt2x2 && !c.ctc_part && !c.se && !lk), OAR_INTERNAL, "conv_igemm: multi-source input on a layer that does not run on the output-stationary bf16x6 kernel");
    OAR_CHECK(!c.accum || rs3, OAR_INTERNAL, "conv_igemm: accumulate flag on a layer that does not run on the row-streaming 3x3 kernel");
    if (lk) {
        conv_lk_x6(s, p, c.N);
    } else if (ws3) {
        conv_igemm_ws3(s, p, nfrag);
    } else if (rs3) {
        conv3x3_n16_x6(s, p, c.N);
    } else if (x6) {
        const int nt = c.ctc_part ? 8 : is1x1 ? ws_x6_tile(p.K, nfrag) : 0;
        OAR_CHECK(((p.Cout & 3) == 0) && ((p.y_ld & 3) == 0) && !c.convt2x2, OAR_INTERNAL, "conv_igemm: bf16x6 weights on an ineligible layer");
        const bool os = !c.ctc_part && (nt == 0 || prefer_os || c.n_msrc > 0 || (os_mode() == 2 && os_x6_eligible(p.M, p.K, p.gemm_cout, c.Cin)));
        OAR_CHECK(!c.se || (!os && !c.ctc_part && is1x1), OAR_INTERNAL, "conv_igemm: gate on a layer the weight-stationary x6 kernel does not take");
        if (os) {
            OAR_CHECK(os_x6_eligible(p.M, p.K, p.gemm_cout, c.Cin, grouped), OAR_INTERNAL, "conv_igemm: bf16x6 weights on a layer neither x6 kernel takes");
            conv_igemm_os_x6(s, p, nfrag, is1x1);
        } else {
            conv_igemm_ws_x6(s, p, nt, (nfrag + nt - 1) / nt, (size_t)nt * p.KC * 3072 + (size_t)nt * 64 + 16);
        }
    } else if (ws) {
        const int wny = (nfrag + ws_nt - 1) / ws_nt;
        const size_t lds = (size_t)ws_nt * p.KC * 1024 + (size_t)ws_nt * 64 + 16;
        if (is1x1) conv_igemm_ws_1x1(s, p, ws_nt, wny, lds);
        else conv_igemm_ws_gen(s, p, ws_nt, wny, lds);
    } else if (small) {
        const dim3 sgrid((unsigned)(((p.M + 63) / 64) * ny));
#define LAUNCH_SMALL(NTV)                                                                                          \
    do {                                                                                                          \
        if (is1x1 && p.se) hipLaunchKernelGGL((conv_igemm_small_kernel<NTV, true, true>), sgrid, dim3(256), 0, s, p); \
        else if (is1x1) hipLaunchKernelGGL((conv_igemm_small_kernel<NTV, true>), sgrid, dim3(256), 0, s, p);   \
        else hipLaunchKernelGGL((conv_igemm_small_kernel<NTV, false>), sgrid, dim3(256), 0, s, p);             \
    } while (0)
        if (NT == 4) LAUNCH_SMALL(4);
        else if (NT == 3) LAUNCH_SMALL(3);
        else if (NT == 2) LAUNCH_SMALL(2);
        else LAUNCH_SMALL(1);
#undef LAUNCH_SMALL
    } else if (NT == 4) LAUNCH(4);
    else if (NT == 3) LAUNCH(3);
    else if (NT == 2) LAUNCH(2);
    else LAUNCH(1);
#undef LAUNCH2
#undef LAUNCH
}

}  // namespace k
}  // namespace oar
