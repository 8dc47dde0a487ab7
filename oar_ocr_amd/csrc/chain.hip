// chain.hip -- a run of sample-local operators as ONE launch (round 3, VERDICT r2 #3: the sub-20 us launch tail).
//
// The SVTR neck of a recognizer (EncoderWithSVTR: 1x3 conv -> 1x1 conv -> 2 x [LN -> QKV -> attention -> proj + residual -> LN ->
// FC1 -> FC2 + residual] -> LN -> 1x1 conv -> concat -> 1x3 conv -> 1x1 conv) touches, per text line, T <= a few dozen tokens of
// <= 512 channels: every operator is a [T x K] x [K x N] product, a row normalisation, a T x T attention or a row copy that only
// reads rows of its OWN sample.  As separate launches that is 22 kernels of 5-30 us each whose grids cannot fill 256 CUs; here ONE
// workgroup (16 waves) owns a sample and walks the operator table, a workgroup barrier between operators.  The planner (engine.cc,
// Planner::fuse_chains) records the operators, keeps every tensor of the run allocated until the run ends (no two of them alias,
// so samples may run at different paces), places the run's internal tensors in LDS, packs its constants into one blob and
// replaces the run by one `chain_run` step.  DESIGN.md 4.12 has the per-operator timings and what was tried.
//
// Arithmetic: products on v_mfma_f32_16x16x4_f32 (exact f32 multiply-add, weights as the A operand, tokens as B: a lane ends up
// with 4 consecutive output channels of one token = one float4 store); split-K partials, when a product has too few tiles for 16
// waves, are reduced through LDS in a fixed order (deterministic).  Attention for T <= 64 runs on the matrix pipe too.  Summation
// orders differ from the operator-by-operator kernels: equal to them to f32 rounding (<= 5e-5 in the tests), not bit for bit.
#include <hip/hip_ext.h>
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "common.h"
#include "kernels_dev.h"

namespace oar {
namespace k {

namespace {
constexpr int kChainThreads = 1024;
constexpr int kWaves = kChainThreads / 64;

// LDS and HBM are addressed through pointers of their own address spaces, by hand: a `cond ? lds_ptr : hbm_ptr` select would make
// every access a FLAT instruction (LDS at L2-hit latency, both wait counters) -- the first version of this kernel ran 3x slower for it.
typedef __attribute__((address_space(3))) float ch_lf;
typedef __attribute__((address_space(3))) f32x4 ch_lf4;
typedef __attribute__((address_space(3))) int ch_li;
typedef __attribute__((address_space(1))) const float ch_gf;
typedef __attribute__((address_space(1))) const f32x4 ch_gf4;
typedef __attribute__((address_space(1))) float ch_gfw;
typedef __attribute__((address_space(1))) f32x4 ch_gf4w;
#define CH_LDS(T, byte) (*reinterpret_cast<T*>((__attribute__((address_space(3))) char*)nullptr + (byte)))
__device__ __forceinline__ float4 ch_f4(f32x4 v) { return make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ f32x4 ch_v4(float4 v) { f32x4 r = {v.x, v.y, v.z, v.w}; return r; }

// one operand of an operator as the workgroup sees it: rows of `ld` floats, in LDS (byte address `l`) or in HBM (the sample's first row)
struct ChView { ch_gf* g; unsigned l; int ld; bool lds; };
__device__ __forceinline__ ChView ch_view(const ChainRef& r, int ld, long row0, const char* arena, const char* input, unsigned lbase) {
    ChView v;
    v.ld = ld; v.lds = r.kind == 3; v.l = lbase + (unsigned)r.v;
    typedef __attribute__((address_space(1))) const char gch;
    gch* base = r.kind == 1 ? (gch*)arena + r.v : r.kind == 2 ? (gch*)input + r.v : (gch*)(unsigned long long)r.v;
    v.g = reinterpret_cast<ch_gf*>(base) + row0 * ld;
    return v;
}
__device__ __forceinline__ float4 ch_ld4(const ChView& v, int idx) {
    if (v.lds) return ch_f4(CH_LDS(ch_lf4, v.l + 4u * (unsigned)idx));
    return ch_f4(*reinterpret_cast<ch_gf4*>(v.g + idx));
}
__device__ __forceinline__ float ch_ld1(const ChView& v, int idx) {
    if (v.lds) return CH_LDS(ch_lf, v.l + 4u * (unsigned)idx);
    return v.g[idx];
}
__device__ __forceinline__ void ch_st4(const ChView& v, int idx, float4 x) {
    if (v.lds) CH_LDS(ch_lf4, v.l + 4u * (unsigned)idx) = ch_v4(x);
    else *reinterpret_cast<ch_gf4w*>(const_cast<ch_gfw*>(v.g) + idx) = ch_v4(x);
}
__device__ __forceinline__ void ch_st1(const ChView& v, int idx, float x) {
    if (v.lds) CH_LDS(ch_lf, v.l + 4u * (unsigned)idx) = x;
    else const_cast<ch_gfw*>(v.g)[idx] = x;
}
template <int CTRL>
__device__ __forceinline__ float ch_dpp(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, false)); }
// all-reduce over the 16 lanes of a DPP row (rotations 8, 4, 2, 1): every lane ends up with the same sum / maximum
__device__ __forceinline__ float ch_row16_sum(float v) {
    v += ch_dpp<0x128>(v); v += ch_dpp<0x124>(v); v += ch_dpp<0x122>(v); v += ch_dpp<0x121>(v);
    return v;
}
// all-reduce over a quad of lanes (quad_perm [1,0,3,2] then [2,3,0,1])
__device__ __forceinline__ float ch_quad_sum(float v) { v += ch_dpp<0xB1>(v); v += ch_dpp<0x4E>(v); return v; }
__device__ __forceinline__ float ch_quad_max(float v) { v = fmaxf(v, ch_dpp<0xB1>(v)); v = fmaxf(v, ch_dpp<0x4E>(v)); return v; }

// the activations a chained product may carry (the planner leaves operators with any other one unfused): apply_act's formulas, the
// switch outside the four elements -- every extra case is inlined into each epilogue of a kernel that has to stay cache-sized
__device__ __forceinline__ f32x4 ch_act4(f32x4 v, int kind, float alpha, float beta) {
    switch (kind) {
        case ACT_RELU:
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = v[i] > 0.f ? v[i] : 0.f;
            break;
        case ACT_HSWISH:
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = v[i] * fminf(fmaxf(v[i] * (1.0f / 6.0f) + 0.5f, 0.f), 1.f);
            break;
        case ACT_HSIGMOID:
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = fminf(fmaxf(v[i] * alpha + beta, 0.f), 1.f);
            break;
        case ACT_SIGMOID:
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = 1.0f / (1.0f + expf(-v[i]));
            break;
        case ACT_SWISH:
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = v[i] * (1.0f / (1.0f + expf(-v[i])));
            break;
        default: break;
    }
    return v;
}

// out[t][n] = act(bias[n] + sum_{tap, c} in[t + tap - pad][c] * w[n][tap * cin + c]) (+ res[t][n]),  t in [0, T)
struct ChEpi { const ChainOpD& op; const ChView& out; const ChView& res; bool has_res; int T; unsigned lbase; int r, q, MT; };
__device__ __forceinline__ void ch_epilogue(const ChEpi& e, f32x4 acc, int nt, int mtile) {
    const int t = mtile * 16 + e.r, n0 = nt * 16 + 4 * e.q;
    if (mtile >= e.MT || t >= e.T) return;
    if (e.op.bias_l >= 0) { const f32x4 b = CH_LDS(ch_lf4, e.lbase + 4u * (unsigned)(e.op.bias_l + n0)); acc += b; }
    acc = ch_act4(acc, e.op.act, e.op.alpha, e.op.beta);
    if (e.has_res) { const float4 v = ch_ld4(e.res, t * e.res.ld + n0); acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w; }
    ch_st4(e.out, t * e.out.ld + n0, make_float4(acc[0], acc[1], acc[2], acc[3]));
}

// General path (tokens in HBM: only when the planner could not stage them in LDS): one work item = one 16 x 16 output tile over one
// K slice, one 16-wide K step at a time.  Compact on purpose -- it is the fallback, and the kernel's code has to stay cache-sized.
__device__ __forceinline__ void ch_gemm_slow(const ChainOpD& op, const ChView& in, const ChView& out, const ChView& res, bool has_res, int T, unsigned lbase, int wave, int lane) {
    const int MT = (T + 15) >> 4, NT = op.N >> 4, KB = op.K >> 4, KS = op.ksplit;
    const int kper = (KB + KS - 1) / KS, tiles = NT * MT, items = tiles * KS;
    const int r = lane & 15, q = lane >> 4;
    const ChEpi epi{op, out, res, has_res, T, lbase, r, q, MT};
    for (int it = wave; it < items; it += kWaves) {
        const int tile = it / KS, ks = it - tile * KS, nt = tile / MT, mt = tile - nt * MT;
        const int t = mt * 16 + r;
        ch_gf* wrow = (ch_gf*)reinterpret_cast<unsigned long long>(op.w) + (long)(nt * 16 + r) * op.K + 4 * q;
        const int kb0 = ks * kper, kb1 = min(KB, kb0 + kper);
        int tap = (kb0 * 16) / op.cin, c0 = kb0 * 16 - tap * op.cin;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma nounroll
        for (int kb = kb0; kb < kb1; ++kb) {
            const float4 a = ch_f4(*reinterpret_cast<ch_gf4*>(wrow + kb * 16));
            const int row = t + tap - op.pad;
            float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row >= 0 && row < T) b = ch_ld4(in, row * in.ld + c0 + 4 * q);
            c0 += 16;
            if (c0 == op.cin) { c0 = 0; ++tap; }
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc, 0, 0, 0);
        }
        if (KS == 1) ch_epilogue(epi, acc, nt, mt);
        else CH_LDS(ch_lf4, lbase + 16u * (unsigned)(it * 64 + lane)) = acc;
    }
    if (KS > 1) {
        __syncthreads();
        for (int tile = wave; tile < tiles; tile += kWaves) {
            f32x4 acc = CH_LDS(ch_lf4, lbase + 16u * (unsigned)(tile * KS * 64 + lane));
            for (int ks = 1; ks < KS; ++ks) acc += CH_LDS(ch_lf4, lbase + 16u * (unsigned)((tile * KS + ks) * 64 + lane));
            ch_epilogue(epi, acc, tile / MT, tile % MT);
        }
    }
}

// Fast path: tokens in LDS.  One work item = 16 output channels x MB (<= 3) token tiles (the weights of a K step are loaded once for
// all of them) over one K slice; items = channel tiles x token groups x K slices, chosen by the planner to occupy the 16 waves in one
// round.  (Requesting the next operator's first weights one operator ahead was tried: the 32 registers it holds across the
// operator push this 1024-thread kernel over its 128 and the spills cost more than the L2 round trip they hide.)
// The K loop is BRANCH-FREE (end of round 5): MB is a template parameter, a K step past the slice loads a clamped (valid) weight address and
// is zeroed by a select, a token row outside the sample reads row 0 and is zeroed by a select.  With `if (m < op.mb)` / `if (g + u < kb1)`
// as run-time branches around loads and MFMAs the compiler split the unrolled body into ~20 blocks, shuffled the accumulators between
// them with v_mov chains and waited vmcnt(0) -- i.e. for the NEXT group's weights, the prefetch -- in front of every MFMA quartet.
template <int MB>
__device__ __forceinline__ void ch_gemm_lds(const ChainOpD& op, const ChView& in, const ChView& out, const ChView& res, bool has_res, int T, unsigned lbase, int wave, int lane) {
    const int MT = (T + 15) >> 4, MG = (MT + MB - 1) / MB, NT = op.N >> 4, KB = op.K >> 4, KS = op.ksplit;
    const int kper = (KB + KS - 1) / KS, items = NT * MG * KS;
    const int r = lane & 15, q = lane >> 4;
    const ChEpi epi{op, out, res, has_res, T, lbase, r, q, MT};
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (int it = wave; it < items; it += kWaves) {
        const int ks = it % KS, rest = it / KS, mg = rest % MG, nt = rest / MG;
        const int kb0 = ks * kper, kb1 = min(KB, kb0 + kper), klast = max(kb1 - 1, 0);
        ch_gf* wrow = (ch_gf*)reinterpret_cast<unsigned long long>(op.w) + (long)(nt * 16 + r) * op.K + 4 * q;
        int tap = (kb0 * 16) / op.cin, c0 = kb0 * 16 - tap * op.cin;
        f32x4 a[4], an[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) an[u] = *reinterpret_cast<ch_gf4*>(wrow + min(kb0 + u, klast) * 16);
        int trow[MB];
#pragma unroll
        for (int m = 0; m < MB; ++m) trow[m] = (mg * MB + m) * 16 + r;
        f32x4 acc[MB];
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[m] = zero4;
        const unsigned inb = in.l + 16u * (unsigned)q;
#pragma nounroll
        for (int g = kb0; g < kb1; g += 4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool live = g + u < kb1;
#pragma unroll
                for (int e = 0; e < 4; ++e) a[u][e] = live ? an[u][e] : 0.f;
            }
            if (g + 4 < kb1) {   // the next group (a branch around the four loads only: a request nobody consumes would still be waited for before its registers are reused)
#pragma unroll
                for (int u = 0; u < 4; ++u) an[u] = *reinterpret_cast<ch_gf4*>(wrow + min(g + 4 + u, klast) * 16);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                f32x4 b[MB];
                const int dt = tap - op.pad;
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    const int row = trow[m] + dt;
                    const bool ok = (unsigned)row < (unsigned)T;   // outside the sample: the convolution's zero padding / a tile row past T (computed, never stored)
                    const f32x4 v = CH_LDS(ch_lf4, inb + 4u * (unsigned)((ok ? row : 0) * in.ld + c0));
#pragma unroll
                    for (int e = 0; e < 4; ++e) b[m][e] = ok ? v[e] : 0.f;
                }
                c0 += 16;
                const bool wrap = c0 >= op.cin;
                c0 = wrap ? 0 : c0;
                tap += wrap ? 1 : 0;
                // (requesting step u + 1's token tiles before the MFMAs of step u -- two register stages -- was measured: no faster, 13 registers more)
#pragma unroll
                for (int m = 0; m < MB; ++m) {
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][0], b[m][0], acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][1], b[m][1], acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][2], b[m][2], acc[m], 0, 0, 0);
                    acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][3], b[m][3], acc[m], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);   // (left alone the scheduler hoists all 12 token loads of the group: 48 registers, spills)
            }
        }
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            if (KS == 1) ch_epilogue(epi, acc[m], nt, mg * MB + m);
            else CH_LDS(ch_lf4, lbase + 16u * (unsigned)(((((nt * MG + mg) * MB + m) * KS) + ks) * 64 + lane)) = acc[m];
            __builtin_amdgcn_sched_barrier(0);   // one tile's epilogue at a time (interleaved, their exp sequences spill)
        }
    }
    if (KS > 1) {
        __syncthreads();
        const int tiles = NT * MG * MB;
        for (int tile = wave; tile < tiles; tile += kWaves) {
            f32x4 acc = CH_LDS(ch_lf4, lbase + 16u * (unsigned)(tile * KS * 64 + lane));
            for (int ks = 1; ks < KS; ++ks) acc += CH_LDS(ch_lf4, lbase + 16u * (unsigned)((tile * KS + ks) * 64 + lane));
            const int m = tile % MB, rest = tile / MB, mg = rest % MG, nt = rest / MG;
            ch_epilogue(epi, acc, nt, mg * MB + m);
        }
    }
}

// One token tile per item and a K slice shorter than one group of four steps (the 32 -> 64 convolutions, the K = 64 projections split in two): the
// guarded loop of rounds 3-4 -- skipping the missing steps costs less here than multiplying zeros (3.2 / 4.4 us against 3.9 / 5.3 per product).
__device__ __forceinline__ void ch_gemm_lds_short(const ChainOpD& op, const ChView& in, const ChView& out, const ChView& res, bool has_res, int T, unsigned lbase, int wave, int lane) {
    constexpr int MBX = 1, MB = 1;
    const int MT = (T + 15) >> 4, MG = (MT + MB - 1) / MB, NT = op.N >> 4, KB = op.K >> 4, KS = op.ksplit;
    const int kper = (KB + KS - 1) / KS, items = NT * MG * KS;
    const int r = lane & 15, q = lane >> 4;
    const ChEpi epi{op, out, res, has_res, T, lbase, r, q, MT};
    const bool lin = op.cin == op.K && op.pad == 0;   // a plain product: row t of every K step (rows past T clamp to the last one: computed, never stored)
    for (int it = wave; it < items; it += kWaves) {
        const int ks = it % KS, rest = it / KS, mg = rest % MG, nt = rest / MG;
        const int kb0 = ks * kper, kb1 = min(KB, kb0 + kper);
        ch_gf* wrow = (ch_gf*)reinterpret_cast<unsigned long long>(op.w) + (long)(nt * 16 + r) * op.K + 4 * q;
        int tap = (kb0 * 16) / op.cin, c0 = kb0 * 16 - tap * op.cin;
        f32x4 a[4], an[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { a[u] = (f32x4){0.f, 0.f, 0.f, 0.f}; if (kb0 + u < kb1) a[u] = *reinterpret_cast<ch_gf4*>(wrow + (kb0 + u) * 16); }
        int trow[MBX];
#pragma unroll
        for (int m = 0; m < MBX; ++m) { trow[m] = (mg * MB + m) * 16 + r; if (lin) trow[m] = min(trow[m], T - 1); }
        f32x4 acc[MBX];
#pragma unroll
        for (int m = 0; m < MBX; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const unsigned inb = in.l + 16u * (unsigned)q;
#pragma nounroll
        for (int g = kb0; g < kb1; g += 4) {
            const bool more = g + 4 < kb1;
            if (more) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { an[u] = (f32x4){0.f, 0.f, 0.f, 0.f}; if (g + 4 + u < kb1) an[u] = *reinterpret_cast<ch_gf4*>(wrow + (g + 4 + u) * 16); }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (g + u < kb1) {
                    f32x4 b[MBX];
                    const int dt = tap - op.pad;
#pragma unroll
                    for (int m = 0; m < MBX; ++m) {
                        const int row = trow[m] + dt;
                        b[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (m < MB && (unsigned)row < (unsigned)T) b[m] = CH_LDS(ch_lf4, inb + 4u * (unsigned)(row * in.ld + c0));
                    }
                    c0 += 16;
                    if (c0 == op.cin) { c0 = 0; ++tap; }
#pragma unroll
                    for (int m = 0; m < MBX; ++m)
                        if (m < MB) {
                            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][0], b[m][0], acc[m], 0, 0, 0);
                            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][1], b[m][1], acc[m], 0, 0, 0);
                            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][2], b[m][2], acc[m], 0, 0, 0);
                            acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][3], b[m][3], acc[m], 0, 0, 0);
                        }
                }
                __builtin_amdgcn_sched_barrier(0);   // (left alone the scheduler hoists all 12 token loads of the group: 48 registers, spills)
            }
            if (more) {
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u] = an[u];
            }
        }
#pragma unroll
        for (int m = 0; m < MBX; ++m)
            if (m < MB) {
                if (KS == 1) ch_epilogue(epi, acc[m], nt, mg * MB + m);
                else CH_LDS(ch_lf4, lbase + 16u * (unsigned)(((((nt * MG + mg) * MB + m) * KS) + ks) * 64 + lane)) = acc[m];
                __builtin_amdgcn_sched_barrier(0);   // one tile's epilogue at a time (interleaved, their exp sequences spill)
            }
    }
    if (KS > 1) {
        __syncthreads();
        const int tiles = NT * MG * MB;
        for (int tile = wave; tile < tiles; tile += kWaves) {
            f32x4 acc = CH_LDS(ch_lf4, lbase + 16u * (unsigned)(tile * KS * 64 + lane));
            for (int ks = 1; ks < KS; ++ks) acc += CH_LDS(ch_lf4, lbase + 16u * (unsigned)((tile * KS + ks) * 64 + lane));
            const int m = tile % MB, rest = tile / MB, mg = rest % MG, nt = rest / MG;
            ch_epilogue(epi, acc, nt, mg * MB + m);
        }
    }
}

// LayerNorm over rows of C floats: 16 lanes per row (4 rows per wave at a time), each lane float4s of the row (C <= 256 and a
// multiple of 4: the row stays in registers) or single floats (any C, re-read per pass); mean, then the variance of the centred
// values, the 16 partial sums combined by DPP rotations
__device__ __forceinline__ void ch_layernorm(const ChainOpD& op, const ChView& x, const ChView& y, int T, unsigned lbase, int wave, int lane) {
    const int C = op.N, sub = lane & 15;
    const float rc = 1.0f / (float)C;
    for (int row = wave * 4 + (lane >> 4); row < T + 3; row += kWaves * 4) {   // (+3: the lanes of a wave stay together for the DPP steps)
        const bool live = row < T;
        const int xr = (live ? row : T - 1) * x.ld;
        if (C <= 256 && (C & 3) == 0) {
            float4 xv[4];
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (64 * j + 4 * sub < C) xv[j] = ch_ld4(x, xr + 64 * j + 4 * sub);
                s += (xv[j].x + xv[j].y) + (xv[j].z + xv[j].w);
            }
            const float mean = ch_row16_sum(s) * rc;
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (64 * j + 4 * sub < C) {
                    const float d0 = xv[j].x - mean, d1 = xv[j].y - mean, d2 = xv[j].z - mean, d3 = xv[j].w - mean;
                    v += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
                }
            const float inv = 1.0f / sqrtf(ch_row16_sum(v) * rc + op.eps);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int i = 64 * j + 4 * sub;
                if (live && i < C) {
                    f32x4 t = {(xv[j].x - mean) * inv, (xv[j].y - mean) * inv, (xv[j].z - mean) * inv, (xv[j].w - mean) * inv};
                    if (op.w_l >= 0) t *= CH_LDS(ch_lf4, lbase + 4u * (unsigned)(op.w_l + i));
                    if (op.bias_l >= 0) t += CH_LDS(ch_lf4, lbase + 4u * (unsigned)(op.bias_l + i));
                    ch_st4(y, row * y.ld + i, ch_f4(t));
                }
            }
        } else {
            float s = 0.f;
#pragma nounroll
            for (int i = sub; i < C; i += 16) s += ch_ld1(x, xr + i);
            const float mean = ch_row16_sum(s) * rc;
            float v = 0.f;
#pragma nounroll
            for (int i = sub; i < C; i += 16) { const float d = ch_ld1(x, xr + i) - mean; v += d * d; }
            const float inv = 1.0f / sqrtf(ch_row16_sum(v) * rc + op.eps);
            if (live) {
#pragma nounroll
                for (int i = sub; i < C; i += 16) {
                    float t = (ch_ld1(x, xr + i) - mean) * inv;
                    if (op.w_l >= 0) t *= CH_LDS(ch_lf, lbase + 4u * (unsigned)(op.w_l + i));
                    if (op.bias_l >= 0) t += CH_LDS(ch_lf, lbase + 4u * (unsigned)(op.bias_l + i));
                    ch_st1(y, row * y.ld + i, t);
                }
            }
        }
    }
}

// softmax(scale * q k^T) v on the matrix pipe (T <= 64): one wave per (head, tile of 16 query rows).  Scores transposed, S^T = K (Q
// scale)^T: keys are the A rows, queries the B columns, so a lane ends up with the scores of keys {16 jt + 4 q + i} for ITS query
// r -- the softmax over keys is 4 JT registers plus two cross-row exchanges, and the probabilities are already laid out as the B
// operand of O^T = V^T P (K index = the lane group's key), whose result is 4 consecutive output channels of one query: one float4.
template <int MT>   // token tiles, 1..4 (a template parameter: `jt < MT` tested at run time around the MFMAs splits the unrolled body into blocks, see ch_gemm_lds)
__device__ __forceinline__ void ch_attention_mfma(const ChainOpD& op, const ChView& qkv, const ChView& out, int T, int wave, int lane) {
    constexpr int JX = MT;
    const int heads = op.heads, hd = op.hd, dim = heads * hd;
    const int r = lane & 15, q = lane >> 4;
    const bool dq = 4 * q < hd;
    for (int u = wave; u < heads * MT; u += kWaves) {
        const int h = u / MT, tt = u - h * MT;
        const int tq = min(tt * 16 + r, T - 1);
        float4 qv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (dq) qv = ch_ld4(qkv, tq * qkv.ld + h * hd + 4 * q);
        qv.x *= op.scale; qv.y *= op.scale; qv.z *= op.scale; qv.w *= op.scale;
        f32x4 sc[JX];
        float m = -3.402823466e38f;
#pragma unroll
        for (int jt = 0; jt < JX; ++jt) {
            sc[jt] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (jt < MT) {
                float4 kv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (dq) kv = ch_ld4(qkv, min(jt * 16 + r, T - 1) * qkv.ld + dim + h * hd + 4 * q);
                sc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.x, qv.x, sc[jt], 0, 0, 0);
                sc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.y, qv.y, sc[jt], 0, 0, 0);
                sc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.z, qv.z, sc[jt], 0, 0, 0);
                sc[jt] = __builtin_amdgcn_mfma_f32_16x16x4f32(kv.w, qv.w, sc[jt], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (jt * 16 + 4 * q + i >= T) sc[jt][i] = -3.402823466e38f;
                    m = fmaxf(m, sc[jt][i]);
                }
            }
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float l = 0.f;
#pragma unroll
        for (int jt = 0; jt < JX; ++jt)
            if (jt < MT) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float p = jt * 16 + 4 * q + i < T ? expf(sc[jt][i] - m) : 0.f;
                    sc[jt][i] = p;
                    l += p;
                }
            }
        l += __shfl_xor(l, 16, 64);
        l += __shfl_xor(l, 32, 64);
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jt = 0; jt < JX; ++jt)
            if (jt < MT) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int j = jt * 16 + 4 * q + i;
                    float vv = 0.f;
                    if (j < T && r < hd) vv = ch_ld1(qkv, j * qkv.ld + 2 * dim + h * hd + r);
                    o = __builtin_amdgcn_mfma_f32_16x16x4f32(vv, sc[jt][i], o, 0, 0, 0);
                }
            }
        const int t = tt * 16 + r;
        if (t < T && dq) {
            const float rl = 1.0f / l;
            ch_st4(out, t * out.ld + h * hd + 4 * q, make_float4(o[0] * rl, o[1] * rl, o[2] * rl, o[3] * rl));
        }
    }
}

// any T: one thread per (head, query row), two passes over the keys (maximum, then exp / sum / weighted V) -- attention_kernel's
// statement sequence (kernels.hip), K and V read where the QKV projection left them; compact, not fast
template <int HD>
__device__ __forceinline__ void ch_attention_any(const ChainOpD& op, const ChView& qkv, const ChView& out, int T) {
    const int heads = op.heads, hd = op.hd, dim = heads * hd;
    for (int i = threadIdx.x; i < heads * T; i += kChainThreads) {
        const int h = i / T, t = i - h * T;
        float qv[HD], o[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) { qv[d] = d < hd ? ch_ld1(qkv, t * qkv.ld + h * hd + d) * op.scale : 0.f; o[d] = 0.f; }
        auto score = [&](int j) __attribute__((always_inline)) {
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < HD; ++d) if (d < hd) a = fmaf(qv[d], ch_ld1(qkv, j * qkv.ld + dim + h * hd + d), a);
            return a;
        };
        float m = -3.402823466e38f;
#pragma nounroll
        for (int j = 0; j < T; ++j) m = fmaxf(m, score(j));
        float l = 0.f;
#pragma nounroll
        for (int j = 0; j < T; ++j) {
            const float p = expf(score(j) - m);
            l += p;
#pragma unroll
            for (int d = 0; d < HD; ++d) if (d < hd) o[d] = fmaf(p, ch_ld1(qkv, j * qkv.ld + 2 * dim + h * hd + d), o[d]);
        }
#pragma unroll
        for (int d = 0; d < HD; ++d) if (d < hd) ch_st1(out, t * out.ld + h * hd + d, o[d] / l);
    }
}

__device__ __forceinline__ void ch_copy(const ChainOpD& op, const ChView& x, const ChView& y, int T) {
    const int c4 = op.N >> 2;
    for (int i = threadIdx.x; i < T * c4; i += kChainThreads) {
        const int t = i / c4, c = i - t * c4;
        ch_st4(y, t * y.ld + 4 * c, ch_ld4(x, t * x.ld + 4 * c));
    }
}

// AveragePool whose window spans the whole height and `kw` columns per token (the recognizer's [6, 2] pool in front of the SVTR neck): the
// sample's [kh][Wp][C] map (Wp = op.pad >= kw * T: a last odd column is dropped, as the pool does) -> T rows of C, pool2d_kernel's statement
// order (rows outside, columns inside, one division at the end).  The window's loads are issued six at a time, then added in that order.
__device__ __forceinline__ void ch_pool(const ChainOpD& op, const ChView& x, const ChView& y, int T) {
    const int c4 = op.N >> 2, kh = op.K, kw = op.cin, Wp = op.pad, win = kh * kw;
    ch_gf* xs = x.g + ((long)blockIdx.x * ((long)kh * Wp - T)) * x.ld;   // x.g is base + sample * T * ld; the sample's map starts at row sample * kh * Wp
    const float div = (float)win;
    for (int i = threadIdx.x; i < T * c4; i += kChainThreads) {
        const int t = i / c4, c = i - t * c4;
        ch_gf* px = xs + ((long)t * kw) * x.ld + 4 * c;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int q0 = 0; q0 < win; q0 += 6) {
            f32x4 v[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int q = min(q0 + u, win - 1), a = q / kw, b = q - a * kw;
                v[u] = *reinterpret_cast<ch_gf4*>(px + ((long)a * Wp + b) * x.ld);
            }
#pragma unroll
            for (int u = 0; u < 6; ++u) if (q0 + u < win) acc += v[u];
        }
        ch_st4(y, t * y.ld + 4 * c, make_float4(acc[0] / div, acc[1] / div, acc[2] / div, acc[3] / div));
    }
}

// consts: the chain's constants in one allocation -- [0, small) floats = biases / LayerNorm affine vectors (copied into LDS at
// `small_l`), the rest = the products' weights.  Before the first operator each workgroup pulls its share of the weights towards
// its XCD's L2 (workgroup b runs on XCD b % 8: the 32 workgroups of an XCD split the blob), so that the operators find them there
// instead of paying an HBM (and TLB) miss each.
template <int HD>
__global__ __launch_bounds__(kChainThreads) void chain_kernel(const ChainOpD* __restrict__ ops, int n_ops, int T, char* arena, const char* input,
                                                               const float* __restrict__ consts, int small, int total_consts, int small_l, int tab_l, int stamp_l, unsigned long long* dbg) {
    extern __shared__ float4 chain_lds[];
    const unsigned lbase = (unsigned)reinterpret_cast<unsigned long long>(chain_lds);   // LDS byte address of the dynamic segment (low half of its flat address)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const long row0 = (long)blockIdx.x * T;
    const unsigned stamps = lbase + 4u * (unsigned)stamp_l;   // OAR_CHAIN_DBG: kept on chip until the end (a store to HBM per operator would be waited for at the next barrier)
    if (dbg && threadIdx.x == 0) CH_LDS(__attribute__((address_space(3))) unsigned long long, stamps) = wall_clock64();
    const unsigned long long cyc0 = dbg ? clock64() : 0;
    float warm = 0.f;
    {
        const int lines = (total_consts - small + 31) / 32;                 // 128-byte lines of weights
        const int parts = min(32, max(1, (int)(gridDim.x / 8)));
        const int part = (int)(blockIdx.x / 8) % parts, per = (lines + parts - 1) / parts;
        for (int i = part * per + (int)threadIdx.x; i < min(lines, (part + 1) * per); i += kChainThreads) warm += consts[small + i * 32];
        for (int i = threadIdx.x; i < small; i += kChainThreads) CH_LDS(ch_lf, lbase + 4u * (unsigned)(small_l + i)) = consts[i];
    }
    // the operator table goes to LDS in one sweep (a descriptor fetched from HBM per operator costs more than most operators do)
    constexpr int kOpWords = (int)(sizeof(ChainOpD) / 4);
    for (int i = threadIdx.x; i < n_ops * kOpWords; i += kChainThreads) CH_LDS(ch_li, lbase + 4u * (unsigned)(tab_l + i)) = reinterpret_cast<const int*>(ops)[i];
    __syncthreads();
    auto read_op = [&](int i) __attribute__((always_inline)) {
        int w[kOpWords];
#pragma unroll
        for (int k = 0; k < kOpWords; ++k) w[k] = __builtin_amdgcn_readfirstlane(CH_LDS(ch_li, lbase + 4u * (unsigned)(tab_l + i * kOpWords + k)));
        ChainOpD op;
        __builtin_memcpy(&op, w, sizeof op);
        return op;
    };
    for (int i = 0; i < n_ops; ++i) {
        const ChainOpD op = read_op(i);
        const ChView in = ch_view(op.in, op.in_ld, row0, arena, input, lbase), out = ch_view(op.out, op.out_ld, row0, arena, input, lbase);
        switch (op.type) {
            case CH_GEMM: {
                const bool has_res = op.res.kind >= 0;
                const ChView res = ch_view(op.res, op.res_ld, row0, arena, input, lbase);
                if (in.lds && op.mb == 1 && ((op.K >> 4) + op.ksplit - 1) / op.ksplit < 4) ch_gemm_lds_short(op, in, out, res, has_res, T, lbase, wave, lane);
                else if (in.lds && op.mb == 1) ch_gemm_lds<1>(op, in, out, res, has_res, T, lbase, wave, lane);
                else if (in.lds && op.mb == 2) ch_gemm_lds<2>(op, in, out, res, has_res, T, lbase, wave, lane);
                else if (in.lds) ch_gemm_lds<3>(op, in, out, res, has_res, T, lbase, wave, lane);
                else ch_gemm_slow(op, in, out, res, has_res, T, lbase, wave, lane);
                break;
            }
            case CH_LN: ch_layernorm(op, in, out, T, lbase, wave, lane); break;
            case CH_ATTN:
                if (T <= 64 && (op.hd & 3) == 0) {
                    if (T <= 16) ch_attention_mfma<1>(op, in, out, T, wave, lane);
                    else if (T <= 32) ch_attention_mfma<2>(op, in, out, T, wave, lane);
                    else if (T <= 48) ch_attention_mfma<3>(op, in, out, T, wave, lane);
                    else ch_attention_mfma<4>(op, in, out, T, wave, lane);
                } else ch_attention_any<HD>(op, in, out, T);
                break;
            case CH_POOL: ch_pool(op, in, out, T); break;
            default: ch_copy(op, in, out, T); break;
        }
        __syncthreads();   // workgroup-scope release / acquire: the next operator reads what this one stored (LDS, or HBM through this CU's L1)
        if (dbg && threadIdx.x == 0 && i < 70) CH_LDS(__attribute__((address_space(3))) unsigned long long, stamps + 8u * (unsigned)(i + 1)) = wall_clock64();
    }
    if (dbg && blockIdx.x == 0 && threadIdx.x == 0)
    {
        for (int i = 0; i <= min(n_ops, 70); ++i) dbg[i] = CH_LDS(__attribute__((address_space(3))) unsigned long long, stamps + 8u * (unsigned)i);
        dbg[n_ops + 1] = clock64() - cyc0;
    }
    if (warm == 1.2345678e-30f && dbg) dbg[0] = 0;   // (keeps the warming loads alive)
}
}  // namespace

size_t chain_lds_bytes(const ChainOpD& op, int T) {   // scratch at the bottom of LDS: split-K partials
    if (op.type == CH_GEMM && op.ksplit > 1) { const int MT = (T + 15) / 16, mb = op.mb < 1 ? 1 : op.mb; return (size_t)(op.N / 16) * ((MT + mb - 1) / mb) * mb * op.ksplit * 64 * 16; }
    return 0;
}

void chain_run(hipStream_t s, const ChainLaunch& L, char* arena, const char* input) {
    if (L.n_samples <= 0 || L.T <= 0 || L.n_ops <= 0) return;
    OAR_CHECK(L.lds + 640 <= 160 * 1024 && L.max_hd <= kChainMaxHd, OAR_INTERNAL, "chain: LDS / head size beyond what the planner may fuse");
    OAR_MAX_LDS_ONCE(chain_kernel<kChainMaxHd>, 160 * 1024);
    ProfScope ps(s, "chain", L.bytes, L.flops, true);
    // OAR_CHAIN_DBG=1: per-operator wall-clock stamps of workgroup 0 (100 MHz constant clock), printed after a synchronising read-back
    static const bool dbg_on = [] { const char* e = getenv("OAR_CHAIN_DBG"); return e && atoi(e) != 0; }();
    unsigned long long* dbg = nullptr;
    if (dbg_on) { OAR_HIP(hipMalloc(&dbg, (size_t)(80 + 4 * L.n_ops + 8) * 8)); OAR_HIP(hipMemset(dbg, 0, (size_t)(80 + 4 * L.n_ops + 8) * 8)); }
    hipExtLaunchKernelGGL((chain_kernel<kChainMaxHd>), dim3((unsigned)L.n_samples), dim3(kChainThreads), L.lds + 640, s, ps.start(), ps.stop(), 0, L.ops, L.n_ops, L.T, arena, input,
                          L.consts, L.small, L.total_consts, L.small_l, L.tab_l, (int)(L.lds / 4), dbg);
    if (dbg_on) {
        const int n_ops = L.n_ops;
        std::vector<unsigned long long> h((size_t)(80 + 4 * n_ops + 8));
        std::vector<ChainOpD> ho((size_t)n_ops);
        OAR_HIP(hipStreamSynchronize(s));
        OAR_HIP(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
        OAR_HIP(hipMemcpy(ho.data(), L.ops, ho.size() * sizeof(ChainOpD), hipMemcpyDeviceToHost));
        OAR_HIP(hipFree(dbg));
        fprintf(stderr, "[chain] n=%d T=%d ops=%d lds=%zu total %.2f us, %llu shader-clock ticks\n", L.n_samples, L.T, n_ops, L.lds, (double)(h[(size_t)n_ops] - h[0]) / 100.0, h[(size_t)n_ops + 1]);
        for (int i = 0; i < n_ops; ++i) fprintf(stderr, "[chain]   op %2d type %d K=%4d N=%4d ks=%d mb=%d in/out/res kind %d/%d/%d  %.2f us\n", i, ho[(size_t)i].type, ho[(size_t)i].K, ho[(size_t)i].N, ho[(size_t)i].ksplit, ho[(size_t)i].mb, ho[(size_t)i].in.kind, ho[(size_t)i].out.kind, ho[(size_t)i].res.kind, (double)(h[(size_t)i + 1] - h[(size_t)i]) / 100.0);
    }
    OAR_HIP(hipGetLastError());
}

}  // namespace k
}  // namespace oar
