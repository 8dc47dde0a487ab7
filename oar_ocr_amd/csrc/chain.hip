// chain.hip -- a run of sample-local operators as ONE launch (round 3, VERDICT r2 #3: the sub-20 us launch tail).
//
// The SVTR neck of a recognizer (EncoderWithSVTR: 1x3 conv -> 1x1 conv -> 2 x [LN -> QKV -> attention -> proj + residual -> LN ->
// FC1 -> FC2 + residual] -> LN -> 1x1 conv -> concat -> 1x3 conv -> 1x1 conv) touches, per text line, T <= a few dozen tokens of
// <= 512 channels: every operator is a [T x K] x [K x N] product, a row normalisation, a T x T attention or a row copy that only
// reads rows of its OWN sample.  As separate launches that is 22 kernels of 5-30 us each whose grids cannot fill 256 CUs; here ONE
// workgroup (16 waves) owns a sample and walks the operator table, a workgroup barrier between operators, the intermediate
// tensors in the planner's arena (they are KBs per sample: L2 / L1 hits).  The planner (engine.cc, Planner::fuse_chains) records the
// operators, keeps every tensor of the run allocated until the run ends (no two of them alias, so samples may run at
// different paces) and replaces the run by one `chain_run` step.
//
// Arithmetic: products on v_mfma_f32_16x16x4_f32 (exact f32 multiply-add, weights as the A operand, tokens as B: a lane ends up
// with 4 consecutive output channels of one token = one float4 store); split-K partials, when a product has too few tiles for 16
// waves, are reduced through LDS in a fixed order (deterministic).  LayerNorm and attention are the statement sequences of
// layernorm_kernel / attention_kernel (kernels.hip), so those results are bit-identical to the unfused path given equal inputs.
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

// out[t][n] = act(bias[n] + sum_{tap, c} in[t + tap - pad][c] * w[n][tap * cin + c]) (+ res[t][n]),  t in [0, T)
// General path (tokens in HBM: only when the planner could not stage them in LDS): one work item = one 16 x 16 output tile over one
// K slice; K walks in groups of four 16-wide steps, the next group's operands in flight while the current group's 16 MFMAs issue.
__device__ __forceinline__ void ch_gemm_slow(const ChainOpD& op, const ChView& in, const ChView& out, const ChView& res, bool has_res, int T, unsigned lbase, int wave, int lane, unsigned ph) {
    const int MT = (T + 15) >> 4, NT = op.N >> 4, KB = op.K >> 4, KS = op.ksplit;
    const int kper = (KB + KS - 1) / KS, tiles = NT * MT, items = tiles * KS;
    const int r = lane & 15, q = lane >> 4;
    auto epilogue = [&](f32x4 acc, int tile) __attribute__((always_inline)) {
        const int nt = tile / MT, mt = tile - nt * MT;
        const int t = mt * 16 + r, n0 = nt * 16 + 4 * q;
        if (t >= T) return;
        if (op.bias_l >= 0) { const float4 b = ch_f4(CH_LDS(ch_lf4, lbase + 4u * (unsigned)(op.bias_l + n0))); acc[0] += b.x; acc[1] += b.y; acc[2] += b.z; acc[3] += b.w; }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = apply_act(acc[i], op.act, op.alpha, op.beta);
        if (has_res) { const float4 v = ch_ld4(res, t * res.ld + n0); acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w; }
        ch_st4(out, t * out.ld + n0, make_float4(acc[0], acc[1], acc[2], acc[3]));
    };
    if (ph && threadIdx.x == 0) CH_LDS(__attribute__((address_space(3))) unsigned long long, ph) = clock64();
    for (int it = wave; it < items; it += kWaves) {
        const int tile = it / KS, ks = it - tile * KS;
        const int nt = tile / MT, mt = tile - nt * MT;
        const int t = mt * 16 + r;
        ch_gf* wrow = (ch_gf*)(unsigned long long)reinterpret_cast<unsigned long long>(op.w) + (long)(nt * 16 + r) * op.K + 4 * q;
        const int kb0 = ks * kper, kb1 = min(KB, kb0 + kper);
        int kb = kb0;                                        // the next step to LOAD
        int tap = (kb * 16) / op.cin, c0 = kb * 16 - tap * op.cin;
        float4 a[4], b[4], an[4], bn[4];
        auto load_group = [&](float4 (&A)[4], float4 (&B)[4]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                A[u] = make_float4(0.f, 0.f, 0.f, 0.f); B[u] = A[u];
                if (kb < kb1) {
                    A[u] = ch_f4(*reinterpret_cast<ch_gf4*>(wrow + kb * 16));
                    const int row = t + tap - op.pad;
                    if (row >= 0 && row < T) B[u] = ch_ld4(in, row * in.ld + c0 + 4 * q);
                    ++kb; c0 += 16;
                    if (c0 == op.cin) { c0 = 0; ++tap; }
                }
            }
        };
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        load_group(a, b);
        if (ph && threadIdx.x == 0 && it == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); CH_LDS(__attribute__((address_space(3))) unsigned long long, ph + 8) = clock64(); }
        for (int g = kb0; g < kb1; g += 4) {
            const bool more = g + 4 < kb1;
            if (more) load_group(an, bn);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].x, b[u].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].y, b[u].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].z, b[u].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u].w, b[u].w, acc, 0, 0, 0);
            }
            if (more) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { a[u] = an[u]; b[u] = bn[u]; }
            }
        }
        if (ph && threadIdx.x == 0 && it == 0) { asm volatile("s_nop 0" : "+v"(acc)); CH_LDS(__attribute__((address_space(3))) unsigned long long, ph + 16) = clock64(); }
        if (KS == 1) epilogue(acc, tile);
        else CH_LDS(ch_lf4, lbase + 16u * (unsigned)(it * 64 + lane)) = acc;
    }
    if (KS > 1) {
        __syncthreads();
        for (int tile = wave; tile < tiles; tile += kWaves) {
            f32x4 acc = CH_LDS(ch_lf4, lbase + 16u * (unsigned)(tile * KS * 64 + lane));
            for (int ks = 1; ks < KS; ++ks) { const f32x4 v = CH_LDS(ch_lf4, lbase + 16u * (unsigned)((tile * KS + ks) * 64 + lane)); acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3]; }
            epilogue(acc, tile);
        }
    }
    if (ph && threadIdx.x == 0) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); CH_LDS(__attribute__((address_space(3))) unsigned long long, ph + 24) = clock64(); }
}

// weights of a work item's first K group (4 steps x float4 per lane), zero beyond the item's K slice: issued one operator ahead
__device__ __forceinline__ void ch_load_a(const ChainOpD& op, int T, int it, int lane, float4 (&A)[4]) {
    const int MT = (T + 15) >> 4, MB = op.mb, MG = (MT + MB - 1) / MB, KB = op.K >> 4, KS = op.ksplit, kper = (KB + KS - 1) / KS;
    const int ks = it % KS, nt = (it / KS) / MG;
    const int kb0 = ks * kper, kb1 = min(KB, kb0 + kper);
    ch_gf* wrow = (ch_gf*)reinterpret_cast<unsigned long long>(op.w) + (long)(nt * 16 + (lane & 15)) * op.K + 4 * (lane >> 4);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        A[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (kb0 + u < kb1) A[u] = ch_f4(*reinterpret_cast<ch_gf4*>(wrow + (kb0 + u) * 16));
    }
}

// Fast path: tokens in LDS.  One work item = 16 output channels x MB token tiles (the weights of a K step are loaded once for all of
// them) over one K slice; items = channel tiles x token groups x K slices, chosen by the planner to occupy the 16 waves.  The first
// weight group of a wave's first item arrives in `pa` (loaded while the previous operator ran).
template <int MB>
__device__ __forceinline__ void ch_gemm_lds(const ChainOpD& op, const ChView& in, const ChView& out, const ChView& res, bool has_res, int T, unsigned lbase, int wave, int lane,
                                            const float4 (&pa)[4], bool have_pa) {
    const int MT = (T + 15) >> 4, MG = (MT + MB - 1) / MB, NT = op.N >> 4, KB = op.K >> 4, KS = op.ksplit;
    const int kper = (KB + KS - 1) / KS, items = NT * MG * KS;
    const int r = lane & 15, q = lane >> 4;
    auto epilogue = [&](f32x4 acc, int nt, int mtile) __attribute__((always_inline)) {
        const int t = mtile * 16 + r, n0 = nt * 16 + 4 * q;
        if (mtile >= MT || t >= T) return;
        if (op.bias_l >= 0) { const float4 b = ch_f4(CH_LDS(ch_lf4, lbase + 4u * (unsigned)(op.bias_l + n0))); acc[0] += b.x; acc[1] += b.y; acc[2] += b.z; acc[3] += b.w; }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = apply_act(acc[i], op.act, op.alpha, op.beta);
        if (has_res) { const float4 v = ch_ld4(res, t * res.ld + n0); acc[0] += v.x; acc[1] += v.y; acc[2] += v.z; acc[3] += v.w; }
        ch_st4(out, t * out.ld + n0, make_float4(acc[0], acc[1], acc[2], acc[3]));
    };
    for (int it = wave; it < items; it += kWaves) {
        const int ks = it % KS, rest = it / KS, mg = rest % MG, nt = rest / MG;
        const int kb0 = ks * kper, kb1 = min(KB, kb0 + kper);
        ch_gf* wrow = (ch_gf*)reinterpret_cast<unsigned long long>(op.w) + (long)(nt * 16 + r) * op.K + 4 * q;
        int tap = (kb0 * 16) / op.cin, c0 = kb0 * 16 - tap * op.cin;
        float4 a[4], an[4];
        if (it == wave && have_pa) {
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = pa[u];
        } else {
#pragma unroll
            for (int u = 0; u < 4; ++u) { a[u] = make_float4(0.f, 0.f, 0.f, 0.f); if (kb0 + u < kb1) a[u] = ch_f4(*reinterpret_cast<ch_gf4*>(wrow + (kb0 + u) * 16)); }
        }
        f32x4 acc[MB];
#pragma unroll
        for (int m = 0; m < MB; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
        const unsigned inb = in.l + 16u * (unsigned)q;
        for (int g = kb0; g < kb1; g += 4) {
            if (g + 4 < kb1) {
#pragma unroll
                for (int u = 0; u < 4; ++u) { an[u] = make_float4(0.f, 0.f, 0.f, 0.f); if (g + 4 + u < kb1) an[u] = ch_f4(*reinterpret_cast<ch_gf4*>(wrow + (g + 4 + u) * 16)); }
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                if (g + 2 * half >= kb1) break;
                f32x4 b[2][MB];
#pragma unroll
                for (int uu = 0; uu < 2; ++uu) {
                    const bool on = g + 2 * half + uu < kb1;
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
                        const int row = (mg * MB + m) * 16 + r + tap - op.pad;
                        b[uu][m] = (f32x4){0.f, 0.f, 0.f, 0.f};
                        if (on && row >= 0 && row < T) b[uu][m] = CH_LDS(ch_lf4, inb + 4u * (unsigned)(row * in.ld + c0));
                    }
                    if (on) { c0 += 16; if (c0 == op.cin) { c0 = 0; ++tap; } }
                }
#pragma unroll
                for (int uu = 0; uu < 2; ++uu) {
                    const float4 av = a[2 * half + uu];
#pragma unroll
                    for (int m = 0; m < MB; ++m) {
                        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, b[uu][m][0], acc[m], 0, 0, 0);
                        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, b[uu][m][1], acc[m], 0, 0, 0);
                        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.z, b[uu][m][2], acc[m], 0, 0, 0);
                        acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.w, b[uu][m][3], acc[m], 0, 0, 0);
                    }
                }
            }
            if (g + 4 < kb1) {
#pragma unroll
                for (int u = 0; u < 4; ++u) a[u] = an[u];
            }
        }
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            if (KS == 1) epilogue(acc[m], nt, mg * MB + m);
            else CH_LDS(ch_lf4, lbase + 16u * (unsigned)(((((nt * MG + mg) * MB + m) * KS) + ks) * 64 + lane)) = acc[m];
        }
    }
    if (KS > 1) {
        __syncthreads();
        const int tiles = NT * MG * MB;
        for (int tile = wave; tile < tiles; tile += kWaves) {
            f32x4 acc = CH_LDS(ch_lf4, lbase + 16u * (unsigned)(tile * KS * 64 + lane));
            for (int ks = 1; ks < KS; ++ks) { const f32x4 v = CH_LDS(ch_lf4, lbase + 16u * (unsigned)((tile * KS + ks) * 64 + lane)); acc[0] += v[0]; acc[1] += v[1]; acc[2] += v[2]; acc[3] += v[3]; }
            const int m = tile % MB, rest = tile / MB, mg = rest % MG, nt = rest / MG;
            epilogue(acc, nt, mg * MB + m);
        }
    }
}

// LayerNorm over rows of C floats: 16 lanes per row (4 rows per wave at a time), the row in registers (C <= 256: 16 per lane) or
// re-read (wider rows); mean, then the variance of the centred values, the 16 partial sums combined by DPP rotations
__device__ __forceinline__ void ch_layernorm(const ChainOpD& op, const ChView& x, const ChView& y, int T, unsigned lbase, int wave, int lane) {
    const int C = op.N, sub = lane & 15;
    const float rc = 1.0f / (float)C;
    for (int row = wave * 4 + (lane >> 4); row < T + 3; row += kWaves * 4) {   // (+3: the lanes of a wave stay together for the DPP steps)
        const bool live = row < T;
        const int xr = (live ? row : T - 1) * x.ld;
        if (C <= 256) {
            float xv[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) { xv[j] = 0.f; if (16 * j + sub < C) xv[j] = ch_ld1(x, xr + 16 * j + sub); }
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) s += xv[j];
            const float mean = ch_row16_sum(s) * rc;
            float v = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) if (16 * j + sub < C) { const float d = xv[j] - mean; v += d * d; }
            const float inv = 1.0f / sqrtf(ch_row16_sum(v) * rc + op.eps);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int i = 16 * j + sub;
                if (live && i < C) {
                    float t = (xv[j] - mean) * inv;
                    if (op.w_l >= 0) t *= CH_LDS(ch_lf, lbase + 4u * (unsigned)(op.w_l + i));
                    if (op.bias_l >= 0) t += CH_LDS(ch_lf, lbase + 4u * (unsigned)(op.bias_l + i));
                    ch_st1(y, row * y.ld + i, t);
                }
            }
        } else {
            float s = 0.f;
            for (int i = sub; i < C; i += 16) s += ch_ld1(x, xr + i);
            const float mean = ch_row16_sum(s) * rc;
            float v = 0.f;
            for (int i = sub; i < C; i += 16) { const float d = ch_ld1(x, xr + i) - mean; v += d * d; }
            const float inv = 1.0f / sqrtf(ch_row16_sum(v) * rc + op.eps);
            if (live)
                for (int i = sub; i < C; i += 16) {
                    float t = (ch_ld1(x, xr + i) - mean) * inv;
                    if (op.w_l >= 0) t *= CH_LDS(ch_lf, lbase + 4u * (unsigned)(op.w_l + i));
                    if (op.bias_l >= 0) t += CH_LDS(ch_lf, lbase + 4u * (unsigned)(op.bias_l + i));
                    ch_st1(y, row * y.ld + i, t);
                }
        }
    }
}

// softmax(scale * q k^T) v per head: a quad of lanes owns one (head, query row), each lane every fourth key; two passes over its
// keys (maximum, then exp / sum / weighted V, the scores recomputed), the quad's partial maxima / sums / outputs combined by DPP,
// lane s of the quad storing output channels [4 s, 4 s + 4) -- K and V are read where the QKV projection left them.
template <int HD>
__device__ __forceinline__ void ch_attention(const ChainOpD& op, const ChView& qkv, const ChView& out, int T) {
    constexpr int H4 = HD / 4;
    const int heads = op.heads, hd = op.hd, dim = heads * hd, h4 = hd >> 2;
    const int total = heads * T * 4;
    for (int i0 = 0; i0 < total; i0 += kChainThreads) {
        const int i = i0 + (int)threadIdx.x;
        const bool live = i < total;
        const int ic = live ? i : total - 1;
        const int sl = ic & 3, ht = ic >> 2, h = ht / T, t = ht - h * T;
        const int qb = t * qkv.ld + h * hd;
        float qv[HD];
#pragma unroll
        for (int d4 = 0; d4 < H4; ++d4) {
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (d4 < h4) v = ch_ld4(qkv, qb + 4 * d4);
            qv[4 * d4] = v.x * op.scale; qv[4 * d4 + 1] = v.y * op.scale; qv[4 * d4 + 2] = v.z * op.scale; qv[4 * d4 + 3] = v.w * op.scale;
        }
        auto score = [&](int j) __attribute__((always_inline)) {   // four independent partial sums (one per float4 of the head), then their sum
            float a[H4];
            const int kb = j * qkv.ld + dim + h * hd;
#pragma unroll
            for (int d4 = 0; d4 < H4; ++d4) {
                a[d4] = 0.f;
                if (d4 < h4) {
                    const float4 kk = ch_ld4(qkv, kb + 4 * d4);
                    a[d4] = fmaf(qv[4 * d4 + 3], kk.w, fmaf(qv[4 * d4 + 2], kk.z, fmaf(qv[4 * d4 + 1], kk.y, qv[4 * d4] * kk.x)));
                }
            }
            float sum = a[0];
#pragma unroll
            for (int d4 = 1; d4 < H4; ++d4) sum += a[d4];
            return sum;
        };
        constexpr int kKeep = 16;               // scores of a lane's keys stay in registers when T <= 64
        const bool keep = T <= 4 * kKeep;
        float sc[kKeep];
        float m = -3.402823466e38f;
        if (keep) {
#pragma unroll
            for (int jj = 0; jj < kKeep; ++jj) { sc[jj] = -3.402823466e38f; if (sl + 4 * jj < T) { sc[jj] = score(sl + 4 * jj); m = fmaxf(m, sc[jj]); } }
        } else {
            for (int j = sl; j < T; j += 4) m = fmaxf(m, score(j));
        }
        m = ch_quad_max(m);
        float l = 0.f;
        float o[HD];
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] = 0.f;
        auto weigh = [&](int j, float sj) __attribute__((always_inline)) {
            const float p = expf(sj - m);
            l += p;
            const int vb = j * qkv.ld + 2 * dim + h * hd;
#pragma unroll
            for (int d4 = 0; d4 < H4; ++d4)
                if (d4 < h4) {
                    const float4 vv = ch_ld4(qkv, vb + 4 * d4);
                    o[4 * d4] = fmaf(p, vv.x, o[4 * d4]); o[4 * d4 + 1] = fmaf(p, vv.y, o[4 * d4 + 1]);
                    o[4 * d4 + 2] = fmaf(p, vv.z, o[4 * d4 + 2]); o[4 * d4 + 3] = fmaf(p, vv.w, o[4 * d4 + 3]);
                }
        };
        if (keep) {
#pragma unroll
            for (int jj = 0; jj < kKeep; ++jj) if (sl + 4 * jj < T) weigh(sl + 4 * jj, sc[jj]);
        } else {
            for (int j = sl; j < T; j += 4) weigh(j, score(j));
        }
        l = ch_quad_sum(l);
#pragma unroll
        for (int d = 0; d < HD; ++d) o[d] = ch_quad_sum(o[d]);
        const float rl = 1.0f / l;
#pragma unroll
        for (int d4 = 0; d4 < H4; ++d4)
            if (live && d4 == sl && d4 < h4)
                ch_st4(out, t * out.ld + h * hd + 4 * d4, make_float4(o[4 * d4] * rl, o[4 * d4 + 1] * rl, o[4 * d4 + 2] * rl, o[4 * d4 + 3] * rl));
    }
}

__device__ __forceinline__ void ch_copy(const ChainOpD& op, const ChView& x, const ChView& y, int T) {
    const int c4 = op.N >> 2;
    for (int i = threadIdx.x; i < T * c4; i += kChainThreads) {
        const int t = i / c4, c = i - t * c4;
        ch_st4(y, t * y.ld + 4 * c, ch_ld4(x, t * x.ld + 4 * c));
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
    float4 pa[4];
    bool have_pa = false;
#pragma unroll
    for (int u = 0; u < 4; ++u) pa[u] = make_float4(0.f, 0.f, 0.f, 0.f);
    ChainOpD op = read_op(0);
    for (int i = 0; i < n_ops; ++i) {
        // the next product's first weights start their trip now (they do not depend on this operator's result)
        ChainOpD nxt = op;
        float4 pn[4];
        bool have_pn = false;
#pragma unroll
        for (int u = 0; u < 4; ++u) pn[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i + 1 < n_ops) {
            nxt = read_op(i + 1);
            if (nxt.type == CH_GEMM && nxt.in.kind == 3) {
                const int MTn = (T + 15) >> 4, items = (nxt.N >> 4) * ((MTn + nxt.mb - 1) / nxt.mb) * nxt.ksplit;
                if (wave < items) ch_load_a(nxt, T, wave, lane, pn);
                have_pn = true;
            }
        }
        const ChView in = ch_view(op.in, op.in_ld, row0, arena, input, lbase), out = ch_view(op.out, op.out_ld, row0, arena, input, lbase);
        switch (op.type) {
            case CH_GEMM: {
                const bool has_res = op.res.kind >= 0;
                const ChView res = ch_view(op.res, op.res_ld, row0, arena, input, lbase);
                if (!in.lds) ch_gemm_slow(op, in, out, res, has_res, T, lbase, wave, lane, 0u);
                else if (op.mb == 1) ch_gemm_lds<1>(op, in, out, res, has_res, T, lbase, wave, lane, pa, have_pa);
                else if (op.mb == 2) ch_gemm_lds<2>(op, in, out, res, has_res, T, lbase, wave, lane, pa, have_pa);
                else ch_gemm_lds<3>(op, in, out, res, has_res, T, lbase, wave, lane, pa, have_pa);
                break;
            }
            case CH_LN: ch_layernorm(op, in, out, T, lbase, wave, lane); break;
            case CH_ATTN: ch_attention<HD>(op, in, out, T); break;
            default: ch_copy(op, in, out, T); break;
        }
        __syncthreads();   // workgroup-scope release / acquire: the next operator reads what this one stored (LDS, or HBM through this CU's L1)
        if (dbg && threadIdx.x == 0 && i < 70) CH_LDS(__attribute__((address_space(3))) unsigned long long, stamps + 8u * (unsigned)(i + 1)) = wall_clock64();
        op = nxt;
#pragma unroll
        for (int u = 0; u < 4; ++u) pa[u] = pn[u];
        have_pa = have_pn;
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
    static const bool once = [] {
        OAR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_kernel<kChainMaxHd>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        return true;
    }();
    (void)once;
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
