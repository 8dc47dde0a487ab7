// attention_x6.hip -- softmax(scale * q k^T) v on the bf16 matrix pipe with f32-equivalent accuracy (the exact 3-way bf16 split
// "bf16x6" of igemm_ws_x6.hip on BOTH products), flash-style: the [T, T] score matrix never exists, K and V stream through LDS in
// blocks of 32 keys.  For the global-mixing blocks of the SVTRv2-class recognizer (head dim 32, T = 480 ... 4800 tokens per crop:
// BASELINE C3), where one head's K and V do not fit the LDS-resident kernel of kernels.hip (T * hd <= 19200) and one thread per query
// on the vector ALU would take 3 T^2 hd FMAs per head.
//
//   * a workgroup (4 waves) owns 128 queries of one (crop, head); a wave owns 32 of them as two B-operand fragments of
//     v_mfma_f32_16x16x32_bf16 (lane (q = lane & 15, g = lane >> 4) holds components 8 g .. 8 g + 7 of query q), split once into three
//     bf16 planes, pre-multiplied by scale * log2(e) so that the soft-max is exp2 of the accumulator;
//   * K block (32 keys x 32 components): A operand of S^T = K Q^T, staged in LDS already split and in fragment order by 128 of the
//     workgroup's threads (one 32-byte group of one key each);  a wave's accumulator lane (q, g) then holds the scores of query q against
//     keys 4 g + r (tile 0) and 16 + 4 g + r (tile 1), r = 0..3;
//   * these 8 probabilities ARE the lane's B-operand slots of the second product O^T = V^T P^T if the 32 keys of the block are taken in
//     the order slot (g, j) -> key (j < 4 ? 4 g + j : 16 + 4 g + j - 4): the contraction index may be permuted freely as long as both
//     operands agree, so V^T is staged in that order (the other 128 threads: 8 strided loads each) and P never crosses lanes;
//   * the running maximum per query needs the four lane groups to agree (they feed one accumulator): two cross-lane maxima per block;
//     the running sum stays per lane and is reduced once, in the epilogue.
// Per 32-key block a wave issues 48 MFMAs (2 query fragments x (2 key tiles + 2 value tiles) x 6 products).
#include "igemm_dev.h"

namespace oar {
namespace k {

typedef __bf16 abf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void att_split3(const float (&f)[8], uint4& h, uint4& m, uint4& l) {
    unsigned hh[8], mm[8], ll[8];
#pragma clang loop unroll(full)
    for (int e = 0; e < 8; ++e) {
        const unsigned u = __float_as_uint(f[e]);
        const unsigned uh = u & 0xFFFF0000u;
        const float r1 = f[e] - __uint_as_float(uh);
        const unsigned um = __float_as_uint(r1) & 0xFFFF0000u;
        const float r2 = r1 - __uint_as_float(um);
        hh[e] = uh; mm[e] = um; ll[e] = __float_as_uint(r2) & 0xFFFF0000u;
    }
    h = make_uint4((hh[0] >> 16) | hh[1], (hh[2] >> 16) | hh[3], (hh[4] >> 16) | hh[5], (hh[6] >> 16) | hh[7]);
    m = make_uint4((mm[0] >> 16) | mm[1], (mm[2] >> 16) | mm[3], (mm[4] >> 16) | mm[5], (mm[6] >> 16) | mm[7]);
    l = make_uint4((ll[0] >> 16) | ll[1], (ll[2] >> 16) | ll[3], (ll[4] >> 16) | ll[5], (ll[6] >> 16) | ll[7]);
}

constexpr int kAttQ = 128;    // queries per workgroup
constexpr int kAttKB = 32;    // keys per block

// qkv: [n][T][3][heads][32] row-major; out: [n][T][heads * 32]
__global__ __launch_bounds__(256, 2) void attention_x6_kernel(const float* __restrict__ qkv, float* __restrict__ out, int T, int heads, float scale_log2e, int q_tiles) {
    __shared__ uint4 kv_lds[2][2][2][3][64];   // [stage][K | V][tile][plane][lane]
    constexpr int HD = 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 15, g = lane >> 4;
    const int qt = (int)(blockIdx.x % (unsigned)q_tiles);
    const int nh = (int)(blockIdx.x / (unsigned)q_tiles);
    const int n = nh / heads, h = nh - n * heads, dim = heads * HD;
    const long row_ld = 3L * dim;
    const float* base = qkv + (long)n * T * row_ld + h * HD;
    const int q0 = qt * kAttQ + wave * 32;

    // ---- this wave's queries: two fragments, split once
    uint4 qp[2][3];
#pragma clang loop unroll(full)
    for (int f = 0; f < 2; ++f) {
        const int q = min(q0 + f * 16 + ql, T - 1);   // clamped rows compute garbage that is never stored
        const float4 a = *reinterpret_cast<const float4*>(base + (long)q * row_ld + 8 * g);
        const float4 b = *reinterpret_cast<const float4*>(base + (long)q * row_ld + 8 * g + 4);
        const float v[8] = {a.x * scale_log2e, a.y * scale_log2e, a.z * scale_log2e, a.w * scale_log2e,
                            b.x * scale_log2e, b.y * scale_log2e, b.z * scale_log2e, b.w * scale_log2e};
        att_split3(v, qp[f][0], qp[f][1], qp[f][2]);
    }

    // ---- staging roles: threads 0..127 one (key tile, lane) slot of K, threads 128..255 one (value tile, lane) slot of V^T
    const bool is_k = tid < 128;
    const int slot = tid & 127, s_tile = slot >> 6, s_lane = slot & 63, s_i = s_lane & 15, s_g = s_lane >> 4;
    float stg[8];
    // (every load is unconditional -- out-of-range keys read key T - 1 and are zeroed afterwards: with a branch per load the compiler waited for each of the
    // eight strided value loads before it issued the next, eight memory round trips per block where the block's products take one)
    auto stage_load = [&](int kb) {
        const int k0 = kb * kAttKB;
        if (is_k) {
            const int key = k0 + s_tile * 16 + s_i;
            const float* src = base + (long)min(key, T - 1) * row_ld + dim + 8 * s_g;
            const float4 a = *reinterpret_cast<const float4*>(src);
            const float4 b = *reinterpret_cast<const float4*>(src + 4);
            const float z = key < T ? 1.f : 0.f;
            stg[0] = a.x * z; stg[1] = a.y * z; stg[2] = a.z * z; stg[3] = a.w * z; stg[4] = b.x * z; stg[5] = b.y * z; stg[6] = b.z * z; stg[7] = b.w * z;
        } else {
            float v[8];
#pragma clang loop unroll(full)
            for (int j = 0; j < 8; ++j) {
                const int key = k0 + (j < 4 ? 4 * s_g + j : 16 + 4 * s_g + (j - 4));
                v[j] = base[(long)min(key, T - 1) * row_ld + 2 * dim + s_tile * 16 + s_i];
            }
#pragma clang loop unroll(full)
            for (int j = 0; j < 8; ++j) {
                const int key = k0 + (j < 4 ? 4 * s_g + j : 16 + 4 * s_g + (j - 4));
                stg[j] = key < T ? v[j] : 0.f;
            }
        }
    };
    auto stage_commit = [&](int st) {
        uint4 ph, pm, pl;
        att_split3(stg, ph, pm, pl);
        uint4* dst = &kv_lds[st][is_k ? 0 : 1][s_tile][0][s_lane];
        dst[0] = ph; dst[64] = pm; dst[128] = pl;
    };

    f32x4 o[2][2];   // [query fragment][value tile]: lane (q, g) holds components 16 dt + 4 g + r of query q
    float m_run[2], l_run[2];
#pragma clang loop unroll(full)
    for (int f = 0; f < 2; ++f) {
        o[f][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; o[f][1] = o[f][0];
        m_run[f] = -INFINITY; l_run[f] = 0.f;
    }
    constexpr int WP[6] = {1, 2, 0, 1, 0, 0};   // (A plane, B plane) = mm, lh, hl, mh, hm, hh: smallest terms first
    constexpr int XP[6] = {1, 0, 2, 0, 1, 0};

    const int n_blocks = (T + kAttKB - 1) / kAttKB;
    stage_load(0);
    stage_commit(0);
    __syncthreads();
    // (requesting the blocks TWO iterations ahead -- a second staging set, 182 registers -- was measured slower: 369 against 312 us per launch; the third wave
    // per SIMD that 161 registers allow hides more than the longer prefetch distance does)
    for (int kb = 0; kb < n_blocks; ++kb) {
        const int st = kb & 1;
        const bool more = kb + 1 < n_blocks;
        if (more) stage_load(kb + 1);
        // ---- S^T = K Q^T (log2 domain)
        uint4 ka[2][3];
#pragma clang loop unroll(full)
        for (int t = 0; t < 2; ++t)
#pragma clang loop unroll(full)
            for (int pq = 0; pq < 3; ++pq) ka[t][pq] = kv_lds[st][0][t][pq][lane];
        f32x4 s[2][2];
#pragma clang loop unroll(full)
        for (int f = 0; f < 2; ++f)
#pragma clang loop unroll(full)
            for (int t = 0; t < 2; ++t) s[f][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma clang loop unroll(full)
        for (int pr = 0; pr < 6; ++pr)
#pragma clang loop unroll(full)
            for (int f = 0; f < 2; ++f)
#pragma clang loop unroll(full)
                for (int t = 0; t < 2; ++t)
                    s[f][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(abf16x8, ka[t][WP[pr]]), __builtin_bit_cast(abf16x8, qp[f][XP[pr]]), s[f][t], 0, 0, 0);
        if (!more && (T & (kAttKB - 1))) {   // key tail of the last block
            const int valid = T - kb * kAttKB;
#pragma clang loop unroll(full)
            for (int f = 0; f < 2; ++f)
#pragma clang loop unroll(full)
                for (int t = 0; t < 2; ++t)
#pragma clang loop unroll(full)
                    for (int r = 0; r < 4; ++r)
                        if (t * 16 + 4 * g + r >= valid) s[f][t][r] = -INFINITY;
        }
        // ---- online soft-max; the probabilities become the B operand of the second product in place
        uint4 pp[2][3];
#pragma clang loop unroll(full)
        for (int f = 0; f < 2; ++f) {
            float mx = fmaxf(fmaxf(fmaxf(s[f][0][0], s[f][0][1]), fmaxf(s[f][0][2], s[f][0][3])), fmaxf(fmaxf(s[f][1][0], s[f][1][1]), fmaxf(s[f][1][2], s[f][1][3])));
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            const float m_new = fmaxf(m_run[f], mx);
            const float alpha = __builtin_amdgcn_exp2f(m_run[f] - m_new);   // first block: exp2(-inf) = 0
            m_run[f] = m_new;
            float p[8];
            float ls = 0.f;
#pragma clang loop unroll(full)
            for (int j = 0; j < 8; ++j) { p[j] = __builtin_amdgcn_exp2f(s[f][j >> 2][j & 3] - m_new); ls += p[j]; }
            l_run[f] = l_run[f] * alpha + ls;
#pragma clang loop unroll(full)
            for (int dt = 0; dt < 2; ++dt)
#pragma clang loop unroll(full)
                for (int r = 0; r < 4; ++r) o[f][dt][r] *= alpha;
            att_split3(p, pp[f][0], pp[f][1], pp[f][2]);
        }
        // ---- O^T += V^T P^T
        uint4 va[2][3];
#pragma clang loop unroll(full)
        for (int t = 0; t < 2; ++t)
#pragma clang loop unroll(full)
            for (int pq = 0; pq < 3; ++pq) va[t][pq] = kv_lds[st][1][t][pq][lane];
#pragma clang loop unroll(full)
        for (int pr = 0; pr < 6; ++pr)
#pragma clang loop unroll(full)
            for (int f = 0; f < 2; ++f)
#pragma clang loop unroll(full)
                for (int dt = 0; dt < 2; ++dt)
                    o[f][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(abf16x8, va[dt][WP[pr]]), __builtin_bit_cast(abf16x8, pp[f][XP[pr]]), o[f][dt], 0, 0, 0);
        if (more) {
            stage_commit(st ^ 1);   // (stage st^1 was last read in iteration kb - 1, before that iteration's barrier)
            __syncthreads();
        }
    }
    // ---- epilogue: total sum over the four lane groups, then one float4 store per (query, value tile)
#pragma clang loop unroll(full)
    for (int f = 0; f < 2; ++f) {
        float l = l_run[f];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const int q = q0 + f * 16 + ql;
        if (q < T) {
            float* y = out + ((long)n * T + q) * dim + h * HD + 4 * g;
#pragma clang loop unroll(full)
            for (int dt = 0; dt < 2; ++dt)
                *reinterpret_cast<float4*>(y + 16 * dt) = make_float4(o[f][dt][0] / l, o[f][dt][1] / l, o[f][dt][2] / l, o[f][dt][3] / l);
        }
    }
}

bool attention_x6_supported(int T, int heads, int hd) { return hd == 32 && T >= 1 && heads >= 1; }

void attention_x6(hipStream_t s, const float* qkv, float* out, int n, int T, int heads, int hd, float scale) {
    OAR_CHECK(attention_x6_supported(T, heads, hd), OAR_INTERNAL, "attention_x6: head dim must be 32");
    const int q_tiles = (T + kAttQ - 1) / kAttQ;
    const long grid = (long)n * heads * q_tiles;
    OAR_CHECK(grid < (1L << 31), OAR_UNSUPPORTED_OP, "attention_x6: too many (crop, head, query tile) workgroups for one launch");
    const double nh = (double)n * heads;
    ProfScope ps(s, "attention_x6", 4.0 * nh * T * 4.0 * hd, 4.0 * nh * T * T * hd);
    hipLaunchKernelGGL(attention_x6_kernel, dim3((unsigned)grid), dim3(256), 0, s, qkv, out, T, heads, scale * 1.4426950408889634f, q_tiles);
}

}  // namespace k
}  // namespace oar
