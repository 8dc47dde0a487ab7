// igemm_ws_x6.hip -- weight-stationary 1x1 conv / Linear on the bf16 matrix pipe with f32-equivalent accuracy.
//
// Same workgroup organisation, tile queue and wait protocol as conv_igemm_ws_kernel (igemm_ws.inc); the arithmetic is the
// exact 3-way bf16 split (igemm.hip, "igemm x6"): every f32 operand x = h + m + l with h, m, l the successive 8-bit
// significand slices (truncation, so the split is exact), and the product accumulates the six terms mm, lh, hl, mh, hm,
// hh in f32 -- the dropped terms are <= 2^-24 relative, the same error class as an f32 FMA chain.  One 32-deep K step of
// a (16 cout x 16 pixel) fragment costs 6 x 16 cycles of v_mfma_f32_16x16x32_bf16 instead of 8 x 32 cycles of
// v_mfma_f32_16x16x4_f32.  Weights are pre-split on the host (3 planes of 8 bf16 per lane and K step), resident in LDS;
// pixels are loaded as 32 contiguous bytes per lane (the four g-lanes of a pixel read one whole 128-byte line) and split
// in registers.
#include "igemm_dev.h"

namespace oar {
namespace k {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct IgemmWsX6P {
    IgemmP g;           // g.KC = ceil(K / 32)
    int ny;             // cout tiles
    int groups;         // teams per XCD; team t handles cout tiles t, t+groups, ...
    long wt_per_xcd;    // wave tiles (16 pixels) per XCD band
    long wt_total;      // ceil(M / 16)
    int se_cap;         // SE variant: gate rows (images) the LDS region after the counter can hold
};

__device__ __forceinline__ void split3(const f32x4& a, const f32x4& b, uint4& h, uint4& m, uint4& l) {
    const float f[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
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
    // bf16 element e = upper half of piece e; two per dword, element 2i in the low half
    h = make_uint4((hh[0] >> 16) | hh[1], (hh[2] >> 16) | hh[3], (hh[4] >> 16) | hh[5], (hh[6] >> 16) | hh[7]);
    m = make_uint4((mm[0] >> 16) | mm[1], (mm[2] >> 16) | mm[3], (mm[4] >> 16) | mm[5], (mm[6] >> 16) | mm[7]);
    l = make_uint4((ll[0] >> 16) | ll[1], (ll[2] >> 16) | ll[3], (ll[4] >> 16) | ll[5], (ll[6] >> 16) | ll[7]);
}

// SE: the input is multiplied by a squeeze-excite gate [image][K] as it is loaded (the Mul between the gate and this conv never runs:
// one read + one write of the whole feature map less).  The gate rows of the images this workgroup's tile range touches sit in LDS
// behind the counter; x * gate is the same v_mul_f32 the stand-alone Mul would have done, so the result is bit-identical to it.
template <int NT, bool CTC = false, bool SE = false>   // CTC: the CTC-head variant (softmax partials instead of logits), its own instantiation so that
__global__ __launch_bounds__(1024) void conv_igemm_ws_x6_kernel(IgemmWsX6P q) {   // the plain kernels keep their register budget
    extern __shared__ uint4 wx_lds[];   // [kc][nf][plane][lane] weights | [nf][16] bias | counter
    const IgemmP& p = q.g;
    const int lane = threadIdx.x & 63;
    const int pl_ = lane & 15, g = lane >> 4;
    const int xcd = (int)(blockIdx.x & 7);
    const int j = (int)(blockIdx.x >> 3);
    const int per_xcd = (int)(gridDim.x >> 3);
    const int team = j % q.groups, member = j / q.groups, team_size = (per_xcd - team + q.groups - 1) / q.groups;
    float* lds_bias = reinterpret_cast<float*>(wx_lds + (long)NT * p.KC * 3 * 64);
    unsigned* ws_ctr = reinterpret_cast<unsigned*>(lds_bias + NT * 16);
    float* se_lds = reinterpret_cast<float*>(ws_ctr + 4);   // SE: [image - se_first][K]
    long wt_begin, wt_count;
    {
        const long x0 = (long)xcd * q.wt_per_xcd, x1 = min(q.wt_total, x0 + q.wt_per_xcd);
        const long n_x = max(0L, x1 - x0), share = (n_x + team_size - 1) / team_size;
        wt_begin = x0 + (long)member * share;
        wt_count = max(0L, min(share, x1 - wt_begin));
    }
    int se_first = 0;
    if (SE && wt_count > 0) {
        se_first = (int)((wt_begin * 16) / p.se_hw);
        const int se_last = (int)(min(p.M - 1, (wt_begin + wt_count) * 16 - 1) / p.se_hw);
        if (se_last - se_first + 1 > q.se_cap) __builtin_trap();   // the host's bound (ws_x6_se_rows) undercounts this workgroup's images: fail loudly, never read unstaged gates
        const int cnt = (se_last - se_first + 1) * p.K;
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) se_lds[i] = p.se[(long)se_first * p.K + i];   // (visible after the barrier below)
    }
    constexpr int NL = 2;   // vector-memory loads per chunk: two float4 of one pixel fragment
    auto issue = [](f32x4& dst, const float* src) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src)); };
    struct Stage { f32x4 a, b; };
    auto x_addr = [&](long row_base, int kc) -> const float* {
        return p.x + row_base + min(kc * 32 + 8 * g, p.K - 8);   // zero-padded K tail of W: re-read a valid group
    };
    auto grab = [&]() -> long {
        unsigned u = 0;
        if (lane == 0) u = atomicAdd(ws_ctr, 1u);
        return (long)(unsigned)__builtin_amdgcn_readfirstlane((int)u);
    };

    for (int ntile = team; ntile < q.ny; ntile += q.groups) {
        const int nf0 = ntile * NT;
        __syncthreads();
        if (threadIdx.x == 0) *ws_ctr = 0;
        if (threadIdx.x < NT * 16) {
            const int c = nf0 * 16 + (int)threadIdx.x;
            lds_bias[threadIdx.x] = (p.bias && c < p.gemm_cout) ? p.bias[c] : 0.f;
        }
        {   // global fragment order [nf][kc][plane][lane] -> LDS [kc][nf][plane][lane]
            const uint4* wsrc = reinterpret_cast<const uint4*>(p.w) + (long)nf0 * p.KC * 3 * 64;
            const int total = NT * p.KC * 3 * 64;
            const int stride = (int)blockDim.x;
            int i0 = threadIdx.x;
            for (; i0 + 7 * stride < total; i0 += 8 * stride) {
                uint4 v[8];
#pragma clang loop unroll(full)
                for (int u = 0; u < 8; ++u) v[u] = wsrc[i0 + u * stride];
#pragma clang loop unroll(full)
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + u * stride, r = i % 192, f = i / 192;   // f = nf * KC + kc, r = plane * 64 + lane
                    const int nf = f / p.KC, kc = f - nf * p.KC;
                    wx_lds[(kc * NT + nf) * 192 + r] = v[u];
                }
            }
            for (; i0 < total; i0 += stride) {
                const int r = i0 % 192, f = i0 / 192;
                const int nf = f / p.KC, kc = f - nf * p.KC;
                wx_lds[(kc * NT + nf) * 192 + r] = wsrc[i0];
            }
        }
        __syncthreads();

        long u = grab();
        if (u >= wt_count) continue;
        long m0 = (wt_begin + u) * 16, m0n = m0;
        long cur_row = min(m0 + pl_, p.M - 1) * (long)p.Cin, nxt_row = cur_row;
        Stage s0, s1, s2;
        auto stage = [&](auto Ic) -> Stage& {
            constexpr int I = decltype(Ic)::value % 3;
            if constexpr (I == 0) return s0; else if constexpr (I == 1) return s1; else return s2;
        };
        { const float* a0 = x_addr(cur_row, 0); issue(s0.a, a0); issue(s0.b, a0 + 4); }
        { const float* a1 = x_addr(cur_row, min(1, p.KC - 1)); issue(s1.a, a1); issue(s1.b, a1 + 4); }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(s0.a), "+v"(s0.b), "+v"(s1.a), "+v"(s1.b));

        const uint4* wl = wx_lds + lane;
        auto run_tile = [&]() {
            const float* se_row = nullptr;
            if (SE) se_row = se_lds + (long)((int)(min(m0 + pl_, p.M - 1) / p.se_hw) - se_first) * p.K;
            f32x4 acc[NT][1];
#pragma clang loop unroll(full)
            for (int nf = 0; nf < NT; ++nf) {
                const float4 bq = *reinterpret_cast<const float4*>(lds_bias + nf * 16 + g * 4);
                acc[nf][0] = (f32x4){bq.x, bq.y, bq.z, bq.w};
            }
            uint4 wr[2][3];   // 2-slot ring of weight fragments (3 planes each)
#pragma clang loop unroll(full)
            for (int pq = 0; pq < 3; ++pq) wr[0][pq] = wl[pq * 64];
            auto chunk = [&](int kc, auto SIc, auto WAITc, auto R0c) {
                constexpr int SI = decltype(SIc)::value, r0 = decltype(R0c)::value;
                constexpr bool WAIT = decltype(WAITc)::value;
                Stage& cs = stage(std::integral_constant<int, SI>{});
                Stage& ns = stage(std::integral_constant<int, SI + 2>{});
                if (WAIT) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(cs.a), "+v"(cs.b) : "n"(NL));
                {
                    const int k2 = kc + 2;
                    const bool in_cur = k2 < p.KC;
                    const float* ap = x_addr(in_cur ? cur_row : nxt_row, in_cur ? k2 : k2 - p.KC);
                    issue(ns.a, ap); issue(ns.b, ap + 4);
                }
                uint4 xs[3];
                if (SE) {
                    const float* sp = se_row + min(kc * 32 + 8 * g, p.K - 8);
                    const f32x4 ga = *reinterpret_cast<const f32x4*>(sp), gb = *reinterpret_cast<const f32x4*>(sp + 4);
                    split3(cs.a * ga, cs.b * gb, xs[0], xs[1], xs[2]);
                } else
                split3(cs.a, cs.b, xs[0], xs[1], xs[2]);
                const int k1 = min(kc + 1, p.KC - 1);
#pragma clang loop unroll(full)
                for (int nf = 0; nf < NT; ++nf) {
                    {   // next fragment of the flattened (kc, nf) order into the other ring slot
                        const int nf2 = (nf + 1) % NT;
                        const int kq = (nf + 1 < NT) ? kc : k1;
                        __builtin_amdgcn_sched_barrier(0);
#pragma clang loop unroll(full)
                        for (int pq = 0; pq < 3; ++pq) wr[(r0 + nf + 1) & 1][pq] = wl[(((long)kq * NT + nf2) * 3 + pq) * 64];
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    const uint4* w = wr[(r0 + nf) & 1];
                    constexpr int WP[6] = {1, 2, 0, 1, 0, 0};   // (w plane, x plane) = mm, lh, hl, mh, hm, hh: smallest terms first
                    constexpr int XP[6] = {1, 0, 2, 0, 1, 0};
#pragma clang loop unroll(full)
                    for (int t = 0; t < 6; ++t)
                        acc[nf][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[WP[t]]), __builtin_bit_cast(bf16x8, xs[XP[t]]), acc[nf][0], 0, 0, 0);
                }
            };
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>;
            using Wy = std::true_type;
            using Wn = std::false_type;
            // ring slot of a chunk's fragment 0 = (kc * NT) & 1: NT even -> always 0; NT odd -> kc & 1 (the trip below
            // advances kc by 6 so that both the 3 pixel stages and the 2 ring slots return to their positions)
            constexpr int O = NT & 1;
            chunk(0, I0{}, Wn{}, I0{});
            chunk(1, I1{}, Wn{}, std::integral_constant<int, O>{});
            int kc = 2;
            for (; kc + 5 < p.KC; kc += 6) {
                chunk(kc, I2{}, Wy{}, I0{}); chunk(kc + 1, I0{}, Wy{}, std::integral_constant<int, O>{}); chunk(kc + 2, I1{}, Wy{}, I0{});
                chunk(kc + 3, I2{}, Wy{}, std::integral_constant<int, O>{}); chunk(kc + 4, I0{}, Wy{}, I0{}); chunk(kc + 5, I1{}, Wy{}, std::integral_constant<int, O>{});
            }
            if (kc < p.KC) chunk(kc, I2{}, Wy{}, I0{});
            if (kc + 1 < p.KC) chunk(kc + 1, I0{}, Wy{}, std::integral_constant<int, O>{});
            if (kc + 2 < p.KC) chunk(kc + 2, I1{}, Wy{}, I0{});
            if (kc + 3 < p.KC) chunk(kc + 3, I2{}, Wy{}, std::integral_constant<int, O>{});
            if (kc + 4 < p.KC) chunk(kc + 4, I0{}, Wy{}, I0{});
            // land the two chunks requested for the next tile BEFORE any store is issued
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(s0.a), "+v"(s0.b), "+v"(s1.a), "+v"(s1.b), "+v"(s2.a), "+v"(s2.b));
            if constexpr (CTC) igemm_ctc_epilogue<NT>(p, acc, m0, pl_, g, nf0, ntile, q.ny);   // tiles of 8 fragments (ctc_tiles)
            else igemm_epilogue<NT, 1, true>(p, acc, m0, pl_, g, nf0, false);
        };

        for (;;) {
            const long un = grab();
            const bool has_next = un < wt_count;
            m0n = (wt_begin + (has_next ? un : u)) * 16;
            nxt_row = min(m0n + pl_, p.M - 1) * (long)p.Cin;
            run_tile();
            if (!has_next) break;
            const int rot = p.KC % 3;   // chunk 0 / 1 of the next tile sit in stages KC%3 / (KC+1)%3: rotate them to 0 / 1
            if (rot == 1) { s0 = s1; s1 = s2; }
            else if (rot == 2) { s1 = s0; s0 = s2; }
            m0 = m0n; cur_row = nxt_row; u = un;
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(s0.a), "+v"(s0.b), "+v"(s1.a), "+v"(s1.b), "+v"(s2.a), "+v"(s2.b));
    }
}

// gate rows (images) one workgroup's tile range can touch, for the launch geometry below
int ws_x6_se_rows(long M, int ny, int hw) {
    const int per_xcd = 32;
    const long wt_total = (M + 15) / 16, wt_per_xcd = (wt_total + 7) / 8;
    const int groups = ny < per_xcd ? ny : per_xcd, team_min = per_xcd / groups;   // the smallest team has floor(32 / groups) members
    const long share = (wt_per_xcd + team_min - 1) / team_min;
    return (int)((share * 16 + hw - 1) / hw) + 1;
}

template <int NT, bool CTC = false, bool SE = false>
static void launch_ws_x6(hipStream_t s, const IgemmP& p, int ny, size_t lds) {
    OAR_MAX_LDS_ONCE((conv_igemm_ws_x6_kernel<NT, CTC, SE>), 160 * 1024);
    IgemmWsX6P q;
    q.g = p; q.ny = ny;
    const int per_xcd = 32;
    q.wt_total = (p.M + 15) / 16;
    q.wt_per_xcd = (q.wt_total + 7) / 8;
    q.groups = ny < per_xcd ? ny : per_xcd;
    q.se_cap = 0;
    if (SE) {
        q.se_cap = ws_x6_se_rows(p.M, ny, p.se_hw);
        lds += (size_t)q.se_cap * p.K * 4;
        OAR_CHECK(lds <= 160 * 1024, OAR_INTERNAL, "conv_igemm_ws_x6: the gate rows do not fit LDS (conv_igemm_se_ok should have said no)");
    }
    hipLaunchKernelGGL((conv_igemm_ws_x6_kernel<NT, CTC, SE>), dim3(per_xcd * 8), dim3(1024), lds, s, q);
}

void conv_igemm_ws_x6(hipStream_t s, const IgemmP& p, int ws_nt, int ny, size_t lds) {
    if (p.ctc_part) { launch_ws_x6<8, true>(s, p, ny, lds); return; }   // conv_igemm passes ws_nt = 8 for CTC heads
    if (p.se) {
        switch (ws_nt) {
            case 8: launch_ws_x6<8, false, true>(s, p, ny, lds); break;
            case 6: launch_ws_x6<6, false, true>(s, p, ny, lds); break;
            default: launch_ws_x6<4, false, true>(s, p, ny, lds); break;
        }
        return;
    }
    switch (ws_nt) {
        case 8: launch_ws_x6<8>(s, p, ny, lds); break;
        case 6: launch_ws_x6<6>(s, p, ny, lds); break;
        default: launch_ws_x6<4>(s, p, ny, lds); break;
    }
}

}  // namespace k
}  // namespace oar
