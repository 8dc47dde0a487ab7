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
// PF (round 6): pixel fragments per wave tile.  At PF = 1 (16 waves per CU, 128 registers each) every weight fragment read from LDS -- three ds_read_b128 -- feeds six MFMAs:
// four SIMDs multiplying at full rate would ask the LDS for exactly its 128 B / clock, and the kernel sits at ~35 % of the matrix pipe.  PF = 2 (8 waves per CU, 32-pixel
// tiles, 64 accumulators) halves the LDS traffic per MFMA -- the output-stationary kernel's ratio, without its per-chunk weight staging and barrier.
template <int NT, bool CTC = false, bool SE = false, int PF = 1>   // CTC: the CTC-head variant (softmax partials instead of logits), its own instantiation so that
__global__ __launch_bounds__(PF == 2 ? 512 : 1024) void conv_igemm_ws_x6_kernel(IgemmWsX6P q) {   // the plain kernels keep their register budget
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
        se_first = (int)((wt_begin * (16 * PF)) / p.se_hw);
        const int se_last = (int)(min(p.M - 1, (wt_begin + wt_count) * (16 * PF) - 1) / p.se_hw);
        if (se_last - se_first + 1 > q.se_cap) __builtin_trap();   // the host's bound (ws_x6_se_rows) undercounts this workgroup's images: fail loudly, never read unstaged gates
        const int cnt = (se_last - se_first + 1) * p.K;
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) se_lds[i] = p.se[(long)se_first * p.K + i];   // (visible after the barrier below)
    }
    constexpr int NL = 2 * PF;   // vector-memory loads per chunk: two float4 per pixel fragment
    auto issue = [](f32x4& dst, const float* src) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src)); };
    struct Stage { f32x4 a[PF], b[PF]; };
    auto tie = [](Stage& st) {   // (an empty asm: the registers of a stage are live here -- keeps the compiler from moving / re-materialising the inline-asm loads)
        if constexpr (PF == 1) asm volatile("" : "+v"(st.a[0]), "+v"(st.b[0]));
        else asm volatile("" : "+v"(st.a[0]), "+v"(st.b[0]), "+v"(st.a[1]), "+v"(st.b[1]));
    };
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
        long m0 = (wt_begin + u) * (16 * PF), m0n = m0;
        long cur_row[PF], nxt_row[PF];
#pragma clang loop unroll(full)
        for (int pf = 0; pf < PF; ++pf) { cur_row[pf] = min(m0 + pf * 16 + pl_, p.M - 1) * (long)p.Cin; nxt_row[pf] = cur_row[pf]; }
        Stage s0, s1, s2;
        auto stage = [&](auto Ic) -> Stage& {
            constexpr int I = decltype(Ic)::value % 3;
            if constexpr (I == 0) return s0; else if constexpr (I == 1) return s1; else return s2;
        };
#pragma clang loop unroll(full)
        for (int pf = 0; pf < PF; ++pf) { const float* a0 = x_addr(cur_row[pf], 0); issue(s0.a[pf], a0); issue(s0.b[pf], a0 + 4); }
#pragma clang loop unroll(full)
        for (int pf = 0; pf < PF; ++pf) { const float* a1 = x_addr(cur_row[pf], min(1, p.KC - 1)); issue(s1.a[pf], a1); issue(s1.b[pf], a1 + 4); }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tie(s0); tie(s1);

        const uint4* wl = wx_lds + lane;
        auto run_tile = [&]() {
            const float* se_row[PF];
#pragma clang loop unroll(full)
            for (int pf = 0; pf < PF; ++pf) se_row[pf] = SE ? se_lds + (long)((int)(min(m0 + pf * 16 + pl_, p.M - 1) / p.se_hw) - se_first) * p.K : nullptr;
            f32x4 acc[NT][PF];
#pragma clang loop unroll(full)
            for (int nf = 0; nf < NT; ++nf) {
                const float4 bq = *reinterpret_cast<const float4*>(lds_bias + nf * 16 + g * 4);
#pragma clang loop unroll(full)
                for (int pf = 0; pf < PF; ++pf) acc[nf][pf] = (f32x4){bq.x, bq.y, bq.z, bq.w};
            }
            uint4 wr[2][3];   // 2-slot ring of weight fragments (3 planes each)
#pragma clang loop unroll(full)
            for (int pq = 0; pq < 3; ++pq) wr[0][pq] = wl[pq * 64];
            auto chunk = [&](int kc, auto SIc, auto WAITc, auto R0c) {
                constexpr int SI = decltype(SIc)::value, r0 = decltype(R0c)::value;
                constexpr bool WAIT = decltype(WAITc)::value;
                Stage& cs = stage(std::integral_constant<int, SI>{});
                Stage& ns = stage(std::integral_constant<int, SI + 2>{});
                if (WAIT) { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NL) : "memory"); tie(cs); }
                {
                    const int k2 = kc + 2;
                    const bool in_cur = k2 < p.KC;
#pragma clang loop unroll(full)
                    for (int pf = 0; pf < PF; ++pf) {
                        const float* ap = x_addr(in_cur ? cur_row[pf] : nxt_row[pf], in_cur ? k2 : k2 - p.KC);
                        issue(ns.a[pf], ap); issue(ns.b[pf], ap + 4);
                    }
                }
                uint4 xs[PF][3];
#pragma clang loop unroll(full)
                for (int pf = 0; pf < PF; ++pf) {
                    if (SE) {
                        const float* sp = se_row[pf] + min(kc * 32 + 8 * g, p.K - 8);
                        const f32x4 ga = *reinterpret_cast<const f32x4*>(sp), gb = *reinterpret_cast<const f32x4*>(sp + 4);
                        split3(cs.a[pf] * ga, cs.b[pf] * gb, xs[pf][0], xs[pf][1], xs[pf][2]);
                    } else {
                        split3(cs.a[pf], cs.b[pf], xs[pf][0], xs[pf][1], xs[pf][2]);
                    }
                }
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
#pragma clang loop unroll(full)
                        for (int pf = 0; pf < PF; ++pf)   // (PF = 2: the two pixel fragments alternate -- no back-to-back MFMAs on one accumulator)
                            acc[nf][pf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[WP[t]]), __builtin_bit_cast(bf16x8, xs[pf][XP[t]]), acc[nf][pf], 0, 0, 0);
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
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            tie(s0); tie(s1); tie(s2);
            if constexpr (CTC) { static_assert(!CTC || PF == 1, "the CTC epilogue is written for one pixel fragment"); igemm_ctc_epilogue<NT>(p, acc, m0, pl_, g, nf0, ntile, q.ny); }   // tiles of 8 fragments (ctc_tiles)
            else igemm_epilogue<NT, PF, true>(p, acc, m0, pl_, g, nf0, false);
        };

        for (;;) {
            const long un = grab();
            const bool has_next = un < wt_count;
            m0n = (wt_begin + (has_next ? un : u)) * (16 * PF);
#pragma clang loop unroll(full)
            for (int pf = 0; pf < PF; ++pf) nxt_row[pf] = min(m0n + pf * 16 + pl_, p.M - 1) * (long)p.Cin;
            run_tile();
            if (!has_next) break;
            const int rot = p.KC % 3;   // chunk 0 / 1 of the next tile sit in stages KC%3 / (KC+1)%3: rotate them to 0 / 1
            if (rot == 1) { s0 = s1; s1 = s2; }
            else if (rot == 2) { s1 = s0; s0 = s2; }
            m0 = m0n; u = un;
#pragma clang loop unroll(full)
            for (int pf = 0; pf < PF; ++pf) cur_row[pf] = nxt_row[pf];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tie(s0); tie(s1); tie(s2);
    }
}

// gate rows (images) one workgroup's tile range can touch, for the launch geometry below (tile = 16 pf pixels)
static int ws_x6_se_rows_pf(long M, int ny, int hw, int pf) {
    const int per_xcd = 32;
    const long wt_total = (M + 16 * pf - 1) / (16 * pf), wt_per_xcd = (wt_total + 7) / 8;
    const int groups = ny < per_xcd ? ny : per_xcd, team_min = per_xcd / groups;   // the smallest team has floor(32 / groups) members
    const long share = (wt_per_xcd + team_min - 1) / team_min;
    return (int)((share * 16 * pf + hw - 1) / hw) + 1;
}
// pixel fragments per wave tile for a launch of M pixels: two once the launch has ~1.5 rounds of 32-pixel tiles for the chip's 2048 resident waves (OAR_WS_X6_PF=1 / 2 forces)
// Measured (profiles/r6/ws_x6_pf_ab.txt): +3 ... +7 % on layers without an expensive activation (the SE blocks' gated 1x1 convolutions: 0.99 -> 0.88 ms per step; SVTRv2's
// projections 196 -> 204 TFLOP/s) -- the kernel was not as LDS-bound as the arithmetic above says -- and MINUS 13 ... 30 % behind a GELU epilogue (256 -> 1024: 178 -> 154
// TFLOP/s), which eight waves per CU hide worse than sixteen: those layers keep PF = 1.
static int ws_x6_pf(long M, bool ctc, int act) {
    static const int force = [] { const char* e = getenv("OAR_WS_X6_PF"); return e ? atoi(e) : 0; }();
    if (ctc) return 1;
    if (force == 1 || force == 2) return force;
    const bool cheap = act == ACT_NONE || act == ACT_RELU || act == ACT_HSWISH || act == ACT_HSIGMOID || act == ACT_LEAKY || act == ACT_CLIP;
    return (cheap && M >= 98304) ? 2 : 1;
}
int ws_x6_se_rows(long M, int ny, int hw) { return std::max(ws_x6_se_rows_pf(M, ny, hw, 1), ws_x6_se_rows_pf(M, ny, hw, 2)); }   // (what conv_igemm_se_ok budgets LDS with: the larger of the two tilings)

template <int NT, bool CTC = false, bool SE = false, int PF = 1>
static void launch_ws_x6(hipStream_t s, const IgemmP& p, int ny, size_t lds) {
    OAR_MAX_LDS_ONCE((conv_igemm_ws_x6_kernel<NT, CTC, SE, PF>), 160 * 1024);
    IgemmWsX6P q;
    q.g = p; q.ny = ny;
    const int per_xcd = 32;
    q.wt_total = (p.M + 16 * PF - 1) / (16 * PF);
    q.wt_per_xcd = (q.wt_total + 7) / 8;
    q.groups = ny < per_xcd ? ny : per_xcd;
    q.se_cap = 0;
    if (SE) {
        q.se_cap = ws_x6_se_rows_pf(p.M, ny, p.se_hw, PF);
        lds += (size_t)q.se_cap * p.K * 4;
        OAR_CHECK(lds <= 160 * 1024, OAR_INTERNAL, "conv_igemm_ws_x6: the gate rows do not fit LDS (conv_igemm_se_ok should have said no)");
    }
    hipLaunchKernelGGL((conv_igemm_ws_x6_kernel<NT, CTC, SE, PF>), dim3(per_xcd * 8), dim3(PF == 2 ? 512 : 1024), lds, s, q);
}

void conv_igemm_ws_x6(hipStream_t s, const IgemmP& p, int ws_nt, int ny, size_t lds) {
    if (p.ctc_part) { launch_ws_x6<8, true>(s, p, ny, lds); return; }   // conv_igemm passes ws_nt = 8 for CTC heads
    const bool two = ws_x6_pf(p.M, false, p.act) == 2;
    if (p.se) {
        switch (ws_nt) {
            case 8: two ? launch_ws_x6<8, false, true, 2>(s, p, ny, lds) : launch_ws_x6<8, false, true>(s, p, ny, lds); break;
            case 6: two ? launch_ws_x6<6, false, true, 2>(s, p, ny, lds) : launch_ws_x6<6, false, true>(s, p, ny, lds); break;
            default: two ? launch_ws_x6<4, false, true, 2>(s, p, ny, lds) : launch_ws_x6<4, false, true>(s, p, ny, lds); break;
        }
        return;
    }
    switch (ws_nt) {
        case 8: two ? launch_ws_x6<8, false, false, 2>(s, p, ny, lds) : launch_ws_x6<8>(s, p, ny, lds); break;
        case 6: two ? launch_ws_x6<6, false, false, 2>(s, p, ny, lds) : launch_ws_x6<6>(s, p, ny, lds); break;
        default: two ? launch_ws_x6<4, false, false, 2>(s, p, ny, lds) : launch_ws_x6<4>(s, p, ny, lds); break;
    }
}

}  // namespace k
}  // namespace oar
