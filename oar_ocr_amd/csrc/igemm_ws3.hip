// igemm_ws3.hip -- weight-stationary implicit GEMM for 3x3 / stride 1 / pad 1 "same" convolutions (the DB head's
// 64 -> 16 conv at quarter resolution: 1.7 ms of the step on the generic path, whose pixel loads cover the 9 taps
// separately -- the L1/L2 path carried 9x the input, HBM 3x).
//
// For one kernel row kh the three horizontal taps of a run of consecutive output pixels read the SAME input pixels
// shifted by one: only the centre tap's fragments (plus one pixel of halo on each side of the wave tile) are loaded;
// the kw = 0 / kw = 2 operands are produced in registers with DPP row shifts (a pixel fragment is one 16-lane row per
// channel quad g, so row_shr:1 / row_shl:1 move "pixel p-1 / p+1" into lane p; the row's first / last lane takes its
// value from the neighbouring fragment via row_ror).  Zero padding is a select on the shifted registers.  Everything
// else -- LDS-resident weights and bias, persistent 16-wave workgroups with an LDS tile queue, inline-asm loads with
// hand-counted waits, next tile's first steps landed before the stores -- is conv_igemm_ws_kernel's scheme
// (igemm_ws.inc).  A pipeline step is (kh, 16-channel chunk): PF centre loads + 2 halo loads feed 3*PF*NT*4 MFMAs.
#include "igemm_dev.h"

namespace oar {
namespace k {

struct IgemmWs3P {
    IgemmP g;           // g.KC = K/16 = 9 * Cin/16 weight chunks in (kh, kw, ci) order
    int ny;
    int groups;
    long wt_per_xcd;
    long wt_total;
};

template <class CTRL>
__device__ __forceinline__ f32x4 dpp4(const f32x4& old, const f32x4& src, CTRL) {
    f32x4 r;
#pragma clang loop unroll(full)
    for (int e = 0; e < 4; ++e)
        r[e] = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old[e]), __float_as_int(src[e]), CTRL::value, 0xF, 0xF, false));
    return r;
}

template <int NT, int PF>
__global__ __launch_bounds__(1024) void conv_igemm_ws3_kernel(IgemmWs3P q) {
    extern __shared__ float4 ws_lds[];   // [kc][nf][lane] weights | [nf][16] bias | counter
    const IgemmP& p = q.g;
    const int lane = threadIdx.x & 63;
    const int pl_ = lane & 15, g = lane >> 4;
    const int xcd = (int)(blockIdx.x & 7);
    const int j = (int)(blockIdx.x >> 3);
    const int per_xcd = (int)(gridDim.x >> 3);
    const int team = j % q.groups, member = j / q.groups, team_size = (per_xcd - team + q.groups - 1) / q.groups;
    float* lds_bias = reinterpret_cast<float*>(ws_lds + (long)NT * p.KC * 64);
    unsigned* ws_ctr = reinterpret_cast<unsigned*>(lds_bias + NT * 16);
    // The XCD's band of tiles is dealt CYCLICALLY to its workgroups (tile = band start + u * team_size + member): at any
    // moment the ~512 tiles in flight on the XCD are one compact run of ~30 image rows, so the kh = 0 / 2 re-reads of
    // the rows above and below hit this XCD's L2.  (With one contiguous sub-band per workgroup each of the 32 workgroups
    // kept its own three rows alive -- 12 MB against a 4 MB L2 -- and the input was fetched 3x.)
    long wt_begin, wt_count, wt_stride;
    {
        const long x0 = (long)xcd * q.wt_per_xcd, x1 = min(q.wt_total, x0 + q.wt_per_xcd);
        const long n_x = max(0L, x1 - x0);
        wt_stride = team_size;
        wt_begin = x0 + member;
        wt_count = n_x > member ? (n_x - member + team_size - 1) / team_size : 0;
    }
    constexpr int NL = PF + 2;                 // loads per step: PF centre fragments + left / right halo pixel
    const int CC = p.Cin >> 4;                 // 16-channel chunks per tap
    const int NSTEP = 3 * CC;                  // (kh, chunk) steps per tile
    using DppRor1 = std::integral_constant<int, 0x121>;
    using DppRor15 = std::integral_constant<int, 0x12F>;
    using DppShr1 = std::integral_constant<int, 0x111>;
    using DppShl1 = std::integral_constant<int, 0x101>;

    auto issue = [](f32x4& dst, const float* src) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src)); };
    struct Stage { f32x4 c[PF]; f32x4 hl, hr; };
    struct Tile { long m0; long m[PF]; };
    auto tile_of = [&](long wt, Tile& t) {
        t.m0 = wt * (PF * 16);
#pragma clang loop unroll(full)
        for (int pf = 0; pf < PF; ++pf) t.m[pf] = min(t.m0 + pf * 16 + pl_, p.M - 1);
    };
    // requests the PF + 2 loads of step `st` (= kh * CC + chunk) of tile t into stage s
    auto request = [&](const Tile& t, int st, Stage& s) {
        const int kh = st / CC, c = st - kh * CC;
        const long dm = (long)(kh - 1) * p.W;
        const int ch = 16 * c + 4 * g;
#pragma clang loop unroll(full)
        for (int pf = 0; pf < PF; ++pf) issue(s.c[pf], p.x + min(max(t.m[pf] + dm, 0L), p.M - 1) * p.Cin + ch);
        issue(s.hl, p.x + min(max(t.m0 - 1 + dm, 0L), p.M - 1) * p.Cin + ch);
        issue(s.hr, p.x + min(max(t.m0 + PF * 16 + dm, 0L), p.M - 1) * p.Cin + ch);
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
        {
            const float4* wsrc = reinterpret_cast<const float4*>(p.w) + (long)nf0 * p.KC * 64;
            const int total = NT * p.KC * 64;
            const int stride = (int)blockDim.x;
            int i0 = threadIdx.x;
            for (; i0 + 7 * stride < total; i0 += 8 * stride) {
                float4 v[8];
#pragma clang loop unroll(full)
                for (int u = 0; u < 8; ++u) v[u] = wsrc[i0 + u * stride];
#pragma clang loop unroll(full)
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + u * stride, l = i & 63, f = i >> 6;   // f = nf * KC + kc
                    const int nf = f / p.KC, kc = f - nf * p.KC;
                    ws_lds[(kc * NT + nf) * 64 + l] = v[u];
                }
            }
            for (; i0 < total; i0 += stride) {
                const int l = i0 & 63, f = i0 >> 6;
                const int nf = f / p.KC, kc = f - nf * p.KC;
                ws_lds[(kc * NT + nf) * 64 + l] = wsrc[i0];
            }
        }
        __syncthreads();

        long u = grab();
        if (u >= wt_count) continue;
        Tile cur, nxt;
        tile_of(wt_begin + u * wt_stride, cur);
        Stage s0, s1, s2;
        auto stage = [&](auto Ic) -> Stage& {
            constexpr int I = decltype(Ic)::value % 3;
            if constexpr (I == 0) return s0; else if constexpr (I == 1) return s1; else return s2;
        };
        auto hold_all = [&]() {   // s_waitcnt vmcnt(0) tied to every stage register
            asm volatile("s_waitcnt vmcnt(0)");
#pragma clang loop unroll(full)
            for (int pf = 0; pf < PF; ++pf) asm volatile("" : "+v"(s0.c[pf]), "+v"(s1.c[pf]), "+v"(s2.c[pf]));
            asm volatile("" : "+v"(s0.hl), "+v"(s0.hr), "+v"(s1.hl), "+v"(s1.hr), "+v"(s2.hl), "+v"(s2.hr));
        };
        request(cur, 0, s0);
        request(cur, 1, s1);
        hold_all();

        const float4* wl = ws_lds + lane;
        auto run_tile = [&]() {
            // padding masks of this lane's pixels: row kh valid, left / right neighbour inside the row
            bool rowok[PF][3], lok[PF], rok[PF];
#pragma clang loop unroll(full)
            for (int pf = 0; pf < PF; ++pf) {
                const unsigned m = (unsigned)cur.m[pf];           // M < 2^31 (checked by the host): 32-bit divisions
                const unsigned hw = (unsigned)(p.H * p.W);
                const unsigned r = m % hw;
                const int oh = (int)(r / (unsigned)p.W), ow = (int)(r % (unsigned)p.W);
                rowok[pf][0] = oh > 0; rowok[pf][1] = true; rowok[pf][2] = oh < p.H - 1;
                lok[pf] = ow > 0; rok[pf] = ow < p.W - 1;
            }
            f32x4 acc[NT][PF];
#pragma clang loop unroll(full)
            for (int nf = 0; nf < NT; ++nf) {
                const float4 bq = *reinterpret_cast<const float4*>(lds_bias + nf * 16 + g * 4);
#pragma clang loop unroll(full)
                for (int pf = 0; pf < PF; ++pf) acc[nf][pf] = (f32x4){bq.x, bq.y, bq.z, bq.w};
            }
            auto step = [&](int st, auto SIc, auto WAITc) {
                constexpr int SI = decltype(SIc)::value;
                constexpr bool WAIT = decltype(WAITc)::value;
                Stage& cs = stage(std::integral_constant<int, SI>{});
                Stage& ns = stage(std::integral_constant<int, SI + 2>{});
                if (WAIT) {
                    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NL));
#pragma clang loop unroll(full)
                    for (int pf = 0; pf < PF; ++pf) asm volatile("" : "+v"(cs.c[pf]));
                    asm volatile("" : "+v"(cs.hl), "+v"(cs.hr));
                }
                {   // request step st+2 of the stream: this tile's, or the next tile's step st+2-NSTEP
                    const int s2i = st + 2;
                    if (s2i < NSTEP) request(cur, s2i, ns);
                    else request(nxt, s2i - NSTEP, ns);
                }
                const int kh = st / CC, c = st - kh * CC;
                // weight fragments of the three taps of this row for this channel chunk
                float4 w[3][NT];
#pragma clang loop unroll(full)
                for (int kw = 0; kw < 3; ++kw)
#pragma clang loop unroll(full)
                    for (int nf = 0; nf < NT; ++nf) w[kw][nf] = wl[((long)((kh * 3 + kw) * CC + c) * NT + nf) * 64];
#pragma clang loop unroll(full)
                for (int pf = 0; pf < PF; ++pf) {
                    const bool rv = kh == 0 ? rowok[pf][0] : kh == 1 ? rowok[pf][1] : rowok[pf][2];
                    // kw = 0: pixel p-1 (lane 0 of the row takes lane 15 of the previous fragment / the left halo pixel)
                    f32x4 xl = dpp4(dpp4(cs.c[pf], pf == 0 ? cs.hl : cs.c[pf == 0 ? 0 : pf - 1], DppRor1{}), cs.c[pf], DppShr1{});
                    // kw = 2: pixel p+1 (lane 15 takes lane 0 of the next fragment / the right halo pixel)
                    f32x4 xr = dpp4(dpp4(cs.c[pf], pf == PF - 1 ? cs.hr : cs.c[pf == PF - 1 ? pf : pf + 1], DppRor15{}), cs.c[pf], DppShl1{});
                    f32x4 xc = cs.c[pf];
#pragma clang loop unroll(full)
                    for (int e = 0; e < 4; ++e) {
                        xl[e] = (rv && lok[pf]) ? xl[e] : 0.f;
                        xc[e] = rv ? xc[e] : 0.f;
                        xr[e] = (rv && rok[pf]) ? xr[e] : 0.f;
                    }
#pragma clang loop unroll(full)
                    for (int nf = 0; nf < NT; ++nf) {
                        const float wa[4] = {w[0][nf].x, w[0][nf].y, w[0][nf].z, w[0][nf].w};
                        const float wb[4] = {w[1][nf].x, w[1][nf].y, w[1][nf].z, w[1][nf].w};
                        const float wc[4] = {w[2][nf].x, w[2][nf].y, w[2][nf].z, w[2][nf].w};
#pragma clang loop unroll(full)
                        for (int jj = 0; jj < 4; ++jj) acc[nf][pf] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[jj], xl[jj], acc[nf][pf], 0, 0, 0);
#pragma clang loop unroll(full)
                        for (int jj = 0; jj < 4; ++jj) acc[nf][pf] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[jj], xc[jj], acc[nf][pf], 0, 0, 0);
#pragma clang loop unroll(full)
                        for (int jj = 0; jj < 4; ++jj) acc[nf][pf] = __builtin_amdgcn_mfma_f32_16x16x4f32(wc[jj], xr[jj], acc[nf][pf], 0, 0, 0);
                    }
                }
            };
            using I0 = std::integral_constant<int, 0>;
            using I1 = std::integral_constant<int, 1>;
            using I2 = std::integral_constant<int, 2>;
            using Wy = std::true_type;
            using Wn = std::false_type;
            step(0, I0{}, Wn{});
            step(1, I1{}, Wn{});
            int st = 2;
            for (; st + 2 < NSTEP; st += 3) { step(st, I2{}, Wy{}); step(st + 1, I0{}, Wy{}); step(st + 2, I1{}, Wy{}); }
            if (st < NSTEP) step(st, I2{}, Wy{});
            if (st + 1 < NSTEP) step(st + 1, I0{}, Wy{});
            hold_all();   // next tile's steps 0 / 1 landed BEFORE any store is issued
            igemm_epilogue<NT, PF, true>(p, acc, cur.m0, pl_, g, nf0, false);
        };

        for (;;) {
            const long un = grab();
            const bool has_next = un < wt_count;
            tile_of(wt_begin + (has_next ? un : u) * wt_stride, nxt);
            run_tile();
            if (!has_next) break;
            const int rot = NSTEP % 3;   // = 0 (NSTEP = 3 * CC), kept for clarity: steps 0 / 1 of the next tile sit in stages 0 / 1
            if (rot == 1) { s0 = s1; s1 = s2; }
            else if (rot == 2) { s1 = s0; s0 = s2; }
            cur = nxt; u = un;
        }
        hold_all();
    }
}

bool conv_igemm_ws3_eligible(const IgemmP& p, int nfrag) {
    static const bool on = [] { const char* e = getenv("OAR_IGEMM_WS3"); return !e || atoi(e) != 0; }();
    const bool vec_ok = ((p.Cout & 3) == 0) && ((p.y_ld & 3) == 0);
    return on && vec_ok && !p.convt && p.kh == 3 && p.kw == 3 && p.sh == 1 && p.sw == 1 && p.dh == 1 && p.dw == 1 && p.pt == 1 && p.pl == 1 &&
           p.H == p.Ho && p.W == p.Wo && (p.Cin & 15) == 0 && nfrag <= 2 && p.M >= 100000 && p.M < (1L << 31) && (size_t)nfrag * p.KC * 1024 <= 150 * 1024;
}

template <int NT>
static void launch_ws3(hipStream_t s, const IgemmP& p, size_t lds) {
    OAR_MAX_LDS_ONCE((conv_igemm_ws3_kernel<NT, 2>), 160 * 1024);
    IgemmWs3P q;
    q.g = p; q.ny = 1; q.groups = 1;
    q.wt_total = (p.M + 31) / 32;
    q.wt_per_xcd = (q.wt_total + 7) / 8;
    hipLaunchKernelGGL((conv_igemm_ws3_kernel<NT, 2>), dim3(256), dim3(1024), lds, s, q);
}

void conv_igemm_ws3(hipStream_t s, const IgemmP& p, int nfrag) {
    const size_t lds = (size_t)nfrag * p.KC * 1024 + (size_t)nfrag * 64 + 16;
    if (nfrag == 1) launch_ws3<1>(s, p, lds);
    else launch_ws3<2>(s, p, lds);
}

}  // namespace k
}  // namespace oar
