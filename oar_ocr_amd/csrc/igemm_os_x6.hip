// igemm_os_x6.hip -- output-stationary implicit-GEMM conv / Linear on the bf16 matrix pipe with f32-equivalent accuracy
// (the exact 3-way bf16 split "bf16x6" of igemm_ws_x6.hip), for the layers whose weights do NOT fit LDS: long-K 1x1 /
// Linear layers (K >= 384) and k x k convolutions (K = kh * kw * Cin up to a few thousand).
//
// The weight-stationary kernels keep a cout tile of W resident in LDS for the whole kernel and stream pixels past it; that
// needs NT * K * 96 bytes of LDS and stops at K ~ 200 for 128 couts.  Here the accumulators are the stationary operand:
//   * a workgroup (4 waves) owns a 128-pixel x (BNF * 16)-cout output tile; wave w owns pixels [32 w, 32 w + 32) x all couts
//     of the tile (2 x BNF accumulator fragments, 64 VGPRs at BNF = 8).  Two workgroups share a CU: they drift out of phase,
//     so the VALU / barrier phases of one overlap the MFMA phase of the other (a scheduler-level interleave of the split
//     with the MFMAs of the SAME wave was tried -- sched_group_barrier, pipelined split -- and the compiler undid it);
//   * K is walked in chunks of 32.  Each wave loads ITS OWN two pixel fragments of the chunk straight from HBM / L2 into
//     registers (one 32-byte group per lane and fragment, three register stages = two chunks of prefetch) and splits them
//     into the three bf16 planes in registers -- pixel data never touches LDS;
//   * the chunk's weights (pre-split on the host into planes, fragment order) are shared by all 8 waves: the workgroup
//     copies them L2 -> LDS (BNF * 3 KB per chunk, double buffered, one barrier per chunk).  One weight-fragment read from
//     LDS (3 planes) feeds 12 MFMAs (2 pixel fragments x 6 products): half the LDS traffic per MFMA of the
//     weight-stationary x6 kernel, whose LDS read rate equals its MFMA rate.
// Waits are placed by hand (the loads are inline asm): per chunk the only vector-memory wait is vmcnt(4) before the LDS copy
// of the NEXT chunk's weights, which leaves the four pixel loads of chunk kc + 2 in flight; the barrier is a bare
// s_waitcnt lgkmcnt(0) + s_barrier, so it does not drain them either.
#include "igemm_dev.h"

namespace oar {
namespace k {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct IgemmOsP {
    IgemmP g;            // g.KC = ceil(K / 32); g.w in x6 fragment order
    int ny;              // cout tiles of BNF fragments
    int nfrag_alloc;     // cout fragments present in g.w (rows are padded to 64)
    long m_tiles;        // ceil(M / pixels per workgroup)
    long m_per_xcd;      // pixel tiles per XCD band
};

__device__ __forceinline__ void os_split3(const f32x4& a, const f32x4& b, uint4& h, uint4& m, uint4& l) {
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
    h = make_uint4((hh[0] >> 16) | hh[1], (hh[2] >> 16) | hh[3], (hh[4] >> 16) | hh[5], (hh[6] >> 16) | hh[7]);
    m = make_uint4((mm[0] >> 16) | mm[1], (mm[2] >> 16) | mm[3], (mm[4] >> 16) | mm[5], (mm[6] >> 16) | mm[7]);
    l = make_uint4((ll[0] >> 16) | ll[1], (ll[2] >> 16) | ll[3], (ll[4] >> 16) | ll[5], (ll[6] >> 16) | ll[7]);
}

constexpr int kOsThreads = 256;   // 4 waves: two workgroups per CU drift out of phase, so one's split / barrier overlaps the other's MFMAs

template <int BNF, bool IS1X1>
__global__ __launch_bounds__(kOsThreads, 2) void conv_igemm_os_x6_kernel(IgemmOsP q) {
    extern __shared__ uint4 w_lds[];   // [2 stages][BNF][3 planes][64 lanes]
    const IgemmP& p = q.g;
    constexpr int PF = 2;
    constexpr int WCH = BNF * 192;                 // uint4 per weight chunk
    constexpr int NTHR = kOsThreads;
    constexpr int NWL = (WCH + NTHR - 1) / NTHR;   // weight loads per thread and chunk
    constexpr int NXL = 2 * PF;                    // pixel loads per lane and chunk
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pl_ = lane & 15, g = lane >> 4;
    // workgroup -> (pixel tile, cout tile): the cout tiles of one pixel tile are neighbours on one XCD (they re-read the same pixels)
    const int xcd = (int)(blockIdx.x & 7);
    const long j = (long)(blockIdx.x >> 3);
    const long m_local = j / q.ny;
    const int ntile = (int)(j - m_local * q.ny);
    const long m_tile = (long)xcd * q.m_per_xcd + m_local;
    if (m_local >= q.m_per_xcd || m_tile >= q.m_tiles) return;
    const int nf0 = ntile * BNF;
    const long m0 = m_tile * (NTHR / 2) + wave * 32;

    // ---- this lane's two pixels
    long pix_base[PF], mrow[PF];
    int ih0[PF], iw0[PF];
#pragma clang loop unroll(full)
    for (int pf = 0; pf < PF; ++pf) {
        const long m = min(m0 + pf * 16 + pl_, p.M - 1);   // clamped rows compute garbage that is never stored
        mrow[pf] = m;
        if (IS1X1) { pix_base[pf] = m * (long)p.x_ld; ih0[pf] = 0; iw0[pf] = 0; }
        else {
            const long hw = (long)p.Ho * p.Wo;
            const long n = m / hw, r = m - n * hw;
            const int oh = (int)(r / p.Wo), ow = (int)(r - (long)oh * p.Wo);
            pix_base[pf] = n * (long)p.H * p.W * p.x_ld;
            ih0[pf] = oh * p.sh - p.pt; iw0[pf] = ow * p.sw - p.pl;
        }
    }
    auto issue = [](f32x4& dst, const float* src) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src)); };
    auto issue_w = [](u32x4& dst, const uint4* src) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(src)); };
    struct Stage { f32x4 a[PF], b[PF]; unsigned okmask; };
    auto x_request = [&](Stage& st, int kc) {
        const int k = min(kc * 32 + 8 * g, p.K - 8);   // zero-padded K tail of W: re-read a valid group
        st.okmask = 0;
#pragma clang loop unroll(full)
        for (int pf = 0; pf < PF; ++pf) {
            const float* src;
            if (IS1X1) {
                if (p.n_msrc > 0) {   // the input is a channel concat that was never materialised: this lane's 8-channel group lies in ONE source (8 | every width)
                    const float* sp = p.msrc[0];
                    int sc = p.msrc_c[0], k0s = 0, acc_c = p.msrc_c[0];
#pragma clang loop unroll(full)
                    for (int qi = 1; qi < 8; ++qi) {   // (static indices into the kernel arguments: a chain of selects, no register-indexed access)
                        const bool hit = qi < p.n_msrc && k >= acc_c;
                        sp = hit ? p.msrc[qi] : sp; sc = hit ? p.msrc_c[qi] : sc; k0s = hit ? acc_c : k0s;
                        acc_c += p.msrc_c[qi];
                    }
                    src = sp + mrow[pf] * (long)sc + (k - k0s);
                } else {
                    src = p.x + pix_base[pf] + k;
                }
                st.okmask |= 1u << pf;
            }
            else {
                const int tap = (int)__umulhi((unsigned)k, p.cin_magic), ci = k - tap * p.Cin;   // (Cin >= 4 on these kernels: the magic number exists)
                const int tap_h = igemm_tap_h(p, tap), tap_w = tap - tap_h * p.kw;
                const int ih = ih0[pf] + tap_h * p.dh, iw = iw0[pf] + tap_w * p.dw;
                const bool ok = ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;
                const int ihc = min(max(ih, 0), p.H - 1), iwc = min(max(iw, 0), p.W - 1);
                src = p.x + pix_base[pf] + ((long)ihc * p.W + iwc) * p.x_ld + ci;
                st.okmask |= (ok ? 1u : 0u) << pf;
            }
            issue(st.a[pf], src); issue(st.b[pf], src + 4);
        }
    };
    // weights of chunk kc: thread t copies uint4 #(t + 512 u) of the chunk's [BNF][3][64] block
    const uint4* wsrc = reinterpret_cast<const uint4*>(p.w);
    u32x4 wreg[NWL];
    auto w_request = [&](int kc) {
#pragma clang loop unroll(full)
        for (int u = 0; u < NWL; ++u) {
            const int i = min(tid + NTHR * u, WCH - 1);
            const int nf = i / 192, r = i - nf * 192;
            const int nfg = min(nf0 + nf, q.nfrag_alloc - 1);   // fragments past the allocation: any valid address (their results are never stored)
            issue_w(wreg[u], wsrc + ((long)nfg * p.KC + kc) * 192 + r);
        }
    };
    auto w_commit = [&](int stage) {
#pragma clang loop unroll(full)
        for (int u = 0; u < NWL; ++u) {
            const int i = tid + NTHR * u;
            if (i < WCH) w_lds[stage * WCH + i] = make_uint4(wreg[u][0], wreg[u][1], wreg[u][2], wreg[u][3]);
        }
    };

    f32x4 acc[BNF][PF];
#pragma clang loop unroll(full)
    for (int nf = 0; nf < BNF; ++nf) {
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        const int c = (nf0 + nf) * 16 + g * 4;
        if (p.bias && c < p.gemm_cout) b = *reinterpret_cast<const float4*>(p.bias + c);
#pragma clang loop unroll(full)
        for (int pf = 0; pf < PF; ++pf) acc[nf][pf] = (f32x4){b.x, b.y, b.z, b.w};
    }

    Stage s0, s1, s2;   // raw pixels of chunk j live in stage j % 3
    auto tie = [](Stage& st) { asm volatile("" : "+v"(st.a[0]), "+v"(st.b[0]), "+v"(st.a[1]), "+v"(st.b[1])); };

    // prologue: weights of chunk 0 -> LDS stage 0; pixels of chunks 0 and 1 requested
    w_request(0);
    x_request(s0, 0);
    x_request(s1, min(1, p.KC - 1));
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NXL) : "memory");   // the weight loads are the oldest
#pragma clang loop unroll(full)
    for (int u = 0; u < NWL; ++u) asm volatile("" : "+v"(wreg[u]));
    w_commit(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    const uint4* wl = w_lds + lane;
    // chunk kc: `cs` holds its raw pixels, `rq` receives the request for chunk kc + 2
    auto chunk = [&](int kc, Stage& cs, Stage& rq, bool last) {
        // requests first: next chunk's weights, then the pixels two chunks ahead (issue order = completion-count order)
        if (!last) w_request(kc + 1);
        x_request(rq, min(kc + 2, p.KC - 1));
        // The pixels of this chunk were requested two chunks ago: the wait at the end of the previous chunk covered them -- except
        // in chunk 0, where they come from the prologue and everything issued after them may stay in flight
        if (kc == 0) {
            if (last) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NXL) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NXL + NWL) : "memory");
        }
        tie(cs);
        uint4 xs[PF][3];
#pragma clang loop unroll(full)
        for (int pf = 0; pf < PF; ++pf) {
            f32x4 a = cs.a[pf], b = cs.b[pf];
            if (!IS1X1 && !((cs.okmask >> pf) & 1u)) { a = (f32x4){0.f, 0.f, 0.f, 0.f}; b = a; }   // padding taps
            os_split3(a, b, xs[pf][0], xs[pf][1], xs[pf][2]);
        }
        const uint4* ws = wl + (kc & 1) * WCH;
        constexpr int WP[6] = {1, 2, 0, 1, 0, 0};       // (w plane, x plane) = mm, lh, hl, mh, hm, hh: smallest terms first
        constexpr int XP[6] = {1, 0, 2, 0, 1, 0};
        uint4 wr[2][3];   // weight fragments: the one in use and the next one (its LDS read overlaps the MFMAs of this one)
#pragma clang loop unroll(full)
        for (int pq = 0; pq < 3; ++pq) wr[0][pq] = ws[pq * 64];
#pragma clang loop unroll(full)
        for (int nf = 0; nf < BNF; ++nf) {
            if (nf + 1 < BNF) {   // fenced like the ring of igemm_ws_x6.hip: without the fences the scheduler sinks these reads below the MFMAs
                __builtin_amdgcn_sched_barrier(0);
#pragma clang loop unroll(full)
                for (int pq = 0; pq < 3; ++pq) wr[(nf + 1) & 1][pq] = ws[((nf + 1) * 3 + pq) * 64];
                __builtin_amdgcn_sched_barrier(0);
            }
            const uint4* w = wr[nf & 1];
#pragma clang loop unroll(full)
            for (int t = 0; t < 6; ++t)
#pragma clang loop unroll(full)
                for (int pf = 0; pf < PF; ++pf)   // the two pixel fragments alternate: no back-to-back MFMAs on one accumulator
                    acc[nf][pf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[WP[t]]), __builtin_bit_cast(bf16x8, xs[pf][XP[t]]), acc[nf][pf], 0, 0, 0);
        }
        if (!last) {
            // next chunk's weights have to be in registers now; the NXL pixel loads issued after them may stay in flight
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NXL) : "memory");
#pragma clang loop unroll(full)
            for (int u = 0; u < NWL; ++u) asm volatile("" : "+v"(wreg[u]));
            w_commit((kc + 1) & 1);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
    };
    int kc = 0;
    for (; kc + 2 < p.KC; kc += 3) {
        chunk(kc, s0, s2, false);
        chunk(kc + 1, s1, s0, false);
        chunk(kc + 2, s2, s1, kc + 3 >= p.KC);
    }
    // tail: after a multiple of three chunks the stages are back in place (chunk kc in s0, kc + 1 in s1)
    if (kc < p.KC) chunk(kc, s0, s2, kc + 1 >= p.KC);
    if (kc + 1 < p.KC) chunk(kc + 1, s1, s0, true);
    // drain the over-requested pixel loads before the registers are reused by the epilogue
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(s0.a[0]), "+v"(s0.b[0]), "+v"(s1.a[0]), "+v"(s1.b[0]), "+v"(s2.a[0]), "+v"(s2.b[0]));
    asm volatile("" : "+v"(s0.a[1]), "+v"(s0.b[1]), "+v"(s1.a[1]), "+v"(s1.b[1]), "+v"(s2.a[1]), "+v"(s2.b[1]));
    igemm_epilogue<BNF, PF, true>(p, acc, m0, pl_, g, nf0, false);
}

template <int BNF>
static void launch_os_x6(hipStream_t s, const IgemmP& p, int nfrag, int nfrag_alloc, bool is1x1) {
    IgemmOsP q;
    q.g = p;
    q.ny = (nfrag + BNF - 1) / BNF;
    q.nfrag_alloc = nfrag_alloc;
    q.m_tiles = (p.M + kOsThreads / 2 - 1) / (kOsThreads / 2);
    q.m_per_xcd = (q.m_tiles + 7) / 8;
    const size_t lds = (size_t)2 * BNF * 192 * sizeof(uint4);
    const dim3 grid((unsigned)(q.m_per_xcd * q.ny * 8));
    if (is1x1) hipLaunchKernelGGL((conv_igemm_os_x6_kernel<BNF, true>), grid, dim3(kOsThreads), lds, s, q);
    else hipLaunchKernelGGL((conv_igemm_os_x6_kernel<BNF, false>), grid, dim3(kOsThreads), lds, s, q);
}

// cout fragments per workgroup tile: the one with less padded (wasted) MFMA work, 8 on ties; 2 for the 32-channel groups of a grouped convolution
int igemm_os_x6_tile(int nfrag) {
    if (nfrag <= 2) return 2;
    const int p8 = (nfrag + 7) / 8 * 8, p4 = (nfrag + 3) / 4 * 4;
    return p4 < p8 ? 4 : 8;
}

void conv_igemm_os_x6(hipStream_t s, const IgemmP& p, int nfrag, bool is1x1) {
    const int nfrag_alloc = (p.gemm_cout + 63) / 64 * 4;
    const int tile = igemm_os_x6_tile(nfrag);
    if (tile == 8) launch_os_x6<8>(s, p, nfrag, nfrag_alloc, is1x1);
    else if (tile == 4) launch_os_x6<4>(s, p, nfrag, nfrag_alloc, is1x1);
    else launch_os_x6<2>(s, p, nfrag, nfrag_alloc, is1x1);
}

}  // namespace k
}  // namespace oar
