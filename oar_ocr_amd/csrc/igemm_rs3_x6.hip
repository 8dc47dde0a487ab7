// igemm_rs3_x6.hip -- 3x3 / stride 1 / pad 1 convolution with at most 16 output channels on the bf16 matrix pipe, ROW-STREAMING (round 4).
// The DB head's 64 -> 16 convolution at quarter resolution (M = 403 200 pixels, K = 576) ran on conv_igemm_ws3_kernel: f32 MFMAs (32 clocks each,
// 3.6 M of them = 47 us of matrix pipe on their own), 116 us per launch, 1.27 ms of the step.  This kernel computes the same products as bf16x6
// (the three exact truncation pieces of both operands, six MFMAs of 16 clocks per 32-deep k-step) and arranges the data so that the split is paid
// ONCE per input element instead of once per tap:
//
//   * a wave owns a strip of 16 output columns and walks down it; the three input rows a row of outputs needs live in the wave's own LDS ring as
//     PACKED bf16 PLANES ([plane h / m / l][pixel][channel], 144 B per pixel so that the 16 pixels of a 16-byte-per-lane read hit distinct banks):
//     a new input row is loaded f32 (buffer loads: pixels / rows outside the image read as zeros), split, written to the slot of the row that died;
//   * the B operand of tap (kh, kw), channel chunk c, piece p is then ONE ds_read_b128 at (row kh, pixel n + kw): the column shift is an address;
//   * the weights' h and m pieces (K = 9 Cin: 18 k-steps x 2 pieces x 16 B per lane at Cin = 64) stay in REGISTERS for the whole launch -- 144
//     VGPRs, 241 in all: two waves fit a SIMD, and six waves a workgroup (three 23 KB rings more would not fit LDS); the l pieces sit in an
//     18 KB LDS table; no barrier after the table is filled;
//   * six accumulators, one per product class (mm, lh, hl, mh, hm, hh): consecutive MFMAs never depend on each other, and the classes are added
//     smallest first at the end.
// Work item = (image, segment of R rows, strip), dealt in XCD bands as dsblock_rs.inc does.  Algorithmic bytes: 4 (M Cin + M Cout).
#include <hip/hip_ext.h>

#include "igemm_dev.h"

namespace oar {
namespace k {

typedef unsigned r3_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned r3_u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 r3_bf16x8 __attribute__((ext_vector_type(8)));

struct Rs3P {
    const float* x; float* y; const float4* w; const float* bias;
    int N, H, W, Cin, Cout, y_ld;
    int x_ld;            // floats between two pixels of x (>= Cin): a pass over a 64- or 32-channel slice of a wider tensor (round 6: Cin = 96 ... 256 as several passes)
    int accum;           // 1: this pass adds its sum to what y already holds (the previous passes' partial sums); bias / activation belong to the first / last pass (host)
    int act; float alpha, beta;
    int R, segs, tiles_x, items, per_xcd;
    unsigned img_bytes, y_bytes;
};

namespace {
constexpr unsigned kR3Oob = 0x40000000u, kR3OobSt = 0x80000000u;
constexpr int kR3IW = 18, kR3PxB = 144, kR3PlB = kR3IW * kR3PxB, kR3RowB = 3 * kR3PlB, kR3Waves = 6;

__device__ __forceinline__ r3_u32x4 r3_lds4(unsigned off) {
    return *reinterpret_cast<const __attribute__((address_space(3))) r3_u32x4*>((__attribute__((address_space(3))) const char*)nullptr + off);
}
__device__ __forceinline__ void r3_lds_w2(unsigned off, r3_u32x2 v) {
    *reinterpret_cast<__attribute__((address_space(3))) r3_u32x2*>((__attribute__((address_space(3))) char*)nullptr + off) = v;
}

// CC32 = Cin / 32 (k-steps per tap)
// DBG (timing ablations, wrong results; OAR_RS3_DBG): 1 no MFMA, 2 no split / ring writes, 4 no operand reads, 8 no stores, 16 no row loads
template <int CC32, int DBG = 0>
__global__ __launch_bounds__(kR3Waves * 64, 2) void conv3x3_n16_x6_kernel(Rs3P p) {
    constexpr int QPP = CC32 * 8;                           // float4 quads per pixel
    constexpr int NU = kR3IW * QPP;                         // quads of one strip row
    constexpr int NJ = (NU + 63) / 64;                      // loads per lane and row
    constexpr int KC = 9 * CC32;
    extern __shared__ float4 r3_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, n = lane & 15;
    const unsigned ring0 = (unsigned)(wave * 3 * kR3RowB);

    // ---- this lane's weight pieces: A operands of all 9 * CC32 k-steps (IGEMM_W_X6: [kc][piece][lane] 16 bytes)
    // (the h and m pieces: 144 registers at Cin = 64.  With the l pieces as well the kernel needed more than the 256 registers vector instructions
    // can address and shuttled weights through accumulation registers before every MFMA; the l piece feeds one product in six and is read from an
    // LDS table shared by the four waves instead)
    r3_u32x4 wreg[KC][2];
#pragma unroll
    for (int kc = 0; kc < KC; ++kc)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const float4 v = p.w[(kc * 3 + s) * 64 + lane];
            wreg[kc][s] = (r3_u32x4){__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
        }
    constexpr unsigned kWl = (unsigned)(kR3Waves * 3 * kR3RowB);   // [kc][lane] 16 bytes, behind the rings
    for (int i = tid; i < KC * 64; i += kR3Waves * 64) r3_lds[(kWl >> 4) + i] = p.w[((i >> 6) * 3 + 2) * 64 + (i & 63)];
    __syncthreads();
    const float4 bq = (p.bias && g * 4 < p.Cout) ? *reinterpret_cast<const float4*>(p.bias + g * 4) : make_float4(0.f, 0.f, 0.f, 0.f);

    const int xcd = (int)(blockIdx.x & 7), wgx = (int)(blockIdx.x >> 3), wgs = (int)(gridDim.x >> 3);
    const int b0 = xcd * p.per_xcd, b1 = min(p.items, b0 + p.per_xcd);
    const int J = wgs * kR3Waves, j0 = wgx * kR3Waves + wave;
    const int row_bytes = p.W * p.x_ld * 4, orow_bytes = p.W * p.y_ld * 4;
    const __amdgpu_buffer_rsrc_t ysrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)p.y_bytes, 0x00020000);
    const unsigned rd_lane = (unsigned)(n * kR3PxB + g * 16);   // + row slot + piece * kR3PlB + kw * kR3PxB + c * 64

    for (int item = b0 + j0; item < b1; item += J) {
        const int tx = item % p.tiles_x, q_ = item / p.tiles_x, seg = q_ % p.segs, img = q_ / p.segs;
        const int o_begin = seg * p.R, o_end = min(p.H, o_begin + p.R);
        const int x0 = tx * 16 - 1;
        float* ximg = const_cast<float*>(p.x) + (long)img * (p.img_bytes >> 2);
        // per-lane source offsets inside a row / destination offsets inside a ring slot
        unsigned src[NJ], dst[NJ];
#pragma unroll
        for (int u = 0; u < NJ; ++u) {
            const int q = u * 64 + lane, px = q / QPP, cq = q - px * QPP;
            const bool ok = q < NU && (unsigned)(x0 + px) < (unsigned)p.W;
            src[u] = ok ? (unsigned)(((x0 + px) * p.x_ld + cq * 4) * 4) : kR3Oob;
            dst[u] = q < NU ? (unsigned)(px * kR3PxB + cq * 8) : 0xFFFFFFFFu;
        }
        r3_u32x4 in[NJ];
        auto load_row = [&](int y, r3_u32x4 (&in)[NJ]) __attribute__((always_inline)) {
            const bool rok = (unsigned)y < (unsigned)p.H;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(ximg, 0, rok ? (int)p.img_bytes : 0, 0x00020000);
            const int so = rok ? y * row_bytes : 0;
#pragma unroll
            for (int u = 0; u < NJ; ++u) in[u] = (DBG & 16) ? (r3_u32x4){(unsigned)y, 1u, 2u, 3u} : __builtin_amdgcn_raw_buffer_load_b128(rs, (int)src[u], so, 0);
        };
        // exact three-way bf16 split of the loaded quads -> the three planes of ring slot `slot`
        auto put_row = [&](unsigned slot, const r3_u32x4 (&in)[NJ]) __attribute__((always_inline)) {
#pragma unroll
            for (int u = 0; u < NJ; ++u) {
                if constexpr ((DBG & 2) != 0) { if (in[u][0] == 0x12345u) r3_lds_w2(slot, (r3_u32x2){in[u][1], in[u][2]}); continue; }
                unsigned hb[4], mb[4], lb[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned ux = in[u][e];
                    const float xf = __uint_as_float(ux);
                    const float r1 = xf - __uint_as_float(ux & 0xFFFF0000u);
                    const unsigned u1 = __float_as_uint(r1);
                    const float r2 = r1 - __uint_as_float(u1 & 0xFFFF0000u);
                    hb[e] = ux; mb[e] = u1; lb[e] = __float_as_uint(r2);
                }
                if (dst[u] != 0xFFFFFFFFu) {
                    const unsigned a = slot + dst[u];
                    r3_lds_w2(a, (r3_u32x2){__builtin_amdgcn_perm(hb[1], hb[0], 0x07060302u), __builtin_amdgcn_perm(hb[3], hb[2], 0x07060302u)});
                    r3_lds_w2(a + kR3PlB, (r3_u32x2){__builtin_amdgcn_perm(mb[1], mb[0], 0x07060302u), __builtin_amdgcn_perm(mb[3], mb[2], 0x07060302u)});
                    r3_lds_w2(a + 2 * kR3PlB, (r3_u32x2){__builtin_amdgcn_perm(lb[1], lb[0], 0x07060302u), __builtin_amdgcn_perm(lb[3], lb[2], 0x07060302u)});
                }
            }
        };
        // ---- warm-up: input rows o_begin - 1, o_begin, o_begin + 1 into slots 0, 1, 2
        {   // (the three rows are requested together: one memory latency per item, not three)
            r3_u32x4 in1[NJ], in2[NJ];
            load_row(o_begin - 1, in);
            load_row(o_begin, in1);
            load_row(o_begin + 1, in2);
            put_row(ring0, in);
            put_row(ring0 + (unsigned)kR3RowB, in1);
            put_row(ring0 + (unsigned)(2 * kR3RowB), in2);
        }
        int s0 = 0;   // ring slot of the row above the output row
        const int ox = tx * 16 + n;
        const unsigned st_lane = (ox < p.W && g * 4 < p.Cout) ? (unsigned)((ox * p.y_ld + g * 4) * 4) : kR3OobSt;
        unsigned st_row = (unsigned)((img * p.H + o_begin) * orow_bytes);
#pragma unroll 1
        for (int r = o_begin; r < o_end; ++r) {
            load_row(r + 2, in);                               // lands while this row is multiplied
            r3_u32x4 prev = (r3_u32x4){0u, 0u, 0u, 0u};       // partial sums of the earlier channel slices (out-of-range lanes read zeros)
            if (p.accum) prev = __builtin_amdgcn_raw_buffer_load_b128(ysrc, (int)(st_lane + st_row), 0, 0);
            f32x4 acc[6];                                      // one accumulator per product class: six independent MFMA chains
            acc[5] = (f32x4){bq.x, bq.y, bq.z, bq.w};
            acc[0] = acc[1] = acc[2] = acc[3] = acc[4] = (f32x4){0.f, 0.f, 0.f, 0.f};
            unsigned rd = ring0 + rd_lane;
            unsigned slot_of[3];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) slot_of[kh] = (unsigned)((s0 + kh >= 3 ? s0 + kh - 3 : s0 + kh) * kR3RowB);
            // k-step kc = (kh, kw, c): its three pieces are read two k-steps ahead; the reads of kc + 3 are ordered behind the DATA of kc (an empty asm,
            // not volatile: a dependence, not a scheduling barrier), so at most three k-steps of operands are ever in flight
            r3_u32x4 xs[3][4];   // x pieces h, m, l and the weights' l piece
            auto fetch = [&](int kc, r3_u32x4 (&dstv)[4]) __attribute__((always_inline)) {
                const int kh = kc / (3 * CC32), kw = (kc / CC32) % 3, c = kc % CC32;
                if constexpr ((DBG & 4) != 0) { for (int s_ = 0; s_ < 4; ++s_) dstv[s_] = (r3_u32x4){rd + kc, 1u, 2u, (unsigned)s_}; return; }
#pragma unroll
                for (int s_ = 0; s_ < 3; ++s_) dstv[s_] = r3_lds4(rd + slot_of[kh] + (unsigned)(s_ * kR3PlB + kw * kR3PxB + c * 64));
                dstv[3] = r3_lds4(rd - rd_lane - ring0 + kWl + (unsigned)(kc * 1024 + lane * 16));
            };
            fetch(0, xs[0]);
            fetch(1, xs[1]);
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                if (kc + 2 < KC) fetch(kc + 2, xs[(kc + 2) % 3]);
                // six products, smallest first: (w piece, x piece) = mm, lh, hl, mh, hm, hh
                constexpr int WPL[6] = {1, 2, 0, 1, 0, 0}, XPL[6] = {1, 0, 2, 0, 1, 0};
#pragma unroll
                for (int t = 0; t < 6; ++t)
                    if constexpr ((DBG & 1) != 0) acc[t][0] += __uint_as_float(xs[kc % 3][XPL[t]][0] ^ wreg[kc][0][t & 3]);
                    else acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(r3_bf16x8, WPL[t] == 2 ? xs[kc % 3][3] : wreg[kc][WPL[t] == 2 ? 0 : WPL[t]]), __builtin_bit_cast(r3_bf16x8, xs[kc % 3][XPL[t]]), acc[t], 0, 0, 0);
                asm("" : "+v"(rd) : "v"(xs[kc % 3][2]));
            }
            f32x4 o = ((((acc[0] + acc[1]) + acc[2]) + acc[3]) + acc[4]) + acc[5];   // smallest classes first
            r3_u32x4 v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __float_as_uint(apply_act(o[e] + __uint_as_float(prev[e]), p.act, p.alpha, p.beta));
            // every read of the oldest row has returned (its data fed the MFMAs above): its slot takes row r + 2
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            put_row(ring0 + (unsigned)(s0 * kR3RowB), in);
            __builtin_amdgcn_raw_buffer_store_b128(v, ysrc, (int)(((DBG & 8) && v[0] != 0x12345u) ? kR3OobSt : st_lane + st_row), 0, 0);
            st_row += (unsigned)orow_bytes;
            s0 = s0 + 1 == 3 ? 0 : s0 + 1;
        }
    }
}
}  // namespace

bool conv3x3_n16_x6_eligible(long M, int Cin, int Cout, long img_px, int y_ld) {
    static const bool on = [] { const char* e = getenv("OAR_IGEMM_RS3"); return !e || atoi(e) != 0; }();
    return on && (Cin == 32 || Cin == 64) && Cout >= 4 && Cout <= 16 && (Cout & 3) == 0 && (y_ld & 3) == 0 && M >= 100000 && img_px * Cin * 4 < (1L << 29) && M * y_ld * 4 < (1L << 31);
}
// wider inputs as passes over 64- / 32-channel slices (each pass: weights' h and m pieces in registers): the slice widths, empty when not eligible
std::vector<int> conv3x3_n16_x6_slices(long M, int Cin, int Cout, long img_px, int y_ld) {
    std::vector<int> v;
    if (Cin <= 64 || Cin > 256 || Cin % 32 != 0 || img_px * Cin * 4 >= (1L << 29) || !conv3x3_n16_x6_eligible(M, 64, Cout, img_px, y_ld)) return v;
    for (int c = 0; c < Cin; c += 64) v.push_back(std::min(64, Cin - c));
    return v;
}

void conv3x3_n16_x6(hipStream_t s, const IgemmP& g, int n_images) {
    Rs3P p{};
    p.x = g.x; p.y = g.y; p.w = reinterpret_cast<const float4*>(g.w); p.bias = g.bias;
    p.N = n_images; p.H = g.H; p.W = g.W; p.Cin = g.Cin; p.Cout = g.Cout; p.y_ld = g.y_ld;
    p.x_ld = g.x_ld > 0 ? g.x_ld : g.Cin; p.accum = g.accum;
    p.act = g.act; p.alpha = g.alpha; p.beta = g.beta;
    p.tiles_x = (g.W + 15) / 16;
    // rows per item: the fewest wave rounds, then the least warm-up (three rows per item)
    const long waves = 256L * kR3Waves;
    double best = 1e30;
    for (int R = std::min(g.H, 4); R <= g.H; ++R) {
        const int segs = (g.H + R - 1) / R;
        const long items = (long)n_images * segs * p.tiles_x;
        const long rounds = (items + waves - 1) / waves;
        const double cost = (double)rounds * (R + 2.5);
        if (cost < best - 1e-9) { best = cost; p.R = R; p.segs = segs; p.items = (int)items; }
    }
    p.per_xcd = (p.items + 7) / 8;
    p.img_bytes = (unsigned)((long)g.H * g.W * p.x_ld * 4);
    p.y_bytes = (unsigned)((long)n_images * g.H * g.W * g.y_ld * 4);
    const size_t lds = (size_t)kR3Waves * 3 * kR3RowB + (size_t)9 * (g.Cin / 32) * 1024;   // rings + the weights' l pieces
    auto launch = [&](auto kernel) {
        OAR_MAX_LDS_ONCE(kernel, 160 * 1024);
        hipLaunchKernelGGL(kernel, dim3(256), dim3(kR3Waves * 64), lds, s, p);
    };
    static const int dbg = [] { const char* e = getenv("OAR_RS3_DBG"); return e ? atoi(e) : 0; }();
    if (dbg && g.Cin == 64) {
        switch (dbg) {
            case 1: launch(conv3x3_n16_x6_kernel<2, 1>); return;
            case 2: launch(conv3x3_n16_x6_kernel<2, 2>); return;
            case 4: launch(conv3x3_n16_x6_kernel<2, 4>); return;
            case 8: launch(conv3x3_n16_x6_kernel<2, 8>); return;
            case 16: launch(conv3x3_n16_x6_kernel<2, 16>); return;
            case 24: launch(conv3x3_n16_x6_kernel<2, 24>); return;
            case 26: launch(conv3x3_n16_x6_kernel<2, 26>); return;
            case 27: launch(conv3x3_n16_x6_kernel<2, 27>); return;
            case 31: launch(conv3x3_n16_x6_kernel<2, 31>); return;
            default: break;
        }
    }
    if (g.Cin == 64) launch(conv3x3_n16_x6_kernel<2>);
    else launch(conv3x3_n16_x6_kernel<1>);
}

}  // namespace k
}  // namespace oar
