// igemm_lk_x6.hip -- large-kernel "same" convolutions (k x k, stride 1, pad k / 2; the 9 x 9 convolutions of the LK-PAN neck: K = 81 Cin up to 20 736)
// on the bf16 matrix pipe with f32-equivalent accuracy (bf16x6), from an LDS-STAGED im2col TILE (round 6; BASELINE C3).
//
// The output-stationary kernel (igemm_os_x6.hip) ran these layers at 0.41 of the bf16x6 roof: every wave fetches its pixel fragments from L2 once
// PER TAP -- each input element 81 times -- and splits them into the three bf16 pieces 81 times: 43 B / clock / CU of L1 traffic and as many
// vector-ALU clocks for the split as the matrix pipe needs for the products.  Here both are paid once per input element and tile:
//
//   * a workgroup (4 waves) owns an 8-row x 32-column output tile of one image and all (<= 64) output channels; a wave owns 2 rows = 4 pixel
//     fragments x 4 cout fragments = 16 accumulators;
//   * for each 32-channel chunk of the input the tile's HALO ((8 + k - 1) x (32 + k - 1) pixels; out-of-image pixels zero) is loaded ONCE, split
//     exactly into the three bf16 planes and stored to LDS as [plane][pixel][32 channels]; the 16-byte channel group of a pixel is XOR-swizzled with
//     bits 2..3 of the pixel index, so that the 16 lanes of a fragment read (16 consecutive pixels, one group) hit 16 distinct 16-byte bank groups;
//   * the B operand of tap (dy, dx) is then ONE ds_read_b128 per plane at pixel + dy * halo_width + dx: the im2col column never exists anywhere;
//   * the tap's weights (4 cout fragments x 3 planes, 12 KB, IGEMM_W_X6 order) are loaded by every wave into registers one tap ahead of their use (two
//     register sets; L1 / L2 hits: all waves of all workgroups read the same rows); no barrier inside a chunk -- 12 ds_read_b128 per 96 MFMAs.
// Per chunk and tile: 80 KB of input read, k * k * 96 MFMAs per wave.  LDS: 3 planes x 640 pixels x 64 B = 120 KB at k = 9.
#include "igemm_dev.h"

namespace oar {
namespace k {

typedef __bf16 lk_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned lk_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned lk_u32x2 __attribute__((ext_vector_type(2)));

struct LkP {
    IgemmP g;            // g.w in IGEMM_W_X6 order, g.KC = KS * KS * Cin / 32
    int N;               // images
    int tiles_x, tiles_y;
    long tiles;          // N * tiles_y * tiles_x
    long per_xcd;
    int nfrag_alloc;
};

namespace {
__device__ __forceinline__ lk_u32x4 lk_lds4(unsigned off) {
    return *reinterpret_cast<const __attribute__((address_space(3))) lk_u32x4*>((__attribute__((address_space(3))) const char*)nullptr + off);
}
__device__ __forceinline__ void lk_lds_w2(unsigned off, lk_u32x2 v) {
    *reinterpret_cast<__attribute__((address_space(3))) lk_u32x2*>((__attribute__((address_space(3))) char*)nullptr + off) = v;
}
__device__ __forceinline__ void lk_lds_w4(unsigned off, lk_u32x4 v) {
    *reinterpret_cast<__attribute__((address_space(3))) lk_u32x4*>((__attribute__((address_space(3))) char*)nullptr + off) = v;
}

constexpr int kLkTW = 32, kLkThreads = 256;

// KS: kernel size; TH: tile rows (6 / 8 / 12: TH / 2 pixel fragments per wave, dealt round-robin over the tile's 2 TH fragments); NF: cout fragments (4, or 2 for
// the 32-channel groups of a grouped convolution -- the 5 x 5 local mixing of SVTRv2, which reads its 32 input channels out of the full tensor through g.x_ld)
template <int KS, int TH, int NF>
__global__ __launch_bounds__(kLkThreads, 1) void conv_lk_x6_kernel(LkP q) {
    constexpr int PAD = KS / 2, HWD = kLkTW + KS - 1, HHT = TH + KS - 1, NPX = HWD * HHT;
    constexpr int PF = TH / 2;                             // pixel fragments per wave
    constexpr unsigned PLB = (unsigned)NPX * 64u;          // bytes of one plane of the halo tile
    constexpr int NQ = NPX * 8;                            // float4 quads of one chunk of the halo
    constexpr int NSL = (NQ + kLkThreads - 1) / kLkThreads;
    constexpr int NSH = (NSL + 1) / 2;                     // staged in two halves: NSH quads per thread in flight
    extern __shared__ float4 lk_lds[];
    const IgemmP& p = q.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, g = lane >> 4;
    const int xcd = (int)(blockIdx.x & 7);
    const long j = (long)(blockIdx.x >> 3);
    const long tile = (long)xcd * q.per_xcd + j;
    if (j >= q.per_xcd || tile >= q.tiles) return;
    const int tx = (int)(tile % q.tiles_x);
    const long t2 = tile / q.tiles_x;
    const int ty = (int)(t2 % q.tiles_y), img = (int)(t2 / q.tiles_y);
    const int y0 = ty * TH, x0 = tx * kLkTW;
    const int CC = p.Cin >> 5;
    const float* ximg = p.x + (long)img * p.H * p.W * p.x_ld;

    // ---- per-lane halo pixel of each of this wave's PF pixel fragments at tap (0, 0): tile fragment wave + 4 f = (row, 16-column half)
    int hp0[PF];
#pragma clang loop unroll(full)
    for (int f = 0; f < PF; ++f) hp0[f] = ((wave + 4 * f) >> 1) * HWD + ((wave + 4 * f) & 1) * 16 + n16;

    f32x4 acc[PF][NF];   // [pixel fragment][cout fragment]
#pragma clang loop unroll(full)
    for (int f = 0; f < PF; ++f)
#pragma clang loop unroll(full)
        for (int nf = 0; nf < NF; ++nf) acc[f][nf] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // weights of (tap, chunk): every wave loads the NF x 3 fragments it multiplies with straight into registers (IGEMM_W_X6 order: one coalesced 1 KB row per
    // fragment and plane, the same addresses for all waves and workgroups -> L1 / L2 hits), one tap ahead of their use.  (A first version passed them through a
    // triple-buffered LDS stage with a workgroup barrier per tap: same speed to 1 %, and the barrier tied the four waves together at every tap.)
    const uint4* wsrc = reinterpret_cast<const uint4*>(p.w) + lane;
    auto w_load = [&](lk_u32x4 (&wf)[NF][3], int tap, int c) {
        const long kc = (long)tap * CC + c;
#pragma clang loop unroll(full)
        for (int nf = 0; nf < NF; ++nf) {
            const int nfg = min(nf, q.nfrag_alloc - 1);
#pragma clang loop unroll(full)
            for (int pl = 0; pl < 3; ++pl) {
                const uint4 v = wsrc[((long)nfg * p.KC + kc) * 192 + pl * 64];
                wf[nf][pl] = (lk_u32x4){v.x, v.y, v.z, v.w};
            }
        }
    };
    // one chunk of the halo: global f32 -> three bf16 planes in LDS
    auto stage_halo = [&](int c) {
#pragma clang loop unroll(full)
        for (int half = 0; half < 2; ++half) {
            float4 v[NSH];
#pragma clang loop unroll(full)
            for (int u = 0; u < NSH; ++u) {
                const int i = tid + kLkThreads * (half * NSH + u);
                const int hp = i >> 3, qd = i & 7;
                const int hy = hp / HWD, hx = hp - hy * HWD;
                const int iy = y0 - PAD + hy, ix = x0 - PAD + hx;
                const bool ok = (half * NSH + u) < NSL && i < NQ && (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
                v[u] = ok ? *reinterpret_cast<const float4*>(ximg + ((long)iy * p.W + ix) * p.x_ld + c * 32 + qd * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma clang loop unroll(full)
            for (int u = 0; u < NSH; ++u) {
                const int i = tid + kLkThreads * (half * NSH + u);
                if ((half * NSH + u) >= NSL || i >= NQ) continue;
                const int hp = i >> 3, qd = i & 7;
                const float f[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                unsigned hb[4], mb[4], lb[4];
#pragma clang loop unroll(full)
                for (int e = 0; e < 4; ++e) {
                    const unsigned ux = __float_as_uint(f[e]);
                    const float r1 = f[e] - __uint_as_float(ux & 0xFFFF0000u);
                    const unsigned u1 = __float_as_uint(r1);
                    const float r2 = r1 - __uint_as_float(u1 & 0xFFFF0000u);
                    hb[e] = ux; mb[e] = u1; lb[e] = __float_as_uint(r2);
                }
                const unsigned off = (unsigned)(hp * 64 + ((((qd >> 1) ^ (hp >> 2)) & 3) << 4) + (qd & 1) * 8);
                lk_lds_w2(off, (lk_u32x2){__builtin_amdgcn_perm(hb[1], hb[0], 0x07060302u), __builtin_amdgcn_perm(hb[3], hb[2], 0x07060302u)});
                lk_lds_w2(off + PLB, (lk_u32x2){__builtin_amdgcn_perm(mb[1], mb[0], 0x07060302u), __builtin_amdgcn_perm(mb[3], mb[2], 0x07060302u)});
                lk_lds_w2(off + 2 * PLB, (lk_u32x2){__builtin_amdgcn_perm(lb[1], lb[0], 0x07060302u), __builtin_amdgcn_perm(lb[3], lb[2], 0x07060302u)});
            }
        }
    };

    constexpr int WP[6] = {1, 2, 0, 1, 0, 0};   // (w plane, x plane) = mm, lh, hl, mh, hm, hh: smallest terms first
    constexpr int XP[6] = {1, 0, 2, 0, 1, 0};
    auto x_read = [&](lk_u32x4 (&dst)[3], int f, int tapoff) {
        const int hp = hp0[f] + tapoff;
        const unsigned a = (unsigned)(hp * 64 + (((g ^ (hp >> 2)) & 3) << 4));
#pragma clang loop unroll(full)
        for (int pl = 0; pl < 3; ++pl) dst[pl] = lk_lds4(a + (unsigned)pl * PLB);
    };

    auto tap_mfma = [&](const lk_u32x4 (&wf)[NF][3], int tap) {
        const int dy = tap / KS, dx = tap - dy * KS;
        const int tapoff = dy * HWD + dx;
        lk_u32x4 xf[2][3];
        x_read(xf[0], 0, tapoff);
#pragma clang loop unroll(full)
        for (int f = 0; f < PF; ++f) {
            if (f + 1 < PF) x_read(xf[(f + 1) & 1], f + 1, tapoff);   // the next fragment's reads overlap this one's 6 NF MFMAs
#pragma clang loop unroll(full)
            for (int t = 0; t < 6; ++t)
#pragma clang loop unroll(full)
                for (int nf = 0; nf < NF; ++nf)   // the cout fragments alternate: no back-to-back MFMAs on one accumulator
                    acc[f][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(lk_bf16x8, wf[nf][WP[t]]), __builtin_bit_cast(lk_bf16x8, xf[f & 1][XP[t]]), acc[f][nf], 0, 0, 0);
        }
    };
    constexpr int KK = KS * KS;
    lk_u32x4 wa[NF][3], wb[NF][3];
    w_load(wa, 0, 0);
    for (int c = 0; c < CC; ++c) {
        if (c > 0) __syncthreads();   // every wave is done with the previous chunk's halo
        stage_halo(c);
        __syncthreads();
        int tap = 0;
#pragma clang loop unroll(disable)
        for (; tap + 1 < KK; tap += 2) {
            w_load(wb, tap + 1, c);
            tap_mfma(wa, tap);
            if (tap + 2 < KK) w_load(wa, tap + 2, c);
            else if (c + 1 < CC) w_load(wa, 0, c + 1);
            tap_mfma(wb, tap + 1);
        }
        if (tap < KK) {   // odd tap count: the last tap runs from wa; the next chunk's first tap is requested behind it
            tap_mfma(wa, tap);
            if (c + 1 < CC) w_load(wa, 0, c + 1);
        }
    }

    // ---- epilogue: bias + residual + activation, one float4 per (pixel, 4 couts)
#pragma clang loop unroll(full)
    for (int f = 0; f < PF; ++f) {
        const int oy = y0 + ((wave + 4 * f) >> 1), ox = x0 + ((wave + 4 * f) & 1) * 16 + n16;
        if (oy >= p.H || ox >= p.W) continue;
        const long pix = ((long)img * p.H + oy) * p.W + ox;
#pragma clang loop unroll(full)
        for (int nf = 0; nf < NF; ++nf) {
            const int co = nf * 16 + g * 4;
            if (co >= p.gemm_cout) continue;
            float o[4] = {acc[f][nf][0], acc[f][nf][1], acc[f][nf][2], acc[f][nf][3]};
            if (p.bias) { const float4 b = *reinterpret_cast<const float4*>(p.bias + co); o[0] += b.x; o[1] += b.y; o[2] += b.z; o[3] += b.w; }
            if (p.res) { const float4 r = *reinterpret_cast<const float4*>(p.res + pix * p.y_ld + co); o[0] += r.x; o[1] += r.y; o[2] += r.z; o[3] += r.w; }
#pragma clang loop unroll(full)
            for (int e = 0; e < 4; ++e) o[e] = apply_act(o[e], p.act, p.alpha, p.beta);
            *reinterpret_cast<float4*>(p.y + pix * p.y_ld + co) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}
}  // namespace

// k x k / stride 1 / pad k / 2 / dilation 1 convolutions the LDS-tiled kernel takes: k = 9 with Cout <= 64 (LK-PAN), k = 5 with Cout <= 32 (one 32-channel group
// of SVTRv2's local mixing; Cin is then the group's 32 channels), 32 | Cin, float4 channel groups; OAR_IGEMM_LK=0 keeps them on the output-stationary kernel.
// Tile rows: the one of 12 / 8 / 6 that wastes the fewest rows of the map (k = 9: 8, the halo of a 12-row tile would not fit LDS).
static int lk_tile_rows(int ks, int H) {
    if (ks == 9) return 8;
    int best = 8, waste = 1 << 30;
    static const int pref = [] { const char* e = getenv("OAR_IGEMM_LK5_TH"); return e ? atoi(e) : 0; }();   // (A/B knob: force 6 / 8 / 12)
    if (pref == 6 || pref == 8 || pref == 12) return pref;
    for (int th : {6, 8, 12}) { const int w = (H + th - 1) / th * th - H; if (w < waste) { waste = w; best = th; } }   // ties: the smaller tile (two workgroups per CU)
    return best;
}
bool conv_lk_x6_eligible(int kh, int kw, int sh, int sw, int pt, int pl, int dh, int dw, int H, int W, int Ho, int Wo, int Cin, int Cout, int y_ld, long M) {
    static const bool on = [] { const char* e = getenv("OAR_IGEMM_LK"); return !e || atoi(e) != 0; }();
    const bool k9 = kh == 9 && kw == 9 && pt == 4 && pl == 4 && Cout <= 64;
    // k = 5 / one 32-channel group (OAR_IGEMM_LK5=0: back to the output-stationary kernel).  With the weights through an LDS stage and 12-row tiles (one workgroup
    // per CU) it measured EQUAL to that kernel (157 against 146-149 us per group at 319 488 pixels: 25 taps of one chunk do not amortise a halo staging nothing
    // overlaps); with the weights in registers and 6-row tiles -- 69 KB of LDS, two workgroups per CU, one stages while the other multiplies -- 57 against 77 us
    // at 159 744 pixels
    static const bool k5_on = [] { const char* e = getenv("OAR_IGEMM_LK5"); return !e || atoi(e) != 0; }();
    const bool k5 = k5_on && kh == 5 && kw == 5 && pt == 2 && pl == 2 && Cout <= 32 && Cin == 32;
    if (!(on && (k9 || k5) && sh == 1 && sw == 1 && dh == 1 && dw == 1 && Ho == H && Wo == W && (Cin & 31) == 0 && Cin >= 32 &&
          (Cout & 3) == 0 && (y_ld & 3) == 0 && W >= 16 && H >= 4 && M >= 2048 && (long)H * W * Cin < (1L << 31))) return false;
    // one workgroup per CU (129 ... 156 KB of LDS): below ~half a chip of tiles the per-tile kernels, which spread a small map over every CU, are faster
    // (8 x 40 x 40 pixels = 80 tiles of the 9 x 9 kernel: 0.73 ms here against 0.53 ms on the f32 per-tile kernel)
    const int th = lk_tile_rows(kh, H);
    return (M / ((long)H * W)) * ((H + th - 1) / th) * ((W + kLkTW - 1) / kLkTW) >= 128;
}

template <int KS, int TH, int NF>
static void launch_lk(hipStream_t s, const IgemmP& p, int n_images) {
    LkP q;
    q.g = p;
    q.N = n_images;
    q.tiles_x = (p.W + kLkTW - 1) / kLkTW;
    q.tiles_y = (p.H + TH - 1) / TH;
    q.tiles = (long)n_images * q.tiles_x * q.tiles_y;
    q.per_xcd = (q.tiles + 7) / 8;
    q.nfrag_alloc = (p.gemm_cout + 63) / 64 * 4;
    const size_t lds = (size_t)3 * (kLkTW + KS - 1) * (TH + KS - 1) * 64;
    OAR_CHECK(lds <= 160 * 1024, OAR_INTERNAL, "conv_lk_x6: tile does not fit LDS");
    OAR_MAX_LDS_ONCE((conv_lk_x6_kernel<KS, TH, NF>), 160 * 1024);
    hipLaunchKernelGGL((conv_lk_x6_kernel<KS, TH, NF>), dim3((unsigned)(q.per_xcd * 8)), dim3(kLkThreads), lds, s, q);
}

void conv_lk_x6(hipStream_t s, const IgemmP& p, int n_images) {
    if (p.kh == 9) return launch_lk<9, 8, 4>(s, p, n_images);
    OAR_CHECK(p.kh == 5 && p.Cin == 32 && p.gemm_cout <= 32, OAR_INTERNAL, "conv_lk_x6: not a shape conv_lk_x6_eligible accepts");
    switch (lk_tile_rows(5, p.H)) {
        case 12: return launch_lk<5, 12, 2>(s, p, n_images);
        case 6: return launch_lk<5, 6, 2>(s, p, n_images);
        default: return launch_lk<5, 8, 2>(s, p, n_images);
    }
}

}  // namespace k
}  // namespace oar
