// ctc_head_x6.hip -- the CTC head (Linear K -> V whose logits only feed the fused softmax / arg max tail) as an OUTPUT-STATIONARY bf16x6 kernel
// (round 5; VERDICT r4 #5b).  Replaces, for K <= 64, conv_igemm_ws_kernel<8, 1, true>: 123 us per recognition batch (10 240 rows x 6 912 columns x
// K = 64) on the f32 matrix instruction -- 128 v_mfma_f32_16x16x4_f32 of 32 clocks per (16 rows x 128 columns) tile, 65 % issue stall.
//
//   * A wave owns 2 x 16 ROWS (time steps) for the whole launch: their K activations are loaded once, split exactly into three bf16 pieces and kept
//     in registers as the B operands of v_mfma_f32_16x16x32_bf16 (KC x 3 x 4 registers).
//   * A workgroup is 12 waves x 2 row tiles = 384 rows and walks a RANGE of 128-column tiles (the softmax-partial tile of ctc_combine: 8 cout fragments).  A tile's
//     weights -- 8 fragments x KC k-steps x 3 planes x 1 KB, contiguous in IGEMM_W_X6 order -- arrive by LDS-DMA into a double buffer, three
//     pieces per wave, while the previous tile is multiplied: the 1.7 MB weight matrix is streamed once per workgroup from L2, never re-staged per
//     row tile.  One workgroup barrier per tile.
//   * Per tile and wave: 8 x KC x 6 MFMAs (the six significant products, smallest first) into 8 accumulators that start from the bias (LDS), then
//     igemm_ctc_epilogue's arithmetic: per row max / sum exp / LAST arg max over the tile's valid columns -> one float4 of partials.  With four
//     waves per SIMD one wave's epilogue (32 v_exp_f32 + compares) runs under the others' MFMAs.
//   * Grid: ceil(M / 384) row blocks x column splits so that ~240 workgroups exist; no logits ever reach HBM.
// Algorithmic work: 2 M K V flops; bytes 4 M K (activations) + 6 K V per workgroup row block (weights, L2) + 16 M V / 128 (partials).
#include <hip/hip_ext.h>

#include "common.h"
#include "igemm_dev.h"
#include "kernels.h"

namespace oar {
namespace k {

namespace {
typedef unsigned ch_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 ch_bf16x8 __attribute__((ext_vector_type(8)));

struct CtcHeadP {
    const float* x; const float4* w; const float* bias; float* part;
    long M; int K; int valid; int ny; int tiles_per_wg; unsigned w_bytes; int n_padded;
};

__device__ __forceinline__ void ch_dma(unsigned voff, ch_u32x4 rsrc, unsigned soff, unsigned lds_dst) {
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(soff), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ ch_u32x4 ch_lds4(unsigned off) {
    return *reinterpret_cast<const __attribute__((address_space(3))) ch_u32x4*>((__attribute__((address_space(3))) const char*)nullptr + off);
}

constexpr int kChWaves = 12;   // waves per workgroup (three per SIMD: 170 registers each)
constexpr int kChRT = 2;       // 16-row tiles per wave: every weight fragment read from LDS feeds kChRT MFMAs (at one tile per wave the kernel ran at the LDS read rate:
                               // 512 B of A operand per MFMA x 4 SIMDs / 16.5 clocks = 124 B / clock / CU)

template <int KC>
__global__ __launch_bounds__(kChWaves * 64, 1) void ctc_head_x6_kernel(CtcHeadP p) {
    constexpr int TILE = 8 * KC * 3 * 1024;          // bytes of one tile's weights
    constexpr int NPIECE = TILE / 1024;              // DMA pieces per tile
    constexpr int PER_WAVE = (NPIECE + kChWaves - 1) / kChWaves;
    constexpr unsigned BIAS0 = 2u * TILE;            // + this workgroup's bias range [tiles_per_wg][128] f32
    extern __shared__ float4 ch_lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, pl = lane & 15;
    const int t0 = (int)blockIdx.x * p.tiles_per_wg, t1 = min(p.ny, t0 + p.tiles_per_wg);
    if (t0 >= t1) return;
    const long row0 = (long)blockIdx.y * (kChWaves * kChRT * 16) + wave * (kChRT * 16) + pl;   // row of tile rt: row0 + 16 rt
    const unsigned long wbase = reinterpret_cast<unsigned long>(p.w);
    const ch_u32x4 wsrc = {(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)wbase), (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(wbase >> 32)) & 0xFFFFu, p.w_bytes, 0x00020000u};
    auto request = [&](int t, int buf) __attribute__((always_inline)) {   // this wave's pieces of tile t (pieces past the matrix read as zeros)
#pragma unroll
        for (int j = 0; j < PER_WAVE; ++j) {
            const int piece = j * kChWaves + wave;
            if (piece < NPIECE) ch_dma((unsigned)(lane * 16), wsrc, (unsigned)t * (unsigned)TILE + (unsigned)(piece * 1024), (unsigned)(buf * TILE + piece * 1024));
        }
    };
    request(t0, 0);
    // bias of the workgroup's columns (zeros past the padded width never matter: those columns are not valid)
    for (int i = tid; i < (t1 - t0) * 128; i += kChWaves * 64)
        reinterpret_cast<float*>(reinterpret_cast<char*>(ch_lds) + BIAS0)[i] = (p.bias && t0 * 128 + i < p.n_padded) ? p.bias[t0 * 128 + i] : 0.f;   // (the last tile may reach past the padded width)
    // ---- this wave's rows: lane (g, pl) holds k = 32 kc + 8 g .. + 7 of row pl, split exactly into three bf16 pieces (B operands)
    ch_u32x4 B[kChRT][KC][3];
#pragma unroll
    for (int rt = 0; rt < kChRT; ++rt)
#pragma unroll
    for (int kc = 0; kc < KC; ++kc) {
        const long row = row0 + 16 * rt;
        float v[8];
        const int k0 = kc * 32 + g * 8;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < p.M && k0 + h * 4 < p.K) q = *reinterpret_cast<const float4*>(p.x + row * p.K + k0 + h * 4);
            v[h * 4 + 0] = q.x; v[h * 4 + 1] = q.y; v[h * 4 + 2] = q.z; v[h * 4 + 3] = q.w;
        }
        unsigned hb[8], mb[8], lb[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const unsigned ux = __float_as_uint(v[e]);
            const float r1 = v[e] - __uint_as_float(ux & 0xFFFF0000u);
            const unsigned u1 = __float_as_uint(r1);
            const float r2 = r1 - __uint_as_float(u1 & 0xFFFF0000u);
            hb[e] = ux; mb[e] = u1; lb[e] = __float_as_uint(r2);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            B[rt][kc][0][q] = __builtin_amdgcn_perm(hb[2 * q + 1], hb[2 * q], 0x07060302u);
            B[rt][kc][1][q] = __builtin_amdgcn_perm(mb[2 * q + 1], mb[2 * q], 0x07060302u);
            B[rt][kc][2][q] = __builtin_amdgcn_perm(lb[2 * q + 1], lb[2 * q], 0x07060302u);
        }
    }
    // six products per (fragment, k-step), smallest terms first: (w plane, x plane) = mm, lh, hl, mh, hm, hh
    constexpr int WPL[6] = {1, 2, 0, 1, 0, 0}, XPL[6] = {1, 0, 2, 0, 1, 0};
    for (int t = t0; t < t1; ++t) {
        const int buf = (t - t0) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile t (and its partials of tile t - 1) are done
        __syncthreads();                                    // everybody's are, and everybody is through with the other buffer
        if (t + 1 < t1) request(t + 1, buf ^ 1);
        const unsigned wb = (unsigned)(buf * TILE) + (unsigned)lane * 16u, bb = BIAS0 + (unsigned)((t - t0) * 512 + g * 16);
        f32x4 acc[kChRT][8];
#pragma unroll
        for (int nf = 0; nf < 8; ++nf) {
            const ch_u32x4 b = ch_lds4(bb + (unsigned)(nf * 64));
#pragma unroll
            for (int rt = 0; rt < kChRT; ++rt) acc[rt][nf] = (f32x4){__uint_as_float(b[0]), __uint_as_float(b[1]), __uint_as_float(b[2]), __uint_as_float(b[3])};
        }
#pragma unroll
        for (int nf = 0; nf < 8; ++nf)
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                ch_u32x4 A[3];
#pragma unroll
                for (int s = 0; s < 3; ++s) A[s] = ch_lds4(wb + (unsigned)(((nf * KC + kc) * 3 + s) * 1024));
#pragma unroll
                for (int q = 0; q < 6; ++q)
#pragma unroll
                    for (int rt = 0; rt < kChRT; ++rt)
                        acc[rt][nf] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(ch_bf16x8, A[WPL[q]]), __builtin_bit_cast(ch_bf16x8, B[rt][kc][XPL[q]]), acc[rt][nf], 0, 0, 0);
            }
        // ---- softmax partials of the tile (the arithmetic of igemm_ctc_epilogue): lane (g, pl) holds columns t * 128 + nf * 16 + g * 4 + r of row pl
#pragma unroll
        for (int rt = 0; rt < kChRT; ++rt) {
            const long row = row0 + 16 * rt;
            float m = -3.402823466e38f;
#pragma unroll
            for (int nf = 0; nf < 8; ++nf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = t * 128 + nf * 16 + g * 4 + r;
                    if (c < p.valid) m = fmaxf(m, acc[rt][nf][r]);
                }
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            float sum = 0.f;
            int last = -1;
#pragma unroll
            for (int nf = 0; nf < 8; ++nf)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int c = t * 128 + nf * 16 + g * 4 + r;
                    if (c < p.valid) {
                        const float e = __expf(acc[rt][nf][r] - m);
                        sum += e;
                        if (e == 1.0f) last = c;   // columns ascend with (nf, r): the last hit is the largest
                    }
                }
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            last = max(last, __shfl_xor(last, 16, 64));
            last = max(last, __shfl_xor(last, 32, 64));
            if (g == 0 && row < p.M) reinterpret_cast<float4*>(p.part)[row * p.ny + t] = make_float4(m, sum, __int_as_float(last), 0.f);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

template <typename K>
void launch_ch(K kernel, hipStream_t s, const CtcHeadP& p, dim3 grid, size_t lds, hipEvent_t e0, hipEvent_t e1) {
    OAR_MAX_LDS_ONCE(kernel, 160 * 1024);
    hipExtLaunchKernelGGL(kernel, grid, dim3(kChWaves * 64), lds, s, e0, e1, 0, p);
}
}  // namespace

// K a multiple of 32 up to 96 (the tile's 8 x KC x 3 KB of weights, double-buffered, must fit LDS: 144 KB at K = 96, round 6); OAR_CTC_HEAD_OS=0 keeps the weight-stationary kernels
bool ctc_head_x6_supported(long M, int K, int n_padded) {
    static const bool on = [] { const char* e = getenv("OAR_CTC_HEAD_OS"); return !e || atoi(e) != 0; }();
    return on && M > 0 && (K == 32 || K == 64 || K == 96) && n_padded >= 128 && (n_padded & 15) == 0 && M * (long)ctc_tiles(n_padded) * 16 < (1L << 40);
}

void ctc_head_x6(hipStream_t s, const float* x, const float* w_x6, const float* bias, float* part, long M, int K, int n_padded, int valid) {
    CtcHeadP p{};
    p.x = x; p.w = reinterpret_cast<const float4*>(w_x6); p.bias = bias; p.part = part;
    p.M = M; p.K = K; p.valid = valid; p.ny = ctc_tiles(n_padded); p.n_padded = n_padded;
    const int KC = K / 32;
    const long rows64 = ((long)n_padded + 63) / 64 * 64;   // engine.cc to_fragment_x6 pads the rows (couts) to 64
    p.w_bytes = (unsigned)(rows64 / 16 * KC * 3 * 1024);
    const int row_blocks = (int)((M + kChWaves * kChRT * 16 - 1) / (kChWaves * kChRT * 16));
    int splits = std::max(1, std::min(p.ny, (240 + row_blocks - 1) / row_blocks));
    const int bias_cap = (int)((160 * 1024 - (size_t)2 * 8 * KC * 3 * 1024) / 512);   // the bias range of a workgroup lives in LDS behind the two weight buffers (K = 96: 32 tiles)
    p.tiles_per_wg = std::min((p.ny + splits - 1) / splits, std::min(96, bias_cap));
    splits = (p.ny + p.tiles_per_wg - 1) / p.tiles_per_wg;
    const size_t lds = (size_t)2 * 8 * KC * 3 * 1024 + (size_t)p.tiles_per_wg * 512;
    OAR_CHECK(lds <= 160 * 1024, OAR_INTERNAL, "ctc_head_x6: the column range of a workgroup does not fit LDS");
    const double flops = 2.0 * (double)M * K * n_padded, bytes = 4.0 * (double)M * K + 6.0 * (double)K * n_padded + 16.0 * (double)M * p.ny;
    ProfScope ps(s, "ctc_head_x6", bytes, flops, true);
    if (KC == 1) launch_ch(ctc_head_x6_kernel<1>, s, p, dim3((unsigned)splits, (unsigned)row_blocks), lds, ps.start(), ps.stop());
    else if (KC == 3) launch_ch(ctc_head_x6_kernel<3>, s, p, dim3((unsigned)splits, (unsigned)row_blocks), lds, ps.start(), ps.stop());
    else launch_ch(ctc_head_x6_kernel<2>, s, p, dim3((unsigned)splits, (unsigned)row_blocks), lds, ps.start(), ps.stop());
}

}  // namespace k
}  // namespace oar
