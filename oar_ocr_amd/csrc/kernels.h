// kernels.h -- launchers for the gfx950 NN kernels (all activations f32, NHWC for rank-4 feature maps).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

namespace oar {
namespace k {

enum ActKind : int { ACT_NONE = 0, ACT_RELU, ACT_HSWISH, ACT_HSIGMOID, ACT_SIGMOID, ACT_SWISH, ACT_LEAKY, ACT_CLIP, ACT_TANH, ACT_GELU_ERF,
                     // element-wise math (decomposed GELU / LayerNorm exports, UVDoc)
                     ACT_ERF, ACT_SQRT, ACT_EXP, ACT_ABS, ACT_NEG, ACT_RECIP, ACT_LOG, ACT_GELU_TANH, ACT_SOFTPLUS, ACT_FLOOR, ACT_CEIL, ACT_ROUND, ACT_NOT };
struct Act {
    int kind = ACT_NONE;
    float alpha = 0.f, beta = 0.f;  // HardSigmoid(alpha,beta) / LeakyRelu(alpha) / Clip(alpha=min,beta=max)
};

struct ConvP {
    int N, H, W, Cin;       // input NHWC
    int Ho, Wo, Cout;       // output NHWC
    int kh, kw, sh, sw, pt, pl, dh, dw, groups;
    Act act;
    const float* x;
    const float* w;         // layout depends on the kernel (see each launcher)
    const float* bias;      // may be null
    const float* residual;  // may be null; same shape/ld as y
    float* y;
    int y_ld;               // channel stride of the output buffer (>= Cout); y points at channel offset already
    int convt2x2;           // igemm only: output scatter of a 2x2/stride-2 ConvTranspose (Cout is the real Cout)
    int w_fmt;              // igemm only: weight fragment layout, one of IGEMM_W_* (chosen by igemm_weight_format)
    float* ctc_part;        // igemm only, Linear feeding the fused CTC tail: softmax partials [rows][ctc_tiles()] float4
    int ctc_valid;          //   instead of logits (y is not written); ctc_valid = number of real classes
    const float* se;        // igemm only (bf16x6 weight-stationary 1x1): squeeze-excite gate [N][Cin] multiplied into the input on load
    int res_up;             // igemm only (f32 per-tile kernels): f > 1 => `residual` is [N][Ho / f][Wo / f][Cout], added through the nearest index map (h / f, w / f)
                            // -- the top-down sum of an FPN without materialising the upsampled tensor
    int n_msrc;             // igemm only, output-stationary bf16x6 kernel, 1x1 layers: > 0 => the input is the channel concat of n_msrc channels-last maps that was never
    const float* msrc[8];   //   materialised (PP-HGNetV2's 7-way aggregation): source i has msrc_c[i] channels (8 | msrc_c[i]) at msrc[i], pixel stride msrc_c[i];
    int msrc_c[8];          //   x is ignored, Cin = the sum
    int accum;              // igemm only, row-streaming 3x3 kernel: 1 = add to what y already holds (a later pass over a channel slice of the input, ConvP::x_ld)
    int x_ld;               // igemm only, output-stationary bf16x6 kernel / row-streaming 3x3 kernel: floats between two pixels of x when that is not Cin (0 = Cin): one group of a
                            // grouped convolution reads its Cin channels out of the full tensor
    float* gap_part;        // depthwise only (conv_dw_gap_tiles(p) > 0): per-tile sums of the activated output, [N][tiles][Cout] -- the squeeze of an
                            // SE block without a second read of the feature map (global_avgpool_finish reduces them)
};
enum : int { IGEMM_W_K16 = 0, IGEMM_W_X6 = 1, IGEMM_W_X6RS = 2, IGEMM_W_X6CS = 3 };   // X6CS: per-chunk weight blocks of dsblock_cs.inc (depthwise taps + bias + pointwise pieces)
//   // X6RS: the bf16x6 fragments of dsblock_rs.inc (k-step = two 16-channel chunks)

// Implicit-GEMM conv on the matrix cores. groups == 1, Cin % 4 == 0. Weight layouts (ConvP::w_fmt):
//   K16: f32 fragments  Wf[cout/16][K/16][lane][4 f32]              -> v_mfma_f32_16x16x4_f32 (exact f32 FMA chain)
//   X6 : bf16x6 fragments Wx[cout/16][K/32][3 planes][lane][8 bf16] -> v_mfma_f32_16x16x32_bf16 x 6 (f32-equivalent),
//        weight-stationary kernel only (wide 1x1 / Linear layers)
// Rows are padded to 64 couts, K to the chunk size, with zeros.
void conv_igemm(hipStream_t s, const ConvP& p);
// chosen per layer AND input shape at plan time: M = GEMM rows (pixels), N = couts
// same3x3_px: pixels of ONE image when the layer is a 3x3 / stride 1 / pad 1 / dilation 1 convolution without residual (0 otherwise): the row-streaming
// bf16x6 kernel of igemm_rs3_x6.hip takes such layers with <= 16 output channels
int igemm_weight_format(long M, int K, int N, bool is1x1, int Cin = 0, long same3x3_px = 0, bool lk_ok = false);
// lk_ok: conv_lk_x6_eligible said yes for this layer (large-kernel same convolution from an LDS-staged halo tile, igemm_lk_x6.hip)
bool conv_lk_x6_eligible(int kh, int kw, int sh, int sw, int pt, int pl, int dh, int dw, int H, int W, int Ho, int Wo, int Cin, int Cout, int y_ld, long M);
// one group of a grouped k x k convolution (ConvP::x_ld) as its own implicit GEMM on the output-stationary bf16x6 kernel: M pixels, K = kh * kw * Cin_g, N = Cout_g >= 32
bool conv_grouped_x6_ok(long M, int K, int N, int Cin);
// may a 1x1 layer of this shape read a never-materialised channel concat (ConvP::msrc)?  It then runs on the output-stationary bf16x6 kernel whatever its K
bool conv_msrc_ok(long M, int K, int N);
// a 3x3 / stride 1 / pad 1 convolution with <= 16 output channels and 96 ... 256 input channels as passes of the row-streaming bf16x6 kernel over 64- / 32-channel
// slices of the input (ConvP::x_ld, ConvP::accum): the slice widths, empty when the layer is not eligible.  img_px = pixels of one image
std::vector<int> conv3x3_n16_slices(long M, int Cin, int Cout, long img_px, int y_ld);
// may a 1x1 conv of this shape take ConvP::se (the gate of a squeeze-excite block folded into its input load)?  hw = pixels per image
bool conv_igemm_se_ok(long M, int K, int N, int hw);   // Cin: input channels of a k x k conv (0: treat as not eligible for the x6 path)
// Depthwise conv. w: [kh][kw][C]. C % 4 == 0.
void conv_dw(hipStream_t s, const ConvP& p);
// tiles per image of the variant conv_dw will launch for p when it can also emit ConvP::gap_part (0: it cannot -- pool separately)
int conv_dw_gap_tiles(const ConvP& p);
// mean over the feature map from conv_dw's per-tile sums: y[n][c] = (sum over tiles of part[n][t][c]) / hw, fixed order
void global_avgpool_finish(hipStream_t s, const float* part, float* y, int N, int tiles, int C, int hw);
// Direct conv for everything else (small Cin, odd channels, grouped). w: [kh][kw][Cin/g][Cout].
void conv_direct(hipStream_t s, const ConvP& p);
// RGB stem with the page normalisation folded in: the stem reads the u8 HWC pages themselves and computes
// (float)v * alpha[c] + beta[c] for tensor channel c = page channel src[c] on the fly -- the same two f32 operations
// pp::normalize performs, so the result is bit-identical to normalize + conv -- instead of a 12-bytes-per-pixel f32 tensor
// being written and read back.  Up to 32 separately allocated pages per launch.
// `dev` == nullptr: st.pages[0..n) are H x W x 3 pages of one size, normalised as x * alpha + beta (the detector).  `dev` != nullptr
// (the recognizer): image i is dev[i].ptr, H rows of dev[i].w <= W pixels -- columns past its own width read as zero AFTER
// normalisation, which is how the CRNN input is padded (crnn.rs:98-121) -- and a byte v becomes ((v / 255 - 0.5) / 0.5), the
// recognizer's expression, through a 256-entry table.
struct StemImg { const uint8_t* ptr; int32_t w; int32_t pad; };
struct StemU8 { const uint8_t* pages[32]; int src[3]; float alpha[3], beta[3]; const StemImg* dev = nullptr; };
void conv_smallcin_u8(hipStream_t s, const ConvP& p, const StemU8& st);
// Two stacked ConvTranspose 2x2 / stride 2 layers as one kernel -- the tail of the DB head (C/4 -> C/4 -> 1 channels): the
// intermediate map, 4x the input's pixels, never leaves the registers.  x: [N, H, W, C0] NHWC; w1: [C0][4][C1] (4 = a * 2 + b of
// the first layer), w2: [C1][4][C2]; y: [N, 4H, 4W, C2] NHWC.  C0, C1 % 4 == 0, C1 <= 32, C2 <= 4.
struct ConvT2Pair {
    const float* x; float* y;
    const float *w1, *b1, *w2, *b2;   // biases may be null
    int N, H, W, C0, C1, C2;
    Act act1, act2;
};
bool convt2x2_pair_supported(int C0, int C1, int C2);
void convt2x2_pair(hipStream_t s, const ConvT2Pair& p);
// General ConvTranspose (gather form). w: [kh][kw][Cin][Cout] (groups == 1), output_padding folded in Ho/Wo.
void convt_direct(hipStream_t s, const ConvP& p);

struct PoolP {
    int N, H, W, C, Ho, Wo, kh, kw, sh, sw, pt, pl;
    int pb, pr;              // bottom / right padding: the divisor of count_include_pad counts padding, not the ceil_mode overhang beyond it
    int is_max, count_include_pad;
    const float* x;
    float* y;
};
void pool2d(hipStream_t s, const PoolP& p);
// partial: scratch of N * global_avgpool_splits(N, HW, C) * C floats (may be null when the split count is 1)
int global_avgpool_splits(int N, int HW, int C);
void global_avgpool(hipStream_t s, const float* x, float* y, int N, int HW, int C, float* partial = nullptr);

// Resize NHWC. mode 0 = nearest, 1 = bilinear. ctm: 0 = asymmetric, 1 = half_pixel, 2 = align_corners,
// 3 = pytorch_half_pixel. nearest_mode: 0 = floor, 1 = round_prefer_floor, 2 = round_prefer_ceil, 3 = ceil.
// source index of output index o for mode = nearest (host evaluation of the kernel's own index map)
int resize_nearest_index(int o, float scale, int in, int out, int ctm, int nearest_mode);
// y = a op nearest_upsample(b), integer factors (fh, fw), NHWC, C % 4 == 0: b is [N, Ho / fh, Wo / fw, C]
void binary_upsampled(hipStream_t s, const float* a, const float* b, float* y, int N, int Ho, int Wo, int C, int fh, int fw, int op);
void resize(hipStream_t s, const float* x, float* y, int N, int H, int W, int C, int Ho, int Wo, float scale_h,
            float scale_w, int mode, int ctm, int nearest_mode, int y_ld);
// Channel concat of channels-last maps in ONE launch: source i contributes c[i] channels (c[i] % 4 == 0) at channel offset off[i], read at
// pixel (oh / fh[i], ow / fw[i]) of its [N][Ho / fh][Wo / fw][c] map (integer-factor nearest upsampling with coordinate mode asymmetric /
// floor; f = 1: a plain copy).  The DB neck's Concat(up8(p5), up4(p4), up2(p3), p2) was three resize launches and a copy.
struct ConcatGatherP { const float* x[8]; int c[8], off[8], fh[8], fw[8]; int n_src, N, Ho, Wo, C; };
void concat_gather(hipStream_t s, const ConcatGatherP& p, float* y);

void unary(hipStream_t s, const float* x, float* y, int64_t n, Act act);
// y = op(a, b) with numpy broadcasting over up to 6 dims. op: 0 add, 1 sub, 2 mul, 3 div, 4 pow, 5 prelu, 6 max, 7 min,
// 8 equal, 9 less, 10 greater, 11 and, 12 or (comparisons / logic give 1.0f or 0.0f: bool tensors live as f32 on the device). Strides in elements
// (0 for broadcast dims); output is contiguous with dims `dims`.
void binary(hipStream_t s, const float* a, const float* b, float* y, int op, int rank, const int64_t* dims,
            const int64_t* sa, const int64_t* sb, Act post);
// Copies `rows` rows of `c` floats from x (ld x_ld) to y (ld y_ld): channel concat / strided views.
void copy2d(hipStream_t s, const float* x, float* y, int64_t rows, int c, int x_ld, int y_ld);
// Generic permute: y (contiguous, dims out_dims) = x indexed with in_strides (already permuted), rank <= 6.
void permute(hipStream_t s, const float* x, float* y, int rank, const int64_t* out_dims, const int64_t* in_strides);

// Batched GEMM, row-major: C[b] (MxN) = alpha * A[b] (MxK) * B[b] (KxN or NxK if transB) (+ bias[N]) (+ residual) then act.
// batch strides in elements (0 = shared). VALU-tiled kernel for the small attention products.
struct GemmP {
    int batch, M, N, K, transB;
    int64_t sA, sB, sC;
    float alpha;
    const float *A, *B, *bias, *residual;
    float* C;
    Act act;
};
void gemm_batched(hipStream_t s, const GemmP& p);

void layernorm(hipStream_t s, const float* x, const float* gamma, const float* beta, float* y, int64_t rows, int C, float eps);
void softmax_lastdim(hipStream_t s, const float* x, float* y, int64_t rows, int C);
// squeeze-excite gate on a pooled vector: y[n] = act2(W2 act1(W1 x[n] + b1) + b2); w1 = W1 [Cmid][C], w2 = W2 TRANSPOSED [Cmid][Cout]
// tiles > 0: x is not the pooled vector but the per-tile channel sums a pooling depthwise conv left ([N][tiles][C], conv_dw with gap_part);
// the kernel reduces them itself, in global_avgpool_finish's order, and divides by hw -- the squeeze costs no launch of its own
void se_fc(hipStream_t s, const float* x, const float* w1, const float* b1, Act act1, const float* w2, const float* b2, Act act2, float* y, int N, int C,
           int Cmid, int Cout, int tiles = 0, int hw = 0);
// ONNX Pad on a contiguous tensor of rank <= 6: out_dims[d] = in_dims[d] + before[d] + after[d] (negative = crop);
// mode 0 constant (value), 1 reflect, 2 edge
void pad_nd(hipStream_t s, const float* x, float* y, int rank, const int64_t* in_dims, const int64_t* out_dims, const int64_t* before, int mode, float value);
// reduction over the last axis: x [rows][C] -> y [rows]; mode 0 mean, 1 sum, 2 max, 3 min, 4 prod
void reduce_lastdim(hipStream_t s, const float* x, float* y, int64_t rows, int C, int mode);
void argreduce_lastdim(hipStream_t s, const float* x, float* y, int64_t rows, int C, bool is_min, bool select_last);
inline void reduce_mean_lastdim(hipStream_t s, const float* x, float* y, int64_t rows, int C) { reduce_lastdim(s, x, y, rows, C, 0); }
// y = cond != 0 ? a : b with numpy broadcasting over up to 6 dims (strides in elements, 0 = broadcast)
void where(hipStream_t s, const float* cond, const float* a, const float* b, float* y, int rank, const int64_t* dims, const int64_t* sc, const int64_t* sa, const int64_t* sb);
// ONNX GridSample (4-D): x [N][H][W][C] channels-last, grid [N][Ho][Wo][2] (x, y in [-1, 1]) -> y [N][Ho][Wo][C].
// mode 0 bilinear / 1 nearest; padding 0 zeros / 1 border / 2 reflection
void grid_sample(hipStream_t s, const float* x, const float* grid, float* y, int N, int H, int W, int C, int Ho, int Wo, int mode, int padding, int align_corners);
// softmax(scale * q k^T) v per (image, head); qkv [n][T][3][heads][hd] row-major, out [n][T][heads][hd]; hd <= 64
void attention(hipStream_t s, const float* qkv, float* out, int n, int T, int heads, int hd, float scale);
// attention_x6.hip: the same operator flash-style on the bf16 matrix pipe (bf16x6 products), any T, head dim 32 -- what `attention` launches for
// hd == 32 unless OAR_ATTN_X6=0; attention_fits(T, hd): can `attention` run this shape at all (either kernel)?
bool attention_x6_supported(int T, int heads, int hd);
void attention_x6(hipStream_t s, const float* qkv, float* out, int n, int T, int heads, int hd, float scale);
bool attention_fits(int T, int heads, int hd);
// A run of sample-local operators as one launch (chain.hip): one workgroup per sample walks the table.  Every tensor is a row-major
// [n_samples * T rows][ld floats] view; sample s owns rows [s * T, (s + 1) * T).
enum ChainType : int { CH_GEMM = 0, CH_LN = 1, CH_ATTN = 2, CH_COPY = 3, CH_POOL = 4 };   // CH_POOL: AveragePool K x cin (rows x columns per token) that leaves one row: in = the sample's [K][pad][N] map, pad = its width >= cin * T
struct ChainRef { unsigned long long v = 0; int kind = -1; };   // kind: -1 none, 0 absolute device pointer, 1 arena-relative byte offset, 2 primary-input-relative
struct ChainOpD {
    int type = 0;
    int K = 0, N = 0;            // GEMM: reduction length (taps * cin) / output channels; LN, COPY: N = row length
    int cin = 0, pad = 0;        // GEMM over a 1 x k convolution: k = K / cin taps, input row t + tap - pad (zero outside the sample)
    int ksplit = 1;              // GEMM: K slices reduced through LDS
    int mb = 1;                  // GEMM: token tiles (of 16 rows) one work item covers, 1..3 (4 would spill)
    int in_ld = 0, out_ld = 0, res_ld = 0;
    int act = 0; float alpha = 0.f, beta = 0.f;
    float eps = 0.f, scale = 0.f;   // LN epsilon / attention scale
    int heads = 0, hd = 0;
    ChainRef in, out, res;
    const float* w = nullptr;    // GEMM: [N][K] f32, K index = tap * cin + c (inside the chain's constant blob)
    int bias_l = -1;             // GEMM bias [N] / LN beta: float offset in LDS (the kernel copies the small constants there), -1 = none
    int w_l = -1;                // LN gamma: float offset in LDS, -1 = none
};
static_assert(sizeof(ChainOpD) % 4 == 0, "ChainOpD is copied to LDS word by word");
struct ChainLaunch {
    const ChainOpD* ops = nullptr; int n_ops = 0, n_samples = 0, T = 0, max_hd = 0;
    size_t lds = 0;              // dynamic LDS: split-K scratch | resident tensors | small constants | operator table
    const float* consts = nullptr; int small = 0, total_consts = 0, small_l = 0;   // constant blob (floats): [0, small) -> LDS at float offset small_l
    int tab_l = 0;               // float offset in LDS of the operator table copy
    double bytes = 0, flops = 0;
};
constexpr size_t kChainLdsBudget = 159 * 1024;   // LDS a chain may plan with (the launch adds 640 bytes of debug stamps)
constexpr int kChainMaxHd = 16;   // attention head size the chain kernel is instantiated for (a 32-wide variant spills: wider heads stay unfused)
inline bool chain_act_ok(int kind) { return kind == ACT_NONE || kind == ACT_RELU || kind == ACT_HSWISH || kind == ACT_HSIGMOID || kind == ACT_SIGMOID || kind == ACT_SWISH; }
size_t chain_lds_bytes(const ChainOpD& op, int T);
void chain_run(hipStream_t s, const ChainLaunch& L, char* arena, const char* input);
// softmax over the last dim fused with CTC argmax (last max index wins) -- see kernels.hip
// CTC head without logits: conv_igemm with ConvP::ctc_part set writes {max, sum exp, last arg max} per (row, cout tile
// of 128 columns); ctc_combine merges the tiles of each row into the arg max index and its softmax probability.
int ctc_tiles(int n_padded);   // cout tiles per row for a (16-padded) class count
bool ctc_partials_supported(int K);
bool ctc_partials_supported_x6(int K);
void ctc_combine(hipStream_t s, const float* part, int64_t rows, int tiles, int64_t* idx, float* prob);
// ctc_head_x6.hip: the CTC head as an output-stationary bf16x6 kernel (K = 32 / 64; IGEMM_W_X6 weights; same partials as above)
bool ctc_head_x6_supported(long M, int K, int n_padded);
void ctc_head_x6(hipStream_t s, const float* x, const float* w_x6, const float* bias, float* part, long M, int K, int n_padded, int valid);
void softmax_argmax(hipStream_t s, const float* logits, int64_t rows, int C, int ld, int64_t* idx, float* prob);   // ld = row stride

}  // namespace k
}  // namespace oar
