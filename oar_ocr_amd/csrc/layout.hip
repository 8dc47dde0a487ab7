// layout.hip -- kernels of the layout-detection path (layout.h): filtered resize with host taps, LayoutPostProcess per image.
// Byte / index work with the reference's f32 statements (-ffp-contract=off): bit-exact against the oracle restatement.
#include "layout.h"

namespace oar {
namespace pp {

namespace {
inline unsigned grid_for(long work, int block = 256, long cap = 256L * 32) {
    long g = (work + block - 1) / block;
    return (unsigned)(g < 1 ? 1 : g > cap ? cap : g);
}

// vertical pass: one thread = one (output row, source column); three channels
__global__ __launch_bounds__(256) void resize_v_kernel(const uint8_t* __restrict__ src, int w, int nh, const FilterTaps* __restrict__ taps, const float* __restrict__ wts,
                                                       int max_taps, float* __restrict__ tmp) {
    const long total = (long)nh * w;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int oy = (int)(i / w), x = (int)(i - (long)oy * w);
        const FilterTaps t = taps[oy];
        const float* wv = wts + (long)oy * max_taps;
        float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
        for (int k = 0; k < t.n; ++k) {
            const uint8_t* p = src + ((long)(t.left + k) * w + x) * 3;
            const float wk = wv[k];
            t0 += (float)p[0] * wk; t1 += (float)p[1] * wk; t2 += (float)p[2] * wk;
        }
        float* o = tmp + i * 3;
        o[0] = t0; o[1] = t1; o[2] = t2;
    }
}
// horizontal pass: one thread = one output pixel
__global__ __launch_bounds__(256) void resize_h_kernel(const float* __restrict__ tmp, int w, int nw, int nh, const FilterTaps* __restrict__ taps, const float* __restrict__ wts,
                                                       int max_taps, uint8_t* __restrict__ dst) {
    const long total = (long)nh * nw;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int y = (int)(i / nw), ox = (int)(i - (long)y * nw);
        const FilterTaps t = taps[ox];
        const float* wv = wts + (long)ox * max_taps;
        float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;
        for (int k = 0; k < t.n; ++k) {
            const float* p = tmp + ((long)y * w + (t.left + k)) * 3;
            const float wk = wv[k];
            t0 += p[0] * wk; t1 += p[1] * wk; t2 += p[2] * wk;
        }
        auto q = [](float v) { return (uint8_t)roundf(fminf(fmaxf(v, 0.0f), 255.0f)); };
        uint8_t* o = dst + i * 3;
        o[0] = q(t0); o[1] = q(t1); o[2] = q(t2);
    }
}

// ---- LayoutPostProcess helpers (processors/layout_postprocess.rs)
__device__ __forceinline__ float rclamp(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }   // f32::clamp (NaN stays NaN)
__device__ __forceinline__ bool valid_score(float s) { return isfinite(s) && s >= 0.0f && s <= 1.0f + 1.1920929e-7f; }   // :460-462
__device__ __forceinline__ bool valid_class(float raw, int nc) {                                                        // :464-470
    if (!isfinite(raw)) return false;
    const int c = (int)roundf(raw);
    return c >= 0 && c < nc + 5;
}
__device__ __forceinline__ void convert(float x1, float y1, float x2, float y2, float ow, float oh, float* o) {        // :423-454
    const bool normalized = x2 <= 1.05f && y2 <= 1.05f && x1 >= -0.05f && y1 >= -0.05f && ow > 0.0f && oh > 0.0f;
    if (normalized) { o[0] = rclamp(x1, 0.0f, 1.0f) * ow; o[1] = rclamp(y1, 0.0f, 1.0f) * oh; o[2] = rclamp(x2, 0.0f, 1.0f) * ow; o[3] = rclamp(y2, 0.0f, 1.0f) * oh; }
    else { o[0] = rclamp(x1, 0.0f, ow); o[1] = rclamp(y1, 0.0f, oh); o[2] = rclamp(x2, 0.0f, ow); o[3] = rclamp(y2, 0.0f, oh); }
}
__device__ __forceinline__ bool valid_box(const float* b) { return b[2] > b[0] && b[3] > b[1] && isfinite(b[0]) && isfinite(b[1]) && isfinite(b[2]) && isfinite(b[3]); }

// one workgroup per image
__global__ __launch_bounds__(256) void layout_post_kernel(LayoutPostP p) {
    extern __shared__ unsigned char lp_lds[];
    unsigned char* sup = lp_lds;                                        // [rows] suppressed flags (by candidate row)
    __shared__ int s_nvalid, s_nkeep, s_cur, s_go;
    const int img = blockIdx.x, tid = threadIdx.x;
    const float* pred = p.pred + (long)img * p.rows * p.feat;
    float* cand = p.cand + (long)img * p.rows * 8;
    int* sorted = p.sorted + (long)img * p.rows;
    int* keep = p.keep + (long)img * p.max_det;
    const float ow = p.src_wh[img * 2], oh = p.src_wh[img * 2 + 1];
    if (tid == 0) { s_nvalid = 0; s_nkeep = 0; }
    // ---- phase A: parse every row (process_picodet :99-211 / process_pp_doclayout :232-333)
    for (int r = tid; r < p.rows; r += 256) {
        const float* row = pred + (long)r * p.feat;
        float b[4] = {0.f, 0.f, 0.f, 0.f}, score = 0.f;
        int cls = 0;
        bool ok = false;
        if (p.model_type == 2) {
            if (p.feat >= 6) {
                const float cf = row[0];
                const int ci = cf != cf ? 0 : cf >= 2147483648.0f ? 2147483647 : cf <= -2147483648.0f ? (-2147483647 - 1) : (int)cf;   // `as i32`
                score = row[1];
                if (!(score < p.score_thr || ci < 0 || ci >= p.num_classes)) {
                    convert(row[2], row[3], row[4], row[5], ow, oh, b);
                    ok = valid_box(b);
                    cls = ci;
                }
            }
        } else if (p.feat == 4 + p.num_classes) {
            int best = 0; float bs = -INFINITY;
            for (int c = 0; c < p.num_classes; ++c) if (row[4 + c] > bs) { bs = row[4 + c]; best = c; }
            if (!(bs < p.score_thr)) { convert(row[0], row[1], row[2], row[3], ow, oh, b); ok = valid_box(b); cls = best; score = bs; }
        } else if (p.feat >= 6) {
            // parse_compact_prediction (:372-421): (class, score, box) / (box, score, class) / (score, class, box)
            const int order[3][6] = {{0, 1, 2, 3, 4, 5}, {5, 4, 0, 1, 2, 3}, {1, 0, 2, 3, 4, 5}};
            bool parsed = false;
            float x[4] = {0.f, 0.f, 0.f, 0.f};
            for (int f = 0; f < 3 && !parsed; ++f) {
                const float s = row[order[f][1]], c = row[order[f][0]];
                const bool sv = p.model_type == 1 ? isfinite(s) : valid_score(s);
                if (sv && valid_class(c, p.num_classes)) {
                    const int ci = (int)roundf(c);
                    if (ci >= 0) {
                        cls = ci; score = p.model_type == 1 ? rclamp(s, 0.0f, 1.0f) : s;
                        x[0] = row[order[f][2]]; x[1] = row[order[f][3]]; x[2] = row[order[f][4]]; x[3] = row[order[f][5]];
                        parsed = true;
                    }
                }
            }
            if (parsed && !(score < p.score_thr || cls >= p.num_classes)) { convert(x[0], x[1], x[2], x[3], ow, oh, b); ok = valid_box(b); }
        }
        float* c8 = cand + (long)r * 8;
        c8[0] = b[0]; c8[1] = b[1]; c8[2] = b[2]; c8[3] = b[3]; c8[4] = score; c8[5] = __int_as_float(cls); c8[6] = ok ? 1.0f : 0.0f; c8[7] = 0.0f;
        sup[r] = 0;
    }
    __syncthreads();
    // ---- phase B: stable descending rank by score among the valid rows (sort_by(partial_cmp) of compute_nms_keep_indices, :488-494):
    // a precedes b iff score_b < score_a, ties keep row order
    for (int i = tid; i < p.rows; i += 256) {
        if (cand[(long)i * 8 + 6] == 0.0f) continue;
        const float si = cand[(long)i * 8 + 4];
        int rank = 0;
        for (int j = 0; j < p.rows; ++j) {
            if (cand[(long)j * 8 + 6] == 0.0f) continue;
            const float sj = cand[(long)j * 8 + 4];
            // j before i: (s_i < s_j) or (not (s_j < s_i) and j < i)   [incomparable (NaN) counts as equal]
            if (si < sj || (!(sj < si) && j < i)) ++rank;
        }
        sorted[rank] = i;
        atomicAdd(&s_nvalid, 1);
    }
    __syncthreads();
    const int nv = s_nvalid;
    // ---- phase C: greedy class-aware suppression (:496-545)
    for (int pos = 0; pos < nv; ++pos) {
        if (tid == 0) {
            const int i = sorted[pos];
            s_go = 0; s_cur = i;
            if (!sup[i]) {
                keep[s_nkeep++] = i;
                s_go = s_nkeep >= p.max_det ? 2 : 1;
            }
        }
        __syncthreads();
        const int go = s_go;
        if (go == 2) break;
        if (go == 1) {
            const int i = s_cur;
            const float* bi = cand + (long)i * 8;
            const float ix1 = bi[0], iy1 = bi[1], ix2 = bi[2], iy2 = bi[3];
            const int ic = __float_as_int(bi[5]);
            const float area_i = (ix2 - ix1) * (iy2 - iy1);
            for (int q = pos + 1 + tid; q < nv; q += 256) {
                const int j = sorted[q];
                const float* bj = cand + (long)j * 8;
                if (sup[j] || __float_as_int(bj[5]) != ic) continue;
                const float ax = fmaxf(ix1, bj[0]), ay = fmaxf(iy1, bj[1]), bx = fminf(ix2, bj[2]), by = fminf(iy2, bj[3]);
                if (ax >= bx || ay >= by) continue;
                const float inter = (bx - ax) * (by - ay);
                const float area_j = (bj[2] - bj[0]) * (bj[3] - bj[1]);
                const float uni = area_i + area_j - inter;
                if (uni > 0.0f && inter / uni > p.nms_thr) sup[j] = 1;
            }
        }
        __syncthreads();
    }
    __syncthreads();
    if (tid == 0) p.n_keep[img] = s_nkeep;
}
}  // namespace

void resize_filter(hipStream_t s, const uint8_t* src, int w, int h, uint8_t* dst, int nw, int nh, const FilterTaps* tv, const float* wv, int max_tv,
                   const FilterTaps* th, const float* wh, int max_th, float* tmp) {
    if ((long)nw * nh == 0) return;
    (void)h;
    ProfScope ps(s, "resize_filter", 3.0 * ((double)w * h + (double)nw * nh) + 24.0 * (double)w * nh, 0.0);
    hipLaunchKernelGGL(resize_v_kernel, dim3(grid_for((long)nh * w)), dim3(256), 0, s, src, w, nh, tv, wv, max_tv, tmp);
    hipLaunchKernelGGL(resize_h_kernel, dim3(grid_for((long)nh * nw)), dim3(256), 0, s, tmp, w, nw, nh, th, wh, max_th, dst);
}

void layout_postprocess(hipStream_t s, const LayoutPostP& p, int n_images) {
    if (n_images == 0) return;
    ProfScope ps(s, "layout_post", 4.0 * (double)n_images * p.rows * p.feat, 0.0);
    hipLaunchKernelGGL(layout_post_kernel, dim3(n_images), dim3(256), (size_t)((p.rows + 15) & ~15), s, p);
}

}  // namespace pp
}  // namespace oar
