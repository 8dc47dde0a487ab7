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

// ------------------------------------------------------------------------------------------ PP-DocLayout adapter post-processing
// LayoutDetectionAdapter::postprocess_pp_doclayout (domain/adapters/layout_detection_adapter.rs:631-846) up to and including the reading-order
// sort, one workgroup per image, every list operation as a parallel predicate over the candidate rows instead of the reference's Vec rebuilds:
//   A  parse + per-class threshold + coordinate conversion + validity                       (:683-732)      one row per thread
//   B  paddlex_layout_nms (:884-935): stable descending rank by counting, then the in-place marking form of the reference (a lane per later
//      candidate tests the selected box; same-class IoU >= 0.6 or cross-class >= 0.98 suppresses; IoU is paddlex_iou's "+ 1" form, :937-953)
//   C  filter_large_image_boxes (:955-995): page-sized "image" boxes dropped unless nothing else is left
//   D  apply_paddlex_merge_modes (:997-1100): per configured class, containment (>= 0.9 of the inner box's area) against the list as it was
//      before the step; Large drops the contained, Small drops a container that is not itself contained
//   E  reading order (:785-811): stable rank by f32::total_cmp on the order column(s), by counting
// List state = one byte per candidate row in LDS (bit 0 candidate, bit 1 alive); positions are recomputed by counting when an order is needed.
__device__ __forceinline__ float ppd_iou(const float* a, const float* b) {
    const float iw = fmaxf(fminf(a[2], b[2]) - fmaxf(a[0], b[0]) + 1.0f, 0.0f), ih = fmaxf(fminf(a[3], b[3]) - fmaxf(a[1], b[1]) + 1.0f, 0.0f);
    const float inter = iw * ih;
    const float uni = (a[2] - a[0] + 1.0f) * (a[3] - a[1] + 1.0f) + (b[2] - b[0] + 1.0f) * (b[3] - b[1] + 1.0f) - inter;
    return uni > 0.0f ? inter / uni : 0.0f;
}
__device__ __forceinline__ bool ppd_contained(const float* in, const float* out) {
    const float area = (in[2] - in[0]) * (in[3] - in[1]);
    if (area <= 0.0f) return false;
    const float iw = fmaxf(fminf(in[2], out[2]) - fmaxf(in[0], out[0]), 0.0f), ih = fmaxf(fminf(in[3], out[3]) - fmaxf(in[1], out[1]), 0.0f);
    return (iw * ih) / area >= 0.9f;
}
__device__ __forceinline__ int ppd_key(float v) { const int b = __float_as_int(v); return b ^ (int)(((unsigned)(b >> 31)) >> 1); }   // f32::total_cmp

__global__ __launch_bounds__(256) void ppdoc_post_kernel(PpDocPostP p) {
    extern __shared__ unsigned char pd_lds[];
    unsigned char* st = pd_lds;                       // [rows]: bit 0 candidate (survived A), bit 1 alive, bit 2 scratch
    __shared__ int s_n, s_go, s_cur, s_any;
    const int img = blockIdx.x, tid = threadIdx.x;
    const float* pred = p.pred + (long)img * p.rows * p.feat;
    float* cand = p.cand + (long)img * p.rows * 8;
    int* sorted = p.sorted + (long)img * p.rows;
    int* keep = p.keep + (long)img * p.rows;
    const float ow = p.src_wh[img * 2], oh = p.src_wh[img * 2 + 1];
    if (tid == 0) { s_n = 0; s_any = 0; }
    // ---- A
    for (int r = tid; r < p.rows; r += 256) {
        const float* row = pred + (long)r * p.feat;
        float b[4] = {0.f, 0.f, 0.f, 0.f};
        const float cf = row[0], score = row[1];
        const int ci = cf != cf ? 0 : cf >= 2147483648.0f ? 2147483647 : cf <= -2147483648.0f ? (-2147483647 - 1) : (int)cf;   // `as i32`
        bool ok = false;
        if (ci >= 0 && ci < p.num_classes) {
            float thr = fmaxf(p.score_thr, 0.0f);
            if (p.class_thr) { const float t = p.class_thr[ci]; if (t == t) thr = t; }
            if (!(score < thr)) { convert(row[2], row[3], row[4], row[5], ow, oh, b); ok = valid_box(b); }
        }
        float* c8 = cand + (long)r * 8;
        c8[0] = b[0]; c8[1] = b[1]; c8[2] = b[2]; c8[3] = b[3]; c8[4] = score; c8[5] = __int_as_float(ci);
        c8[6] = p.feat >= 7 ? row[6] : 0.0f; c8[7] = p.feat >= 8 ? row[7] : 0.0f;
        st[r] = ok ? 3 : 0;
    }
    __syncthreads();
    // ---- B
    if (p.layout_nms) {
        for (int i = tid; i < p.rows; i += 256) {
            if (!(st[i] & 1)) continue;
            const float si = cand[(long)i * 8 + 4];
            int rank = 0;
            for (int j = 0; j < p.rows; ++j) {
                if (!(st[j] & 1)) continue;
                const float sj = cand[(long)j * 8 + 4];
                if (si < sj || (!(sj < si) && j < i)) ++rank;      // j sorts before i (unordered pairs keep row order)
            }
            sorted[rank] = i;
            atomicAdd(&s_n, 1);
        }
        __syncthreads();
        const int nv = s_n;
        for (int pos = 0; pos < nv; ++pos) {
            if (tid == 0) { s_cur = sorted[pos]; s_go = (st[s_cur] & 2) ? 1 : 0; }
            __syncthreads();
            if (s_go) {
                const int i = s_cur;
                const float* bi = cand + (long)i * 8;
                const int ic = __float_as_int(bi[5]);
                for (int q = pos + 1 + tid; q < nv; q += 256) {
                    const int j = sorted[q];
                    if (!(st[j] & 2)) continue;
                    const float* bj = cand + (long)j * 8;
                    const float iou = ppd_iou(bi, bj), thr = __float_as_int(bj[5]) == ic ? 0.6f : 0.98f;
                    if (iou >= thr || iou != iou) st[j] &= ~2;
                }
            }
            __syncthreads();
        }
    } else {
        // selection order = row order
        for (int i = tid; i < p.rows; i += 256) {
            if (!(st[i] & 1)) continue;
            int rank = 0;
            for (int j = 0; j < i; ++j) rank += st[j] & 1;
            sorted[rank] = i;
            atomicAdd(&s_n, 1);
        }
        __syncthreads();
    }
    const int nv = s_n;
    __syncthreads();
    // ---- C (list = alive entries of sorted[0 .. nv))
    if (p.image_class >= 0) {
        if (tid == 0) { s_go = 0; s_cur = 0; }
        __syncthreads();
        const float thr = ow > oh ? 0.82f : 0.93f, img_area = ow * oh;
        int alive = 0, kept = 0;
        for (int q = tid; q < nv; q += 256) {
            const int i = sorted[q];
            if (!(st[i] & 2)) continue;
            ++alive;
            const float* b = cand + (long)i * 8;
            bool k = true;
            if (__float_as_int(b[5]) == p.image_class) {
                const float xmin = fmaxf(b[0], 0.0f), ymin = fmaxf(b[1], 0.0f), xmax = fminf(b[2], ow), ymax = fminf(b[3], oh);
                k = (xmax - xmin) * (ymax - ymin) <= thr * img_area;
            }
            if (k) { ++kept; st[i] |= 4; } else st[i] &= ~4;
        }
        atomicAdd(&s_go, alive); atomicAdd(&s_cur, kept);
        __syncthreads();
        if (s_go > 1 && s_cur > 0) {
            for (int q = tid; q < nv; q += 256) { const int i = sorted[q]; if ((st[i] & 2) && !(st[i] & 4)) st[i] &= ~2; }
        }
        __syncthreads();
    }
    // ---- D
    if (p.merge_mode) {
        for (int q = tid; q < nv; q += 256) { const int i = sorted[q]; st[i] = (st[i] & ~4) | ((st[i] & 2) ? 4 : 0); }   // bit 2 = the list before this step
        __syncthreads();
        for (int q = tid; q < nv; q += 256) {
            const int x = sorted[q];
            if (!(st[x] & 4)) continue;
            const float* bx = cand + (long)x * 8;
            const int cx = __float_as_int(bx[5]);
            bool drop = false;
            for (int c = 0; c < p.num_classes && !drop; ++c) {
                const int mode = p.merge_mode[c];
                if (mode != 0 && mode != 2) continue;                      // Union / not configured
                bool contained = false, contains = false;
                for (int r = 0; r < nv; ++r) {
                    const int y = sorted[r];
                    if (y == x || !(st[y] & 4)) continue;
                    const float* by = cand + (long)y * 8;
                    const int cy = __float_as_int(by[5]);
                    // pair (i = x, j = y): is x contained by y?   pair (i = y, j = x): does x contain y?
                    const bool skip_xy = p.formula_class >= 0 && cx == p.formula_class && cy != p.formula_class;
                    const bool skip_yx = p.formula_class >= 0 && cy == p.formula_class && cx != p.formula_class;
                    if (mode == 0) {
                        if (!skip_xy && cy == c && ppd_contained(bx, by)) contained = true;
                        if (!skip_yx && cx == c && ppd_contained(by, bx)) contains = true;
                    } else {
                        if (!skip_xy && cx == c && ppd_contained(bx, by)) contained = true;
                        if (!skip_yx && cy == c && ppd_contained(by, bx)) contains = true;
                    }
                }
                if (mode == 0) drop = contained;
                else drop = !(!contains || contained);
            }
            if (drop) st[x] &= ~2;
        }
        __syncthreads();
    }
    // ---- E: final order.  Position among the alive entries of the selection order, or the stable rank by the order column(s)
    if (tid == 0) s_go = 0;
    __syncthreads();
    for (int q = tid; q < nv; q += 256) {
        const int i = sorted[q];
        if (!(st[i] & 2)) continue;
        int rank = 0;
        if (p.feat == 7 || p.feat == 8) {
            const int ki = ppd_key(cand[(long)i * 8 + 6]), ri = ppd_key(cand[(long)i * 8 + 7]);
            for (int r = 0; r < nv; ++r) {
                const int j = sorted[r];
                if (!(st[j] & 2) || j == i) continue;
                const int kj = ppd_key(cand[(long)j * 8 + 6]), rj = ppd_key(cand[(long)j * 8 + 7]);
                const bool less = kj < ki || (kj == ki && p.feat == 8 && rj < ri);
                const bool equal = kj == ki && (p.feat != 8 || rj == ri);
                if (less || (equal && r < q)) ++rank;
            }
        } else {
            for (int r = 0; r < q; ++r) rank += (st[sorted[r]] & 2) ? 1 : 0;
        }
        keep[rank] = i;
        atomicAdd(&s_go, 1);
    }
    __syncthreads();
    if (tid == 0) p.n_keep[img] = s_go;
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

void ppdoc_postprocess(hipStream_t s, const PpDocPostP& p, int n_images) {
    if (n_images == 0) return;
    ProfScope ps(s, "layout_post", 4.0 * (double)n_images * p.rows * p.feat, 0.0);
    hipLaunchKernelGGL(ppdoc_post_kernel, dim3(n_images), dim3(256), (size_t)((p.rows + 15) & ~15), s, p);
}

}  // namespace pp
}  // namespace oar
