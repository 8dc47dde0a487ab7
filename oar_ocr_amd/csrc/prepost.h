// prepost.h -- launchers for the pre/post-processing HIP kernels (byte / index work: bit-exact vs the reference).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace oar {
namespace pp {

// a4 processors/simd.rs:87-123. rgb: [n_pix*3] u8 (any number of tightly packed images back to back).
// layout 0: out[c*plane + p] per image (CHW); layout 1: out[p*3 + c] (HWC / NHWC).
void normalize(hipStream_t s, const uint8_t* rgb, float* out, int64_t n_images, int64_t plane, const int src[3],
               const float alpha[3], const float beta[3], int layout);
// same arithmetic, n_pages (<= 32) separately allocated device pages of `plane` pixels each in ONE launch (host array of device pointers)
void normalize_pages(hipStream_t s, const uint8_t* const* d_pages, int n_pages, float* out, int64_t plane, const int src[3], const float alpha[3],
                     const float beta[3], int layout);

struct CropDesc {          // one recognizer input crop (device-resident u8 HWC)
    const uint8_t* src;
    int32_t w, h;          // source crop size
    int32_t rw;            // resized width (<= tensor width)
    int32_t flip;          // 1: the crop is read as its imageops::rotate180 (text-line orientation class 1, src/oarocr/ocr.rs:785-788)
};
// a16 models/recognition/crnn.rs:98-121 + simd.rs:248-308: Triangle resize to (rw x img_h), BGR (v/255-0.5)/0.5,
// zero padding to Wt.  out layout: NHWC [n][img_h][Wt][3] (channel c = source channel 2-c) or NCHW when nchw != 0.
void rec_pack(hipStream_t s, const CropDesc* d_descs, int n, int img_h, int Wt, float* out, int nchw);
// The resize half of rec_pack alone: crop i -> img_h rows of d_descs[i].rw RGB pixels at d_dst[i] (u8, tightly packed).  The fused
// recognizer stem (k::conv_smallcin_u8 with StemU8::dev) normalises and pads while it reads them: the f32 input tensor -- four
// times the bytes, written once and read once -- never exists.
struct ResizedImg { uint8_t* ptr; int32_t w; int32_t pad; };   // same layout as k::StemImg (kernels.h): one table serves both kernels
void rec_resize_u8(hipStream_t s, const CropDesc* d_descs, const ResizedImg* d_dst, int n, int img_h, int max_rw);

// image 0.25.6 imageops::resize(Triangle) on u8 RGB (processors/resize_detection.rs:314).
void resize_triangle(hipStream_t s, const uint8_t* src, int w, int h, uint8_t* dst, int nw, int nh);

// a7 processors/db_postprocess.rs:185-221
void threshold(hipStream_t s, const float* pred, uint8_t* mask, int64_t n, float thresh);

// n_images masks of height x width bytes -> bit planes of ceil(width / 8) bytes per row (pixel x = bit x & 7 of byte x >> 3): the
// form in which a mask is read back by the host border follower (8x less PCIe traffic; db_host.h find_contours_band_bits)
void pack_mask_bits(hipStream_t s, const uint8_t* mask, uint8_t* bits, int n_images, int height, int width);
// a7 for the default detector route in ONE launch: net_out = [n][C][H][W] (image stride img_stride floats), channel 0 -> probs [n][H][W] and its
// threshold as a bit plane (same bits as threshold + pack_mask_bits)
void db_keep_and_pack(hipStream_t s, const float* net_out, int64_t img_stride, float* probs, uint8_t* bits, int n_images, int height, int width, float thresh);

// DBPostProcess::dilate_mask_img (processors/db_mask.rs:11: imageproc morphology::dilate, Norm::LInf, k = 1) on n_images
// masks of height x width each: a pixel becomes 255 when any pixel of its 3 x 3 neighbourhood inside the image is non-zero.
void dilate3x3(hipStream_t s, const uint8_t* mask, uint8_t* out, int n_images, int height, int width);

// a8 on the GPU (contours.hip): imageproc find_contours (processors/db_bitmap.rs:100), one wavefront per segment (a rectangle of the
// mask bounded by blank rows / columns; see contours.hip for why segments are independent and how discovery order is recovered).
// masks: n_pages device masks of H x W.  Device work buffers: rows [n_pages * H] bytes, band_y [n_pages * trace_max_bands(H) * 2],
// n_bands [n_pages], lists [2 * list_cap] words, scratch [n_pages * H * W * 2] words, ctrl [kTraceCtlWords] words, table_dev
// [table_cap].  Host-visible (pinned) outputs: ctrl_host [kTraceCtlWords], table [table_cap], packed [packed_cap_words].
// After the stream has drained: ctrl_host[kTraceCtlSegments] records are valid in `table` (when kTraceCtlOverflow is set, or the
// count exceeds table_cap, the table was too small: follow those pages on the host).  Record r: flags == 0 -> its word stream
// ([hole << 31 | n_points][x | y << 16]...) is packed[off .. off + used); flags != 0 -> follow rows [y0, y1) x columns [x0, x1)
// of page `page` on the host.  The borders of a page in find_contours order = all its records' borders sorted by their first
// point (y, then x).
struct SegRec { int32_t page, y0, y1, x0, x1; uint32_t off, used, n_contours, flags; };
enum { kTraceCtlTotal = 0, kTraceCtlNSmall = 1, kTraceCtlNLarge = 2, kTraceCtlCurSmall = 3, kTraceCtlCurLarge = 4, kTraceCtlSegments = 5, kTraceCtlOverflow = 6,
       kTraceCtlWords = 8 };
inline int trace_max_bands(int H) { return H / 2 + 1; }
void trace_contours(hipStream_t s, const uint8_t* masks, int n_pages, int H, int W, uint8_t* rows, int32_t* band_y, int32_t* n_bands, uint32_t* lists,
                    uint32_t list_cap, uint32_t* scratch, uint32_t* packed, uint32_t packed_cap_words, uint32_t* ctrl, uint32_t* ctrl_host, SegRec* table_dev,
                    SegRec* table, uint32_t table_cap);

// a18 processors/decode.rs:452-501: last index of the row maximum + the maximum
void ctc_argmax(hipStream_t s, const float* probs, int64_t rows, int vocab, int64_t* idx, float* prob);

struct ScoreBox {          // a10 processors/db_score.rs:34-134
    float pts[8];
    int32_t image;         // which prob map of the batch
    int32_t pad;
};
void box_scores(hipStream_t s, const float* pred, int height, int width, const ScoreBox* d_boxes, int n_boxes, float* d_scores);

// a11 on the GPU: DBPostProcess::unclip (processors/db_bitmap.rs:279-368) of every mini box of a sub-batch, in the same round trip
// as its score.  Same f64 operation sequence as host::unclip (db_host.cc), one lane per box; the vertices land on the 1/100 px
// integer grid, which absorbs the last-ulp differences between the device's and glibc's acos / sin / cos / atan2 / hypot
// (tests/test_gpu_kernels.py: identical to the host routine on 20 000 random boxes).  n_pts: >= 3 polygon, 0 = dropped by the
// reference's degeneracy tests, -1 = not handled here (a reflex corner or more than kUnclipMaxPts vertices): the host routine runs.
constexpr int kUnclipMaxPts = 80;
struct UnclipOut { int32_t n_pts; int32_t pad; float pts[kUnclipMaxPts * 2]; };
void unclip_quads(hipStream_t s, const ScoreBox* d_boxes, int n_boxes, float ratio, UnclipOut* d_out);

// Scanline mean over an arbitrary polygon: box_score_slow (processors/db_score.rs:139-181, the contour itself is the
// polygon) and the polygon (seal) path's box_score_fast on the approximated contour (db_bitmap.rs:49).  Same arithmetic and
// summation order as box_scores; polygon i = pts[poly[i].first .. first + count) (x, y pairs).
struct PolyDesc { int32_t first, count, image, pad; };
void poly_scores(hipStream_t s, const float* pred, int height, int width, const float* d_pts_xy, const PolyDesc* d_polys, int n_polys, float* d_scores);

struct WarpDesc {          // a14 utils/transform.rs:76-191 (plan computed on the host)
    const uint8_t* page;   // device u8 HWC
    int32_t page_w, page_h;
    int32_t left, top, cw, ch;   // AABB crop inside the page
    int32_t ow, oh;              // warp output size BEFORE the optional rotate270
    int32_t rot;                 // 1: output is rotate270 of the warp (out dims oh x ow)
    int32_t mode;                // 1 axis-aligned copy, 2 perspective bicubic
    float inv[9];                // inverse homography (dst -> src), row-major
    int64_t out_off;             // byte offset of this crop in the output pool
};
void rotate_crops(hipStream_t s, const WarpDesc* d_descs, int n, uint8_t* out_pool, int max_out_pixels);

// ---- config 5
struct ClsDesc {           // one classifier input image (device-resident u8 HWC)
    const uint8_t* src;
    int32_t w, h;          // source size
    int32_t nw, nh;        // Triangle-resize target (pp_lcnet.rs:158-163 short-edge rule, or the input size itself)
    int32_t x1, y1;        // centre-crop origin inside the resized image (pp_lcnet.rs:166-169)
    int32_t pad;
};
// a22 models/classification/pp_lcnet.rs:139-196: resize (Triangle) + centre crop + `v*alpha[c] + beta[c]` in RGB order.
// out: NHWC [n][crop_h][crop_w][3] (nchw == 0) or NCHW.
void cls_pack(hipStream_t s, const ClsDesc* d_descs, int n, int crop_h, int crop_w, const float alpha[3], const float beta[3], float* out, int nchw);
// image 0.25 imageops::rotate90 / 180 / 270 (clockwise quarter turns) on u8 RGB; dst is (h x w) for quarter 1, 3.
void rotate_rgb(hipStream_t s, const uint8_t* src, int w, int h, int quarter, uint8_t* dst);
// a23 processors/simd.rs:327-348: planes (B, G, R) f32 -> interleaved RGB8 with `(v*scale).clamp(0,255) as u8`.
void bgr_planes_to_rgb(hipStream_t s, const float* planes, int64_t plane, float scale, uint8_t* out);

}  // namespace pp
}  // namespace oar
