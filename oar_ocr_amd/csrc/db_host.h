// db_host.h -- host-side (serial, per-contour) geometry of DB post-processing and crop planning.
// These steps are pointer-chasing / tiny-N float work with libm calls (atan2f/cosf/sinf/hypotf) whose exact
// results are part of the box contract, so they stay on the host (one worker per page) -- see DESIGN.md.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace oar {
namespace host {

struct Pt { float x, y; };

struct Contour {
    std::vector<Pt> pts;   // border pixels in tracing order
    bool simplified = false;   // pts already is simplify_chain(border pixels) (find_contours_band_bits with corners_only)
    bool hole = false;
    int parent = -1;
};

// imageproc 0.27 `find_contours::<u32>` call at processors/db_bitmap.rs:100 (Suzuki-Abe, raster discovery order,
// hole borders included).  Stops after max_contours contours (the reference `take(max_candidates)` only ever
// consumes that many).
std::vector<Contour> find_contours(const uint8_t* mask, int width, int height, size_t max_contours);
// Same result, restricted to rows [y0, y1) of the image (contour coordinates stay global).  Connected components
// cannot cross a fully-blank row, so an image cut at blank rows can be traced band by band, in parallel, and the
// per-band results concatenated in band order are exactly find_contours' raster discovery order.
std::vector<Contour> find_contours_band(const uint8_t* mask, int width, int height, int y0, int y1, size_t max_contours, int32_t* scratch);
// The same two helpers on a bit-packed mask (what the detector reads back: 8x less PCIe traffic than the byte mask): pixel x of
// row y is bit (x & 7) of byte bits[y * row_bytes + (x >> 3)]; bits past `width` in a row's last byte must be zero.
// corners_only: a contour whose simplify_chain keeps >= 3 points is returned as those points (`simplified`), which is all the
// Quad / fast-score candidate stage reads; the others whole.
std::vector<Contour> find_contours_band_bits(const uint8_t* bits, int row_bytes, int width, int y0, int y1, size_t max_contours, bool corners_only = false);
std::vector<int> blank_row_bands_bits(const uint8_t* bits, int row_bytes, int height, int max_bands);
// Row cuts for find_contours_band: returns band boundaries (first = 0, last = height); every interior boundary is a
// row whose pixels are all zero; at most max_bands bands of roughly equal foreground-row count.
std::vector<int> blank_row_bands(const uint8_t* mask, int width, int height, int max_bands);

struct MinAreaRect { float cx, cy, w, h, angle; };
std::vector<Pt> convex_hull(const std::vector<Pt>& src);                 // processors/geometry.rs:226-271
MinAreaRect min_area_rect(const std::vector<Pt>& src);                   // processors/geometry.rs:310-441
std::vector<Pt> simplify_chain(const std::vector<Pt>& p);                // processors/db_bitmap.rs:207-239
// processors/db_bitmap.rs:164-205,253-277: ordered mini box + min side; false when rejected.
bool mini_box(const std::vector<Pt>& pts, Pt out[4], float& min_side);
// db_bitmap.rs:153-162: simplify the border chain, mini box of what is left (of the whole chain when fewer than 3 points are)
bool contour_mini_box(const Contour& c, Pt out[4], float& min_side);
// processors/db_bitmap.rs:279-368 (clipper2 inflate, Round join, precision 2)
std::vector<Pt> unclip(const Pt box[4], float ratio);
// processors/sorting.rs:35-84: returns the permutation
std::vector<int> sort_quad_boxes(const std::vector<float>& boxes8);

// ---- polygon (seal text) branch, poly_host.cc
float perimeter(const std::vector<Pt>& pts);                                   // processors/geometry.rs:161-171
std::vector<Pt> approx_poly_dp(const std::vector<Pt>& pts, float epsilon);     // processors/geometry.rs:453-561
// processors/db_bitmap.rs:279-368 for any polygon: Clipper2 round-join offset + the outline its closing union keeps; empty when
// the reference would drop the box (degenerate input, or an offset that is not exactly one path)
std::vector<Pt> unclip_poly(const std::vector<Pt>& poly, float ratio);
// processors/sorting.rs:100-118: stable by min y; box i = points [offsets[i], offsets[i + 1]) of pts_xy
std::vector<int> sort_poly_boxes(const std::vector<float>& pts_xy, const std::vector<uint32_t>& offsets);
int ring_outline_for_tests(const int64_t* xy, int n, int negative, int64_t* out_xy, int out_cap);
int offset_ring_for_tests(const int64_t* xy, int n, double radius, int64_t* out_xy, int out_cap);

struct CropPlan {        // utils/transform.rs:76-191
    int mode = 0;        // 0 failed, 1 axis aligned, 2 perspective
    int left = 0, top = 0, cw = 0, ch = 0, ow = 0, oh = 0, rot = 0;
    float inv[9] = {0};
    int out_w() const { return rot ? oh : ow; }
    int out_h() const { return rot ? ow : oh; }
};
CropPlan plan_crop(int img_w, int img_h, const float box8[8]);
// utils/bbox_crop.rs:26-72: the polygon's bounding rectangle as an axis-aligned crop (mode 1), mode 0 where the reference errs
CropPlan plan_bbox_crop(int img_w, int img_h, const float* pts_xy, int n_points);

// processors/resize_detection.rs:243-319 (type0). Returns true when a resize is needed.
bool det_resize_dims(uint32_t w, uint32_t h, uint32_t limit_side_len, int limit_type, uint32_t max_side_limit,
                     uint32_t& rh, uint32_t& rw);
// models/recognition/crnn.rs:80-103
int rec_tensor_width(const std::vector<uint32_t>& ws, const std::vector<uint32_t>& hs, int img_h, int img_w, int max_img_w,
                     std::vector<int32_t>& resized_w);

}  // namespace host
}  // namespace oar
