// layout.h -- layout detection: PicoDet / RT-DETR / PP-DocLayout graphs behind LayoutDetectionAdapter's model half.
//   ScaleAwareDetectorModel::forward (oar-ocr-core/src/models/detection/scale_aware_detector.rs:169-440): DetResizeForTest Type1 (resize_exact
//   to the model's image_shape with Lanczos3 / CatmullRom), NormalizeImage, graph("image", "scale_factor"[, "im_shape"]) -> [M, 6|7|8];
//   LayoutPostProcess::apply (processors/layout_postprocess.rs:60-634): row parsing, score filter, coordinate conversion, class-aware NMS.
// Everything between the u8 pages and the kept (box, class, score) lists stays in HBM.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>
#include <mutex>
#include <vector>

#include "common.h"
#include "engine.h"

namespace oar {

struct LayoutCfg {
    int device_id = 0;
    uint32_t input_h = 800, input_w = 608;   // image_shape (h, w): PicoDet 800 x 608, PP-DocLayout 800 x 800
    int filter = 2;                          // 0 Triangle, 1 CatmullRom, 2 Lanczos3
    bool bgr = true;                         // ColorOrder of the tensor (PicoDet BGR, PP-DocLayout RGB)
    float scale = 1.0f / 255.0f;
    float mean[3] = {0.485f, 0.456f, 0.406f}, stdv[3] = {0.229f, 0.224f, 0.225f};   // RGB statistics (permuted for BGR)
    uint32_t num_classes = 5;
    int model_type = 0;                      // 0 picodet / standard, 1 rtdetr, 2 pp-doclayout
    float score_threshold = 0.5f, nms_threshold = 0.5f;
    uint32_t max_detections = 100;
};

struct LayoutOut {                           // LayoutPostprocessOutput: per image, in the reference's order
    std::vector<uint32_t> offsets;           // n_images + 1
    std::vector<float> boxes;                // 4 per detection: x1 y1 x2 y2 (original-image pixels)
    std::vector<int32_t> classes;
    std::vector<float> scores;
    uint32_t feature_dim = 0;                // 7 / 8: the graph carried reading-order columns (is_reading_order_sorted)
};

namespace pp {
struct FilterTaps { int left, n; };          // per output coordinate; weights at [o * max_taps .. + n)
// image 0.25.6 imageops::resize, two passes with host-computed taps (layout.cc filter_taps: the Lanczos window needs sinf, and the
// weights must be the host libm's to the bit): vertical u8 -> f32, horizontal f32 -> u8 with clamp + round.
void resize_filter(hipStream_t s, const uint8_t* src, int w, int h, uint8_t* dst, int nw, int nh, const FilterTaps* tv, const float* wv, int max_tv,
                   const FilterTaps* th, const float* wh, int max_th, float* tmp);
struct LayoutPostP {
    const float* pred;       // [n_images][rows][feat]
    int rows, feat, num_classes, model_type, max_det;
    float score_thr, nms_thr;
    const float* src_wh;     // [n_images][2] original width, height
    float* cand;             // scratch [n_images][rows][8]: x1 y1 x2 y2 score class valid row
    int* sorted;             // scratch [n_images][rows]
    int* keep;               // out [n_images][max_det] candidate rows in keep order
    int* n_keep;             // out [n_images]
};
struct PpDocPostP {           // the PP-DocLayout adapter's own post-processing (layout.hip ppdoc_post_kernel)
    const float* pred;       // [n_images][rows][feat], feat 6 | 7 | 8
    int rows, feat, num_classes;
    float score_thr;
    const float* class_thr;  // [num_classes] per-class threshold, NaN = not configured; may be null
    int layout_nms, image_class, formula_class;
    const int* merge_mode;   // [num_classes]: -1 not configured, 0 Large, 1 Union, 2 Small; may be null
    const float* src_wh;     // [n_images][2]
    float* cand;             // scratch [n_images][rows][8]: x1 y1 x2 y2 score class order0 order1
    int* sorted;             // scratch [n_images][rows]
    int* keep;               // out [n_images][rows]: candidate rows in final order
    int* n_keep;             // out [n_images]
};
void ppdoc_postprocess(hipStream_t s, const PpDocPostP& p, int n_images);
// LayoutPostProcess for every image of a batch: one workgroup per image (parse -> stable rank by score -> greedy class-aware NMS)
void layout_postprocess(hipStream_t s, const LayoutPostP& p, int n_images);
}  // namespace pp

namespace host {
// apply_nms_with_merge (processors/layout_postprocess.rs:743-841); returns the number of boxes written
int nms_with_merge(const float* boxes, const int32_t* classes, const float* scores, int n, const int32_t* mode_of_class, int num_classes, float nms_thr, int max_det,
                   float* out_boxes, int32_t* out_classes, float* out_scores);
// taps of one axis: image's sample loop (same statements as the oracle, kept in C++ for the product); returns max taps
int filter_taps(int filter, int in_len, int out_len, std::vector<pp::FilterTaps>& taps, std::vector<float>& weights);
}  // namespace host

class LayoutDetector {
   public:
    LayoutDetector(const uint8_t* onnx, size_t len, const LayoutCfg& cfg);
    struct Image { const uint8_t* host = nullptr; const uint8_t* dev = nullptr; uint32_t w = 0, h = 0; };
    void run(const std::vector<Image>& images, LayoutOut& out);
    // PP-DocLayout: the adapter's post-processing instead of LayoutPostProcess (host arrays are copied to the device per call)
    struct PpDocCfg { float score_threshold = 0.5f; const float* class_thr = nullptr; bool layout_nms = true; int image_class = -1, formula_class = -1; const int32_t* merge_mode = nullptr; };
    void run_ppdoc(const std::vector<Image>& images, const PpDocCfg& pc, LayoutOut& out);
    // the preprocessed tensor of ONE image, [3, H, W] f32 on the host (parity tests)
    void preprocess_only(const Image& im, std::vector<float>& chw);
    Engine& engine() { return *eng_; }

   private:
    const float* preprocess(const std::vector<Image>& images, size_t i0, size_t n, std::vector<float>& scale_factor, std::vector<float>& src_wh);
    void run_impl(const std::vector<Image>& images, const PpDocCfg* pc, LayoutOut& out);
    std::unique_ptr<Engine> eng_;
    LayoutCfg cfg_;
    bool wants_im_shape_ = false;
    DevBuf stage_dev_, resized_dev_, tmp_f32_, input_f32_, taps_dev_, aux_dev_, cand_dev_, sorted_dev_, keep_dev_;
    PinBuf stage_host_;
    std::mutex mu_;
};

}  // namespace oar
