// pipeline.h -- the MI355X stand-ins for the reference's text-detection / text-recognition adapters and
// for OAROCR::predict (src/oarocr/ocr.rs:518-659).
#pragma once
#include <atomic>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "db_host.h"
#include "engine.h"
#include "prepost.h"

namespace oar {

// Small fork-join pool for the host-side geometry. Workers SPIN for a short window after each job before they
// go to sleep on a condition variable: the work arrives as ~1 ms bursts every few ms, and waking parked cores
// (deep C-states, min clocks) was measured to cost more than the work itself.
class ThreadPool {
   public:
    explicit ThreadPool(int n);
    ~ThreadPool();
    void parallel_for(int count, const std::function<void(int)>& fn);
    int size() const { return (int)workers_.size(); }

   private:
    void loop();
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_;
    const std::function<void(int)>* fn_ = nullptr;
    std::atomic<int> gen_{0}, next_{0}, done_{0}, count_{0};
    std::atomic<bool> stop_{false};
    std::mutex err_mu_;
    std::exception_ptr err_;
};

struct PageRef {              // one input page (RGB8 HWC)
    const uint8_t* host = nullptr;
    const uint8_t* dev = nullptr;
    uint32_t w = 0, h = 0;
};

struct DetBoxes {             // per image, discovery order
    std::vector<float> pts;   // n*8
    std::vector<float> scores;
};

// TextDetectionAdapter + DBModel (domain/adapters/text_detection_adapter.rs:36-79, models/detection/db.rs:281-335)
class Detector {
   public:
    Detector(const uint8_t* onnx, size_t len, const oar_det_cfg& cfg);
    // Pages may be host or device resident. dev_pages_out (optional) receives the device pointer of each page
    // (uploaded copies stay valid until the next call).
    void run(const std::vector<PageRef>& pages, float thresh, float box_thresh, float unclip, std::vector<DetBoxes>& out,
             std::vector<const uint8_t*>* dev_pages_out = nullptr);
    Engine& engine() { return *eng_; }
    static void postprocess_host(const float* pred, int H, int W, uint32_t src_w, uint32_t src_h, float thresh, float box_thresh,
                                 float unclip, uint32_t max_candidates, DetBoxes& out);
    ThreadPool& pool() { return *pool_; }

   private:
    void run_group(const std::vector<int>& idx, const std::vector<PageRef>& pages, uint32_t rh, uint32_t rw, float thresh,
                   float box_thresh, float unclip, std::vector<DetBoxes>& out);
    std::unique_ptr<Engine> eng_;
    std::unique_ptr<ThreadPool> pool_;
    oar_det_cfg cfg_;
    DevBuf pages_dev_, resized_dev_, input_f32_, mask_dev_, boxes_dev_, scores_dev_, probs_keep_;
    // image as the resize stage sees it: the page itself, or its black-padded copy when h + w < 64
    // (DetResizeForTest::image_padding, processors/resize_detection.rs:174-176,204-220)
    DevBuf padded_dev_;
    std::vector<const uint8_t*> det_src_;
    std::vector<uint32_t> det_w_, det_h_;
    std::vector<hipEvent_t> sub_events_;
    PinBuf mask_host_, boxes_host_, scores_host_;
    std::vector<const uint8_t*> page_ptrs_;
    std::mutex mu_;
};

struct RecOut {
    uint32_t T = 0, V = 0, Wt = 0;
    std::vector<int64_t> idx;  // n*T
    std::vector<float> prob;
};

// TextRecognitionAdapter + CRNNModel (domain/adapters/text_recognition_adapter.rs:35-111, models/recognition/crnn.rs:247-293)
class Recognizer {
   public:
    Recognizer(const uint8_t* onnx, size_t len, const oar_rec_cfg& cfg);
    struct Crop { const uint8_t* host = nullptr; const uint8_t* dev = nullptr; uint32_t w = 0, h = 0; };
    void run(const std::vector<Crop>& crops, RecOut& out);
    // Several recognition batches back to back on the engine stream with ONE synchronisation at the end
    // (the reference runs them serially under the session lock, src/oarocr/ocr.rs:827-841).
    void run_batches(const std::vector<std::vector<Crop>>& batches, std::vector<RecOut>& outs);
    // test hook: packed input tensor only
    void pack_only(const std::vector<Crop>& crops, std::vector<float>& nchw, uint32_t& Wt);
    Engine& engine() { return *eng_; }

   private:
    const float* pack(const std::vector<Crop>& crops, int& Wt, bool nchw, size_t desc_slot = 0, size_t stage_slot = 0, int lane = 0);
    Engine& lane_engine(int lane) { return lane == 0 ? *eng_ : *lanes_[lane - 1]; }
    std::unique_ptr<Engine> eng_;
    // Recognition batches are independent: they are dealt round-robin to `1 + lanes_.size()` engines (own stream, own
    // arena, own input tensor), so one batch's short kernels and dispatch gaps overlap with another batch's work.
    std::vector<std::unique_ptr<Engine>> lanes_;
    std::vector<std::unique_ptr<DevBuf>> lane_in_;
    std::vector<hipEvent_t> lane_done_;
    oar_rec_cfg cfg_;
    DevBuf crops_dev_, descs_dev_, input_f32_, idx_dev_, prob_dev_;
    PinBuf descs_host_, idx_host_, prob_host_, stage_host_;
    std::mutex mu_;
};

struct OcrRegion {
    float pts[8];
    float det_score;
    uint32_t crop_w, crop_h, T;
    float max_wh_ratio;
    std::vector<int64_t> idx;
    std::vector<float> prob;
};

// OAROCR (src/oarocr/ocr.rs)
class Ocr {
   public:
    Ocr(const uint8_t* det, size_t det_len, const uint8_t* rec, size_t rec_len, const oar_ocr_cfg& cfg);
    void predict(const std::vector<PageRef>& pages, std::vector<std::vector<OcrRegion>>& out);
    Detector& det() { return *det_; }
    Recognizer& rec() { return *rec_; }

   private:
    std::unique_ptr<Detector> det_;
    std::unique_ptr<Recognizer> rec_;
    oar_ocr_cfg cfg_;
    DevBuf crop_pool_, warp_descs_dev_;
    PinBuf warp_descs_host_;
    std::mutex mu_;
};

}  // namespace oar
