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
    // Workers poll for work only while the pool is ACTIVE (a detector call is in flight); otherwise they park on the condition
    // variable at once.  A polling worker burns a whole CPU: under a container CPU quota (cgroup cpu.max) sixteen of them
    // polling through the recognition phase as well were enough to get the process throttled for tens of milliseconds.
    void set_active(bool on);
    struct ActiveScope { ThreadPool& p; explicit ActiveScope(ThreadPool& pool) : p(pool) { p.set_active(true); } ~ActiveScope() { p.set_active(false); } };
    // CPUs this process may really use: min(affinity mask, cgroup CPU quota), at least 1
    static int available_cpus();
    // test hooks (oar_host_pool_selftest only; set before the first parallel_for, never in production): called by a worker
    // between observing a new generation and reading the job descriptor / by the publisher between writing the descriptor
    // and opening the claim word -- the two windows of the stale-descriptor race.
    std::function<void()> selftest_worker_delay_, selftest_publish_delay_;

   private:
    void loop();
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::atomic<const std::function<void(int)>*> fn_{nullptr};   // published before gen_ (release), read after it (acquire)
    std::atomic<int> gen_{0}, done_{0}, count_{0};
    // (generation << 32) | next index.  An index is claimed by CAS, so a worker that is still holding an OLD job's
    // descriptor can never take an index of the next job (with a bare counter it could, and then ran the old, already
    // destroyed closure on it).
    std::atomic<uint64_t> state_{0};
    bool claim(int gen, int count, int& index);
    std::atomic<bool> stop_{false};
    std::atomic<int> active_{0};
    std::mutex err_mu_;
    std::exception_ptr err_;
};

struct PageRef {              // one input page (RGB8 HWC)
    const uint8_t* host = nullptr;
    const uint8_t* dev = nullptr;
    uint32_t w = 0, h = 0;
};

struct DetBoxes {             // per image, discovery order
    std::vector<float> pts;   // BoxType::Quad: n * 8; BoxType::Poly: 2 * sum(counts)
    std::vector<float> scores;
    std::vector<uint32_t> counts;   // BoxType::Poly: vertices per box (empty for quads)
};

// TextDetectionAdapter + DBModel (domain/adapters/text_detection_adapter.rs:36-79, models/detection/db.rs:281-335)
// Buffers of pp::trace_contours for up to `pages` masks of H x W (contours.hip): device work space shared by the sub-batches of a
// detector call (they are traced one after the other on one stream), host-visible results per page / per sub-batch.
struct ContourBufs {
    DevBuf rows, band_y, n_bands, lists, scratch, ctrl, table_dev;
    PinBuf ctrl_host, table, packed;
    static constexpr uint32_t kSegsPerPage = 4096;   // segment-table entries per page (more segments: that page is followed on the host)
    size_t packed_words_per_page = 0;
    void reserve(int pages, int sub_pages, int H, int W, int n_sub);
    bool fits(int pages, int sub_pages, int H, int W, int n_sub) const;
};

class Detector {
   public:
    Detector(const uint8_t* onnx, size_t len, const oar_det_cfg& cfg);
    ~Detector();
    // Pages may be host or device resident. dev_pages_out (optional) receives the device pointer of each page
    // (uploaded copies stay valid until the next call).
    // on_ready(first, count) (optional) is called as soon as the boxes of pages [first, first + count) are final, in
    // page order and while the GPU is still busy with later pages; dev_pages_out is filled before the first call.
    using ReadyFn = std::function<void(int first, int count)>;
    void run(const std::vector<PageRef>& pages, float thresh, float box_thresh, float unclip, std::vector<DetBoxes>& out,
             std::vector<const uint8_t*>* dev_pages_out = nullptr, const ReadyFn& on_ready = nullptr);
    Engine& engine() { return *eng_; }
    static void postprocess_host(const float* pred, int H, int W, uint32_t src_w, uint32_t src_h, float thresh, float box_thresh,
                                 float unclip, uint32_t max_candidates, DetBoxes& out, int score_mode = 0, int use_dilation = 0, int box_type = 0);
    ThreadPool& pool() { return *pool_; }
    // a8 through the GPU tracer for one device-resident mask (test hook + postprocess_host)
    static std::vector<host::Contour> trace_device_mask(const uint8_t* d_mask, int H, int W, uint32_t max_contours, bool gpu_contours = true);

   private:
    void run_group(const std::vector<int>& idx, const std::vector<PageRef>& pages, uint32_t rh, uint32_t rw, float thresh,
                   float box_thresh, float unclip, std::vector<DetBoxes>& out, const ReadyFn& on_ready);
    std::unique_ptr<Engine> eng_;
    std::unique_ptr<ThreadPool> pool_;
    oar_det_cfg cfg_;
    DevBuf pages_dev_, resized_dev_, input_f32_, mask_dev_, probs_keep_;
    DevBuf mask_dil_;   // use_dilation: the dilated masks (what the host traces)
    DevBuf mask_bits_;  // the traced masks as bit planes: what is read back (mask_host_ holds bits, row_bytes = ceil(W / 8))
    // image as the resize stage sees it: the page itself, or its black-padded copy when h + w < 64
    // (DetResizeForTest::image_padding, processors/resize_detection.rs:174-176,204-220)
    DevBuf padded_dev_;
    std::vector<const uint8_t*> det_src_;
    std::vector<uint32_t> det_w_, det_h_;
    std::vector<hipEvent_t> sub_events_, mask_ready_, score_done_;
    // box-score round trip of each sub-batch (its own slots: the next sub-batch's is enqueued before this one is read)
    struct ScoreSlot {
        PinBuf boxes_host, scores_host; DevBuf boxes_dev, scores_dev; std::vector<size_t> base; size_t total = 0;
        PinBuf poly_pts_host, poly_desc_host; DevBuf poly_pts_dev, poly_desc_dev;   // ScoreMode::Slow: the contours themselves
        PinBuf unclip_host; DevBuf unclip_dev; bool unclipped = false;              // a11 on the GPU: pp::UnclipOut per candidate
    };
    std::vector<std::unique_ptr<ScoreSlot>> score_slots_;
    std::vector<float> finish_pts_;       // finish(): per-candidate results of the chunk-parallel unclip / second mini box
    std::vector<uint8_t> finish_ok_;
    // mask read-back runs on its own stream: a D2H copy is a blit KERNEL on the queue it is issued to (150 us per 9-page
    // sub-batch over PCIe), which on the engine stream held up the next sub-batch's network
    hipStream_t copy_stream_ = nullptr;
    hipStream_t upload_stream2_ = nullptr; // second uploader thread's stream (alternate pages of a sub-batch)
    hipStream_t upload_stream_ = nullptr;  // host pages -> HBM, one sub-batch ahead of the network
    hipEvent_t stage_free_ = nullptr;
    std::vector<hipEvent_t> upload_done_, upload_done2_;
    std::vector<const uint8_t*> upload_src_;   // per page: host source still to be uploaded (nullptr = resident)
    hipStream_t score_stream_ = nullptr;   // box-score round trips (not behind the queued mask copies of later sub-batches)
    PinBuf mask_host_;
    ContourBufs trace_;     // a8 on the GPU (OAR_GPU_CONTOURS=0: trace on the host from the read-back mask instead)
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
    struct Crop { const uint8_t* host = nullptr; const uint8_t* dev = nullptr; uint32_t w = 0, h = 0; bool flip = false; };   // flip: recognise rotate180 of the crop
    void run(const std::vector<Crop>& crops, RecOut& out);
    // Several recognition batches back to back on the engine stream with ONE synchronisation at the end
    // (the reference runs them serially under the session lock, src/oarocr/ocr.rs:827-841).
    void run_batches(const std::vector<std::vector<Crop>>& batches, std::vector<RecOut>& outs);
    // test hook: packed input tensor only
    void pack_only(const std::vector<Crop>& crops, std::vector<float>& nchw, uint32_t& Wt);
    Engine& engine() { return *eng_; }

   private:
    const float* pack(const std::vector<Crop>& crops, int& Wt, bool nchw, size_t desc_slot = 0, size_t stage_slot = 0, int lane = 0);
    // the fused-stem form of pack: crops resized to u8 only (pp::rec_resize_u8); returns the device table the stem reads
    const pp::ResizedImg* pack_u8(const std::vector<Crop>& crops, int& Wt, size_t desc_slot, size_t stage_slot, int lane);
    DevBuf imgs_dev_; PinBuf imgs_host_;   // pp::ResizedImg per crop of a run (parallel to the CropDesc table)
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

// PP-LCNet image classifier adapter: DocumentOrientationAdapter / TextLineOrientationAdapter
// (models/classification/pp_lcnet.rs:139-330, utils/topk.rs:181-199).  SURVEY 8a row a22.
struct ClsCfg {
    int32_t device_id = 0;
    uint32_t input_h = 224, input_w = 224;
    uint32_t resize_short = 256;   // 0: direct resize to (input_w, input_h) (text-line orientation: 160 x 80)
    uint32_t topk = 1;
    uint32_t batch = 64;
};
struct ClsOut { std::vector<int32_t> ids; std::vector<float> scores; uint32_t topk = 0, n_classes = 0; };   // n * topk each
class Classifier {
   public:
    Classifier(const uint8_t* onnx, size_t len, const ClsCfg& cfg);
    struct Image { const uint8_t* host = nullptr; const uint8_t* dev = nullptr; uint32_t w = 0, h = 0; };
    void run(const std::vector<Image>& images, ClsOut& out);
    // test hook: the packed input tensor (NCHW) of the images
    void pack_only(const std::vector<Image>& images, std::vector<float>& nchw);
    Engine& engine() { return *eng_; }
    const ClsCfg& cfg() const { return cfg_; }

   private:
    const float* pack(const std::vector<Image>& images, size_t i0, size_t n, bool nchw);
    std::unique_ptr<Engine> eng_;
    ClsCfg cfg_;
    DevBuf stage_dev_, descs_dev_, input_f32_;
    PinBuf stage_host_, descs_host_, probs_host_;
    std::mutex mu_;
};

// UVDoc rectifier adapter (models/rectification/uvdoc.rs:82-109,166-207).  SURVEY 8a row a23.
struct RectCfg { int32_t device_id = 0; uint32_t target_h = 512, target_w = 512; };
class Rectifier {
   public:
    Rectifier(const uint8_t* onnx, size_t len, const RectCfg& cfg);
    // src: device u8 HWC (w x h); dst: device u8 HWC of the SAME size (caller-allocated).  Enqueued on stream(), not synchronised.
    void run_device(const uint8_t* src, uint32_t w, uint32_t h, uint8_t* dst);
    void run_host(const uint8_t* src, uint32_t w, uint32_t h, uint8_t* dst);
    struct Page { const uint8_t* src = nullptr; uint32_t w = 0, h = 0; uint8_t* dst = nullptr; };
    void run_device_batch(const std::vector<Page>& pages);   // same as run_device per page, the network batched over pages
    Engine& engine() { return *eng_; }

   private:
    void run_device_locked(const uint8_t* src, uint32_t w, uint32_t h, uint8_t* dst);   // mu_ held by the caller
    std::unique_ptr<Engine> eng_;
    RectCfg cfg_;
    DevBuf resized_dev_, input_f32_, out_u8_, io_dev_;
    std::mutex mu_;
};

struct OcrRegion {
    float pts[8];
    std::vector<float> poly;   // BoxType::Poly (seal text): the polygon's (x, y) pairs; pts then holds the box only when it has 4 points
    float det_score;
    uint32_t crop_w, crop_h, T;
    float max_wh_ratio;
    std::vector<int64_t> idx;
    std::vector<float> prob;
    float line_angle = -1.0f;   // text-line orientation (0 / 180), -1 when that stage is not attached (ocr.rs:782-783)
};

// OAROCR (src/oarocr/ocr.rs)
class Ocr {
   public:
    Ocr(const uint8_t* det, size_t det_len, const uint8_t* rec, size_t rec_len, const oar_ocr_cfg& cfg);
    struct PageMeta { float angle = -1.0f; bool rectified = false; uint32_t rotated_w = 0, rotated_h = 0; };
    // meta_out (optional) receives what the optional stages did to each page, copied while the call still holds the lock
    void predict(const std::vector<PageRef>& pages, std::vector<std::vector<OcrRegion>>& out, std::vector<PageMeta>* meta_out = nullptr);
    // Optional stages of OAROCR (src/oarocr/preprocess.rs:59-97, src/oarocr/ocr.rs:760-790); not owned.
    void attach(Classifier* doc_orientation, Rectifier* rectifier, Classifier* line_orientation) { doc_cls_ = doc_orientation; rect_ = rectifier; line_cls_ = line_orientation; }
    Detector& det() { return *det_; }
    Recognizer& rec() { return *rec_; }

   private:
    void predict_core(const std::vector<PageRef>& pages, std::vector<std::vector<OcrRegion>>& out);
    void preprocess_pages(const std::vector<PageRef>& pages, std::vector<PageRef>& cur);
    std::unique_ptr<Detector> det_;
    std::unique_ptr<Recognizer> rec_;
    oar_ocr_cfg cfg_;
    DevBuf crop_pool_, warp_descs_dev_;
    Classifier* doc_cls_ = nullptr;
    Rectifier* rect_ = nullptr;
    Classifier* line_cls_ = nullptr;
    std::vector<PageMeta> meta_;
    DevBuf pre_pages_;               // corrected pages (device)
    DevBuf upload_pages_;
    PinBuf warp_descs_host_;
    std::mutex mu_;
};

}  // namespace oar
