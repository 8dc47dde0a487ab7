// layout.cc -- host side of the layout-detection path (layout.h): resize taps, batching, graph inputs, result assembly.
#include "layout.h"

#include <algorithm>
#include <cmath>
#include <cstring>

#include "prepost.h"

namespace oar {

namespace host {

namespace {
// image 0.25.6 imageops::sample kernels (FilterType::{Triangle, CatmullRom, Lanczos3}); evaluated with the host libm because the
// Lanczos window is sinf-based and the weights have to be the reference platform's to the last bit
struct Filter {
    int kind;
    float support() const { return kind == 0 ? 1.0f : kind == 1 ? 2.0f : 3.0f; }
    static float sinc(float t) {
        const float a = t * 3.14159274f;
        return t == 0.0f ? 1.0f : std::sin(a) / a;
    }
    float operator()(float x) const {
        const float a = std::fabs(x);
        switch (kind) {
            case 0: return a < 1.0f ? 1.0f - a : 0.0f;
            case 2: return a < 3.0f ? sinc(x) * sinc(x / 3.0f) : 0.0f;
            default: {   // bc_cubic_spline(x, 0, 0.5)
                float k = 0.0f;
                if (a < 1.0f) k = 9.0f * (a * a * a) + -15.0f * (a * a) + 6.0f;
                else if (a < 2.0f) k = -3.0f * (a * a * a) + 15.0f * (a * a) + -24.0f * a + 12.0f;
                return k / 6.0f;
            }
        }
    }
};
}  // namespace

int filter_taps(int filter, int in_len, int out_len, std::vector<pp::FilterTaps>& taps, std::vector<float>& weights) {
    const Filter f{filter};
    const float ratio = (float)in_len / (float)out_len;
    const float sratio = ratio < 1.0f ? 1.0f : ratio;
    const float reach = f.support() * sratio;
    // widest window any output can have: 2 * reach + 2 source samples
    const int max_taps = std::min(in_len, (int)std::ceil(2.0f * reach) + 2);
    taps.assign((size_t)out_len, pp::FilterTaps{0, 0});
    weights.assign((size_t)out_len * max_taps, 0.0f);
    for (int o = 0; o < out_len; ++o) {
        float centre = ((float)o + 0.5f) * ratio;
        long lo = (long)std::floor(centre - reach);
        lo = std::min<long>(std::max<long>(lo, 0), (long)in_len - 1);
        long hi = (long)std::ceil(centre + reach);
        hi = std::min<long>(std::max<long>(hi, lo + 1), (long)in_len);
        centre -= 0.5f;
        float* w = weights.data() + (size_t)o * max_taps;
        OAR_CHECK(hi - lo <= max_taps, OAR_INTERNAL, "resize filter: tap window larger than planned");
        float total = 0.0f;
        for (long i = lo; i < hi; ++i) { const float v = f(((float)i - centre) / sratio); w[i - lo] = v; total += v; }
        for (long i = 0; i < hi - lo; ++i) w[i] /= total;
        taps[(size_t)o] = pp::FilterTaps{(int)lo, (int)(hi - lo)};
    }
    return max_taps;
}

}  // namespace host

LayoutDetector::LayoutDetector(const uint8_t* onnx, size_t len, const LayoutCfg& cfg) : cfg_(cfg) {
    OAR_CHECK(cfg_.input_h > 0 && cfg_.input_w > 0 && cfg_.input_h <= 8192 && cfg_.input_w <= 8192, OAR_INVALID_INPUT, "layout: image_shape must be in 1..=8192");
    OAR_CHECK(cfg_.filter >= 0 && cfg_.filter <= 2, OAR_INVALID_INPUT, "layout: resize filter must be 0 (Triangle), 1 (CatmullRom) or 2 (Lanczos3)");
    OAR_CHECK(cfg_.model_type >= 0 && cfg_.model_type <= 2, OAR_INVALID_INPUT, "layout: model_type must be 0 (picodet), 1 (rtdetr) or 2 (pp-doclayout)");
    OAR_CHECK(cfg_.num_classes > 0 && cfg_.max_detections > 0 && cfg_.max_detections <= 4096, OAR_INVALID_INPUT, "layout: num_classes / max_detections out of range");
    for (int c = 0; c < 3; ++c) OAR_CHECK(cfg_.stdv[c] > 0.0f, OAR_INVALID_INPUT, "layout: std must be positive");   // ScaleAwareDetectorPreprocessConfig::validate
    eng_.reset(new Engine(onnx, len, cfg_.device_id));
    // the graph's declared inputs decide the inference mode (pp_doclayout.rs:60-70): "image", "scale_factor" [, "im_shape"]
    const auto& ins = eng_->input_infos();
    OAR_CHECK(ins.size() == 2 || ins.size() == 3, OAR_MODEL_LOAD, "layout: the graph must declare image + scale_factor (+ im_shape) inputs");
    OAR_CHECK(ins[0].name == "image", OAR_MODEL_LOAD, "layout: the first graph input must be \"image\"");
    bool have_sf = false;
    for (size_t i = 1; i < ins.size(); ++i) {
        if (ins[i].name == "scale_factor") have_sf = true;
        else if (ins[i].name == "im_shape") wants_im_shape_ = true;
        else fail(OAR_MODEL_LOAD, "layout: unexpected graph input \"" + ins[i].name + "\"");
    }
    OAR_CHECK(have_sf, OAR_MODEL_LOAD, "layout: the graph declares no \"scale_factor\" input");
}

const float* LayoutDetector::preprocess(const std::vector<Image>& images, size_t i0, size_t n, std::vector<float>& scale_factor, std::vector<float>& src_wh) {
    hipStream_t s = eng_->stream();
    const int th = (int)cfg_.input_h, tw = (int)cfg_.input_w;
    const size_t plane = (size_t)th * tw;
    // pages: small ones are padded on the host first (DetResizeForTest::image_padding, resize_detection.rs:174-176,204-220)
    struct Src { const uint8_t* dev; int w, h; };
    std::vector<Src> srcs(n);
    std::vector<std::vector<uint8_t>> padded(n);
    size_t stage = 0;
    for (size_t i = 0; i < n; ++i) {
        const Image& im = images[i0 + i];
        OAR_CHECK(im.w > 0 && im.h > 0 && (im.host || im.dev), OAR_INVALID_INPUT, "layout: empty image");
        uint32_t w = im.w, h = im.h;
        if (im.w + im.h < 64) {
            OAR_CHECK(im.host, OAR_UNSUPPORTED_OP, "layout: device pages smaller than 64 pixels in total must be passed from the host (they are padded first)");
            w = std::max<uint32_t>(32, im.w); h = std::max<uint32_t>(32, im.h);
            padded[i].assign((size_t)w * h * 3, 0);
            for (uint32_t y = 0; y < im.h; ++y) std::memcpy(padded[i].data() + (size_t)y * w * 3, im.host + (size_t)y * im.w * 3, (size_t)im.w * 3);
        }
        srcs[i] = Src{im.dev, (int)w, (int)h};
        if (!im.dev || !padded[i].empty()) stage += ((size_t)w * h * 3 + 255) & ~(size_t)255;
        // ImageScaleInfo (src dims before padding) and the scale factors from the shapes the model sees (scale_aware_detector.rs:171-174, 221-246)
        src_wh.push_back((float)im.w); src_wh.push_back((float)im.h);
        scale_factor.push_back((float)th / (float)im.h); scale_factor.push_back((float)tw / (float)im.w);   // orig_shapes are the caller's images (:171-174)
    }
    OAR_HIP(hipStreamSynchronize(s));   // staging is reused by every batch
    stage_dev_.reserve(stage); stage_host_.reserve(stage);
    resized_dev_.reserve(plane * 3 * n); input_f32_.reserve(plane * 12 * n);
    int max_w = 1;
    for (const Src& q : srcs) max_w = std::max(max_w, q.w);
    tmp_f32_.reserve((size_t)th * max_w * 12);   // the vertical pass's f32 image: th rows of the widest source
    size_t off = 0;
    for (size_t i = 0; i < n; ++i) {
        const Image& im = images[i0 + i];
        if (im.dev && padded[i].empty()) continue;
        const size_t bytes = (size_t)srcs[i].w * srcs[i].h * 3;
        std::memcpy(stage_host_.as<uint8_t>() + off, padded[i].empty() ? im.host : padded[i].data(), bytes);
        srcs[i].dev = stage_dev_.as<uint8_t>() + off;
        off += (bytes + 255) & ~(size_t)255;
    }
    if (stage) OAR_HIP(hipMemcpyAsync(stage_dev_.p, stage_host_.p, stage, hipMemcpyHostToDevice, s));
    // taps per distinct source size (pages of one batch usually share it)
    struct TapSet { int w, h, max_v, max_h; size_t tv, wv, th, wh; };
    std::vector<TapSet> sets;
    std::vector<uint8_t> blob;
    auto put = [&](const void* p, size_t bytes) { const size_t at = (blob.size() + 255) & ~(size_t)255; blob.resize(at + bytes); std::memcpy(blob.data() + at, p, bytes); return at; };
    for (size_t i = 0; i < n; ++i) {
        if (srcs[i].w == tw && srcs[i].h == th) continue;
        bool have = false;
        for (auto& t : sets) have = have || (t.w == srcs[i].w && t.h == srcs[i].h);
        if (have) continue;
        std::vector<pp::FilterTaps> tv, thh;
        std::vector<float> wv, wh;
        TapSet t{srcs[i].w, srcs[i].h, 0, 0, 0, 0, 0, 0};
        t.max_v = host::filter_taps(cfg_.filter, srcs[i].h, th, tv, wv);
        t.max_h = host::filter_taps(cfg_.filter, srcs[i].w, tw, thh, wh);
        t.tv = put(tv.data(), tv.size() * sizeof(pp::FilterTaps)); t.wv = put(wv.data(), wv.size() * 4);
        t.th = put(thh.data(), thh.size() * sizeof(pp::FilterTaps)); t.wh = put(wh.data(), wh.size() * 4);
        sets.push_back(t);
    }
    if (!blob.empty()) {
        taps_dev_.reserve(blob.size());
        OAR_HIP(hipMemcpyAsync(taps_dev_.p, blob.data(), blob.size(), hipMemcpyHostToDevice, s));
        OAR_HIP(hipStreamSynchronize(s));   // blob is a local
    }
    for (size_t i = 0; i < n; ++i) {
        uint8_t* dst = resized_dev_.as<uint8_t>() + i * plane * 3;
        if (srcs[i].w == tw && srcs[i].h == th) {   // "Return original if no resize is needed" (resize_detection.rs:356-358)
            OAR_HIP(hipMemcpyAsync(dst, srcs[i].dev, plane * 3, hipMemcpyDeviceToDevice, s));
            continue;
        }
        const TapSet* t = nullptr;
        for (auto& q : sets) if (q.w == srcs[i].w && q.h == srcs[i].h) t = &q;
        const uint8_t* tb = taps_dev_.as<uint8_t>();
        pp::resize_filter(s, srcs[i].dev, srcs[i].w, srcs[i].h, dst, tw, th, reinterpret_cast<const pp::FilterTaps*>(tb + t->tv), reinterpret_cast<const float*>(tb + t->wv), t->max_v,
                          reinterpret_cast<const pp::FilterTaps*>(tb + t->th), reinterpret_cast<const float*>(tb + t->wh), t->max_h, tmp_f32_.as<float>());
    }
    // NormalizeImage::with_color_order_from_rgb_stats (normalization.rs:241-273): statistics permuted into the output channel order, CHW
    int srcc[3];
    float alpha[3], beta[3];
    for (int c = 0; c < 3; ++c) {
        const int sc = cfg_.bgr ? 2 - c : c;
        srcc[c] = sc;
        alpha[c] = cfg_.scale / cfg_.stdv[sc];
        beta[c] = -cfg_.mean[sc] / cfg_.stdv[sc];
    }
    pp::normalize(s, resized_dev_.as<uint8_t>(), input_f32_.as<float>(), (int64_t)n, (int64_t)plane, srcc, alpha, beta, 0);
    return input_f32_.as<float>();
}

void LayoutDetector::preprocess_only(const Image& im, std::vector<float>& chw) {
    std::lock_guard<std::mutex> lk(mu_);
    OAR_HIP(hipSetDevice(eng_->device()));
    std::vector<float> sf, wh;
    const float* d = preprocess({im}, 0, 1, sf, wh);
    chw.resize((size_t)3 * cfg_.input_h * cfg_.input_w);
    OAR_HIP(hipMemcpyAsync(chw.data(), d, chw.size() * 4, hipMemcpyDeviceToHost, eng_->stream()));
    OAR_HIP(hipStreamSynchronize(eng_->stream()));
}

void LayoutDetector::run(const std::vector<Image>& images, LayoutOut& out) { run_impl(images, nullptr, out); }
void LayoutDetector::run_ppdoc(const std::vector<Image>& images, const PpDocCfg& pc, LayoutOut& out) { run_impl(images, &pc, out); }

void LayoutDetector::run_impl(const std::vector<Image>& images, const PpDocCfg* pc, LayoutOut& out) {
    std::lock_guard<std::mutex> lk(mu_);
    OAR_HIP(hipSetDevice(eng_->device()));
    hipStream_t s = eng_->stream();
    out = LayoutOut();
    out.offsets.assign(1, 0);
    OAR_CHECK(!images.empty(), OAR_INVALID_INPUT, "images must not be empty");   // validate_non_empty (scale_aware_detector.rs:170)
    constexpr size_t kBatch = 8;   // LayoutDetectionAdapter::recommended_batch_size
    for (size_t i0 = 0; i0 < images.size(); i0 += kBatch) {
        const size_t n = std::min(kBatch, images.size() - i0);
        std::vector<float> sf, wh;
        const float* in = preprocess(images, i0, n, sf, wh);
        // auxiliary graph inputs: scale_factor [n, 2] = (scale_y, scale_x); im_shape [n, 2] = resized (h, w)
        std::vector<float> aux(sf);
        for (size_t i = 0; i < n; ++i) { aux.push_back((float)cfg_.input_h); aux.push_back((float)cfg_.input_w); }
        aux.insert(aux.end(), wh.begin(), wh.end());
        aux_dev_.reserve(aux.size() * 4);
        OAR_HIP(hipMemcpyAsync(aux_dev_.p, aux.data(), aux.size() * 4, hipMemcpyHostToDevice, s));
        OAR_HIP(hipStreamSynchronize(s));   // aux is a local
        const float* d_sf = aux_dev_.as<float>();
        const float* d_shape = d_sf + 2 * n;
        const float* d_wh = d_shape + 2 * n;
        std::vector<const float*> ins{in};
        std::vector<std::vector<int64_t>> dims{{(int64_t)n, 3, (int64_t)cfg_.input_h, (int64_t)cfg_.input_w}};
        for (size_t k = 1; k < eng_->input_infos().size(); ++k) {
            ins.push_back(eng_->input_infos()[k].name == "im_shape" ? d_shape : d_sf);
            dims.push_back({(int64_t)n, 2});
        }
        const Plan& plan = eng_->run_multi(ins, dims);
        OAR_CHECK(!plan.outputs.empty(), OAR_INVALID_INPUT, "No output tensors available from model");   // scale_aware_detector.rs:303-305
        const PlanOutput& po = plan.outputs[0];
        OAR_CHECK(!po.on_host, OAR_UNSUPPORTED_OP, "layout: the detection output was constant-folded on the host");
        // 2-D [n * boxes, 6 | 7 | 8] or 4-D [n, boxes, 1, feat] (scale_aware_detector.rs:307-352); a 3-D [n, boxes, feat] is taken the same way
        int rows = 0, feat = 0;
        if (po.dims.size() == 2) {
            feat = (int)po.dims[1];
            OAR_CHECK(feat == 6 || feat == 7 || feat == 8, OAR_INVALID_INPUT, "Expected box dimension 6, 7, or 8, got " + std::to_string(feat));
            OAR_CHECK(po.dims[0] % (int64_t)n == 0, OAR_INVALID_INPUT, "2D detector output has " + std::to_string(po.dims[0]) + " boxes, not divisible by batch size " + std::to_string(n));
            rows = (int)(po.dims[0] / (int64_t)n);
        } else if (po.dims.size() == 4 || po.dims.size() == 3) {
            OAR_CHECK(po.dims[0] == (int64_t)n, OAR_SHAPE_MISMATCH, "layout: output batch differs from the input batch");
            feat = (int)po.dims.back();
            rows = 1;
            for (size_t d = 1; d + 1 < po.dims.size(); ++d) rows *= (int)po.dims[d];
        } else {
            fail(OAR_INVALID_INPUT, "layout: unexpected output rank " + std::to_string(po.dims.size()));
        }
        out.feature_dim = (uint32_t)feat;
        if (rows == 0 || feat == 0) { for (size_t i = 0; i < n; ++i) out.offsets.push_back((uint32_t)out.scores.size()); continue; }
        OAR_CHECK(rows <= 16384, OAR_UNSUPPORTED_OP, "layout: more than 16384 candidate rows per image");   // (one workgroup ranks an image: O(rows^2); rows bytes of LDS)
        if (pc) {
            // ---- the PP-DocLayout adapter's post-processing (layout.hip ppdoc_post_kernel): the kept rows come back in their final order
            OAR_CHECK(feat >= 6 && feat <= 8, OAR_INVALID_INPUT, "pp-doclayout: expected 6, 7 or 8 prediction columns, got " + std::to_string(feat));
            cand_dev_.reserve((size_t)n * rows * 8 * 4); sorted_dev_.reserve((size_t)n * rows * 4); keep_dev_.reserve((size_t)n * (rows + 1) * 4);
            const size_t nc = cfg_.num_classes;
            DevBuf cfg_dev;
            cfg_dev.reserve(nc * 8 + 16);
            if (pc->class_thr) OAR_HIP(hipMemcpyAsync(cfg_dev.p, pc->class_thr, nc * 4, hipMemcpyHostToDevice, s));
            if (pc->merge_mode) OAR_HIP(hipMemcpyAsync(cfg_dev.as<uint8_t>() + nc * 4, pc->merge_mode, nc * 4, hipMemcpyHostToDevice, s));
            pp::PpDocPostP q{};
            q.pred = eng_->out_ptr(po.loc); q.rows = rows; q.feat = feat; q.num_classes = (int)nc; q.score_thr = pc->score_threshold;
            q.class_thr = pc->class_thr ? cfg_dev.as<float>() : nullptr; q.layout_nms = pc->layout_nms ? 1 : 0; q.image_class = pc->image_class; q.formula_class = pc->formula_class;
            q.merge_mode = pc->merge_mode ? reinterpret_cast<const int*>(cfg_dev.as<uint8_t>() + nc * 4) : nullptr;
            q.src_wh = d_wh; q.cand = cand_dev_.as<float>(); q.sorted = sorted_dev_.as<int>(); q.keep = keep_dev_.as<int>(); q.n_keep = keep_dev_.as<int>() + (size_t)n * rows;
            pp::ppdoc_postprocess(s, q, (int)n);
            std::vector<int> keep((size_t)n * (rows + 1));
            std::vector<float> cand((size_t)n * rows * 8);
            OAR_HIP(hipMemcpyAsync(keep.data(), keep_dev_.p, keep.size() * 4, hipMemcpyDeviceToHost, s));
            OAR_HIP(hipMemcpyAsync(cand.data(), cand_dev_.p, cand.size() * 4, hipMemcpyDeviceToHost, s));
            OAR_HIP(hipStreamSynchronize(s));   // (also: the configuration copies have left their host sources)
            for (size_t i = 0; i < n; ++i) {
                const int nk = keep[(size_t)n * rows + i];
                for (int k = 0; k < nk; ++k) {
                    const float* c8 = cand.data() + (i * (size_t)rows + (size_t)keep[i * (size_t)rows + k]) * 8;
                    out.boxes.insert(out.boxes.end(), c8, c8 + 4);
                    out.scores.push_back(c8[4]);
                    int32_t cls; std::memcpy(&cls, &c8[5], 4);
                    out.classes.push_back(cls);
                }
                out.offsets.push_back((uint32_t)out.scores.size());
            }
            continue;
        }
        cand_dev_.reserve((size_t)n * rows * 8 * 4); sorted_dev_.reserve((size_t)n * rows * 4); keep_dev_.reserve((size_t)n * (cfg_.max_detections + 1) * 4);
        pp::LayoutPostP p{};
        p.pred = eng_->out_ptr(po.loc); p.rows = rows; p.feat = feat; p.num_classes = (int)cfg_.num_classes; p.model_type = cfg_.model_type; p.max_det = (int)cfg_.max_detections;
        p.score_thr = cfg_.score_threshold; p.nms_thr = cfg_.nms_threshold; p.src_wh = d_wh;
        p.cand = cand_dev_.as<float>(); p.sorted = sorted_dev_.as<int>(); p.keep = keep_dev_.as<int>(); p.n_keep = keep_dev_.as<int>() + (size_t)n * cfg_.max_detections;
        pp::layout_postprocess(s, p, (int)n);
        std::vector<int> keep((size_t)n * (cfg_.max_detections + 1));
        std::vector<float> cand((size_t)n * rows * 8), pred;
        OAR_HIP(hipMemcpyAsync(keep.data(), keep_dev_.p, keep.size() * 4, hipMemcpyDeviceToHost, s));
        OAR_HIP(hipMemcpyAsync(cand.data(), cand_dev_.p, cand.size() * 4, hipMemcpyDeviceToHost, s));
        const bool reading_order = cfg_.model_type == 2 && feat == 8;
        if (reading_order) { pred.resize((size_t)n * rows * feat); OAR_HIP(hipMemcpyAsync(pred.data(), p.pred, pred.size() * 4, hipMemcpyDeviceToHost, s)); }
        OAR_HIP(hipStreamSynchronize(s));
        for (size_t i = 0; i < n; ++i) {
            const int nk = keep[(size_t)n * cfg_.max_detections + i];
            std::vector<int> rowsk(keep.begin() + (long)(i * cfg_.max_detections), keep.begin() + (long)(i * cfg_.max_detections) + nk);
            if (reading_order && nk > 1) {   // (col, row) ascending by f32::total_cmp, stable (layout_postprocess.rs:309-320)
                auto key = [](float v) { int32_t b; std::memcpy(&b, &v, 4); return b ^ (int32_t)(((uint32_t)(b >> 31)) >> 1); };
                const float* pr = pred.data() + i * (size_t)rows * feat;
                std::stable_sort(rowsk.begin(), rowsk.end(), [&](int a, int b) {
                    const int32_t ca = key(pr[(size_t)a * feat + 6]), cb = key(pr[(size_t)b * feat + 6]);
                    if (ca != cb) return ca < cb;
                    return key(pr[(size_t)a * feat + 7]) < key(pr[(size_t)b * feat + 7]);
                });
            }
            for (int r : rowsk) {
                const float* c8 = cand.data() + (i * (size_t)rows + (size_t)r) * 8;
                out.boxes.insert(out.boxes.end(), c8, c8 + 4);
                out.scores.push_back(c8[4]);
                int32_t cls; std::memcpy(&cls, &c8[5], 4);
                out.classes.push_back(cls);
            }
            out.offsets.push_back((uint32_t)out.scores.size());
        }
    }
    if (Profiler::get().enabled) Profiler::get().flush();
}


namespace host {
// apply_nms_with_merge: groups grow greedily in score order -- a seed box absorbs every later-ranked box of its class that overlaps the GROWING merged box
// by more than the threshold; the group's box is merged per the class's mode, its score is the group maximum, its place the earliest input index.
int nms_with_merge(const float* boxes, const int32_t* classes, const float* scores, int n, const int32_t* mode_of_class, int num_classes, float nms_thr, int max_det,
                   float* out_boxes, int32_t* out_classes, float* out_scores) {
    if (n <= 0) return 0;
    struct Group { float b[4]; int32_t cls; float score; int first; };
    std::vector<int> order((size_t)n);
    for (int i = 0; i < n; ++i) order[(size_t)i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return scores[b] < scores[a]; });   // descending; unordered pairs stay put
    std::vector<char> used((size_t)n, 0);
    std::vector<Group> groups;
    auto area = [](const float* b) { return (b[2] - b[0]) * (b[3] - b[1]); };
    auto iou = [&](const float* a, const float* b) {
        const float x0 = std::max(a[0], b[0]), y0 = std::max(a[1], b[1]), x1 = std::min(a[2], b[2]), y1 = std::min(a[3], b[3]);
        if (x1 <= x0 || y1 <= y0) return 0.0f;
        const float inter = (x1 - x0) * (y1 - y0), uni = area(a) + area(b) - inter;
        return uni > 0.0f ? inter / uni : 0.0f;
    };
    for (int seed : order) {
        if (used[(size_t)seed]) continue;
        used[(size_t)seed] = 1;
        Group gq;
        std::memcpy(gq.b, boxes + 4 * (size_t)seed, 16); gq.cls = classes[seed]; gq.score = scores[seed]; gq.first = seed;
        const int mode = gq.cls >= 0 && gq.cls < num_classes ? mode_of_class[gq.cls] : 0;
        for (int other : order) {
            if (other == seed || used[(size_t)other] || classes[other] != gq.cls) continue;
            const float* ob = boxes + 4 * (size_t)other;
            if (!(iou(gq.b, ob) > nms_thr)) continue;
            if (mode == 1) { gq.b[0] = std::min(gq.b[0], ob[0]); gq.b[1] = std::min(gq.b[1], ob[1]); gq.b[2] = std::max(gq.b[2], ob[2]); gq.b[3] = std::max(gq.b[3], ob[3]); }
            else if (mode == 2 ? !(area(gq.b) <= area(ob)) : !(area(gq.b) >= area(ob))) std::memcpy(gq.b, ob, 16);
            gq.score = std::max(gq.score, scores[other]);
            gq.first = std::min(gq.first, other);
            used[(size_t)other] = 1;
        }
        groups.push_back(gq);
    }
    if ((int)groups.size() > max_det) groups.resize((size_t)std::max(max_det, 0));   // the groups are in score order: the best max_det survive
    std::stable_sort(groups.begin(), groups.end(), [](const Group& a, const Group& b) { return a.first < b.first; });
    for (size_t i = 0; i < groups.size(); ++i) { std::memcpy(out_boxes + 4 * i, groups[i].b, 16); out_classes[i] = groups[i].cls; out_scores[i] = groups[i].score; }
    return (int)groups.size();
}
}  // namespace host

}  // namespace oar
