// c_api.cc -- extern "C" boundary (include/oar_mi355x.h). No exception crosses it.
#include <condition_variable>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <thread>

#include "engine.h"
#include "layout.h"
#include "pipeline.h"

namespace oar {
const std::string& last_error();
extern std::atomic<int> g_inject_batched_det_failures;   // pipeline.cc
}
using namespace oar;

struct oar_engine { std::unique_ptr<Engine> e; PinBuf view; };   // view: staging for oar_engine_run_first_f32
struct oar_det { std::unique_ptr<Detector> d; };
struct oar_rec { std::unique_ptr<Recognizer> r; };
// One pipeline handle = `lanes` complete pipelines (own engines, streams, staging, geometry pool) over the same model bytes.  Lane 0 serves
// the synchronous oar_ocr_predict; oar_ocr_predict_async hands a call to the next lane's worker thread, so call k + 1's upload and
// detection run while call k recognises (the ~15 % of the GPU a synchronous predict leaves idle).  A call is still one
// OAROCR::predict on one lane: crops are pooled across the pages of THAT call only (ocr.rs:594-634), results are those of the
// synchronous call.
struct OcrLane {
    std::unique_ptr<Ocr> o;
    std::thread worker;
    std::mutex mu;
    std::condition_variable cv;
    struct Job {
        uint64_t ticket = 0;
        std::vector<PageRef> pages;
        bool done = false;
        oar_status status = OAR_OK;
        std::string error;
        std::vector<std::vector<OcrRegion>> res;
        std::vector<Ocr::PageMeta> meta;
    };
    std::deque<std::shared_ptr<Job>> queue;
    bool stop = false;
};
struct oar_ocr {
    std::unique_ptr<Ocr> o;                         // lane 0
    std::vector<std::unique_ptr<OcrLane>> lanes;    // all lanes (lanes[0]->o is null: it borrows `o`)
    std::mutex mu;                                  // tickets / job table
    std::condition_variable done_cv;
    uint64_t next_ticket = 1;
    std::map<uint64_t, std::shared_ptr<OcrLane::Job>> jobs;
    Ocr& lane_ocr(size_t i) { return i == 0 ? *o : *lanes[i]->o; }
    ~oar_ocr() {
        for (auto& l : lanes) {
            { std::lock_guard<std::mutex> lk(l->mu); l->stop = true; }
            l->cv.notify_all();
            if (l->worker.joinable()) l->worker.join();
        }
    }
};
struct oar_cls { std::unique_ptr<Classifier> c; };
struct oar_rect { std::unique_ptr<Rectifier> r; };
struct oar_layout { std::unique_ptr<LayoutDetector> l; };

#include "jpeg_decode.h"
#include "jpeg_dev.h"
namespace oar { namespace img {
bool is_png(const uint8_t* b, size_t n);
const char* sniff(const uint8_t* b, size_t n);
void decode_png(const uint8_t* b, size_t n, std::vector<uint8_t>& rgb, uint32_t& width, uint32_t& height);
bool decode_misc(const uint8_t* b, size_t n, std::vector<uint8_t>& rgb, uint32_t& width, uint32_t& height);   // image_misc_decode.cc: BMP, PNM, TIFF, GIF
} }

namespace {
template <typename F>
oar_status guard(F&& f) {
    try {
        f();
        return OAR_OK;
    } catch (const Error& e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::bad_alloc&) {
        set_last_error("host allocation failed");
        return OAR_OOM;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return OAR_INTERNAL;
    } catch (...) {
        set_last_error("unknown error");
        return OAR_INTERNAL;
    }
}
template <typename T>
T* cmalloc(size_t n) {
    T* p = (T*)std::malloc(std::max<size_t>(n, 1) * sizeof(T));
    if (!p) throw std::bad_alloc();
    return p;
}
void require_device() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) fail(OAR_DEVICE, "no HIP device visible: libOarMi355x has no CPU fallback");
}
void fill_det_result(const std::vector<DetBoxes>& boxes, oar_det_result* out) {
    std::memset(out, 0, sizeof *out);
    out->n_images = (uint32_t)boxes.size();
    size_t total = 0;
    for (auto& b : boxes) total += b.scores.size();
    out->n_boxes = (uint32_t)total;
    bool poly = false;
    size_t floats = 0;
    for (auto& b : boxes) { poly = poly || !b.counts.empty(); floats += b.pts.size(); }
    out->n_points = (uint32_t)(floats / 2);
    out->box_offsets = cmalloc<uint32_t>(boxes.size() + 1);
    out->points = cmalloc<float>(floats);
    out->scores = cmalloc<float>(total);
    if (poly) out->point_offsets = cmalloc<uint32_t>(total + 1);
    size_t k = 0, f = 0;
    for (size_t i = 0; i < boxes.size(); ++i) {
        out->box_offsets[i] = (uint32_t)k;
        if (poly) {
            size_t at = f / 2;
            for (size_t b = 0; b < boxes[i].counts.size(); ++b) { out->point_offsets[k + b] = (uint32_t)at; at += boxes[i].counts[b]; }
        }
        std::memcpy(out->points + f, boxes[i].pts.data(), boxes[i].pts.size() * sizeof(float));
        std::memcpy(out->scores + k, boxes[i].scores.data(), boxes[i].scores.size() * sizeof(float));
        k += boxes[i].scores.size();
        f += boxes[i].pts.size();
    }
    out->box_offsets[boxes.size()] = (uint32_t)k;
    if (poly) out->point_offsets[total] = (uint32_t)(f / 2);
}
}  // namespace

extern "C" {

size_t oar_last_error(char* buf, size_t cap) {
    const std::string& m = last_error();
    if (buf && cap) {
        size_t n = std::min(cap - 1, m.size());
        std::memcpy(buf, m.data(), n);
        buf[n] = 0;
    }
    return m.size();
}

size_t oar_version(char* buf, size_t cap) {
    char tmp[256];
    int n = 0;
    hipDeviceProp_t prop;
    if (hipGetDeviceCount(&n) == hipSuccess && n > 0 && hipGetDeviceProperties(&prop, 0) == hipSuccess)
        snprintf(tmp, sizeof tmp, "libOarMi355x 0.1.0 %s %s CUs=%d", prop.gcnArchName, prop.name, prop.multiProcessorCount);
    else
        snprintf(tmp, sizeof tmp, "libOarMi355x 0.1.0 (no HIP device)");
    size_t len = std::strlen(tmp);
    if (buf && cap) {
        size_t c = std::min(cap - 1, len);
        std::memcpy(buf, tmp, c);
        buf[c] = 0;
    }
    return len;
}

int oar_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ---------------------------------------------------------------------------------------------- Seam A
oar_status oar_engine_create(const uint8_t* onnx, size_t onnx_len, const oar_engine_cfg* cfg, oar_engine** out) {
    return guard([&] {
        OAR_CHECK(out, OAR_INVALID_INPUT, "oar_engine_create: out is null");
        *out = nullptr;
        std::unique_ptr<oar_engine> h(new oar_engine());
        OAR_CHECK(!cfg || cfg->precision == OAR_PRECISION_F32, OAR_INVALID_INPUT, "oar_engine_create: unsupported precision (only OAR_PRECISION_F32 = 0 exists: f32 / bf16x6 arithmetic)");
        h->e.reset(new Engine(onnx, onnx_len, cfg ? cfg->device_id : 0, cfg ? static_cast<hipStream_t>(cfg->stream) : nullptr));
        if (cfg && cfg->profile) Profiler::get().enabled = true;
        *out = h.release();
    });
}
void oar_engine_destroy(oar_engine* e) { delete e; }

oar_status oar_engine_input_name(const oar_engine* e, char* buf, size_t cap) {
    return guard([&] {
        OAR_CHECK(e && buf && cap, OAR_INVALID_INPUT, "oar_engine_input_name: bad arguments");
        snprintf(buf, cap, "%s", e->e->input_name().c_str());
    });
}

namespace {
// copies every plan output back to the host (F32 as floats; I64 either from the plan-time host value or by converting the
// integer-valued f32 device tensor).  The caller holds the engine mutex and synchronises the stream afterwards.
struct PendingI64 { oar_tensor* t; std::vector<float> tmp; int64_t n; };
void fetch_outputs(Engine& E, const Plan& p, oar_tensor* outs, int32_t max_out, int32_t* n_out) {
    OAR_CHECK((int)p.outputs.size() <= max_out, OAR_INVALID_INPUT, "oar_engine_run: outs too small");
    for (int i = 0; i < max_out; ++i) std::memset(&outs[i], 0, sizeof outs[i]);
    std::vector<PendingI64> pend;
    pend.reserve(p.outputs.size());
    try {
        for (size_t i = 0; i < p.outputs.size(); ++i) {
            const PlanOutput& po = p.outputs[i];
            oar_tensor& t = outs[i];
            OAR_CHECK(po.dims.size() <= 8, OAR_UNSUPPORTED_OP, "graph output of rank > 8");
            t.rank = (int32_t)po.dims.size();
            int64_t n = 1;
            for (size_t k = 0; k < po.dims.size(); ++k) { t.dims[k] = po.dims[k]; n *= po.dims[k]; }
            snprintf(t.name, sizeof t.name, "%s", po.name.c_str());
            t.dtype = po.dtype == 7 ? OAR_DTYPE_I64 : OAR_DTYPE_F32;
            if (po.dtype == 7) {
                t.data_i64 = cmalloc<int64_t>((size_t)std::max<int64_t>(n, 1));
                if (po.on_host) {
                    for (int64_t k = 0; k < n; ++k) t.data_i64[k] = (int64_t)po.host_vals[k];
                } else {
                    pend.push_back(PendingI64{&t, std::vector<float>((size_t)n), n});
                    if (n) OAR_HIP(hipMemcpyAsync(pend.back().tmp.data(), E.out_ptr(po.loc), (size_t)n * 4, hipMemcpyDeviceToHost, E.stream()));
                }
            } else {
                t.data = cmalloc<float>((size_t)std::max<int64_t>(n, 1));
                if (po.on_host) {
                    for (int64_t k = 0; k < n; ++k) t.data[k] = (float)po.host_vals[k];
                } else if (n) {
                    OAR_HIP(hipMemcpyAsync(t.data, E.out_ptr(po.loc), (size_t)n * 4, hipMemcpyDeviceToHost, E.stream()));
                }
            }
        }
        OAR_HIP(hipStreamSynchronize(E.stream()));
    } catch (...) {
        (void)hipStreamSynchronize(E.stream());
        for (int i = 0; i < max_out; ++i) oar_tensor_free(&outs[i]);
        throw;
    }
    // integer outputs live on the device as f32: exact up to 2^24, beyond that the value the graph meant is gone -- fail instead of
    // handing back a rounded neighbour (index-valued outputs, ArgMax and friends, are far below that)
    bool lossy = false;
    for (auto& q : pend)
        for (int64_t k = 0; k < q.n; ++k) {
            const float v = q.tmp[(size_t)k];
            lossy = lossy || !(std::fabs(v) <= 16777216.0f);
            q.t->data_i64[k] = (int64_t)std::llround((double)v);
        }
    if (lossy) {
        for (int i = 0; i < max_out; ++i) oar_tensor_free(&outs[i]);
        fail(OAR_UNSUPPORTED_OP, "an integer graph output exceeds 2^24 (or is not finite): it cannot be represented by the engine's f32 device tensors");
    }
    *n_out = (int32_t)p.outputs.size();
}

// uploads the named inputs in the order the model declares them and runs the plan; `bufs` keeps the device copies alive
const Plan& run_named(Engine& E, const oar_input* inputs, int32_t n_in, std::vector<std::unique_ptr<DevBuf>>& bufs) {
    const auto& infos = E.input_infos();
    OAR_CHECK(inputs && n_in > 0, OAR_INVALID_INPUT, "No inputs provided for inference");   // ort_infer_execution.rs:125-129
    OAR_CHECK((size_t)n_in == infos.size(), OAR_INVALID_INPUT,
              "the model declares " + std::to_string(infos.size()) + " input(s), " + std::to_string(n_in) + " given");
    std::vector<const float*> ptrs(infos.size(), nullptr);
    std::vector<std::vector<int64_t>> dims(infos.size());
    for (size_t k = 0; k < infos.size(); ++k) bufs.emplace_back(new DevBuf());
    for (int32_t i = 0; i < n_in; ++i) {
        const oar_input& in = inputs[i];
        OAR_CHECK(in.data && in.dims && in.rank >= 1 && in.rank <= 8, OAR_INVALID_INPUT, "input " + std::to_string(i) + ": bad tensor");
        size_t slot = infos.size();
        if (!in.name || !in.name[0]) slot = 0;
        else for (size_t k = 0; k < infos.size(); ++k) if (infos[k].name == in.name) slot = k;
        OAR_CHECK(slot < infos.size(), OAR_INVALID_INPUT, std::string("the model has no input named '") + (in.name ? in.name : "") + "'");
        OAR_CHECK(ptrs[slot] == nullptr, OAR_INVALID_INPUT, "input '" + infos[slot].name + "' given twice");
        // oar_input carries f32 only: a model that declares an integer / double / half input would be fed reinterpreted floats
        OAR_CHECK(infos[slot].elem_type == 0 || infos[slot].elem_type == 1, OAR_UNSUPPORTED_OP,
                  "input '" + infos[slot].name + "' is declared with ONNX element type " + std::to_string(infos[slot].elem_type) + "; only f32 inputs can be bound");
        int64_t cnt = 1;
        for (int32_t k = 0; k < in.rank; ++k) { OAR_CHECK(in.dims[k] >= 0, OAR_INVALID_INPUT, "negative dimension"); cnt *= in.dims[k]; }
        dims[slot].assign(in.dims, in.dims + in.rank);
        bufs[slot]->reserve((size_t)std::max<int64_t>(cnt, 1) * 4);
        if (cnt) OAR_HIP(hipMemcpyAsync(bufs[slot]->p, in.data, (size_t)cnt * 4, hipMemcpyHostToDevice, E.stream()));
        ptrs[slot] = bufs[slot]->as<float>();
    }
    return E.run_multi(ptrs, dims);
}
}  // namespace

oar_status oar_engine_run(oar_engine* e, const float* input, const int64_t* dims, int32_t rank, oar_tensor* outs, int32_t max_out,
                          int32_t* n_out) {
    return guard([&] {
        OAR_CHECK(e && input && dims && rank > 0 && rank <= 8 && outs && n_out, OAR_INVALID_INPUT, "oar_engine_run: bad arguments");
        Engine& E = *e->e;
        std::lock_guard<std::mutex> lk(E.mutex());
        OAR_HIP(hipSetDevice(E.device()));
        OAR_CHECK(E.input_infos().size() == 1, OAR_INVALID_INPUT, "oar_engine_run: the model declares several inputs, use oar_engine_run_named");
        std::vector<int64_t> d(dims, dims + rank);
        int64_t cnt = 1;
        for (auto v : d) { OAR_CHECK(v >= 0, OAR_INVALID_INPUT, "negative dimension"); cnt *= v; }
        DevBuf din;
        din.reserve((size_t)std::max<int64_t>(cnt, 1) * 4);
        OAR_HIP(hipMemcpyAsync(din.p, input, (size_t)cnt * 4, hipMemcpyHostToDevice, E.stream()));
        const Plan& p = E.run(din.as<float>(), d, false);
        fetch_outputs(E, p, outs, max_out, n_out);
        if (Profiler::get().enabled) Profiler::get().flush();
    });
}

oar_status oar_engine_run_named(oar_engine* e, const oar_input* inputs, int32_t n_in, oar_tensor* outs, int32_t max_out, int32_t* n_out) {
    return guard([&] {
        OAR_CHECK(e && outs && n_out, OAR_INVALID_INPUT, "oar_engine_run_named: bad arguments");
        Engine& E = *e->e;
        std::lock_guard<std::mutex> lk(E.mutex());
        OAR_HIP(hipSetDevice(E.device()));
        std::vector<std::unique_ptr<DevBuf>> bufs;
        const Plan& p = run_named(E, inputs, n_in, bufs);
        fetch_outputs(E, p, outs, max_out, n_out);
        if (Profiler::get().enabled) Profiler::get().flush();
    });
}

oar_status oar_engine_run_first_f32(oar_engine* e, const oar_input* inputs, int32_t n_in, oar_output_view_fn fn, void* user) {
    return guard([&] {
        OAR_CHECK(e && fn, OAR_INVALID_INPUT, "oar_engine_run_first_f32: bad arguments");
        Engine& E = *e->e;
        std::lock_guard<std::mutex> lk(E.mutex());
        OAR_HIP(hipSetDevice(E.device()));
        std::vector<std::unique_ptr<DevBuf>> bufs;
        const Plan& p = run_named(E, inputs, n_in, bufs);
        OAR_CHECK(!p.outputs.empty(), OAR_INTERNAL, "no output returned from inference");
        const PlanOutput& po = p.outputs[0];
        OAR_CHECK(po.dtype == 1, OAR_SHAPE_MISMATCH, "output '" + po.name + "' is not an f32 tensor");   // ort_infer_execution.rs:283-290
        OAR_CHECK(po.dims.size() <= 8, OAR_UNSUPPORTED_OP, "graph output of rank > 8");
        int64_t n = 1;
        for (auto d : po.dims) n *= d;
        e->view.reserve((size_t)std::max<int64_t>(n, 1) * 4);
        if (po.on_host) {
            for (int64_t k = 0; k < n; ++k) e->view.as<float>()[k] = (float)po.host_vals[k];
        } else if (n) {
            OAR_HIP(hipMemcpyAsync(e->view.p, E.out_ptr(po.loc), (size_t)n * 4, hipMemcpyDeviceToHost, E.stream()));
        }
        OAR_HIP(hipStreamSynchronize(E.stream()));
        if (Profiler::get().enabled) Profiler::get().flush();
        const int32_t rc = fn(user, po.dims.data(), (int32_t)po.dims.size(), e->view.as<float>());
        OAR_CHECK(rc == 0, OAR_INVALID_INPUT, "output view callback failed with code " + std::to_string(rc));
    });
}

oar_status oar_engine_io(const oar_engine* e, oar_io_info* inputs, int32_t max_in, int32_t* n_in, oar_io_info* outputs, int32_t max_out,
                         int32_t* n_out) {
    return guard([&] {
        OAR_CHECK(e, OAR_INVALID_INPUT, "oar_engine_io: engine is null");
        auto fill = [](const std::vector<ValueInfo>& src, oar_io_info* dst, int32_t cap, int32_t* cnt) {
            if (cnt) *cnt = (int32_t)src.size();
            if (!dst) return;
            OAR_CHECK((int32_t)src.size() <= cap, OAR_INVALID_INPUT, "oar_engine_io: array too small");
            for (size_t i = 0; i < src.size(); ++i) {
                oar_io_info& o = dst[i];
                std::memset(&o, 0, sizeof o);
                snprintf(o.name, sizeof o.name, "%s", src[i].name.c_str());
                o.dtype = src[i].elem_type;
                o.rank = src[i].has_shape && src[i].dims.size() <= 8 ? (int32_t)src[i].dims.size() : -1;
                for (int32_t k = 0; k < o.rank; ++k) o.dims[k] = src[i].dims[(size_t)k];
            }
        };
        fill(e->e->input_infos(), inputs, max_in, n_in);
        fill(e->e->output_infos(), outputs, max_out, n_out);
    });
}

void oar_tensor_free(oar_tensor* t) {
    if (!t) return;
    if (t->data) { std::free(t->data); t->data = nullptr; }
    if (t->data_i64) { std::free(t->data_i64); t->data_i64 = nullptr; }
}

oar_status oar_engine_cache_stats(oar_engine* e, uint64_t* cached_plans, uint64_t* evicted_plans) {
    return guard([&] {
        OAR_CHECK(e, OAR_INVALID_INPUT, "oar_engine_cache_stats: engine is null");
        std::lock_guard<std::mutex> lk(e->e->mutex());
        if (cached_plans) *cached_plans = e->e->cached_plans();
        if (evicted_plans) *evicted_plans = e->e->evicted_plans();
    });
}

oar_status oar_onnx_inspect(const uint8_t* onnx, size_t onnx_len, char* summary, size_t cap) {
    if (summary && cap) summary[0] = 0;
    return guard([&] {
        OnnxModel m = parse_onnx(onnx, onnx_len);   // throws OAR_MODEL_LOAD on malformed / truncated files
        Engine::validate_model(m);
        std::map<std::string, int> hist;
        for (auto& n : m.nodes) hist[n.op]++;
        std::string text, missing;
        for (auto& kv : hist) {
            const bool ok = Engine::supported_ops().count(kv.first) != 0;
            text += (text.empty() ? "" : " ") + std::string(ok ? "" : "!") + kv.first + ":" + std::to_string(kv.second);
            if (!ok) missing += (missing.empty() ? "" : ", ") + kv.first;
        }
        text = "opset=" + std::to_string(m.opset) + " input=" + m.inputs[0] + " outputs=" + std::to_string(m.outputs.size()) + " initializers=" +
               std::to_string(m.initializers.size()) + " nodes=" + std::to_string(m.nodes.size()) + " | " + text;
        if (summary && cap) snprintf(summary, cap, "%s", text.c_str());
        OAR_CHECK(missing.empty(), OAR_UNSUPPORTED_OP, "operators not implemented: " + missing);
    });
}

oar_status oar_engine_cost(oar_engine* e, const int64_t* dims, int32_t rank, double* flops, double* bytes, int32_t* n_kernels) {
    return guard([&] {
        OAR_CHECK(e && dims && rank > 0, OAR_INVALID_INPUT, "oar_engine_cost: bad arguments");
        std::lock_guard<std::mutex> lk(e->e->mutex());
        const Plan& p = e->e->plan_for(std::vector<int64_t>(dims, dims + rank), rank >= 3);
        if (flops) *flops = p.flops;
        if (bytes) *bytes = p.bytes;
        if (n_kernels) *n_kernels = p.n_kernels;
    });
}

// ---------------------------------------------------------------------------------------------- detection
oar_status oar_det_create(const uint8_t* onnx, size_t onnx_len, const oar_det_cfg* cfg, oar_det** out) {
    return guard([&] {
        OAR_CHECK(out, OAR_INVALID_INPUT, "oar_det_create: out is null");
        *out = nullptr;
        oar_det_cfg c;
        std::memset(&c, 0, sizeof c);
        if (cfg) c = *cfg;
        std::unique_ptr<oar_det> h(new oar_det());
        h->d.reset(new Detector(onnx, onnx_len, c));
        if (c.profile) Profiler::get().enabled = true;
        *out = h.release();
    });
}
void oar_det_destroy(oar_det* d) { delete d; }

oar_status oar_det_run(oar_det* d, const uint8_t* const* rgb, const uint32_t* widths, const uint32_t* heights, uint32_t n_images,
                       float thresh, float box_thresh, float unclip_ratio, oar_det_result* out) {
    return guard([&] {
        OAR_CHECK(d && out && (n_images == 0 || (rgb && widths && heights)), OAR_INVALID_INPUT, "oar_det_run: bad arguments");
        std::vector<PageRef> pages(n_images);
        for (uint32_t i = 0; i < n_images; ++i) { pages[i].host = rgb[i]; pages[i].w = widths[i]; pages[i].h = heights[i]; }
        std::vector<DetBoxes> boxes;
        d->d->run(pages, thresh, box_thresh, unclip_ratio, boxes);
        fill_det_result(boxes, out);
    });
}
void oar_det_result_free(oar_det_result* r) {
    if (!r) return;
    std::free(r->box_offsets); std::free(r->points); std::free(r->scores); std::free(r->point_offsets);
    std::memset(r, 0, sizeof *r);
}

oar_status oar_db_postprocess(const float* pred, uint32_t height, uint32_t width, uint32_t src_w, uint32_t src_h, float thresh,
                              float box_thresh, float unclip_ratio, uint32_t max_candidates, oar_det_result* out) {
    return guard([&] {
        OAR_CHECK(pred && out && height && width, OAR_INVALID_INPUT, "oar_db_postprocess: bad arguments");
        std::vector<DetBoxes> boxes(1);
        Detector::postprocess_host(pred, (int)height, (int)width, src_w, src_h, thresh, box_thresh, unclip_ratio, max_candidates, boxes[0]);
        fill_det_result(boxes, out);
    });
}

oar_status oar_db_postprocess_ex(const float* pred, uint32_t height, uint32_t width, uint32_t src_w, uint32_t src_h, float thresh,
                                 float box_thresh, float unclip_ratio, uint32_t max_candidates, int32_t box_type, int32_t score_mode,
                                 int32_t use_dilation, oar_det_result* out) {
    return guard([&] {
        OAR_CHECK(pred && out && height && width, OAR_INVALID_INPUT, "oar_db_postprocess_ex: bad arguments");
        OAR_CHECK(box_type == 0 || box_type == 1, OAR_INVALID_INPUT, "box_type must be 0 (Quad) or 1 (Poly)");
        OAR_CHECK(score_mode == 0 || score_mode == 1, OAR_INVALID_INPUT, "score_mode must be 0 (fast) or 1 (slow)");
        std::vector<DetBoxes> boxes(1);
        Detector::postprocess_host(pred, (int)height, (int)width, src_w, src_h, thresh, box_thresh, unclip_ratio, max_candidates, boxes[0], score_mode, use_dilation, box_type);
        fill_det_result(boxes, out);
    });
}

// ---------------------------------------------------------------------------------------------- recognition
oar_status oar_rec_create(const uint8_t* onnx, size_t onnx_len, const oar_rec_cfg* cfg, oar_rec** out) {
    return guard([&] {
        OAR_CHECK(out, OAR_INVALID_INPUT, "oar_rec_create: out is null");
        *out = nullptr;
        oar_rec_cfg c;
        std::memset(&c, 0, sizeof c);
        if (cfg) c = *cfg;
        std::unique_ptr<oar_rec> h(new oar_rec());
        h->r.reset(new Recognizer(onnx, onnx_len, c));
        if (c.profile) Profiler::get().enabled = true;
        *out = h.release();
    });
}
void oar_rec_destroy(oar_rec* r) { delete r; }

oar_status oar_rec_run(oar_rec* r, const uint8_t* const* rgb, const uint32_t* widths, const uint32_t* heights, uint32_t n_crops,
                       oar_rec_result* out) {
    return guard([&] {
        OAR_CHECK(r && out && (n_crops == 0 || (rgb && widths && heights)), OAR_INVALID_INPUT, "oar_rec_run: bad arguments");
        std::memset(out, 0, sizeof *out);
        std::vector<Recognizer::Crop> crops(n_crops);
        for (uint32_t i = 0; i < n_crops; ++i) { crops[i].host = rgb[i]; crops[i].w = widths[i]; crops[i].h = heights[i]; }
        RecOut ro;
        r->r->run(crops, ro);
        out->batch = ro.idx.empty() ? 0 : n_crops;   // empty tensor => empty batch (decode.rs:465-472)
        out->seq_len = ro.T; out->vocab = ro.V; out->tensor_width = ro.Wt;
        out->indices = cmalloc<int64_t>(ro.idx.size());
        out->probs = cmalloc<float>(ro.prob.size());
        std::memcpy(out->indices, ro.idx.data(), ro.idx.size() * sizeof(int64_t));
        std::memcpy(out->probs, ro.prob.data(), ro.prob.size() * sizeof(float));
    });
}
void oar_rec_result_free(oar_rec_result* r) {
    if (!r) return;
    std::free(r->indices); std::free(r->probs);
    std::memset(r, 0, sizeof *r);
}

// ---------------------------------------------------------------------------------------------- pipeline
oar_status oar_ocr_create(const uint8_t* det_onnx, size_t det_len, const uint8_t* rec_onnx, size_t rec_len, const oar_ocr_cfg* cfg,
                          oar_ocr** out) {
    return guard([&] {
        OAR_CHECK(out, OAR_INVALID_INPUT, "oar_ocr_create: out is null");
        *out = nullptr;
        oar_ocr_cfg c;
        std::memset(&c, 0, sizeof c);
        if (cfg) c = *cfg;
        if (c.det_thresh == 0.f && c.det_box_thresh == 0.f && c.det_unclip_ratio == 0.f) {
            c.det_thresh = 0.3f; c.det_box_thresh = 0.6f; c.det_unclip_ratio = 2.0f;  // builder defaults, ocr.rs:319-366
        }
        OAR_CHECK(c.lanes <= 8, OAR_INVALID_INPUT, "oar_ocr_cfg.lanes must be in 0..=8");
        const uint32_t n_lanes = c.lanes ? c.lanes : 1;
        if (n_lanes > 1) {   // the lanes share the host: split the geometry pool (16 polling threads per lane would fight)
            int total = c.det.host_threads;
            if (total <= 0) { const char* e = getenv("OAR_HOST_THREADS"); total = e ? atoi(e) : 0; }
            if (total <= 0) total = std::min<int>(ThreadPool::available_cpus(), 16);
            c.det.host_threads = std::max(2, total / (int)n_lanes);
        }
        std::unique_ptr<oar_ocr> h(new oar_ocr());
        h->o.reset(new Ocr(det_onnx, det_len, rec_onnx, rec_len, c));
        for (uint32_t i = 0; i < n_lanes; ++i) {
            std::unique_ptr<OcrLane> l(new OcrLane());
            if (i > 0) l->o.reset(new Ocr(det_onnx, det_len, rec_onnx, rec_len, c));
            h->lanes.push_back(std::move(l));
        }
        oar_ocr* raw = h.get();
        for (uint32_t i = 0; i < n_lanes; ++i) {
            OcrLane* l = raw->lanes[i].get();
            l->worker = std::thread([raw, l, i] {
                for (;;) {
                    std::shared_ptr<OcrLane::Job> job;
                    {
                        std::unique_lock<std::mutex> lk(l->mu);
                        l->cv.wait(lk, [&] { return l->stop || !l->queue.empty(); });
                        if (l->queue.empty()) return;   // stop requested and nothing left to run
                        job = l->queue.front();
                        l->queue.pop_front();
                    }
                    try {
                        raw->lane_ocr(i).predict(job->pages, job->res, &job->meta);
                    } catch (const Error& e) { job->status = e.code; job->error = e.what(); }
                    catch (const std::bad_alloc&) { job->status = OAR_OOM; job->error = "host allocation failed"; }
                    catch (const std::exception& e) { job->status = OAR_INTERNAL; job->error = e.what(); }
                    catch (...) { job->status = OAR_INTERNAL; job->error = "unknown error"; }
                    { std::lock_guard<std::mutex> lk(raw->mu); job->done = true; }
                    raw->done_cv.notify_all();
                }
            });
        }
        if (c.det.profile || c.rec.profile) Profiler::get().enabled = true;
        *out = h.release();
    });
}
void oar_ocr_destroy(oar_ocr* o) { delete o; }

static void fill_ocr_result(const std::vector<std::vector<OcrRegion>>& res, const std::vector<Ocr::PageMeta>& meta, oar_ocr_result* out) {
    std::memset(out, 0, sizeof *out);
    size_t nreg = 0, nctc = 0;
    for (auto& im : res) for (auto& r : im) { ++nreg; nctc += r.idx.size(); }
    out->n_images = (uint32_t)res.size(); out->n_regions = (uint32_t)nreg;
    out->region_offsets = cmalloc<uint32_t>(res.size() + 1);
    bool poly = false;
    size_t npoly_floats = 0;
    for (auto& im : res) for (auto& r : im) { poly = poly || !r.poly.empty(); npoly_floats += r.poly.size(); }
    out->points = cmalloc<float>(poly ? npoly_floats : nreg * 8);
    out->n_points = (uint32_t)(poly ? npoly_floats / 2 : nreg * 4);
    if (poly) out->point_offsets = cmalloc<uint32_t>(nreg + 1);
    size_t pf = 0;
    out->det_scores = cmalloc<float>(nreg);
    out->crop_wh = cmalloc<uint32_t>(nreg * 2);
    out->seq_len = cmalloc<uint32_t>(nreg);
    out->max_wh_ratio = cmalloc<float>(nreg);
    out->ctc_offsets = cmalloc<uint64_t>(nreg + 1);
    out->ctc_indices = cmalloc<int64_t>(nctc);
    out->ctc_probs = cmalloc<float>(nctc);
    out->page_angle = cmalloc<float>(res.size());
    out->page_rectified = cmalloc<uint8_t>(res.size());
    out->line_angle = cmalloc<float>(nreg);
    for (size_t i = 0; i < res.size(); ++i) {
        out->page_angle[i] = i < meta.size() ? meta[i].angle : -1.0f;
        out->page_rectified[i] = i < meta.size() && meta[i].rectified ? 1 : 0;
    }
    size_t k = 0, c = 0;
    for (size_t i = 0; i < res.size(); ++i) {
        out->region_offsets[i] = (uint32_t)k;
        for (auto& r : res[i]) {
            if (poly) {
                out->point_offsets[k] = (uint32_t)(pf / 2);
                std::memcpy(out->points + pf, r.poly.data(), r.poly.size() * sizeof(float));
                pf += r.poly.size();
            } else {
                std::memcpy(out->points + k * 8, r.pts, sizeof r.pts);
            }
            out->det_scores[k] = r.det_score;
            out->crop_wh[k * 2] = r.crop_w; out->crop_wh[k * 2 + 1] = r.crop_h;
            out->seq_len[k] = r.T; out->max_wh_ratio[k] = r.max_wh_ratio; out->line_angle[k] = r.line_angle;
            out->ctc_offsets[k] = c;
            std::memcpy(out->ctc_indices + c, r.idx.data(), r.idx.size() * sizeof(int64_t));
            std::memcpy(out->ctc_probs + c, r.prob.data(), r.prob.size() * sizeof(float));
            c += r.idx.size();
            ++k;
        }
    }
    out->region_offsets[res.size()] = (uint32_t)k;
    if (poly) out->point_offsets[k] = (uint32_t)(pf / 2);
    out->ctc_offsets[k] = c;
}

static oar_status ocr_predict_impl(oar_ocr* o, const uint8_t* const* rgb, const uint32_t* widths, const uint32_t* heights, uint32_t n,
                                   bool device, oar_ocr_result* out) {
    return guard([&] {
        OAR_CHECK(o && out, OAR_INVALID_INPUT, "oar_ocr_predict: bad arguments");
        OAR_CHECK(n > 0 && rgb && widths && heights, OAR_INVALID_INPUT, "OCR Pipeline: images must be a non-empty slice");
        std::vector<PageRef> pages(n);
        for (uint32_t i = 0; i < n; ++i) {
            if (device) pages[i].dev = rgb[i]; else pages[i].host = rgb[i];
            pages[i].w = widths[i]; pages[i].h = heights[i];
        }
        std::vector<std::vector<OcrRegion>> res;
        std::vector<Ocr::PageMeta> meta;
        o->o->predict(pages, res, &meta);
        fill_ocr_result(res, meta, out);
    });
}
oar_status oar_ocr_predict_async(oar_ocr* o, const uint8_t* const* rgb, const uint32_t* widths, const uint32_t* heights, uint32_t n_images,
                                 int32_t device_pages, uint64_t* ticket) {
    return guard([&] {
        OAR_CHECK(o && ticket, OAR_INVALID_INPUT, "oar_ocr_predict_async: bad arguments");
        OAR_CHECK(n_images > 0 && rgb && widths && heights, OAR_INVALID_INPUT, "OCR Pipeline: images must be a non-empty slice");
        auto job = std::make_shared<OcrLane::Job>();
        job->pages.resize(n_images);
        for (uint32_t i = 0; i < n_images; ++i) {
            if (device_pages) job->pages[i].dev = rgb[i]; else job->pages[i].host = rgb[i];
            job->pages[i].w = widths[i]; job->pages[i].h = heights[i];
        }
        size_t lane;
        {
            std::lock_guard<std::mutex> lk(o->mu);
            job->ticket = o->next_ticket++;
            lane = (size_t)(job->ticket % o->lanes.size());   // round robin: consecutive calls land on different lanes
            o->jobs[job->ticket] = job;
        }
        OcrLane* l = o->lanes[lane].get();
        { std::lock_guard<std::mutex> lk(l->mu); l->queue.push_back(job); }
        l->cv.notify_one();
        *ticket = job->ticket;
    });
}

oar_status oar_ocr_wait(oar_ocr* o, uint64_t ticket, oar_ocr_result* out) {
    std::shared_ptr<OcrLane::Job> job;
    oar_status st = guard([&] {
        OAR_CHECK(o && out, OAR_INVALID_INPUT, "oar_ocr_wait: bad arguments");
        std::memset(out, 0, sizeof *out);
        std::unique_lock<std::mutex> lk(o->mu);
        auto it = o->jobs.find(ticket);
        OAR_CHECK(it != o->jobs.end(), OAR_INVALID_INPUT, "oar_ocr_wait: unknown (or already collected) ticket");
        job = it->second;
        o->done_cv.wait(lk, [&] { return job->done; });
        // (the wait released the lock: another thread waiting on the same ticket may have collected it -- `it` is not to be trusted)
        OAR_CHECK(o->jobs.erase(ticket) == 1, OAR_INVALID_INPUT, "oar_ocr_wait: ticket collected by another waiter");
    });
    if (st != OAR_OK) return st;
    if (job->status != OAR_OK) { oar::set_last_error(job->error); return job->status; }
    return guard([&] { fill_ocr_result(job->res, job->meta, out); });
}

oar_status oar_ocr_predict(oar_ocr* o, const uint8_t* const* rgb, const uint32_t* widths, const uint32_t* heights, uint32_t n_images,
                           oar_ocr_result* out) {
    return ocr_predict_impl(o, rgb, widths, heights, n_images, false, out);
}
oar_status oar_ocr_predict_device(oar_ocr* o, const uint8_t* const* d_rgb, const uint32_t* widths, const uint32_t* heights,
                                  uint32_t n_images, oar_ocr_result* out) {
    return ocr_predict_impl(o, d_rgb, widths, heights, n_images, true, out);
}
void oar_ocr_result_free(oar_ocr_result* r) {
    if (!r) return;
    std::free(r->region_offsets); std::free(r->points); std::free(r->det_scores); std::free(r->crop_wh); std::free(r->seq_len);
    std::free(r->max_wh_ratio); std::free(r->ctc_offsets); std::free(r->ctc_indices); std::free(r->ctc_probs);
    std::free(r->page_angle); std::free(r->page_rectified); std::free(r->line_angle); std::free(r->point_offsets);
    std::memset(r, 0, sizeof *r);
}

// ---------------------------------------------------------------------------------------------- config-5 stages
oar_status oar_cls_create(const uint8_t* onnx, size_t onnx_len, const oar_cls_cfg* cfg, oar_cls** out) {
    return guard([&] {
        OAR_CHECK(out, OAR_INVALID_INPUT, "oar_cls_create: out is null");
        *out = nullptr;
        ClsCfg c;
        if (cfg) {
            c.device_id = cfg->device_id;
            if (cfg->input_h && cfg->input_w) { c.input_h = cfg->input_h; c.input_w = cfg->input_w; }
            c.resize_short = cfg->resize_short;
            c.topk = cfg->topk ? cfg->topk : 1;
            c.batch = cfg->batch ? cfg->batch : 64;
        }
        std::unique_ptr<oar_cls> h(new oar_cls());
        h->c.reset(new Classifier(onnx, onnx_len, c));
        *out = h.release();
    });
}
void oar_cls_destroy(oar_cls* c) { delete c; }
static std::vector<Classifier::Image> cls_images(const uint8_t* const* rgb, const uint32_t* widths, const uint32_t* heights, uint32_t n) {
    std::vector<Classifier::Image> v(n);
    for (uint32_t i = 0; i < n; ++i) { v[i].host = rgb[i]; v[i].w = widths[i]; v[i].h = heights[i]; }
    return v;
}
oar_status oar_cls_run(oar_cls* c, const uint8_t* const* rgb, const uint32_t* widths, const uint32_t* heights, uint32_t n_images,
                       oar_cls_result* out) {
    return guard([&] {
        OAR_CHECK(c && out && (n_images == 0 || (rgb && widths && heights)), OAR_INVALID_INPUT, "oar_cls_run: bad arguments");
        std::memset(out, 0, sizeof *out);
        ClsOut co;
        c->c->run(cls_images(rgb, widths, heights, n_images), co);
        out->n_images = n_images; out->topk = co.topk; out->n_classes = co.n_classes;
        out->class_ids = cmalloc<int32_t>(co.ids.size());
        out->scores = cmalloc<float>(co.scores.size());
        std::memcpy(out->class_ids, co.ids.data(), co.ids.size() * sizeof(int32_t));
        std::memcpy(out->scores, co.scores.data(), co.scores.size() * sizeof(float));
    });
}
void oar_cls_result_free(oar_cls_result* r) {
    if (!r) return;
    std::free(r->class_ids); std::free(r->scores);
    std::memset(r, 0, sizeof *r);
}
oar_status oar_cls_preprocess(oar_cls* c, const uint8_t* const* rgb, const uint32_t* widths, const uint32_t* heights, uint32_t n_images,
                              float* out_nchw) {
    return guard([&] {
        OAR_CHECK(c && out_nchw && rgb && widths && heights, OAR_INVALID_INPUT, "oar_cls_preprocess: bad arguments");
        std::vector<float> v;
        c->c->pack_only(cls_images(rgb, widths, heights, n_images), v);
        std::memcpy(out_nchw, v.data(), v.size() * sizeof(float));
    });
}

// ---------------------------------------------------------------------------------------------- layout detection
static LayoutCfg layout_cfg_from(const oar_layout_cfg* cfg) {
    LayoutCfg c;
    if (!cfg) return c;
    c.device_id = cfg->device_id;
    if (cfg->input_h && cfg->input_w) { c.input_h = cfg->input_h; c.input_w = cfg->input_w; }
    c.filter = cfg->resize_filter;
    c.bgr = cfg->color_bgr != 0;
    if (cfg->scale > 0.0f) c.scale = cfg->scale;
    if (cfg->std[0] != 0.0f || cfg->std[1] != 0.0f || cfg->std[2] != 0.0f)
        for (int k = 0; k < 3; ++k) { c.mean[k] = cfg->mean[k]; c.stdv[k] = cfg->std[k]; }
    if (cfg->num_classes) c.num_classes = cfg->num_classes;
    c.model_type = cfg->model_type;
    c.score_threshold = cfg->score_threshold;
    c.nms_threshold = cfg->nms_threshold;
    if (cfg->max_detections) c.max_detections = cfg->max_detections;
    return c;
}
static void fill_layout_result(const LayoutOut& lo, oar_layout_result* out) {
    std::memset(out, 0, sizeof *out);
    out->n_images = (uint32_t)(lo.offsets.size() - 1); out->n_boxes = (uint32_t)lo.scores.size(); out->feature_dim = lo.feature_dim;
    out->box_offsets = cmalloc<uint32_t>(lo.offsets.size());
    std::memcpy(out->box_offsets, lo.offsets.data(), lo.offsets.size() * 4);
    out->boxes = cmalloc<float>(lo.boxes.size()); out->classes = cmalloc<int32_t>(lo.classes.size()); out->scores = cmalloc<float>(lo.scores.size());
    std::memcpy(out->boxes, lo.boxes.data(), lo.boxes.size() * 4);
    std::memcpy(out->classes, lo.classes.data(), lo.classes.size() * 4);
    std::memcpy(out->scores, lo.scores.data(), lo.scores.size() * 4);
}
oar_status oar_layout_create(const uint8_t* onnx, size_t onnx_len, const oar_layout_cfg* cfg, oar_layout** out) {
    return guard([&] {
        OAR_CHECK(out, OAR_INVALID_INPUT, "oar_layout_create: out is null");
        *out = nullptr;
        std::unique_ptr<oar_layout> h(new oar_layout());
        h->l.reset(new LayoutDetector(onnx, onnx_len, layout_cfg_from(cfg)));
        *out = h.release();
    });
}
void oar_layout_destroy(oar_layout* l) { delete l; }
oar_status oar_layout_run(oar_layout* l, const uint8_t* const* rgb, const uint32_t* widths, const uint32_t* heights, uint32_t n_images, oar_layout_result* out) {
    return guard([&] {
        OAR_CHECK(l && out && (n_images == 0 || (rgb && widths && heights)), OAR_INVALID_INPUT, "oar_layout_run: bad arguments");
        std::vector<LayoutDetector::Image> imgs(n_images);
        for (uint32_t i = 0; i < n_images; ++i) { imgs[i].host = rgb[i]; imgs[i].w = widths[i]; imgs[i].h = heights[i]; }
        LayoutOut lo;
        l->l->run(imgs, lo);
        fill_layout_result(lo, out);
    });
}
oar_status oar_layout_run_ppdoc(oar_layout* l, const uint8_t* const* rgb, const uint32_t* widths, const uint32_t* heights, uint32_t n_images, const oar_ppdoc_cfg* cfg,
                                oar_layout_result* out) {
    return guard([&] {
        OAR_CHECK(l && out && cfg && (n_images == 0 || (rgb && widths && heights)), OAR_INVALID_INPUT, "oar_layout_run_ppdoc: bad arguments");
        std::vector<LayoutDetector::Image> imgs(n_images);
        for (uint32_t i = 0; i < n_images; ++i) { imgs[i].host = rgb[i]; imgs[i].w = widths[i]; imgs[i].h = heights[i]; }
        LayoutDetector::PpDocCfg pc;
        pc.score_threshold = cfg->score_threshold; pc.class_thr = cfg->class_thresholds; pc.layout_nms = cfg->layout_nms != 0;
        pc.image_class = cfg->image_class_id; pc.formula_class = cfg->formula_class_id; pc.merge_mode = cfg->class_merge_modes;
        LayoutOut lo;
        l->l->run_ppdoc(imgs, pc, lo);
        fill_layout_result(lo, out);
    });
}
void oar_layout_result_free(oar_layout_result* r) {
    if (!r) return;
    std::free(r->box_offsets); std::free(r->boxes); std::free(r->classes); std::free(r->scores);
    std::memset(r, 0, sizeof *r);
}
oar_status oar_layout_preprocess(oar_layout* l, const uint8_t* rgb, uint32_t width, uint32_t height, float* out_chw) {
    return guard([&] {
        OAR_CHECK(l && rgb && out_chw, OAR_INVALID_INPUT, "oar_layout_preprocess: bad arguments");
        LayoutDetector::Image im;
        im.host = rgb; im.w = width; im.h = height;
        std::vector<float> t;
        l->l->preprocess_only(im, t);
        std::memcpy(out_chw, t.data(), t.size() * 4);
    });
}
oar_status oar_k_resize_filter(const uint8_t* rgb, uint32_t w, uint32_t h, uint32_t nw, uint32_t nh, int32_t filter, uint8_t* out) {
    return guard([&] {
        OAR_CHECK(rgb && out && w && h && nw && nh && filter >= 0 && filter <= 2, OAR_INVALID_INPUT, "oar_k_resize_filter: bad arguments");
        require_device();
        if (nw == w && nh == h) { std::memcpy(out, rgb, (size_t)w * h * 3); return; }
        std::vector<pp::FilterTaps> tv, th;
        std::vector<float> wv, wh;
        const int mv = host::filter_taps(filter, (int)h, (int)nh, tv, wv), mh = host::filter_taps(filter, (int)w, (int)nw, th, wh);
        DevBuf din, dout, dtmp, dtv, dwv, dth, dwh;
        din.reserve((size_t)w * h * 3); dout.reserve((size_t)nw * nh * 3); dtmp.reserve((size_t)nh * w * 12);
        dtv.reserve(tv.size() * sizeof(pp::FilterTaps)); dwv.reserve(wv.size() * 4); dth.reserve(th.size() * sizeof(pp::FilterTaps)); dwh.reserve(wh.size() * 4);
        OAR_HIP(hipMemcpy(din.p, rgb, (size_t)w * h * 3, hipMemcpyHostToDevice));
        OAR_HIP(hipMemcpy(dtv.p, tv.data(), tv.size() * sizeof(pp::FilterTaps), hipMemcpyHostToDevice));
        OAR_HIP(hipMemcpy(dwv.p, wv.data(), wv.size() * 4, hipMemcpyHostToDevice));
        OAR_HIP(hipMemcpy(dth.p, th.data(), th.size() * sizeof(pp::FilterTaps), hipMemcpyHostToDevice));
        OAR_HIP(hipMemcpy(dwh.p, wh.data(), wh.size() * 4, hipMemcpyHostToDevice));
        pp::resize_filter(nullptr, din.as<uint8_t>(), (int)w, (int)h, dout.as<uint8_t>(), (int)nw, (int)nh, dtv.as<pp::FilterTaps>(), dwv.as<float>(), mv, dth.as<pp::FilterTaps>(),
                          dwh.as<float>(), mh, dtmp.as<float>());
        OAR_HIP(hipDeviceSynchronize());
        OAR_HIP(hipMemcpy(out, dout.p, (size_t)nw * nh * 3, hipMemcpyDeviceToHost));
    });
}
oar_status oar_k_layout_postprocess(const float* pred, uint32_t n_images, uint32_t rows, uint32_t feat, const float* src_wh, uint32_t num_classes, int32_t model_type,
                                    float score_threshold, float nms_threshold, uint32_t max_detections, oar_layout_result* out) {
    return guard([&] {
        OAR_CHECK(out && (n_images == 0 || (src_wh && (rows == 0 || feat == 0 || pred))) && max_detections > 0 && max_detections <= 4096 && rows <= 16384, OAR_INVALID_INPUT,
                  "oar_k_layout_postprocess: bad arguments");
        // the row formats the kernel parses: 6 / 7 / 8 compact columns, or 4 box columns + one score per class; any other width yields no detections, as
        // the reference's row parser does (layout_postprocess.rs:277-340) -- every read stays inside the row, so only the size is bounded here
        OAR_CHECK(feat <= 4 + 4096 && num_classes <= 4096, OAR_INVALID_INPUT, "oar_k_layout_postprocess: feat / num_classes out of range");
        require_device();
        LayoutOut lo;
        lo.offsets.assign(1, 0);
        lo.feature_dim = feat;
        if (n_images == 0 || rows == 0 || feat == 0) { for (uint32_t i = 0; i < n_images; ++i) lo.offsets.push_back(0); fill_layout_result(lo, out); return; }
        DevBuf dpred, dwh, dcand, dsorted, dkeep;
        const size_t np = (size_t)n_images * rows * feat;
        dpred.reserve(np * 4); dwh.reserve((size_t)n_images * 8); dcand.reserve((size_t)n_images * rows * 32); dsorted.reserve((size_t)n_images * rows * 4);
        dkeep.reserve((size_t)n_images * (max_detections + 1) * 4);
        OAR_HIP(hipMemcpy(dpred.p, pred, np * 4, hipMemcpyHostToDevice));
        OAR_HIP(hipMemcpy(dwh.p, src_wh, (size_t)n_images * 8, hipMemcpyHostToDevice));
        pp::LayoutPostP p{};
        p.pred = dpred.as<float>(); p.rows = (int)rows; p.feat = (int)feat; p.num_classes = (int)num_classes; p.model_type = model_type; p.max_det = (int)max_detections;
        p.score_thr = score_threshold; p.nms_thr = nms_threshold; p.src_wh = dwh.as<float>();
        p.cand = dcand.as<float>(); p.sorted = dsorted.as<int>(); p.keep = dkeep.as<int>(); p.n_keep = dkeep.as<int>() + (size_t)n_images * max_detections;
        pp::layout_postprocess(nullptr, p, (int)n_images);
        OAR_HIP(hipDeviceSynchronize());
        std::vector<int> keep((size_t)n_images * (max_detections + 1));
        std::vector<float> cand((size_t)n_images * rows * 8);
        OAR_HIP(hipMemcpy(keep.data(), dkeep.p, keep.size() * 4, hipMemcpyDeviceToHost));
        OAR_HIP(hipMemcpy(cand.data(), dcand.p, cand.size() * 4, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n_images; ++i) {
            const int nk = keep[(size_t)n_images * max_detections + i];
            std::vector<int> rk(keep.begin() + (long)((size_t)i * max_detections), keep.begin() + (long)((size_t)i * max_detections) + nk);
            if (model_type == 2 && feat == 8 && nk > 1) {
                auto key = [](float v) { int32_t b; std::memcpy(&b, &v, 4); return b ^ (int32_t)(((uint32_t)(b >> 31)) >> 1); };
                const float* pr = pred + (size_t)i * rows * feat;
                std::stable_sort(rk.begin(), rk.end(), [&](int a, int b) {
                    const int32_t ca = key(pr[(size_t)a * feat + 6]), cb = key(pr[(size_t)b * feat + 6]);
                    if (ca != cb) return ca < cb;
                    return key(pr[(size_t)a * feat + 7]) < key(pr[(size_t)b * feat + 7]);
                });
            }
            for (int r : rk) {
                const float* c8 = cand.data() + ((size_t)i * rows + (size_t)r) * 8;
                lo.boxes.insert(lo.boxes.end(), c8, c8 + 4);
                lo.scores.push_back(c8[4]);
                int32_t cls; std::memcpy(&cls, &c8[5], 4);
                lo.classes.push_back(cls);
            }
            lo.offsets.push_back((uint32_t)lo.scores.size());
        }
        fill_layout_result(lo, out);
    });
}

oar_status oar_rect_create(const uint8_t* onnx, size_t onnx_len, const oar_rect_cfg* cfg, oar_rect** out) {
    return guard([&] {
        OAR_CHECK(out, OAR_INVALID_INPUT, "oar_rect_create: out is null");
        *out = nullptr;
        RectCfg c;
        if (cfg) { c.device_id = cfg->device_id; if (cfg->target_h == OAR_RECT_NATIVE_SIZE || cfg->target_w == OAR_RECT_NATIVE_SIZE) { c.target_h = 0; c.target_w = 0; }   // Rectifier: 0 = feed pages at their own size (uvdoc.rs:86)
                   else if (cfg->target_h && cfg->target_w) { c.target_h = cfg->target_h; c.target_w = cfg->target_w; } }
        std::unique_ptr<oar_rect> h(new oar_rect());
        h->r.reset(new Rectifier(onnx, onnx_len, c));
        *out = h.release();
    });
}
void oar_rect_destroy(oar_rect* r) { delete r; }
oar_status oar_rect_run(oar_rect* r, const uint8_t* rgb, uint32_t width, uint32_t height, uint8_t* out_rgb) {
    return guard([&] {
        OAR_CHECK(r && rgb && out_rgb && width > 0 && height > 0, OAR_INVALID_INPUT, "oar_rect_run: bad arguments");
        r->r->run_host(rgb, width, height, out_rgb);
    });
}

oar_status oar_ocr_attach(oar_ocr* o, oar_cls* doc_orientation, oar_rect* rectifier, oar_cls* line_orientation) {
    return guard([&] {
        OAR_CHECK(o, OAR_INVALID_INPUT, "oar_ocr_attach: pipeline handle is null");
        // the stage handles lock internally (Classifier / Rectifier own a mutex): every lane may borrow the same ones
        for (size_t i = 0; i < std::max<size_t>(o->lanes.size(), 1); ++i)
            o->lane_ocr(i).attach(doc_orientation ? doc_orientation->c.get() : nullptr, rectifier ? rectifier->r.get() : nullptr,
                                  line_orientation ? line_orientation->c.get() : nullptr);
    });
}

oar_status oar_k_rotate_rgb(const uint8_t* rgb, uint32_t w, uint32_t h, int32_t quarter, uint8_t* out) {
    return guard([&] {
        OAR_CHECK(rgb && out && quarter >= 0 && quarter <= 3, OAR_INVALID_INPUT, "oar_k_rotate_rgb: bad arguments");
        require_device();
        const size_t bytes = (size_t)w * h * 3;
        if (bytes == 0) return;
        DevBuf a, b;
        a.reserve(bytes); b.reserve(bytes);
        OAR_HIP(hipMemcpy(a.p, rgb, bytes, hipMemcpyHostToDevice));
        pp::rotate_rgb(nullptr, a.as<uint8_t>(), (int)w, (int)h, quarter, b.as<uint8_t>());
        OAR_HIP(hipDeviceSynchronize());
        OAR_HIP(hipMemcpy(out, b.p, bytes, hipMemcpyDeviceToHost));
    });
}
oar_status oar_k_bgr_planes_to_rgb(const float* planes, uint64_t plane, float scale, uint8_t* out) {
    return guard([&] {
        OAR_CHECK(planes && out, OAR_INVALID_INPUT, "oar_k_bgr_planes_to_rgb: bad arguments");
        require_device();
        if (plane == 0) return;
        DevBuf a, b;
        a.reserve(plane * 12); b.reserve(plane * 3);
        OAR_HIP(hipMemcpy(a.p, planes, plane * 12, hipMemcpyHostToDevice));
        pp::bgr_planes_to_rgb(nullptr, a.as<float>(), (int64_t)plane, scale, b.as<uint8_t>());
        OAR_HIP(hipDeviceSynchronize());
        OAR_HIP(hipMemcpy(out, b.p, plane * 3, hipMemcpyDeviceToHost));
    });
}
oar_status oar_host_rotate_back_points(float* pts, uint32_t n_points, float angle, uint32_t rotated_w, uint32_t rotated_h) {
    return guard([&] {
        OAR_CHECK(pts || n_points == 0, OAR_INVALID_INPUT, "oar_host_rotate_back_points: pts is null");
        const int a = (int)angle;
        for (uint32_t i = 0; i < n_points; ++i) {
            const float x = pts[2 * i], y = pts[2 * i + 1];
            if (a == 90) { pts[2 * i] = (float)rotated_h - y; pts[2 * i + 1] = x; }
            else if (a == 180) { pts[2 * i] = (float)rotated_w - x; pts[2 * i + 1] = (float)rotated_h - y; }
            else if (a == 270) { pts[2 * i] = y; pts[2 * i + 1] = (float)rotated_w - x; }
        }
    });
}

// ---------------------------------------------------------------------------------------------- device helpers
oar_status oar_dev_alloc(int32_t device_id, size_t bytes, void** out) {
    return guard([&] {
        OAR_CHECK(out, OAR_INVALID_INPUT, "oar_dev_alloc: out is null");
        require_device();
        OAR_HIP(hipSetDevice(device_id));
        OAR_HIP(hipMalloc(out, std::max<size_t>(bytes, 16)));
    });
}
oar_status oar_dev_upload(void* dst, const void* src, size_t bytes) {
    return guard([&] { OAR_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice)); });
}
oar_status oar_dev_download(void* dst, const void* src, size_t bytes) {
    return guard([&] { OAR_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost)); });
}
void oar_dev_free(void* p) {
    if (p) (void)hipFree(p);
}
oar_status oar_dev_synchronize(int32_t device_id) {
    return guard([&] {
        require_device();
        OAR_HIP(hipSetDevice(device_id));
        OAR_HIP(hipDeviceSynchronize());
    });
}

// ---------------------------------------------------------------------------------------------- stand-alone kernels
oar_status oar_k_normalize(const uint8_t* rgb, uint32_t w, uint32_t h, const int32_t src_channels[3], const float alpha[3],
                           const float beta[3], int32_t hwc_layout, float* out) {
    return guard([&] {
        OAR_CHECK(rgb && out && src_channels && alpha && beta, OAR_INVALID_INPUT, "oar_k_normalize: bad arguments");
        require_device();
        size_t plane = (size_t)w * h;
        DevBuf din, dout;
        din.reserve(plane * 3); dout.reserve(plane * 12);
        OAR_HIP(hipMemcpy(din.p, rgb, plane * 3, hipMemcpyHostToDevice));
        int src[3] = {src_channels[0], src_channels[1], src_channels[2]};
        pp::normalize(nullptr, din.as<uint8_t>(), dout.as<float>(), 1, (int64_t)plane, src, alpha, beta, hwc_layout ? 1 : 0);
        OAR_HIP(hipMemcpy(out, dout.p, plane * 12, hipMemcpyDeviceToHost));
    });
}

oar_status oar_k_rec_preprocess(const uint8_t* const* rgb, const uint32_t* widths, const uint32_t* heights, uint32_t n, uint32_t img_h,
                                uint32_t img_w, uint32_t max_img_w, float* out_nchw, uint32_t* tensor_width) {
    return oar_k_rec_preprocess_flip(rgb, widths, heights, nullptr, n, img_h, img_w, max_img_w, out_nchw, tensor_width);
}

oar_status oar_k_rec_preprocess_flip(const uint8_t* const* rgb, const uint32_t* widths, const uint32_t* heights, const uint8_t* flips, uint32_t n,
                                     uint32_t img_h, uint32_t img_w, uint32_t max_img_w, float* out_nchw, uint32_t* tensor_width) {
    return guard([&] {
        OAR_CHECK(tensor_width && (n == 0 || (rgb && widths && heights)), OAR_INVALID_INPUT, "oar_k_rec_preprocess: bad arguments");
        require_device();
        // size query when out_nchw is null
        std::vector<uint32_t> ws(widths, widths + n), hs(heights, heights + n);
        std::vector<int32_t> rws;
        int Wt = host::rec_tensor_width(ws, hs, (int)img_h, (int)img_w, (int)max_img_w, rws);
        *tensor_width = (uint32_t)Wt;
        if (!out_nchw || n == 0) return;
        size_t stage = 0;
        for (uint32_t i = 0; i < n; ++i) stage += ((size_t)ws[i] * hs[i] * 3 + 63) & ~(size_t)63;
        DevBuf dc, dd, dout;
        dc.reserve(stage); dd.reserve(n * sizeof(pp::CropDesc)); dout.reserve((size_t)n * 3 * img_h * Wt * 4);
        std::vector<pp::CropDesc> descs(n);
        size_t off = 0;
        for (uint32_t i = 0; i < n; ++i) {
            size_t bytes = (size_t)ws[i] * hs[i] * 3;
            OAR_HIP(hipMemcpy(dc.as<uint8_t>() + off, rgb[i], bytes, hipMemcpyHostToDevice));
            descs[i].src = dc.as<uint8_t>() + off; descs[i].w = (int)ws[i]; descs[i].h = (int)hs[i]; descs[i].rw = rws[i]; descs[i].flip = flips && flips[i] ? 1 : 0;
            off += (bytes + 63) & ~(size_t)63;
        }
        OAR_HIP(hipMemcpy(dd.p, descs.data(), n * sizeof(pp::CropDesc), hipMemcpyHostToDevice));
        pp::rec_pack(nullptr, dd.as<pp::CropDesc>(), (int)n, (int)img_h, Wt, dout.as<float>(), 1);
        OAR_HIP(hipMemcpy(out_nchw, dout.p, (size_t)n * 3 * img_h * Wt * 4, hipMemcpyDeviceToHost));
    });
}

oar_status oar_k_resize_triangle(const uint8_t* rgb, uint32_t w, uint32_t h, uint32_t nw, uint32_t nh, uint8_t* out) {
    return guard([&] {
        OAR_CHECK(rgb && out && w && h && nw && nh, OAR_INVALID_INPUT, "oar_k_resize_triangle: bad arguments");
        require_device();
        DevBuf din, dout;
        din.reserve((size_t)w * h * 3); dout.reserve((size_t)nw * nh * 3);
        OAR_HIP(hipMemcpy(din.p, rgb, (size_t)w * h * 3, hipMemcpyHostToDevice));
        pp::resize_triangle(nullptr, din.as<uint8_t>(), (int)w, (int)h, dout.as<uint8_t>(), (int)nw, (int)nh);
        OAR_HIP(hipMemcpy(out, dout.p, (size_t)nw * nh * 3, hipMemcpyDeviceToHost));
    });
}

oar_status oar_k_threshold(const float* pred, size_t n, float thresh, uint8_t* mask) {
    return guard([&] {
        OAR_CHECK((pred && mask) || n == 0, OAR_INVALID_INPUT, "oar_k_threshold: bad arguments");
        require_device();
        if (n == 0) return;
        DevBuf din, dout;
        din.reserve(n * 4); dout.reserve(n + 4);
        OAR_HIP(hipMemcpy(din.p, pred, n * 4, hipMemcpyHostToDevice));
        pp::threshold(nullptr, din.as<float>(), dout.as<uint8_t>(), (int64_t)n, thresh);
        OAR_HIP(hipMemcpy(mask, dout.p, n, hipMemcpyDeviceToHost));
    });
}

oar_status oar_k_dilate(const uint8_t* mask, uint32_t height, uint32_t width, uint8_t* out) {
    return guard([&] {
        OAR_CHECK(mask && out && height && width, OAR_INVALID_INPUT, "oar_k_dilate: bad arguments");
        require_device();
        const size_t n = (size_t)height * width;
        DevBuf din, dout;
        din.reserve(n); dout.reserve(n);
        OAR_HIP(hipMemcpy(din.p, mask, n, hipMemcpyHostToDevice));
        pp::dilate3x3(nullptr, din.as<uint8_t>(), dout.as<uint8_t>(), 1, (int)height, (int)width);
        OAR_HIP(hipMemcpy(out, dout.p, n, hipMemcpyDeviceToHost));
    });
}

oar_status oar_k_poly_scores(const float* pred, uint32_t height, uint32_t width, const float* pts_xy, const uint32_t* counts, uint32_t n_polys,
                             float* scores) {
    return guard([&] {
        OAR_CHECK(pred && height && width && (n_polys == 0 || (pts_xy && counts && scores)), OAR_INVALID_INPUT, "oar_k_poly_scores: bad arguments");
        require_device();
        if (n_polys == 0) return;
        std::vector<pp::PolyDesc> pd(n_polys);
        size_t at = 0;
        for (uint32_t i = 0; i < n_polys; ++i) { pd[i] = pp::PolyDesc{(int32_t)at, (int32_t)counts[i], 0, 0}; at += counts[i]; }
        const size_t hw = (size_t)height * width;
        DevBuf dpred, dpts, dpd, dsc;
        dpred.reserve(hw * 4); dpts.reserve(at * 8 + 8); dpd.reserve(pd.size() * sizeof(pp::PolyDesc)); dsc.reserve((size_t)n_polys * 4);
        OAR_HIP(hipMemcpy(dpred.p, pred, hw * 4, hipMemcpyHostToDevice));
        if (at) OAR_HIP(hipMemcpy(dpts.p, pts_xy, at * 8, hipMemcpyHostToDevice));
        OAR_HIP(hipMemcpy(dpd.p, pd.data(), pd.size() * sizeof(pp::PolyDesc), hipMemcpyHostToDevice));
        pp::poly_scores(nullptr, dpred.as<float>(), (int)height, (int)width, dpts.as<float>(), dpd.as<pp::PolyDesc>(), (int)n_polys, dsc.as<float>());
        OAR_HIP(hipMemcpy(scores, dsc.p, (size_t)n_polys * 4, hipMemcpyDeviceToHost));
    });
}

oar_status oar_k_ctc_argmax(const float* probs, size_t rows, size_t vocab, int64_t* idx, float* prob) {
    return guard([&] {
        require_device();
        if (rows == 0 || vocab == 0) return;  // empty tensor => no entries (decode.rs:465-472)
        OAR_CHECK(probs && idx && prob, OAR_INVALID_INPUT, "oar_k_ctc_argmax: bad arguments");
        DevBuf din, di, dp;
        din.reserve(rows * vocab * 4); di.reserve(rows * 8); dp.reserve(rows * 4);
        OAR_HIP(hipMemcpy(din.p, probs, rows * vocab * 4, hipMemcpyHostToDevice));
        pp::ctc_argmax(nullptr, din.as<float>(), (int64_t)rows, (int)vocab, di.as<int64_t>(), dp.as<float>());
        OAR_HIP(hipMemcpy(idx, di.p, rows * 8, hipMemcpyDeviceToHost));
        OAR_HIP(hipMemcpy(prob, dp.p, rows * 4, hipMemcpyDeviceToHost));
    });
}

oar_status oar_k_unclip(const float* boxes, uint32_t n_boxes, float ratio, int32_t* counts, float* pts_xy, uint32_t cap_points) {
    return guard([&] {
        require_device();
        OAR_CHECK(n_boxes == 0 || (boxes && counts && pts_xy), OAR_INVALID_INPUT, "oar_k_unclip: bad arguments");
        if (n_boxes == 0) return;
        std::vector<pp::ScoreBox> sb(n_boxes);
        for (uint32_t i = 0; i < n_boxes; ++i) { std::memcpy(sb[i].pts, boxes + (size_t)i * 8, 32); sb[i].image = 0; sb[i].pad = 0; }
        DevBuf db, dout;
        db.reserve(n_boxes * sizeof(pp::ScoreBox)); dout.reserve(n_boxes * sizeof(pp::UnclipOut));
        OAR_HIP(hipMemcpy(db.p, sb.data(), n_boxes * sizeof(pp::ScoreBox), hipMemcpyHostToDevice));
        pp::unclip_quads(nullptr, db.as<pp::ScoreBox>(), (int)n_boxes, ratio, dout.as<pp::UnclipOut>());
        std::vector<pp::UnclipOut> ho(n_boxes);
        OAR_HIP(hipMemcpy(ho.data(), dout.p, n_boxes * sizeof(pp::UnclipOut), hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n_boxes; ++i) {
            counts[i] = ho[i].n_pts;
            const int m = std::min<int>(std::max(ho[i].n_pts, 0), (int)cap_points);
            std::memcpy(pts_xy + (size_t)i * cap_points * 2, ho[i].pts, (size_t)m * 2 * sizeof(float));
        }
    });
}
oar_status oar_k_box_scores(const float* pred, uint32_t height, uint32_t width, const float* boxes, uint32_t n_boxes, float* scores) {
    return guard([&] {
        require_device();
        if (n_boxes == 0) return;
        OAR_CHECK(pred && boxes && scores && height && width, OAR_INVALID_INPUT, "oar_k_box_scores: bad arguments");
        size_t hw = (size_t)height * width;
        DevBuf dpred, db, ds;
        dpred.reserve(hw * 4); db.reserve(n_boxes * sizeof(pp::ScoreBox)); ds.reserve(n_boxes * 4);
        std::vector<pp::ScoreBox> sb(n_boxes);
        for (uint32_t i = 0; i < n_boxes; ++i) { std::memcpy(sb[i].pts, boxes + (size_t)i * 8, 32); sb[i].image = 0; sb[i].pad = 0; }
        OAR_HIP(hipMemcpy(dpred.p, pred, hw * 4, hipMemcpyHostToDevice));
        OAR_HIP(hipMemcpy(db.p, sb.data(), n_boxes * sizeof(pp::ScoreBox), hipMemcpyHostToDevice));
        pp::box_scores(nullptr, dpred.as<float>(), (int)height, (int)width, db.as<pp::ScoreBox>(), (int)n_boxes, ds.as<float>());
        OAR_HIP(hipMemcpy(scores, ds.p, n_boxes * 4, hipMemcpyDeviceToHost));
    });
}

oar_status oar_k_rotate_crop(const uint8_t* rgb, uint32_t w, uint32_t h, const float box[8], uint8_t* out, size_t cap, uint32_t* out_w,
                             uint32_t* out_h) {
    return guard([&] {
        OAR_CHECK(rgb && box && out_w && out_h && w && h, OAR_INVALID_INPUT, "oar_k_rotate_crop: bad arguments");
        require_device();
        host::CropPlan pl = host::plan_crop((int)w, (int)h, box);
        *out_w = *out_h = 0;
        if (pl.mode == 0) return;
        *out_w = (uint32_t)pl.out_w(); *out_h = (uint32_t)pl.out_h();
        size_t bytes = (size_t)pl.out_w() * pl.out_h() * 3;
        OAR_CHECK(out && cap >= bytes, OAR_INVALID_INPUT, "oar_k_rotate_crop: output buffer too small");
        DevBuf dpage, dd, dout;
        dpage.reserve((size_t)w * h * 3); dd.reserve(sizeof(pp::WarpDesc)); dout.reserve(bytes);
        OAR_HIP(hipMemcpy(dpage.p, rgb, (size_t)w * h * 3, hipMemcpyHostToDevice));
        pp::WarpDesc d;
        std::memset(&d, 0, sizeof d);
        d.page = dpage.as<uint8_t>(); d.page_w = (int)w; d.page_h = (int)h;
        d.left = pl.left; d.top = pl.top; d.cw = pl.cw; d.ch = pl.ch; d.ow = pl.ow; d.oh = pl.oh; d.rot = pl.rot; d.mode = pl.mode;
        std::memcpy(d.inv, pl.inv, sizeof d.inv);
        d.out_off = 0;
        OAR_HIP(hipMemcpy(dd.p, &d, sizeof d, hipMemcpyHostToDevice));
        pp::rotate_crops(nullptr, dd.as<pp::WarpDesc>(), 1, dout.as<uint8_t>(), pl.out_w() * pl.out_h());
        OAR_HIP(hipMemcpy(out, dout.p, bytes, hipMemcpyDeviceToHost));
    });
}

// ---------------------------------------------------------------------------------------------- host hooks (no GPU)
int32_t oar_host_candidates(const uint8_t* mask, uint32_t width, uint32_t height, uint32_t max_candidates, int32_t max_bands,
                            float* boxes8, int32_t cap) {
    try {
        // the detector's own route: the mask as a bit plane, row bands cut at blank rows, corner points only (pipeline.cc subbatch_candidates)
        const int row_bytes = ((int)width + 7) / 8;
        std::vector<uint8_t> bits((size_t)row_bytes * height, 0);
        for (uint32_t y = 0; y < height; ++y)
            for (uint32_t x = 0; x < width; ++x)
                if (mask[(size_t)y * width + x]) bits[(size_t)y * row_bytes + (x >> 3)] |= (uint8_t)(1u << (x & 7));
        std::vector<int> cuts = host::blank_row_bands_bits(bits.data(), row_bytes, (int)height, max_bands < 1 ? 1 : max_bands);
        std::vector<host::Contour> cs;
        for (size_t i = 0; i + 1 < cuts.size(); ++i) {
            auto part = host::find_contours_band_bits(bits.data(), row_bytes, (int)width, cuts[i], cuts[i + 1], max_candidates, true);
            for (auto& c : part) { if (cs.size() >= max_candidates) break; cs.push_back(std::move(c)); }
        }
        int32_t n = 0;
        for (auto& c : cs) {
            host::Pt mb[4];
            float ms = 0.f;
            if (!host::contour_mini_box(c, mb, ms) || ms < 3.0f) continue;
            if (n < cap) for (int k = 0; k < 4; ++k) { boxes8[n * 8 + k * 2] = mb[k].x; boxes8[n * 8 + k * 2 + 1] = mb[k].y; }
            ++n;
        }
        return n;
    } catch (...) { return -1; }
}
int32_t oar_host_contours_bits(const uint8_t* mask, uint32_t width, uint32_t height, uint32_t max_contours, int32_t max_bands, int64_t* offsets,
                               int32_t* pts_xy, int32_t* types, int64_t cap_points) {
    try {
        // the detector's read-back format: the byte mask packed to a bit plane (here on the host), followed band by band
        const int row_bytes = ((int)width + 7) / 8;
        std::vector<uint8_t> bits((size_t)row_bytes * height, 0);
        for (uint32_t y = 0; y < height; ++y)
            for (uint32_t x = 0; x < width; ++x)
                if (mask[(size_t)y * width + x]) bits[(size_t)y * row_bytes + (x >> 3)] |= (uint8_t)(1u << (x & 7));
        std::vector<int> cuts = host::blank_row_bands_bits(bits.data(), row_bytes, (int)height, max_bands < 1 ? 1 : max_bands);
        int32_t n = 0;
        int64_t np = 0;
        if (offsets) offsets[0] = 0;
        for (size_t i = 0; i + 1 < cuts.size(); ++i) {
            auto part = host::find_contours_band_bits(bits.data(), row_bytes, (int)width, cuts[i], cuts[i + 1], max_contours);
            for (auto& c : part) {
                if ((uint32_t)n >= max_contours) break;
                for (auto& q : c.pts) {
                    if (pts_xy && np < cap_points) { pts_xy[np * 2] = (int32_t)q.x; pts_xy[np * 2 + 1] = (int32_t)q.y; }
                    ++np;
                }
                if (types) types[n] = c.hole ? 1 : 0;
                ++n;
                if (offsets) offsets[n] = np;
            }
        }
        return n;
    } catch (...) { return -1; }
}
int32_t oar_host_contours(const uint8_t* mask, uint32_t width, uint32_t height, uint32_t max_contours, int32_t max_bands, int64_t* offsets,
                          int32_t* pts_xy, int32_t* types, int64_t cap_points) {
    try {
        std::vector<int32_t> scratch((size_t)width * height);
        std::vector<int> cuts = host::blank_row_bands(mask, (int)width, (int)height, max_bands < 1 ? 1 : max_bands);
        int32_t n = 0;
        int64_t np = 0;
        if (offsets) offsets[0] = 0;
        for (size_t i = 0; i + 1 < cuts.size(); ++i) {
            auto part = host::find_contours_band(mask, (int)width, (int)height, cuts[i], cuts[i + 1], max_contours, scratch.data());
            for (auto& c : part) {
                if ((uint32_t)n >= max_contours) break;
                for (auto& q : c.pts) {
                    if (pts_xy && np < cap_points) { pts_xy[np * 2] = (int32_t)q.x; pts_xy[np * 2 + 1] = (int32_t)q.y; }
                    ++np;
                }
                if (types) types[n] = c.hole ? 1 : 0;
                ++n;
                if (offsets) offsets[n] = np;
            }
        }
        return n;
    } catch (...) { return -1; }
}

oar_status oar_k_contours(const uint8_t* mask, uint32_t width, uint32_t height, uint32_t max_contours, int32_t* n_contours, int64_t* offsets,
                          int32_t* pts_xy, int32_t* types, int64_t cap_points) {
    return guard([&] {
        OAR_CHECK(mask && width > 0 && height > 0 && n_contours, OAR_INVALID_INPUT, "oar_k_contours: bad arguments");
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) fail(OAR_DEVICE, "no HIP device visible: libOarMi355x has no CPU fallback");
        DevBuf dm;
        const size_t hw = (size_t)width * height;
        dm.reserve(hw);
        OAR_HIP(hipMemcpy(dm.p, mask, hw, hipMemcpyHostToDevice));
        std::vector<host::Contour> cs = Detector::trace_device_mask(dm.as<uint8_t>(), (int)height, (int)width, max_contours);
        int64_t np = 0;
        if (offsets) offsets[0] = 0;
        int32_t n = 0;
        for (auto& c : cs) {
            for (auto& q : c.pts) {
                if (pts_xy && np < cap_points) { pts_xy[np * 2] = (int32_t)q.x; pts_xy[np * 2 + 1] = (int32_t)q.y; }
                ++np;
            }
            if (types) types[n] = c.hole ? 1 : 0;
            ++n;
            if (offsets) offsets[n] = np;
        }
        *n_contours = n;
    });
}
int32_t oar_host_unclip(const float box8[8], float ratio, float* out_xy, int32_t cap_points) {
    try {
        host::Pt b[4];
        for (int k = 0; k < 4; ++k) b[k] = {box8[k * 2], box8[k * 2 + 1]};
        std::vector<host::Pt> r = host::unclip(b, ratio);
        for (size_t i = 0; i < r.size() && (int32_t)i < cap_points; ++i) { out_xy[i * 2] = r[i].x; out_xy[i * 2 + 1] = r[i].y; }
        return (int32_t)r.size();
    } catch (...) { return -1; }
}
oar_status oar_image_decode(const uint8_t* bytes, size_t len, uint8_t** rgb, uint32_t* width, uint32_t* height) {
    return guard([&] {
        OAR_CHECK(bytes && rgb && width && height, OAR_INVALID_INPUT, "oar_image_decode: bad arguments");
        *rgb = nullptr; *width = *height = 0;
        if (img::is_jpeg(bytes, len)) {   // baseline / progressive Huffman JPEG (jpeg_decode.cc); pixel half on the host for this host-output entry
            img::JpegImage ji;
            img::jpeg_entropy_decode(bytes, len, ji);
            std::vector<uint8_t> px;
            img::jpeg_render_host(ji, px);
            uint8_t* out = cmalloc<uint8_t>(px.size());
            std::memcpy(out, px.data(), px.size());
            *rgb = out; *width = ji.w; *height = ji.h;
            return;
        }
        std::vector<uint8_t> px;
        uint32_t w = 0, h = 0;
        if (!img::is_png(bytes, len)) {
            if (!img::decode_misc(bytes, len, px, w, h)) {
                const char* what = img::sniff(bytes, len);
                if (what) fail(OAR_UNSUPPORTED_OP, std::string("image load: ") + what + " is not decoded by this library (PNG, JPEG, BMP, PNM, TIFF and GIF are); use the reference's loader for it");
                fail(OAR_INVALID_INPUT, "image load: unrecognised image format");
            }
        } else {
            img::decode_png(bytes, len, px, w, h);
        }
        uint8_t* out = cmalloc<uint8_t>(px.size());
        std::memcpy(out, px.data(), px.size());
        *rgb = out; *width = w; *height = h;
    });
}
void oar_image_free(uint8_t* rgb) { std::free(rgb); }

oar_status oar_image_decode_device(const uint8_t* bytes, size_t len, int32_t device_id, void** dev_rgb, uint32_t* width, uint32_t* height) {
    return guard([&] {
        OAR_CHECK(bytes && dev_rgb && width && height, OAR_INVALID_INPUT, "oar_image_decode_device: bad arguments");
        *dev_rgb = nullptr; *width = *height = 0;
        require_device();
        OAR_HIP(hipSetDevice(device_id));
        if (img::is_jpeg(bytes, len)) {
            // entropy decoding on the host (serial by nature), the arithmetic on the GPU: coefficient planes up, RGB page born in HBM
            img::JpegImage ji;
            img::jpeg_entropy_decode(bytes, len, ji);
            pp::JpegDevPlan plan{};
            plan.ncomp = ji.ncomp; plan.hmax = ji.hmax; plan.vmax = ji.vmax; plan.color = ji.color; plan.w = ji.w; plan.h = ji.h;
            size_t coef_bytes = 0, plane_bytes = 0;
            for (int c = 0; c < ji.ncomp; ++c) { coef_bytes += ji.comp[c].coef.size() * 2; plane_bytes += (size_t)ji.comp[c].bw * ji.comp[c].bh * 64; }
            DevBuf work;   // [coefficients][planes][quantisation tables]
            work.reserve(coef_bytes + plane_bytes + 3 * 64 * 2 + 256);
            uint8_t* base = work.as<uint8_t>();
            size_t co = 0, po = coef_bytes;
            uint16_t qh[3 * 64] = {0};
            for (int c = 0; c < ji.ncomp; ++c) {
                const img::JpegComp& k = ji.comp[c];
                OAR_HIP(hipMemcpyAsync(base + co, k.coef.data(), k.coef.size() * 2, hipMemcpyHostToDevice, nullptr));
                plan.comp[c] = pp::JpegDevComp{reinterpret_cast<const int16_t*>(base + co), base + po, k.h, k.v, k.bw, k.bh, k.dw, k.dh};
                plan.total_blocks += (long)k.bw * k.bh;
                co += k.coef.size() * 2; po += (size_t)k.bw * k.bh * 64;
                std::memcpy(qh + c * 64, k.q, 128);
            }
            const size_t qo = (po + 127) & ~(size_t)127;
            OAR_HIP(hipMemcpyAsync(base + qo, qh, sizeof qh, hipMemcpyHostToDevice, nullptr));
            plan.q = reinterpret_cast<const uint16_t*>(base + qo);
            // the page buffer is owned by a guard until it is handed to the caller: jpeg_render / the profiler scope may throw, and the
            // pageable sources of the copies above (ji, qh) must outlive them on every path -- so the stream is drained before unwinding
            struct PageGuard {
                void* p = nullptr;
                ~PageGuard() { (void)hipStreamSynchronize(nullptr); if (p) (void)hipFree(p); (void)hipGetLastError(); }
            } page;
            OAR_HIP(hipMalloc(&page.p, (size_t)ji.w * ji.h * 3));
            pp::jpeg_render(nullptr, plan, static_cast<uint8_t*>(page.p));
            const hipError_t e = hipStreamSynchronize(nullptr);
            if (e != hipSuccess) fail(OAR_DEVICE, std::string("oar_image_decode_device: ") + hipGetErrorString(e));
            *dev_rgb = page.p; page.p = nullptr; *width = ji.w; *height = ji.h;
            return;
        }
        // every other decoded format: host decode, one upload
        uint8_t* host = nullptr;
        uint32_t w = 0, h = 0;
        const oar_status st = oar_image_decode(bytes, len, &host, &w, &h);
        if (st != OAR_OK) { char msg[512]; oar_last_error(msg, sizeof msg); fail(st, msg); }
        void* out = nullptr;
        const hipError_t e1 = hipMalloc(&out, (size_t)w * h * 3);
        const hipError_t e2 = e1 == hipSuccess ? hipMemcpy(out, host, (size_t)w * h * 3, hipMemcpyHostToDevice) : e1;
        std::free(host);
        if (e2 != hipSuccess) { if (out) (void)hipFree(out); fail(OAR_DEVICE, std::string("oar_image_decode_device: ") + hipGetErrorString(e2)); }
        *dev_rgb = out; *width = w; *height = h;
    });
}

oar_status oar_k_ppdoc_postprocess(const float* pred, uint32_t n_images, uint32_t rows, uint32_t feat, const float* src_wh, uint32_t num_classes, const oar_ppdoc_cfg* cfg,
                                   oar_layout_result* out) {
    return guard([&] {
        OAR_CHECK(out && cfg && (n_images == 0 || (src_wh && (rows == 0 || pred))) && rows <= 16384 && num_classes > 0 && num_classes <= 4096, OAR_INVALID_INPUT,
                  "oar_k_ppdoc_postprocess: bad arguments");
        OAR_CHECK(rows == 0 || (feat >= 6 && feat <= 8), OAR_INVALID_INPUT, "oar_k_ppdoc_postprocess: feat must be 6, 7 or 8");
        require_device();
        LayoutOut lo;
        lo.offsets.assign(1, 0);
        lo.feature_dim = feat;
        if (n_images == 0 || rows == 0) { for (uint32_t i = 0; i < n_images; ++i) lo.offsets.push_back(0); fill_layout_result(lo, out); return; }
        DevBuf dpred, dwh, dcand, dsorted, dkeep, dcfg;
        const size_t np = (size_t)n_images * rows * feat;
        dpred.reserve(np * 4); dwh.reserve((size_t)n_images * 8); dcand.reserve((size_t)n_images * rows * 32); dsorted.reserve((size_t)n_images * rows * 4);
        dkeep.reserve((size_t)n_images * (rows + 1) * 4); dcfg.reserve((size_t)num_classes * 8 + 16);
        OAR_HIP(hipMemcpy(dpred.p, pred, np * 4, hipMemcpyHostToDevice));
        OAR_HIP(hipMemcpy(dwh.p, src_wh, (size_t)n_images * 8, hipMemcpyHostToDevice));
        if (cfg->class_thresholds) OAR_HIP(hipMemcpy(dcfg.p, cfg->class_thresholds, (size_t)num_classes * 4, hipMemcpyHostToDevice));
        if (cfg->class_merge_modes) OAR_HIP(hipMemcpy(dcfg.as<uint8_t>() + (size_t)num_classes * 4, cfg->class_merge_modes, (size_t)num_classes * 4, hipMemcpyHostToDevice));
        pp::PpDocPostP q{};
        q.pred = dpred.as<float>(); q.rows = (int)rows; q.feat = (int)feat; q.num_classes = (int)num_classes; q.score_thr = cfg->score_threshold;
        q.class_thr = cfg->class_thresholds ? dcfg.as<float>() : nullptr; q.layout_nms = cfg->layout_nms ? 1 : 0; q.image_class = cfg->image_class_id; q.formula_class = cfg->formula_class_id;
        q.merge_mode = cfg->class_merge_modes ? reinterpret_cast<const int*>(dcfg.as<uint8_t>() + (size_t)num_classes * 4) : nullptr;
        q.src_wh = dwh.as<float>(); q.cand = dcand.as<float>(); q.sorted = dsorted.as<int>(); q.keep = dkeep.as<int>(); q.n_keep = dkeep.as<int>() + (size_t)n_images * rows;
        pp::ppdoc_postprocess(nullptr, q, (int)n_images);
        OAR_HIP(hipDeviceSynchronize());
        std::vector<int> keep((size_t)n_images * (rows + 1));
        std::vector<float> cand((size_t)n_images * rows * 8);
        OAR_HIP(hipMemcpy(keep.data(), dkeep.p, keep.size() * 4, hipMemcpyDeviceToHost));
        OAR_HIP(hipMemcpy(cand.data(), dcand.p, cand.size() * 4, hipMemcpyDeviceToHost));
        for (uint32_t i = 0; i < n_images; ++i) {
            const int nk = keep[(size_t)n_images * rows + i];
            for (int k = 0; k < nk; ++k) {
                const float* c8 = cand.data() + ((size_t)i * rows + (size_t)keep[(size_t)i * rows + k]) * 8;
                lo.boxes.insert(lo.boxes.end(), c8, c8 + 4);
                lo.scores.push_back(c8[4]);
                int32_t cls; std::memcpy(&cls, &c8[5], 4);
                lo.classes.push_back(cls);
            }
            lo.offsets.push_back((uint32_t)lo.scores.size());
        }
        fill_layout_result(lo, out);
    });
}
int32_t oar_host_nms_with_merge(const float* boxes, const int32_t* classes, const float* scores, uint32_t n, const int32_t* mode_of_class, uint32_t num_classes,
                                float nms_threshold, uint32_t max_detections, float* out_boxes, int32_t* out_classes, float* out_scores) {
    try {
        if (n == 0) return 0;
        if (!boxes || !classes || !scores || !mode_of_class || !out_boxes || !out_classes || !out_scores) return -1;
        return host::nms_with_merge(boxes, classes, scores, (int)n, mode_of_class, (int)num_classes, nms_threshold, (int)max_detections, out_boxes, out_classes, out_scores);
    } catch (...) { return -1; }
}
int32_t oar_host_approx_poly_dp(const float* xy, int32_t n_points, float epsilon, float* out_xy, int32_t cap_points) {
    try {
        std::vector<host::Pt> p(n_points > 0 ? n_points : 0);
        for (int i = 0; i < n_points; ++i) p[i] = {xy[i * 2], xy[i * 2 + 1]};
        std::vector<host::Pt> r = host::approx_poly_dp(p, epsilon);
        for (size_t i = 0; i < r.size() && (int32_t)i < cap_points; ++i) { out_xy[i * 2] = r[i].x; out_xy[i * 2 + 1] = r[i].y; }
        return (int32_t)r.size();
    } catch (...) { return -1; }
}
float oar_host_perimeter(const float* xy, int32_t n_points) {
    try {
        std::vector<host::Pt> p(n_points > 0 ? n_points : 0);
        for (int i = 0; i < n_points; ++i) p[i] = {xy[i * 2], xy[i * 2 + 1]};
        return host::perimeter(p);
    } catch (...) { return -1.0f; }
}
int32_t oar_host_unclip_poly(const float* xy, int32_t n_points, float ratio, float* out_xy, int32_t cap_points) {
    try {
        std::vector<host::Pt> p(n_points > 0 ? n_points : 0);
        for (int i = 0; i < n_points; ++i) p[i] = {xy[i * 2], xy[i * 2 + 1]};
        std::vector<host::Pt> r = host::unclip_poly(p, ratio);
        for (size_t i = 0; i < r.size() && (int32_t)i < cap_points; ++i) { out_xy[i * 2] = r[i].x; out_xy[i * 2 + 1] = r[i].y; }
        return (int32_t)r.size();
    } catch (...) { return -1; }
}
int32_t oar_host_offset_ring(const int64_t* xy, int32_t n_points, double radius, int64_t* out_xy, int32_t cap_points) {
    try { return host::offset_ring_for_tests(xy, n_points, radius, out_xy, cap_points); } catch (...) { return -1; }
}
int32_t oar_host_ring_outline(const int64_t* xy, int32_t n_points, int32_t negative, int64_t* out_xy, int32_t cap_points) {
    try { return host::ring_outline_for_tests(xy, n_points, negative, out_xy, cap_points); } catch (...) { return -1; }
}
void oar_host_sort_poly_boxes(const float* pts_xy, const uint32_t* offsets, int32_t n, int32_t* order) {
    try {
        if (n <= 0) return;
        std::vector<uint32_t> off(offsets, offsets + n + 1);
        std::vector<float> pts(pts_xy, pts_xy + (size_t)off[n] * 2);
        std::vector<int> o = host::sort_poly_boxes(pts, off);
        for (int i = 0; i < n; ++i) order[i] = o[i];
    } catch (...) {}
}
int32_t oar_host_mini_box(const float* xy, int32_t n_points, float box8[8], float* min_side) {
    try {
        std::vector<host::Pt> p(n_points);
        for (int i = 0; i < n_points; ++i) p[i] = {xy[i * 2], xy[i * 2 + 1]};
        host::Pt mb[4];
        float ms = 0.f;
        if (!host::mini_box(p, mb, ms)) return 0;
        for (int k = 0; k < 4; ++k) { box8[k * 2] = mb[k].x; box8[k * 2 + 1] = mb[k].y; }
        if (min_side) *min_side = ms;
        return 1;
    } catch (...) { return -1; }
}
int32_t oar_host_convex_hull(const float* xy, int32_t n_points, float* out_xy, int32_t cap_points) {
    try {
        if (!xy || n_points < 0 || !out_xy) return -1;
        std::vector<host::Pt> p((size_t)n_points);
        for (int i = 0; i < n_points; ++i) p[(size_t)i] = {xy[i * 2], xy[i * 2 + 1]};
        const std::vector<host::Pt> h = host::convex_hull(p);
        if ((int64_t)h.size() > cap_points) return -1;
        for (size_t k = 0; k < h.size(); ++k) { out_xy[k * 2] = h[k].x; out_xy[k * 2 + 1] = h[k].y; }
        return (int32_t)h.size();
    } catch (...) { return -1; }
}
int32_t oar_host_pool_selftest(int32_t threads, int32_t jobs) {
    try {
        // phase 1: back-to-back loops of varying length
        {
            ThreadPool pool(threads);
            for (int j = 0; j < jobs; ++j) {
                if (j == jobs / 2) pool.set_active(true);   // first half with parked workers (every loop wakes them), second half polling
                const int count = 2 + (j * 7) % 37;
                std::vector<std::atomic<int>> seen(count);
                for (auto& a : seen) a.store(0);
                std::atomic<int> bad{0};
                const int tag = j;
                pool.parallel_for(count, [&seen, &bad, count, tag](int i) {
                    if (i < 0 || i >= count || tag < 0) bad.fetch_add(1);
                    else seen[i].fetch_add(1);
                });
                if (bad.load()) return j + 1;
                for (auto& a : seen) if (a.load() != 1) return j + 1;
            }
        }
        // phase 2: the stale-descriptor window.  Counts GROW from job to job, some workers are late between seeing a
        // generation and reading its descriptor, and the publisher dawdles between writing the descriptor and opening the
        // claim word: a late worker of job j must not obtain index count_j of job j (it would run it twice / on the dead
        // closure) although it may already read job j+1's larger count.  Every job's counters live in one long-lived
        // table so that a stray index shows up as a count != 1 instead of a crash.
        {
            ThreadPool pool(threads);
            ThreadPool::ActiveScope hot(pool);
            std::atomic<uint32_t> rng{12345};
            auto jitter = [&rng](int max_spins) {
                uint32_t x = rng.fetch_add(0x9e3779b9u, std::memory_order_relaxed);
                x ^= x >> 15; x *= 0x2c1b3c6du; x ^= x >> 12;
                if ((x & 3) != 0) return;                       // 1 in 4 calls dawdles
                const int spins = (int)((x >> 8) % (uint32_t)max_spins);
                for (volatile int k = 0; k < spins; ++k) {}
            };
            pool.selftest_worker_delay_ = [&] { jitter(20000); };
            pool.selftest_publish_delay_ = [&] { jitter(4000); };
            const int rounds = std::max(1, jobs / 8), kMax = 96;
            std::vector<std::atomic<int>> table((size_t)rounds * kMax);
            for (auto& a : table) a.store(0);
            std::atomic<int> bad{0};
            for (int j = 0; j < rounds; ++j) {
                const int count = 2 + (j % 24) * 4;             // 2, 6, 10, ... 94, then back to 2
                std::atomic<int>* row = table.data() + (size_t)j * kMax;
                pool.parallel_for(count, [row, &bad, count](int i) {
                    if (i < 0 || i >= count) bad.fetch_add(1);
                    else row[i].fetch_add(1);
                });
                if (bad.load()) return 100000 + j + 1;
            }
            for (int j = 0; j < rounds; ++j) {
                const int count = 2 + (j % 24) * 4;
                for (int i = 0; i < kMax; ++i)
                    if (table[(size_t)j * kMax + i].load() != (i < count ? 1 : 0)) return 100000 + j + 1;
            }
        }
        return 0;
    } catch (...) {
        return -1;
    }
}
void oar_host_sort_quad_boxes(const float* boxes8, int32_t n, int32_t* order) {
    try {
        std::vector<float> b(boxes8, boxes8 + (size_t)n * 8);
        std::vector<int> o = host::sort_quad_boxes(b);
        for (int i = 0; i < n; ++i) order[i] = o[i];
    } catch (...) {}
}
void oar_host_plan_crop(uint32_t img_w, uint32_t img_h, const float box8[8], int32_t plan[8], float inv[9]) {
    try {
        host::CropPlan pl = host::plan_crop((int)img_w, (int)img_h, box8);
        plan[0] = pl.mode; plan[1] = pl.left; plan[2] = pl.top; plan[3] = pl.cw; plan[4] = pl.ch;
        plan[5] = pl.mode ? pl.out_w() : 0; plan[6] = pl.mode ? pl.out_h() : 0; plan[7] = pl.rot;
        std::memcpy(inv, pl.inv, sizeof pl.inv);
    } catch (...) { plan[0] = 0; }
}

// ---------------------------------------------------------------------------------------------- test hooks
oar_status oar_debug_inject_failure(const char* site, int32_t count) {
    return guard([&] {
        OAR_CHECK(site, OAR_INVALID_INPUT, "oar_debug_inject_failure: site is null");
        if (std::string(site) == "batched_detection") g_inject_batched_det_failures.store(count);
        else fail(OAR_INVALID_INPUT, std::string("oar_debug_inject_failure: unknown site '") + site + "'");
    });
}

// ---------------------------------------------------------------------------------------------- profiling
void oar_prof_reset(void) {
    try { Profiler::get().reset(); } catch (...) {}
}
void oar_prof_enable(int32_t on) {
    Profiler& p = Profiler::get();
    p.flush();   // captured graphs carry the instrumentation of the epoch they were captured in
    std::lock_guard<std::mutex> lk(p.mu);
    if (p.enabled != (on != 0)) { p.enabled = on != 0; ++p.epoch; }
}
void oar_prof_sampling(int32_t stride, int32_t phase) {
    Profiler& p = Profiler::get();
    p.sample_stride.store(stride < 1 ? 1 : stride, std::memory_order_relaxed);
    p.sample_phase.store(phase, std::memory_order_relaxed);
    p.sample_counter.store(0, std::memory_order_relaxed);
}
void oar_prof_filter(const char* cls) {
    Profiler& p = Profiler::get();
    p.flush();
    std::lock_guard<std::mutex> lk(p.mu);
    std::string f = cls ? cls : "";
    if (p.filter != f) { p.filter = f; ++p.epoch; }
}
int32_t oar_prof_snapshot(oar_prof_entry* entries, int32_t cap) {
    try {
        Profiler& p = Profiler::get();
        p.flush();
        std::lock_guard<std::mutex> lk(p.mu);
        std::vector<oar_prof_entry> v = p.totals;
        std::sort(v.begin(), v.end(), [](const oar_prof_entry& a, const oar_prof_entry& b) { return a.total_ms > b.total_ms; });
        int n = (int)std::min<size_t>(v.size(), (size_t)std::max(cap, 0));
        for (int i = 0; i < n; ++i) entries[i] = v[i];
        return (int32_t)v.size();
    } catch (...) {
        return 0;
    }
}

}  // extern "C"
