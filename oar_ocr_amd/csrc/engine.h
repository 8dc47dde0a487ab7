// engine.h -- ONNX graph -> shape-specialised plan of HIP kernel launches.
// Stands where `OrtInfer` stands in the reference (core/inference/ort_infer_execution.rs:121-306): one f32
// input, all graph outputs.  Feature maps live in HBM as NHWC ("channels-last"); the ONNX (NCHW) view is
// only materialised at the boundary.
#pragma once
#include <cstdlib>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <tuple>
#include <vector>

#include "common.h"
#include "kernels.h"
#include "onnx_parse.h"

namespace oar {

enum class Layout : int { NATIVE = 0, CLAST = 1 };  // CLAST: logical [n, C, s...] stored as [n, s..., C]

struct GNode {  // graph node after load-time rewrites
    std::string op;
    std::vector<std::string> in, out;
    std::map<std::string, Attr> attrs;
    k::Act act;                 // fused activation
    std::string bias;           // fused bias initializer (Linear)
    std::string residual;       // fused residual input
    int id = 0;
    int64_t ai(const char* k, int64_t d) const { auto it = attrs.find(k); return it == attrs.end() ? d : it->second.i; }
    float af(const char* k, float d) const { auto it = attrs.find(k); return it == attrs.end() ? d : it->second.f; }
    std::string as(const char* k, const std::string& d) const { auto it = attrs.find(k); return it == attrs.end() ? d : it->second.s; }
    std::vector<int64_t> ais(const char* k) const { auto it = attrs.find(k); return it == attrs.end() ? std::vector<int64_t>{} : it->second.is; }
    bool has(const char* k) const { return attrs.count(k) != 0; }
};

struct Loc {  // where a value lives at run time
    enum Kind { NONE, INPUT, ARENA, CONST, EXTRA } kind = NONE;
    int64_t off = 0;             // byte offset (ARENA / INPUT / EXTRA)
    int idx = 0;                 // EXTRA: which secondary graph input
    const float* cptr = nullptr; // CONST
};

struct RunCtx {
    hipStream_t s;
    const float* input;
    char* arena;
    const float* const* extra = nullptr;   // secondary graph inputs (Engine::run_multi), device pointers
    const k::StemU8* stem = nullptr;       // Engine::run_stem: the u8 pages the fused stem reads (no f32 input tensor exists)
    const float* at(const Loc& l) const {
        switch (l.kind) {
            case Loc::EXTRA: return reinterpret_cast<const float*>(reinterpret_cast<const char*>(extra[l.idx]) + l.off);
            case Loc::INPUT: return reinterpret_cast<const float*>(reinterpret_cast<const char*>(input) + l.off);
            case Loc::ARENA: return reinterpret_cast<const float*>(arena + l.off);
            case Loc::CONST: return l.cptr;
            default: return nullptr;
        }
    }
    float* mut(const Loc& l) const { return const_cast<float*>(at(l)); }
};

struct PlanOutput {
    std::string name;
    std::vector<int64_t> dims;  // logical (ONNX) dims
    Loc loc;                    // native layout
    Loc loc_clast;              // valid when has_clast: the channels-last copy (no conversion needed)
    bool has_clast = false;
    int dtype = 1;              // onnx elem type the caller sees: 1 = f32, 7 = i64 (device storage is f32 either way)
    bool on_host = false;       // the value was computed at plan time (shape arithmetic): host_vals holds it, loc is unused
    std::vector<double> host_vals;
};

struct Plan {
    std::vector<std::function<void(const RunCtx&)>> steps;
    std::vector<std::string> step_ops;   // OAR_DEBUG_STEPS=1: the graph node (op -> first output) behind each step, for error attribution
    size_t arena_bytes = 0;
    std::vector<PlanOutput> outputs;
    double flops = 0, bytes = 0;
    int n_kernels = 0;
    bool skipped_softmax = false;
    mutable uint64_t last_used = 0;   // Engine's LRU tick
    mutable int runs = 0;   // completed uncaptured runs (a plan is captured into a hipGraph from its second run on)
    int logits_valid = 0;   // > 0: output[0]'s rows are padded; only the first logits_valid columns are logits
    Loc ctc_part;           // kind != NONE: output[0] was never materialised; softmax partials [rows][ctc_tiles] float4 live here
    int ctc_tiles = 0;
};

class Engine {
   public:
    Engine(const uint8_t* onnx, size_t len, int device_id, hipStream_t caller_stream = nullptr);   // caller_stream: oar_engine_cfg.stream (null = own stream)
    ~Engine();
    const std::string& input_name() const { return input_name_; }
    // declared graph inputs (initializers excluded) / outputs, with the shapes the model file declares (-1 = dynamic):
    // OrtInfer::input_names_from_model / primary_input_shape / output_shapes (core/inference/mod.rs:66-112)
    const std::vector<ValueInfo>& input_infos() const { return input_infos_; }
    const std::vector<ValueInfo>& output_infos() const { return output_infos_; }
    int device() const { return device_; }
    hipStream_t stream() const { return stream_; }

    // d_in: device pointer. dims: logical ONNX dims. in_clast: rank-4 input already stored NHWC.
    // Outputs stay on the device (valid until the next run on this engine).
    // skip_final_softmax: when output[0] is produced by a Softmax over its last axis, stop before it and return the
    // logits instead (Plan::skipped_softmax is set) -- the recognizer fuses that softmax with the CTC argmax.
    const Plan& run(const float* d_in, const std::vector<int64_t>& dims, bool in_clast, bool skip_final_softmax = false);
    const Plan& plan_for(const std::vector<int64_t>& dims, bool in_clast, bool skip_final_softmax = false,
                         const std::vector<std::vector<int64_t>>* extra_dims = nullptr, bool stem_u8 = false);
    // The graph input feeds exactly one node, an RGB stem convolution: the detector may hand the engine its u8 pages
    // (run_stem) and skip the normalised f32 input tensor altogether.
    bool stem_fusable() const { return stem_fusable_; }
    // true while plans may be captured / replayed as hipGraphs on stream(): capture is a property of the STREAM, so no other thread may
    // submit to it meanwhile (Detector::run keeps its helper enqueue thread off in that mode)
    static bool graphs_requested() { static const bool on = [] { const char* e = getenv("OAR_HIP_GRAPH"); return e && atoi(e) != 0; }(); return on; }
    bool graphs_enabled() const { return graphs_requested() && graphs_ok_; }
    // dims = {n, 3, H, W}; st.pages[0..n) are device pointers to H x W x 3 u8 pages; st.src / alpha / beta as pp::normalize
    // st.dev (device table of per-image pointers / widths, kernels.h): the recognizer's form, any n
    const Plan& run_stem(const k::StemU8& st, const std::vector<int64_t>& dims, bool skip_final_softmax = false);
    // Several named inputs (OrtInfer::infer, ort_infer_execution.rs:121-219): d_ins[i] / dims[i] belong to input_infos()[i]
    // (the caller has matched the names); input 0 is the primary one, the others are bound as plain f32 device tensors.
    const Plan& run_multi(const std::vector<const float*>& d_ins, const std::vector<std::vector<int64_t>>& dims);
    const float* out_ptr(const Loc& l) const;
    char* arena() const { return arena_.as<char>(); }
    std::mutex& mutex() { return mu_; }
    static const std::set<std::string>& supported_ops();
    static void validate_model(const OnnxModel& m);   // throws OAR_MODEL_LOAD; host-only
    size_t cached_plans() const { return plans_.size(); }
    uint64_t evicted_plans() const { return plans_evicted_; }

   private:
    friend struct Planner;
    void rewrite_graph(OnnxModel& m);
    const float* upload_const(const std::string& key, const std::vector<float>& v);

    int device_ = 0;
    int64_t opset_ = 17;
    hipStream_t stream_ = nullptr;
    bool owns_stream_ = true;   // false: oar_engine_cfg.stream (the caller's)
    std::string input_name_;
    std::vector<std::string> output_names_;
    std::vector<ValueInfo> input_infos_, output_infos_;
    std::vector<const float*> last_extra_;
    bool stem_fusable_ = false;
    std::map<std::string, HostTensor> inits_;
    std::vector<GNode> nodes_;
    std::map<std::string, const float*> dev_consts_;
    std::vector<void*> dev_allocs_;
    std::map<std::string, std::unique_ptr<Plan>> plans_;
    size_t plan_cap_ = 256;              // OAR_PLAN_CACHE: max cached plans (LRU)
    uint64_t tick_ = 0, plans_evicted_ = 0;
    const Plan* last_returned_ = nullptr;
    void evict_plans();
    DevBuf arena_;
    // hipGraph replay (opt-in, OAR_HIP_GRAPH=1; measured at parity with plain launches, see Engine::replay): from its
    // second run a plan is captured once per (input pointer, arena base, profiler epoch) and replayed.
    struct GraphEntry {
        hipGraphExec_t exec = nullptr;
        std::shared_ptr<Profiler::GraphEvents> events;
        int epoch = 0;
    };
    std::map<std::tuple<const Plan*, const void*, const void*>, GraphEntry> graphs_;
    bool graphs_ok_ = true;
    void clear_graphs();
    bool replay(const Plan& p, const RunCtx& c);
    std::mutex mu_;
    const float* last_input_ = nullptr;
};

}  // namespace oar
