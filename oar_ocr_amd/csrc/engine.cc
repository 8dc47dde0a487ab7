#include "engine.h"

#include "dsblock.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <set>
#include <sstream>

namespace oar {

using k::Act;

// =================================================================================================
// construction + load-time graph rewrites
// =================================================================================================
static int64_t numel(const std::vector<int64_t>& d) {
    int64_t n = 1;
    for (auto v : d) n *= v;
    return n;
}

Engine::Engine(const uint8_t* onnx, size_t len, int device_id, hipStream_t caller_stream) : device_(device_id) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        fail(OAR_DEVICE, "no HIP device visible: libOarMi355x has no CPU fallback");
    OAR_CHECK(device_id >= 0 && device_id < ndev, OAR_DEVICE, "device_id out of range");
    OAR_HIP(hipSetDevice(device_));
    if (caller_stream) { stream_ = caller_stream; owns_stream_ = false; }   // oar_engine_cfg.stream: the caller's stream, never destroyed here
    else OAR_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
    OnnxModel m = parse_onnx(onnx, len);
    validate_model(m);
    input_name_ = m.inputs[0];
    output_names_ = m.outputs;
    input_infos_ = m.input_infos;
    output_infos_ = m.output_infos;
    if (const char* e = getenv("OAR_PLAN_CACHE")) { long v = atol(e); if (v >= 2) plan_cap_ = (size_t)v; }
    opset_ = m.opset;
    rewrite_graph(m);
    // fused-stem eligibility: the input is consumed once, by a group-1 convolution with 3 input channels and a small kernel
    {
        int uses = 0;
        const GNode* consumer = nullptr;
        for (const GNode& n : nodes_) {
            for (size_t i = 0; i < n.in.size(); ++i) if (n.in[i] == input_name_) { ++uses; consumer = &n; if (i != 0) uses += 100; }
            if (n.residual == input_name_) uses += 100;
        }
        for (auto& on : output_names_) if (on == input_name_) uses += 100;
        if (uses == 1 && consumer && consumer->op == "Conv" && consumer->ai("group", 1) == 1 && input_infos_.size() == 1) {
            auto it = inits_.find(consumer->in.size() > 1 ? consumer->in[1] : std::string());
            if (it != inits_.end() && it->second.dims.size() == 4 && it->second.dims[1] == 3 && it->second.dims[2] * it->second.dims[3] * 3 <= 128)
                stem_fusable_ = true;
        }
    }
}

Engine::~Engine() {
    (void)hipSetDevice(device_);
    if (stream_) (void)hipStreamSynchronize(stream_);
    clear_graphs();
    Profiler::get().drop_events();   // no pooled event may outlive the stream it was recorded on
    for (void* p : dev_allocs_) (void)hipFree(p);
    if (stream_ && owns_stream_) (void)hipStreamDestroy(stream_);
    (void)hipGetLastError();
}

const float* Engine::upload_const(const std::string& key, const std::vector<float>& v) {
    auto it = dev_consts_.find(key);
    if (it != dev_consts_.end()) return it->second;
    void* p = nullptr;
    size_t bytes = std::max<size_t>(v.size() * sizeof(float), 16);
    OAR_HIP(hipMalloc(&p, bytes));
    dev_allocs_.push_back(p);
    if (!v.empty()) OAR_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    dev_consts_[key] = (const float*)p;
    return (const float*)p;
}

// activations that may be folded into a conv / linear / add epilogue (apply_act).  The rest of is_unary_act only ever runs
// as a stand-alone element-wise kernel (apply_unary): every extra case in apply_act costs registers in each conv epilogue --
// with all of them in, the depthwise kernel lost a wave per SIMD and a quarter of its bandwidth.
static bool is_fusable_act(const std::string& op) {
    return op == "Relu" || op == "HardSwish" || op == "HardSigmoid" || op == "Sigmoid" || op == "LeakyRelu" || op == "Tanh" || op == "Gelu";
}
static bool is_unary_act(const std::string& op) {
    return op == "Relu" || op == "HardSwish" || op == "HardSigmoid" || op == "Sigmoid" || op == "LeakyRelu" || op == "Tanh" || op == "Erf" ||
           op == "Sqrt" || op == "Exp" || op == "Abs" || op == "Neg" || op == "Reciprocal" || op == "Log" || op == "Gelu" || op == "Softplus" ||
           op == "Floor" || op == "Ceil" || op == "Round" || op == "Not";
}
static Act act_of(const GNode& n) {
    Act a;
    if (n.op == "Relu") a.kind = k::ACT_RELU;
    else if (n.op == "HardSwish") a.kind = k::ACT_HSWISH;
    else if (n.op == "HardSigmoid") { a.kind = k::ACT_HSIGMOID; a.alpha = n.af("alpha", 0.2f); a.beta = n.af("beta", 0.5f); }
    else if (n.op == "Sigmoid") a.kind = k::ACT_SIGMOID;
    else if (n.op == "LeakyRelu") { a.kind = k::ACT_LEAKY; a.alpha = n.af("alpha", 0.01f); }
    else if (n.op == "Tanh") a.kind = k::ACT_TANH;
    else if (n.op == "Erf") a.kind = k::ACT_ERF;
    else if (n.op == "Sqrt") a.kind = k::ACT_SQRT;
    else if (n.op == "Exp") a.kind = k::ACT_EXP;
    else if (n.op == "Abs") a.kind = k::ACT_ABS;
    else if (n.op == "Neg") a.kind = k::ACT_NEG;
    else if (n.op == "Reciprocal") a.kind = k::ACT_RECIP;
    else if (n.op == "Log") a.kind = k::ACT_LOG;
    else if (n.op == "Gelu") a.kind = n.as("approximate", "none") == "tanh" ? k::ACT_GELU_TANH : k::ACT_GELU_ERF;
    else if (n.op == "Softplus") a.kind = k::ACT_SOFTPLUS;
    else if (n.op == "Floor") a.kind = k::ACT_FLOOR;
    else if (n.op == "Ceil") a.kind = k::ACT_CEIL;
    else if (n.op == "Round") a.kind = k::ACT_ROUND;
    else if (n.op == "Not") a.kind = k::ACT_NOT;
    return a;
}

void Engine::rewrite_graph(OnnxModel& m) {
    inits_ = std::move(m.initializers);
    std::vector<GNode> nodes;
    for (auto& on : m.nodes) {
        if (on.op == "Constant") {
            auto it = on.attrs.find("value");
            OAR_CHECK(it != on.attrs.end() && it->second.kind == Attr::T, OAR_UNSUPPORTED_OP, "Constant without tensor value");
            HostTensor t = it->second.t;
            t.name = on.outputs[0];
            inits_[t.name] = std::move(t);
            continue;
        }
        if (on.op == "Dropout") { on.op = "Identity"; on.outputs.resize(1); }
        GNode g;
        g.op = on.op; g.in = on.inputs; g.out = on.outputs; g.attrs = on.attrs;
        nodes.push_back(std::move(g));
    }
    std::set<std::string> graph_outs(output_names_.begin(), output_names_.end());
    auto consumers = [&](const std::vector<GNode>& ns) {
        std::map<std::string, std::vector<int>> c;
        for (int i = 0; i < (int)ns.size(); ++i)
            for (auto& s : ns[i].in)
                if (!s.empty()) c[s].push_back(i);
        return c;
    };
    auto is_init = [&](const std::string& s) { return inits_.count(s) != 0; };

    // ---- pass 1: fold BatchNormalization into the producing Conv / ConvTranspose
    {
        auto cons = consumers(nodes);
        std::map<std::string, int> producer;
        for (int i = 0; i < (int)nodes.size(); ++i) for (auto& o : nodes[i].out) producer[o] = i;
        std::vector<bool> dead(nodes.size(), false);
        for (int i = 0; i < (int)nodes.size(); ++i) {
            GNode& bn = nodes[i];
            if (bn.op != "BatchNormalization" || bn.in.size() < 5) continue;
            auto pit = producer.find(bn.in[0]);
            if (pit == producer.end()) continue;
            GNode& cv = nodes[pit->second];
            if (cv.op != "Conv" && cv.op != "ConvTranspose") continue;
            if (cons[bn.in[0]].size() != 1 || graph_outs.count(bn.in[0])) continue;
            if (!is_init(cv.in[1]) || !is_init(bn.in[1]) || !is_init(bn.in[2]) || !is_init(bn.in[3]) || !is_init(bn.in[4])) continue;
            if (cv.in.size() > 2 && !cv.in[2].empty() && !is_init(cv.in[2])) continue;
            const HostTensor& W = inits_[cv.in[1]];
            const auto &ga = inits_[bn.in[1]].f, &be = inits_[bn.in[2]].f, &mu = inits_[bn.in[3]].f, &va = inits_[bn.in[4]].f;
            float eps = bn.af("epsilon", 1e-5f);
            int64_t C = (int64_t)ga.size();
            OAR_CHECK(W.dtype == DType::F32 && W.dims.size() == 4, OAR_MODEL_LOAD, "BN fold: conv weight must be a rank-4 f32 initializer (" + cv.in[1] + ")");
            OAR_CHECK(C > 0 && (int64_t)be.size() == C && (int64_t)mu.size() == C && (int64_t)va.size() == C, OAR_MODEL_LOAD,
                      "BatchNormalization: scale / bias / mean / var must all have C elements (" + bn.out[0] + ")");
            if (cv.in.size() > 2 && !cv.in[2].empty())
                OAR_CHECK((int64_t)inits_[cv.in[2]].f.size() == C, OAR_MODEL_LOAD, "BN fold: conv bias must have C elements (" + cv.in[2] + ")");
            HostTensor W2 = W;
            std::vector<float> b2(C, 0.f);
            if (cv.in.size() > 2 && !cv.in[2].empty()) b2 = inits_[cv.in[2]].f;
            std::vector<float> sc(C);
            for (int64_t c = 0; c < C; ++c) sc[c] = ga[c] / std::sqrt(va[c] + eps);
            if (cv.op == "Conv") {
                OAR_CHECK(W.dims[0] == C, OAR_SHAPE_MISMATCH, "BN fold: channel mismatch");
                int64_t per = numel(W.dims) / C;
                for (int64_t c = 0; c < C; ++c)
                    for (int64_t j = 0; j < per; ++j) W2.f[c * per + j] = W.f[c * per + j] * sc[c];
            } else {
                int64_t g = cv.ai("group", 1);
                if (g != 1) continue;
                OAR_CHECK(W.dims[1] == C, OAR_SHAPE_MISMATCH, "BN fold (convT): channel mismatch");
                int64_t khw = W.dims[2] * W.dims[3];
                for (int64_t ci = 0; ci < W.dims[0]; ++ci)
                    for (int64_t c = 0; c < C; ++c)
                        for (int64_t j = 0; j < khw; ++j) W2.f[(ci * C + c) * khw + j] = W.f[(ci * C + c) * khw + j] * sc[c];
            }
            for (int64_t c = 0; c < C; ++c) b2[c] = (b2[c] - mu[c]) * sc[c] + be[c];
            std::string wn = cv.in[1] + "::bnfold" + std::to_string(i), bnm = wn + "::b";
            W2.name = wn;
            inits_[wn] = std::move(W2);
            HostTensor B; B.name = bnm; B.dtype = DType::F32; B.dims = {C}; B.f = b2;
            inits_[bnm] = std::move(B);
            cv.in.resize(3);
            cv.in[1] = wn; cv.in[2] = bnm;
            cv.out[0] = bn.out[0];
            producer[bn.out[0]] = pit->second;
            dead[i] = true;
        }
        std::vector<GNode> keep;
        for (int i = 0; i < (int)nodes.size(); ++i) if (!dead[i]) keep.push_back(std::move(nodes[i]));
        nodes.swap(keep);
    }

    // ---- pass 2: MatMul(x, Wconst) + Add(const bias)  ->  Linear
    {
        auto cons = consumers(nodes);
        std::vector<bool> dead(nodes.size(), false);
        for (int i = 0; i < (int)nodes.size(); ++i) {
            GNode& mm = nodes[i];
            if (mm.op != "MatMul" || !is_init(mm.in[1]) || inits_[mm.in[1]].dims.size() != 2) continue;
            mm.op = "Linear";
            auto& cs = cons[mm.out[0]];
            if (cs.size() != 1 || graph_outs.count(mm.out[0])) continue;
            GNode& ad = nodes[cs[0]];
            if (ad.op != "Add" || dead[cs[0]]) continue;
            int other = ad.in[0] == mm.out[0] ? 1 : 0;
            if (!is_init(ad.in[other])) continue;
            const HostTensor& b = inits_[ad.in[other]];
            if (numel(b.dims) != inits_[mm.in[1]].dims[1] || b.dims.empty() || b.dims.back() != numel(b.dims)) continue;
            mm.bias = ad.in[other];
            mm.out[0] = ad.out[0];
            dead[cs[0]] = true;
        }
        std::vector<GNode> keep;
        for (int i = 0; i < (int)nodes.size(); ++i) if (!dead[i]) keep.push_back(std::move(nodes[i]));
        nodes.swap(keep);
    }

    // ---- pass 2b: GELU as exporters below opset 20 spell it (SVTRv2's MLPs and conv stem): x * 0.5 * (1 + erf(x / sqrt(2))), in either association
    //   Mul(Mul(x, Add(Erf(Div(x, 1.41421)), 1)), 0.5)   or   Mul(Mul(x, 0.5), Add(Erf(..), 1));  x / sqrt(2) may also be Mul(x, 0.70711)
    // becomes ONE Gelu node (5 element-wise launches over the widest tensor of the block -> an epilogue of the producing Linear / Conv, pass 3).
    // OAR_FUSE_GELU=0 keeps the op-by-op path.
    {
        const char* fe = getenv("OAR_FUSE_GELU");
        const bool fuse = !fe || atoi(fe) != 0;
        auto cons = consumers(nodes);
        std::map<std::string, int> producer;
        for (int i = 0; i < (int)nodes.size(); ++i) for (auto& o : nodes[i].out) producer[o] = i;
        std::vector<bool> dead(nodes.size(), false);
        auto scalar_of = [&](const std::string& v, float& out) {
            auto it = inits_.find(v);
            if (it == inits_.end() || it->second.dtype != DType::F32 || it->second.f.size() != 1) return false;
            out = it->second.f[0];
            return true;
        };
        auto single_use = [&](const std::string& v) { return cons[v].size() == 1 && !graph_outs.count(v); };
        // node `i` is op(x, c) or op(c, x) with a scalar constant close to `want`: returns x's name
        auto scaled = [&](int i, const char* op, float want, bool commutes, std::string& x) {
            const GNode& q = nodes[i];
            if (q.op != op || q.in.size() != 2) return false;
            float c;
            if (scalar_of(q.in[1], c) && std::fabs(c - want) <= 1e-4f * std::fabs(want)) { x = q.in[0]; return true; }
            if (commutes && scalar_of(q.in[0], c) && std::fabs(c - want) <= 1e-4f * std::fabs(want)) { x = q.in[1]; return true; }
            return false;
        };
        for (int i = 0; fuse && i < (int)nodes.size(); ++i) {
            if (nodes[i].op != "Erf" || dead[i] || !single_use(nodes[i].out[0])) continue;
            auto pit = producer.find(nodes[i].in[0]);
            if (pit == producer.end() || dead[pit->second] || !single_use(nodes[i].in[0])) continue;
            const int pre = pit->second;
            std::string x;
            if (!scaled(pre, "Div", 1.41421356f, false, x) && !scaled(pre, "Mul", 0.70710678f, true, x)) continue;
            const int add = cons[nodes[i].out[0]][0];
            std::string e1;
            if (dead[add] || !scaled(add, "Add", 1.0f, true, e1) || e1 != nodes[i].out[0] || !single_use(nodes[add].out[0])) continue;
            const int m1 = cons[nodes[add].out[0]][0];
            if (dead[m1] || nodes[m1].op != "Mul" || nodes[m1].in.size() != 2) continue;
            const std::string other = nodes[m1].in[0] == nodes[add].out[0] ? nodes[m1].in[1] : nodes[m1].in[0];
            int last = -1, half = -1;
            if (other == x) {   // (x * (1 + erf)) * 0.5
                if (!single_use(nodes[m1].out[0])) continue;
                const int m2 = cons[nodes[m1].out[0]][0];
                std::string t;
                if (dead[m2] || !scaled(m2, "Mul", 0.5f, true, t) || t != nodes[m1].out[0]) continue;
                last = m2;
            } else {            // (x * 0.5) * (1 + erf)
                auto hit = producer.find(other);
                std::string t;
                if (hit == producer.end() || dead[hit->second] || !single_use(other) || !scaled(hit->second, "Mul", 0.5f, true, t) || t != x) continue;
                half = hit->second;
                last = m1;
            }
            GNode ge;
            ge.op = "Gelu"; ge.in = {x}; ge.out = {nodes[last].out[0]};
            for (int d : {pre, i, add, m1}) dead[d] = true;
            if (half >= 0) dead[half] = true;
            dead[last] = false;
            nodes[last] = std::move(ge);   // the last node of the pattern: every reader comes after it
        }
        std::vector<GNode> keep;
        for (int i = 0; i < (int)nodes.size(); ++i) if (!dead[i]) keep.push_back(std::move(nodes[i]));
        nodes.swap(keep);
    }

    // ---- pass 3: fuse activations into Conv / ConvTranspose / Linear / Gemm / Add
    {
        bool changed = true;
        while (changed) {
            changed = false;
            auto cons = consumers(nodes);
            std::vector<bool> dead(nodes.size(), false);
            for (int i = 0; i < (int)nodes.size(); ++i) {
                GNode& p = nodes[i];
                if (!(p.op == "Conv" || p.op == "ConvTranspose" || p.op == "Linear" || p.op == "Gemm" || p.op == "Add")) continue;
                if (p.act.kind != k::ACT_NONE) continue;
                const std::string& y = p.out[0];
                if (graph_outs.count(y)) continue;
                auto& cs = cons[y];
                if (cs.size() == 1 && is_fusable_act(nodes[cs[0]].op) && !dead[cs[0]] &&
                    !(nodes[cs[0]].op == "Gelu" && nodes[cs[0]].as("approximate", "none") == "tanh")) {   // (the tanh form only exists in the stand-alone kernel)
                    GNode& a = nodes[cs[0]];
                    p.act = act_of(a);
                    p.out[0] = a.out[0];
                    dead[cs[0]] = true;
                    changed = true;
                    continue;
                }
                if (cs.size() == 2) {
                    // x * HardSigmoid(x) (alpha 1/6, beta .5) = HardSwish ; x * Sigmoid(x) = Swish
                    for (int t = 0; t < 2; ++t) {
                        GNode& g = nodes[cs[t]];
                        GNode& mul = nodes[cs[1 - t]];
                        if (dead[cs[0]] || dead[cs[1]]) break;
                        if (!(g.op == "HardSigmoid" || g.op == "Sigmoid") || mul.op != "Mul") continue;
                        if (graph_outs.count(g.out[0]) || cons[g.out[0]].size() != 1 || cons[g.out[0]][0] != cs[1 - t]) continue;
                        bool ok = (mul.in[0] == y && mul.in[1] == g.out[0]) || (mul.in[1] == y && mul.in[0] == g.out[0]);
                        if (!ok) continue;
                        if (g.op == "HardSigmoid") {
                            float al = g.af("alpha", 0.2f), be = g.af("beta", 0.5f);
                            if (std::fabs(al - 1.0f / 6.0f) > 1e-6f || std::fabs(be - 0.5f) > 1e-6f) continue;
                            p.act.kind = k::ACT_HSWISH;
                        } else {
                            p.act.kind = k::ACT_SWISH;
                        }
                        p.out[0] = mul.out[0];
                        dead[cs[0]] = dead[cs[1]] = true;
                        changed = true;
                        break;
                    }
                }
            }
            std::vector<GNode> keep;
            for (int i = 0; i < (int)nodes.size(); ++i) if (!dead[i]) keep.push_back(std::move(nodes[i]));
            nodes.swap(keep);
        }
    }

    // ---- pass 4: Linear / Conv -> Add(residual): fold the residual into the producer's epilogue (no act between);
    // shapes are only known at plan time: op_conv / op_linear fall back to a separate add when they do not match
    {
        auto cons = consumers(nodes);
        std::vector<bool> dead(nodes.size(), false);
        std::map<std::string, int> producer;
        for (int i = 0; i < (int)nodes.size(); ++i) for (auto& o : nodes[i].out) producer[o] = i;
        for (int i = 0; i < (int)nodes.size(); ++i) {
            GNode& ad = nodes[i];
            if (ad.op != "Add" || ad.act.kind != k::ACT_NONE) continue;
            for (int t = 0; t < 2; ++t) {
                auto pit = producer.find(ad.in[t]);
                if (pit == producer.end() || dead[pit->second]) continue;
                GNode& lin = nodes[pit->second];
                if ((lin.op != "Linear" && lin.op != "Conv") || lin.act.kind != k::ACT_NONE || !lin.residual.empty()) continue;
                if (cons[ad.in[t]].size() != 1 || graph_outs.count(ad.in[t])) continue;
                const std::string& other = ad.in[1 - t];
                if (is_init(other)) continue;
                // move the fused Linear to the Add's position so `other` is already computed
                GNode f = lin;
                f.residual = other;
                f.out[0] = ad.out[0];
                dead[pit->second] = true;
                nodes[i] = std::move(f);
                break;
            }
        }
        std::vector<GNode> keep;
        for (int i = 0; i < (int)nodes.size(); ++i) if (!dead[i]) keep.push_back(std::move(nodes[i]));
        nodes.swap(keep);
    }
    // ---- pass 5: the SVTR global-attention block (EncoderWithSVTR), as the recognizer exports emit it:
    //   Reshape[0,-1,3,h,d] -> Transpose[2,0,3,1,4] -> Split(axis 0) -> 3 x Squeeze[0] -> (Mul by a scalar on q) ->
    //   MatMul(q, Transpose[0,1,3,2](k)) -> Softmax(-1) -> MatMul(., v) -> Transpose[0,2,1,3] -> Reshape[0,-1,h*d]
    // becomes ONE Attention node that reads the [n, T, 3*h*d] projection in place (12 glue kernels + 2 batched GEMMs +
    // softmax -> 1 kernel).  OAR_FUSE_ATTENTION=0 keeps the op-by-op path.
    {
        const char* fe = getenv("OAR_FUSE_ATTENTION");
        const bool fuse = !fe || atoi(fe) != 0;
        auto cons = consumers(nodes);
        std::map<std::string, int> producer;
        for (int i = 0; i < (int)nodes.size(); ++i) for (auto& o : nodes[i].out) producer[o] = i;
        std::vector<bool> dead(nodes.size(), false);
        auto prod = [&](const std::string& v, const char* op) -> int {
            auto it = producer.find(v);
            if (it == producer.end() || dead[it->second] || nodes[it->second].op != op) return -1;
            return it->second;
        };
        auto single_use = [&](const std::string& v) { return cons[v].size() == 1 && !graph_outs.count(v); };
        auto ints_of = [&](const std::string& v) -> std::vector<int64_t> {
            auto it = inits_.find(v);
            return it == inits_.end() ? std::vector<int64_t>{} : it->second.i;
        };
        auto perm_is = [&](const GNode& t, std::initializer_list<int64_t> want) { return t.ais("perm") == std::vector<int64_t>(want); };
        auto squeeze0 = [&](const GNode& q) {
            std::vector<int64_t> ax = q.in.size() > 1 ? ints_of(q.in[1]) : q.ais("axes");
            return ax.size() == 1 && ax[0] == 0;
        };
        for (int i = 0; fuse && i < (int)nodes.size(); ++i) {
            if (nodes[i].op != "Softmax" || dead[i]) continue;
            const GNode& sm = nodes[i];
            const int64_t sax = sm.ai("axis", -1);
            if (sax != -1 && sax != 3) continue;
            const int mm1 = prod(sm.in[0], "MatMul");
            if (mm1 < 0 || !single_use(sm.in[0]) || !single_use(sm.out[0])) continue;
            const int mm2 = cons[sm.out[0]][0];
            if (nodes[mm2].op != "MatMul" || nodes[mm2].in[0] != sm.out[0]) continue;
            // q side: Squeeze [-> Mul scalar]
            float scale = 1.0f;
            std::string qv = nodes[mm1].in[0];
            int mul = prod(qv, "Mul");
            if (mul >= 0) {
                if (!single_use(qv)) continue;
                int ci = is_init(nodes[mul].in[1]) ? 1 : is_init(nodes[mul].in[0]) ? 0 : -1;
                if (ci < 0) continue;
                const HostTensor& sc = inits_[nodes[mul].in[ci]];
                if (sc.dtype != DType::F32 || sc.f.size() != 1) continue;
                scale = sc.f[0];
                qv = nodes[mul].in[1 - ci];
            }
            const int sq_q = prod(qv, "Squeeze");
            const int tk = prod(nodes[mm1].in[1], "Transpose");
            if (sq_q < 0 || tk < 0 || !single_use(qv) || !single_use(nodes[mm1].in[1]) || !perm_is(nodes[tk], {0, 1, 3, 2})) continue;
            const int sq_k = prod(nodes[tk].in[0], "Squeeze");
            const int sq_v = prod(nodes[mm2].in[1], "Squeeze");
            if (sq_k < 0 || sq_v < 0 || !single_use(nodes[tk].in[0]) || !single_use(nodes[mm2].in[1])) continue;
            if (!squeeze0(nodes[sq_q]) || !squeeze0(nodes[sq_k]) || !squeeze0(nodes[sq_v])) continue;
            const int sp = prod(nodes[sq_q].in[0], "Split");
            if (sp < 0 || nodes[sp].out.size() != 3 || nodes[sp].ai("axis", 0) != 0) continue;
            if (nodes[sq_q].in[0] != nodes[sp].out[0] || nodes[sq_k].in[0] != nodes[sp].out[1] || nodes[sq_v].in[0] != nodes[sp].out[2]) continue;
            if (!single_use(nodes[sp].out[0]) || !single_use(nodes[sp].out[1]) || !single_use(nodes[sp].out[2])) continue;
            const int t1 = prod(nodes[sp].in[0], "Transpose");
            if (t1 < 0 || !single_use(nodes[sp].in[0]) || !perm_is(nodes[t1], {2, 0, 3, 1, 4})) continue;
            const int r1 = prod(nodes[t1].in[0], "Reshape");
            if (r1 < 0 || !single_use(nodes[t1].in[0]) || nodes[r1].in.size() < 2) continue;
            const std::vector<int64_t> shp = ints_of(nodes[r1].in[1]);
            if (shp.size() != 5 || shp[0] != 0 || shp[1] != -1 || shp[2] != 3 || shp[3] <= 0 || shp[4] <= 0 || shp[4] > 64) continue;
            // output side: Transpose[0,2,1,3] -> Reshape[0,-1,h*d]
            if (!single_use(nodes[mm2].out[0])) continue;
            const int t2 = cons[nodes[mm2].out[0]][0];
            if (nodes[t2].op != "Transpose" || !perm_is(nodes[t2], {0, 2, 1, 3}) || !single_use(nodes[t2].out[0])) continue;
            const int r2 = cons[nodes[t2].out[0]][0];
            if (nodes[r2].op != "Reshape" || nodes[r2].in.size() < 2) continue;
            const std::vector<int64_t> shp2 = ints_of(nodes[r2].in[1]);
            if (shp2.size() != 3 || shp2[0] != 0 || shp2[1] != -1 || shp2[2] != shp[3] * shp[4]) continue;
            GNode at;
            at.op = "Attention";
            at.in = {nodes[r1].in[0]};
            at.out = {nodes[r2].out[0]};
            Attr ah; ah.kind = Attr::I; ah.i = shp[3]; at.attrs["heads"] = ah;
            Attr ad; ad.kind = Attr::I; ad.i = shp[4]; at.attrs["head_dim"] = ad;
            Attr as; as.kind = Attr::F; as.f = scale; at.attrs["scale"] = as;
            for (int d : {r1, t1, sp, sq_q, sq_k, sq_v, tk, mm1, i, mm2, t2}) dead[d] = true;
            if (mul >= 0) dead[mul] = true;
            nodes[r2] = std::move(at);   // the last node of the block: its input is long computed
        }
        std::vector<GNode> keep;
        for (int i = 0; i < (int)nodes.size(); ++i) if (!dead[i]) keep.push_back(std::move(nodes[i]));
        nodes.swap(keep);
    }
    // ---- pass 6: squeeze-excite gate: GlobalAveragePool -> Conv 1x1 (+act) -> Conv 1x1 (+act) on the pooled vector
    // becomes one SEGate node (two GEMVs per image in one kernel instead of two implicit-GEMM launches).
    {
        const char* fe = getenv("OAR_FUSE_SE");
        const bool fuse = !fe || atoi(fe) != 0;
        auto cons = consumers(nodes);
        std::map<std::string, int> producer;
        for (int i = 0; i < (int)nodes.size(); ++i) for (auto& o : nodes[i].out) producer[o] = i;
        std::vector<bool> dead(nodes.size(), false);
        auto plain_1x1 = [&](const GNode& c) {
            if (c.op != "Conv" || c.in.size() < 2 || !is_init(c.in[1]) || !c.residual.empty() || c.ai("group", 1) != 1) return false;
            const HostTensor& w = inits_[c.in[1]];
            if (w.dims.size() != 4 || w.dims[2] != 1 || w.dims[3] != 1) return false;
            for (auto v : c.ais("strides")) if (v != 1) return false;
            for (auto v : c.ais("pads")) if (v != 0) return false;
            return c.as("auto_pad", "NOTSET") == "NOTSET" && (c.in.size() < 3 || c.in[2].empty() || is_init(c.in[2]));
        };
        for (int i = 0; fuse && i < (int)nodes.size(); ++i) {
            if (dead[i] || nodes[i].op != "GlobalAveragePool") continue;
            const std::string& pooled = nodes[i].out[0];
            if (cons[pooled].size() != 1 || graph_outs.count(pooled)) continue;
            const int a = cons[pooled][0];
            if (!plain_1x1(nodes[a]) || nodes[a].in[0] != pooled) continue;
            const std::string& mid = nodes[a].out[0];
            if (cons[mid].size() != 1 || graph_outs.count(mid)) continue;
            const int b = cons[mid][0];
            if (!plain_1x1(nodes[b]) || nodes[b].in[0] != mid) continue;
            if (inits_[nodes[b].in[1]].dims[1] != inits_[nodes[a].in[1]].dims[0]) continue;
            GNode g;
            g.op = "SEGate";
            g.in = {pooled, nodes[a].in[1], nodes[a].in.size() > 2 ? nodes[a].in[2] : std::string(), nodes[b].in[1], nodes[b].in.size() > 2 ? nodes[b].in[2] : std::string()};
            g.out = {nodes[b].out[0]};
            g.act = nodes[b].act;
            Attr k1; k1.kind = Attr::I; k1.i = nodes[a].act.kind; g.attrs["act1"] = k1;
            Attr al; al.kind = Attr::F; al.f = nodes[a].act.alpha; g.attrs["act1_alpha"] = al;
            Attr be; be.kind = Attr::F; be.f = nodes[a].act.beta; g.attrs["act1_beta"] = be;
            dead[a] = true;
            nodes[b] = std::move(g);
        }
        std::vector<GNode> keep;
        for (int i = 0; i < (int)nodes.size(); ++i) if (!dead[i]) keep.push_back(std::move(nodes[i]));
        nodes.swap(keep);
    }
    // ---- pass 7: depthwise-separable block: Conv (depthwise k x k, folded activation) -> Conv (1 x 1, folded activation /
    // residual) with nothing else reading the expanded tensor becomes ONE DSBlock node (csrc/dsblock.inc: the depthwise
    // output stays in LDS and feeds the matrix pipe).  Shapes are only known at plan time: op_dsblock falls back to the two
    // convolutions when dsblock_eligible() says no.  OAR_FUSE_DSBLOCK=0 keeps them apart.
    {
        const char* fe = getenv("OAR_FUSE_DSBLOCK");
        const bool fuse = !fe || atoi(fe) != 0;
        auto cons = consumers(nodes);
        std::vector<bool> dead(nodes.size(), false);
        auto ints = [](const GNode& c, const char* k, int64_t dflt) { auto v = c.ais(k); return v.empty() ? std::vector<int64_t>{dflt, dflt} : v; };
        for (int i = 0; fuse && i < (int)nodes.size(); ++i) {
            const GNode& d = nodes[i];
            if (dead[i] || d.op != "Conv" || d.in.size() < 2 || !is_init(d.in[1]) || !d.residual.empty()) continue;
            const HostTensor& wd = inits_[d.in[1]];
            if (wd.dims.size() != 4 || wd.dims[1] != 1 || wd.dims[2] != wd.dims[3] || (wd.dims[2] != 3 && wd.dims[2] != 5)) continue;
            if (d.ai("group", 1) != wd.dims[0] || wd.dims[0] < 8) continue;
            for (auto v : d.ais("dilations")) if (v != 1) goto next_node;
            if (d.as("auto_pad", "NOTSET") != "NOTSET" || (d.in.size() > 2 && !d.in[2].empty() && !is_init(d.in[2]))) continue;
            {
                const std::string& mid = d.out[0];
                if (graph_outs.count(mid) || cons[mid].size() != 1) continue;
                const int j = cons[mid][0];
                GNode& pw = nodes[j];
                if (dead[j] || pw.op != "Conv" || pw.in[0] != mid || pw.in.size() < 2 || !is_init(pw.in[1]) || pw.ai("group", 1) != 1) continue;
                const HostTensor& wp = inits_[pw.in[1]];
                if (wp.dims.size() != 4 || wp.dims[2] != 1 || wp.dims[3] != 1 || wp.dims[1] != wd.dims[0]) continue;
                bool plain = pw.as("auto_pad", "NOTSET") == "NOTSET" && (pw.in.size() < 3 || pw.in[2].empty() || is_init(pw.in[2]));
                for (auto v : ints(pw, "strides", 1)) plain = plain && v == 1;
                for (auto v : pw.ais("pads")) plain = plain && v == 0;
                if (!plain) continue;
                GNode f;
                f.op = "DSBlock";
                f.in = {d.in[0], d.in[1], d.in.size() > 2 ? d.in[2] : std::string(), pw.in[1], pw.in.size() > 2 ? pw.in[2] : std::string()};
                f.out = {pw.out[0]};
                f.attrs = d.attrs;                      // strides / pads / kernel_shape of the depthwise conv
                f.act = pw.act; f.residual = pw.residual;
                Attr k1; k1.kind = Attr::I; k1.i = d.act.kind; f.attrs["act1"] = k1;
                Attr al; al.kind = Attr::F; al.f = d.act.alpha; f.attrs["act1_alpha"] = al;
                Attr be; be.kind = Attr::F; be.f = d.act.beta; f.attrs["act1_beta"] = be;
                Attr mn; mn.kind = Attr::S; mn.s = mid; f.attrs["mid_name"] = mn;
                dead[i] = true;
                nodes[j] = std::move(f);   // at the pointwise conv's position: a folded residual is computed by then
            }
        next_node:;
        }
        std::vector<GNode> keep;
        for (int i = 0; i < (int)nodes.size(); ++i) if (!dead[i]) keep.push_back(std::move(nodes[i]));
        nodes.swap(keep);
    }
    // ---- pass 8: squeeze-excite scale into the pointwise conv behind it: Mul(x, SEGate(...)) whose only reader is a plain 1x1 Conv
    // becomes that Conv with the gate as a 4th input (csrc/igemm_ws_x6.hip multiplies it into the pixels as they are loaded: the scaled
    // feature map is never written or re-read).  Shapes / kernel choice are plan-time matters: op_conv runs the Mul after all when the
    // bf16x6 weight-stationary kernel does not take the layer.  OAR_FUSE_SE_SCALE=0 keeps the Mul.
    {
        const char* fe = getenv("OAR_FUSE_SE_SCALE");
        const bool fuse = !fe || atoi(fe) != 0;
        auto cons = consumers(nodes);
        std::map<std::string, int> producer;
        for (int i = 0; i < (int)nodes.size(); ++i) for (auto& o : nodes[i].out) producer[o] = i;
        std::vector<bool> dead(nodes.size(), false);
        for (int i = 0; fuse && i < (int)nodes.size(); ++i) {
            const GNode& m = nodes[i];
            if (m.op != "Mul" || m.in.size() != 2 || m.act.kind != k::ACT_NONE || !m.residual.empty() || graph_outs.count(m.out[0])) continue;
            int gi = -1;
            for (int t = 0; t < 2; ++t) { auto pit = producer.find(m.in[t]); if (pit != producer.end() && nodes[pit->second].op == "SEGate") gi = t; }
            if (gi < 0 || cons[m.out[0]].size() != 1) continue;
            GNode& c = nodes[cons[m.out[0]][0]];
            if (c.op != "Conv" || c.in.size() < 2 || c.in.size() > 3 || c.in[0] != m.out[0] || !is_init(c.in[1]) || c.ai("group", 1) != 1) continue;
            const HostTensor& w = inits_[c.in[1]];
            if (w.dims.size() != 4 || w.dims[2] != 1 || w.dims[3] != 1) continue;
            bool plain = c.as("auto_pad", "NOTSET") == "NOTSET";
            for (auto v : c.ais("strides")) plain = plain && v == 1;
            for (auto v : c.ais("pads")) plain = plain && v == 0;
            if (!plain) continue;
            c.in.resize(4);
            c.in[0] = m.in[1 - gi];
            c.in[3] = m.in[gi];
            dead[i] = true;
        }
        std::vector<GNode> keep;
        for (int i = 0; i < (int)nodes.size(); ++i) if (!dead[i]) keep.push_back(std::move(nodes[i]));
        nodes.swap(keep);
    }
    // ---- pass 9: the squeeze of a squeeze-excite block from the depthwise conv's own epilogue: GlobalAveragePool(dw conv output) is
    // dropped and its output becomes a second output of the conv node (conv_dw_tiled_kernel<..., GAP> writes per-tile sums,
    // global_avgpool_finish reduces them: the feature map is not read a second time).  op_conv pools separately when the launched
    // depthwise variant has no pooled form.  OAR_FUSE_SE_POOL=0 keeps the GlobalAveragePool.
    {
        const char* fe = getenv("OAR_FUSE_SE_POOL");
        const bool fuse = !fe || atoi(fe) != 0;
        std::map<std::string, int> producer;
        for (int i = 0; i < (int)nodes.size(); ++i) for (auto& o : nodes[i].out) producer[o] = i;
        std::vector<bool> dead(nodes.size(), false);
        for (int i = 0; fuse && i < (int)nodes.size(); ++i) {
            const GNode& gp = nodes[i];
            if (gp.op != "GlobalAveragePool" || gp.in.size() != 1 || graph_outs.count(gp.out[0])) continue;
            auto pit = producer.find(gp.in[0]);
            if (pit == producer.end() || pit->second > i) continue;
            GNode& d = nodes[pit->second];
            if (d.op != "Conv" || d.out.size() != 1 || d.in.size() < 2 || d.in.size() > 3 || !is_init(d.in[1]) || !d.residual.empty()) continue;
            const HostTensor& wd = inits_[d.in[1]];
            if (wd.dims.size() != 4 || wd.dims[1] != 1 || d.ai("group", 1) != wd.dims[0] || wd.dims[0] < 8) continue;   // depthwise
            d.out.push_back(gp.out[0]);
            dead[i] = true;
            // round 5: when the pooled vector only feeds a fused squeeze-excite gate the conv hands over its per-tile sums as they are and
            // se_fc reduces them (no global_avgpool_finish launch: 4 fewer 6 us kernels and kernel boundaries per recognition batch).
            // OAR_FUSE_SE_POOL=1 keeps the finishing launch.
            int readers = 0, gate = -1;
            for (int j = 0; j < (int)nodes.size(); ++j) {
                if (dead[j]) continue;
                for (size_t q = 0; q < nodes[j].in.size(); ++q) if (nodes[j].in[q] == gp.out[0]) { ++readers; if (nodes[j].op == "SEGate" && q == 0) gate = j; }
                if (nodes[j].residual == gp.out[0]) ++readers;
            }
            // (a pooled vector that is ALSO a graph output must exist as [N, C, 1, 1] means, not as raw tile sums: ADVICE r5)
            if (readers == 1 && gate >= 0 && !graph_outs.count(gp.out[0]) && !(fe && atoi(fe) == 1)) { Attr a; a.kind = Attr::I; a.i = 1; d.attrs["gap_raw"] = a; }
        }
        std::vector<GNode> keep;
        for (int i = 0; i < (int)nodes.size(); ++i) if (!dead[i]) keep.push_back(std::move(nodes[i]));
        nodes.swap(keep);
    }
    for (int i = 0; i < (int)nodes.size(); ++i) nodes[i].id = i;
    nodes_ = std::move(nodes);
}

// =================================================================================================
// planner
// =================================================================================================
struct TInfo {
    std::vector<int64_t> dims;  // logical
    Layout layout = Layout::NATIVE;
    Loc loc;
    bool host_int = false;           // value known at plan time (shape arithmetic): lives on the host, never in HBM
    std::vector<int64_t> hv;         //   as integers (floats truncated toward zero)
    bool host_f = false;             //   the host value is floating point: hd is authoritative
    std::vector<double> hd;
    const HostTensor* ht = nullptr;  // f32 initializer
    bool is_int = false;             // device tensor whose f32 values are integers by construction (ArgMax, integer Cast of one)
    bool u8_stem = false;            // the graph input of a run_stem plan: u8 pages read by the fused stem, no f32 tensor behind it
    std::string root;                // storage root (for liveness)
    int gap_tiles = 0, gap_hw = 0;   // [n, tiles * C, 1, 1]: per-tile channel sums of a pooling depthwise conv, still to be reduced and divided by gap_hw (SEGate does)
    size_t bytes() const { return (size_t)std::max<int64_t>(numel(dims), 1) * 4; }
};

struct Arena {
    struct Blk { size_t off, sz; };
    std::vector<Blk> free_;
    size_t top = 0;
    static size_t al(size_t n) { return (n + 255) & ~(size_t)255; }
    size_t alloc(size_t n) {
        n = al(std::max<size_t>(n, 4));
        int best = -1;
        for (int i = 0; i < (int)free_.size(); ++i)
            if (free_[i].sz >= n && (best < 0 || free_[i].sz < free_[best].sz)) best = i;
        if (best >= 0) {
            size_t off = free_[best].off;
            if (free_[best].sz == n) free_.erase(free_.begin() + best);
            else { free_[best].off += n; free_[best].sz -= n; }
            return off;
        }
        size_t off = top;
        top += n;
        return off;
    }
    void release(size_t off, size_t n) {
        n = al(std::max<size_t>(n, 4));
        free_.push_back({off, n});
        std::sort(free_.begin(), free_.end(), [](const Blk& a, const Blk& b) { return a.off < b.off; });
        std::vector<Blk> m;
        for (auto& b : free_) {
            if (!m.empty() && m.back().off + m.back().sz == b.off) m.back().sz += b.sz;
            else m.push_back(b);
        }
        if (!m.empty() && m.back().off + m.back().sz == top) { top = m.back().off; m.pop_back(); }
        free_.swap(m);
    }
};

static std::vector<int64_t> clast_phys_dims(const std::vector<int64_t>& d) {  // [n,C,s...] -> [n,s...,C]
    std::vector<int64_t> p;
    p.push_back(d[0]);
    for (size_t i = 2; i < d.size(); ++i) p.push_back(d[i]);
    p.push_back(d[1]);
    return p;
}
static std::vector<int64_t> contig_strides(const std::vector<int64_t>& d) {
    std::vector<int64_t> s(d.size(), 1);
    for (int i = (int)d.size() - 2; i >= 0; --i) s[i] = s[i + 1] * d[i + 1];
    return s;
}

struct Planner {
    Engine& E;
    Plan& P;
    std::map<std::string, TInfo> vals;
    std::map<std::string, int> last_use;      // by root name
    std::map<std::string, size_t> root_bytes; // arena-resident roots
    std::map<std::string, size_t> root_off;
    Arena arena;
    int cur = 0;
    std::vector<std::pair<size_t, size_t>> temps;  // freed after the current node

    Planner(Engine& e, Plan& p) : E(e), P(p) { opset = e.opset_; }

    // ------------------------------------------------------------------ deferred nearest Resize
    // A Resize with exactly one consumer is not run where it stands: its consumer either absorbs it (Concat on the channel axis:
    // the resize writes straight into its slot of the concatenated tensor; Add: the sum reads the low-resolution operand through
    // the index map) or, failing that, runs it first.  compute_last_use keeps the resize's input alive until that consumer.
    struct PendingResize {
        Loc xin;
        std::vector<int64_t> dims;   // logical output dims [N, C, Ho, Wo]
        int N, H, W, C, Ho, Wo, imode, ictm, inm;
        float sh, sw;
        int fh = 0, fw = 0;          // > 0: the index map is exactly o / f (integer-factor nearest upsampling)
    };
    std::map<std::string, PendingResize> pending_resize;
    std::map<std::string, int> sole_consumer;   // Resize output -> index of its only consumer node
    void run_resize(const PendingResize& r, Loc yl, int64_t y_off, int y_ld) {
        const PendingResize q = r;
        step([=](const RunCtx& c) { k::resize(c.s, c.at(q.xin), c.mut(yl) + y_off, q.N, q.H, q.W, q.C, q.Ho, q.Wo, q.sh, q.sw, q.imode, q.ictm, q.inm, y_ld); }, 0,
             4.0 * ((double)q.N * q.H * q.W * q.C + (double)q.N * q.Ho * q.Wo * q.C));
    }
    // A 2x2 / stride-2 ConvTranspose whose only consumer is another one (the DB head's tail) is held back the same way: the
    // consumer runs both as k::convt2x2_pair, anything else runs it first.
    std::map<std::string, const GNode*> pending_convt;
    // Round 6: a plain channel concat of channels-last maps whose ONLY reader is a 1x1 convolution the output-stationary bf16x6 kernel can take (PP-HGNetV2's
    // aggregation: seven maps, up to 3328 channels) is not written at all: the convolution reads its K dimension out of the sources (ConvP::msrc).  Anything
    // else that asks for the value first gets the gather launch.  compute_last_use keeps the sources alive until that reader.
    struct PendingConcat { std::vector<Loc> src; std::vector<int> c; std::vector<int64_t> dims; };
    std::map<std::string, PendingConcat> pending_concat;
    std::map<std::string, int> concat_consumer;   // Concat output -> index of its only consumer node
    void emit_concat_gather(const PendingConcat& pc, Loc yl) {
        k::ConcatGatherP gp{};
        gp.n_src = (int)pc.src.size(); gp.N = (int)pc.dims[0]; gp.Ho = (int)pc.dims[2]; gp.Wo = (int)pc.dims[3]; gp.C = (int)pc.dims[1];
        int off = 0;
        for (size_t i = 0; i < pc.src.size(); ++i) { gp.c[i] = pc.c[i]; gp.off[i] = off; gp.fh[i] = 1; gp.fw[i] = 1; off += pc.c[i]; }
        const std::vector<Loc> src = pc.src;
        step([=](const RunCtx& c) {
            k::ConcatGatherP q = gp;
            for (int i = 0; i < q.n_src; ++i) q.x[i] = c.at(src[(size_t)i]);
            k::concat_gather(c.s, q, c.mut(yl));
        }, 0, 8.0 * (double)numel(pc.dims));
    }
    void materialise_concat(const std::string& name) {
        auto it = pending_concat.find(name);
        if (it == pending_concat.end()) return;
        const PendingConcat pc = it->second;
        pending_concat.erase(it);
        vals.erase(name);
        TInfo& y = new_out(name, pc.dims, Layout::CLAST);
        emit_concat_gather(pc, y.loc);
    }
    const PendingResize* peek_pending(const std::string& name) const {
        auto it = pending_resize.find(name);
        return it == pending_resize.end() ? nullptr : &it->second;
    }
    void materialise_pending(const std::string& name) {
        auto ct = pending_convt.find(name);
        if (ct != pending_convt.end()) {
            const GNode* node = ct->second;
            pending_convt.erase(ct);
            vals.erase(name);
            op_convt(*node, /*may_defer=*/false);
            return;
        }
        auto it = pending_resize.find(name);
        if (it == pending_resize.end()) return;
        const PendingResize r = it->second;
        pending_resize.erase(it);
        TInfo& y = new_out(name, r.dims, Layout::CLAST);
        run_resize(r, y.loc, 0, r.C);
    }

    // ------------------------------------------------------------------ values
    TInfo& get(const std::string& name) {
        if (!pending_resize.empty() || !pending_convt.empty()) materialise_pending(name);
        if (!pending_concat.empty()) materialise_concat(name);
        auto it = vals.find(name);
        if (it != vals.end()) return it->second;
        auto ii = E.inits_.find(name);
        OAR_CHECK(ii != E.inits_.end(), OAR_MODEL_LOAD, "graph references unknown value '" + name + "'");
        TInfo t;
        t.dims = ii->second.dims;
        if (ii->second.dtype == DType::F32) {
            t.ht = &ii->second;
            t.loc.kind = Loc::CONST;
            t.loc.cptr = E.upload_const("init:" + name, ii->second.f);
        } else {
            t.host_int = true;
            t.hv = ii->second.i;
        }
        return vals[name] = t;
    }
    bool has_input(const GNode& n, size_t i) const { return i < n.in.size() && !n.in[i].empty(); }

    Loc alloc_arena(size_t bytes, const std::string& root) {
        Loc l;
        l.kind = Loc::ARENA;
        l.off = (int64_t)arena.alloc(bytes);
        P.arena_bytes = std::max(P.arena_bytes, arena.top);
        if (!root.empty()) { root_bytes[root] = bytes; root_off[root] = (size_t)l.off; }
        return l;
    }
    Loc alloc_temp(size_t bytes) {
        Loc l = alloc_arena(bytes, "");
        temps.push_back({(size_t)l.off, bytes});
        return l;
    }
    TInfo& new_out(const std::string& name, const std::vector<int64_t>& dims, Layout lay) {
        TInfo t;
        t.dims = dims; t.layout = lay; t.root = name;
        t.loc = alloc_arena(t.bytes(), name);
        return vals[name] = t;
    }
    // A node planned as several operators leaves its result as a VIEW of a temporary (e.g. pool -> "out::kept" -> Squeeze -> "out"): the arena bytes are then booked
    // under the temporary's name, which compute_last_use() knows nothing about -- release_dead() would free them right after the node although "out" is read later.
    // The node's output takes the booking over (round 6; tools/op_fuzz.py found a reduce whose result was overwritten by the next node's output).
    void adopt_root(const std::string& out, const std::string& tmp) {
        auto it = vals.find(out);
        if (it == vals.end() || tmp == out || it->second.root != tmp) return;
        auto rb = root_bytes.find(tmp);
        if (rb != root_bytes.end()) { root_bytes[out] = rb->second; root_off[out] = root_off[tmp]; root_bytes.erase(tmp); root_off.erase(tmp); }
        it->second.root = out;
    }
    TInfo& alias_out(const std::string& name, const TInfo& src, const std::vector<int64_t>& dims, Layout lay, int64_t byte_off = 0) {
        TInfo t;
        t.dims = dims; t.layout = lay; t.root = src.root; t.loc = src.loc; t.ht = nullptr;
        if (t.loc.kind == Loc::CONST) t.loc.cptr = (const float*)((const char*)t.loc.cptr + byte_off);
        else t.loc.off += byte_off;
        return vals[name] = t;
    }
    void step(std::function<void(const RunCtx&)> f, double flops = 0, double bytes = 0) {
        P.steps.push_back(std::move(f));
        P.flops += flops; P.bytes += bytes; P.n_kernels++;
        step_chain.push_back(-1);
        step_node.push_back(cur);
        static const bool dbg_steps = [] { const char* e = getenv("OAR_DEBUG_STEPS"); return e && atoi(e) != 0; }();
        if (dbg_steps) {
            const GNode* g = cur >= 0 && cur < (int)E.nodes_.size() ? &E.nodes_[(size_t)cur] : nullptr;
            P.step_ops.push_back(g ? g->op + " -> " + (g->out.empty() ? std::string("?") : g->out[0]) : std::string("?"));
        }
    }

    // ------------------------------------------------------------------ sample-local chains (csrc/chain.hip)
    // Operators that only read rows of their own sample (1 x k convolutions over H == 1 maps, Linear, LayerNorm, the fused SVTR
    // attention, row copies of a channel concat) record a ChainRec next to their step.  While the newest step is such an operator,
    // build() releases nothing (every tensor a run touches stays allocated until the run ends, so no two of them share arena bytes
    // and the samples of a fused run may advance at different paces); fuse_chains() then replaces each run by ONE chain_run step.
    struct ChainRec {
        k::ChainOpD d;
        Loc in, out, res;
        int64_t rows = 0;                 // rows of the operator's tensors (all samples)
        int64_t fix_n = 0, fix_T = 0;     // > 0: the operator fixes the partition (n samples of T rows); 0: any partition of its rows
        std::function<std::vector<float>()> make_w;   // GEMM weights as [N][K] f32 (host), built only when a chain is really formed
        const float* dev_bias = nullptr;  // GEMM bias / LN beta (device constant of N floats)
        const float* dev_gamma = nullptr; // LN gamma
        double flops = 0, bytes = 0;
        int node = 0;                     // graph node the operator belongs to
        std::string out_root;             // storage root of the tensor it writes (liveness: is it read after the run?)
    };
    std::vector<int> step_chain;          // parallel to P.steps: index into chain_recs, -1 = not chainable
    std::vector<int> step_node;           // parallel to P.steps: the graph node that emitted the step
    std::vector<ChainRec> chain_recs;
    bool chain_on = [] { const char* e = getenv("OAR_FUSE_CHAIN"); return !e || atoi(e) != 0; }();
    static bool chain_loc_ok(const Loc& l) { return l.kind == Loc::ARENA || l.kind == Loc::INPUT || (l.kind == Loc::CONST && l.cptr); }
    static k::ChainRef chain_ref(const Loc& l) {
        k::ChainRef r;
        if (l.kind == Loc::ARENA) { r.kind = 1; r.v = (unsigned long long)l.off; }
        else if (l.kind == Loc::INPUT) { r.kind = 2; r.v = (unsigned long long)l.off; }
        else if (l.kind == Loc::CONST) { r.kind = 0; r.v = (unsigned long long)reinterpret_cast<uintptr_t>(l.cptr); }
        return r;
    }
    void step_chainable(std::function<void(const RunCtx&)> f, ChainRec rec, double flops, double bytes) {
        step(std::move(f), flops, bytes);
        if (!chain_on) return;
        rec.flops = flops; rec.bytes = bytes; rec.node = cur;
        step_chain.back() = (int)chain_recs.size();
        chain_recs.push_back(std::move(rec));
    }
    bool chain_run_open() const { return chain_on && !step_chain.empty() && step_chain.back() >= 0; }
    static std::vector<float> chain_weight_conv(const HostTensor& W) {   // [Cout][Cin][1][kw] -> [Cout][kw * Cin]
        const int64_t Co = W.dims[0], Ci = W.dims[1], kw = W.dims[3], K = kw * Ci;
        std::vector<float> w((size_t)Co * K);
        for (int64_t co = 0; co < Co; ++co)
            for (int64_t ci = 0; ci < Ci; ++ci)
                for (int64_t b = 0; b < kw; ++b) w[(size_t)co * K + b * Ci + ci] = W.f[(size_t)((co * Ci + ci) * kw + b)];
        return w;
    }
    static std::vector<float> chain_weight_linear(const HostTensor& B, bool transB) {   // -> [N][K]
        const int64_t K = transB ? B.dims[1] : B.dims[0], N = transB ? B.dims[0] : B.dims[1];
        std::vector<float> w((size_t)N * K);
        for (int64_t kk = 0; kk < K; ++kk)
            for (int64_t nn = 0; nn < N; ++nn) w[(size_t)nn * K + kk] = transB ? B.f[(size_t)(nn * K + kk)] : B.f[(size_t)(kk * N + nn)];
        return w;
    }
    static constexpr int kMinChain = 3;
    bool chain_live_after(const std::string& root, int last_node) const {   // is the tensor read by a node after `last_node` (or a graph output)?
        if (root == "\x01stage") return false;   // a staging copy of fuse_chains: lives inside the run by construction
        auto it = last_use.find(root);
        return root.empty() || it == last_use.end() || it->second > last_node;
    }
    // Tensors a chain writes and nobody reads after it live in the workgroup's LDS (one sample's rows, each row padded by 4 floats so
    // that the 16 rows of a B operand start in different banks): the operators' references are rewritten to LDS offsets (ChainRef
    // kind 3) and their row strides to the padded one.  Liveness-based first-fit over the operator order; when the budget is
    // exceeded the largest tensor goes back to the arena and the placement is redone.  Returns the dynamic LDS size of the launch.
    // Split-K partials live at LDS offset 0, but only DURING their operator: `op_scratch[j]` bytes are reserved for operator j alone (a pseudo tensor
    // pinned at offset 0 with the lifetime [j, j]), so a tensor that is dead by then -- or born later -- may occupy the same bytes (round 5: with the pooling
    // operator inside the chain its 40 KB output pushed the 80 KB concat back to the arena while 24 KB of partials that no operator near it uses sat idle).
    size_t chain_place_in_lds(std::vector<k::ChainOpD>& ops, const std::vector<std::string>& out_roots, int last_node, int64_t n, int64_t T, const std::vector<size_t>& op_scratch, size_t budget) {
        size_t scratch = 0;
        for (size_t b : op_scratch) scratch = std::max(scratch, b);
        const char* lds_env = getenv("OAR_CHAIN_LDS");   // read per plan: tests A/B the placements inside one process
        const bool lds_on = !lds_env || atoi(lds_env) != 0;
        if (!lds_on || scratch >= budget) return scratch;
        struct Use { int op; int which; int64_t off; int ld; int width; };   // which: 0 in, 1 out, 2 res; width: columns the view covers
        std::vector<Use> uses;
        for (int j = 0; j < (int)ops.size(); ++j) {
            const k::ChainOpD& d = ops[(size_t)j];
            const int win = d.type == k::CH_GEMM ? d.cin : d.type == k::CH_ATTN ? 3 * d.heads * d.hd : d.N;
            if (d.in.kind == 1 && d.type != k::CH_POOL) uses.push_back({j, 0, (int64_t)d.in.v, d.in_ld, win});   // (a pool reads the sample's whole map from HBM: not a [n T][ld] tensor of the run)
            if (d.out.kind == 1) uses.push_back({j, 1, (int64_t)d.out.v, d.out_ld, d.N});
            if (d.res.kind == 1) uses.push_back({j, 2, (int64_t)d.res.v, d.res_ld, d.N});
        }
        struct Root { int64_t base, end; int ld; bool bad = false, written = false, live_out = false, in_lds = false; int first = 1 << 30, last = -1; size_t bytes = 0, off = 0; };
        std::vector<Root> roots;
        std::vector<int> order(uses.size());
        for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
        std::sort(order.begin(), order.end(), [&](int a, int b) { return uses[(size_t)a].off < uses[(size_t)b].off; });
        std::vector<int> root_of(uses.size(), -1);
        for (int ui : order) {
            const Use& u = uses[(size_t)ui];
            const int64_t end = u.off + ((n * T - 1) * u.ld + u.width) * 4;
            if (roots.empty() || u.off >= roots.back().end) { Root r; r.base = u.off; r.end = end; r.ld = u.ld; roots.push_back(r); }
            Root& r = roots.back();
            r.end = std::max(r.end, end);
            if (u.ld != r.ld || u.off - r.base >= (int64_t)r.ld * 4) r.bad = true;
            root_of[(size_t)ui] = (int)roots.size() - 1;
        }
        for (size_t ui = 0; ui < uses.size(); ++ui) {
            Root& r = roots[(size_t)root_of[ui]];
            const Use& u = uses[ui];
            r.first = std::min(r.first, u.op); r.last = std::max(r.last, u.op);
            if (u.which == 1) {
                r.written = true;
                if (chain_live_after(out_roots[(size_t)u.op], last_node)) r.live_out = true;
            }
        }
        for (Root& r : roots) {
            r.bytes = (size_t)T * (size_t)(r.ld + 4) * 4;
            r.in_lds = r.written && !r.live_out && !r.bad;
            // a tensor written in the run must be written before it is read there (it is: the planner emits in dependency order)
        }
        size_t peak = 0;
        for (;;) {
            // first-fit by first definition; a slot is reusable by an operator's OUTPUT only when its tensor's last use is an earlier operator.
            // Candidates a tensor must avoid: every tensor placed before it whose life reaches its first operator, and the partials of every
            // operator inside its own life.
            std::vector<int> idx;
            for (int i = 0; i < (int)roots.size(); ++i) if (roots[(size_t)i].in_lds) idx.push_back(i);
            std::sort(idx.begin(), idx.end(), [&](int a, int b) { return roots[(size_t)a].first < roots[(size_t)b].first; });
            std::vector<int> placed;
            peak = scratch;
            for (int i : idx) {
                Root& r = roots[(size_t)i];
                std::vector<std::pair<size_t, size_t>> busy;   // [begin, end) byte ranges taken while r lives
                for (int a : placed) if (roots[(size_t)a].last >= r.first) busy.push_back({roots[(size_t)a].off, roots[(size_t)a].off + roots[(size_t)a].bytes});
                size_t part = 0;
                for (int j = std::max(0, r.first); j <= r.last && j < (int)op_scratch.size(); ++j) part = std::max(part, op_scratch[(size_t)j]);
                if (part) busy.push_back({0, part});
                std::sort(busy.begin(), busy.end());
                size_t off = 0;
                for (auto& b : busy) { if (off + r.bytes <= b.first) break; off = std::max(off, b.second); }
                r.off = off;
                peak = std::max(peak, off + r.bytes);
                placed.push_back(i);
            }
            if (peak <= budget || idx.empty()) break;
            int big = idx[0];
            for (int i : idx) if (roots[(size_t)i].bytes > roots[(size_t)big].bytes) big = i;
            roots[(size_t)big].in_lds = false;
        }
        for (size_t ui = 0; ui < uses.size(); ++ui) {
            const Root& r = roots[(size_t)root_of[ui]];
            if (!r.in_lds) continue;
            const Use& u = uses[ui];
            k::ChainOpD& d = ops[(size_t)u.op];
            k::ChainRef ref; ref.kind = 3; ref.v = (unsigned long long)(r.off + (size_t)(u.off - r.base));
            if (u.which == 0) { d.in = ref; d.in_ld = r.ld + 4; }
            else if (u.which == 1) { d.out = ref; d.out_ld = r.ld + 4; }
            else { d.res = ref; d.res_ld = r.ld + 4; }
        }
        return peak;
    }
    void fuse_chains() {
        if (!chain_on || chain_recs.empty()) return;
        const int ns = (int)P.steps.size();
        std::vector<std::function<void(const RunCtx&)>> out;
        int fused_away = 0;
        int a = 0;
        std::vector<std::string> out_ops;
        const bool names = P.step_ops.size() == P.steps.size();
        auto keep = [&](int i) { out.push_back(std::move(P.steps[(size_t)i])); if (names) out_ops.push_back(P.step_ops[(size_t)i]); };
        while (a < ns) {
            if (step_chain[(size_t)a] < 0) { keep(a++); continue; }
            int b = a;
            while (b < ns && step_chain[(size_t)b] >= 0) ++b;
            // run [a, b): chains grow around each operator that fixes the partition
            int i = a;
            while (i < b) {
                int kf = i;
                while (kf < b && chain_recs[(size_t)step_chain[(size_t)kf]].fix_n == 0) ++kf;
                if (kf == b) break;
                const int64_t n = chain_recs[(size_t)step_chain[(size_t)kf]].fix_n, T = chain_recs[(size_t)step_chain[(size_t)kf]].fix_T;
                auto compat = [&](int j) {
                    const ChainRec& r = chain_recs[(size_t)step_chain[(size_t)j]];
                    return r.fix_n ? (r.fix_n == n && r.fix_T == T) : (r.rows == n * T);
                };
                if (n <= 0 || T <= 0 || T > 4096 || !compat(kf)) { for (int j = i; j <= kf; ++j) keep(j); i = kf + 1; continue; }
                int lo = kf, hi = kf + 1;
                while (lo > i && compat(lo - 1)) --lo;
                while (hi < b && compat(hi)) ++hi;
                for (int j = i; j < lo; ++j) keep(j);
                // one workgroup (one CU's f32 matrix pipe: 0.6 TFLOP/s) per sample: beyond ~32 MFLOP per sample the operator-by-operator
                // kernels, which spread every product over the whole chip, win again (a server-size SVTR neck is ~120 MFLOP per line)
                double run_flops = 0;
                for (int j = lo; j < hi; ++j) run_flops += chain_recs[(size_t)step_chain[(size_t)j]].flops;
                const char* mf_env = getenv("OAR_CHAIN_MAX_MFLOP");   // (read per plan: the tests lift the cap for their long-line cases)
                const double max_mflop = mf_env ? atof(mf_env) : 32.0;
                if (hi - lo < kMinChain || run_flops / (double)n > max_mflop * 1e6) { for (int j = lo; j < hi; ++j) keep(j); i = hi; continue; }
                std::vector<k::ChainOpD> ops;
                std::vector<std::string> out_roots;
                std::vector<const ChainRec*> recs;
                double flops = 0, bytes = 0;
                int max_hd = 0;
                int last_node = chain_recs[(size_t)step_chain[(size_t)(hi - 1)]].node;
                if ((size_t)hi < step_node.size() && step_node[(size_t)hi] == last_node) --last_node;   // the run was cut inside a node: that node's tensors stay in the arena
                for (int j = lo; j < hi; ++j) {
                    ChainRec& r = chain_recs[(size_t)step_chain[(size_t)j]];
                    k::ChainOpD d = r.d;
                    d.in = chain_ref(r.in); d.out = chain_ref(r.out); d.res = chain_ref(r.res);
                    if (d.type == k::CH_ATTN) max_hd = std::max(max_hd, d.hd);
                    flops += r.flops; bytes += r.bytes;
                    ops.push_back(d); out_roots.push_back(r.out_root); recs.push_back(&r);
                }
                // copy elision: a Concat input produced inside the run by one operator and read by nothing else is written where the copy
                // would have put it (the producer gets the copy's destination view) and the copy disappears
                for (int j = 0; j < (int)ops.size();) {
                    const k::ChainOpD& cp = ops[(size_t)j];
                    int prod = -1, readers = 0;
                    if (cp.type == k::CH_COPY && cp.in.kind == 1) {
                        for (int q = 0; q < (int)ops.size(); ++q) {
                            const k::ChainOpD& o = ops[(size_t)q];
                            if (q < j && o.out.kind == 1 && o.out.v == cp.in.v && o.out_ld == cp.in_ld && o.N == cp.N && o.type != k::CH_COPY) prod = q;
                            if (q != j && ((o.in.kind == 1 && o.in.v == cp.in.v) || (o.res.kind == 1 && o.res.v == cp.in.v))) ++readers;
                        }
                    }
                    if (prod >= 0 && readers == 0 && !chain_live_after(out_roots[(size_t)prod], last_node)) {
                        ops[(size_t)prod].out = cp.out; ops[(size_t)prod].out_ld = cp.out_ld;
                        out_roots[(size_t)prod] = out_roots[(size_t)j];
                        ops.erase(ops.begin() + j); out_roots.erase(out_roots.begin() + j); recs.erase(recs.begin() + j);
                        continue;
                    }
                    ++j;
                }
                // the chain's constants in one allocation: biases / LayerNorm affine vectors first (the kernel keeps them in LDS), then weights
                std::vector<float> blob;
                auto fetch = [&](const float* dev, int cnt) {   // a device constant back to the host (plan time only)
                    const int at = (int)blob.size();
                    blob.resize((size_t)at + (size_t)((cnt + 3) & ~3), 0.f);
                    OAR_HIP(hipMemcpy(blob.data() + at, dev, (size_t)cnt * 4, hipMemcpyDeviceToHost));
                    return at;
                };
                for (size_t q = 0; q < ops.size(); ++q) {
                    if (recs[q]->dev_bias && ops[q].type != k::CH_COPY) ops[q].bias_l = fetch(recs[q]->dev_bias, ops[q].N);
                    if (recs[q]->dev_gamma && ops[q].type == k::CH_LN) ops[q].w_l = fetch(recs[q]->dev_gamma, ops[q].N);
                }
                const int small = (int)blob.size();
                // a product whose tokens come from HBM (the run's input) gets them staged into LDS first: a row copy in front of it, its
                // destination a transient LDS tensor (a made-up arena address far above the real arena names it for the placement below)
                const unsigned long long kStageBase = 1ull << 50;
                for (size_t q = 0; q < ops.size(); ++q) {
                    if (ops[q].type != k::CH_GEMM || (ops[q].in.kind != 1 && ops[q].in.kind != 2)) continue;
                    bool produced_here = false;   // (tensors written earlier in the run are placed in LDS as they are)
                    for (size_t e = 0; e < q && ops[q].in.kind == 1; ++e) {
                        if (ops[e].out.kind != 1) continue;
                        const unsigned long long a0 = ops[q].in.v, a1 = a0 + (unsigned long long)(((n * T - 1) * ops[q].in_ld + ops[q].cin) * 4);
                        const unsigned long long b0 = ops[e].out.v, b1 = b0 + (unsigned long long)(((n * T - 1) * ops[e].out_ld + ops[e].N) * 4);
                        produced_here = produced_here || (a0 < b1 && b0 < a1);
                    }
                    if (produced_here) continue;
                    k::ChainOpD cp;
                    cp.type = k::CH_COPY; cp.N = ops[q].in_ld; cp.in_ld = ops[q].in_ld; cp.out_ld = ops[q].in_ld;
                    cp.in = ops[q].in;
                    cp.out.kind = 1; cp.out.v = kStageBase + ((unsigned long long)q << 40);
                    ops[q].in = cp.out;
                    ops.insert(ops.begin() + (long)q, cp); out_roots.insert(out_roots.begin() + (long)q, std::string("\x01stage")); recs.insert(recs.begin() + (long)q, nullptr);
                    ++q;
                }
                const int MT = (int)((T + 15) / 16);
                std::vector<size_t> w_at(ops.size(), 0);
                for (size_t q = 0; q < ops.size(); ++q) {
                    k::ChainOpD& d = ops[q];
                    if (d.type != k::CH_GEMM) continue;
                    blob.resize((blob.size() + 63) & ~(size_t)63, 0.f);
                    w_at[q] = blob.size();
                    const std::vector<float> w = recs[q]->make_w();
                    OAR_CHECK(w.size() == (size_t)d.N * (size_t)d.K, OAR_INTERNAL, "chain: weight size");
                    blob.insert(blob.end(), w.begin(), w.end());
                    // work items = channel tiles x token groups (mb tiles each) x K slices: enough of them for the 16 waves, partials <= 24 KB
                    // narrow products whose 16 x 16 tiles fit one round of the 16 waves take one tile per item (shortest dependent chain);
                    // the others share each weight load between up to 3 token tiles
                    const int NT = d.N / 16, KB = d.K / 16;
                    d.mb = (NT * MT <= 16 && KB <= 8) ? 1 : (MT <= 3 ? MT : (MT % 3 == 0 ? 3 : (MT % 2 == 0 ? 2 : 3)));
                    const int MG = (MT + d.mb - 1) / d.mb;
                    // (a 48 KB cap -- 16 items instead of 8 for the two long products K = 768 / 1536 -> 32 -- was measured: no faster per product, and the
                    // partials then push the 80 KB concat out of LDS: profiles/r5/chain_r5_endgame.txt)
                    int ks = 1;
                    while (NT * MG * ks < 16 && KB / (ks * 2) >= 2 && NT * MG * d.mb * ks * 2 <= 24) ks *= 2;
                    d.ksplit = ks;
                }
                std::vector<size_t> op_scratch(ops.size(), 0);
                for (size_t q = 0; q < ops.size(); ++q) op_scratch[q] = (k::chain_lds_bytes(ops[q], (int)T) + 255) & ~(size_t)255;
                auto consts = std::make_shared<DevBuf>();
                consts->reserve(blob.size() * 4 + 256);
                OAR_HIP(hipMemcpy(consts->p, blob.data(), blob.size() * 4, hipMemcpyHostToDevice));
                for (size_t q = 0; q < ops.size(); ++q) if (ops[q].type == k::CH_GEMM) ops[q].w = consts->as<float>() + w_at[q];
                const size_t small_bytes = (((size_t)small * 4 + 255) & ~(size_t)255) + ((ops.size() * sizeof(k::ChainOpD) + 255) & ~(size_t)255);   // + the table copy
                const size_t tens_end = chain_place_in_lds(ops, out_roots, last_node, n, T, op_scratch, k::kChainLdsBudget - small_bytes);
                for (size_t q = 0; q + 1 < ops.size();) {   // a staging copy that found no room: drop it, the product reads HBM (slow path)
                    if (ops[q].type == k::CH_COPY && ops[q].out.kind == 1 && ops[q].out.v >= kStageBase) {
                        ops[q + 1].in = ops[q].in; ops[q + 1].in_ld = ops[q].in_ld;
                        ops.erase(ops.begin() + (long)q); out_roots.erase(out_roots.begin() + (long)q); recs.erase(recs.begin() + (long)q); w_at.erase(w_at.begin() + (long)q);
                        continue;
                    }
                    ++q;
                }
                const int small_l = (int)(((tens_end + 255) & ~(size_t)255) / 4);
                for (auto& d : ops) { if (d.bias_l >= 0) d.bias_l += small_l; if (d.w_l >= 0) d.w_l += small_l; }
                auto table = std::make_shared<DevBuf>();
                table->reserve(ops.size() * sizeof(k::ChainOpD));
                OAR_HIP(hipMemcpy(table->p, ops.data(), ops.size() * sizeof(k::ChainOpD), hipMemcpyHostToDevice));
                k::ChainLaunch L;
                L.ops = table->as<k::ChainOpD>(); L.n_ops = (int)ops.size(); L.n_samples = (int)n; L.T = (int)T; L.max_hd = max_hd;
                L.lds = (size_t)small_l * 4 + small_bytes;
                L.tab_l = small_l + (int)((((size_t)small * 4 + 255) & ~(size_t)255) / 4);
                L.consts = consts->as<float>(); L.small = small; L.total_consts = (int)blob.size(); L.small_l = small_l;
                L.bytes = bytes; L.flops = flops;
                const int n_ops = hi - lo;
                out.push_back([L, table, consts](const RunCtx& c) { k::chain_run(c.s, L, c.arena, reinterpret_cast<const char*>(c.input)); });
                if (names) out_ops.push_back("chain of " + std::to_string(hi - lo) + " steps from " + P.step_ops[(size_t)lo]);
                fused_away += n_ops - 1;
                i = hi;
            }
            for (int j = i; j < b; ++j) keep(j);
            a = b;
        }
        P.steps.swap(out);
        if (names) P.step_ops.swap(out_ops);
        P.n_kernels -= fused_away;
        step_chain.assign(P.steps.size(), -1);
        step_node.assign(P.steps.size(), -1);
    }

    // ------------------------------------------------------------------ layout conversions
    Loc to_clast_loc(const TInfo& t) {  // native [n,C,s..] -> channels-last copy in a temp
        if (t.layout == Layout::CLAST) return t.loc;
        OAR_CHECK(t.dims.size() >= 3 && t.dims.size() <= 5, OAR_UNSUPPORTED_OP, "channels-last conversion needs rank 3..5");
        int r = (int)t.dims.size();
        bool trivial = t.dims[1] == 1;
        int64_t sp = 1;
        for (int i = 2; i < r; ++i) sp *= t.dims[i];
        if (trivial || sp == 1) return t.loc;
        Loc tmp = alloc_temp(t.bytes());
        std::vector<int64_t> od = clast_phys_dims(t.dims);
        std::vector<int64_t> ns = contig_strides(t.dims), is(r);
        is[0] = ns[0];
        for (int i = 2; i < r; ++i) is[i - 1] = ns[i];
        is[r - 1] = ns[1];
        Loc src = t.loc;
        step([=](const RunCtx& c) { k::permute(c.s, c.at(src), c.mut(tmp), r, od.data(), is.data()); }, 0, 2.0 * t.bytes());
        return tmp;
    }
    Loc to_native_loc(const TInfo& t, Loc dst = Loc()) {
        bool need = t.layout == Layout::CLAST;
        int r = (int)t.dims.size();
        if (need) {
            int64_t sp = 1;
            for (int i = 2; i < r; ++i) sp *= t.dims[i];
            if (t.dims[1] == 1 || sp == 1) need = false;
        }
        if (!need) {
            if (dst.kind == Loc::NONE) return t.loc;
            Loc src = t.loc;
            int64_t n = numel(t.dims);
            step([=](const RunCtx& c) { k::copy2d(c.s, c.at(src), c.mut(dst), 1, (int)n, (int)n, (int)n); }, 0, 2.0 * t.bytes());
            return dst;
        }
        if (dst.kind == Loc::NONE) dst = alloc_temp(t.bytes());
        std::vector<int64_t> pd = clast_phys_dims(t.dims), ps = contig_strides(pd), is(r);
        is[0] = ps[0];
        is[1] = ps[r - 1];
        for (int i = 2; i < r; ++i) is[i] = ps[i - 1];
        std::vector<int64_t> od = t.dims;
        Loc src = t.loc;
        step([=](const RunCtx& c) { k::permute(c.s, c.at(src), c.mut(dst), r, od.data(), is.data()); }, 0, 2.0 * t.bytes());
        return dst;
    }

    // ------------------------------------------------------------------ weights
    // [rows][K] row-major GEMM weights -> MFMA fragment order Wf[rows/16][KC][lane][4]
    // (lane = (row & 15) + 16 * g holds k = 16*kc + 4*g + j, j = 0..3); rows padded to 64, K to 16.
    static std::vector<float> to_fragment_order(const std::vector<float>& w, int64_t rows, int64_t K) {
        int64_t Rp = (rows + 63) / 64 * 64, KC = (K + 15) / 16;
        std::vector<float> f((size_t)Rp * KC * 16, 0.f);
        for (int64_t r = 0; r < rows; ++r)
            for (int64_t k = 0; k < K; ++k) {
                int64_t nf = r / 16, c = r % 16, kc = k / 16, g = (k % 16) / 4, j = k % 4;
                f[(size_t)(((nf * KC + kc) * 64 + (c + 16 * g)) * 4 + j)] = w[(size_t)r * K + k];
            }
        return f;
    }
    // [rows][K] f32 GEMM weights -> bf16x6 fragment order Wx[rows/16][K/32][3 planes][lane][8 bf16]
    // (lane = (row & 15) + 16 * g holds k = 32*kc + 8*g + e, e = 0..7; plane 0/1/2 = the h/m/l truncation pieces)
    static std::vector<float> to_fragment_x6(const std::vector<float>& w, int64_t rows, int64_t K) {
        int64_t Rp = (rows + 63) / 64 * 64, KC = (K + 31) / 32;
        std::vector<uint16_t> f((size_t)Rp * KC * 32 * 3, 0);
        for (int64_t r = 0; r < rows; ++r)
            for (int64_t k = 0; k < K; ++k) {
                float x = w[(size_t)r * K + k];
                uint32_t u, um, ul;
                std::memcpy(&u, &x, 4);
                uint32_t uh = u & 0xFFFF0000u;
                float fh; std::memcpy(&fh, &uh, 4);
                float r1 = x - fh;
                std::memcpy(&um, &r1, 4); um &= 0xFFFF0000u;
                float fm; std::memcpy(&fm, &um, 4);
                float r2 = r1 - fm;
                std::memcpy(&ul, &r2, 4); ul &= 0xFFFF0000u;
                int64_t nf = r / 16, c = r % 16, kc = k / 32, g = (k % 32) / 8, e = k % 8;
                size_t lane = (size_t)(c + 16 * g);
                size_t base = (size_t)((nf * KC + kc) * 3) * 64 * 8;
                f[base + (0 * 64 + lane) * 8 + e] = (uint16_t)(uh >> 16);
                f[base + (1 * 64 + lane) * 8 + e] = (uint16_t)(um >> 16);
                f[base + (2 * 64 + lane) * 8 + e] = (uint16_t)(ul >> 16);
            }
        std::vector<float> out(f.size() / 2);
        std::memcpy(out.data(), f.data(), f.size() * 2);
        return out;
    }
    // [rows][K] f32 GEMM weights -> the bf16x6 fragments of the row-streaming separable block (dsblock_rs.inc): Wr[rows/16][KP][3 planes][lane][8 bf16],
    // KP = ceil(ceil(K/16)/2) k-steps of two 16-channel chunks; lane = (row & 15) + 16 * g holds, for e = 0..3, k = 32*kp + 4*g + e (first chunk) and,
    // for e = 4..7, k = 32*kp + 16 + 4*g + (e - 4) (second chunk) -- the order in which that kernel's lanes hold the depthwise outputs
    static std::vector<float> to_fragment_x6_rs(const std::vector<float>& w, int64_t rows, int64_t K) {
        int64_t Rp = (rows + 63) / 64 * 64, KP = ((K + 15) / 16 + 1) / 2;
        std::vector<uint16_t> f((size_t)Rp * KP * 32 * 3, 0);
        for (int64_t r = 0; r < rows; ++r)
            for (int64_t k = 0; k < K; ++k) {
                float x = w[(size_t)r * K + k];
                uint32_t u, um, ul;
                std::memcpy(&u, &x, 4);
                uint32_t uh = u & 0xFFFF0000u;
                float fh; std::memcpy(&fh, &uh, 4);
                float r1 = x - fh;
                std::memcpy(&um, &r1, 4); um &= 0xFFFF0000u;
                float fm; std::memcpy(&fm, &um, 4);
                float r2 = r1 - fm;
                std::memcpy(&ul, &r2, 4); ul &= 0xFFFF0000u;
                int64_t nf = r / 16, c = r % 16, kp = k / 32, half = (k % 32) / 16, g = (k % 16) / 4, e = half * 4 + k % 4;
                size_t lane = (size_t)(c + 16 * g);
                size_t base = (size_t)((nf * KP + kp) * 3) * 64 * 8;
                f[base + (0 * 64 + lane) * 8 + e] = (uint16_t)(uh >> 16);
                f[base + (1 * 64 + lane) * 8 + e] = (uint16_t)(um >> 16);
                f[base + (2 * 64 + lane) * 8 + e] = (uint16_t)(ul >> 16);
            }
        std::vector<float> out(f.size() / 2);
        std::memcpy(out.data(), f.data(), f.size() * 2);
        return out;
    }
    static std::vector<float> to_fragments(int fmt, const std::vector<float>& w, int64_t rows, int64_t K) {
        if (fmt == k::IGEMM_W_X6RS) return to_fragment_x6_rs(w, rows, K);
        return fmt == k::IGEMM_W_X6 ? to_fragment_x6(w, rows, K) : to_fragment_order(w, rows, K);
    }
    const float* conv_weight_igemm(const GNode& n, const HostTensor& W, int fmt) {
        std::string key = "igemm" + std::to_string(fmt) + ":" + n.in[1];
        auto it = E.dev_consts_.find(key);
        if (it != E.dev_consts_.end()) return it->second;
        int64_t Co = W.dims[0], Ci = W.dims[1], kh = W.dims[2], kw = W.dims[3];
        int64_t K = kh * kw * Ci, Cp = (Co + 63) / 64 * 64;
        std::vector<float> w((size_t)Cp * K, 0.f);
        for (int64_t co = 0; co < Co; ++co)
            for (int64_t ci = 0; ci < Ci; ++ci)
                for (int64_t a = 0; a < kh; ++a)
                    for (int64_t b = 0; b < kw; ++b)
                        w[(size_t)co * K + (a * kw + b) * Ci + ci] = W.f[((co * Ci + ci) * kh + a) * kw + b];
        (void)Cp;
        w.resize((size_t)Co * K);
        return E.upload_const(key, to_fragments(fmt, w, Co, K));
    }
    // IGEMM_W_X6CS (dsblock_cs.inc): one block per 16-channel chunk c --
    //   [tap t = 0 .. ks*ks-1 | bias][g = 0..3] float4 = depthwise weights / bias of channels 16 c + 4 g .. + 3        (rounded up to 1 KB)
    //   per cout fragment f: lane (g, m = cout 16 f + m) -> 8 bytes w_l, 8 bytes w_h  (1 KB), then 8 bytes w_m (512 B)   (rounded up to 1 KB)
    // and after the last block the pointwise bias [Cout]
    // where w_h / w_m / w_l are the bf16 truncation pieces of W_pw[16 f + m][16 c + 4 g + e], e = 0..3
    const float* conv_weight_cs(const GNode& dwn, const GNode& pwn, const HostTensor& WD, const HostTensor& WP, const std::vector<float>* bias_dw, const std::vector<float>* bias_pw) {
        std::string key = "dscs:" + dwn.in[1] + ":" + pwn.in[1];
        auto it = E.dev_consts_.find(key);
        if (it != E.dev_consts_.end()) return it->second;
        const int64_t C = WD.dims[0], ks = WD.dims[2], Co = WP.dims[0], nch = C / 16, nft = Co / 16;
        const size_t blk = k::dsblock_cs_block_bytes((int)ks, (int)nft), dwb = (size_t)(((ks * ks + 1) * 64 + 1023) / 1024 * 1024);
        std::vector<float> out(blk * (size_t)nch / 4 + (size_t)Co, 0.f);   // + the pointwise bias [Cout] (zeros when the block has none)
        if (bias_pw) std::copy(bias_pw->begin(), bias_pw->begin() + Co, out.begin() + (ptrdiff_t)(blk * (size_t)nch / 4));
        auto split = [](float x, uint16_t& h, uint16_t& m, uint16_t& l) {
            uint32_t u, um, ul;
            std::memcpy(&u, &x, 4);
            const uint32_t uh = u & 0xFFFF0000u;
            float fh; std::memcpy(&fh, &uh, 4);
            const float r1 = x - fh;
            std::memcpy(&um, &r1, 4); um &= 0xFFFF0000u;
            float fm; std::memcpy(&fm, &um, 4);
            const float r2 = r1 - fm;
            std::memcpy(&ul, &r2, 4);
            h = (uint16_t)(uh >> 16); m = (uint16_t)(um >> 16); l = (uint16_t)(ul >> 16);
        };
        for (int64_t c = 0; c < nch; ++c) {
            char* b = reinterpret_cast<char*>(out.data()) + (size_t)c * blk;
            float* dw = reinterpret_cast<float*>(b);
            for (int64_t t = 0; t <= ks * ks; ++t)
                for (int64_t q = 0; q < 16; ++q) {
                    const int64_t ch = c * 16 + q;
                    dw[t * 16 + q] = t < ks * ks ? WD.f[(size_t)(ch * ks * ks + t)] : (bias_dw ? (*bias_dw)[(size_t)ch] : 0.f);
                }
            for (int64_t f = 0; f < nft; ++f) {
                uint16_t* lh = reinterpret_cast<uint16_t*>(b + dwb + (size_t)f * 1536);
                uint16_t* md = reinterpret_cast<uint16_t*>(b + dwb + (size_t)f * 1536 + 1024);
                for (int64_t g = 0; g < 4; ++g)
                    for (int64_t m = 0; m < 16; ++m)
                        for (int64_t e = 0; e < 4; ++e) {
                            uint16_t h, mm, l;
                            split(WP.f[(size_t)((f * 16 + m) * C + c * 16 + g * 4 + e)], h, mm, l);
                            const size_t lane = (size_t)(g * 16 + m);
                            lh[lane * 8 + e] = l; lh[lane * 8 + 4 + e] = h; md[lane * 4 + e] = mm;
                        }
            }
        }
        return E.upload_const(key, out);
    }
    const float* conv_weight_dw(const GNode& n, const HostTensor& W) {
        std::string key = "dw:" + n.in[1];
        auto it = E.dev_consts_.find(key);
        if (it != E.dev_consts_.end()) return it->second;
        int64_t C = W.dims[0], kh = W.dims[2], kw = W.dims[3];
        std::vector<float> w((size_t)C * kh * kw);
        for (int64_t c = 0; c < C; ++c)
            for (int64_t a = 0; a < kh; ++a)
                for (int64_t b = 0; b < kw; ++b) w[(size_t)(a * kw + b) * C + c] = W.f[(c * kh + a) * kw + b];
        return E.upload_const(key, w);
    }
    const float* conv_weight_direct(const GNode& n, const HostTensor& W) {
        std::string key = "direct:" + n.in[1];
        auto it = E.dev_consts_.find(key);
        if (it != E.dev_consts_.end()) return it->second;
        int64_t Co = W.dims[0], cpg = W.dims[1], kh = W.dims[2], kw = W.dims[3];
        std::vector<float> w((size_t)Co * cpg * kh * kw);
        for (int64_t co = 0; co < Co; ++co)
            for (int64_t ci = 0; ci < cpg; ++ci)
                for (int64_t a = 0; a < kh; ++a)
                    for (int64_t b = 0; b < kw; ++b)
                        w[(size_t)(((a * kw + b) * cpg + ci) * Co + co)] = W.f[((co * cpg + ci) * kh + a) * kw + b];
        return E.upload_const(key, w);
    }
    const float* convt_weight(const GNode& n, const HostTensor& W, bool igemm, int fmt = 0) {
        std::string key = (igemm ? "convt_ig" + std::to_string(fmt) + ":" : std::string("convt_dir:")) + n.in[1];
        auto it = E.dev_consts_.find(key);
        if (it != E.dev_consts_.end()) return it->second;
        int64_t Ci = W.dims[0], Co = W.dims[1], kh = W.dims[2], kw = W.dims[3];
        std::vector<float> w;
        if (igemm) {
            int64_t rows = kh * kw * Co;
            w.assign((size_t)rows * Ci, 0.f);
            for (int64_t ci = 0; ci < Ci; ++ci)
                for (int64_t co = 0; co < Co; ++co)
                    for (int64_t a = 0; a < kh; ++a)
                        for (int64_t b = 0; b < kw; ++b)
                            w[(size_t)((a * kw + b) * Co + co) * Ci + ci] = W.f[((ci * Co + co) * kh + a) * kw + b];
            w = to_fragments(fmt, w, rows, Ci);
        } else {
            w.assign((size_t)kh * kw * Ci * Co, 0.f);
            for (int64_t ci = 0; ci < Ci; ++ci)
                for (int64_t co = 0; co < Co; ++co)
                    for (int64_t a = 0; a < kh; ++a)
                        for (int64_t b = 0; b < kw; ++b)
                            w[(size_t)(((a * kw + b) * Ci + ci) * Co + co)] = W.f[((ci * Co + co) * kh + a) * kw + b];
        }
        return E.upload_const(key, w);
    }
    const float* linear_weight(const std::string& name, const HostTensor& B, bool transB, int fmt) {
        std::string key = "lin" + std::to_string(fmt) + ":" + (transB ? "t:" : "") + name;
        auto it = E.dev_consts_.find(key);
        if (it != E.dev_consts_.end()) return it->second;
        int64_t K = transB ? B.dims[1] : B.dims[0], N = transB ? B.dims[0] : B.dims[1];
        std::vector<float> w((size_t)N * K, 0.f);
        for (int64_t kk = 0; kk < K; ++kk)
            for (int64_t nn = 0; nn < N; ++nn) w[(size_t)nn * K + kk] = transB ? B.f[nn * K + kk] : B.f[kk * N + nn];
        return E.upload_const(key, to_fragments(fmt, w, N, K));
    }

    // ------------------------------------------------------------------ ops
    void get_pads(const GNode& n, int64_t H, int64_t W, int64_t kh, int64_t kw, int64_t sh, int64_t sw, int64_t dh, int64_t dw,
                  int64_t& pt, int64_t& pl, int64_t& pb, int64_t& pr) {
        std::vector<int64_t> pads = n.ais("pads");
        pt = pl = pb = pr = 0;
        if (pads.size() == 4) { pt = pads[0]; pl = pads[1]; pb = pads[2]; pr = pads[3]; }
        std::string ap = n.as("auto_pad", "NOTSET");
        if (ap == "SAME_UPPER" || ap == "SAME_LOWER") {
            int64_t oh = (H + sh - 1) / sh, ow = (W + sw - 1) / sw;
            int64_t th = std::max<int64_t>((oh - 1) * sh + (kh - 1) * dh + 1 - H, 0), tw = std::max<int64_t>((ow - 1) * sw + (kw - 1) * dw + 1 - W, 0);
            if (ap == "SAME_UPPER") { pt = th / 2; pb = th - pt; pl = tw / 2; pr = tw - pl; }
            else { pb = th / 2; pt = th - pb; pr = tw / 2; pl = tw - pr; }
        }
    }

    void op_conv(const GNode& n) {
        if (n.in.size() > 3 && !n.in[3].empty()) {   // rewrite pass 8: a squeeze-excite gate rides on this 1x1 conv
            TInfo xg = get(n.in[0]), gt = get(n.in[3]);
            const TInfo& wg = get(n.in[1]);
            bool ok = xg.dims.size() == 4 && !xg.host_int && !gt.host_int && wg.ht && wg.ht->dims.size() == 4 && gt.loc.kind != Loc::NONE && n.residual.empty();
            if (ok) {
                const int64_t N = xg.dims[0], C = xg.dims[1], HW = xg.dims[2] * xg.dims[3];
                ok = numel(gt.dims) == N * C && gt.dims.size() >= 2 && gt.dims[0] == N && gt.dims[1] == C && wg.ht->dims[1] == C && C % 8 == 0 &&
                     k::conv_igemm_se_ok(N * HW, (int)C, (int)wg.ht->dims[0], (int)HW);
            }
            if (!ok) {   // the Mul after all, then the plain conv
                GNode mul;
                mul.op = "Mul"; mul.in = {n.in[0], n.in[3]}; mul.out = {n.out[0] + "::se"};
                op_binary(mul, 2);
                GNode c2 = n;
                c2.in.resize(3);
                c2.in[0] = mul.out[0];
                op_conv(c2);
                return;
            }
        }
        PendingConcat msrc;   // non-empty: the input is a deferred concat this convolution reads source by source
        {
            auto pit = pending_concat.find(n.in[0]);
            if (pit != pending_concat.end()) {
                auto wi = E.inits_.find(n.in.size() > 1 ? n.in[1] : std::string());
                const std::vector<int64_t>& pd = pit->second.dims;
                const bool still_ok = wi != E.inits_.end() && wi->second.dims.size() == 4 && wi->second.dims[2] == 1 && wi->second.dims[3] == 1 && n.residual.empty() &&
                                      !(n.in.size() > 3 && !n.in[3].empty()) && n.out.size() == 1 &&
                                      k::conv_msrc_ok((long)(pd[0] * pd[2] * pd[3]), (int)pd[1], (int)wi->second.dims[0]);
                if (getenv("OAR_DEBUG_CONCAT")) fprintf(stderr, "  conv %s reads pending concat: still_ok=%d (w4=%d res=%d gate=%d nout=%zu)\n", n.out[0].c_str(), (int)still_ok, (int)(wi != E.inits_.end() && wi->second.dims.size() == 4), (int)!n.residual.empty(), (int)(n.in.size() > 3 && !n.in[3].empty()), n.out.size());
                if (still_ok) { msrc = pit->second; TInfo keep = vals[n.in[0]]; pending_concat.erase(pit); vals[n.in[0]] = keep; }
            }
        }
        TInfo x = get(n.in[0]);
        OAR_CHECK(x.dims.size() == 4, OAR_UNSUPPORTED_OP, "Conv: only 2-D convolutions are supported");
        const TInfo& wt = get(n.in[1]);
        OAR_CHECK(wt.ht, OAR_UNSUPPORTED_OP, "Conv: weights must be an initializer");
        const HostTensor& W = *wt.ht;
        OAR_CHECK(W.dims.size() == 4 && W.dims[0] > 0 && W.dims[1] > 0 && W.dims[2] > 0 && W.dims[3] > 0, OAR_MODEL_LOAD, "Conv: weight must be rank 4 with positive dims at " + n.out[0]);
        int64_t N = x.dims[0], Cin = x.dims[1], H = x.dims[2], Wd = x.dims[3];
        int64_t Cout = W.dims[0], kh = W.dims[2], kw = W.dims[3], g = n.ai("group", 1);
        OAR_CHECK(g >= 1 && W.dims[1] * g == Cin && Cout % g == 0, OAR_SHAPE_MISMATCH, "Conv: weight/input channel mismatch at " + n.out[0]);
        if (has_input(n, 2)) { const TInfo& bt = get(n.in[2]); OAR_CHECK(bt.ht && (int64_t)bt.ht->f.size() == Cout, OAR_MODEL_LOAD, "Conv: bias must be an f32 initializer of Cout elements at " + n.out[0]); }
        auto st = n.ais("strides"), dl = n.ais("dilations");
        int64_t sh = st.size() == 2 ? st[0] : 1, sw = st.size() == 2 ? st[1] : 1;
        int64_t dh = dl.size() == 2 ? dl[0] : 1, dw = dl.size() == 2 ? dl[1] : 1;
        int64_t pt, pl, pb, pr;
        get_pads(n, H, Wd, kh, kw, sh, sw, dh, dw, pt, pl, pb, pr);
        int64_t Ho = (H + pt + pb - dh * (kh - 1) - 1) / sh + 1, Wo = (Wd + pl + pr - dw * (kw - 1) - 1) / sw + 1;
        if (x.u8_stem) {   // Engine::run_stem: the stem reads the u8 pages and normalises on the fly (kernels.h StemU8)
            OAR_CHECK(g == 1 && Cin == 3 && kh * kw * 3 <= 128 && n.residual.empty(), OAR_INTERNAL, "run_stem on a graph whose first node is not an RGB stem");
            const float* sb = has_input(n, 2) ? get(n.in[2]).loc.cptr : nullptr;
            TInfo& ys = new_out(n.out[0], {N, Cout, Ho, Wo}, Layout::CLAST);
            k::ConvP sp{};
            sp.N = (int)N; sp.H = (int)H; sp.W = (int)Wd; sp.Cin = 3; sp.Ho = (int)Ho; sp.Wo = (int)Wo; sp.Cout = (int)Cout;
            sp.kh = (int)kh; sp.kw = (int)kw; sp.sh = (int)sh; sp.sw = (int)sw; sp.pt = (int)pt; sp.pl = (int)pl; sp.dh = (int)dh; sp.dw = (int)dw;
            sp.groups = 1; sp.act = n.act; sp.bias = sb; sp.y_ld = (int)Cout; sp.convt2x2 = 0;
            sp.w = conv_weight_direct(n, W);
            const Loc ysl = ys.loc;
            step([=](const RunCtx& c) {
                OAR_CHECK(c.stem != nullptr, OAR_INTERNAL, "stem plan run without pages");
                k::ConvP q = sp;
                q.y = c.mut(ysl);
                k::conv_smallcin_u8(c.s, q, *c.stem);
            }, 2.0 * N * Ho * Wo * Cout * 3 * kh * kw, 3.0 * N * H * Wd + 4.0 * N * Ho * Wo * Cout);
            return;
        }
        Loc xin = msrc.src.empty() ? to_clast_loc(x) : Loc();   // (a deferred concat has no storage of its own)
        const float* bias = has_input(n, 2) ? get(n.in[2]).loc.cptr : nullptr;
        Loc res;
        int res_up = 0;
        if (!n.residual.empty()) {
            // the other operand of the folded Add is a deferred nearest upsampling by an integer factor (the top-down path of an FPN): the
            // conv's epilogue reads the low-resolution tensor through the index map -- the upsampled tensor is never written or re-read
            const char* up_env = getenv("OAR_FUSE_RES_UP");   // (read per plan: the test compares both forms in one process)
            const bool up_on = !up_env || atoi(up_env) != 0;
            if (const PendingResize* pr = peek_pending(n.residual)) {
                const bool f32_tile = g == 1 && Cin % 4 == 0 && Cout % 4 == 0 &&
                                      k::igemm_weight_format((long)(N * Ho * Wo), (int)(kh * kw * Cin), (int)Cout, kh == 1 && kw == 1 && sh == 1 && sw == 1 && pt == 0 && pl == 0, (int)Cin) == k::IGEMM_W_K16;
                if (up_on && f32_tile && pr->fh > 1 && pr->fh == pr->fw && pr->N == N && pr->C == Cout && pr->Ho == Ho && pr->Wo == Wo && Ho % pr->fh == 0 && Wo % pr->fw == 0 &&
                    pr->H * pr->fh == Ho && pr->W * pr->fw == Wo && N * Ho * Wo < ((int64_t)1 << 31)) {
                    res = pr->xin;
                    res_up = pr->fh;
                    pending_resize.erase(n.residual);
                }
            }
        }
        if (!n.residual.empty() && res_up == 0) {
            TInfo r = get(n.residual);
            if (r.host_int || r.dims != std::vector<int64_t>{N, Cout, Ho, Wo}) {
                // the folded Add broadcasts: plan the conv and the add separately after all
                GNode pre = n;
                pre.residual.clear();
                pre.out[0] = n.out[0] + "::pre";
                op_conv(pre);
                GNode add;
                add.op = "Add"; add.in = {pre.out[0], n.residual}; add.out = {n.out[0]};
                op_binary(add, 0);
                return;
            }
            res = to_clast_loc(r);
        }
        TInfo& y = new_out(n.out[0], {N, Cout, Ho, Wo}, Layout::CLAST);
        k::ConvP p{};
        p.N = (int)N; p.H = (int)H; p.W = (int)Wd; p.Cin = (int)Cin; p.Ho = (int)Ho; p.Wo = (int)Wo; p.Cout = (int)Cout;
        p.kh = (int)kh; p.kw = (int)kw; p.sh = (int)sh; p.sw = (int)sw; p.pt = (int)pt; p.pl = (int)pl; p.dh = (int)dh; p.dw = (int)dw;
        p.groups = (int)g; p.act = n.act; p.bias = bias; p.y_ld = (int)Cout; p.convt2x2 = 0; p.res_up = res_up;
        int kind;  // 0 igemm, 1 dw, 2 direct, 3 grouped igemm
        // a grouped k x k convolution with wide groups (SVTRv2's local mixing: 5x5, 32 channels per group) is one implicit GEMM per group on the
        // output-stationary bf16x6 kernel, which reads its Cin / g channels out of the full tensor (ConvP::x_ld) and writes its Cout / g channels
        // into the full output (y_ld); the direct kernel it ran on before took 5.3 ms per layer (6.7 TFLOP/s).  OAR_GROUPED_X6=0 restores that
        const int64_t cg = Cin / g, og = Cout / g;
        const char* gx_env = getenv("OAR_GROUPED_X6");
        const bool grouped_x6 = g > 1 && g < Cin && cg % 8 == 0 && og % 4 == 0 && !(n.in.size() > 3 && !n.in[3].empty()) && res_up == 0 && !(gx_env && gx_env[0] == '0') &&
                                k::conv_grouped_x6_ok((long)(N * Ho * Wo), (int)(kh * kw * cg), (int)og, (int)cg);
        std::vector<const float*> wgrp;
        std::vector<int> slices3;
        if (grouped_x6) {
            kind = 3;
            p.w_fmt = k::IGEMM_W_X6;
            const size_t per = (size_t)(og * cg * kh * kw);
            for (int64_t gi = 0; gi < g; ++gi) {
                HostTensor Wg;
                Wg.dtype = DType::F32; Wg.dims = {og, cg, kh, kw};
                Wg.f.assign(W.f.begin() + (ptrdiff_t)(gi * per), W.f.begin() + (ptrdiff_t)((gi + 1) * per));
                GNode gn = n;
                gn.in[1] = n.in[1] + "::group" + std::to_string(gi);
                wgrp.push_back(conv_weight_igemm(gn, Wg, k::IGEMM_W_X6));
            }
        }
        else if (g == 1 && Cout <= 16 && Cin > 64 && kh == 3 && kw == 3 && sh == 1 && sw == 1 && pt == 1 && pl == 1 && dh == 1 && dw == 1 && Ho == H && Wo == Wd && n.residual.empty() &&
                 !(n.in.size() > 3 && !n.in[3].empty()) && !(slices3 = k::conv3x3_n16_slices((long)(N * Ho * Wo), (int)Cin, (int)Cout, (long)(H * Wd), (int)(4 * Cout))).empty()) {
            // the row-streaming bf16x6 kernel keeps a slice's weights in registers: 96 ... 256 input channels run as passes over 64- / 32-channel slices, every pass
            // after the first adding to the partial sums in y (the 0.445 M-parameter detector's neck: 96 -> 16 at quarter resolution ran on the f32 kernel, 201 us per 8 pages)
            kind = 4;
            p.w_fmt = k::IGEMM_W_X6;
            int64_t c0 = 0;
            for (int cs : slices3) {
                HostTensor Ws;
                Ws.dtype = DType::F32; Ws.dims = {Cout, (int64_t)cs, kh, kw};
                Ws.f.resize((size_t)(Cout * cs * 9));
                for (int64_t co = 0; co < Cout; ++co)
                    for (int64_t ci = 0; ci < cs; ++ci)
                        for (int64_t t = 0; t < 9; ++t) Ws.f[(size_t)((co * cs + ci) * 9 + t)] = W.f[(size_t)((co * Cin + c0 + ci) * 9 + t)];
                GNode gn = n;
                gn.in[1] = n.in[1] + "::slice" + std::to_string(c0);
                wgrp.push_back(conv_weight_igemm(gn, Ws, k::IGEMM_W_X6));
                c0 += cs;
            }
        }
        else if (g == 1 && Cin % 4 == 0) {
            kind = 0;
            const bool is1x1 = kh == 1 && kw == 1 && sh == 1 && sw == 1 && pt == 0 && pl == 0;
            const bool same3x3 = kh == 3 && kw == 3 && sh == 1 && sw == 1 && pt == 1 && pl == 1 && dh == 1 && dw == 1 && Ho == H && Wo == Wd && n.residual.empty();
            const bool lk_ok = !(n.in.size() > 3 && !n.in[3].empty()) && res_up == 0 &&
                               k::conv_lk_x6_eligible((int)kh, (int)kw, (int)sh, (int)sw, (int)pt, (int)pl, (int)dh, (int)dw, (int)H, (int)Wd, (int)Ho, (int)Wo, (int)Cin, (int)Cout, (int)(4 * Cout), (long)(N * Ho * Wo));
            p.w_fmt = k::igemm_weight_format((long)(N * Ho * Wo), (int)(kh * kw * Cin), (int)Cout, is1x1, (int)Cin, same3x3 ? (long)(H * Wd) : 0, lk_ok);
            p.w = conv_weight_igemm(n, W, p.w_fmt);
        }
        else if (g == Cin && g == Cout && Cout % 4 == 0) { kind = 1; p.w = conv_weight_dw(n, W); }
        else { kind = 2; p.w = conv_weight_direct(n, W); }
        Loc yl = y.loc;
        bool has_res = res.kind != Loc::NONE;
        Loc gate;   // [N][Cin] squeeze-excite gate folded into the load (checked above)
        if (n.in.size() > 3 && !n.in[3].empty()) {
            OAR_CHECK(kind == 0 && (p.w_fmt == k::IGEMM_W_X6 || p.w_fmt == k::IGEMM_W_K16), OAR_INTERNAL, "Conv: gate on a layer that is on neither gated kernel");
            gate = get(n.in[3]).loc;
        }
        const bool has_gate = gate.kind != Loc::NONE;
        double flops = 2.0 * N * Ho * Wo * Cout * (Cin / g) * kh * kw;
        double bytes = 4.0 * (N * H * Wd * Cin + N * Ho * Wo * Cout * (has_res ? 2 : 1) + numel(W.dims));
        if (n.out.size() > 1) {   // rewrite pass 9: out[1] = GlobalAveragePool(out[0])
            const int tiles = kind == 1 ? k::conv_dw_gap_tiles(p) : 0;
            if (tiles > 0 && Cout <= 1024 && n.ai("gap_raw", 0) != 0) {   // the only reader is an SEGate: it reduces the tile sums itself
                TInfo& gy = new_out(n.out[1], {N, (int64_t)tiles * Cout, 1, 1}, Layout::CLAST);
                gy.gap_tiles = tiles; gy.gap_hw = (int)(Ho * Wo);
                const Loc part = gy.loc;
                step([=](const RunCtx& c) {
                    k::ConvP q = p;
                    q.x = c.at(xin); q.y = c.mut(yl); q.residual = has_res ? c.at(res) : nullptr; q.gap_part = c.mut(part);
                    k::conv_dw(c.s, q);
                }, flops, bytes + 4.0 * N * tiles * Cout);
                return;
            }
            if (tiles > 0 && Cout <= 1024) {
                TInfo& gy = new_out(n.out[1], {N, Cout, 1, 1}, Layout::CLAST);
                const Loc gl = gy.loc, part = alloc_temp((size_t)N * tiles * Cout * sizeof(float));
                const int hw = (int)(Ho * Wo);
                step([=](const RunCtx& c) {
                    k::ConvP q = p;
                    q.x = c.at(xin); q.y = c.mut(yl); q.residual = has_res ? c.at(res) : nullptr; q.gap_part = c.mut(part);
                    k::conv_dw(c.s, q);
                }, flops, bytes + 4.0 * N * tiles * Cout);
                step([=](const RunCtx& c) { k::global_avgpool_finish(c.s, c.at(part), c.mut(gl), (int)N, tiles, (int)Cout, hw); }, 0, 4.0 * N * tiles * Cout);
                return;
            }
        }
        const bool from_concat = !msrc.src.empty();
        OAR_CHECK(!from_concat || (kind == 0 && p.w_fmt == k::IGEMM_W_X6), OAR_INTERNAL, "Conv: deferred concat in front of a layer that is not on the bf16x6 kernels");
        auto run = [=](const RunCtx& c) {
            k::ConvP q = p;
            q.y = c.mut(yl); q.residual = has_res ? c.at(res) : nullptr; q.se = has_gate ? c.at(gate) : nullptr;
            if (from_concat) {
                q.x = nullptr; q.n_msrc = (int)msrc.src.size();
                for (int i = 0; i < q.n_msrc; ++i) { q.msrc[i] = c.at(msrc.src[(size_t)i]); q.msrc_c[i] = msrc.c[(size_t)i]; }
            } else {
                q.x = c.at(xin);
            }
            if (kind == 3) {
                const int ng = p.groups, cgi = p.Cin / ng, ogi = p.Cout / ng;
                for (int gi = 0; gi < ng; ++gi) {
                    k::ConvP r = q;
                    r.groups = 1; r.Cin = cgi; r.Cout = ogi; r.x_ld = p.Cin;
                    r.x = q.x + (size_t)gi * cgi; r.y = q.y + (size_t)gi * ogi; r.w = wgrp[(size_t)gi];
                    r.bias = q.bias ? q.bias + (size_t)gi * ogi : nullptr;
                    r.residual = q.residual ? q.residual + (size_t)gi * ogi : nullptr;
                    k::conv_igemm(c.s, r);
                }
                return;
            }
            if (kind == 4) {
                int c0 = 0;
                for (size_t i = 0; i < slices3.size(); ++i) {
                    k::ConvP r = q;
                    r.Cin = slices3[i]; r.x_ld = p.Cin; r.x = q.x + c0; r.w = wgrp[i]; r.accum = i > 0 ? 1 : 0;
                    if (i > 0) r.bias = nullptr;
                    if (i + 1 < slices3.size()) r.act = k::Act{};
                    k::conv_igemm(c.s, r);
                    c0 += slices3[i];
                }
                return;
            }
            if (kind == 0) k::conv_igemm(c.s, q);
            else if (kind == 1) k::conv_dw(c.s, q);
            else k::conv_direct(c.s, q);
        };
        auto pool_after = [&]() {   // the pooled output without a pooled kernel variant: an ordinary GlobalAveragePool of out[0]
            if (n.out.size() < 2) return;
            GNode gp;
            gp.op = "GlobalAveragePool"; gp.in = {n.out[0]}; gp.out = {n.out[1]};
            op_gap(gp);
        };
        // a 1 x k convolution over a one-row map is a product over the rows of each sample: chainable (chain.hip)
        if (kind == 0 && !has_gate && H == 1 && Ho == 1 && kh == 1 && pt == 0 && pb == 0 && sw == 1 && dw == 1 && Wo == Wd && Cin % 16 == 0 && Cout % 16 == 0 && k::chain_act_ok(n.act.kind) &&
            chain_loc_ok(xin) && (!has_res || chain_loc_ok(res))) {
            ChainRec r;
            r.d.type = k::CH_GEMM; r.d.K = (int)(kw * Cin); r.d.N = (int)Cout; r.d.cin = (int)Cin; r.d.pad = (int)pl;
            r.d.in_ld = (int)Cin; r.d.out_ld = (int)Cout; r.d.res_ld = (int)Cout;
            r.d.act = n.act.kind; r.d.alpha = n.act.alpha; r.d.beta = n.act.beta; r.dev_bias = bias;
            r.in = xin; r.out = yl; if (has_res) r.res = res;
            r.rows = N * Wd; r.out_root = n.out[0];
            if (kw > 1) { r.fix_n = N; r.fix_T = Wd; }
            const HostTensor* Wp = &W;
            r.make_w = [Wp]() { return chain_weight_conv(*Wp); };
            step_chainable(run, std::move(r), flops, bytes);
            pool_after();
            return;
        }
        step(run, flops, bytes);
        pool_after();
    }

    // fused depthwise-separable block (rewrite pass 7)
    // shape part of a DSBlock node's parameters for an [N, C, H, W] input; false when the node is not a plain block
    bool dsblock_shape(const GNode& n, int64_t N, int64_t C, int64_t H, int64_t W, k::DsBlockP& p) {
        const TInfo &wdt = get(n.in[1]), &wpt = get(n.in[3]);
        if (!wdt.ht || !wpt.ht) return false;
        const HostTensor &WD = *wdt.ht, &WP = *wpt.ht;
        const int64_t ks = WD.dims[2], Cout = WP.dims[0];
        if (WD.dims[0] != C || WP.dims[1] != C) return false;
        auto st = n.ais("strides");
        const int64_t sh = st.size() == 2 ? st[0] : 1, sw = st.size() == 2 ? st[1] : 1;
        int64_t pt, pl, pb, pr;
        get_pads(n, H, W, ks, ks, sh, sw, 1, 1, pt, pl, pb, pr);
        const int64_t Ho = (H + pt + pb - (ks - 1) - 1) / sh + 1, Wo = (W + pl + pr - (ks - 1) - 1) / sw + 1;
        p = k::DsBlockP{};
        p.N = (int)N; p.H = (int)H; p.W = (int)W; p.C = (int)C; p.Ho = (int)Ho; p.Wo = (int)Wo; p.Cout = (int)Cout;
        p.ks = (int)ks; p.sh = (int)sh; p.sw = (int)sw; p.pt = (int)pt; p.pl = (int)pl;
        p.act1.kind = (int)n.ai("act1", 0); p.act1.alpha = n.af("act1_alpha", 0.f); p.act1.beta = n.af("act1_beta", 0.f);
        p.act2 = n.act; p.y_ld = (int)Cout;
        p.has_res = !n.residual.empty();
        return Ho > 0 && Wo > 0;
    }
    // DSBlock n followed by a DSBlock that is the ONLY reader of its output: one launch, the tensor between them never exists (csrc/dsblock_rs2.inc).
    // Decided here, at plan time, because eligibility depends on the shapes; the second node is then skipped by build().
    // May DSBlock node i be fused with node i + 1 as far as the GRAPH is concerned: i + 1 is a DSBlock reading i's output as its data input and nothing else does.
    bool dsblock_link_ok(int i) {
        if (i + 1 >= (int)E.nodes_.size()) return false;
        const GNode &n = E.nodes_[i], &m = E.nodes_[i + 1];
        if (n.op != "DSBlock" || m.op != "DSBlock" || !n.residual.empty() || !m.residual.empty() || m.in.empty() || m.in[0] != n.out[0]) return false;
        for (auto& o : E.output_names_) if (o == n.out[0]) return false;
        for (int j = 0; j < (int)E.nodes_.size(); ++j) {   // nobody else reads the intermediate tensor
            if (j == i + 1) continue;
            const GNode& q = E.nodes_[j];
            for (auto& s : q.in) if (s == n.out[0]) return false;
            if (q.residual == n.out[0]) return false;
        }
        for (size_t k = 1; k < m.in.size(); ++k) if (m.in[k] == n.out[0]) return false;
        return true;
    }
    // The pairing of a maximal run of chained DSBlocks, decided ONCE when its first block is planned (the round-5 rule decided per node --
    // "defer when the next pair is wider" -- and in a run of four or more blocks of growing width every block deferred, so only the last pair fused):
    // the set of disjoint adjacent pairs that keeps the most intermediate-tensor elements out of HBM (a three-line dynamic programme over the run).
    // OAR_DSBLOCK_RS2_FIRST=1: greedy from the front instead.
    std::map<int, bool> dsblock_pair_plan;   // node index -> fuse with the next node
    void plan_dsblock_run(int first, const TInfo& x) {
        std::vector<k::DsBlockP> ps;
        std::vector<bool> link;
        int64_t N = x.dims[0], C = x.dims[1], H = x.dims[2], W = x.dims[3];
        for (int i = first; i < (int)E.nodes_.size(); ++i) {
            const GNode& q = E.nodes_[i];
            k::DsBlockP p;
            if (q.op != "DSBlock" || q.in.size() < 5 || !dsblock_shape(q, N, C, H, W, p)) break;
            ps.push_back(p);
            C = p.Cout; H = p.Ho; W = p.Wo;
            const bool ok = dsblock_link_ok(i);
            link.push_back(ok);
            if (!ok) break;
        }
        const int R = (int)ps.size();
        std::vector<double> w(std::max(R - 1, 0), -1.0);   // elements of the tensor between block j and j + 1 when that pair can fuse
        for (int j = 0; j + 1 < R; ++j)
            if (link[j] && k::dsblock2_eligible(ps[j], ps[j + 1])) w[j] = (double)N * ps[j].Cout * ps[j].Ho * ps[j].Wo;
        static const bool greedy_first = [] { const char* e = getenv("OAR_DSBLOCK_RS2_FIRST"); return e && atoi(e) != 0; }();
        std::vector<bool> take(std::max(R - 1, 0), false);
        if (greedy_first) {
            for (int j = 0; j + 1 < R; ++j) if (w[j] > 0) { take[j] = true; ++j; }
        } else {
            std::vector<double> best(R + 1, 0.0);   // best[j]: blocks j.. of the run
            for (int j = R - 2; j >= 0; --j) best[j] = std::max(best[j + 1], w[j] > 0 ? w[j] + best[std::min(j + 2, R)] : 0.0);
            for (int j = 0; j + 1 < R;) {
                if (w[j] > 0 && w[j] + best[std::min(j + 2, R)] >= best[j + 1]) { take[j] = true; j += 2; }
                else ++j;
            }
        }
        for (int j = 0; j < R; ++j) dsblock_pair_plan[first + j] = j + 1 < R && take[j];
    }
    bool try_dsblock_pair(const GNode& n, const TInfo& x) {
        if (cur + 1 >= (int)E.nodes_.size() || !n.residual.empty() || x.dims.size() != 4 || x.host_int) return false;
        if (!dsblock_pair_plan.count(cur)) plan_dsblock_run(cur, x);
        if (!dsblock_pair_plan[cur]) return false;
        const GNode& m = E.nodes_[cur + 1];
        k::DsBlockP pa, pb;
        const int64_t N = x.dims[0], C = x.dims[1], H = x.dims[2], W = x.dims[3];
        if (!dsblock_shape(n, N, C, H, W, pa) || !dsblock_shape(m, N, pa.Cout, pa.Ho, pa.Wo, pb) || !k::dsblock2_eligible(pa, pb)) return false;
        auto weights = [&](const GNode& d, k::DsBlockP& p) {
            const HostTensor &WD = *get(d.in[1]).ht, &WP = *get(d.in[3]).ht;
            GNode dwn, pwn;
            dwn.op = "Conv"; dwn.in = {d.in[0], d.in[1]};
            pwn.op = "Conv"; pwn.in = {d.out[0] + "::dw", d.in[3]};
            if (!d.in[2].empty()) OAR_CHECK((int64_t)get(d.in[2]).ht->f.size() == p.C, OAR_MODEL_LOAD, "DSBlock: depthwise bias size");
            if (!d.in[4].empty()) OAR_CHECK((int64_t)get(d.in[4]).ht->f.size() == p.Cout, OAR_MODEL_LOAD, "DSBlock: pointwise bias size");
            p.wd = conv_weight_dw(dwn, WD);
            p.bd = d.in[2].empty() ? nullptr : get(d.in[2]).loc.cptr;
            p.wp = conv_weight_igemm(pwn, WP, k::IGEMM_W_X6RS);
            p.bp = d.in[4].empty() ? nullptr : get(d.in[4]).loc.cptr;
        };
        weights(n, pa);
        weights(m, pb);
        Loc xin = to_clast_loc(x);
        TInfo& y = new_out(m.out[0], {N, (int64_t)pb.Cout, (int64_t)pb.Ho, (int64_t)pb.Wo}, Layout::CLAST);
        Loc yl = y.loc;
        const double px = (double)N * H * W;
        const double flops = 2.0 * px * (pa.C * (9.0 + pa.Cout) + pb.C * (9.0 + pb.Cout));
        const double bytes = 4.0 * px * (pa.C + pb.Cout);
        step([=](const RunCtx& c) {
            k::DsBlockP qa = pa, qb = pb;
            qa.x = c.at(xin); qb.y = c.mut(yl);
            k::dsblock2(c.s, qa, qb);
        }, flops, bytes);
        fused_into_prev.insert(cur + 1);
        return true;
    }

    void op_dsblock(const GNode& n) {
        TInfo x = get(n.in[0]);
        if (try_dsblock_pair(n, x)) return;
        const TInfo &wdt = get(n.in[1]), &wpt = get(n.in[3]);
        OAR_CHECK(wdt.ht && wpt.ht, OAR_UNSUPPORTED_OP, "DSBlock: weights must be initializers");
        const HostTensor &WD = *wdt.ht, &WP = *wpt.ht;
        // the two convolutions this node stands for (used when the fused kernel does not take the shape)
        GNode dwn, pwn;
        dwn.op = "Conv"; dwn.in = {n.in[0], n.in[1]}; if (!n.in[2].empty()) dwn.in.push_back(n.in[2]);
        dwn.out = {n.as("mid_name", n.out[0] + "::dw")}; dwn.attrs = n.attrs;
        dwn.act.kind = (int)n.ai("act1", 0); dwn.act.alpha = n.af("act1_alpha", 0.f); dwn.act.beta = n.af("act1_beta", 0.f);
        pwn.op = "Conv"; pwn.in = {dwn.out[0], n.in[3]}; if (!n.in[4].empty()) pwn.in.push_back(n.in[4]);
        pwn.out = n.out; pwn.act = n.act; pwn.residual = n.residual;
        Attr g1; g1.kind = Attr::I; g1.i = 1; pwn.attrs["group"] = g1;
        auto unfused = [&]() { op_conv(dwn); op_conv(pwn); };
        if (x.dims.size() != 4 || x.host_int) return unfused();
        const int64_t N = x.dims[0], C = x.dims[1], H = x.dims[2], W = x.dims[3], ks = WD.dims[2], Cout = WP.dims[0];
        OAR_CHECK(WD.dims[0] == C && WP.dims[1] == C, OAR_SHAPE_MISMATCH, "DSBlock: channel mismatch at " + n.out[0]);
        auto st = n.ais("strides");
        const int64_t sh = st.size() == 2 ? st[0] : 1, sw = st.size() == 2 ? st[1] : 1;
        int64_t pt, pl, pb, pr;
        get_pads(n, H, W, ks, ks, sh, sw, 1, 1, pt, pl, pb, pr);
        const int64_t Ho = (H + pt + pb - (ks - 1) - 1) / sh + 1, Wo = (W + pl + pr - (ks - 1) - 1) / sw + 1;
        k::DsBlockP p{};
        p.N = (int)N; p.H = (int)H; p.W = (int)W; p.C = (int)C; p.Ho = (int)Ho; p.Wo = (int)Wo; p.Cout = (int)Cout;
        p.ks = (int)ks; p.sh = (int)sh; p.sw = (int)sw; p.pt = (int)pt; p.pl = (int)pl;
        p.act1 = dwn.act; p.act2 = n.act; p.y_ld = (int)Cout;
        p.has_res = !n.residual.empty();
        Loc res;
        if (!n.residual.empty()) {
            TInfo r = get(n.residual);
            if (r.host_int || r.dims != std::vector<int64_t>{N, Cout, Ho, Wo}) return unfused();   // broadcasting add: op_conv splits it off
        }
        if (Ho <= 0 || Wo <= 0 || !k::dsblock_eligible(p)) return unfused();
        if (!n.residual.empty()) res = to_clast_loc(get(n.residual));
        Loc xin = to_clast_loc(x);
        if (!n.in[2].empty()) OAR_CHECK((int64_t)get(n.in[2]).ht->f.size() == C, OAR_MODEL_LOAD, "DSBlock: depthwise bias size");
        if (!n.in[4].empty()) OAR_CHECK((int64_t)get(n.in[4]).ht->f.size() == Cout, OAR_MODEL_LOAD, "DSBlock: pointwise bias size");
        p.wd = conv_weight_dw(dwn, WD);
        p.bd = n.in[2].empty() ? nullptr : get(n.in[2]).loc.cptr;
        const int wfmt = k::dsblock_wp_format(p);
        p.wp = wfmt == k::IGEMM_W_X6CS ? conv_weight_cs(dwn, pwn, WD, WP, n.in[2].empty() ? nullptr : &get(n.in[2]).ht->f, n.in[4].empty() ? nullptr : &get(n.in[4]).ht->f) : conv_weight_igemm(pwn, WP, wfmt);
        p.bp = n.in[4].empty() ? nullptr : get(n.in[4]).loc.cptr;
        TInfo& y = new_out(n.out[0], {N, Cout, Ho, Wo}, Layout::CLAST);
        Loc yl = y.loc;
        const bool has_res = res.kind != Loc::NONE;
        const double flops = 2.0 * N * Ho * Wo * C * (ks * ks + (double)Cout);
        const double bytes = 4.0 * (N * H * W * C + N * Ho * Wo * Cout * (has_res ? 2 : 1) + numel(WD.dims) + numel(WP.dims));
        step([=](const RunCtx& c) {
            k::DsBlockP q = p;
            q.x = c.at(xin); q.y = c.mut(yl); q.residual = has_res ? c.at(res) : nullptr;
            k::dsblock(c.s, q);
        }, flops, bytes);
    }

    // shape facts of a ConvTranspose node that decide whether it is the plain 2x2 / stride-2 kind
    bool convt_is_2x2s2(const GNode& n, int64_t& cin, int64_t& cout) {
        auto wi = E.inits_.find(n.in.size() > 1 ? n.in[1] : std::string());
        if (wi == E.inits_.end() || wi->second.dtype != DType::F32 || wi->second.dims.size() != 4) return false;
        const auto& d = wi->second.dims;
        auto st = n.ais("strides"), dl = n.ais("dilations"), pads = n.ais("pads"), op = n.ais("output_padding");
        const bool s2 = st.size() == 2 && st[0] == 2 && st[1] == 2;
        bool zero = true;
        for (auto v : pads) zero = zero && v == 0;
        for (auto v : op) zero = zero && v == 0;
        for (auto v : dl) zero = zero && v == 1;
        cin = d[0]; cout = d[1];
        return d[2] == 2 && d[3] == 2 && s2 && zero && n.ai("group", 1) == 1 && n.residual.empty();
    }
    void op_convt(const GNode& n, bool may_defer = true) {
        static const bool pair_on = [] { const char* e = getenv("OAR_FUSE_CONVT_PAIR"); return !e || atoi(e) != 0; }();
        // (a) this node's input is a held-back ConvTranspose: run the pair as one kernel
        if (pair_on && !n.in.empty() && pending_convt.count(n.in[0])) {
            const GNode& a = *pending_convt[n.in[0]];
            int64_t c0 = 0, c1 = 0, c1b = 0, c2 = 0;
            if (convt_is_2x2s2(a, c0, c1) && convt_is_2x2s2(n, c1b, c2) && c1 == c1b && k::convt2x2_pair_supported((int)c0, (int)c1, (int)c2)) {
                TInfo x = get(a.in[0]);
                if (x.dims.size() == 4 && x.dims[1] == c0 && x.layout == Layout::CLAST && !x.host_int) {
                    pending_convt.erase(n.in[0]);
                    vals.erase(n.in[0]);
                    const HostTensor& W1 = E.inits_.at(a.in[1]);
                    const HostTensor& W2 = E.inits_.at(n.in[1]);
                    auto pack = [&](const std::string& key, const HostTensor& W) {   // [Cin][Cout][2][2] -> [Cin][a * 2 + b][Cout]
                        auto it = E.dev_consts_.find(key);
                        if (it != E.dev_consts_.end()) return it->second;
                        const int64_t Ci = W.dims[0], Co = W.dims[1];
                        std::vector<float> w((size_t)Ci * 4 * Co);
                        for (int64_t ci = 0; ci < Ci; ++ci)
                            for (int64_t co = 0; co < Co; ++co)
                                for (int64_t q = 0; q < 4; ++q) w[(size_t)((ci * 4 + q) * Co + co)] = W.f[(size_t)((ci * Co + co) * 4 + q)];
                        return E.upload_const(key, w);
                    };
                    k::ConvT2Pair cp{};
                    cp.w1 = pack("convt_pair1:" + a.in[1], W1);
                    cp.w2 = pack("convt_pair2:" + n.in[1], W2);
                    cp.b1 = has_input(a, 2) ? get(a.in[2]).loc.cptr : nullptr;
                    cp.b2 = has_input(n, 2) ? get(n.in[2]).loc.cptr : nullptr;
                    cp.N = (int)x.dims[0]; cp.H = (int)x.dims[2]; cp.W = (int)x.dims[3]; cp.C0 = (int)c0; cp.C1 = (int)c1; cp.C2 = (int)c2;
                    cp.act1 = a.act; cp.act2 = n.act;
                    const Loc xin = x.loc;
                    TInfo& y = new_out(n.out[0], {x.dims[0], c2, x.dims[2] * 4, x.dims[3] * 4}, Layout::CLAST);
                    const Loc yl = y.loc;
                    const double px = (double)x.dims[0] * x.dims[2] * x.dims[3];
                    step([=](const RunCtx& c) { k::ConvT2Pair q = cp; q.x = c.at(xin); q.y = c.mut(yl); k::convt2x2_pair(c.s, q); },
                         2.0 * px * 4.0 * ((double)c0 * c1 + 4.0 * c1 * c2), 4.0 * px * (c0 + 16.0 * c2));
                    return;
                }
            }
        }
        // (b) this node may itself be held back for its consumer
        if (pair_on && may_defer && sole_consumer.count(n.out[0]) && E.nodes_[sole_consumer[n.out[0]]].op == "ConvTranspose") {
            int64_t c0 = 0, c1 = 0, c1b = 0, c2 = 0;
            const GNode& b = E.nodes_[sole_consumer[n.out[0]]];
            const bool feeds_first = !b.in.empty() && b.in[0] == n.out[0];
            auto xi = vals.find(n.in[0]);
            if (feeds_first && convt_is_2x2s2(n, c0, c1) && convt_is_2x2s2(b, c1b, c2) && c1 == c1b && k::convt2x2_pair_supported((int)c0, (int)c1, (int)c2) &&
                xi != vals.end() && !peek_pending(n.in[0]) && !pending_convt.count(n.in[0]) && xi->second.dims.size() == 4 && xi->second.layout == Layout::CLAST) {
                pending_convt[n.out[0]] = &n;
                TInfo t;   // known shape, no storage
                t.dims = {xi->second.dims[0], c1, xi->second.dims[2] * 2, xi->second.dims[3] * 2};
                t.layout = Layout::CLAST; t.root = "";
                vals[n.out[0]] = t;
                return;
            }
        }
        TInfo x = get(n.in[0]);
        OAR_CHECK(x.dims.size() == 4, OAR_UNSUPPORTED_OP, "ConvTranspose: only 2-D");
        const TInfo& wt = get(n.in[1]);
        OAR_CHECK(wt.ht, OAR_UNSUPPORTED_OP, "ConvTranspose: weights must be an initializer");
        const HostTensor& W0 = *wt.ht;
        OAR_CHECK(W0.dims.size() == 4 && W0.dims[0] == x.dims[1] && W0.dims[1] > 0 && W0.dims[2] > 0 && W0.dims[3] > 0, OAR_MODEL_LOAD, "ConvTranspose: weight must be [Cin, Cout / group, kh, kw] at " + n.out[0]);
        const int64_t grp = n.ai("group", 1);
        OAR_CHECK(grp >= 1 && x.dims[1] % grp == 0, OAR_SHAPE_MISMATCH, "ConvTranspose: group must divide Cin at " + n.out[0]);
        // group > 1 (round 6; no PP-OCR graph has one): the weights as the block-diagonal [Cin, group * Cout / group, kh, kw] tensor of the equivalent dense layer
        HostTensor Wg;
        if (grp > 1) {
            const int64_t cig = W0.dims[0] / grp, cog = W0.dims[1], kk = W0.dims[2] * W0.dims[3];
            Wg.dtype = W0.dtype; Wg.dims = {W0.dims[0], grp * cog, W0.dims[2], W0.dims[3]};
            Wg.f.assign((size_t)(W0.dims[0] * grp * cog * kk), 0.f);
            for (int64_t ci = 0; ci < W0.dims[0]; ++ci)
                for (int64_t co = 0; co < cog; ++co)
                    for (int64_t t = 0; t < kk; ++t) Wg.f[(size_t)((ci * grp * cog + (ci / cig) * cog + co) * kk + t)] = W0.f[(size_t)((ci * cog + co) * kk + t)];
        }
        const HostTensor& W = grp > 1 ? Wg : W0;
        if (has_input(n, 2)) { const TInfo& bt = get(n.in[2]); OAR_CHECK(bt.ht && (int64_t)bt.ht->f.size() == W.dims[1], OAR_MODEL_LOAD, "ConvTranspose: bias must be an f32 initializer of Cout elements at " + n.out[0]); }
        int64_t N = x.dims[0], Cin = x.dims[1], H = x.dims[2], Wd = x.dims[3];
        int64_t Cout = W.dims[1], kh = W.dims[2], kw = W.dims[3];
        auto st = n.ais("strides"), dl = n.ais("dilations"), pads = n.ais("pads"), op = n.ais("output_padding");
        int64_t sh = st.size() == 2 ? st[0] : 1, sw = st.size() == 2 ? st[1] : 1;
        int64_t dh = dl.size() == 2 ? dl[0] : 1, dw = dl.size() == 2 ? dl[1] : 1;
        int64_t pt = pads.size() == 4 ? pads[0] : 0, pl = pads.size() == 4 ? pads[1] : 0, pb = pads.size() == 4 ? pads[2] : 0, pr = pads.size() == 4 ? pads[3] : 0;
        int64_t oph = op.size() == 2 ? op[0] : 0, opw = op.size() == 2 ? op[1] : 0;
        int64_t Ho = (H - 1) * sh - pt - pb + dh * (kh - 1) + oph + 1, Wo = (Wd - 1) * sw - pl - pr + dw * (kw - 1) + opw + 1;
        Loc xin = to_clast_loc(x);
        const float* bias = has_input(n, 2) ? get(n.in[2]).loc.cptr : nullptr;
        TInfo& y = new_out(n.out[0], {N, Cout, Ho, Wo}, Layout::CLAST);
        bool fast = kh == 2 && kw == 2 && sh == 2 && sw == 2 && pt == 0 && pl == 0 && pb == 0 && pr == 0 && oph == 0 && opw == 0 && Cin % 4 == 0 && dh == 1 && dw == 1;
        k::ConvP p{};
        p.N = (int)N; p.H = (int)H; p.W = (int)Wd; p.Cin = (int)Cin; p.Ho = (int)Ho; p.Wo = (int)Wo; p.Cout = (int)Cout;
        p.kh = (int)kh; p.kw = (int)kw; p.sh = (int)sh; p.sw = (int)sw; p.pt = (int)pt; p.pl = (int)pl; p.dh = (int)dh; p.dw = (int)dw;
        p.groups = 1; p.act = n.act; p.bias = bias; p.y_ld = (int)Cout; p.convt2x2 = fast ? 1 : 0;
        p.w_fmt = 0;   // the 2x2 scatter epilogue runs on the f32 kernels
        p.w = convt_weight(n, W, fast, p.w_fmt);
        Loc yl = y.loc;
        double flops = 2.0 * N * H * Wd * Cin * Cout * kh * kw;
        double bytes = 4.0 * (N * H * Wd * Cin + N * Ho * Wo * Cout + numel(W.dims));
        step([=](const RunCtx& c) {
            k::ConvP q = p;
            q.x = c.at(xin); q.y = c.mut(yl);
            if (fast) k::conv_igemm(c.s, q); else k::convt_direct(c.s, q);
        }, flops, bytes);
    }

    void op_bn(const GNode& n) {  // leftover BN: per-channel affine as a 1x1 depthwise conv
        TInfo x = get(n.in[0]);
        OAR_CHECK(x.dims.size() == 4, OAR_UNSUPPORTED_OP, "BatchNormalization: rank-4 only");
        const auto &ga = get(n.in[1]), &be = get(n.in[2]), &mu = get(n.in[3]), &va = get(n.in[4]);
        OAR_CHECK(ga.ht && be.ht && mu.ht && va.ht, OAR_UNSUPPORTED_OP, "BatchNormalization: params must be initializers");
        int64_t C = x.dims[1];
        OAR_CHECK((int64_t)ga.ht->f.size() == C && (int64_t)be.ht->f.size() == C && (int64_t)mu.ht->f.size() == C && (int64_t)va.ht->f.size() == C, OAR_MODEL_LOAD,
                  "BatchNormalization: scale / bias / mean / var must all have C elements (" + n.out[0] + ")");
        float eps = n.af("epsilon", 1e-5f);
        std::vector<float> sc(C), sh(C);
        for (int64_t c = 0; c < C; ++c) { sc[c] = ga.ht->f[c] / std::sqrt(va.ht->f[c] + eps); sh[c] = be.ht->f[c] - mu.ht->f[c] * sc[c]; }
        const float* dsc = E.upload_const("bn_sc:" + n.out[0], sc);
        const float* dsh = E.upload_const("bn_sh:" + n.out[0], sh);
        Loc xin = to_clast_loc(x);
        TInfo& y = new_out(n.out[0], x.dims, Layout::CLAST);
        k::ConvP p{};
        p.N = (int)x.dims[0]; p.H = p.Ho = (int)x.dims[2]; p.W = p.Wo = (int)x.dims[3]; p.Cin = p.Cout = (int)C;
        p.kh = p.kw = p.sh = p.sw = p.dh = p.dw = 1; p.groups = (int)C; p.act = n.act; p.w = dsc; p.bias = dsh; p.y_ld = (int)C;
        Loc yl = y.loc;
        bool dwok = C % 4 == 0;
        step([=](const RunCtx& c) {
            k::ConvP q = p; q.x = c.at(xin); q.y = c.mut(yl);
            if (dwok) k::conv_dw(c.s, q); else k::conv_direct(c.s, q);
        }, 0, 8.0 * numel(x.dims));
    }

    void op_unary(const GNode& n, Act a) {
        TInfo x = get(n.in[0]);
        TInfo& y = new_out(n.out[0], x.dims, x.layout);
        Loc xl = x.loc, yl = y.loc;
        int64_t cnt = numel(x.dims);
        step([=](const RunCtx& c) { k::unary(c.s, c.at(xl), c.mut(yl), cnt, a); }, 0, 8.0 * cnt);
    }

    // host permutation of a constant to channels-last physical order
    const float* const_clast(const std::string& name, const HostTensor& t, const std::vector<int64_t>& dims_aligned) {
        std::string key = "clast:" + name + ":" + std::to_string(dims_aligned.size());
        auto it = E.dev_consts_.find(key);
        if (it != E.dev_consts_.end()) return it->second;
        int r = (int)dims_aligned.size();
        std::vector<int64_t> pd = clast_phys_dims(dims_aligned), ns = contig_strides(dims_aligned), is(r);
        is[0] = ns[0];
        for (int i = 2; i < r; ++i) is[i - 1] = ns[i];
        is[r - 1] = ns[1];
        int64_t total = numel(pd);
        std::vector<float> out((size_t)total);
        std::vector<int64_t> idx(r, 0);
        for (int64_t i = 0; i < total; ++i) {
            int64_t rem = i, off = 0;
            for (int d = r - 1; d >= 0; --d) { int64_t q = rem / pd[d]; off += (rem - q * pd[d]) * is[d]; rem = q; }
            out[i] = t.f[off];
        }
        return E.upload_const(key, out);
    }

    void op_binary(const GNode& n, int op) {
        // a + upsample(b) (FPN top-down path): the sum reads b at (oh / f, ow / f) -- no upsampled tensor
        if (op == 0 && !pending_resize.empty() && n.act.kind == k::ACT_NONE) {
            for (int side = 0; side < 2; ++side) {
                const PendingResize* pr = peek_pending(n.in[side]);
                if (!pr || peek_pending(n.in[1 - side]) || !pr->fh || (pr->C & 3)) continue;
                const TInfo& o = get(n.in[1 - side]);
                if (o.host_int || o.ht || o.layout != Layout::CLAST || o.dims != pr->dims) continue;
                const PendingResize r = *pr;
                pending_resize.erase(n.in[side]);
                Loc al = o.loc;
                TInfo& y = new_out(n.out[0], r.dims, Layout::CLAST);
                Loc yl = y.loc;
                const double cnt = (double)numel(r.dims);
                step([=](const RunCtx& c) { k::binary_upsampled(c.s, c.at(al), c.at(r.xin), c.mut(yl), r.N, r.Ho, r.Wo, r.C, r.fh, r.fw, 0); }, cnt,
                     4.0 * (2.0 * cnt + cnt / (r.fh * r.fw)));
                return;
            }
        }
        TInfo a = get(n.in[0]), b = get(n.in[1]);
        // a plan-time value meeting a device tensor (e.g. a scale computed from Shape): materialise it as an f32 constant
        if (a.host_int) { a.loc = host_to_device(n.in[0], a); a.host_int = false; a.layout = Layout::NATIVE; }
        if (b.host_int) { b.loc = host_to_device(n.in[1], b); b.host_int = false; b.layout = Layout::NATIVE; }
        int r = (int)std::max(a.dims.size(), b.dims.size());
        auto align = [&](const std::vector<int64_t>& d) { std::vector<int64_t> o(r - d.size(), 1); o.insert(o.end(), d.begin(), d.end()); return o; };
        std::vector<int64_t> ad = align(a.dims), bd = align(b.dims), od(r);
        for (int i = 0; i < r; ++i) {
            OAR_CHECK(ad[i] == bd[i] || ad[i] == 1 || bd[i] == 1, OAR_SHAPE_MISMATCH, "binary: shapes do not broadcast at " + n.out[0]);
            od[i] = std::max(ad[i], bd[i]);
        }
        bool clast = (a.layout == Layout::CLAST && (int)a.dims.size() == r) || (b.layout == Layout::CLAST && (int)b.dims.size() == r);
        Loc al = a.loc, bl = b.loc;
        std::vector<int64_t> pad, pbd, pod;  // physical dims
        if (clast) {
            auto fix = [&](TInfo& t, const std::vector<int64_t>& d, const std::string& nm, Loc& l) {
                if (t.layout == Layout::CLAST && (int)t.dims.size() == r) return;
                // count non-1 dims: with <= 1 the memory order is unaffected by the permutation
                int non1 = 0;
                for (auto v : d) if (v != 1) ++non1;
                if (non1 <= 1) return;
                if (t.ht) { l.kind = Loc::CONST; l.cptr = const_clast(nm, *t.ht, d); return; }
                TInfo tmp = t; tmp.dims = d; tmp.layout = Layout::NATIVE;
                l = to_clast_loc(tmp);
            };
            fix(a, ad, n.in[0], al);
            fix(b, bd, n.in[1], bl);
            pad = clast_phys_dims(ad); pbd = clast_phys_dims(bd); pod = clast_phys_dims(od);
        } else {
            if (a.layout == Layout::CLAST) { al = to_native_loc(a); }
            if (b.layout == Layout::CLAST) { bl = to_native_loc(b); }
            pad = ad; pbd = bd; pod = od;
        }
        auto bstr = [&](const std::vector<int64_t>& d) {
            std::vector<int64_t> s = contig_strides(d);
            for (size_t i = 0; i < d.size(); ++i) if (d[i] == 1) s[i] = 0;
            return s;
        };
        std::vector<int64_t> sa = bstr(pad), sb = bstr(pbd);
        // commutative ops: put the full-size operand first (fast paths key on `a`)
        bool commut = op == 0 || op == 2 || op == 6 || op == 7 || op == 8 || op == 11 || op == 12;
        if (commut && numel(pad) < numel(pbd)) { std::swap(al, bl); std::swap(sa, sb); }
        TInfo& y = new_out(n.out[0], od, clast ? Layout::CLAST : Layout::NATIVE);
        Loc yl = y.loc;
        Act post = n.act;
        int64_t cnt = numel(od);
        step([=](const RunCtx& c) { k::binary(c.s, c.at(al), c.at(bl), c.mut(yl), op, r, pod.data(), sa.data(), sb.data(), post); }, (double)cnt, 12.0 * cnt);
    }

    // ReduceMean / ReduceSum / ReduceMax / ReduceMin / ReduceProd over the trailing axes (decomposed LayerNorm / softmax
    // exports), or ReduceMean over the spatial axes of an NCHW tensor (= GlobalAveragePool).  mode: k::reduce_lastdim's.
    void op_reduce(const GNode& n, int mode) {
        TInfo x = get(n.in[0]);
        const int r = (int)x.dims.size();
        std::vector<int64_t> axes = has_input(n, 1) ? get(n.in[1]).hv : n.ais("axes");
        if (axes.empty() && n.ai("noop_with_empty_axes", 0) == 0) for (int i = 0; i < r; ++i) axes.push_back(i);
        for (auto& a : axes) if (a < 0) a += r;
        std::sort(axes.begin(), axes.end());
        const bool keep = n.ai("keepdims", 1) != 0;
        bool spatial = mode == 0 && r == 4 && axes.size() == 2 && axes[0] == 2 && axes[1] == 3;
        if (spatial) {
            if (keep) return op_gap(n);
            GNode gp = n;   // keepdims = 0: the pooled [N, C, 1, 1] map, squeezed
            gp.out = {n.out[0] + "::kept"};
            op_gap(gp);
            GNode sq;
            sq.op = "Squeeze"; sq.in = {gp.out[0]}; sq.out = {n.out[0]};
            Attr ax; ax.kind = Attr::IS; ax.is = {2, 3}; sq.attrs["axes"] = ax;
            op_squeeze(sq);
            adopt_root(n.out[0], gp.out[0]);
            return;
        }
        bool trailing = !axes.empty();
        for (size_t i = 0; i < axes.size(); ++i) trailing = trailing && axes[i] == r - (int64_t)axes.size() + (int64_t)i;
        if (!trailing && (mode == 0 || mode == 2) && r == 4 && axes.size() == 1 && (axes[0] == 2 || axes[0] == 3) && !x.host_int) {
            // mean / max over ONE spatial axis of a feature map (SVTRv2's `x.mean(2)` in front of its CTC head): a pooling window as tall (wide)
            // as the map, on the channels-last tensor where it lies; keepdims = 0 is then a Squeeze of a channels-last view
            GNode pl;
            pl.op = mode == 0 ? "AveragePool" : "MaxPool"; pl.in = {n.in[0]}; pl.out = {keep ? n.out[0] : n.out[0] + "::kept"};
            Attr ks; ks.kind = Attr::IS; ks.is = axes[0] == 2 ? std::vector<int64_t>{x.dims[2], 1} : std::vector<int64_t>{1, x.dims[3]};
            pl.attrs["kernel_shape"] = ks; pl.attrs["strides"] = ks;
            op_pool(pl, mode == 2);
            if (!keep) {
                GNode sq;
                sq.op = "Squeeze"; sq.in = {pl.out[0]}; sq.out = {n.out[0]};
                Attr ax; ax.kind = Attr::IS; ax.is = {axes[0]}; sq.attrs["axes"] = ax;
                op_squeeze(sq);
                adopt_root(n.out[0], pl.out[0]);
            }
            return;
        }
        if (!trailing && !axes.empty() && !x.host_int) {
            // any other axis set (round 6; no PP-OCR graph has one -- tools/op_fuzz.py asked): the reduced axes are moved behind the others by a Transpose, reduced as
            // trailing axes, and keepdims = 1 is a view of the result with the ones back in place
            std::vector<int64_t> perm, od;
            for (int i = 0; i < r; ++i) if (!std::binary_search(axes.begin(), axes.end(), (int64_t)i)) perm.push_back(i);
            for (auto a : axes) perm.push_back(a);
            for (int i = 0; i < r; ++i) { if (!std::binary_search(axes.begin(), axes.end(), (int64_t)i)) od.push_back(x.dims[i]); else if (keep) od.push_back(1); }
            GNode tr;
            tr.op = "Transpose"; tr.in = {n.in[0]}; tr.out = {n.out[0] + "::moved"};
            Attr pa; pa.kind = Attr::IS; pa.is = perm; tr.attrs["perm"] = pa;
            op_transpose(tr);
            GNode rd = n;
            rd.in = {tr.out[0]}; rd.out = {keep ? n.out[0] + "::reduced" : n.out[0]};
            std::vector<int64_t> tail;
            for (size_t i = 0; i < axes.size(); ++i) tail.push_back(r - (int64_t)axes.size() + (int64_t)i);
            Attr ta; ta.kind = Attr::IS; ta.is = tail; rd.attrs["axes"] = ta;
            Attr kd; kd.kind = Attr::I; kd.i = 0; rd.attrs["keepdims"] = kd;
            op_reduce(rd, mode);
            if (keep) { GNode v; v.op = "Reshape"; v.in = {rd.out[0]}; v.out = {n.out[0]}; view_native(v, get(rd.out[0]), od); adopt_root(n.out[0], rd.out[0]); }
            return;
        }
        OAR_CHECK(trailing, OAR_UNSUPPORTED_OP, n.op + ": only the trailing axes (or H, W / one of them for ReduceMean / ReduceMax of an NCHW tensor) are supported");
        Loc xin = to_native_loc(x);
        int64_t C = 1;
        for (auto a : axes) C *= x.dims[a];
        const int64_t rows = numel(x.dims) / std::max<int64_t>(C, 1);
        std::vector<int64_t> od(x.dims.begin(), x.dims.end() - (int64_t)axes.size());
        if (keep) for (size_t i = 0; i < axes.size(); ++i) od.push_back(1);
        TInfo& y = new_out(n.out[0], od, Layout::NATIVE);
        Loc yl = y.loc;
        step([=](const RunCtx& c) { k::reduce_lastdim(c.s, c.at(xin), c.mut(yl), rows, (int)C, mode); }, (double)rows * C, 4.0 * rows * (C + 1));
    }

    // ArgMax / ArgMin over the last axis
    void op_argreduce(const GNode& n, bool is_min) {
        TInfo x = get(n.in[0]);
        const int r = (int)x.dims.size();
        int64_t axis = n.ai("axis", 0);
        if (axis < 0) axis += r;
        OAR_CHECK(r >= 1 && axis == r - 1, OAR_UNSUPPORTED_OP, n.op + ": only the last axis is supported");
        Loc xin = to_native_loc(x);
        const int64_t C = x.dims[axis], rows = numel(x.dims) / std::max<int64_t>(C, 1);
        OAR_CHECK(C >= 1 && C < (1 << 24), OAR_UNSUPPORTED_OP, n.op + ": axis length must be in [1, 2^24)");
        std::vector<int64_t> od(x.dims.begin(), x.dims.end() - 1);
        if (n.ai("keepdims", 1) != 0) od.push_back(1);
        const bool last = n.ai("select_last_index", 0) != 0;
        TInfo& y = new_out(n.out[0], od, Layout::NATIVE);
        y.is_int = true;
        Loc yl = y.loc;
        step([=](const RunCtx& c) { k::argreduce_lastdim(c.s, c.at(xin), c.mut(yl), rows, (int)C, is_min, last); }, (double)rows * C, 4.0 * rows * (C + 1));
    }

    // ------------------------------------------------------------------ broadcast copies (Expand / Tile)
    // y (contiguous, dims od) = x read through `in_strides` (0 = broadcast); dims of size 1 are dropped and neighbours that
    // are contiguous in x are merged so that real-world ranks fit the 6-d permute kernel
    void strided_copy(const std::string& out, Loc xin, std::vector<int64_t> od_full, std::vector<int64_t> vd, std::vector<int64_t> vs) {
        std::vector<int64_t> d, st;
        for (size_t i = 0; i < vd.size(); ++i) {
            if (vd[i] == 1) continue;
            if (!d.empty() && st.back() == vs[i] * vd[i] && vs[i] != 0) { d.back() *= vd[i]; st.back() = vs[i]; }
            else if (!d.empty() && st.back() == 0 && vs[i] == 0) d.back() *= vd[i];
            else { d.push_back(vd[i]); st.push_back(vs[i]); }
        }
        if (d.empty()) { d.push_back(1); st.push_back(0); }
        OAR_CHECK(d.size() <= 6, OAR_UNSUPPORTED_OP, "Expand / Tile: more than 6 effective dimensions at " + out);
        TInfo& y = new_out(out, od_full, Layout::NATIVE);
        Loc yl = y.loc;
        const int r = (int)d.size();
        step([=](const RunCtx& c) { k::permute(c.s, c.at(xin), c.mut(yl), r, d.data(), st.data()); }, 0, 8.0 * numel(od_full));
    }
    void op_expand(const GNode& n) {
        TInfo x = get(n.in[0]);
        const TInfo& sh = get(n.in[1]);
        OAR_CHECK(sh.host_int, OAR_UNSUPPORTED_OP, "Expand: shape must be known on the host");
        const int r = (int)std::max(x.dims.size(), sh.hv.size());
        std::vector<int64_t> xd(r - x.dims.size(), 1), td(r - sh.hv.size(), 1), od(r);
        xd.insert(xd.end(), x.dims.begin(), x.dims.end());
        td.insert(td.end(), sh.hv.begin(), sh.hv.end());
        for (int i = 0; i < r; ++i) {
            OAR_CHECK(xd[i] == td[i] || xd[i] == 1 || td[i] == 1, OAR_SHAPE_MISMATCH, "Expand: shapes do not broadcast at " + n.out[0]);
            od[i] = std::max(xd[i], td[i]);
        }
        Loc xin = to_native_loc(x);
        std::vector<int64_t> st = contig_strides(xd);
        for (int i = 0; i < r; ++i) if (xd[i] == 1) st[i] = 0;
        strided_copy(n.out[0], xin, od, od, st);
    }
    void op_tile(const GNode& n) {
        TInfo x = get(n.in[0]);
        const TInfo& rp = get(n.in[1]);
        OAR_CHECK(rp.host_int && rp.hv.size() == x.dims.size(), OAR_UNSUPPORTED_OP, "Tile: repeats must be host ints, one per axis");
        Loc xin = to_native_loc(x);
        std::vector<int64_t> xs = contig_strides(x.dims), od, vd, vs;
        for (size_t i = 0; i < x.dims.size(); ++i) {
            OAR_CHECK(rp.hv[i] >= 1, OAR_INVALID_INPUT, "Tile: repeats must be >= 1");
            od.push_back(x.dims[i] * rp.hv[i]);
            vd.push_back(rp.hv[i]); vs.push_back(0);          // [repeat][extent] with stride [0][s]
            vd.push_back(x.dims[i]); vs.push_back(xs[i]);
        }
        strided_copy(n.out[0], xin, od, vd, vs);
    }
    void op_constant_of_shape(const GNode& n) {
        const TInfo& sh = get(n.in[0]);
        OAR_CHECK(sh.host_int, OAR_UNSUPPORTED_OP, "ConstantOfShape: shape must be known on the host");
        double v = 0.0; bool is_f = true;
        auto it = n.attrs.find("value");
        if (it != n.attrs.end() && it->second.kind == Attr::T) {
            const HostTensor& t = it->second.t;
            if (t.dtype == DType::F32) v = t.f.empty() ? 0.0 : t.f[0];
            else { v = t.i.empty() ? 0.0 : (double)t.i[0]; is_f = false; }
        }
        for (auto d : sh.hv) OAR_CHECK(d >= 0 && d <= (int64_t)1 << 28, OAR_SHAPE_MISMATCH, "ConstantOfShape: negative or unreasonable dimension");   // (two negative dims would multiply to a plausible count)
        const int64_t cnt = numel(sh.hv);
        OAR_CHECK(cnt >= 0 && cnt <= (int64_t)1 << 28, OAR_SHAPE_MISMATCH, "ConstantOfShape: unreasonable element count");
        if (!is_f || cnt <= 64) {   // shape plumbing stays on the host
            TInfo o; o.host_int = true; o.dims = sh.hv; o.host_f = is_f;
            o.hv.assign((size_t)cnt, (int64_t)v); o.hd.assign((size_t)cnt, v);
            vals[n.out[0]] = o; return;
        }
        TInfo t;
        t.dims = sh.hv; t.layout = Layout::NATIVE; t.loc.kind = Loc::CONST;
        t.loc.cptr = E.upload_const("cos:" + std::to_string(v) + ":" + std::to_string(cnt), std::vector<float>((size_t)cnt, (float)v));
        vals[n.out[0]] = t;
    }
    // device value of a host-known tensor (a bool / int mask or scalar that a device op consumes)
    Loc host_to_device(const std::string& name, const TInfo& t) {
        std::vector<float> f(t.hv.size());
        for (size_t i = 0; i < f.size(); ++i) f[i] = t.host_f && i < t.hd.size() ? (float)t.hd[i] : (float)t.hv[i];
        std::string key = "hostval:" + name + ":";
        for (float v : f) key += std::to_string(v) + ",";
        Loc l; l.kind = Loc::CONST; l.cptr = E.upload_const(key, f);
        return l;
    }
    void op_where(const GNode& n) {
        TInfo c = get(n.in[0]), a = get(n.in[1]), b = get(n.in[2]);
        auto dev = [&](const std::string& nm, TInfo& t) -> Loc { return t.host_int ? host_to_device(nm, t) : to_native_loc(t); };
        Loc cl, al = dev(n.in[1], a), bl = dev(n.in[2], b);
        if (c.host_int) cl = host_to_device(n.in[0], c);
        else if (c.ht == nullptr && c.loc.kind != Loc::NONE) cl = to_native_loc(c);
        else cl = c.loc;   // a bool initializer arrives as host ints (handled above); f32 0/1 masks as constants
        const int r = (int)std::max({c.dims.size(), a.dims.size(), b.dims.size()});
        OAR_CHECK(r <= 6, OAR_UNSUPPORTED_OP, "Where: rank > 6");
        auto align = [&](const std::vector<int64_t>& d) { std::vector<int64_t> o(r - d.size(), 1); o.insert(o.end(), d.begin(), d.end()); return o; };
        std::vector<int64_t> cd = align(c.dims), ad = align(a.dims), bd = align(b.dims), od(r);
        for (int i = 0; i < r; ++i) {
            od[i] = std::max({cd[i], ad[i], bd[i]});
            OAR_CHECK((cd[i] == od[i] || cd[i] == 1) && (ad[i] == od[i] || ad[i] == 1) && (bd[i] == od[i] || bd[i] == 1), OAR_SHAPE_MISMATCH, "Where: shapes do not broadcast at " + n.out[0]);
        }
        auto bstr = [&](const std::vector<int64_t>& d) { std::vector<int64_t> st = contig_strides(d); for (size_t i = 0; i < d.size(); ++i) if (d[i] == 1) st[i] = 0; return st; };
        std::vector<int64_t> sc = bstr(cd), sa = bstr(ad), sb = bstr(bd);
        TInfo& y = new_out(n.out[0], od, Layout::NATIVE);
        Loc yl = y.loc;
        step([=](const RunCtx& cx) { k::where(cx.s, cx.at(cl), cx.at(al), cx.at(bl), cx.mut(yl), r, od.data(), sc.data(), sa.data(), sb.data()); }, 0, 16.0 * numel(od));
    }

    // ------------------------------------------------------------------ host evaluation (shape arithmetic)
    // Paddle2ONNX / torch exports compute Reshape / Resize / Slice arguments with little integer (and sometimes float)
    // graphs hanging off Shape nodes.  Every value reachable from Shape + constants only is evaluated here, at plan time,
    // with numpy broadcasting; nothing of it is ever launched.
    static std::vector<double> host_values(const TInfo& t) {
        std::vector<double> v;
        if (t.host_int) {
            if (t.host_f && t.hd.size() == t.hv.size()) return t.hd;
            v.assign(t.hv.begin(), t.hv.end());
        } else if (t.ht) {
            v.assign(t.ht->f.begin(), t.ht->f.end());
        }
        return v;
    }
    bool host_evaluable(const TInfo& t) const { return t.host_int || (t.ht && t.ht->f.size() <= 64); }
    void set_host(const std::string& name, const std::vector<int64_t>& dims, const std::vector<double>& v, bool is_float) {
        TInfo o;
        o.host_int = true; o.host_f = is_float; o.dims = dims; o.hd = v;
        o.hv.resize(v.size());
        for (size_t i = 0; i < v.size(); ++i) o.hv[i] = (int64_t)v[i];   // truncation toward zero, like Cast
        vals[name] = o;
    }
    static bool float_like(const TInfo& t) { return t.host_int ? t.host_f : t.ht != nullptr; }
    // returns true when the node was evaluated on the host
    bool op_host(const GNode& n) {
        const std::string& op = n.op;
        std::vector<TInfo> in;
        for (auto& sname : n.in) in.push_back(sname.empty() ? TInfo() : get(sname));
        auto bin = [&](int code) {
            const TInfo &a = in[0], &b = in[1];
            std::vector<double> av = host_values(a), bv = host_values(b);
            const int r = (int)std::max(a.dims.size(), b.dims.size());
            std::vector<int64_t> ad(r - a.dims.size(), 1), bd(r - b.dims.size(), 1), od(r);
            ad.insert(ad.end(), a.dims.begin(), a.dims.end());
            bd.insert(bd.end(), b.dims.begin(), b.dims.end());
            for (int i = 0; i < r; ++i) {
                OAR_CHECK(ad[i] == bd[i] || ad[i] == 1 || bd[i] == 1, OAR_SHAPE_MISMATCH, "host " + op + ": shapes do not broadcast at " + n.out[0]);
                od[i] = std::max(ad[i], bd[i]);
            }
            std::vector<int64_t> sa = contig_strides(ad), sb = contig_strides(bd);
            const bool fl = (float_like(a) || float_like(b)) && code <= 6;
            std::vector<double> out((size_t)numel(od));
            for (int64_t i = 0; i < (int64_t)out.size(); ++i) {
                int64_t rem = i, oa = 0, ob = 0;
                for (int d = r - 1; d >= 0; --d) { const int64_t q = rem / od[d], ix = rem - q * od[d]; rem = q; oa += (ad[d] == 1 ? 0 : ix) * sa[d]; ob += (bd[d] == 1 ? 0 : ix) * sb[d]; }
                const double x = av[(size_t)oa], y = bv[(size_t)ob];
                double v = 0;
                switch (code) {
                    case 0: v = x + y; break;
                    case 1: v = x - y; break;
                    case 2: v = x * y; break;
                    case 3: v = fl ? x / y : (y != 0 ? (double)((int64_t)x / (int64_t)y) : 0.0); break;   // integer Div truncates
                    case 4: v = std::pow(x, y); break;
                    case 5: v = std::max(x, y); break;
                    case 6: v = std::min(x, y); break;
                    case 7: v = x == y; break;
                    case 8: v = x < y; break;
                    case 9: v = x > y; break;
                    case 10: v = (x != 0) && (y != 0); break;
                    default: v = (x != 0) || (y != 0); break;
                }
                out[(size_t)i] = fl ? (double)(float)v : v;   // f32 tensors stay f32
            }
            set_host(n.out[0], od, out, fl);
            return true;
        };
        auto un = [&](double (*f)(double)) {
            std::vector<double> v = host_values(in[0]);
            for (auto& e : v) e = f(e);
            set_host(n.out[0], in[0].dims, v, float_like(in[0]));
            return true;
        };
        if (op == "Add") return bin(0);
        if (op == "Sub") return bin(1);
        if (op == "Mul") return bin(2);
        if (op == "Div") return bin(3);
        if (op == "Pow") return bin(4);
        if (op == "Max") return bin(5);
        if (op == "Min") return bin(6);
        if (op == "Equal") return bin(7);
        if (op == "Less") return bin(8);
        if (op == "Greater") return bin(9);
        if (op == "And") return bin(10);
        if (op == "Or") return bin(11);
        if (op == "Neg") return un([](double v) { return -v; });
        if (op == "Abs") return un([](double v) { return std::fabs(v); });
        if (op == "Floor") return un([](double v) { return std::floor(v); });
        if (op == "Ceil") return un([](double v) { return std::ceil(v); });
        if (op == "Round") return un([](double v) { return std::nearbyint(v); });
        if (op == "Sqrt") return un([](double v) { return (double)std::sqrt((float)v); });
        if (op == "Not") return un([](double v) { return v != 0 ? 0.0 : 1.0; });
        if (op == "Identity") { set_host(n.out[0], in[0].dims, host_values(in[0]), float_like(in[0])); return true; }
        if (op == "Cast") {
            const int64_t to = n.ai("to", 1);
            std::vector<double> v = host_values(in[0]);
            const bool to_f = to == 1 || to == 10 || to == 11 || to == 16;
            if (!to_f) for (auto& e : v) e = to == 9 ? (double)(e != 0) : (double)(int64_t)e;
            else if (to == 1) for (auto& e : v) e = (double)(float)e;
            set_host(n.out[0], in[0].dims, v, to_f);
            return true;
        }
        if (op == "Where") {
            std::vector<double> c = host_values(in[0]), a = host_values(in[1]), b = host_values(in[2]);
            const size_t cnt = std::max({c.size(), a.size(), b.size()});
            if (!((c.size() == cnt || c.size() == 1) && (a.size() == cnt || a.size() == 1) && (b.size() == cnt || b.size() == 1))) return false;   // device path broadcasts
            std::vector<double> out(cnt);
            for (size_t i = 0; i < cnt; ++i) out[i] = c[c.size() == 1 ? 0 : i] != 0 ? a[a.size() == 1 ? 0 : i] : b[b.size() == 1 ? 0 : i];
            const TInfo& big = c.size() == cnt ? in[0] : a.size() == cnt ? in[1] : in[2];
            set_host(n.out[0], big.dims, out, float_like(in[1]) || float_like(in[2]));
            return true;
        }
        if (op == "Expand") {
            if (!in[1].host_int) return false;
            std::vector<double> v = host_values(in[0]);
            const int64_t cnt = numel(in[1].hv);
            if (!(v.size() == 1 || ((int64_t)v.size() == cnt && in[0].dims.size() <= 1)) || cnt > 4096) return false;   // real data: device path
            std::vector<double> out((size_t)cnt);
            for (int64_t i = 0; i < cnt; ++i) out[(size_t)i] = v[v.size() == 1 ? 0 : (size_t)i];
            set_host(n.out[0], in[1].hv, out, float_like(in[0]));
            return true;
        }
        if (op == "Tile") {
            if (!(in[0].dims.size() <= 1 && in[1].hv.size() == 1 && in[1].host_int)) return false;   // real data: device path
            std::vector<double> v = host_values(in[0]), out;
            // (a shape-arithmetic vector: the repeat count comes from the model file -- bounded, and never negative)
            OAR_CHECK(in[1].hv[0] >= 0 && in[1].hv[0] * (int64_t)std::max<size_t>(v.size(), 1) <= 4096, OAR_MODEL_LOAD, "Tile: unreasonable repeat count of a host vector at " + n.out[0]);
            for (int64_t k = 0; k < in[1].hv[0]; ++k) out.insert(out.end(), v.begin(), v.end());
            set_host(n.out[0], {(int64_t)out.size()}, out, float_like(in[0]));
            return true;
        }
        if (op == "Range") {
            const double a = host_values(in[0])[0], lim = host_values(in[1])[0], d = host_values(in[2])[0];
            OAR_CHECK(d != 0, OAR_INVALID_INPUT, "Range: delta 0");
            std::vector<double> out;
            const int64_t cnt = std::max<int64_t>((int64_t)std::ceil((lim - a) / d), 0);
            OAR_CHECK(cnt <= 1 << 20, OAR_SHAPE_MISMATCH, "Range: too long");
            for (int64_t i = 0; i < cnt; ++i) out.push_back(a + (double)i * d);
            set_host(n.out[0], {cnt}, out, float_like(in[0]));
            return true;
        }
        if (op == "ReduceProd" || op == "ReduceSum" || op == "ReduceMax" || op == "ReduceMin") {
            std::vector<double> v = host_values(in[0]);
            if (in[0].dims.size() > 1) return false;
            double acc = op == "ReduceProd" ? 1.0 : op == "ReduceSum" ? 0.0 : op == "ReduceMax" ? -INFINITY : INFINITY;
            for (double e : v) acc = op == "ReduceProd" ? acc * e : op == "ReduceSum" ? acc + e : op == "ReduceMax" ? std::max(acc, e) : std::min(acc, e);
            set_host(n.out[0], n.ai("keepdims", 1) ? std::vector<int64_t>{1} : std::vector<int64_t>{}, {acc}, float_like(in[0]));
            return true;
        }
        return false;
    }

    // GridSample (UVDoc's final un-warp): X [N,C,H,W], grid [N,Ho,Wo,2] -> [N,C,Ho,Wo]
    void op_grid_sample(const GNode& n) {
        TInfo x = get(n.in[0]), g = get(n.in[1]);
        OAR_CHECK(x.dims.size() == 4 && g.dims.size() == 4 && g.dims[3] == 2 && g.dims[0] == x.dims[0], OAR_UNSUPPORTED_OP, "GridSample: 4-D input and [N,Ho,Wo,2] grid only");
        const std::string mode = n.as("mode", "linear"), pad = n.as("padding_mode", "zeros");
        const int imode = (mode == "linear" || mode == "bilinear") ? 0 : mode == "nearest" ? 1 : -1;
        const int ipad = pad == "zeros" ? 0 : pad == "border" ? 1 : pad == "reflection" ? 2 : -1;
        OAR_CHECK(imode >= 0 && ipad >= 0, OAR_UNSUPPORTED_OP, "GridSample: mode " + mode + " / padding_mode " + pad);
        const int align = (int)n.ai("align_corners", 0);
        Loc xin = to_clast_loc(x), gin = to_native_loc(g);
        const int64_t N = x.dims[0], C = x.dims[1], H = x.dims[2], W = x.dims[3], Ho = g.dims[1], Wo = g.dims[2];
        TInfo& y = new_out(n.out[0], {N, C, Ho, Wo}, Layout::CLAST);
        Loc yl = y.loc;
        step([=](const RunCtx& c) { k::grid_sample(c.s, c.at(xin), c.at(gin), c.mut(yl), (int)N, (int)H, (int)W, (int)C, (int)Ho, (int)Wo, imode, ipad, align); },
             8.0 * N * Ho * Wo * C, 4.0 * N * Ho * Wo * (2 + 5 * C));
    }

    // fused squeeze-excite gate (rewrite pass 6): pooled [n,C,1,1] -> [n,Cout,1,1]
    void op_se_gate(const GNode& n) {
        TInfo x = get(n.in[0]);
        OAR_CHECK(x.dims.size() == 4 && x.dims[2] == 1 && x.dims[3] == 1, OAR_SHAPE_MISMATCH, "SEGate: input must be [n, C, 1, 1]");
        const TInfo &w1 = get(n.in[1]), &w2 = get(n.in[3]);
        OAR_CHECK(w1.ht && w2.ht, OAR_UNSUPPORTED_OP, "SEGate: weights must be initializers");
        const int tiles = x.gap_tiles, gap_hw = x.gap_hw;   // > 0: the input still is the producing conv's tile sums
        const int64_t N = x.dims[0], C = tiles > 0 ? x.dims[1] / tiles : x.dims[1], Cmid = w1.ht->dims[0], Cout = w2.ht->dims[0];
        OAR_CHECK(w1.ht->dims[1] == C && w2.ht->dims[1] == Cmid, OAR_SHAPE_MISMATCH, "SEGate: weight shapes");
        const float* b1 = has_input(n, 2) ? get(n.in[2]).loc.cptr : nullptr;
        const float* b2 = has_input(n, 4) ? get(n.in[4]).loc.cptr : nullptr;
        const float* w1p = w1.loc.cptr;
        std::vector<float> w2t((size_t)(Cmid * Cout));   // [Cout][Cmid] -> [Cmid][Cout]: the kernel's threads run along Cout
        for (int64_t c = 0; c < Cout; ++c)
            for (int64_t j = 0; j < Cmid; ++j) w2t[(size_t)(j * Cout + c)] = w2.ht->f[(size_t)(c * Cmid + j)];
        const float* w2p = E.upload_const("segate_w2t:" + n.in[3], w2t);
        Act a1; a1.kind = (int)n.ai("act1", 0); a1.alpha = n.af("act1_alpha", 0.f); a1.beta = n.af("act1_beta", 0.f);
        Act a2 = n.act;
        Loc xin = x.loc;   // [n, C, 1, 1]: the same bytes in either layout
        TInfo& y = new_out(n.out[0], {N, Cout, 1, 1}, Layout::CLAST);
        Loc yl = y.loc;
        step([=](const RunCtx& c) { k::se_fc(c.s, c.at(xin), w1p, b1, a1, w2p, b2, a2, c.mut(yl), (int)N, (int)C, (int)Cmid, (int)Cout, tiles, gap_hw); },
             2.0 * N * Cmid * (C + Cout), 4.0 * (N * (C + Cout) + Cmid * (C + Cout)));
    }

    // Pad (UVDoc's reflect-padded convolutions export as Pad(reflect) + Conv).  Works on the physical layout: a
    // channels-last tensor is padded in place of a conversion (the pads are permuted with the dims).
    void op_pad(const GNode& n) {
        TInfo x = get(n.in[0]);
        const int r = (int)x.dims.size();
        std::vector<int64_t> pads = has_input(n, 1) ? get(n.in[1]).hv : n.ais("pads");
        std::vector<int64_t> axes;
        if (has_input(n, 3)) axes = get(n.in[3]).hv;
        std::vector<int64_t> before(r, 0), after(r, 0);
        if (axes.empty()) {
            OAR_CHECK((int)pads.size() == 2 * r, OAR_SHAPE_MISMATCH, "Pad: pads must have 2 * rank entries");
            for (int d = 0; d < r; ++d) { before[d] = pads[d]; after[d] = pads[r + d]; }
        } else {
            OAR_CHECK(pads.size() == 2 * axes.size(), OAR_SHAPE_MISMATCH, "Pad: pads must have 2 * len(axes) entries");
            for (size_t k = 0; k < axes.size(); ++k) { int64_t a = axes[k] < 0 ? axes[k] + r : axes[k]; before[a] = pads[k]; after[a] = pads[axes.size() + k]; }
        }
        const std::string mode = n.as("mode", "constant");
        const int imode = mode == "constant" ? 0 : mode == "reflect" ? 1 : mode == "edge" ? 2 : -1;
        OAR_CHECK(imode >= 0, OAR_UNSUPPORTED_OP, "Pad: mode " + mode);
        float value = n.af("value", 0.0f);
        if (has_input(n, 2)) { const TInfo& v = get(n.in[2]); if (v.ht && !v.ht->f.empty()) value = v.ht->f[0]; }
        std::vector<int64_t> od(r);
        for (int d = 0; d < r; ++d) {
            od[d] = x.dims[d] + before[d] + after[d];
            OAR_CHECK(od[d] >= 0, OAR_SHAPE_MISMATCH, "Pad: negative output dimension");
            OAR_CHECK(imode != 1 || (before[d] < std::max<int64_t>(x.dims[d], 1) && after[d] < std::max<int64_t>(x.dims[d], 1)) || x.dims[d] == 1, OAR_UNSUPPORTED_OP,
                      "Pad: reflect padding wider than the axis");
        }
        const bool clast = x.layout == Layout::CLAST && r >= 3;
        std::vector<int64_t> pin = clast ? clast_phys_dims(x.dims) : x.dims, pout = clast ? clast_phys_dims(od) : od, pbef = clast ? clast_phys_dims(before) : before;
        Loc xin = x.loc;
        TInfo& y = new_out(n.out[0], od, clast ? Layout::CLAST : Layout::NATIVE);
        Loc yl = y.loc;
        step([=](const RunCtx& c) { k::pad_nd(c.s, c.at(xin), c.mut(yl), r, pin.data(), pout.data(), pbef.data(), imode, value); }, 0, 8.0 * numel(od));
    }

    void op_gap(const GNode& n) {
        TInfo x = get(n.in[0]);
        OAR_CHECK(x.dims.size() == 4, OAR_UNSUPPORTED_OP, "GlobalAveragePool: rank-4 only");
        Loc xin = to_clast_loc(x);
        int64_t N = x.dims[0], C = x.dims[1], HW = x.dims[2] * x.dims[3];
        TInfo& y = new_out(n.out[0], {N, C, 1, 1}, Layout::CLAST);
        Loc yl = y.loc;
        const int splits = k::global_avgpool_splits((int)N, (int)HW, (int)C);
        Loc part;
        if (splits > 1) part = alloc_temp((size_t)N * splits * C * sizeof(float));
        step([=](const RunCtx& c) { k::global_avgpool(c.s, c.at(xin), c.mut(yl), (int)N, (int)HW, (int)C, splits > 1 ? c.mut(part) : nullptr); }, 0, 4.0 * numel(x.dims));
    }

    void op_pool(const GNode& n, bool is_max) {
        TInfo x = get(n.in[0]);
        OAR_CHECK(x.dims.size() == 4, OAR_UNSUPPORTED_OP, "Pool: rank-4 only");
        auto ks = n.ais("kernel_shape"), st = n.ais("strides");
        OAR_CHECK(ks.size() == 2, OAR_UNSUPPORTED_OP, "Pool: kernel_shape");
        int64_t kh = ks[0], kw = ks[1], sh = st.size() == 2 ? st[0] : 1, sw = st.size() == 2 ? st[1] : 1;
        int64_t pt, pl, pb, pr;
        get_pads(n, x.dims[2], x.dims[3], kh, kw, sh, sw, 1, 1, pt, pl, pb, pr);
        bool ceil_mode = n.ai("ceil_mode", 0) != 0;
        auto od = [&](int64_t in, int64_t k_, int64_t s_, int64_t p0, int64_t p1) {
            int64_t t = in + p0 + p1 - k_;
            int64_t o = (ceil_mode ? (t + s_ - 1) / s_ : t / s_) + 1;
            if (ceil_mode && (o - 1) * s_ >= in + p0) --o;
            return o;
        };
        int64_t Ho = od(x.dims[2], kh, sh, pt, pb), Wo = od(x.dims[3], kw, sw, pl, pr);
        Loc xin = to_clast_loc(x);
        TInfo& y = new_out(n.out[0], {x.dims[0], x.dims[1], Ho, Wo}, Layout::CLAST);
        k::PoolP p{};
        p.N = (int)x.dims[0]; p.H = (int)x.dims[2]; p.W = (int)x.dims[3]; p.C = (int)x.dims[1]; p.Ho = (int)Ho; p.Wo = (int)Wo;
        p.kh = (int)kh; p.kw = (int)kw; p.sh = (int)sh; p.sw = (int)sw; p.pt = (int)pt; p.pl = (int)pl; p.pb = (int)pb; p.pr = (int)pr;
        p.is_max = is_max; p.count_include_pad = (int)n.ai("count_include_pad", 0);
        Loc yl = y.loc;
        auto run = [=](const RunCtx& c) { k::PoolP q = p; q.x = c.at(xin); q.y = c.mut(yl); k::pool2d(c.s, q); };
        const double bytes = 4.0 * (numel(x.dims) + numel(y.dims));
        // round 5: an average pool that leaves ONE row of tokens per sample (the recognizer's [6, 2] pool in front of the SVTR neck) only
        // reads its own sample: as the first operator of a sample-local chain (chain.hip CH_POOL) it costs no launch.  OAR_CHAIN_POOL=0 keeps it apart.
        const char* cpe = getenv("OAR_CHAIN_POOL");
        if (!(cpe && cpe[0] == '0') && !is_max && !ceil_mode && Ho == 1 && kh == x.dims[2] && kw == sw && kw * Wo <= x.dims[3] && x.dims[3] - kw * Wo < kw && pt == 0 && pl == 0 && pb == 0 && pr == 0 &&
            (p.C & 3) == 0 && kh * kw <= 64 && chain_loc_ok(xin)) {
            ChainRec r;
            r.d.type = k::CH_POOL; r.d.K = (int)kh; r.d.cin = (int)kw; r.d.pad = (int)x.dims[3]; r.d.N = p.C; r.d.in_ld = p.C; r.d.out_ld = p.C;
            r.in = xin; r.out = yl; r.rows = x.dims[0] * Wo; r.fix_n = x.dims[0]; r.fix_T = Wo; r.out_root = n.out[0];
            step_chainable(run, std::move(r), 0, bytes);
        } else {
            step(run, 0, bytes);
        }
    }

    void op_resize(const GNode& n) {
        TInfo x = get(n.in[0]);
        OAR_CHECK(x.dims.size() == 4, OAR_UNSUPPORTED_OP, "Resize: rank-4 only");
        int64_t H = x.dims[2], W = x.dims[3], Ho, Wo;
        float sh, sw;
        if (has_input(n, 3)) {
            const TInfo& sz = get(n.in[3]);
            OAR_CHECK(sz.host_int && sz.hv.size() == 4, OAR_UNSUPPORTED_OP, "Resize: sizes must be a host int tensor");
            Ho = sz.hv[2]; Wo = sz.hv[3]; sh = (float)Ho / (float)H; sw = (float)Wo / (float)W;
        } else {
            OAR_CHECK(has_input(n, 2), OAR_UNSUPPORTED_OP, "Resize: neither scales nor sizes");
            const TInfo& sc = get(n.in[2]);
            std::vector<double> scv = host_values(sc);   // an initializer, or computed from Shape on the host
            OAR_CHECK(host_evaluable(sc) && scv.size() == 4, OAR_UNSUPPORTED_OP, "Resize: scales must be 4 floats known at plan time");
            sh = (float)scv[2]; sw = (float)scv[3];
            Ho = (int64_t)std::floor((float)H * sh); Wo = (int64_t)std::floor((float)W * sw);
        }
        std::string mode = n.as("mode", "nearest"), ctm = n.as("coordinate_transformation_mode", "half_pixel"), nm = n.as("nearest_mode", "round_prefer_floor");
        int imode = mode == "nearest" ? 0 : 1;
        OAR_CHECK(mode == "nearest" || mode == "linear", OAR_UNSUPPORTED_OP, "Resize: mode " + mode);
        int ictm = ctm == "asymmetric" ? 0 : ctm == "half_pixel" ? 1 : ctm == "align_corners" ? 2 : ctm == "pytorch_half_pixel" ? 3 : -1;
        OAR_CHECK(ictm >= 0, OAR_UNSUPPORTED_OP, "Resize: coordinate_transformation_mode " + ctm);
        int inm = nm == "floor" ? 0 : nm == "round_prefer_floor" ? 1 : nm == "round_prefer_ceil" ? 2 : 3;
        static const bool defer_on = [] { const char* e = getenv("OAR_DEFER_RESIZE"); return !e || atoi(e) != 0; }();
        if (defer_on && imode == 0 && x.layout == Layout::CLAST && sole_consumer.count(n.out[0])) {
            PendingResize r;
            r.xin = x.loc; r.dims = {x.dims[0], x.dims[1], Ho, Wo};
            r.N = (int)x.dims[0]; r.H = (int)H; r.W = (int)W; r.C = (int)x.dims[1]; r.Ho = (int)Ho; r.Wo = (int)Wo;
            r.imode = imode; r.ictm = ictm; r.inm = inm; r.sh = sh; r.sw = sw;
            // integer-factor upsampling whose index map is o / f: checked on the map itself, whatever the attributes say
            auto factor = [&](int in, int out, float scale) {
                if (in <= 0 || out % in != 0) return 0;
                const int f = out / in;
                for (int o = 0; o < out; ++o) if (k::resize_nearest_index(o, scale, in, out, ictm, inm) != o / f) return 0;
                return f;
            };
            r.fh = factor(r.H, r.Ho, sh); r.fw = factor(r.W, r.Wo, sw);
            if (!r.fh || !r.fw) r.fh = r.fw = 0;
            pending_resize[n.out[0]] = r;
            TInfo t;   // known shape, no storage yet
            t.dims = r.dims; t.layout = Layout::CLAST; t.root = "";
            vals[n.out[0]] = t;
            return;
        }
        Loc xin = to_clast_loc(x);
        TInfo& y = new_out(n.out[0], {x.dims[0], x.dims[1], Ho, Wo}, Layout::CLAST);
        Loc yl = y.loc;
        int N = (int)x.dims[0], C = (int)x.dims[1];
        step([=](const RunCtx& c) { k::resize(c.s, c.at(xin), c.mut(yl), N, (int)H, (int)W, C, (int)Ho, (int)Wo, sh, sw, imode, ictm, inm, C); }, 0,
             4.0 * (numel(x.dims) + numel(y.dims)));
    }

    void op_concat(const GNode& n) {
        // deferred resizes among the inputs: absorbed when this is a channel concat of channels-last tensors with float4-aligned
        // slots, run on the spot otherwise
        bool absorb = false;
        if (!pending_resize.empty()) {
            int64_t axis0 = n.ai("axis", 0);
            bool ok = true, any = false;
            int64_t total_c = 0;
            for (auto& s : n.in) {
                const PendingResize* pr = peek_pending(s);
                any = any || pr;
                auto it = vals.find(s);
                if (it == vals.end() || it->second.dims.size() != 4 || it->second.layout != Layout::CLAST || it->second.host_int) { ok = false; break; }
                if ((it->second.dims[1] & 3) != 0) ok = false;
                total_c += it->second.dims[1];
            }
            absorb = any && ok && (axis0 == 1 || axis0 == -3) && (total_c & 3) == 0;
            if (!absorb) for (auto& s : n.in) materialise_pending(s);
        }
        std::vector<TInfo> xs;
        for (auto& s : n.in) { auto it = vals.find(s); xs.push_back(absorb && peek_pending(s) ? it->second : get(s)); }
        bool all_host = true, any_host = false, any_f = false;
        for (auto& t : xs) { all_host = all_host && host_evaluable(t) && t.dims.size() <= 1; any_host = any_host || t.host_int; any_f = any_f || float_like(t); }
        if (all_host && any_host) {   // 1-D shape / scale vectors
            std::vector<double> v;
            for (auto& t : xs) { std::vector<double> e = host_values(t); v.insert(v.end(), e.begin(), e.end()); }
            set_host(n.out[0], {(int64_t)v.size()}, v, any_f);
            return;
        }
        int r = (int)xs[0].dims.size();
        int64_t axis = n.ai("axis", 0);
        if (axis < 0) axis += r;
        bool clast = false;
        for (auto& t : xs) clast = clast || (t.layout == Layout::CLAST);
        std::vector<int64_t> od = xs[0].dims;
        od[axis] = 0;
        for (auto& t : xs) od[axis] += t.dims[axis];
        if (clast && r >= 3) {
            // physical axis of the logical axis
            int pax = axis == 0 ? 0 : (axis == 1 ? r - 1 : (int)axis - 1);
            std::vector<int64_t> pod = clast_phys_dims(od);
            int64_t outer = 1, inner = 1;
            for (int i = 0; i < pax; ++i) outer *= pod[i];
            for (int i = pax + 1; i < r; ++i) inner *= pod[i];
            std::vector<Loc> ins;
            for (size_t i = 0; i < xs.size(); ++i) ins.push_back(absorb && peek_pending(n.in[i]) ? Loc() : to_clast_loc(xs[i]));
            int64_t coff = 0, total = pod[pax] * inner;
            // round 5: every absorbed resize an integer-factor nearest upsampling (the DB neck: up8 / up4 / up2 of three pyramid levels next to
            // the finest one) -> the whole concat is ONE gather launch instead of one launch per input.  OAR_CONCAT_GATHER=0 restores those.
            // round 6: the same launch for a plain channel concat of three or more maps (PP-HGNetV2's blocks concatenate seven: one gather instead of seven
            // strided copies)
            bool plain = !absorb && r == 4 && inner == 1 && xs.size() >= 3 && (od[1] & 3) == 0;
            for (auto& t : xs) plain = plain && t.dims.size() == 4 && (t.dims[1] & 3) == 0;
            // (round 6, later) ... or no launch at all when the only reader is a 1x1 convolution that can read the sources itself (pending_concat above)
            static const bool defer_on = [] { const char* e = getenv("OAR_CONCAT_DEFER"); return !(e && e[0] == '0'); }();
            if (getenv("OAR_DEBUG_CONCAT")) fprintf(stderr, "concat %s: plain=%d absorb=%d r=%d inner=%ld n=%zu consumer=%d\n", n.out[0].c_str(), (int)plain, (int)absorb, r, (long)inner, xs.size(), concat_consumer.count(n.out[0]) ? concat_consumer[n.out[0]] : -1);
            if (plain && defer_on && xs.size() <= 8 && concat_consumer.count(n.out[0])) {
                const GNode& cn = E.nodes_[concat_consumer[n.out[0]]];
                auto wi = E.inits_.find(cn.in.size() > 1 ? cn.in[1] : std::string());
                auto attr_ok = [&](const char* a, int64_t want) { auto v = cn.ais(a); for (auto e : v) if (e != want) return false; return true; };
                bool ok = wi != E.inits_.end() && wi->second.dims.size() == 4 && wi->second.dims[2] == 1 && wi->second.dims[3] == 1 && wi->second.dims[1] == od[1] && cn.ai("group", 1) == 1 &&
                          attr_ok("strides", 1) && attr_ok("pads", 0) && attr_ok("dilations", 1) && cn.residual.empty() && !(cn.in.size() > 3 && !cn.in[3].empty()) &&
                          cn.as("auto_pad", "NOTSET") == "NOTSET";
                for (auto& t : xs) ok = ok && t.layout == Layout::CLAST && (t.dims[1] & 7) == 0 && t.dims[0] == od[0] && t.dims[2] == od[2] && t.dims[3] == od[3] && !t.host_int;   // (channels-last already: a converted copy would be a temporary of THIS node)
                if (getenv("OAR_DEBUG_CONCAT")) {
                    fprintf(stderr, "  ok before msrc check = %d (K=%ld N=%ld) res=%d gate=%d autopad=%s\n", (int)ok, (long)od[1], wi != E.inits_.end() ? (long)wi->second.dims[0] : -1L, (int)!cn.residual.empty(), (int)(cn.in.size() > 3 && !cn.in[3].empty()), cn.as("auto_pad", "NOTSET").c_str());
                    for (auto& t : xs) fprintf(stderr, "    src layout=%d C=%ld host=%d\n", (int)(t.layout == Layout::CLAST), (long)t.dims[1], (int)t.host_int);
                }
                ok = ok && k::conv_msrc_ok((long)(od[0] * od[2] * od[3]), (int)od[1], (int)wi->second.dims[0]) && (wi->second.dims[0] & 3) == 0;
                if (ok) {
                    PendingConcat pc;
                    for (size_t i = 0; i < xs.size(); ++i) { pc.src.push_back(ins[i]); pc.c.push_back((int)xs[i].dims[1]); }
                    pc.dims = od;
                    TInfo t;
                    t.dims = od; t.layout = Layout::CLAST;
                    vals[n.out[0]] = t;          // shape only: no storage unless somebody materialises it
                    pending_concat[n.out[0]] = std::move(pc);
                    return;
                }
            }
            TInfo& y = new_out(n.out[0], od, Layout::CLAST);   // (below the deferral: a deferred concat owns no storage)
            Loc yl = y.loc;
            if ((absorb || plain) && r == 4 && inner == 1 && xs.size() <= 8) {
                static const bool gather_on = [] { const char* e = getenv("OAR_CONCAT_GATHER"); return !(e && e[0] == '0'); }();
                bool ok = gather_on;
                for (size_t i = 0; ok && i < xs.size(); ++i) {
                    const PendingResize* pr = peek_pending(n.in[i]);
                    if (pr) ok = pr->fh > 0 && pr->fw > 0 && pr->imode == 0 && pr->Ho == od[2] && pr->Wo == od[3] && pr->H * pr->fh == pr->Ho && pr->W * pr->fw == pr->Wo && pr->N == od[0];
                    else ok = xs[i].dims[0] == od[0] && xs[i].dims[2] == od[2] && xs[i].dims[3] == od[3];
                }
                if (ok) {
                    k::ConcatGatherP gp{};
                    std::vector<Loc> src(xs.size());
                    gp.n_src = (int)xs.size(); gp.N = (int)od[0]; gp.Ho = (int)od[2]; gp.Wo = (int)od[3]; gp.C = (int)od[1];
                    double bytes = 4.0 * (double)numel(od);
                    for (size_t i = 0; i < xs.size(); ++i) {
                        const PendingResize* pr = peek_pending(n.in[i]);
                        src[i] = pr ? pr->xin : ins[i];
                        gp.c[i] = (int)xs[i].dims[1]; gp.off[i] = (int)coff; gp.fh[i] = pr ? pr->fh : 1; gp.fw[i] = pr ? pr->fw : 1;
                        bytes += 4.0 * (double)numel(xs[i].dims) / (gp.fh[i] * gp.fw[i]);
                        coff += xs[i].dims[1];
                    }
                    for (auto& nm : n.in) pending_resize.erase(nm);
                    step([=](const RunCtx& c) {
                        k::ConcatGatherP q = gp;
                        for (int i = 0; i < q.n_src; ++i) q.x[i] = c.at(src[(size_t)i]);
                        k::concat_gather(c.s, q, c.mut(yl));
                    }, 0, bytes);
                    return;
                }
            }
            for (size_t i = 0; i < xs.size(); ++i) {
                int64_t w = xs[i].dims[axis] * inner;
                Loc il = ins[i];
                int64_t off = coff;
                if (absorb && peek_pending(n.in[i])) {   // the resize writes its pixels' channel group in place
                    run_resize(*peek_pending(n.in[i]), yl, off, (int)total);
                    pending_resize.erase(n.in[i]);
                } else {
                    concat_copy_step(il, yl, off, outer, w, total, n.out[0]);
                }
                coff += w;
            }
            return;
        }
        int64_t outer = 1, inner = 1;
        for (int i = 0; i < axis; ++i) outer *= od[i];
        for (int i = (int)axis + 1; i < r; ++i) inner *= od[i];
        std::vector<Loc> ins;
        for (auto& t : xs) ins.push_back(to_native_loc(t));
        TInfo& y = new_out(n.out[0], od, Layout::NATIVE);
        Loc yl = y.loc;
        int64_t coff = 0, total = od[axis] * inner;
        for (size_t i = 0; i < xs.size(); ++i) {
            int64_t w = xs[i].dims[axis] * inner;
            Loc il = ins[i];
            int64_t off = coff;
            concat_copy_step(il, yl, off, outer, w, total, n.out[0]);
            coff += w;
        }
    }
    // one input of a Concat: `outer` rows of w floats into columns [off, off + w) of rows of `total` floats
    void concat_copy_step(Loc il, Loc yl, int64_t off, int64_t outer, int64_t w, int64_t total, const std::string& out_name) {
        auto run = [=](const RunCtx& c) { k::copy2d(c.s, c.at(il), c.mut(yl) + off, outer, (int)w, (int)w, (int)total); };
        if ((w & 3) == 0 && (total & 3) == 0 && (off & 3) == 0 && chain_loc_ok(il) && yl.kind == Loc::ARENA) {
            ChainRec r;
            r.d.type = k::CH_COPY; r.d.N = (int)w; r.d.in_ld = (int)w; r.d.out_ld = (int)total;
            r.in = il; r.out = yl; r.out.off += off * 4; r.rows = outer; r.out_root = out_name;
            step_chainable(run, std::move(r), 0, 8.0 * outer * w);
        } else {
            step(run, 0, 8.0 * outer * w);
        }
    }

    // Reshape-like ops produce views of the native buffer.
    void view_native(const GNode& n, const TInfo& x, const std::vector<int64_t>& od) {
        OAR_CHECK(numel(od) == numel(x.dims), OAR_SHAPE_MISMATCH, "reshape: element count mismatch at " + n.out[0]);
        if (x.layout == Layout::CLAST) {
            int r = (int)x.dims.size();
            int64_t sp = 1;
            for (int i = 2; i < r; ++i) sp *= x.dims[i];
            // [n, C, spatial...] -> [n, C, other spatial grouping] (SVTRv2's map <-> token moves: Reshape [n,C,H,W] -> [n,C,H*W] in front of a
            // Transpose to [n,H*W,C], and back): channels-last keeps the spatial positions in row-major order whatever their grouping -- a view
            if (od.size() >= 3 && od[0] == x.dims[0] && od[1] == x.dims[1] && !x.ht) { alias_out(n.out[0], x, od, Layout::CLAST); return; }
            if (x.dims[1] != 1 && sp != 1) {  // genuine transpose needed
                TInfo& y = new_out(n.out[0], od, Layout::NATIVE);
                to_native_loc(x, y.loc);
                return;
            }
        }
        alias_out(n.out[0], x, od, Layout::NATIVE);
    }

    void op_reshape(const GNode& n) {
        TInfo x = get(n.in[0]);
        const TInfo& sh = get(n.in[1]);
        OAR_CHECK(sh.host_int, OAR_UNSUPPORTED_OP, "Reshape: shape must be known on the host");
        if (x.host_int) { TInfo o = x; o.dims.assign(sh.hv.begin(), sh.hv.end()); vals[n.out[0]] = o; return; }
        std::vector<int64_t> od(sh.hv.size());
        int64_t known = 1; int infer = -1;
        for (size_t i = 0; i < od.size(); ++i) {
            int64_t v = sh.hv[i];
            if (v == 0 && n.ai("allowzero", 0) == 0) v = x.dims[i];
            if (v == -1) { infer = (int)i; v = 1; }
            od[i] = v; known *= v;
        }
        if (infer >= 0) od[infer] = numel(x.dims) / std::max<int64_t>(known, 1);
        view_native(n, x, od);
    }

    void op_flatten(const GNode& n) {
        TInfo x = get(n.in[0]);
        int64_t ax = n.ai("axis", 1);
        if (ax < 0) ax += (int64_t)x.dims.size();
        int64_t a = 1, b = 1;
        for (int i = 0; i < (int)x.dims.size(); ++i) (i < ax ? a : b) *= x.dims[i];
        view_native(n, x, {a, b});
    }

    std::vector<int64_t> axes_of(const GNode& n, size_t input_idx) {
        if (has_input(n, input_idx)) { const TInfo& a = get(n.in[input_idx]); OAR_CHECK(a.host_int, OAR_UNSUPPORTED_OP, "axes must be host ints"); return a.hv; }
        return n.ais("axes");
    }

    void op_squeeze(const GNode& n) {
        TInfo x = get(n.in[0]);
        std::vector<int64_t> axes = axes_of(n, 1);
        int r = (int)x.dims.size();
        std::set<int> ax;
        if (axes.empty()) { for (int i = 0; i < r; ++i) if (x.dims[i] == 1) ax.insert(i); }
        for (auto a : axes) ax.insert((int)(a < 0 ? a + r : a));
        std::vector<int64_t> od;
        for (int i = 0; i < r; ++i) if (!ax.count(i)) od.push_back(x.dims[i]);
        if (x.host_int) { TInfo o = x; o.dims = od; vals[n.out[0]] = o; return; }
        // channels-last stays channels-last when only spatial axes are removed and rank stays >= 3
        if (x.layout == Layout::CLAST && !ax.count(0) && !ax.count(1) && od.size() >= 3) { alias_out(n.out[0], x, od, Layout::CLAST); return; }
        view_native(n, x, od);
    }

    void op_unsqueeze(const GNode& n) {
        TInfo x = get(n.in[0]);
        std::vector<int64_t> axes = axes_of(n, 1);
        int r = (int)x.dims.size() + (int)axes.size();
        std::set<int> ax;
        for (auto a : axes) ax.insert((int)(a < 0 ? a + r : a));
        std::vector<int64_t> od;
        size_t j = 0;
        for (int i = 0; i < r; ++i) od.push_back(ax.count(i) ? 1 : x.dims[j++]);
        if (x.host_int) { TInfo o = x; o.dims = od; vals[n.out[0]] = o; return; }
        if (x.layout == Layout::CLAST && !ax.count(0) && !ax.count(1) && r <= 5) { alias_out(n.out[0], x, od, Layout::CLAST); return; }
        view_native(n, x, od);
    }

    void op_transpose(const GNode& n) {
        TInfo x = get(n.in[0]);
        int r = (int)x.dims.size();
        std::vector<int64_t> perm = n.ais("perm");
        if (perm.empty()) for (int i = r - 1; i >= 0; --i) perm.push_back(i);
        std::vector<int64_t> od(r);
        for (int i = 0; i < r; ++i) od[i] = x.dims[perm[i]];
        // lazy channels-last moves: [n,C,s..] <-> [n,s..,C]
        bool to_last = r >= 3 && perm[0] == 0 && perm[r - 1] == 1;
        if (to_last) for (int i = 1; i < r - 1; ++i) to_last = to_last && perm[i] == i + 1;
        bool from_last = r >= 3 && perm[0] == 0 && perm[1] == r - 1;
        if (from_last) for (int i = 2; i < r; ++i) from_last = from_last && perm[i] == i - 1;
        if (x.layout == Layout::CLAST && to_last) { alias_out(n.out[0], x, od, Layout::NATIVE); return; }
        if (x.layout == Layout::NATIVE && from_last && r <= 5 && !x.ht) { alias_out(n.out[0], x, od, Layout::CLAST); return; }
        Loc xin = to_native_loc(x);
        std::vector<int64_t> ns = contig_strides(x.dims), is(r);
        for (int i = 0; i < r; ++i) is[i] = ns[perm[i]];
        TInfo& y = new_out(n.out[0], od, Layout::NATIVE);
        Loc yl = y.loc;
        step([=](const RunCtx& c) { k::permute(c.s, c.at(xin), c.mut(yl), r, od.data(), is.data()); }, 0, 8.0 * numel(od));
    }

    // strided sub-tensor copy (or alias when the slice is a contiguous prefix-dim block)
    void slice_out(const std::string& out, const TInfo& x, Loc xin, int axis, int64_t start, int64_t len) {
        std::vector<int64_t> od = x.dims;
        od[axis] = len;
        std::vector<int64_t> ns = contig_strides(x.dims);
        bool outer_all_one = true;
        for (int i = 0; i < axis; ++i) outer_all_one = outer_all_one && x.dims[i] == 1;
        if (outer_all_one) {
            TInfo tmp = x; tmp.loc = xin; tmp.layout = Layout::NATIVE;
            if (xin.kind != x.loc.kind || xin.off != x.loc.off) tmp.root = "";  // temp buffers cannot be aliased safely
            if (!tmp.root.empty() || xin.kind == Loc::CONST) { alias_out(out, tmp, od, Layout::NATIVE, start * ns[axis] * 4); return; }
        }
        TInfo& y = new_out(out, od, Layout::NATIVE);
        Loc yl = y.loc;
        int r = (int)od.size();
        int64_t off = start * ns[axis];
        step([=](const RunCtx& c) { k::permute(c.s, c.at(xin) + off, c.mut(yl), r, od.data(), ns.data()); }, 0, 8.0 * numel(od));
    }

    void op_split(const GNode& n) {
        TInfo x = get(n.in[0]);
        int r = (int)x.dims.size();
        int64_t axis = n.ai("axis", 0);
        if (axis < 0) axis += r;
        std::vector<int64_t> parts;
        if (has_input(n, 1)) parts = get(n.in[1]).hv;
        else if (n.has("split")) parts = n.ais("split");
        else parts.assign(n.out.size(), x.dims[axis] / (int64_t)n.out.size());
        Loc xin = to_native_loc(x);
        int64_t start = 0;
        for (size_t i = 0; i < n.out.size(); ++i) { slice_out(n.out[i], x, xin, (int)axis, start, parts[i]); start += parts[i]; }
    }

    void op_slice(const GNode& n) {
        TInfo x = get(n.in[0]);
        std::vector<int64_t> starts = get(n.in[1]).hv, ends = get(n.in[2]).hv, axes, steps;
        if (has_input(n, 3)) axes = get(n.in[3]).hv; else for (size_t i = 0; i < starts.size(); ++i) axes.push_back((int64_t)i);
        if (has_input(n, 4)) steps = get(n.in[4]).hv; else steps.assign(starts.size(), 1);
        int r = (int)x.dims.size();
        if (x.host_int) {
            OAR_CHECK(r == 1 && axes.size() == 1 && steps[0] == 1, OAR_UNSUPPORTED_OP, "Slice: host tensors support 1-D unit-step only");
            int64_t d = x.dims[0], s = starts[0] < 0 ? starts[0] + d : starts[0], e = ends[0] < 0 ? ends[0] + d : ends[0];
            s = std::min(std::max<int64_t>(s, 0), d); e = std::min(std::max<int64_t>(e, 0), d);
            std::vector<double> xv = host_values(x);
            set_host(n.out[0], {std::max<int64_t>(e - s, 0)}, std::vector<double>(xv.begin() + s, xv.begin() + std::max(s, e)), x.host_f);
            return;
        }
        OAR_CHECK(axes.size() == 1 && steps[0] == 1, OAR_UNSUPPORTED_OP, "Slice: single axis, unit step only");
        int64_t ax = axes[0] < 0 ? axes[0] + r : axes[0], d = x.dims[ax];
        int64_t s = starts[0] < 0 ? starts[0] + d : starts[0], e = ends[0] < 0 ? ends[0] + d : ends[0];
        s = std::min(std::max<int64_t>(s, 0), d); e = std::min(std::max<int64_t>(e, 0), d);
        Loc xin = to_native_loc(x);
        slice_out(n.out[0], x, xin, (int)ax, s, std::max<int64_t>(e - s, 0));
    }

    void op_gather(const GNode& n) {
        TInfo x = get(n.in[0]);
        const TInfo& idx = get(n.in[1]);
        OAR_CHECK(idx.host_int, OAR_UNSUPPORTED_OP, "Gather: indices must be host ints");
        int64_t axis = n.ai("axis", 0);
        if (x.host_int) {
            OAR_CHECK(x.dims.size() <= 1, OAR_UNSUPPORTED_OP, "Gather: host data rank > 1");
            std::vector<double> xv = host_values(x), ov;
            for (auto i : idx.hv) {
                const int64_t j = i < 0 ? i + (int64_t)xv.size() : i;
                OAR_CHECK(j >= 0 && j < (int64_t)xv.size(), OAR_SHAPE_MISMATCH, "Gather: index out of range at " + n.out[0]);
                ov.push_back(xv[(size_t)j]);
            }
            set_host(n.out[0], idx.dims, ov, x.host_f);
            return;
        }
        int r = (int)x.dims.size();
        if (axis < 0) axis += r;
        OAR_CHECK(idx.hv.size() == 1, OAR_UNSUPPORTED_OP, "Gather: only a single index is supported on device tensors");
        int64_t i = idx.hv[0] < 0 ? idx.hv[0] + x.dims[axis] : idx.hv[0];
        Loc xin = to_native_loc(x);
        std::string tmpn = n.out[0] + "::gather_slice";
        slice_out(tmpn, x, xin, (int)axis, i, 1);
        TInfo s = vals[tmpn];
        std::vector<int64_t> od;
        for (int d = 0; d < r; ++d) if (d != axis || !idx.dims.empty()) od.push_back(d == axis ? 1 : x.dims[d]);
        alias_out(n.out[0], s, od, Layout::NATIVE);
        adopt_root(n.out[0], tmpn);   // (when the slice was a copy: the bytes are this node's output)
    }

    // true when `name` is consumed by nothing but the Softmax that produces graph output 0
    bool feeds_final_softmax(const std::string& name) {
        int users = 0; bool ok = false;
        for (const GNode& m : E.nodes_)
            for (const std::string& in : m.in)
                if (in == name) { ++users; ok = m.op == "Softmax" && !m.out.empty() && m.out[0] == E.output_names_[0]; }
        for (const std::string& on : E.output_names_) if (on == name) return false;
        return users == 1 && ok;
    }
    void op_linear(const GNode& n, bool gemm) {
        TInfo a = get(n.in[0]);
        const TInfo& bt = get(n.in[1]);
        OAR_CHECK(bt.ht, OAR_UNSUPPORTED_OP, "Linear: B must be an initializer");
        OAR_CHECK(bt.ht->dims.size() == 2 && !a.dims.empty(), OAR_MODEL_LOAD, "Linear: B must be a rank-2 initializer at " + n.out[0]);
        bool transB = gemm && n.ai("transB", 0) != 0;
        OAR_CHECK(!gemm || n.ai("transA", 0) == 0, OAR_UNSUPPORTED_OP, "Gemm: transA");
        int64_t K = transB ? bt.ht->dims[1] : bt.ht->dims[0], N = transB ? bt.ht->dims[0] : bt.ht->dims[1];
        OAR_CHECK(a.dims.back() == K, OAR_SHAPE_MISMATCH, "Linear: inner dimension mismatch at " + n.out[0]);
        Loc ain = to_native_loc(a);
        int64_t M = numel(a.dims) / K;
        std::vector<int64_t> od = a.dims;
        od.back() = N;
        const float* bias = nullptr;
        float alpha = gemm ? n.af("alpha", 1.0f) : 1.0f;
        if (!n.bias.empty()) { const TInfo& bb = get(n.bias); OAR_CHECK(bb.ht && (int64_t)bb.ht->f.size() == N, OAR_MODEL_LOAD, "Linear: bias must have N elements at " + n.out[0]); bias = bb.loc.cptr; }
        if (gemm && has_input(n, 2)) {
            const TInfo& c = get(n.in[2]);
            OAR_CHECK(c.ht && numel(c.dims) == N, OAR_UNSUPPORTED_OP, "Gemm: C must be a constant of N elements");
            const float beta = n.af("beta", 1.0f);
            if (beta == 1.0f) bias = c.loc.cptr;
            else if (beta != 0.0f) {   // beta * C folded into a constant of its own (round 6)
                std::vector<float> bc(c.ht->f);
                for (auto& v : bc) v *= beta;
                bias = E.upload_const("gemm_beta_c:" + n.out[0], bc);
            }
        }
        Loc res;
        if (!n.residual.empty()) { TInfo r = get(n.residual); OAR_CHECK(numel(r.dims) == M * N, OAR_SHAPE_MISMATCH, "Linear: residual shape"); res = to_native_loc(r); }
        bool has_res = res.kind != Loc::NONE;
        // Logits that only feed the fused softmax+argmax tail are produced with their row padded to a multiple of 16
        // channels (zero weight rows, zero bias): every store is a full float4 and the wide layer qualifies for the
        // weight-stationary kernel; the tail reads the first N columns of each row (Plan::logits_valid).
        int64_t Np = N;
        if (skip_final_softmax && !has_res && K % 4 == 0 && alpha == 1.0f && (N & 15) != 0 && feeds_final_softmax(n.out[0])) {
            Np = (N + 15) / 16 * 16;
            od.back() = Np;
            P.logits_valid = (int)N;
            if (bias) {
                std::string key = "biaspad:" + n.out[0];
                auto it = E.dev_consts_.find(key);
                if (it == E.dev_consts_.end()) {
                    const HostTensor* hb = !n.bias.empty() ? get(n.bias).ht : get(n.in[2]).ht;
                    OAR_CHECK(hb, OAR_INTERNAL, "Linear: bias initializer missing");
                    std::vector<float> bp((size_t)Np, 0.f);
                    for (int64_t i = 0; i < N; ++i) bp[(size_t)i] = hb->f[(size_t)i];
                    bias = E.upload_const(key, bp);
                } else bias = it->second;
            }
        }
        TInfo& y = new_out(n.out[0], od, Layout::NATIVE);
        Loc yl = y.loc;
        Act act = n.act;
        double flops = 2.0 * M * N * K, bytes = 4.0 * (M * K + M * N * (has_res ? 2 : 1) + K * N);
        if (K % 4 == 0 && alpha == 1.0f) {
            // (a CTC head with a short K runs on the output-stationary bf16x6 kernel, ctc_head_x6.hip: bf16x6 fragments whatever igemm_weight_format says)
            const bool ctc_head = P.logits_valid > 0 && od.back() == Np && n.act.kind == k::ACT_NONE && !has_res && k::ctc_head_x6_supported((long)M, (int)K, (int)Np) &&
                                  k::ctc_partials_supported_x6((int)K) && [] { const char* e = getenv("OAR_CTC_PARTIALS"); return !e || atoi(e) != 0; }();
            const int fmt = ctc_head ? k::IGEMM_W_X6 : k::igemm_weight_format((long)M, (int)K, (int)Np, true);
            const float* w = linear_weight(n.in[1], *bt.ht, transB, fmt);
            k::ConvP p{};
            p.w_fmt = fmt;
            N = Np;
            // CTC head: when the logits only feed the fused tail, do not even write them -- per-tile softmax partials
            Loc part;
            if (P.logits_valid > 0 && od.back() == Np && act.kind == k::ACT_NONE &&
                (fmt == k::IGEMM_W_K16 ? k::ctc_partials_supported((int)K) : k::ctc_partials_supported_x6((int)K))) {
                static const bool on = [] { const char* e = getenv("OAR_CTC_PARTIALS"); return !e || atoi(e) != 0; }();
                if (on) {
                    P.ctc_tiles = k::ctc_tiles((int)Np);
                    part = alloc_arena((size_t)M * P.ctc_tiles * 16, "");   // no root: stays allocated until the end of the plan
                    P.ctc_part = part;
                }
            }
            const bool has_part = part.kind != Loc::NONE;
            p.ctc_valid = P.logits_valid;
            p.N = 1; p.H = 1; p.W = (int)M; p.Cin = (int)K; p.Ho = 1; p.Wo = (int)M; p.Cout = (int)N;
            p.kh = p.kw = p.sh = p.sw = p.dh = p.dw = 1; p.groups = 1; p.act = act; p.w = w; p.bias = bias; p.y_ld = (int)N;
            auto run = [=](const RunCtx& c) { k::ConvP q = p; q.x = c.at(ain); q.y = c.mut(yl); q.residual = has_res ? c.at(res) : nullptr; q.ctc_part = has_part ? c.mut(part) : nullptr; k::conv_igemm(c.s, q); };
            if (!has_part && P.logits_valid == 0 && K % 16 == 0 && N % 16 == 0 && k::chain_act_ok(act.kind) && chain_loc_ok(ain) && (!has_res || chain_loc_ok(res))) {
                ChainRec r;   // rows x K times K x N: any partition of the rows into samples will do
                r.d.type = k::CH_GEMM; r.d.K = (int)K; r.d.N = (int)N; r.d.cin = (int)K; r.d.pad = 0;
                r.d.in_ld = (int)K; r.d.out_ld = (int)N; r.d.res_ld = (int)N;
                r.d.act = act.kind; r.d.alpha = act.alpha; r.d.beta = act.beta; r.dev_bias = bias;
                r.in = ain; r.out = yl; if (has_res) r.res = res;
                r.rows = M; r.out_root = n.out[0];
                const HostTensor* Bp = bt.ht;
                r.make_w = [Bp, transB]() { return chain_weight_linear(*Bp, transB); };
                step_chainable(run, std::move(r), flops, bytes);
            } else {
                step(run, flops, bytes);
            }
        } else {
            const float* w = bt.loc.cptr;
            k::GemmP g{};
            g.batch = 1; g.M = (int)M; g.N = (int)N; g.K = (int)K; g.transB = transB; g.alpha = alpha; g.B = w; g.bias = bias; g.act = act;
            step([=](const RunCtx& c) { k::GemmP q = g; q.A = c.at(ain); q.C = c.mut(yl); q.residual = has_res ? c.at(res) : nullptr; k::gemm_batched(c.s, q); }, flops, bytes);
        }
    }

    void op_matmul(const GNode& n) {
        TInfo a = get(n.in[0]), b = get(n.in[1]);
        OAR_CHECK(a.dims.size() >= 2 && b.dims.size() >= 2, OAR_UNSUPPORTED_OP, "MatMul: rank < 2");
        Loc al = to_native_loc(a), bl = to_native_loc(b);
        int64_t M = a.dims[a.dims.size() - 2], K = a.dims.back(), N = b.dims.back();
        OAR_CHECK(b.dims[b.dims.size() - 2] == K, OAR_SHAPE_MISMATCH, "MatMul: inner dimension mismatch at " + n.out[0]);
        std::vector<int64_t> ba(a.dims.begin(), a.dims.end() - 2), bb(b.dims.begin(), b.dims.end() - 2);
        int64_t na = numel(ba), nb = numel(bb);
        OAR_CHECK(na == nb || nb == 1 || na == 1, OAR_UNSUPPORTED_OP, "MatMul: general batch broadcasting is not supported");
        std::vector<int64_t> od = na >= nb ? ba : bb;
        od.push_back(M); od.push_back(N);
        int64_t batch = std::max(na, nb);
        TInfo& y = new_out(n.out[0], od, Layout::NATIVE);
        Loc yl = y.loc;
        k::GemmP g{};
        g.batch = (int)batch; g.M = (int)M; g.N = (int)N; g.K = (int)K; g.transB = 0; g.alpha = 1.0f;
        g.sA = na == 1 ? 0 : M * K; g.sB = nb == 1 ? 0 : K * N; g.sC = M * N; g.act = n.act;
        step([=](const RunCtx& c) { k::GemmP q = g; q.A = c.at(al); q.B = c.at(bl); q.C = c.mut(yl); k::gemm_batched(c.s, q); },
             2.0 * batch * M * N * K, 4.0 * batch * (M * K + K * N + M * N));
    }

    void op_softmax(const GNode& n) {
        TInfo x = get(n.in[0]);
        int r = (int)x.dims.size();
        int64_t axis = n.ai("axis", E_opset13() ? -1 : 1);
        if (axis < 0) axis += r;
        Loc xin = to_native_loc(x);
        int64_t C, rows;
        if (E_opset13() && axis != r - 1 && !x.host_int) {
            // an inner axis (round 6): Transpose it to the end, soft-max there, Transpose back
            std::vector<int64_t> perm;
            for (int i = 0; i < r; ++i) if (i != axis) perm.push_back(i);
            perm.push_back(axis);
            std::vector<int64_t> inv(r);
            for (int i = 0; i < r; ++i) inv[perm[i]] = i;
            GNode t1; t1.op = "Transpose"; t1.in = {n.in[0]}; t1.out = {n.out[0] + "::moved"};
            Attr p1; p1.kind = Attr::IS; p1.is = perm; t1.attrs["perm"] = p1;
            op_transpose(t1);
            GNode sm = n; sm.in = {t1.out[0]}; sm.out = {n.out[0] + "::sm"};
            Attr la; la.kind = Attr::I; la.i = -1; sm.attrs["axis"] = la;
            op_softmax(sm);
            GNode t2; t2.op = "Transpose"; t2.in = {sm.out[0]}; t2.out = {n.out[0]};
            Attr p2; p2.kind = Attr::IS; p2.is = inv; t2.attrs["perm"] = p2;
            op_transpose(t2);
            return;
        }
        if (E_opset13()) {
            OAR_CHECK(axis == r - 1, OAR_UNSUPPORTED_OP, "Softmax: only the last axis is supported");
            C = x.dims.back(); rows = numel(x.dims) / std::max<int64_t>(C, 1);
        } else {  // opset < 13: flatten to 2-D at `axis`
            rows = 1; C = 1;
            for (int i = 0; i < r; ++i) (i < axis ? rows : C) *= x.dims[i];
        }
        TInfo& y = new_out(n.out[0], x.dims, Layout::NATIVE);
        Loc yl = y.loc;
        step([=](const RunCtx& c) { k::softmax_lastdim(c.s, c.at(xin), c.mut(yl), rows, (int)C); }, 4.0 * rows * C, 8.0 * rows * C);
    }
    // fused SVTR attention (rewrite pass 5): x [n, T, 3*h*d] (q | k | v, each [h][d]) -> [n, T, h*d]
    void op_attention(const GNode& n) {
        TInfo x = get(n.in[0]);
        const int64_t h = n.ai("heads", 1), d = n.ai("head_dim", 1);
        OAR_CHECK(x.dims.size() == 3 && x.dims[2] == 3 * h * d, OAR_SHAPE_MISMATCH, "Attention: input must be [n, T, 3*heads*head_dim]");
        const int64_t N = x.dims[0], T = x.dims[1];
        OAR_CHECK(k::attention_fits((int)T, (int)h, (int)d), OAR_UNSUPPORTED_OP, "Attention: head_dim must be in 1..=64");
        Loc xin = to_native_loc(x);
        TInfo& y = new_out(n.out[0], {N, T, h * d}, Layout::NATIVE);
        Loc yl = y.loc;
        const float scale = n.af("scale", 1.0f);
        auto run = [=](const RunCtx& c) { k::attention(c.s, c.at(xin), c.mut(yl), (int)N, (int)T, (int)h, (int)d, scale); };
        const double flops = 4.0 * N * h * T * T * d, bytes = 4.0 * N * T * 4 * h * d;
        if (d <= k::kChainMaxHd && d % 4 == 0 && chain_loc_ok(xin)) {
            ChainRec r;
            r.d.type = k::CH_ATTN; r.d.heads = (int)h; r.d.hd = (int)d; r.d.scale = scale; r.d.N = (int)(h * d);
            r.d.in_ld = (int)(3 * h * d); r.d.out_ld = (int)(h * d);
            r.in = xin; r.out = yl; r.rows = N * T; r.fix_n = N; r.fix_T = T; r.out_root = n.out[0];
            step_chainable(run, std::move(r), flops, bytes);
        } else {
            step(run, flops, bytes);
        }
    }
    bool E_opset13() const { return opset >= 13 || opset == 0; }
    int64_t opset = 17;

    void op_layernorm(const GNode& n) {
        TInfo x = get(n.in[0]);
        int r = (int)x.dims.size();
        int64_t axis = n.ai("axis", -1);
        if (axis < 0) axis += r;
        OAR_CHECK(axis == r - 1, OAR_UNSUPPORTED_OP, "LayerNormalization: only the last axis is supported");
        Loc xin = to_native_loc(x);
        const float* g = has_input(n, 1) ? get(n.in[1]).loc.cptr : nullptr;
        const float* b = has_input(n, 2) ? get(n.in[2]).loc.cptr : nullptr;
        int64_t C = x.dims.back(), rows = numel(x.dims) / std::max<int64_t>(C, 1);
        float eps = n.af("epsilon", 1e-5f);
        TInfo& y = new_out(n.out[0], x.dims, Layout::NATIVE);
        Loc yl = y.loc;
        auto run = [=](const RunCtx& c) { k::layernorm(c.s, c.at(xin), g, b, c.mut(yl), rows, (int)C, eps); };
        if (chain_loc_ok(xin) && C > 0) {
            ChainRec r;
            r.d.type = k::CH_LN; r.d.N = (int)C; r.d.in_ld = (int)C; r.d.out_ld = (int)C; r.d.eps = eps; r.dev_gamma = g; r.dev_bias = b;
            r.in = xin; r.out = yl; r.rows = rows; r.out_root = n.out[0];
            step_chainable(run, std::move(r), 8.0 * rows * C, 8.0 * rows * C);
        } else {
            step(run, 8.0 * rows * C, 8.0 * rows * C);
        }
    }

    void op_host_arith(const GNode& n, int op) {
        const TInfo &a = get(n.in[0]), &b = get(n.in[1]);
        TInfo o; o.host_int = true;
        size_t cnt = std::max(a.hv.size(), b.hv.size());
        for (size_t i = 0; i < cnt; ++i) {
            int64_t x = a.hv[a.hv.size() == 1 ? 0 : i], y = b.hv[b.hv.size() == 1 ? 0 : i], v = 0;
            switch (op) { case 0: v = x + y; break; case 1: v = x - y; break; case 2: v = x * y; break; default: v = y ? x / y : 0; }
            o.hv.push_back(v);
        }
        o.dims = a.hv.size() >= b.hv.size() ? a.dims : b.dims;
        vals[n.out[0]] = o;
    }

    // ------------------------------------------------------------------ driver
    void compute_last_use() {
        static const std::set<std::string> alias_ops = {"Reshape", "Flatten", "Squeeze", "Unsqueeze", "Transpose", "Identity", "Split", "Slice", "Gather", "Cast"};
        std::map<std::string, int> lu;
        for (int i = 0; i < (int)E.nodes_.size(); ++i) {
            const GNode& n = E.nodes_[i];
            for (auto& s : n.in) if (!s.empty()) lu[s] = i;
            if (!n.residual.empty()) lu[n.residual] = i;
        }
        for (auto& o : E.output_names_) lu[o] = 1 << 30;
        // a Resize with a single consumer may run at that consumer (PendingResize): its input has to live until then
        sole_consumer.clear();
        {
            std::map<std::string, std::pair<int, int>> uses;   // value -> (count, last consumer)
            for (int i = 0; i < (int)E.nodes_.size(); ++i) {
                const GNode& n = E.nodes_[i];
                for (auto& s : n.in) if (!s.empty()) { auto& u = uses[s]; ++u.first; u.second = i; }
                if (!n.residual.empty()) { auto& u = uses[n.residual]; ++u.first; u.second = i; }
            }
            for (int i = 0; i < (int)E.nodes_.size(); ++i) {
                const GNode& n = E.nodes_[i];
                if ((n.op != "Resize" && n.op != "ConvTranspose") || n.out.empty() || n.in.empty()) continue;
                auto u = uses.find(n.out[0]);
                if (u == uses.end() || u->second.first != 1) continue;
                if (std::find(E.output_names_.begin(), E.output_names_.end(), n.out[0]) != E.output_names_.end()) continue;
                sole_consumer[n.out[0]] = u->second.second;
                int& l = lu[n.in[0]];
                l = std::max(l, u->second.second);
            }
        }
        concat_consumer.clear();
        {
            std::map<std::string, std::pair<int, int>> uses;
            for (int i = 0; i < (int)E.nodes_.size(); ++i) {
                const GNode& n = E.nodes_[i];
                for (auto& s : n.in) if (!s.empty()) { auto& u = uses[s]; ++u.first; u.second = i; }
                if (!n.residual.empty()) { auto& u = uses[n.residual]; ++u.first; u.second = i; }
            }
            for (int i = 0; i < (int)E.nodes_.size(); ++i) {
                const GNode& n = E.nodes_[i];
                if (n.op != "Concat" || n.out.empty() || n.in.size() < 3) continue;
                auto u = uses.find(n.out[0]);
                if (u == uses.end() || u->second.first != 1) continue;
                if (std::find(E.output_names_.begin(), E.output_names_.end(), n.out[0]) != E.output_names_.end()) continue;
                const GNode& cn = E.nodes_[u->second.second];
                if (cn.op != "Conv" || cn.in.empty() || cn.in[0] != n.out[0]) continue;
                concat_consumer[n.out[0]] = u->second.second;
                for (auto& sname : n.in) { int& l = lu[sname]; l = std::max(l, u->second.second); }   // the sources live until the convolution that reads them
            }
        }
        for (int i = (int)E.nodes_.size() - 1; i >= 0; --i) {
            const GNode& n = E.nodes_[i];
            if (!alias_ops.count(n.op) || n.in.empty()) continue;
            int m = lu.count(n.in[0]) ? lu[n.in[0]] : i;
            for (auto& o : n.out) if (lu.count(o)) m = std::max(m, lu[o]);
            lu[n.in[0]] = m;
        }
        last_use = lu;
    }

    void release_dead(int i) {
        for (auto& t : temps) arena.release(t.first, t.second);
        temps.clear();
        std::vector<std::string> dead;
        for (auto& rb : root_bytes) {
            auto it = last_use.find(rb.first);
            int lu = it == last_use.end() ? i : it->second;
            if (lu <= i) dead.push_back(rb.first);
        }
        for (auto& d : dead) { arena.release(root_off[d], root_bytes[d]); root_bytes.erase(d); root_off.erase(d); }
    }

    bool skip_final_softmax = false;
    std::set<int> fused_into_prev;   // node indices whose work was planned by their predecessor
    bool stem_u8 = false;
    void build(const std::vector<int64_t>& in_dims, bool in_clast, const std::vector<std::vector<int64_t>>* extra_dims) {
        compute_last_use();
        TInfo in;
        in.u8_stem = stem_u8;
        in.dims = in_dims;
        in.layout = (in_clast && in_dims.size() >= 3) ? Layout::CLAST : Layout::NATIVE;
        in.loc.kind = Loc::INPUT;
        in.root = "";
        vals[E.input_name_] = in;
        for (size_t i = 1; extra_dims && i < extra_dims->size() && i < E.input_infos_.size(); ++i) {
            TInfo ex;
            ex.dims = (*extra_dims)[i];
            ex.layout = Layout::NATIVE;
            ex.loc.kind = Loc::EXTRA;
            ex.loc.idx = (int)i;
            ex.root = "";
            vals[E.input_infos_[i].name] = ex;
        }
        for (int i = 0; i < (int)E.nodes_.size(); ++i) {
            const GNode& n = E.nodes_[i];
            cur = i;
            if (fused_into_prev.count(i)) { release_dead(i); continue; }   // planned together with the node before it (op_dsblock: two blocks, one launch)
            if (skip_final_softmax && n.op == "Softmax" && !n.out.empty() && n.out[0] == E.output_names_[0]) {
                TInfo x = get(n.in[0]);
                int r = (int)x.dims.size();
                int64_t axis = n.ai("axis", -1);
                if (axis < 0) axis += r;
                if (E_opset13() && axis == r - 1 && x.layout == Layout::NATIVE) {
                    alias_out(n.out[0], x, x.dims, Layout::NATIVE);
                    P.skipped_softmax = true;
                    release_dead(i);
                    continue;
                }
            }
            dispatch(n);
            if (!chain_run_open()) release_dead(i);   // an open run of chainable operators keeps its tensors (see ChainRec)
        }
        for (size_t oi = 0; oi < E.output_names_.size(); ++oi) {
            const std::string& on = E.output_names_[oi];
            auto it = vals.find(on);
            OAR_CHECK(it != vals.end(), OAR_MODEL_LOAD, "graph output '" + on + "' was never produced");
            TInfo t = it->second;
            PlanOutput po;
            po.name = on; po.dims = t.dims;
            const int declared = oi < E.output_infos_.size() ? E.output_infos_[oi].elem_type : 0;
            if (t.host_int) {   // known at plan time (Shape and friends): TensorOutput::I64 unless the value is a float
                po.on_host = true;
                po.dtype = t.host_f ? 1 : 7;
                if (t.host_f) po.host_vals = t.hd;
                else po.host_vals.assign(t.hv.begin(), t.hv.end());
                OAR_CHECK((int64_t)po.host_vals.size() == numel(t.dims), OAR_INTERNAL, "host output '" + on + "': value / shape mismatch");
                P.outputs.push_back(po);
                continue;
            }
            po.dtype = (t.is_int || declared == 7 || declared == 6) ? 7 : 1;
            if (t.layout == Layout::CLAST) { po.has_clast = true; po.loc_clast = t.loc; }
            Loc nat = to_native_loc(t);
            if (nat.kind == Loc::ARENA && nat.off != t.loc.off) {
                // keep the converted copy alive: move it out of the temp list
                for (auto it2 = temps.begin(); it2 != temps.end(); ++it2)
                    if ((int64_t)it2->first == nat.off) { temps.erase(it2); break; }
            }
            po.loc = nat;
            P.outputs.push_back(po);
        }
        P.arena_bytes = std::max<size_t>(P.arena_bytes, 256);
        fuse_chains();
    }

    void dispatch(const GNode& n) {
        const std::string& op = n.op;
        // shape / kind of an input without forcing a deferred Resize to run (its consumer decides that)
        auto info = [&](const std::string& s) -> const TInfo& { return (peek_pending(s) || pending_convt.count(s) || pending_concat.count(s)) ? vals.find(s)->second : get(s); };
        bool host_inputs = !n.in.empty();
        for (auto& s : n.in) if (!s.empty()) host_inputs = host_inputs && info(s).host_int;
        if (op == "Shape") {
            const TInfo& x = get(n.in[0]);
            TInfo o; o.host_int = true; o.hv = x.dims; o.dims = {(int64_t)x.dims.size()};
            vals[n.out[0]] = o; return;
        }
        // shape arithmetic: every input is a host value or a small constant, and at least one really is a host value
        bool evaluable = !n.in.empty(), any_host = false;
        for (auto& s : n.in) if (!s.empty()) { const TInfo& t = info(s); evaluable = evaluable && host_evaluable(t); any_host = any_host || t.host_int; }
        if (evaluable && (any_host || op == "Range") && op_host(n)) return;
        (void)host_inputs;
        if (op == "ConstantOfShape") return op_constant_of_shape(n);
        if (op == "Conv") return op_conv(n);
        if (op == "ConvTranspose") return op_convt(n);
        if (op == "BatchNormalization") return op_bn(n);
        if (is_unary_act(op)) return op_unary(n, act_of(n));
        if (op == "Clip") {
            Act a; a.kind = k::ACT_CLIP; a.alpha = n.af("min", -3.402823466e38f); a.beta = n.af("max", 3.402823466e38f);
            if (has_input(n, 1)) { const TInfo& t = get(n.in[1]); OAR_CHECK(t.ht, OAR_UNSUPPORTED_OP, "Clip: dynamic min"); a.alpha = t.ht->f[0]; }
            if (has_input(n, 2)) { const TInfo& t = get(n.in[2]); OAR_CHECK(t.ht, OAR_UNSUPPORTED_OP, "Clip: dynamic max"); a.beta = t.ht->f[0]; }
            return op_unary(n, a);
        }
        if (op == "Add") return op_binary(n, 0);
        if (op == "Sub") return op_binary(n, 1);
        if (op == "Mul") return op_binary(n, 2);
        if (op == "Div") return op_binary(n, 3);
        if (op == "Pow") return op_binary(n, 4);
        if (op == "PRelu") return op_binary(n, 5);
        if (op == "Max") return op_binary(n, 6);
        if (op == "Min") return op_binary(n, 7);
        if (op == "Equal") return op_binary(n, 8);
        if (op == "Less") return op_binary(n, 9);
        if (op == "Greater") return op_binary(n, 10);
        if (op == "And") return op_binary(n, 11);
        if (op == "Or") return op_binary(n, 12);
        if (op == "ReduceMean") return op_reduce(n, 0);
        if (op == "ReduceSum") return op_reduce(n, 1);
        if (op == "ReduceMax") return op_reduce(n, 2);
        if (op == "ArgMax") return op_argreduce(n, false);
        if (op == "ArgMin") return op_argreduce(n, true);
        if (op == "ReduceMin") return op_reduce(n, 3);
        if (op == "ReduceProd") return op_reduce(n, 4);
        if (op == "Expand") return op_expand(n);
        if (op == "Tile") return op_tile(n);
        if (op == "Where") return op_where(n);
        if (op == "GridSample") return op_grid_sample(n);
        if (op == "Pad") return op_pad(n);
        if (op == "GlobalAveragePool") return op_gap(n);
        if (op == "AveragePool") return op_pool(n, false);
        if (op == "MaxPool") return op_pool(n, true);
        if (op == "Resize") return op_resize(n);
        if (op == "Concat") return op_concat(n);
        if (op == "Reshape") return op_reshape(n);
        if (op == "Flatten") return op_flatten(n);
        if (op == "Squeeze") return op_squeeze(n);
        if (op == "Unsqueeze") return op_unsqueeze(n);
        if (op == "Transpose") return op_transpose(n);
        if (op == "Split") return op_split(n);
        if (op == "Slice") return op_slice(n);
        if (op == "Gather") return op_gather(n);
        if (op == "Linear") return op_linear(n, false);
        if (op == "Gemm") return op_linear(n, true);
        if (op == "MatMul") return op_matmul(n);
        if (op == "Softmax") return op_softmax(n);
        if (op == "Attention") return op_attention(n);
        if (op == "SEGate") return op_se_gate(n);
        if (op == "DSBlock") return op_dsblock(n);
        if (op == "LayerNormalization") return op_layernorm(n);
        if (op == "Identity") { const TInfo& x = get(n.in[0]); if (x.host_int) { vals[n.out[0]] = x; } else { TInfo xx = x; alias_out(n.out[0], xx, xx.dims, xx.layout); } return; }
        if (op == "Cast") {   // device tensors are f32 whatever they are called: float <-> bool casts of 0/1 masks are aliases
            const TInfo& x = get(n.in[0]);
            const int64_t to = n.ai("to", 1);
            OAR_CHECK(to == 1 || to == 9 || to == 10 || to == 11 || x.is_int, OAR_UNSUPPORTED_OP, "Cast of a device tensor to an integer type");
            TInfo xx = x; alias_out(n.out[0], xx, xx.dims, xx.layout).is_int = x.is_int; return;
        }
        fail(OAR_UNSUPPORTED_OP, "operator '" + op + "' is not implemented (node output " + (n.out.empty() ? "?" : n.out[0]) + ")");
    }
};

// Host-only structural checks of the initializers the load-time rewrites and the planner index by their DECLARED dims
// (a malformed file must end in OAR_MODEL_LOAD, not in an out-of-bounds read).
void Engine::validate_model(const OnnxModel& m) {
    auto init = [&](const std::string& name) -> const HostTensor* {
        auto it = m.initializers.find(name);
        return it == m.initializers.end() ? nullptr : &it->second;
    };
    auto in = [&](const OnnxNode& n, size_t i) -> const HostTensor* { return i < n.inputs.size() && !n.inputs[i].empty() ? init(n.inputs[i]) : nullptr; };
    for (const OnnxNode& n : m.nodes) {
        const std::string at = " (node output " + (n.outputs.empty() ? std::string("?") : n.outputs[0]) + ")";
        OAR_CHECK(!n.outputs.empty(), OAR_MODEL_LOAD, "onnx: node '" + n.op + "' without outputs");
        if (n.op == "Conv" || n.op == "ConvTranspose") {
            OAR_CHECK(n.inputs.size() >= 2, OAR_MODEL_LOAD, n.op + ": needs at least 2 inputs" + at);
            if (const HostTensor* w = in(n, 1)) {
                OAR_CHECK(w->dtype == DType::F32 && w->dims.size() == 4, OAR_MODEL_LOAD, n.op + ": weight must be a rank-4 f32 tensor" + at);
                for (int64_t d : w->dims) OAR_CHECK(d > 0, OAR_MODEL_LOAD, n.op + ": weight has an empty dimension" + at);
                if (const HostTensor* b = in(n, 2)) {
                    const int64_t cout = n.op == "Conv" ? w->dims[0] : w->dims[1] * n.ai("group", 1);
                    OAR_CHECK(b->dtype == DType::F32 && (int64_t)b->f.size() == cout, OAR_MODEL_LOAD, n.op + ": bias must hold Cout f32 values" + at);
                }
            }
        } else if (n.op == "BatchNormalization") {
            OAR_CHECK(n.inputs.size() >= 5, OAR_MODEL_LOAD, "BatchNormalization: needs 5 inputs" + at);
            int64_t c = -1;
            for (size_t i = 1; i < 5; ++i)
                if (const HostTensor* t = in(n, i)) {
                    OAR_CHECK(t->dtype == DType::F32 && !t->f.empty(), OAR_MODEL_LOAD, "BatchNormalization: parameters must be non-empty f32 tensors" + at);
                    if (c < 0) c = (int64_t)t->f.size();
                    OAR_CHECK((int64_t)t->f.size() == c, OAR_MODEL_LOAD, "BatchNormalization: scale / bias / mean / var differ in length" + at);
                }
        } else if (n.op == "Gemm" || n.op == "MatMul") {
            OAR_CHECK(n.inputs.size() >= 2, OAR_MODEL_LOAD, n.op + ": needs 2 inputs" + at);
            if (const HostTensor* b = in(n, 1)) {
                OAR_CHECK(b->dtype == DType::F32 && b->dims.size() >= 1, OAR_MODEL_LOAD, n.op + ": constant B must be an f32 tensor" + at);
                if (n.op == "Gemm") OAR_CHECK(b->dims.size() == 2, OAR_MODEL_LOAD, "Gemm: B must be rank 2" + at);
            }
        } else if (n.op == "PRelu" || n.op == "LayerNormalization") {
            OAR_CHECK(n.inputs.size() >= 2, OAR_MODEL_LOAD, n.op + ": needs 2 inputs" + at);
        }
    }
}

// every operator name dispatch() (or a load-time rewrite) accepts -- oar_onnx_inspect reports the rest
const std::set<std::string>& Engine::supported_ops() {
    static const std::set<std::string> ops = {
        "Constant", "Dropout", "Identity", "Cast", "Shape", "Conv", "ConvTranspose", "BatchNormalization", "Relu", "HardSwish", "HardSigmoid", "Sigmoid",
        "LeakyRelu", "Tanh", "Erf", "Sqrt", "Exp", "Abs", "Neg", "Reciprocal", "Log", "Gelu", "Softplus", "Clip", "Add", "Sub", "Mul", "Div", "Pow", "PRelu",
        "ReduceMean", "GridSample", "Pad", "GlobalAveragePool", "AveragePool", "MaxPool", "Resize", "Concat", "Reshape", "Flatten", "Squeeze", "Unsqueeze",
        "Transpose", "Split", "Slice", "Gather", "Gemm", "MatMul", "Softmax", "LayerNormalization", "Max", "Min", "Equal", "Less", "Greater", "And", "Or", "Not",
        "Floor", "Ceil", "Round", "ReduceSum", "ReduceMax", "ReduceMin", "ReduceProd", "Expand", "Tile", "Where", "ConstantOfShape", "Range", "ArgMax", "ArgMin"};
    return ops;
}

const Plan& Engine::plan_for(const std::vector<int64_t>& dims, bool in_clast, bool skip_final_softmax,
                             const std::vector<std::vector<int64_t>>* extra_dims, bool stem_u8) {
    std::ostringstream key;
    key << (in_clast ? "L" : "N") << (skip_final_softmax ? "S" : "") << (stem_u8 ? "U" : "");
    for (auto d : dims) key << "x" << d;
    for (size_t i = 1; extra_dims && i < extra_dims->size(); ++i) {
        key << "|";
        for (auto d : (*extra_dims)[i]) key << "x" << d;
    }
    auto it = plans_.find(key.str());
    if (it != plans_.end()) { it->second->last_used = ++tick_; last_returned_ = it->second.get(); return *it->second; }
    OAR_HIP(hipSetDevice(device_));
    std::unique_ptr<Plan> p(new Plan());
    Planner pl(*this, *p);
    pl.skip_final_softmax = skip_final_softmax;
    pl.stem_u8 = stem_u8;
    pl.build(dims, in_clast, extra_dims);
    evict_plans();
    p->last_used = ++tick_;
    const Plan& ref = *p;
    plans_[key.str()] = std::move(p);
    last_returned_ = &ref;
    return ref;
}

// The plan cache is keyed on exact input dims (every (batch, Wt) pair of the recognizer, every (sub-batch, rh, rw) of the
// detector), so a long-running server on heterogeneous pages would grow it without bound: keep at most plan_cap_ plans,
// least-recently-used out first.  The plan returned by the PREVIOUS plan_for / run call is never the victim, so a `const
// Plan&` stays valid across one further call on the same engine (callers copy what they need before that).
void Engine::evict_plans() {
    while (plans_.size() >= plan_cap_) {
        auto victim = plans_.end();
        for (auto it = plans_.begin(); it != plans_.end(); ++it) {
            if (it->second.get() == last_returned_) continue;
            if (victim == plans_.end() || it->second->last_used < victim->second->last_used) victim = it;
        }
        if (victim == plans_.end()) return;
        const Plan* dead = victim->second.get();
        bool synced = false;
        for (auto g = graphs_.begin(); g != graphs_.end();) {   // captured graphs of the evicted plan go with it
            if (std::get<0>(g->first) != dead) { ++g; continue; }
            if (!synced) { OAR_HIP(hipStreamSynchronize(stream_)); synced = true; }
            if (g->second.events) Profiler::get().release(*g->second.events);
            if (g->second.exec) (void)hipGraphExecDestroy(g->second.exec);
            g = graphs_.erase(g);
        }
        plans_.erase(victim);
        ++plans_evicted_;
    }
}

void Engine::clear_graphs() {
    for (auto& kv : graphs_) {
        if (kv.second.events) Profiler::get().release(*kv.second.events);
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    }
    graphs_.clear();
}

// Replays (capturing first if needed) the plan as a hipGraph.  Returns false when the plan has to be enqueued kernel by
// kernel (first run, graphs disabled, capture failed).
bool Engine::replay(const Plan& p, const RunCtx& c) {
    // Opt-in (OAR_HIP_GRAPH=1).  Measured on ROCm 7.2 / MI355X with bench.py: 22.8-22.9 ms per step with replay vs
    // 21.9-23.0 ms without.  The rocprofv3 kernel trace explains it: dependent kernels enqueued one by one already start
    // back to back (median gap 0.00 us), so a graph of kernel nodes has no dispatch gap left to remove; with external
    // event-record nodes (profiler on) replay is 6 % slower.
    if (!graphs_requested() || !graphs_ok_ || p.runs == 0) return false;
    Profiler& prof = Profiler::get();
    const auto key = std::make_tuple(&p, (const void*)c.input, (const void*)c.arena);
    auto it = graphs_.find(key);
    if (it != graphs_.end() && it->second.epoch != prof.epoch) {
        if (it->second.events) prof.release(*it->second.events);
        (void)hipGraphExecDestroy(it->second.exec);
        graphs_.erase(it);
        it = graphs_.end();
    }
    if (it == graphs_.end()) {
        if (graphs_.size() >= 64) { OAR_HIP(hipStreamSynchronize(stream_)); clear_graphs(); }
        GraphEntry ge;
        ge.events = std::make_shared<Profiler::GraphEvents>();
        ge.epoch = prof.epoch;
        if (hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); graphs_ok_ = false; return false; }
        Profiler::capturing = ge.events.get();
        hipGraph_t g = nullptr;
        bool ok = true;
        try {
            for (auto& st : p.steps) st(c);
        } catch (...) {
            ok = false;
        }
        Profiler::capturing = nullptr;
        if (hipStreamEndCapture(stream_, &g) != hipSuccess || !g) ok = false;
        if (ok && hipGraphInstantiate(&ge.exec, g, nullptr, nullptr, 0) != hipSuccess) ok = false;
        if (g) (void)hipGraphDestroy(g);
        if (!ok) {   // fall back to plain launches for the lifetime of this engine
            (void)hipGetLastError();
            prof.release(*ge.events);
            graphs_ok_ = false;
            return false;
        }
        {
            std::lock_guard<std::mutex> lk(prof.mu);
            prof.graphs.push_back(ge.events);
        }
        it = graphs_.emplace(key, ge).first;
    }
    GraphEntry& ge = it->second;
    if (ge.events->launched) prof.harvest(*ge.events);   // the replay below re-records the same events
    OAR_HIP(hipGraphLaunch(ge.exec, stream_));
    if (!ge.events->ev.empty()) ge.events->launched = true;
    return true;
}

const Plan& Engine::run(const float* d_in, const std::vector<int64_t>& dims, bool in_clast, bool skip_final_softmax) {
    const Plan& p = plan_for(dims, in_clast, skip_final_softmax);
    OAR_HIP(hipSetDevice(device_));
    if (arena_.cap < p.arena_bytes) {
        OAR_HIP(hipStreamSynchronize(stream_));
        clear_graphs();   // they bake the old arena base into their kernel arguments
        arena_.reserve(p.arena_bytes);
    }
    RunCtx c{stream_, d_in, arena_.as<char>()};
    last_input_ = d_in;
    if (!replay(p, c)) {
        if (!p.step_ops.empty() && p.step_ops.size() == p.steps.size()) {   // OAR_DEBUG_STEPS=1 (and no chain fusion re-indexing): which step leaves a HIP error behind?
            for (size_t i = 0; i < p.steps.size(); ++i) {
                p.steps[i](c);
                const hipError_t e = hipGetLastError();
                if (e != hipSuccess) fail(OAR_DEVICE, "HIP error after step " + std::to_string(i) + " (" + p.step_ops[i] + "): " + hipGetErrorString(e));
            }
        } else {
            for (auto& st : p.steps) st(c);
        }
        ++p.runs;
    }
    OAR_HIP(hipGetLastError());
    return p;
}

const Plan& Engine::run_multi(const std::vector<const float*>& d_ins, const std::vector<std::vector<int64_t>>& dims) {
    OAR_CHECK(!d_ins.empty() && d_ins.size() == dims.size() && d_ins.size() == input_infos_.size(), OAR_INVALID_INPUT,
              "run_multi: one tensor per declared graph input is required");
    if (d_ins.size() == 1) return run(d_ins[0], dims[0], false);
    const Plan& p = plan_for(dims[0], false, false, &dims);
    OAR_HIP(hipSetDevice(device_));
    if (arena_.cap < p.arena_bytes) {
        OAR_HIP(hipStreamSynchronize(stream_));
        clear_graphs();
        arena_.reserve(p.arena_bytes);
    }
    last_extra_ = d_ins;
    RunCtx c{stream_, d_ins[0], arena_.as<char>(), last_extra_.data()};
    last_input_ = d_ins[0];
    for (auto& st : p.steps) st(c);   // no hipGraph replay here: a captured graph bakes in one input pointer only
    ++p.runs;
    OAR_HIP(hipGetLastError());
    return p;
}

const Plan& Engine::run_stem(const k::StemU8& st, const std::vector<int64_t>& dims, bool skip_final_softmax) {
    OAR_CHECK(stem_fusable_, OAR_INTERNAL, "run_stem on a graph without a fusable RGB stem");
    OAR_CHECK(dims.size() == 4 && dims[1] == 3 && dims[0] >= 1 && (st.dev || dims[0] <= 32), OAR_INVALID_INPUT, "run_stem: dims must be {n, 3, H, W} (n <= 32 pages by value)");
    const Plan& p = plan_for(dims, true, skip_final_softmax, nullptr, true);
    OAR_HIP(hipSetDevice(device_));
    if (arena_.cap < p.arena_bytes) {
        OAR_HIP(hipStreamSynchronize(stream_));
        clear_graphs();
        arena_.reserve(p.arena_bytes);
    }
    RunCtx c{stream_, nullptr, arena_.as<char>(), nullptr, &st};
    last_input_ = nullptr;
    for (auto& stp : p.steps) stp(c);   // no hipGraph replay: the page pointers change from call to call
    ++p.runs;
    OAR_HIP(hipGetLastError());
    return p;
}

const float* Engine::out_ptr(const Loc& l) const {
    RunCtx c{stream_, last_input_, arena_.as<char>(), last_extra_.empty() ? nullptr : last_extra_.data()};
    return c.at(l);
}

}  // namespace oar
