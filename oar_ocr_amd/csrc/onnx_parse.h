// onnx_parse.h -- hand-rolled protobuf wire reader for the subset of onnx.proto the engine needs.
// Replaces the model-loading half of `OrtInfer::from_config` (core/inference/session.rs:30-44):
// same `.onnx` bytes in (ModelSource::{Path,Memory}, core/inference/model_source.rs:21-28).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace oar {

enum class DType : int { F32 = 1, I32 = 6, I64 = 7, BOOL = 9, F64 = 11 };

struct HostTensor {
    std::string name;
    DType dtype = DType::F32;
    std::vector<int64_t> dims;
    std::vector<float> f;     // F32 payload
    std::vector<int64_t> i;   // integer payload (I32/I64/BOOL widened)
    int64_t numel() const {
        int64_t n = 1;
        for (auto d : dims) n *= d;
        return n;
    }
};

struct Attr {
    enum Kind { NONE, F, I, S, T, FS, IS } kind = NONE;
    float f = 0;
    int64_t i = 0;
    std::string s;
    HostTensor t;
    std::vector<float> fs;
    std::vector<int64_t> is;
};

struct OnnxNode {
    std::string op, name;
    std::vector<std::string> inputs, outputs;
    std::map<std::string, Attr> attrs;
    int64_t ai(const char* k, int64_t dflt) const {
        auto it = attrs.find(k);
        return it == attrs.end() ? dflt : it->second.i;
    }
    float af(const char* k, float dflt) const {
        auto it = attrs.find(k);
        return it == attrs.end() ? dflt : it->second.f;
    }
    std::string as(const char* k, const std::string& dflt) const {
        auto it = attrs.find(k);
        return it == attrs.end() ? dflt : it->second.s;
    }
    std::vector<int64_t> ais(const char* k, std::vector<int64_t> dflt = {}) const {
        auto it = attrs.find(k);
        return it == attrs.end() ? dflt : it->second.is;
    }
    bool has(const char* k) const { return attrs.count(k) != 0; }
};

struct ValueInfo {  // a graph input / output as declared (ValueInfoProto): dynamic dimensions are -1
    std::string name;
    int elem_type = 0;          // onnx TensorProto.DataType (1 = f32, 7 = i64); 0 = not declared
    bool has_shape = false;
    std::vector<int64_t> dims;
};

struct OnnxModel {
    std::vector<OnnxNode> nodes;
    std::map<std::string, HostTensor> initializers;
    std::vector<std::string> inputs;   // non-initializer graph inputs
    std::vector<std::string> outputs;
    std::vector<ValueInfo> input_infos, output_infos;   // same order as inputs / outputs
    int64_t opset = 0;
};

OnnxModel parse_onnx(const uint8_t* data, size_t len);

}  // namespace oar
