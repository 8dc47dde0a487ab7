#include "onnx_parse.h"

#include <cstring>

#include "common.h"

namespace oar {
namespace {

struct Rd {
    const uint8_t* p;
    const uint8_t* e;
    bool done() const { return p >= e; }
    uint64_t varint() {
        uint64_t r = 0;
        int s = 0;
        while (true) {
            if (p >= e) fail(OAR_MODEL_LOAD, "onnx: truncated varint");
            uint8_t c = *p++;
            r |= (uint64_t)(c & 0x7F) << s;
            if (!(c & 0x80)) return r;
            s += 7;
            if (s > 63) fail(OAR_MODEL_LOAD, "onnx: varint overflow");
        }
    }
    // Reads a key; returns field number and wire type.
    void key(uint32_t& field, uint32_t& wt) {
        uint64_t k = varint();
        field = (uint32_t)(k >> 3);
        wt = (uint32_t)(k & 7);
    }
    Rd sub() {
        uint64_t n = varint();
        if ((uint64_t)(e - p) < n) fail(OAR_MODEL_LOAD, "onnx: truncated length-delimited field");
        Rd r{p, p + n};
        p += n;
        return r;
    }
    void skip(uint32_t wt) {
        switch (wt) {
            case 0: varint(); break;
            case 1: need(8); p += 8; break;
            case 2: sub(); break;
            case 5: need(4); p += 4; break;
            default: fail(OAR_MODEL_LOAD, "onnx: unsupported wire type");
        }
    }
    void need(size_t n) {
        if ((size_t)(e - p) < n) fail(OAR_MODEL_LOAD, "onnx: truncated fixed field");
    }
    float f32() {
        need(4);
        float v;
        memcpy(&v, p, 4);
        p += 4;
        return v;
    }
    std::string str() {
        Rd s = sub();
        return std::string((const char*)s.p, (size_t)(s.e - s.p));
    }
};

void read_ints(Rd& r, uint32_t wt, std::vector<int64_t>& out) {
    if (wt == 0) {
        out.push_back((int64_t)r.varint());
    } else if (wt == 2) {
        Rd s = r.sub();
        while (!s.done()) out.push_back((int64_t)s.varint());
    } else {
        fail(OAR_MODEL_LOAD, "onnx: bad int field");
    }
}
void read_floats(Rd& r, uint32_t wt, std::vector<float>& out) {
    if (wt == 5) {
        out.push_back(r.f32());
    } else if (wt == 2) {
        Rd s = r.sub();
        while (!s.done()) out.push_back(s.f32());
    } else {
        fail(OAR_MODEL_LOAD, "onnx: bad float field");
    }
}

HostTensor parse_tensor(Rd r) {
    HostTensor t;
    int dt = 1;
    const uint8_t* raw = nullptr;
    size_t raw_n = 0;
    std::vector<float> fdata;
    std::vector<int64_t> i32data, i64data;
    while (!r.done()) {
        uint32_t f, wt;
        r.key(f, wt);
        switch (f) {
            case 1: read_ints(r, wt, t.dims); break;
            case 2: dt = (int)r.varint(); break;
            case 4: read_floats(r, wt, fdata); break;
            case 5: read_ints(r, wt, i32data); break;
            case 7: read_ints(r, wt, i64data); break;
            case 8: t.name = r.str(); break;
            case 9: {
                Rd s = r.sub();
                raw = s.p;
                raw_n = (size_t)(s.e - s.p);
                break;
            }
            default: r.skip(wt);
        }
    }
    // dims come from the file: every later index computation trusts numel(), so it is validated here, once
    int64_t n = 1;
    OAR_CHECK(t.dims.size() <= 8, OAR_MODEL_LOAD, "onnx: tensor rank > 8 in " + t.name);
    for (int64_t d : t.dims) {
        OAR_CHECK(d >= 0, OAR_MODEL_LOAD, "onnx: negative dimension in initializer " + t.name);
        OAR_CHECK(d == 0 || n <= (int64_t)1 << 33, OAR_MODEL_LOAD, "onnx: initializer element count overflows in " + t.name);
        n *= d;
    }
    OAR_CHECK(n <= (int64_t)1 << 33, OAR_MODEL_LOAD, "onnx: initializer too large: " + t.name);
    auto need_raw = [&](size_t elem) { OAR_CHECK(raw_n == (size_t)n * elem, OAR_MODEL_LOAD, "onnx: raw_data size does not match dims in " + t.name); };
    auto need_len = [&](size_t have) { OAR_CHECK((int64_t)have == n, OAR_MODEL_LOAD, "onnx: typed data length does not match dims in " + t.name); };
    switch (dt) {
        case 1:
            t.dtype = DType::F32;
            if (raw) {
                need_raw(4);
                t.f.resize(n);
                if (n) memcpy(t.f.data(), raw, raw_n);
            } else {
                need_len(fdata.size());
                t.f = fdata;
            }
            break;
        case 7:
            t.dtype = DType::I64;
            if (raw) {
                need_raw(8);
                t.i.resize(n);
                if (n) memcpy(t.i.data(), raw, raw_n);
            } else {
                need_len(i64data.size());
                t.i = i64data;
            }
            break;
        case 6:
            t.dtype = DType::I32;
            if (raw) {
                need_raw(4);
                t.i.resize(n);
                for (int64_t k = 0; k < n; ++k) {
                    int32_t v;
                    memcpy(&v, raw + k * 4, 4);
                    t.i[k] = v;
                }
            } else {
                need_len(i32data.size());
                t.i.resize(i32data.size());
                for (size_t k = 0; k < i32data.size(); ++k) t.i[k] = (int32_t)i32data[k];
            }
            break;
        case 9:
            t.dtype = DType::BOOL;
            if (raw) {
                need_raw(1);
                t.i.resize(n);
                for (int64_t k = 0; k < n; ++k) t.i[k] = raw[k];
            } else {
                need_len(i32data.size());
                t.i = i32data;
            }
            break;
        case 11: {  // double -> f32
            t.dtype = DType::F32;
            OAR_CHECK(raw != nullptr || n == 0, OAR_MODEL_LOAD, "onnx: f64 tensor without raw_data in " + t.name);
            if (raw) need_raw(8);
            t.f.resize(n);
            for (int64_t k = 0; k < n; ++k) {
                double v;
                memcpy(&v, raw + k * 8, 8);
                t.f[k] = (float)v;
            }
            break;
        }
        default: fail(OAR_UNSUPPORTED_OP, "onnx: tensor data_type " + std::to_string(dt) + " not supported (" + t.name + ")");
    }
    return t;
}

void parse_attr(Rd r, std::string& name, Attr& a) {
    int type = 0;
    bool has_f = false, has_i = false, has_s = false, has_t = false;
    while (!r.done()) {
        uint32_t f, wt;
        r.key(f, wt);
        switch (f) {
            case 1: name = r.str(); break;
            case 2: a.f = r.f32(); has_f = true; break;
            case 3: a.i = (int64_t)r.varint(); has_i = true; break;
            case 4: a.s = r.str(); has_s = true; break;
            case 5: a.t = parse_tensor(r.sub()); has_t = true; break;
            case 7: read_floats(r, wt, a.fs); break;
            case 8: read_ints(r, wt, a.is); break;
            case 20: type = (int)r.varint(); break;
            default: r.skip(wt);
        }
    }
    switch (type) {
        case 1: a.kind = Attr::F; break;
        case 2: a.kind = Attr::I; break;
        case 3: a.kind = Attr::S; break;
        case 4: a.kind = Attr::T; break;
        case 6: a.kind = Attr::FS; break;
        case 7: a.kind = Attr::IS; break;
        default:
            a.kind = has_t ? Attr::T : has_s ? Attr::S : !a.is.empty() ? Attr::IS : !a.fs.empty() ? Attr::FS : has_i ? Attr::I : has_f ? Attr::F : Attr::NONE;
    }
}

OnnxNode parse_node(Rd r) {
    OnnxNode n;
    while (!r.done()) {
        uint32_t f, wt;
        r.key(f, wt);
        switch (f) {
            case 1: n.inputs.push_back(r.str()); break;
            case 2: n.outputs.push_back(r.str()); break;
            case 3: n.name = r.str(); break;
            case 4: n.op = r.str(); break;
            case 5: {
                std::string k;
                Attr a;
                parse_attr(r.sub(), k, a);
                n.attrs[k] = std::move(a);
                break;
            }
            default: r.skip(wt);
        }
    }
    return n;
}

// ValueInfoProto { name = 1, type = 2: TypeProto { tensor_type = 1: Tensor { elem_type = 1, shape = 2: TensorShapeProto {
// dim = 1: Dimension { dim_value = 1 | dim_param = 2 } } } } }
ValueInfo parse_value_info(Rd r) {
    ValueInfo vi;
    while (!r.done()) {
        uint32_t f, wt;
        r.key(f, wt);
        if (f == 1 && wt == 2) vi.name = r.str();
        else if (f == 2 && wt == 2) {
            Rd ty = r.sub();
            while (!ty.done()) {
                uint32_t f2, wt2;
                ty.key(f2, wt2);
                if (!(f2 == 1 && wt2 == 2)) { ty.skip(wt2); continue; }
                Rd tt = ty.sub();
                while (!tt.done()) {
                    uint32_t f3, wt3;
                    tt.key(f3, wt3);
                    if (f3 == 1 && wt3 == 0) vi.elem_type = (int)tt.varint();
                    else if (f3 == 2 && wt3 == 2) {
                        vi.has_shape = true;
                        Rd sh = tt.sub();
                        while (!sh.done()) {
                            uint32_t f4, wt4;
                            sh.key(f4, wt4);
                            if (!(f4 == 1 && wt4 == 2)) { sh.skip(wt4); continue; }
                            Rd dm = sh.sub();
                            int64_t v = -1;
                            while (!dm.done()) {
                                uint32_t f5, wt5;
                                dm.key(f5, wt5);
                                if (f5 == 1 && wt5 == 0) v = (int64_t)dm.varint();
                                else dm.skip(wt5);
                            }
                            OAR_CHECK(vi.dims.size() < 16, OAR_MODEL_LOAD, "onnx: value_info rank > 16 in " + vi.name);
                            vi.dims.push_back(v);
                        }
                    } else tt.skip(wt3);
                }
            }
        } else r.skip(wt);
    }
    return vi;
}

void parse_graph(Rd r, OnnxModel& m) {
    std::vector<ValueInfo> inputs;
    while (!r.done()) {
        uint32_t f, wt;
        r.key(f, wt);
        switch (f) {
            case 1: m.nodes.push_back(parse_node(r.sub())); break;
            case 5: {
                HostTensor t = parse_tensor(r.sub());
                std::string nm = t.name;
                m.initializers[nm] = std::move(t);
                break;
            }
            case 11: inputs.push_back(parse_value_info(r.sub())); break;
            case 12: m.output_infos.push_back(parse_value_info(r.sub())); m.outputs.push_back(m.output_infos.back().name); break;
            default: r.skip(wt);
        }
    }
    for (auto& i : inputs)
        if (!m.initializers.count(i.name)) { m.inputs.push_back(i.name); m.input_infos.push_back(i); }
}

}  // namespace

OnnxModel parse_onnx(const uint8_t* data, size_t len) {
    OAR_CHECK(data && len > 0, OAR_MODEL_LOAD, "onnx: empty model buffer");
    OnnxModel m;
    Rd r{data, data + len};
    bool have_graph = false;
    while (!r.done()) {
        uint32_t f, wt;
        r.key(f, wt);
        if (f == 7 && wt == 2) {
            parse_graph(r.sub(), m);
            have_graph = true;
        } else if (f == 8 && wt == 2) {
            Rd s = r.sub();
            std::string domain;
            int64_t ver = 0;
            while (!s.done()) {
                uint32_t f2, wt2;
                s.key(f2, wt2);
                if (f2 == 1) domain = s.str();
                else if (f2 == 2) ver = (int64_t)s.varint();
                else s.skip(wt2);
            }
            if (domain.empty() || domain == "ai.onnx") m.opset = ver;
        } else {
            r.skip(wt);
        }
    }
    OAR_CHECK(have_graph, OAR_MODEL_LOAD, "onnx: ModelProto has no graph");
    OAR_CHECK(!m.inputs.empty(), OAR_MODEL_LOAD, "onnx: graph has no runtime input");
    OAR_CHECK(!m.outputs.empty(), OAR_MODEL_LOAD, "onnx: graph has no output");
    return m;
}

}  // namespace oar
