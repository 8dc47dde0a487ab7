"""Oracle for the network forward: a tiny ONNX reader + PyTorch-CPU fp32 interpreter.

TEST INFRASTRUCTURE ONLY (see oracle/oar_oracle.c header).

The reference runs the detector/recognizer graphs through ONNX Runtime (third-party; `ort =2.0.0-rc.13`,
oar-ocr-core/Cargo.toml:51; call sites core/inference/ort_infer_execution.rs:178,281).  Neither ORT nor
the reference's model files exist in this image, and the reference has no test at this boundary, so the
network oracle is "parity unpinned": ONNX operator semantics are restated on torch CPU fp32 kernels,
an implementation that is independent of the HIP engine.
"""
from __future__ import annotations

import struct

import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- protobuf reader
def _read_varint(b, i):
    r = 0
    s = 0
    while True:
        c = b[i]
        i += 1
        r |= (c & 0x7F) << s
        if not c & 0x80:
            return r, i
        s += 7


def _fields(b):
    i = 0
    n = len(b)
    while i < n:
        key, i = _read_varint(b, i)
        f, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _read_varint(b, i)
        elif wt == 1:
            v = b[i:i + 8]
            i += 8
        elif wt == 2:
            ln, i = _read_varint(b, i)
            v = b[i:i + ln]
            i += ln
        elif wt == 5:
            v = b[i:i + 4]
            i += 4
        else:
            raise ValueError(f"wire type {wt}")
        yield f, wt, v


def _sint(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _packed_ints(wt, v):
    if wt == 0:
        return [_sint(v)]
    out = []
    i = 0
    while i < len(v):
        x, i = _read_varint(v, i)
        out.append(_sint(x))
    return out


def _parse_tensor(b):
    dims, dt, name, raw = [], 1, "", None
    f32, i32, i64 = [], [], []
    for f, wt, v in _fields(b):
        if f == 1:
            dims += _packed_ints(wt, v)
        elif f == 2:
            dt = v
        elif f == 8:
            name = bytes(v).decode()
        elif f == 9:
            raw = bytes(v)
        elif f == 4:
            f32 += list(struct.unpack(f"<{len(v) // 4}f", v)) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif f == 5:
            i32 += _packed_ints(wt, v)
        elif f == 7:
            i64 += _packed_ints(wt, v)
    np_dt = {1: np.float32, 6: np.int32, 7: np.int64, 9: np.bool_, 11: np.float64}[dt]
    if raw is not None:
        arr = np.frombuffer(raw, dtype=np_dt).copy()
    elif dt == 1:
        arr = np.array(f32, np.float32)
    elif dt == 7:
        arr = np.array(i64, np.int64)
    else:
        arr = np.array(i32, np_dt)
    return name, arr.reshape(dims)


def _parse_attr(b):
    name, val = "", None
    floats, ints = [], []
    typ = 0
    for f, wt, v in _fields(b):
        if f == 1:
            name = bytes(v).decode()
        elif f == 2:
            val = struct.unpack("<f", v)[0]
        elif f == 3:
            val = _sint(v)
        elif f == 4:
            val = bytes(v).decode()
        elif f == 5:
            val = _parse_tensor(v)[1]
        elif f == 7:
            floats += list(struct.unpack(f"<{len(v) // 4}f", v)) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif f == 8:
            ints += _packed_ints(wt, v)
        elif f == 20:
            typ = v
    if typ == 6 or (val is None and floats):
        val = floats
    elif typ == 7 or (val is None and ints):
        val = ints
    elif val is None and typ == 7:
        val = []
    return name, val


def _parse_node(b):
    n = {"inputs": [], "outputs": [], "op": "", "attrs": {}, "name": ""}
    for f, wt, v in _fields(b):
        if f == 1:
            n["inputs"].append(bytes(v).decode())
        elif f == 2:
            n["outputs"].append(bytes(v).decode())
        elif f == 3:
            n["name"] = bytes(v).decode()
        elif f == 4:
            n["op"] = bytes(v).decode()
        elif f == 5:
            k, val = _parse_attr(v)
            n["attrs"][k] = val
    return n


def _parse_value_info(b):
    for f, wt, v in _fields(b):
        if f == 1:
            return bytes(v).decode()
    return ""


def parse_model(data: bytes):
    graph = None
    for f, wt, v in _fields(memoryview(data)):
        if f == 7:
            graph = v
    assert graph is not None, "no graph"
    nodes, inits, inputs, outputs = [], {}, [], []
    for f, wt, v in _fields(graph):
        if f == 1:
            nodes.append(_parse_node(v))
        elif f == 5:
            k, arr = _parse_tensor(v)
            inits[k] = arr
        elif f == 11:
            inputs.append(_parse_value_info(v))
        elif f == 12:
            outputs.append(_parse_value_info(v))
    inputs = [i for i in inputs if i not in inits]
    return {"nodes": nodes, "inits": inits, "inputs": inputs, "outputs": outputs}


# --------------------------------------------------------------------------- interpreter
def _resize(x, node, env):
    a = node["attrs"]
    mode = a.get("mode", "nearest")
    ins = node["inputs"]
    scales = env.get(ins[2]) if len(ins) > 2 and ins[2] else None
    sizes = env.get(ins[3]) if len(ins) > 3 and ins[3] else None
    n, c, h, w = x.shape
    if sizes is not None and sizes.numel():
        oh, ow = int(sizes[2]), int(sizes[3])
        sh, sw = oh / h, ow / w
    else:
        sh, sw = float(scales[2]), float(scales[3])
        oh, ow = int(np.floor(h * sh)), int(np.floor(w * sw))
    ctm = a.get("coordinate_transformation_mode", "half_pixel")
    if mode == "nearest":
        nm = a.get("nearest_mode", "round_prefer_floor")

        def idx(o, s, n_in):
            # float32 coordinate arithmetic, as ONNX Runtime's (and the engine's): with sizes 21 -> 3 the scale 3 / 21 rounds UP in f32 and 1 / scale lands just below 7 --
            # the floor is 6 there and 7 in float64 (tools/op_fuzz.py, round 6: a knife edge of non-integer nearest reductions; PP-OCR graphs only enlarge by integers)
            o = np.arange(o, dtype=np.float32)
            s = np.float32(s)
            if ctm == "asymmetric":
                x_ = o / s
            elif ctm in ("half_pixel", "pytorch_half_pixel"):
                x_ = (o + np.float32(0.5)) / s - np.float32(0.5)
            else:
                raise NotImplementedError(ctm)
            if nm == "floor":
                r = np.floor(x_)
            elif nm == "ceil":
                r = np.ceil(x_)
            elif nm == "round_prefer_floor":
                r = np.ceil(x_ - 0.5)
            else:
                r = np.floor(x_ + 0.5)
            return torch.from_numpy(np.clip(r, 0, n_in - 1).astype(np.int64))

        return x[:, :, idx(oh, sh, h)][:, :, :, idx(ow, sw, w)]
    if mode == "linear":
        # ONNX Resize-13 "linear" as the specification states it (and ONNX Runtime computes it): the source coordinate comes from the SCALE the node was given
        # (out / in only when `sizes` was given) -- F.interpolate(size=...) derives it from out / in, which differs whenever floor(in * scale) / in != scale
        # (odd lengths halved, x1.5 of an odd length; found by tools/op_fuzz.py in round 6) --, is clamped to [0, in - 1], and the two neighbours are blended.
        def axis(o_len, s, n_in):
            o = np.arange(o_len, dtype=np.float64)
            if ctm == "asymmetric":
                x_ = o / s
            elif ctm == "half_pixel":
                x_ = (o + 0.5) / s - 0.5
            elif ctm == "pytorch_half_pixel":
                x_ = (o + 0.5) / s - 0.5 if o_len > 1 else np.zeros_like(o)
            elif ctm == "align_corners":
                x_ = o * (n_in - 1) / (o_len - 1) if o_len > 1 else np.zeros_like(o)
            else:
                raise NotImplementedError((mode, ctm))
            x_ = np.clip(x_.astype(np.float32), 0.0, float(n_in - 1))
            i0 = np.floor(x_).astype(np.int64)
            i1 = np.minimum(i0 + 1, n_in - 1)
            return torch.from_numpy(i0), torch.from_numpy(i1), torch.from_numpy((x_ - i0).astype(np.float32))
        y0, y1, fy = axis(oh, sh, h)
        x0, x1, fx = axis(ow, sw, w)
        fy = fy.view(1, 1, -1, 1)
        fx = fx.view(1, 1, 1, -1)
        top = x[:, :, y0][:, :, :, x0] * (1 - fx) + x[:, :, y0][:, :, :, x1] * fx
        bot = x[:, :, y1][:, :, :, x0] * (1 - fx) + x[:, :, y1][:, :, :, x1] * fx
        return top * (1 - fy) + bot * fy
    raise NotImplementedError((mode, ctm))


def run(model, feeds: dict, want=None):
    """model: parse_model() dict (or raw bytes).  feeds: name -> np.ndarray.  Returns list of np outputs."""
    if isinstance(model, (bytes, bytearray)):
        model = parse_model(model)
    env = {k: torch.from_numpy(v) for k, v in model["inits"].items()}
    for k, v in feeds.items():
        env[k] = torch.from_numpy(np.ascontiguousarray(v))
    with torch.no_grad():
        for nd in model["nodes"]:
            op, a = nd["op"], nd["attrs"]
            x = [env[i] if i else None for i in nd["inputs"]]
            if op == "Conv":
                pads = a.get("pads", [0, 0, 0, 0])
                inp = x[0]
                if pads[0] != pads[2] or pads[1] != pads[3]:
                    inp = F.pad(inp, (pads[1], pads[3], pads[0], pads[2]))
                    pad = (0, 0)
                else:
                    pad = (pads[0], pads[1])
                y = F.conv2d(inp, x[1], x[2] if len(x) > 2 else None, stride=tuple(a.get("strides", [1, 1])), padding=pad,
                             dilation=tuple(a.get("dilations", [1, 1])), groups=a.get("group", 1))
            elif op == "ConvTranspose":
                pads = a.get("pads", [0, 0, 0, 0])
                y = F.conv_transpose2d(x[0], x[1], x[2] if len(x) > 2 else None, stride=tuple(a.get("strides", [1, 1])),
                                       padding=(pads[0], pads[1]), output_padding=tuple(a.get("output_padding", [0, 0])),
                                       groups=a.get("group", 1), dilation=tuple(a.get("dilations", [1, 1])))
            elif op == "BatchNormalization":
                y = F.batch_norm(x[0], x[3], x[4], x[1], x[2], False, 0.0, a.get("epsilon", 1e-5))
            elif op == "Relu":
                y = F.relu(x[0])
            elif op == "HardSigmoid":
                y = torch.clamp(x[0] * a.get("alpha", 0.2) + a.get("beta", 0.5), 0.0, 1.0)
            elif op == "HardSwish":
                y = x[0] * torch.clamp(x[0] * (1.0 / 6.0) + 0.5, 0.0, 1.0)
            elif op == "Sigmoid":
                y = torch.sigmoid(x[0])
            elif op == "Tanh":
                y = torch.tanh(x[0])
            elif op == "Erf":
                y = torch.erf(x[0])
            elif op == "Sqrt":
                y = torch.sqrt(x[0])
            elif op == "Exp":
                y = torch.exp(x[0])
            elif op == "Abs":
                y = torch.abs(x[0])
            elif op == "Neg":
                y = -x[0]
            elif op == "Reciprocal":
                y = 1.0 / x[0]
            elif op == "Log":
                y = torch.log(x[0])
            elif op == "Softplus":
                y = F.softplus(x[0])
            elif op == "Gelu":
                y = F.gelu(x[0], approximate=a.get("approximate", "none"))
            elif op == "Pad":
                pads = [int(v) for v in x[1]] if len(x) > 1 and x[1] is not None else list(a.get("pads"))
                r = x[0].dim()
                axes = [int(v) % r for v in x[3]] if len(x) > 3 and x[3] is not None else list(range(r))
                before, after = [0] * r, [0] * r
                for k_, ax in enumerate(axes):
                    before[ax], after[ax] = pads[k_], pads[len(axes) + k_]
                mode = {"constant": "constant", "reflect": "reflect", "edge": "replicate"}[a.get("mode", "constant")]
                value = float(x[2]) if len(x) > 2 and x[2] is not None else float(a.get("value", 0.0))
                flat = []
                for d in range(r - 1, -1, -1):      # torch order: last axis first
                    flat += [before[d], after[d]]
                while len(flat) > 2 and flat[-1] == 0 and flat[-2] == 0:
                    flat = flat[:-2]
                y = F.pad(x[0], flat, mode=mode, value=value) if mode == "constant" else F.pad(x[0], flat, mode=mode)
            elif op == "PRelu":
                y = torch.where(x[0] > 0, x[0], x[0] * x[1])
            elif op == "GridSample":
                mode = {"linear": "bilinear", "bilinear": "bilinear", "nearest": "nearest"}[a.get("mode", "linear")]
                y = F.grid_sample(x[0], x[1], mode=mode, padding_mode=a.get("padding_mode", "zeros"), align_corners=bool(a.get("align_corners", 0)))
            elif op == "Clip":
                lo = float(x[1]) if len(x) > 1 and x[1] is not None else a.get("min", -3.4e38)
                hi = float(x[2]) if len(x) > 2 and x[2] is not None else a.get("max", 3.4e38)
                y = torch.clamp(x[0], lo, hi)
            elif op == "LeakyRelu":
                y = F.leaky_relu(x[0], a.get("alpha", 0.01))
            elif op in ("Add", "Mul", "Sub", "Div", "Pow"):
                f = {"Add": torch.add, "Mul": torch.mul, "Sub": torch.sub, "Div": torch.div, "Pow": torch.pow}[op]
                y = f(x[0], x[1])
            elif op == "GlobalAveragePool":
                y = x[0].mean(dim=(2, 3), keepdim=True)
            elif op in ("AveragePool", "MaxPool"):
                k = tuple(a["kernel_shape"])
                s = tuple(a.get("strides", [1, 1]))
                pads = a.get("pads", [0, 0, 0, 0])
                cm = bool(a.get("ceil_mode", 0))
                if op == "AveragePool":
                    y = F.avg_pool2d(x[0], k, s, (pads[0], pads[1]), ceil_mode=cm, count_include_pad=bool(a.get("count_include_pad", 0)))
                elif pads[0] != pads[2] or pads[1] != pads[3]:   # e.g. the "SAME" padding of a 2x2 / stride-1 pool: bottom / right only
                    y = F.max_pool2d(F.pad(x[0], (pads[1], pads[3], pads[0], pads[2]), value=float("-inf")), k, s, 0, ceil_mode=cm)
                else:
                    y = F.max_pool2d(x[0], k, s, (pads[0], pads[1]), ceil_mode=cm)
            elif op == "Resize":
                y = _resize(x[0], nd, env)
            elif op == "Concat":
                y = torch.cat(x, dim=a["axis"])
            elif op == "Identity":
                y = x[0]
            elif op == "Reshape":
                shp = [int(v) for v in x[1]]
                shp = [x[0].shape[i] if v == 0 else v for i, v in enumerate(shp)]
                y = x[0].reshape(shp)
            elif op == "Flatten":
                ax = a.get("axis", 1)
                y = x[0].reshape(int(np.prod(x[0].shape[:ax])), -1)
            elif op == "Transpose":
                y = x[0].permute(a["perm"]).contiguous()
            elif op == "Squeeze":
                axes = [int(v) for v in x[1]] if len(x) > 1 and x[1] is not None else a.get("axes")
                y = x[0]
                for ax in sorted([(v + y.dim()) % y.dim() for v in axes], reverse=True):
                    y = y.squeeze(ax)
            elif op == "Unsqueeze":
                axes = [int(v) for v in x[1]] if len(x) > 1 and x[1] is not None else a.get("axes")
                y = x[0]
                for ax in sorted(axes):
                    y = y.unsqueeze(ax)
            elif op == "Split":
                ax = a.get("axis", 0)
                if len(x) > 1 and x[1] is not None:
                    parts = torch.split(x[0], [int(v) for v in x[1]], dim=ax)
                else:
                    parts = torch.chunk(x[0], len(nd["outputs"]), dim=ax)
                for o, p in zip(nd["outputs"], parts):
                    env[o] = p.contiguous()
                continue
            elif op == "Slice":
                starts, ends = [int(v) for v in x[1]], [int(v) for v in x[2]]
                axes = [int(v) for v in x[3]] if len(x) > 3 and x[3] is not None else list(range(len(starts)))
                steps = [int(v) for v in x[4]] if len(x) > 4 and x[4] is not None else [1] * len(starts)
                sl = [slice(None)] * x[0].dim()
                for s_, e_, ax, st in zip(starts, ends, axes, steps):
                    dim = x[0].shape[ax]
                    e_ = max(min(e_, dim), -dim)
                    sl[ax] = slice(s_, e_, st)
                y = x[0][tuple(sl)].contiguous()
            elif op == "Gather":
                ax = a.get("axis", 0)
                idx = x[1].long()
                y = torch.index_select(x[0], ax, idx.reshape(-1)).reshape(
                    list(x[0].shape[:ax]) + list(idx.shape) + list(x[0].shape[ax + 1:]))
            elif op == "Shape":
                y = torch.tensor(list(x[0].shape), dtype=torch.int64)
            elif op == "Cast":
                to = {1: torch.float32, 6: torch.int32, 7: torch.int64, 9: torch.bool, 11: torch.float64}[a["to"]]
                y = x[0].to(to)
            elif op == "Constant":
                y = torch.from_numpy(np.asarray(a["value"]))
            elif op == "MatMul":
                y = torch.matmul(x[0], x[1])
            elif op == "Gemm":
                A = x[0].t() if a.get("transA", 0) else x[0]
                B = x[1].t() if a.get("transB", 0) else x[1]
                y = a.get("alpha", 1.0) * (A @ B)
                if len(x) > 2 and x[2] is not None:
                    y = y + a.get("beta", 1.0) * x[2]
            elif op == "Softmax":
                y = torch.softmax(x[0], dim=a.get("axis", -1))
            elif op == "LayerNormalization":
                ax = a.get("axis", -1)
                shape = x[0].shape[ax:] if ax < 0 else x[0].shape[ax:]
                y = F.layer_norm(x[0], tuple(shape), x[1], x[2] if len(x) > 2 else None, a.get("epsilon", 1e-5))
            elif op in ("ReduceMean", "ReduceSum", "ReduceMax", "ReduceMin", "ReduceProd"):
                axes = a.get("axes") or ([int(v) for v in x[1]] if len(x) > 1 and x[1] is not None else list(range(x[0].dim())))
                kd = bool(a.get("keepdims", 1))
                if op == "ReduceMean":
                    y = x[0].mean(dim=axes, keepdim=kd)
                elif op == "ReduceSum":
                    y = x[0].sum(dim=axes, keepdim=kd)
                elif op == "ReduceMax":
                    y = torch.amax(x[0], dim=axes, keepdim=kd)
                elif op == "ReduceMin":
                    y = torch.amin(x[0], dim=axes, keepdim=kd)
                else:
                    y = x[0]
                    for ax in sorted([(v + y.dim()) % y.dim() for v in axes], reverse=True):
                        y = y.prod(dim=ax, keepdim=kd)
            elif op in ("ArgMax", "ArgMin"):
                ax = a.get("axis", 0)
                xx = x[0].flip(ax) if a.get("select_last_index", 0) else x[0]
                y = (torch.argmax if op == "ArgMax" else torch.argmin)(xx, dim=ax, keepdim=bool(a.get("keepdims", 1)))
                if a.get("select_last_index", 0):
                    y = x[0].shape[ax] - 1 - y
            elif op in ("Max", "Min"):
                y = (torch.maximum if op == "Max" else torch.minimum)(*torch.broadcast_tensors(x[0], x[1]))
            elif op in ("Equal", "Less", "Greater"):
                y = {"Equal": torch.eq, "Less": torch.lt, "Greater": torch.gt}[op](x[0], x[1])
            elif op in ("And", "Or"):
                y = torch.logical_and(x[0], x[1]) if op == "And" else torch.logical_or(x[0], x[1])
            elif op == "Not":
                y = torch.logical_not(x[0])
            elif op == "Where":
                y = torch.where(x[0].bool(), x[1], x[2])
            elif op == "Expand":
                shp = [int(v) for v in x[1]]
                y = x[0] * torch.ones(shp, dtype=x[0].dtype) if x[0].dtype != torch.bool else x[0].expand(torch.broadcast_shapes(x[0].shape, tuple(shp)))
            elif op == "Tile":
                y = x[0].repeat([int(v) for v in x[1]])
            elif op == "ConstantOfShape":
                v = a.get("value")
                v = np.asarray(v).reshape(-1) if v is not None else np.zeros(1, np.float32)
                y = torch.full([int(s_) for s_ in x[0]], v[0].item(), dtype=torch.from_numpy(v).dtype)
            elif op == "Range":
                y = torch.arange(x[0].item(), x[1].item(), x[2].item(), dtype=x[0].dtype)
            elif op == "Floor":
                y = torch.floor(x[0])
            elif op == "Ceil":
                y = torch.ceil(x[0])
            elif op == "Round":
                y = torch.round(x[0])
            else:
                raise NotImplementedError(op)
            env[nd["outputs"][0]] = y
    names = want or model["outputs"]
    return [env[n].numpy() for n in names]
