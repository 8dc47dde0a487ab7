"""A SECOND, torch-free evaluator of the ONNX graphs the product runs (TEST INFRASTRUCTURE ONLY -- nothing under oar_ocr_amd/ imports this).

Why: every network-parity claim of this repository is "HIP engine vs oracle/onnx_ref.py", a torch-CPU interpreter.  ONNX Runtime -- the
reference's own engine (oar-ocr-core/src/core/inference/ort_infer_execution.rs:178,281) -- cannot be installed here, so this file restates the
operator semantics a second time, from the ONNX operator specification, in plain numpy with float64 accumulation and none of torch's kernels
(convolutions as explicit tap sums, pooling as window loops, LayerNormalization / Softmax / Resize spelled out).  tests/test_oracle_second_opinion_cpu.py
holds onnx_ref.py to it on every synthetic graph family: an error in how onnx_ref maps an ONNX attribute onto a torch call (padding order,
coordinate modes, count_include_pad, axis conventions, ...) shows up there instead of being baked into both sides of a GPU parity test.

It shares onnx_ref's protobuf reader (parse_model) and nothing else."""
from __future__ import annotations

import numpy as np

from .onnx_ref import parse_model

F64 = np.float64


def _conv(x, w, b, a):
    n, c, h, wd = x.shape
    co, cpg, kh, kw = w.shape
    g = a.get("group", 1)
    sh, sw = a.get("strides", [1, 1])
    dh, dw = a.get("dilations", [1, 1])
    pt, pl, pb, pr = a.get("pads", [0, 0, 0, 0])
    xp = np.zeros((n, c, h + pt + pb, wd + pl + pr), F64)
    xp[:, :, pt:pt + h, pl:pl + wd] = x
    ho = (h + pt + pb - dh * (kh - 1) - 1) // sh + 1
    wo = (wd + pl + pr - dw * (kw - 1) - 1) // sw + 1
    cog = co // g
    y = np.zeros((n, g, cog, ho, wo), F64)
    wg = w.astype(F64).reshape(g, cog, cpg, kh, kw)
    for i in range(kh):
        for j in range(kw):
            patch = xp[:, :, i * dh:i * dh + sh * (ho - 1) + 1:sh, j * dw:j * dw + sw * (wo - 1) + 1:sw].reshape(n, g, cpg, ho, wo)
            y += np.einsum("ngchw,goc->ngohw", patch, wg[:, :, :, i, j])
    y = y.reshape(n, co, ho, wo)
    if b is not None:
        y += b.astype(F64).reshape(1, co, 1, 1)
    return y


def _conv_transpose(x, w, b, a):
    n, c, h, wd = x.shape
    ci, cog, kh, kw = w.shape          # [Cin, Cout / group, kh, kw]
    g = a.get("group", 1)
    sh, sw = a.get("strides", [1, 1])
    dh, dw = a.get("dilations", [1, 1])
    pt, pl, pb, pr = a.get("pads", [0, 0, 0, 0])
    oph, opw = a.get("output_padding", [0, 0])
    ho = (h - 1) * sh - pt - pb + dh * (kh - 1) + oph + 1
    wo = (wd - 1) * sw - pl - pr + dw * (kw - 1) + opw + 1
    full = np.zeros((n, g * cog, (h - 1) * sh + dh * (kh - 1) + 1 + oph, (wd - 1) * sw + dw * (kw - 1) + 1 + opw), F64)
    cig = ci // g
    xg = x.astype(F64).reshape(n, g, cig, h, wd)
    wg = w.astype(F64).reshape(g, cig, cog, kh, kw)
    for i in range(kh):
        for j in range(kw):
            contrib = np.einsum("ngchw,gco->ngohw", xg, wg[:, :, :, i, j]).reshape(n, g * cog, h, wd)
            full[:, :, i * dh:i * dh + sh * (h - 1) + 1:sh, j * dw:j * dw + sw * (wd - 1) + 1:sw] += contrib
    y = full[:, :, pt:pt + ho, pl:pl + wo]
    if b is not None:
        y = y + b.astype(F64).reshape(1, -1, 1, 1)
    return y


def _pool(x, a, kind):
    n, c, h, w = x.shape
    kh, kw = a["kernel_shape"]
    sh, sw = a.get("strides", [1, 1])
    pt, pl, pb, pr = a.get("pads", [0, 0, 0, 0])
    ceil = bool(a.get("ceil_mode", 0))
    rnd = (lambda v: -(-v // 1)) if ceil else (lambda v: v // 1)
    ho = int(rnd((h + pt + pb - kh) / sh)) + 1 if not ceil else int(np.ceil((h + pt + pb - kh) / sh)) + 1
    wo = int(rnd((w + pl + pr - kw) / sw)) + 1 if not ceil else int(np.ceil((w + pl + pr - kw) / sw)) + 1
    if ceil:   # a last window that would start beyond the input + leading padding is dropped (ONNX / torch output-shape rule)
        if (ho - 1) * sh >= h + pt:
            ho -= 1
        if (wo - 1) * sw >= w + pl:
            wo -= 1
    include = bool(a.get("count_include_pad", 0))
    y = np.zeros((n, c, ho, wo), F64)
    for oh in range(ho):
        for ow in range(wo):
            y0, x0 = oh * sh - pt, ow * sw - pl
            ya, yb, xa, xb = max(y0, 0), min(y0 + kh, h), max(x0, 0), min(x0 + kw, w)
            win = x[:, :, ya:yb, xa:xb].astype(F64)
            if kind == "max":
                y[:, :, oh, ow] = win.max(axis=(2, 3))
            else:
                # count_include_pad: the window clipped to the padded extent -- padding counts, a ceil_mode overhang beyond it does not (ONNX opset >= 19 reference, torch)
                cnt = (min(y0 + kh, h + pb) - y0) * (min(x0 + kw, w + pr) - x0) if include else (yb - ya) * (xb - xa)
                y[:, :, oh, ow] = win.sum(axis=(2, 3)) / cnt
    return y


def _resize(x, node, env):
    a = node["attrs"]
    ins = node["inputs"]
    scales = env.get(ins[2]) if len(ins) > 2 and ins[2] else None
    sizes = env.get(ins[3]) if len(ins) > 3 and ins[3] else None
    n, c, h, w = x.shape
    if sizes is not None and np.size(sizes):
        oh, ow = int(sizes[2]), int(sizes[3])
        sh, sw = oh / h, ow / w
    else:
        sh, sw = float(scales[2]), float(scales[3])
        oh, ow = int(np.floor(h * sh)), int(np.floor(w * sw))
    mode = a.get("mode", "nearest")
    ctm = a.get("coordinate_transformation_mode", "half_pixel")

    def src(o_len, s, n_in):
        o = np.arange(o_len, dtype=F64)
        if ctm == "asymmetric":
            return o / s
        if ctm == "align_corners":
            return o * (n_in - 1) / (o_len - 1) if o_len > 1 else np.zeros(o_len)
        if ctm == "pytorch_half_pixel":
            return (o + 0.5) / s - 0.5 if o_len > 1 else np.zeros(o_len)
        if ctm == "half_pixel":
            return (o + 0.5) / s - 0.5
        raise NotImplementedError(ctm)
    if mode == "nearest":
        nm = a.get("nearest_mode", "round_prefer_floor")

        def idx(o_len, s, n_in):
            v = src(o_len, s, n_in)
            r = {"floor": np.floor(v), "ceil": np.ceil(v), "round_prefer_floor": np.ceil(v - 0.5), "round_prefer_ceil": np.floor(v + 0.5)}[nm]
            return np.clip(r, 0, n_in - 1).astype(np.int64)
        return x[:, :, idx(oh, sh, h)][:, :, :, idx(ow, sw, w)]
    if mode == "linear":
        def taps(o_len, s, n_in):
            v = np.clip(src(o_len, s, n_in), 0, n_in - 1)      # ONNX clamps the source coordinate to the image for linear mode
            i0 = np.floor(v).astype(np.int64)
            i1 = np.minimum(i0 + 1, n_in - 1)
            return i0, i1, v - i0
        y0, y1, fy = taps(oh, sh, h)
        x0, x1, fx = taps(ow, sw, w)
        xd = x.astype(F64)
        top = xd[:, :, y0][:, :, :, x0] * (1 - fx) + xd[:, :, y0][:, :, :, x1] * fx
        bot = xd[:, :, y1][:, :, :, x0] * (1 - fx) + xd[:, :, y1][:, :, :, x1] * fx
        return top * (1 - fy)[None, None, :, None] + bot * fy[None, None, :, None]
    raise NotImplementedError(mode)


def _grid_sample(x, grid, a):
    """bilinear, padding zeros / border, align_corners 0 / 1, per the ONNX GridSample specification"""
    pad = a.get("padding_mode", "zeros")
    assert a.get("mode", "linear") in ("linear", "bilinear") and pad in ("zeros", "border")
    n, c, h, w = x.shape
    ac = bool(a.get("align_corners", 0))

    def unnorm(v, size):
        return (v + 1) / 2 * (size - 1) if ac else ((v + 1) * size - 1) / 2
    gx, gy = unnorm(grid[..., 0].astype(F64), w), unnorm(grid[..., 1].astype(F64), h)
    if pad == "border":     # the sampling position itself is clamped to the image
        gx, gy = np.clip(gx, 0, w - 1), np.clip(gy, 0, h - 1)
    x0, y0 = np.floor(gx).astype(np.int64), np.floor(gy).astype(np.int64)
    out = np.zeros((n, c) + gx.shape[1:], F64)
    xd = x.astype(F64)
    for dy in (0, 1):
        for dx in (0, 1):
            xi, yi = x0 + dx, y0 + dy
            wgt = (1 - np.abs(gx - xi)) * (1 - np.abs(gy - yi))
            ok = (xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)
            xc, yc = np.clip(xi, 0, w - 1), np.clip(yi, 0, h - 1)
            for b in range(n):
                out[b] += xd[b][:, yc[b], xc[b]] * (wgt[b] * ok[b])[None]
    return out


def run(model, feeds: dict):
    """model: parse_model() dict (or raw bytes).  feeds: name -> np.ndarray.  Returns the graph outputs as float32 / integer arrays."""
    if isinstance(model, (bytes, bytearray)):
        model = parse_model(model)
    env = {k: np.asarray(v) for k, v in model["inits"].items()}
    for k, v in feeds.items():
        env[k] = np.asarray(v)
    for nd in model["nodes"]:
        op, a = nd["op"], nd["attrs"]
        x = [env[i] if i else None for i in nd["inputs"]]
        if op == "Conv":
            y = _conv(x[0].astype(F64), x[1], x[2] if len(x) > 2 else None, a)
        elif op == "ConvTranspose":
            y = _conv_transpose(x[0], x[1], x[2] if len(x) > 2 else None, a)
        elif op == "BatchNormalization":
            sc, bi, mu, var = (t.astype(F64).reshape(1, -1, 1, 1) for t in x[1:5])
            y = (x[0].astype(F64) - mu) / np.sqrt(var + a.get("epsilon", 1e-5)) * sc + bi
        elif op == "Relu":
            y = np.maximum(x[0], 0)
        elif op == "HardSigmoid":
            y = np.clip(x[0].astype(F64) * a.get("alpha", 0.2) + a.get("beta", 0.5), 0.0, 1.0)
        elif op == "HardSwish":
            y = x[0].astype(F64) * np.clip(x[0].astype(F64) / 6.0 + 0.5, 0.0, 1.0)
        elif op == "Sigmoid":
            y = 1.0 / (1.0 + np.exp(-x[0].astype(F64)))
        elif op == "Tanh":
            y = np.tanh(x[0].astype(F64))
        elif op == "Softplus":
            y = np.logaddexp(0.0, x[0].astype(F64))
        elif op == "PRelu":
            y = np.where(x[0] > 0, x[0], x[0].astype(F64) * x[1])
        elif op in ("Add", "Mul", "Sub", "Div"):
            f = {"Add": np.add, "Mul": np.multiply, "Sub": np.subtract, "Div": np.divide}[op]
            both_int = all(np.issubdtype(t.dtype, np.integer) for t in x[:2])
            y = f(x[0], x[1]) if both_int and op != "Div" else f(x[0].astype(F64), x[1].astype(F64))
        elif op == "GlobalAveragePool":
            y = x[0].astype(F64).mean(axis=(2, 3), keepdims=True)
        elif op == "AveragePool":
            y = _pool(x[0], a, "avg")
        elif op == "MaxPool":
            y = _pool(x[0], a, "max")
        elif op == "Resize":
            y = _resize(x[0], nd, env)
        elif op == "GridSample":
            y = _grid_sample(x[0], x[1], a)
        elif op == "Concat":
            y = np.concatenate([t.astype(F64) if t.dtype.kind == "f" else t for t in x], axis=a["axis"])
        elif op == "Identity":
            y = x[0]
        elif op == "Reshape":
            shp = [int(v) for v in x[1]]
            y = x[0].reshape([x[0].shape[i] if v == 0 else v for i, v in enumerate(shp)])
        elif op == "Flatten":
            ax = a.get("axis", 1)
            y = x[0].reshape(int(np.prod(x[0].shape[:ax])), -1)
        elif op == "Transpose":
            y = np.transpose(x[0], a["perm"])
        elif op == "Squeeze":
            axes = [int(v) for v in x[1]] if len(x) > 1 and x[1] is not None else a.get("axes")
            y = np.squeeze(x[0], axis=tuple(int(v) for v in axes))
        elif op == "Unsqueeze":
            axes = [int(v) for v in x[1]] if len(x) > 1 and x[1] is not None else a.get("axes")
            y = x[0]
            for ax in sorted(int(v) for v in axes):
                y = np.expand_dims(y, ax)
        elif op == "Split":
            ax = a.get("axis", 0)
            if len(x) > 1 and x[1] is not None:
                parts = np.split(x[0], np.cumsum([int(v) for v in x[1]])[:-1], axis=ax)
            else:
                parts = np.split(x[0], len(nd["outputs"]), axis=ax)
            for o, p in zip(nd["outputs"], parts):
                env[o] = p
            continue
        elif op == "Slice":
            starts, ends = [int(v) for v in x[1]], [int(v) for v in x[2]]
            axes = [int(v) for v in x[3]] if len(x) > 3 and x[3] is not None else list(range(len(starts)))
            steps = [int(v) for v in x[4]] if len(x) > 4 and x[4] is not None else [1] * len(starts)
            sl = [slice(None)] * x[0].ndim
            for s_, e_, ax, st in zip(starts, ends, axes, steps):
                dim = x[0].shape[ax]
                sl[ax] = slice(s_, max(min(e_, dim), -dim), st)
            y = x[0][tuple(sl)]
        elif op == "Cast":
            y = x[0].astype({1: np.float32, 6: np.int32, 7: np.int64, 9: np.bool_, 11: np.float64}[a["to"]])
        elif op == "MatMul":
            y = np.matmul(x[0].astype(F64), x[1].astype(F64))
        elif op == "Gemm":
            A = x[0].astype(F64).T if a.get("transA", 0) else x[0].astype(F64)
            B = x[1].astype(F64).T if a.get("transB", 0) else x[1].astype(F64)
            y = a.get("alpha", 1.0) * (A @ B)
            if len(x) > 2 and x[2] is not None:
                y = y + a.get("beta", 1.0) * x[2]
        elif op == "Softmax":
            ax = a.get("axis", -1)
            z = x[0].astype(F64)
            e = np.exp(z - z.max(axis=ax, keepdims=True))
            y = e / e.sum(axis=ax, keepdims=True)
        elif op == "LayerNormalization":
            ax = a.get("axis", -1)
            z = x[0].astype(F64)
            axes = tuple(range(ax % z.ndim, z.ndim))
            mu = z.mean(axis=axes, keepdims=True)
            var = ((z - mu) ** 2).mean(axis=axes, keepdims=True)
            y = (z - mu) / np.sqrt(var + a.get("epsilon", 1e-5)) * x[1]
            if len(x) > 2 and x[2] is not None:
                y = y + x[2]
        elif op == "ReduceMax":
            axes = a.get("axes") or ([int(v) for v in x[1]] if len(x) > 1 and x[1] is not None else list(range(x[0].ndim)))
            y = x[0].max(axis=tuple(axes), keepdims=bool(a.get("keepdims", 1)))
        elif op == "ArgMax":
            ax = a.get("axis", 0)
            if a.get("select_last_index", 0):
                y = x[0].shape[ax] - 1 - np.argmax(np.flip(x[0], ax), axis=ax)
            else:
                y = np.argmax(x[0], axis=ax)          # first maximum, as the specification says
            if a.get("keepdims", 1):
                y = np.expand_dims(y, ax)
            y = y.astype(np.int64)
        else:
            raise NotImplementedError(f"onnx_np: operator {op}")
        env[nd["outputs"][0]] = y
    outs = []
    for o in model["outputs"]:
        v = env[o["name"] if isinstance(o, dict) else o]
        outs.append(v.astype(np.float32) if v.dtype == F64 else v)
    return outs
