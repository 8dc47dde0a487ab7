"""Randomised single-layer sweep of the engine's convolution / attention dispatch against the torch-CPU oracle (round 6): random shapes around every
eligibility boundary of the bf16x6 kernels (weight-stationary one- and two-fragment tiles, output-stationary, row-streaming 3x3, LDS-tiled large kernel,
grouped, multi-source concat reads, streaming attention), random epilogues (bias, ReLU / hard-swish / GELU / none, residual).  Reports, per case, the kernel
classes that ran and the largest |difference| relative to the output scale; the tolerance is the engine tests' 2e-4.
usage: python tools/op_fuzz.py [n_cases] [seed] [kind]      kind: p2o | neck | net | convmisc | convt | gemm | gridsample | eltwise | reduce | resize | shape | svtr_block | dbhead | fpn | pool | dsblock | dschain | conv1x1 | conv3x3 | convk | grouped | concat | attention | all"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
from oar_ocr_amd import api
from oar_ocr_amd.synth.onnx_writer import GraphBuilder
from oracle import onnx_ref

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
only = sys.argv[3] if len(sys.argv) > 3 else "all"
rng = np.random.default_rng(seed)
TOL = 2e-4
import os
DRY = os.environ.get("OP_FUZZ_DRY") == "1"
RESHAPE = os.environ.get("OP_FUZZ_RESHAPE") == "1"   # run every case on two input shapes through ONE engine (A, B, A again)
RESHAPE_KINDS = {"conv1x1", "conv3x3", "convk", "grouped", "concat", "dsblock", "dschain", "net", "attention", "svtr_block", "neck", "convmisc", "eltwise", "dbhead"}


def act(g, y, kind):
    if kind == "relu":
        return g.op("Relu", [y])
    if kind == "hswish":
        return g.op("HardSwish", [y])
    if kind == "gelu":
        e = g.op("Erf", [g.op("Div", [y, g.init(np.array(np.sqrt(2.0), np.float32), "c")])])
        return g.op("Mul", [g.op("Mul", [y, g.op("Add", [e, g.init(np.array(1.0, np.float32), "c")])]), g.init(np.array(0.5, np.float32), "c")])
    return y


def conv(g, x, cin, cout, k, stride=1, groups=1, bias=True, pad=None):
    pad = k // 2 if pad is None else pad
    w = (rng.standard_normal((cout, cin // groups, k, k)) * np.sqrt(1.0 / (k * k * cin // groups))).astype(np.float32)
    ins = [x, g.init(w)] + ([g.init((0.2 * rng.standard_normal(cout)).astype(np.float32))] if bias else [])
    return g.op("Conv", ins, kernel_shape=[k, k], strides=[stride, stride], pads=[pad] * 4, group=groups, dilations=[1, 1])


g_last_c = [0]


def stem(g, c):
    """a channels-last producer in front of the layer under test (graph inputs are NCHW)"""
    g_last_c[0] = c
    return g.op("Relu", [conv(g, "x", 8, c, 1, bias=False)])


def case_conv1x1():
    cin = int(rng.choice([24, 40, 48, 64, 96, 104, 128, 192, 256, 320, 384, 512, 768]))
    cout = int(rng.choice([16, 24, 48, 64, 96, 112, 128, 192, 256, 384, 512, 1024]))
    n, h, w = int(rng.integers(1, 5)), int(rng.integers(8, 200)), int(rng.integers(8, 260))
    a, res = str(rng.choice(["none", "relu", "hswish", "gelu"])), bool(rng.random() < 0.3) and cin == cout
    g = GraphBuilder("f")
    g.add_input("x", ["N", 8, "H", "W"])
    t = stem(g, cin)
    y = act(g, conv(g, t, cin, cout, 1, bias=bool(rng.random() < 0.8)), a)
    if res:
        y = g.op("Add", [y, t])
    g.add_output(y, ["N", cout, "H", "W"])
    return f"1x1 {cin}->{cout} {n}x{h}x{w} {a}{' +res' if res else ''}", g.model(), (n, 8, h, w)


def case_conv3x3():
    cin = int(rng.choice([16, 32, 48, 64, 96, 128, 160, 256]))
    cout = int(rng.choice([8, 16, 32, 48, 64, 96, 128, 256]))
    n, h, w = int(rng.integers(1, 4)), int(rng.integers(6, 150)), int(rng.integers(6, 200))
    s = int(rng.choice([1, 1, 1, 2]))
    a = str(rng.choice(["none", "relu", "hswish"]))
    g = GraphBuilder("f")
    g.add_input("x", ["N", 8, "H", "W"])
    y = act(g, conv(g, stem(g, cin), cin, cout, 3, stride=s), a)
    g.add_output(y, ["N", cout, "H2", "W2"])
    return f"3x3/{s} {cin}->{cout} {n}x{h}x{w} {a}", g.model(), (n, 8, h, w)


def case_convk():
    k = int(rng.choice([5, 7, 9, 9, 9]))
    cin = int(rng.choice([32, 64, 96, 128, 256]))
    cout = int(rng.choice([16, 32, 48, 64, 80, 128]))
    n, h, w = int(rng.integers(1, 9)), int(rng.integers(9, 130)), int(rng.integers(9, 170))
    a = str(rng.choice(["none", "relu"]))
    g = GraphBuilder("f")
    g.add_input("x", ["N", 8, "H", "W"])
    y = act(g, conv(g, stem(g, cin), cin, cout, k, bias=bool(rng.random() < 0.7)), a)
    g.add_output(y, ["N", cout, "H", "W"])
    return f"{k}x{k} {cin}->{cout} {n}x{h}x{w} {a}", g.model(), (n, 8, h, w)


def case_grouped():
    cg = int(rng.choice([8, 16, 32, 32, 32, 64]))
    groups = int(rng.integers(2, 9))
    c = cg * groups
    k = int(rng.choice([3, 5, 5]))
    n, h, w = int(rng.integers(1, 9)), int(rng.integers(6, 40)), int(rng.integers(20, 900))
    res = bool(rng.random() < 0.4)
    g = GraphBuilder("f")
    g.add_input("x", ["N", 8, "H", "W"])
    t = stem(g, c)
    y = conv(g, t, c, c, k, groups=groups)
    if res:
        y = g.op("Add", [y, t])
    g.add_output(y, ["N", c, "H", "W"])
    return f"grouped {k}x{k} {groups}x{cg} {n}x{h}x{w}{' +res' if res else ''}", g.model(), (n, 8, h, w)


def case_concat():
    nsrc = int(rng.integers(2, 9))
    chans = [int(rng.choice([8, 16, 24, 40, 48, 64])) for _ in range(nsrc)]
    cout = int(rng.choice([32, 48, 64, 128, 256]))
    n, h, w = int(rng.integers(1, 4)), int(rng.integers(10, 200)), int(rng.integers(10, 260))
    g = GraphBuilder("f")
    g.add_input("x", ["N", 8, "H", "W"])
    t = stem(g, chans[0])
    outs = [t]
    for i in range(1, nsrc):
        t = g.op("Relu", [conv(g, t, chans[i - 1], chans[i], int(rng.choice([1, 3])))])
        outs.append(t)
    a = str(rng.choice(["none", "relu", "hswish"]))
    y = act(g, conv(g, g.op("Concat", outs, axis=1), sum(chans), cout, 1), a)
    g.add_output(y, ["N", cout, "H", "W"])
    return f"concat {chans}->{cout} {n}x{h}x{w} {a}", g.model(), (n, 8, h, w)


def case_attention():
    heads = int(rng.integers(1, 13))
    hd = int(rng.choice([32, 32, 32, 16, 64]))
    dim = heads * hd
    n, T = int(rng.integers(1, 4)), int(rng.choice([int(rng.integers(2, 40)), int(rng.integers(33, 700)), int(rng.integers(700, 2600))]))
    gain = float(rng.choice([1.0, 1.0, 3.0]))
    g = GraphBuilder("f")
    g.add_input("x", ["N", "T", dim])
    w = (gain * rng.standard_normal((dim, 3 * dim)) / np.sqrt(dim)).astype(np.float32)
    qkv = g.op("Add", [g.op("MatMul", ["x", g.init(w)]), g.init((0.1 * rng.standard_normal(3 * dim)).astype(np.float32))])
    qkv = g.op("Transpose", [g.op("Reshape", [qkv, g.init(np.array([0, -1, 3, heads, hd], np.int64), "shape")])], perm=[2, 0, 3, 1, 4])
    q, k, v = g.op("Split", [qkv], n_out=3, axis=0)
    ax0 = g.init(np.array([0], np.int64), "axes")
    q, k, v = g.op("Squeeze", [q, ax0]), g.op("Squeeze", [k, ax0]), g.op("Squeeze", [v, ax0])
    q = g.op("Mul", [q, g.init(np.array(hd ** -0.5, np.float32), "scale")])
    att = g.op("Softmax", [g.op("MatMul", [q, g.op("Transpose", [k], perm=[0, 1, 3, 2])])], axis=-1)
    o = g.op("Reshape", [g.op("Transpose", [g.op("MatMul", [att, v])], perm=[0, 2, 1, 3]), g.init(np.array([0, -1, dim], np.int64), "shape")])
    g.add_output(o, ["N", "T", dim])
    return f"attention {heads}x{hd} {n}x{T} gain {gain}", g.model(), (n, T, dim)


def _ds(g, t, c, cout, k, stride, a, se):
    """depthwise k x k + act -> [squeeze-excite gate] -> pointwise 1 x 1 + act: the PP-LCNet block"""
    w = (rng.standard_normal((c, 1, k, k)) * np.sqrt(1.0 / (k * k))).astype(np.float32)
    d = g.op("Conv", [t, g.init(w), g.init((0.1 * rng.standard_normal(c)).astype(np.float32))], kernel_shape=[k, k], strides=list(stride), pads=[k // 2] * 4, group=c, dilations=[1, 1])
    d = act(g, d, a)
    if se:
        r = max(4, c // 4)
        p = g.op("GlobalAveragePool", [d])
        f = g.op("Relu", [conv(g, p, c, r, 1)])
        f = g.op("HardSigmoid", [conv(g, f, r, c, 1)], alpha=1.0 / 6.0, beta=0.5)
        d = g.op("Mul", [d, f])
    return act(g, conv(g, d, c, cout, 1), a)


def case_dsblock():
    c = int(rng.choice([8, 16, 24, 32, 48, 64, 80, 96, 128, 160, 192, 256, 320]))
    cout = int(rng.choice([16, 24, 32, 48, 64, 80, 96, 128, 160, 192, 256, 320]))
    k = int(rng.choice([3, 3, 5]))
    stride = [(1, 1), (1, 1), (2, 2), (2, 1), (1, 2)][int(rng.integers(0, 5))]
    a = str(rng.choice(["hswish", "hswish", "relu", "none"]))
    se = bool(rng.random() < 0.35)
    res = stride == (1, 1) and c == cout and bool(rng.random() < 0.5)
    n, h, w = int(rng.integers(1, 9)), int(rng.integers(5, 130)), int(rng.integers(5, 330))
    g = GraphBuilder("f")
    g.add_input("x", ["N", 8, "H", "W"])
    t = stem(g, c)
    y = _ds(g, t, c, cout, k, stride, a, se)
    if res:
        y = g.op("Add", [y, t])
    g.add_output(y, ["N", cout, "H2", "W2"])
    return f"dsblock {c}->{cout} k{k} s{stride} {a}{' se' if se else ''}{' +res' if res else ''} {n}x{h}x{w}", g.model(), (n, 8, h, w)


def case_dschain():
    """two to four consecutive blocks (pair fusion of the row-streaming kernels, the chunk-streamed / producer-consumer kernels at 192 channels)"""
    depth = int(rng.integers(2, 5))
    c = int(rng.choice([16, 24, 48, 96, 192]))
    n, h, w = int(rng.integers(1, 7)), int(rng.integers(6, 60)), int(rng.integers(16, 330))
    g = GraphBuilder("f")
    g.add_input("x", ["N", 8, "H", "W"])
    t = stem(g, c)
    desc = [str(c)]
    for _ in range(depth):
        cout = int(rng.choice([c, c, min(2 * c, 192)]))
        k = 5 if c >= 192 and rng.random() < 0.7 else 3
        t = _ds(g, t, c, cout, k, (1, 1), "hswish", bool(rng.random() < 0.2))
        c = cout
        desc.append(f"k{k}:{c}")
    g.add_output(t, ["N", c, "H", "W"])
    return f"dschain {'-'.join(desc)} {n}x{h}x{w}", g.model(), (n, 8, h, w)


def _linear(g, x, cin, cout, bias=True):
    w = (rng.standard_normal((cin, cout)) / np.sqrt(cin)).astype(np.float32)
    y = g.op("MatMul", [x, g.init(w)])
    return g.op("Add", [y, g.init((0.1 * rng.standard_normal(cout)).astype(np.float32))]) if bias else y


def _ln(g, x, dim):
    return g.op("LayerNormalization", [x, g.init((1.0 + 0.1 * rng.standard_normal(dim)).astype(np.float32)), g.init((0.1 * rng.standard_normal(dim)).astype(np.float32))], axis=-1, epsilon=1e-5)


def _mha(g, x, dim, heads):
    hd = dim // heads
    qkv = g.op("Transpose", [g.op("Reshape", [_linear(g, x, dim, 3 * dim), g.init(np.array([0, -1, 3, heads, hd], np.int64), "shape")])], perm=[2, 0, 3, 1, 4])
    q, k, v = g.op("Split", [qkv], n_out=3, axis=0)
    ax0 = g.init(np.array([0], np.int64), "axes")
    q, k, v = g.op("Squeeze", [q, ax0]), g.op("Squeeze", [k, ax0]), g.op("Squeeze", [v, ax0])
    q = g.op("Mul", [q, g.init(np.array(hd ** -0.5, np.float32), "scale")])
    att = g.op("Softmax", [g.op("MatMul", [q, g.op("Transpose", [k], perm=[0, 1, 3, 2])])], axis=-1)
    o = g.op("Reshape", [g.op("Transpose", [g.op("MatMul", [att, v])], perm=[0, 2, 1, 3]), g.init(np.array([0, -1, dim], np.int64), "shape")])
    return _linear(g, o, dim, dim)


def case_svtr_block():
    """one or two transformer blocks on [N, T, dim] tokens: pre- or post-LayerNorm, MLP ratio 2 or 4, GELU / swish-free activations the exporters emit"""
    heads = int(rng.integers(1, 13))
    hd = int(rng.choice([32, 32, 16, 8, 64, 15]))
    dim = heads * hd
    n, T = int(rng.integers(1, 40)), int(rng.choice([int(rng.integers(1, 41)), int(rng.integers(41, 200)), int(rng.integers(200, 900))]))
    if n * T > 12000:
        n = max(1, 12000 // T)
    pre, ratio, a = bool(rng.random() < 0.6), int(rng.choice([2, 4])), str(rng.choice(["gelu", "relu", "hswish"]))
    g = GraphBuilder("f")
    g.add_input("x", ["N", "T", dim])
    t = "x"
    for _ in range(int(rng.integers(1, 3))):
        if pre:
            t = g.op("Add", [t, _mha(g, _ln(g, t, dim), dim, heads)])
            t = g.op("Add", [t, _linear(g, act(g, _linear(g, _ln(g, t, dim), dim, ratio * dim), a), ratio * dim, dim)])
        else:
            t = _ln(g, g.op("Add", [t, _mha(g, t, dim, heads)]), dim)
            t = _ln(g, g.op("Add", [t, _linear(g, act(g, _linear(g, t, dim, ratio * dim), a), ratio * dim, dim)]), dim)
    g.add_output(t, ["N", "T", dim])
    return f"svtr {'pre' if pre else 'post'}-LN {heads}x{hd} mlp{ratio} {a} {n}x{T}", g.model(), (n, T, dim)


def case_dbhead():
    """DB head: 3x3 conv (cin -> cin / 4) + ReLU -> ConvTranspose 2x2 / 2 + ReLU -> ConvTranspose 2x2 / 2 -> Sigmoid"""
    cin = int(rng.choice([16, 24, 64, 96, 256]))
    mid = max(4, cin // 4)
    n, h, w = int(rng.integers(1, 5)), int(rng.integers(4, 120)), int(rng.integers(4, 160))
    g = GraphBuilder("f")
    g.add_input("x", ["N", 8, "H", "W"])
    t = g.op("Relu", [conv(g, stem(g, cin), cin, mid, 3)])
    w1 = (rng.standard_normal((mid, mid, 2, 2)) * np.sqrt(1.0 / mid)).astype(np.float32)
    t = g.op("Relu", [g.op("ConvTranspose", [t, g.init(w1), g.init((0.1 * rng.standard_normal(mid)).astype(np.float32))], kernel_shape=[2, 2], strides=[2, 2], pads=[0, 0, 0, 0], group=1, dilations=[1, 1])])
    w2 = (rng.standard_normal((mid, 1, 2, 2)) * np.sqrt(1.0 / mid)).astype(np.float32)
    t = g.op("Sigmoid", [g.op("ConvTranspose", [t, g.init(w2), g.init(np.array([0.05], np.float32))], kernel_shape=[2, 2], strides=[2, 2], pads=[0, 0, 0, 0], group=1, dilations=[1, 1])])
    g.add_output(t, ["N", 1, "H4", "W4"])
    return f"dbhead {cin}->{mid}->1 {n}x{h}x{w}", g.model(), (n, 8, h, w)


def case_fpn():
    """top-down FPN sums over three or four levels (lateral 1x1 + nearest x2 + Add), 3x3 smoothing per level, nearest x2 / x4 / x8 to the finest level, Concat"""
    levels = int(rng.integers(3, 5))
    chans = [int(rng.choice([16, 24, 48, 96])) * (1 << i) for i in range(levels)]
    width = int(rng.choice([24, 48, 64, 96]))
    outc = int(rng.choice([8, 16, 24]))
    n = int(rng.integers(1, 4))
    h, w = int(rng.integers(1, 12)) * (1 << (levels - 1)), int(rng.integers(1, 16)) * (1 << (levels - 1))
    g = GraphBuilder("f")
    g.add_input("x", ["N", 8, "H", "W"])
    feats, t, c = [], "x", 8
    for i in range(levels):
        t = g.op("Relu", [conv(g, t, c, chans[i], 3, stride=1 if i == 0 else 2)])
        c = chans[i]
        feats.append(t)
    scales = g.init(np.array([1, 1, 2, 2], np.float32), "scales")
    top = conv(g, feats[-1], chans[-1], width, 1, bias=False)
    ins = [top]
    for i in range(levels - 2, -1, -1):
        up = g.op("Resize", [top, "", scales], mode="nearest", nearest_mode="floor", coordinate_transformation_mode="asymmetric")
        top = g.op("Add", [conv(g, feats[i], chans[i], width, 1, bias=False), up])
        ins.append(top)
    outs = []
    for j, t in enumerate(ins):   # ins[0] is the coarsest level
        p = conv(g, t, width, outc, 3, bias=False)
        f = 1 << (levels - 1 - j)
        if f > 1:
            p = g.op("Resize", [p, "", g.init(np.array([1, 1, f, f], np.float32), "scales")], mode="nearest", nearest_mode="floor", coordinate_transformation_mode="asymmetric")
        outs.append(p)
    y = g.op("Concat", outs[::-1], axis=1)
    g.add_output(y, ["N", outc * levels, "H", "W"])
    return f"fpn {chans}->{width}->{outc} {n}x{h}x{w}", g.model(), (n, 8, h, w)


def case_pool():
    c = int(rng.choice([8, 16, 24, 64, 96, 192]))
    kind = str(rng.choice(["AveragePool", "MaxPool"]))
    kh, kw = int(rng.integers(1, 5)), int(rng.integers(1, 5))
    sh, sw = int(rng.integers(1, kh + 1)), int(rng.integers(1, kw + 1))
    ph, pw = int(rng.integers(0, kh // 2 + 1)), int(rng.integers(0, kw // 2 + 1))
    ceil_mode = int(rng.random() < 0.3)
    n, h, w = int(rng.integers(1, 5)), int(rng.integers(kh + 1, 80)), int(rng.integers(kw + 1, 120))
    g = GraphBuilder("f")
    g.add_input("x", ["N", 8, "H", "W"])
    attrs = dict(kernel_shape=[kh, kw], strides=[sh, sw], pads=[ph, pw, ph, pw], ceil_mode=ceil_mode)
    if kind == "AveragePool":
        attrs["count_include_pad"] = int(rng.random() < 0.5)
    y = g.op(kind, [stem(g, c)], **attrs)
    y = g.op("Relu", [conv(g, y, c, 16, 1)])
    g.add_output(y, ["N", 16, "H2", "W2"])
    return f"{kind} {kh}x{kw}/{sh}x{sw} pad {ph},{pw} ceil {ceil_mode} {attrs.get('count_include_pad', '-')} c{c} {n}x{h}x{w}", g.model(), (n, 8, h, w)


def case_eltwise():
    """binary operators with ONNX broadcasting: per-channel / per-sample / per-pixel / scalar / full second operands, constants and computed tensors"""
    c = int(rng.choice([3, 8, 16, 24, 50, 96]))
    n, h, w = int(rng.integers(1, 5)), int(rng.integers(1, 60)), int(rng.integers(1, 90))
    g = GraphBuilder("f")
    g.add_input("x", ["N", 8, "H", "W"])
    y = stem(g, c)
    desc = []
    for _ in range(int(rng.integers(1, 4))):
        op = str(rng.choice(["Add", "Mul", "Sub", "Div", "Max", "Min"]))
        form = str(rng.choice(["chan", "chan3", "scalar", "pix", "gap", "full", "row"]))
        if form == "chan":
            b = g.init((1.5 + 0.3 * rng.standard_normal((1, c, 1, 1))).astype(np.float32))
        elif form == "chan3":
            b = g.init((1.5 + 0.3 * rng.standard_normal((c, 1, 1))).astype(np.float32))
        elif form == "scalar":
            b = g.init(np.array(1.25, np.float32), "c")
        elif form == "pix":    # computed [N, 1, H, W]
            b = g.op("Add", [g.op("Sigmoid", [conv(g, y, c, 1, 1)]), g.init(np.array(1.0, np.float32), "c")])
        elif form == "gap":    # computed [N, C, 1, 1]
            b = g.op("Add", [g.op("Sigmoid", [g.op("GlobalAveragePool", [y])]), g.init(np.array(1.0, np.float32), "c")])
        elif form == "row":    # constant [W]-less: [1, 1, 1, 1] degenerate
            b = g.init(np.array([[[[2.0]]]], np.float32))
        else:
            b = g.op("Add", [g.op("Sigmoid", [conv(g, y, c, c, 1)]), g.init(np.array(1.0, np.float32), "c")])
        swap = bool(rng.random() < 0.3) and op != "Div"   # (constant / activation is a division by values at and near zero: ill-conditioned, not a statement about the operator)
        y = g.op(op, [b, y] if swap else [y, b])
        desc.append(f"{op}:{form}{'~' if swap else ''}")
    g.add_output(y, ["N", c, "H", "W"])
    return f"eltwise c{c} {n}x{h}x{w} {' '.join(desc)}", g.model(), (n, 8, h, w)


def case_reduce():
    """ReduceMean / Sum / Max / Min over random axes of 4-D and 3-D tensors, keepdims on and off; Softmax and LayerNormalization over odd widths"""
    g = GraphBuilder("f")
    if rng.random() < 0.5:
        c = int(rng.choice([4, 10, 24, 33]))
        n, h, w = int(rng.integers(1, 4)), int(rng.integers(1, 30)), int(rng.integers(1, 50))
        g.add_input("x", ["N", 8, "H", "W"])
        t = stem(g, c)
        axes = sorted(set(int(a) for a in rng.choice([1, 2, 3], size=int(rng.integers(1, 3)), replace=False)))
        kd = int(rng.random() < 0.6)
        op = str(rng.choice(["ReduceMean", "ReduceSum", "ReduceMax", "ReduceMin"]))
        y = g.op(op, [t], axes=axes, keepdims=kd)
        g.add_output(y, ["A", "B", "C", "D"][: 4 if kd else 4 - len(axes)])
        return f"{op} axes {axes} keepdims {kd} c{c} {n}x{h}x{w}", g.model(), (n, 8, h, w)
    dim = int(rng.choice([7, 24, 33, 96, 120, 257]))
    n, T = int(rng.integers(1, 5)), int(rng.integers(1, 70))
    g.add_input("x", ["N", "T", dim])
    t = _linear(g, "x", dim, dim)
    which = str(rng.choice(["ln", "softmax_last", "softmax_mid", "reduce_last", "reduce_mid"]))
    if which == "ln":
        y = _ln(g, t, dim)
    elif which == "softmax_last":
        y = g.op("Softmax", [t], axis=-1)
    elif which == "softmax_mid":
        y = g.op("Softmax", [t], axis=1)
    elif which == "reduce_last":
        y = g.op("ReduceMean", [t], axes=[-1], keepdims=int(rng.random() < 0.5))
    else:
        y = g.op("ReduceMax", [t], axes=[1], keepdims=int(rng.random() < 0.5))
    g.add_output(y, ["A", "B", "C"])
    return f"seq {which} dim {dim} {n}x{T}", g.model(), (n, T, dim)


def case_resize():
    """Resize: nearest (asymmetric / floor, half_pixel / round_prefer_floor) and linear (half_pixel, align_corners, asymmetric, pytorch_half_pixel), scales and sizes,
    integer and fractional factors, enlargements and reductions"""
    c = int(rng.choice([3, 8, 16, 40]))
    n, h, w = int(rng.integers(1, 4)), int(rng.integers(2, 50)), int(rng.integers(2, 70))
    mode = str(rng.choice(["nearest", "linear"]))
    if mode == "nearest":
        ctm, nm = [("asymmetric", "floor"), ("half_pixel", "round_prefer_floor"), ("asymmetric", "round_prefer_floor")][int(rng.integers(0, 3))]
    else:
        ctm, nm = str(rng.choice(["half_pixel", "align_corners", "asymmetric", "pytorch_half_pixel"])), None
    fy, fx = [float(rng.choice([0.5, 1.0, 1.5, 2.0, 2.5, 3.0, 4.0])) for _ in range(2)]
    use_sizes = bool(rng.random() < 0.4)
    g = GraphBuilder("f")
    g.add_input("x", ["N", 8, "H", "W"])
    t = stem(g, c) if rng.random() < 0.6 else "x"
    cc = c if t != "x" else 8
    attrs = dict(mode=mode, coordinate_transformation_mode=ctm)
    if nm:
        attrs["nearest_mode"] = nm
    if use_sizes:
        oh, ow = max(1, int(h * fy)), max(1, int(w * fx))
        y = g.op("Resize", [t, "", "", g.init(np.array([n, cc, oh, ow], np.int64), "sizes")], **attrs)
    else:
        y = g.op("Resize", [t, "", g.init(np.array([1, 1, fy, fx], np.float32), "scales")], **attrs)
    y = g.op("Relu", [conv(g, y, cc, 8, 1)])
    g.add_output(y, ["N", 8, "H2", "W2"])
    return f"resize {mode} {ctm} {nm} x{fy},{fx} {'sizes' if use_sizes else 'scales'} c{cc} {n}x{h}x{w}", g.model(), (n, 8, h, w)


def case_shape():
    """layout plumbing the exporters emit around the recognizer head and the necks: Slice / Split / Concat on several axes, Transpose, Squeeze / Unsqueeze, Reshape with 0 / -1, Pad"""
    c = int(rng.choice([8, 16, 24, 48]))
    n, h, w = int(rng.integers(1, 4)), int(rng.integers(2, 20)), int(rng.integers(4, 60))
    g = GraphBuilder("f")
    g.add_input("x", ["N", 8, "H", "W"])
    t = stem(g, c)
    which = str(rng.choice(["slice_concat", "split_c", "rec_head", "pad", "transpose"]))
    if which == "slice_concat":
        a0, a1 = int(rng.integers(0, w // 2)), int(rng.integers(w // 2 + 1, w + 1))
        s1 = g.op("Slice", [t, g.init(np.array([a0], np.int64), "i"), g.init(np.array([a1], np.int64), "i"), g.init(np.array([3], np.int64), "i")])
        s2 = g.op("Slice", [t, g.init(np.array([0], np.int64), "i"), g.init(np.array([c // 2], np.int64), "i"), g.init(np.array([1], np.int64), "i")])
        y1 = g.op("Concat", [s1, s1], axis=3)
        y2 = g.op("Concat", [s2, t], axis=1)
        g.add_output(y1, ["A", "B", "C", "D"])
        y = y2
    elif which == "split_c":
        a, b = g.op("Split", [t, g.init(np.array([c // 2, c - c // 2], np.int64), "split")], n_out=2, axis=1)
        y = g.op("Concat", [g.op("Relu", [b]), g.op("Sigmoid", [a])], axis=1)
    elif which == "rec_head":
        p = g.op("AveragePool", [t], kernel_shape=[h, 1], strides=[h, 1], pads=[0, 0, 0, 0])      # [N, C, 1, W]
        sq = g.op("Squeeze", [p, g.init(np.array([2], np.int64), "axes")])                           # [N, C, W]
        tr = g.op("Transpose", [sq], perm=[0, 2, 1])                                               # [N, W, C]
        y = g.op("Softmax", [_linear(g, tr, c, 37)], axis=2)
    elif which == "pad":
        pads = [0, 0, int(rng.integers(0, 3)), int(rng.integers(0, 3)), 0, 0, int(rng.integers(0, 3)), int(rng.integers(0, 3))]
        y = g.op("Pad", [t, g.init(np.array(pads, np.int64), "pads"), g.init(np.array(0.5, np.float32), "c")], mode="constant")
        y = g.op("Relu", [conv(g, y, c, 8, 3)])
    else:
        perm = [[0, 2, 3, 1], [0, 3, 1, 2], [0, 1, 3, 2], [1, 0, 2, 3]][int(rng.integers(0, 4))]
        y = g.op("Transpose", [t], perm=perm)
        y = g.op("Reshape", [y, g.init(np.array([0, -1], np.int64), "shape")])
    g.add_output(y, ["A", "B", "C", "D"])
    return f"shape {which} c{c} {n}x{h}x{w}", g.model(), (n, 8, h, w)


def case_convmisc():
    """convolutions off the beaten path: rectangular kernels, dilation, asymmetric padding, mixed strides, group counts between 1 and Cin"""
    cin = int(rng.choice([4, 8, 16, 24, 32, 64]))
    groups = int(rng.choice([g_ for g_ in (1, 1, 2, 4, cin) if cin % g_ == 0]))
    cout = groups * int(rng.choice([1, 2, 3, 5, 8])) if groups == cin else groups * int(rng.choice([2, 4, 6, 12]))
    kh, kw = int(rng.choice([1, 2, 3, 5, 7])), int(rng.choice([1, 2, 3, 5, 7]))
    sh, sw = int(rng.choice([1, 1, 2])), int(rng.choice([1, 1, 2]))
    dh, dw = int(rng.choice([1, 1, 2])), int(rng.choice([1, 1, 2]))
    pads = [int(rng.integers(0, kh)), int(rng.integers(0, kw)), int(rng.integers(0, kh)), int(rng.integers(0, kw))]
    n, h, w = int(rng.integers(1, 4)), int(rng.integers(dh * (kh - 1) + 1, 60)), int(rng.integers(dw * (kw - 1) + 1, 80))
    a = str(rng.choice(["none", "relu", "hswish"]))
    g = GraphBuilder("f")
    g.add_input("x", ["N", 8, "H", "W"])
    t = stem(g, cin)
    wt = (rng.standard_normal((cout, cin // groups, kh, kw)) * np.sqrt(1.0 / (kh * kw * cin // groups))).astype(np.float32)
    y = g.op("Conv", [t, g.init(wt), g.init((0.2 * rng.standard_normal(cout)).astype(np.float32))], kernel_shape=[kh, kw], strides=[sh, sw], pads=pads, group=groups, dilations=[dh, dw])
    y = act(g, y, a)
    g.add_output(y, ["N", cout, "H2", "W2"])
    return f"conv {kh}x{kw}/{sh}x{sw} d{dh}x{dw} pads {pads} g{groups} {cin}->{cout} {n}x{h}x{w} {a}", g.model(), (n, 8, h, w)


def case_convt():
    cin = int(rng.choice([4, 8, 16, 32]))
    groups = int(rng.choice([1, 1, 2]))
    cout = groups * int(rng.choice([1, 2, 4, 8]))
    k, st = int(rng.choice([2, 3, 4])), int(rng.choice([1, 2, 2]))
    p = int(rng.integers(0, (k + 1) // 2))
    op_ = int(rng.integers(0, st))
    n, h, w = int(rng.integers(1, 4)), int(rng.integers(2, 40)), int(rng.integers(2, 50))
    g = GraphBuilder("f")
    g.add_input("x", ["N", 8, "H", "W"])
    wt = (rng.standard_normal((cin, cout // groups, k, k)) * np.sqrt(1.0 / cin)).astype(np.float32)
    y = g.op("ConvTranspose", [stem(g, cin), g.init(wt), g.init((0.1 * rng.standard_normal(cout)).astype(np.float32))], kernel_shape=[k, k], strides=[st, st], pads=[p] * 4,
             output_padding=[op_, op_], group=groups, dilations=[1, 1])
    y = g.op("Relu", [y])
    g.add_output(y, ["N", cout, "H2", "W2"])
    return f"convT k{k} s{st} p{p} op{op_} g{groups} {cin}->{cout} {n}x{h}x{w}", g.model(), (n, 8, h, w)


def case_gemm():
    which = str(rng.choice(["gemm", "matmul3", "bmm", "classifier"]))
    g = GraphBuilder("f")
    if which == "classifier":   # PP-LCNet tail: GAP -> 1x1 conv + hard-swish -> Flatten -> Gemm -> Softmax
        c, hid, ncls = int(rng.choice([16, 64, 160])), int(rng.choice([32, 128, 320])), int(rng.choice([2, 4, 10]))
        n, h, w = int(rng.integers(1, 9)), int(rng.integers(1, 30)), int(rng.integers(1, 40))
        g.add_input("x", ["N", 8, "H", "W"])
        t = g.op("HardSwish", [conv(g, g.op("GlobalAveragePool", [stem(g, c)]), c, hid, 1)])
        f = g.op("Flatten", [t], axis=1)
        wq = (rng.standard_normal((ncls, hid)) / np.sqrt(hid)).astype(np.float32)
        y = g.op("Softmax", [g.op("Gemm", [f, g.init(wq), g.init((0.1 * rng.standard_normal(ncls)).astype(np.float32))], transB=1)], axis=1)
        g.add_output(y, ["N", ncls])
        return f"classifier {c}->{hid}->{ncls} {n}x{h}x{w}", g.model(), (n, 8, h, w)
    K, M = int(rng.choice([7, 24, 64, 120, 256, 384])), int(rng.choice([5, 16, 37, 96, 257]))
    n, T = int(rng.integers(1, 6)), int(rng.integers(1, 90))
    g.add_input("x", ["N", "T", K])
    if which == "gemm":
        f = g.op("Reshape", ["x", g.init(np.array([-1, K], np.int64), "shape")])
        tb = int(rng.random() < 0.5)
        wq = (rng.standard_normal((M, K) if tb else (K, M)) / np.sqrt(K)).astype(np.float32)
        al, be = float(rng.choice([1.0, 0.5])), float(rng.choice([1.0, 0.0, 2.0]))
        y = g.op("Gemm", [f, g.init(wq), g.init((0.1 * rng.standard_normal(M)).astype(np.float32))], transB=tb, alpha=al, beta=be)
        g.add_output(y, ["R", M])
        return f"Gemm K{K} M{M} transB {tb} alpha {al} beta {be} rows {n * T}", g.model(), (n, T, K)
    if which == "matmul3":
        y = act(g, _linear(g, "x", K, M, bias=bool(rng.random() < 0.7)), str(rng.choice(["none", "relu", "gelu", "hswish"])))
        g.add_output(y, ["N", "T", M])
        return f"MatMul [N,T,{K}] x [{K},{M}] {n}x{T}", g.model(), (n, T, K)
    a = _linear(g, "x", K, M)                                        # [N, T, M]
    b = g.op("Transpose", [_linear(g, "x", K, M)], perm=[0, 2, 1])   # [N, M, T]
    y = g.op("MatMul", [a, b])                                       # [N, T, T]
    z = g.op("MatMul", [g.op("Softmax", [y], axis=-1), a])           # [N, T, M]
    g.add_output(z, ["N", "T", M])
    return f"batched MatMul T{T} M{M} K{K} n{n}", g.model(), (n, T, K)


def case_gridsample():
    c = int(rng.choice([3, 8, 16]))
    n, h, w = int(rng.integers(1, 4)), int(rng.integers(2, 40)), int(rng.integers(2, 50))
    mode, padm, ac = str(rng.choice(["bilinear", "nearest"])), str(rng.choice(["zeros", "border"])), int(rng.random() < 0.5)
    g = GraphBuilder("f")
    g.add_input("x", ["N", 8, "H", "W"])
    t = stem(g, c)
    gr = g.op("Mul", [g.op("Tanh", [conv(g, "x", 8, 2, 3)]), g.init(np.array(1.2, np.float32), "c")])   # [N, 2, H, W] in (-1.2, 1.2): some samples fall outside
    gr = g.op("Transpose", [gr], perm=[0, 2, 3, 1])
    y = g.op("GridSample", [t, gr], mode=mode, padding_mode=padm, align_corners=ac)
    g.add_output(y, ["N", c, "H", "W"])
    return f"GridSample {mode} {padm} align {ac} c{c} {n}x{h}x{w}", g.model(), (n, 8, h, w)


def case_net():
    """a random small network: a DAG of 8 - 20 operators over tensors of several resolutions -- separable / dense / strided convolutions, squeeze-excite, residual sums,
    nearest x2 + sum (FPN), concatenations of several levels, pooling, transposed convolution -- with two or three graph outputs taken anywhere.  What it is after is the
    planner: arena reuse across branches, deferred Resize / Concat / ConvTranspose operands meeting consumers that cannot absorb them, channels-last <-> native moves."""
    n = int(rng.integers(1, 4))
    lv0 = int(rng.integers(2, 4))                         # the input is 2^lv0 x the coarsest grid
    big = 6 if os.environ.get("OP_FUZZ_BIG") == "1" else 1     # OP_FUZZ_BIG=1: maps large enough for the weight-stationary / bf16x6 kernels and the deferred concat reads
    gh, gw = big * int(rng.integers(1, 6)), big * int(rng.integers(1, 8))
    h, w = gh << lv0, gw << lv0
    g = GraphBuilder("f")
    g.add_input("x", ["N", 8, "H", "W"])
    pool = [("x", 8, 0)]                                   # (name, channels, level: resolution = input >> level)
    t = stem(g, int(rng.choice([16, 24, 32])))
    pool.append((t, int(g_last_c[0]), 0))
    desc = []
    for _ in range(int(rng.integers(8, 21))):
        name, c, lv = pool[int(rng.integers(1, len(pool)))]
        op = str(rng.choice(["ds", "ds", "c3", "c1", "down", "se", "res", "up_add", "cat", "pool", "convt", "gap_mul"]))
        if op == "ds":
            co = int(rng.choice([c, 2 * c if c <= 96 else c, 24, 48]))
            y = _ds(g, name, c, co, int(rng.choice([3, 5])), (1, 1), "hswish", bool(rng.random() < 0.3)); pool.append((y, co, lv))
        elif op == "c3":
            co = int(rng.choice([16, 32, 64]))
            y = act(g, conv(g, name, c, co, 3), str(rng.choice(["relu", "hswish", "none"]))); pool.append((y, co, lv))
        elif op == "c1":
            co = int(rng.choice([8, 24, 96, 128]))
            y = act(g, conv(g, name, c, co, 1), str(rng.choice(["relu", "none", "gelu"]))); pool.append((y, co, lv))
        elif op == "down" and lv < lv0:
            co = int(rng.choice([c, 2 * c if c <= 64 else c]))
            y = g.op("Relu", [conv(g, name, c, co, 3, stride=2)]); pool.append((y, co, lv + 1))
        elif op == "se":
            p = g.op("GlobalAveragePool", [name])
            f = g.op("HardSigmoid", [conv(g, g.op("Relu", [conv(g, p, c, max(4, c // 4), 1)]), max(4, c // 4), c, 1)], alpha=1.0 / 6.0, beta=0.5)
            pool.append((g.op("Mul", [name, f]), c, lv))
        elif op == "res":
            same = [q for q in pool[1:] if q[1] == c and q[2] == lv and q[0] != name]
            if same:
                pool.append((g.op("Add", [name, same[int(rng.integers(0, len(same)))][0]]), c, lv))
        elif op == "up_add" and lv > 0:
            finer = [q for q in pool[1:] if q[2] == lv - 1]
            if finer:
                fq = finer[int(rng.integers(0, len(finer)))]
                up = g.op("Resize", [name, "", g.init(np.array([1, 1, 2, 2], np.float32), "scales")], mode="nearest", nearest_mode="floor", coordinate_transformation_mode="asymmetric")
                lat = conv(g, fq[0], fq[1], c, 1, bias=False)
                pool.append((g.op("Add", [lat, up]) if rng.random() < 0.7 else g.op("Add", [up, lat]), c, lv - 1))
        elif op == "cat":
            parts = []
            for q in pool[1:]:
                if q[2] >= lv and len(parts) < 4 and rng.random() < 0.5:
                    pq = q[0]
                    if q[2] > lv:
                        f = 1 << (q[2] - lv)
                        pq = g.op("Resize", [pq, "", g.init(np.array([1, 1, f, f], np.float32), "scales")], mode="nearest", nearest_mode="floor", coordinate_transformation_mode="asymmetric")
                    parts.append((pq, q[1]))
            if len(parts) >= 2:
                pool.append((g.op("Concat", [p_[0] for p_ in parts], axis=1), sum(p_[1] for p_ in parts), lv))
        elif op == "pool" and lv < lv0:
            kind = str(rng.choice(["AveragePool", "MaxPool"]))
            pool.append((g.op(kind, [name], kernel_shape=[2, 2], strides=[2, 2], pads=[0, 0, 0, 0]), c, lv + 1))
        elif op == "convt" and lv > 0 and c % 4 == 0:
            co = int(rng.choice([8, 16, c]))
            wt = (rng.standard_normal((c, co, 2, 2)) * np.sqrt(1.0 / c)).astype(np.float32)
            y = g.op("Relu", [g.op("ConvTranspose", [name, g.init(wt), g.init((0.1 * rng.standard_normal(co)).astype(np.float32))], kernel_shape=[2, 2], strides=[2, 2], pads=[0, 0, 0, 0], group=1, dilations=[1, 1])])
            pool.append((y, co, lv - 1))
        elif op == "gap_mul":
            pool.append((g.op("Mul", [name, g.op("Sigmoid", [g.op("GlobalAveragePool", [name])])]), c, lv))
        else:
            continue
        desc.append(op)
    outs = {len(pool) - 1}
    for _ in range(int(rng.integers(1, 3))):
        outs.add(int(rng.integers(1, len(pool))))
    for i in sorted(outs):
        g.add_output(pool[i][0], ["N", pool[i][1], "Ho", "Wo"])
    return f"net {n}x{h}x{w} {' '.join(desc)} outs {sorted(outs)}", g.model(), (n, 8, h, w)


def _swish(g, y):
    return g.op("Mul", [y, g.op("Sigmoid", [y])])


def _conv1k(g, x, cin, cout, k):
    w = (rng.standard_normal((cout, cin, 1, k)) * np.sqrt(1.0 / (k * cin))).astype(np.float32)
    return g.op("Conv", [x, g.init(w), g.init((0.1 * rng.standard_normal(cout)).astype(np.float32))], kernel_shape=[1, k], strides=[1, 1], pads=[0, k // 2, 0, k // 2], group=1, dilations=[1, 1])


def case_neck():
    """the recognizer's sample-local tail (EncoderWithSVTR) at random widths: [6, 2] pool -> 1x3 conv -> 1x1 -> tokens -> 1..3 pre-LN blocks -> LN -> 1x1 -> concat with the
    pooled map -> 1x3 -> 1x1 -> tokens -> CTC Linear (+ Softmax): the operator run the engine fuses into one chain launch per batch (LDS placement, split-K, attention tiles)"""
    c = int(rng.choice([64, 96, 128, 192, 256, 320, 512]))
    heads = int(rng.choice([1, 2, 4, 8]))
    hd = int(rng.choice([8, 15, 16, 32]))
    dim = heads * hd
    outc = int(rng.choice([32, 64, 96, 120]))
    vocab = int(rng.choice([37, 97, 500, 6625]))
    n, T = int(rng.integers(1, 70)), int(rng.choice([int(rng.integers(1, 41)), int(rng.integers(41, 81)), int(rng.integers(81, 200))]))
    if n * T > 4000:
        n = max(1, 4000 // T)
    hrows = int(rng.choice([1, 2, 6]))
    g = GraphBuilder("f")
    g.add_input("x", ["N", 8, "H", "W"])
    t = stem(g, c)
    x = g.op("AveragePool", [t], kernel_shape=[hrows, 2], strides=[hrows, 2], pads=[0, 0, 0, 0]) if hrows > 1 or rng.random() < 0.5 else t
    pooled_w = 2 if x != t else 1
    hmap = x
    z = _swish(g, _conv1k(g, x, c, max(8, c // 8), 3))
    z = _swish(g, conv(g, z, max(8, c // 8), dim, 1))
    z = g.op("Transpose", [g.op("Squeeze", [z, g.init(np.array([2], np.int64), "axes")])], perm=[0, 2, 1])
    for _ in range(int(rng.integers(1, 4))):
        z = g.op("Add", [z, _mha(g, _ln(g, z, dim), dim, heads)])
        y = _linear(g, _swish(g, _linear(g, _ln(g, z, dim), dim, 2 * dim)), 2 * dim, dim)
        z = g.op("Add", [z, y])
    z = _ln(g, z, dim)
    z = g.op("Unsqueeze", [g.op("Transpose", [z], perm=[0, 2, 1]), g.init(np.array([2], np.int64), "axes")])
    z = _swish(g, conv(g, z, dim, c, 1))
    z = g.op("Concat", [hmap, z], axis=1)
    z = _swish(g, _conv1k(g, z, 2 * c, max(8, c // 8), 3))
    z = _swish(g, conv(g, z, max(8, c // 8), outc, 1))
    z = g.op("Transpose", [g.op("Squeeze", [z, g.init(np.array([2], np.int64), "axes")])], perm=[0, 2, 1])
    y = g.op("Softmax", [_linear(g, z, outc, vocab)], axis=2)
    g.add_output(y, ["N", "T", vocab])
    return f"neck c{c} {heads}x{hd} out{outc} V{vocab} pool[{hrows},{pooled_w}] {n}x{T}", g.model(), (n, 8, hrows, T * pooled_w)


def case_p2o():
    """shape arithmetic as the Paddle / PyTorch exporters write it: Shape -> Slice / Gather -> Concat -> Reshape / Resize(sizes) / Expand, activations and normalisations that do
    not fold (Clip, PRelu, LeakyRelu, BatchNormalization behind an Add), ArgMax tails"""
    c = int(rng.choice([8, 16, 24, 48]))
    n, h, w = int(rng.integers(1, 4)), int(rng.integers(2, 24)), int(rng.integers(2, 40))
    which = str(rng.choice(["tokens", "resize_to_ref", "flatten_math", "expand_pos", "acts", "bn_after_add", "argmax"]))
    i64 = np.int64
    g = GraphBuilder("f")
    g.add_input("x", ["N", 8, "H", "W"])
    t = stem(g, c)
    ci = lambda v: g.init(np.array(v, i64), "i")
    if which == "tokens":      # [N, C, H, W] -> [N, H*W, C] with the batch read from Shape
        shp = g.op("Shape", [t])
        nb = g.op("Slice", [shp, ci([0]), ci([1]), ci([0])])
        tgt = g.op("Concat", [nb, ci([c]), ci([-1])], axis=0)
        y = g.op("Transpose", [g.op("Reshape", [t, tgt])], perm=[0, 2, 1])
        y = _ln(g, y, c)
        g.add_output(y, ["N", "T", c])
    elif which == "resize_to_ref":   # coarse map resized to the spatial size of a finer one, sizes = Concat(Shape(coarse)[:2], Shape(ref)[2:])
        coarse = g.op("Relu", [conv(g, t, c, c, 3, stride=2)])
        sizes = g.op("Concat", [g.op("Slice", [g.op("Shape", [coarse]), ci([0]), ci([2]), ci([0])]), g.op("Slice", [g.op("Shape", [t]), ci([2]), ci([4]), ci([0])])], axis=0)
        mode = str(rng.choice(["nearest", "linear"]))
        attrs = dict(mode=mode, coordinate_transformation_mode="asymmetric" if mode == "nearest" else "half_pixel")
        if mode == "nearest":
            attrs["nearest_mode"] = "floor"
        up = g.op("Resize", [coarse, "", "", sizes], **attrs)
        y = g.op("Add", [up, t])
        g.add_output(y, ["N", c, "H", "W"])
    elif which == "flatten_math":    # [N, C, H, W] -> [N, C*H*W] through Gather + Mul on shape scalars, then a Linear on a fixed-size map
        hh, ww = 3, 5
        p = g.op("Resize", [t, "", "", ci([n, c, hh, ww])], mode="nearest", coordinate_transformation_mode="asymmetric", nearest_mode="floor")
        shp = g.op("Shape", [p])
        d1 = g.op("Gather", [shp, ci(1)], axis=0)
        d2 = g.op("Gather", [shp, ci(2)], axis=0)
        d3 = g.op("Gather", [shp, ci(3)], axis=0)
        tot = g.op("Unsqueeze", [g.op("Mul", [g.op("Mul", [d1, d2]), d3]), ci([0])])
        y = g.op("Reshape", [p, g.op("Concat", [ci([-1]), tot], axis=0)])
        y = _linear(g, y, c * hh * ww, 10)
        g.add_output(y, ["N", 10])
    elif which == "expand_pos":      # a learned [1, C, 1, W0] embedding resized to W, expanded over the batch and the rows, added
        pos = g.init((0.3 * rng.standard_normal((1, c, 1, 1))).astype(np.float32))
        shp = g.op("Shape", [t])
        y = g.op("Add", [t, g.op("Expand", [pos, shp])])
        y = g.op("Mul", [y, g.op("Tile", [g.init((1.0 + 0.1 * rng.standard_normal((1, c, 1, 1))).astype(np.float32)), ci([1, 1, 1, 1])])])
        g.add_output(y, ["N", c, "H", "W"])
    elif which == "acts":
        y = g.op("Clip", [conv(g, t, c, c, 3), g.init(np.array(-0.5, np.float32), "c"), g.init(np.array(1.5, np.float32), "c")])
        y = g.op("PRelu", [conv(g, y, c, c, 1), g.init((0.25 + 0.05 * rng.standard_normal((c, 1, 1))).astype(np.float32))])
        y = g.op("LeakyRelu", [conv(g, y, c, 16, 3)], alpha=0.1)
        y = g.op("Tanh", [y])
        g.add_output(y, ["N", 16, "H", "W"])
    elif which == "bn_after_add":
        y = g.op("Add", [conv(g, t, c, c, 3), t])
        y = g.op("BatchNormalization", [y, g.init((1.0 + 0.1 * rng.standard_normal(c)).astype(np.float32)), g.init((0.1 * rng.standard_normal(c)).astype(np.float32)),
                                        g.init((0.1 * rng.standard_normal(c)).astype(np.float32)), g.init((1.0 + 0.2 * rng.random(c)).astype(np.float32))], epsilon=1e-5)
        y = g.op("Relu", [y])
        g.add_output(y, ["N", c, "H", "W"])
    else:                              # class map: ArgMax over channels is not the last axis of an NCHW tensor -> Transpose first, as exporters do
        y = g.op("Transpose", [conv(g, t, c, 5, 1)], perm=[0, 2, 3, 1])
        y = g.op("Softmax", [y], axis=3)
        g.add_output(y, ["N", "H", "W", 5])
    return f"p2o {which} c{c} {n}x{h}x{w}", g.model(), (n, 8, h, w)


KINDS = {"p2o": case_p2o, "neck": case_neck, "net": case_net, "convmisc": case_convmisc, "convt": case_convt, "gemm": case_gemm, "gridsample": case_gridsample, "eltwise": case_eltwise, "reduce": case_reduce, "resize": case_resize, "shape": case_shape, "svtr_block": case_svtr_block, "dbhead": case_dbhead, "fpn": case_fpn, "pool": case_pool, "dsblock": case_dsblock, "dschain": case_dschain, "conv1x1": case_conv1x1, "conv3x3": case_conv3x3, "convk": case_convk, "grouped": case_grouped, "concat": case_concat, "attention": case_attention}
names = list(KINDS) if only == "all" else [only]
bad = 0
worst = {}
seen = {}
t0 = time.time()
for i in range(n_cases):
    kind = names[i % len(names)]
    label, model, shape = KINDS[kind]()
    x = rng.standard_normal(shape).astype(np.float32)
    if DRY:   # (no GPU: the graphs only -- build, parse, evaluate on the CPU)
        r = onnx_ref.run(model, {"x": x})
        print("dry", label, [a.shape for a in r], flush=True)
        continue
    try:
        eng = api.OrtInfer(model, profile=True)
        if RESHAPE and kind in RESHAPE_KINDS:
            # the same engine on a second input shape, then the first again: plans, constants laid out per shape and arena offsets must not leak from one shape to the other
            shape2 = (shape[0] + 1, 8, shape[2], 2 * shape[3]) if len(shape) == 4 else (shape[0] + 1, 2 * shape[1], shape[2])
            x2 = rng.standard_normal(shape2).astype(np.float32)
            first = eng.infer(x)
            got2 = eng.infer(x2)
            ref2 = onnx_ref.run(model, {eng.input_name(): x2})
            for (_, a), r in zip(got2, ref2):
                assert a.shape == r.shape, (a.shape, r.shape)
                e2 = float(np.abs(a - r).max()) / max(1.0, float(np.abs(r).max())) if r.size else 0.0
                assert e2 <= TOL, f"second shape {shape2}: {e2:.3e}"
            again = eng.infer(x)
            for (_, a), (_, b) in zip(first, again):
                assert np.array_equal(a, b), "first shape again: not the bytes of its first run"
        api.prof_enable(True); api.prof_reset()
        got = eng.infer(x)
        classes = sorted(e["name"] for e in api.prof_snapshot() if e["launches"] > 0)
        api.prof_enable(False)
        ref = onnx_ref.run(model, {eng.input_name(): x})
        err = 0.0
        for (_, a), r in zip(got, ref):
            assert a.shape == r.shape, (a.shape, r.shape)
            if r.size:
                fin = np.isfinite(r)
                if not fin.all():   # a division by an exact zero: the engine must be non-finite in the same places (inf of the same sign, nan where nan)
                    same = np.array_equal(np.isnan(a), np.isnan(r)) and np.array_equal(a[np.isinf(r)], r[np.isinf(r)]) and np.isfinite(a[fin]).all()
                    err = max(err, 0.0 if same else 1.0)
                if fin.any():
                    err = max(err, float(np.abs(a[fin] - r[fin]).max()) / max(1.0, float(np.abs(r[fin]).max())))
        ok = err <= TOL
        eng.close() if hasattr(eng, "close") else None
    except Exception as e:   # an engine error on a supported graph is a failure of the sweep, not of the harness
        ok, err, classes = False, float("nan"), [f"EXC {type(e).__name__}: {str(e)[:200]}"]
    for c in classes:
        seen[c] = seen.get(c, 0) + 1
    worst[kind] = max(worst.get(kind, 0.0), err if err == err else 1.0)
    if os.environ.get("OP_FUZZ_VERBOSE") == "1":
        print(f"{'ok  ' if ok else 'BAD '} case {i} [{label}] err {err:.2e} {classes}", flush=True)
    if not ok:
        bad += 1
        print(f"FAIL case {i} [{label}] err {err:.3e} classes {classes}", flush=True)
print(f"{n_cases - bad}/{n_cases} cases within {TOL} of the oracle in {time.time() - t0:.0f} s (seed {seed}); worst relative difference per kind: "
      + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))
print("kernel classes exercised: " + ", ".join(f"{k} x{v}" for k, v in sorted(seen.items())))
sys.exit(1 if bad else 0)
