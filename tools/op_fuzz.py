"""Randomised single-layer sweep of the engine's convolution / attention dispatch against the torch-CPU oracle (round 6): random shapes around every
eligibility boundary of the bf16x6 kernels (weight-stationary one- and two-fragment tiles, output-stationary, row-streaming 3x3, LDS-tiled large kernel,
grouped, multi-source concat reads, streaming attention), random epilogues (bias, ReLU / hard-swish / GELU / none, residual).  Reports, per case, the kernel
classes that ran and the largest |difference| relative to the output scale; the tolerance is the engine tests' 2e-4.
usage: python tools/op_fuzz.py [n_cases] [seed] [kind]      kind: dsblock | dschain | conv1x1 | conv3x3 | convk | grouped | concat | attention | all"""
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


def stem(g, c):
    """a channels-last producer in front of the layer under test (graph inputs are NCHW)"""
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


KINDS = {"dsblock": case_dsblock, "dschain": case_dschain, "conv1x1": case_conv1x1, "conv3x3": case_conv3x3, "convk": case_convk, "grouped": case_grouped, "concat": case_concat, "attention": case_attention}
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
        api.prof_enable(True); api.prof_reset()
        got = eng.infer(x)
        classes = sorted(e["name"] for e in api.prof_snapshot() if e["launches"] > 0)
        api.prof_enable(False)
        ref = onnx_ref.run(model, {eng.input_name(): x})
        err = 0.0
        for (_, a), r in zip(got, ref):
            assert a.shape == r.shape, (a.shape, r.shape)
            if r.size:
                err = max(err, float(np.abs(a - r).max()) / max(1.0, float(np.abs(r).max())))
        ok = err <= TOL and all(np.isfinite(a).all() for _, a in got)
        eng.close() if hasattr(eng, "close") else None
    except Exception as e:   # an engine error on a supported graph is a failure of the sweep, not of the harness
        ok, err, classes = False, float("nan"), [f"EXC {type(e).__name__}: {str(e)[:200]}"]
    for c in classes:
        seen[c] = seen.get(c, 0) + 1
    worst[kind] = max(worst.get(kind, 0.0), err if err == err else 1.0)
    if not ok:
        bad += 1
        print(f"FAIL case {i} [{label}] err {err:.3e} classes {classes}", flush=True)
print(f"{n_cases - bad}/{n_cases} cases within {TOL} of the oracle in {time.time() - t0:.0f} s (seed {seed}); worst relative difference per kind: "
      + ", ".join(f"{k} {v:.2e}" for k, v in worst.items()))
print("kernel classes exercised: " + ", ".join(f"{k} x{v}" for k, v in sorted(seen.items())))
sys.exit(1 if bad else 0)
