"""Synthetic PP-OCR-style ONNX graphs (random-init weights of the PP-OCR architecture family).

The reference treats every model as an opaque `.onnx` with one f32 NCHW input named "x" and uses
output[0] only (oar-ocr-core/src/models/detection/db.rs:388-390, recognition/crnn.rs:273-279).  The
real files are not in this container, so bench/tests use graphs of the same
topology family, sized to the byte sizes pinned in core/download/registry.rs:83-84
(pp-ocrv6_tiny_det 1.78 MB ~ 0.44 M params; pp-ocrv6_tiny_rec 4.46 MB ~ 1.1 M params, V = 6906).

Detector: PP-LCNetV3-style depthwise-separable backbone (strides 4/8/16/32) -> RSE-FPN-style neck
(1x1 laterals, nearest x2 top-down adds, 3x3 smooth, concat at stride 4) -> DB head (3x3 conv, BN,
ReLU, 2x ConvTranspose 2x2 s2, Sigmoid).  Recognizer: LCNetV3-style backbone with (2,1)/(1,2)
strides (H 48 -> 1, W -> W/8), SVTR neck (2 global-attention blocks) and CTC Linear+Softmax head.

Random weights give a meaningless probability map, so the detector carries one hand-set "ink" path:
channel 0 of every layer on the stride-4 route propagates a blurred darkness signal (all other
channels are random) and the last ConvTranspose reads it with a large gain.  The network therefore
genuinely computes text-like blobs from a synthetic page -- nothing is injected after the forward.
"""
from __future__ import annotations

import numpy as np

from .onnx_writer import GraphBuilder, node

DB_MEAN = np.array([0.485, 0.456, 0.406], np.float32)
DB_STD = np.array([0.229, 0.224, 0.225], np.float32)
INK_AMPL = 12.0  # channel-0 amplitude for a fully dark neighbourhood


class _Net:
    def __init__(self, name, seed, opset=17, decomposed_hswish=True):
        self.g = GraphBuilder(name, opset)
        self.rng = np.random.default_rng(seed)
        self.decomposed_hswish = decomposed_hswish

    # ------------------------------------------------------------------ weights
    def _w(self, shape, fan_in, gain=2.0):
        return (self.rng.standard_normal(shape) * np.sqrt(gain / fan_in)).astype(np.float32)

    def _b(self, n, scale=0.05):
        return (self.rng.standard_normal(n) * scale).astype(np.float32)

    # ------------------------------------------------------------------ activations
    def act(self, x, kind):
        g = self.g
        if kind is None:
            return x
        if kind == "relu":
            return g.op("Relu", [x])
        if kind == "hswish":
            if self.decomposed_hswish:
                hs = g.op("HardSigmoid", [x], alpha=1.0 / 6.0, beta=0.5)
                return g.op("Mul", [x, hs])
            return g.op("HardSwish", [x])
        if kind == "hsigmoid":
            return g.op("HardSigmoid", [x], alpha=0.2, beta=0.5)
        if kind == "sigmoid":
            return g.op("Sigmoid", [x])
        if kind == "swish":
            s = g.op("Sigmoid", [x])
            return g.op("Mul", [x, s])
        if kind == "gelu":   # the erf form exporters below opset 20 write: x * 0.5 * (1 + erf(x / sqrt(2)))
            e = g.op("Erf", [g.op("Div", [x, g.init(np.array(np.sqrt(2.0), np.float32), "c")])])
            t = g.op("Add", [e, g.init(np.array(1.0, np.float32), "c")])
            return g.op("Mul", [g.op("Mul", [x, t]), g.init(np.array(0.5, np.float32), "c")])
        raise ValueError(kind)

    # ------------------------------------------------------------------ layers
    def conv(self, x, cin, cout, k, stride=1, groups=1, act=None, ink=None, pad=None, bias=True, w=None, b=None, pads4=None, gain=2.0):
        """ink: None | 'pass' (out0 <- box-filter of in0) | 'center' (out0 <- in0 through the centre tap: large kernels would blur the
        ink away) | 'zero' (out0 == 0) for the channel-0 ink path.  pads4: ONNX pads [top, left, bottom, right] when they are not symmetric
        (the "SAME" padding of an even kernel pads bottom / right only)."""
        kh, kw = (k, k) if isinstance(k, int) else k
        sh, sw = (stride, stride) if isinstance(stride, int) else stride
        if pad is None:
            pad = (kh // 2, kw // 2)
        cpg = cin // groups
        if w is None:
            w = self._w((cout, cpg, kh, kw), cpg * kh * kw, gain)
        if b is None:
            b = self._b(cout)
        if ink == "pass":
            w[0] = 0.0
            w[0, 0] = 1.0 / (kh * kw)
            b[0] = 0.0
        elif ink == "center":
            w[0] = 0.0
            w[0, 0, kh // 2, kw // 2] = 1.0
            b[0] = 0.0
        elif ink == "zero":
            w[0] = 0.0
            b[0] = 0.0
        ins = [x, self.g.init(w)]
        if bias:
            ins.append(self.g.init(b, "b"))
        y = self.g.op("Conv", ins, kernel_shape=[kh, kw], strides=[sh, sw], pads=list(pads4) if pads4 is not None else [pad[0], pad[1], pad[0], pad[1]],
                      group=groups, dilations=[1, 1])
        return self.act(y, act)

    def se(self, x, c, ink=False):
        """SE block: GAP -> 1x1 -> ReLU -> 1x1 -> HardSigmoid(0.2, 0.5) -> Mul"""
        g = self.g
        r = max(c // 4, 8)
        p = g.op("GlobalAveragePool", [x])
        w1, b1 = self._w((r, c, 1, 1), c), self._b(r)
        w2, b2 = self._w((c, r, 1, 1), r), self._b(c)
        if ink:  # keep channel 0 unscaled: hard-sigmoid saturates at 1 for input >= 2.5
            w2[0] = 0.0
            b2[0] = 3.0
        h = g.op("Conv", [p, g.init(w1), g.init(b1, "b")], kernel_shape=[1, 1], strides=[1, 1], pads=[0, 0, 0, 0], group=1, dilations=[1, 1])
        h = g.op("Relu", [h])
        h = g.op("Conv", [h, g.init(w2), g.init(b2, "b")], kernel_shape=[1, 1], strides=[1, 1], pads=[0, 0, 0, 0], group=1, dilations=[1, 1])
        h = self.act(h, "hsigmoid")
        return g.op("Mul", [x, h])

    def ds_block(self, x, cin, cout, k, stride, use_se=False, ink=None, act="hswish"):
        """DepthwiseSeparable: dw kxk (+act) [SE] pw 1x1 (+act)  (PP-LCNet)"""
        x = self.conv(x, cin, cin, k, stride, groups=cin, act=act, ink=ink)
        if use_se:
            x = self.se(x, cin, ink=ink == "pass")
        x = self.conv(x, cin, cout, 1, 1, act=act, ink=ink)
        return x

    def bn(self, x, c, ink=False):
        g = self.g
        gamma = (1.0 + 0.1 * self.rng.standard_normal(c)).astype(np.float32)
        beta = (0.05 * self.rng.standard_normal(c)).astype(np.float32)
        mean = (0.05 * self.rng.standard_normal(c)).astype(np.float32)
        var = (1.0 + 0.1 * self.rng.random(c)).astype(np.float32)
        if ink:
            gamma[0], beta[0], mean[0], var[0] = 1.0, 0.0, 0.0, 1.0 - 1e-5
        return g.op("BatchNormalization", [x, g.init(gamma), g.init(beta), g.init(mean), g.init(var)], epsilon=1e-5)

    def conv_transpose2x2(self, x, cin, cout, ink_gain=None, ink_bias=0.0, noise=1.0):
        """ConvTranspose 2x2 stride 2.  ink_gain: output channel 0 = ink_gain * in0 (+ noise-scaled random
        contributions from the other input channels when cout == 1) + ink_bias."""
        w = self._w((cin, cout, 2, 2), cin) * np.float32(noise)
        b = self._b(cout)
        if ink_gain is not None:
            if cout > 1:
                w[1:, 0] = 0.0          # out0 reads only in0
            w[0, 0] = ink_gain
            b[0] = ink_bias
        return self.g.op("ConvTranspose", [x, self.g.init(w.astype(np.float32)), self.g.init(b, "b")], kernel_shape=[2, 2],
                         strides=[2, 2], pads=[0, 0, 0, 0], group=1, dilations=[1, 1])


def build_det(size="tiny", seed=0, soft=False):
    """DB text detector.  Returns (onnx_bytes, info).  soft: a shallow final gain and 30x the weight on the random channels -- the probability
    map is no longer near-binary (thousands of pixels within 0.05 of the 0.3 threshold, blob borders that depend on the random-weight
    channels): the regime real PP-OCR weights put the post-processing in.
    size "server_hgnet" (and its narrow test twin "hgnet_small"): the PP-OCRv5 server detector's family, see build_det_hgnet."""
    if size in _HGNET_CFG:
        return build_det_hgnet(size, seed=seed, soft=soft)
    cfg = {
        #        stem  c2   c3   c4    c5   neck  p
        "tiny": (16, 24, 32, 64, 128, 256, 64, 16),
        # the same topology widened to the parameter count of the file it stands for: pp-ocrv6_tiny_det.onnx is 1 780 590 bytes ~ 0.445 M f32 parameters
        # (reference registry.rs:83); "tiny" above has 0.288 M.  447 089 parameters.  bench.py reports both (VERDICT r4 #7b).
        "tiny_full": (16, 32, 48, 80, 160, 320, 96, 16),
        "server": (32, 64, 128, 256, 512, 1024, 256, 64),
    }[size]
    stem, b2, c2, c3, c4, c5, nk, pc = cfg
    n = _Net(f"synth_db_{size}", seed, decomposed_hswish=True)
    g = n.g
    g.add_input("x", ["N", 3, "H", "W"])
    # stem: channel 0 = INK_AMPL * mean darkness of the 3x3 window (see module docstring)
    w = n._w((stem, 3, 3, 3), 27)
    b = n._b(stem)
    # input plane c holds BGR[c] normalised with mean/std in OUTPUT order (db.rs:404-415)
    for c in range(3):
        w[0, c] = -INK_AMPL * DB_STD[c] / 27.0
    b[0] = INK_AMPL * (1.0 - float(DB_MEAN.mean()))
    x = n.conv("x", 3, stem, 3, 2, act="hswish", w=w, b=b)
    x = n.ds_block(x, stem, b2, 3, 1, ink="pass")
    x = n.ds_block(x, b2, c2, 3, 2, ink="pass")
    f2 = n.ds_block(x, c2, c2, 3, 1, ink="pass")              # stride 4
    x = n.ds_block(f2, c2, c3, 3, 2)
    f3 = n.ds_block(x, c3, c3, 3, 1)                          # stride 8
    x = n.ds_block(f3, c3, c4, 3, 2)
    x = n.ds_block(x, c4, c4, 5, 1)
    f4 = n.ds_block(x, c4, c4, 5, 1)                          # stride 16
    x = n.ds_block(f4, c4, c5, 5, 2, use_se=True)
    f5 = n.ds_block(x, c5, c5, 5, 1, use_se=True)             # stride 32
    # neck
    in5 = n.conv(f5, c5, nk, 1, ink="zero")
    in4 = n.conv(f4, c4, nk, 1, ink="zero")
    in3 = n.conv(f3, c3, nk, 1, ink="zero")
    in2 = n.conv(f2, c2, nk, 1, ink="pass")

    def up(t, s):
        return g.op("Resize", [t, "", g.init(np.array([1, 1, s, s], np.float32), "scales")], mode="nearest",
                    coordinate_transformation_mode="asymmetric", nearest_mode="floor")

    out4 = g.op("Add", [in4, up(in5, 2)])
    out3 = g.op("Add", [in3, up(out4, 2)])
    out2 = g.op("Add", [in2, up(out3, 2)])
    p5 = n.conv(in5, nk, pc, 3)
    p4 = n.conv(out4, nk, pc, 3)
    p3 = n.conv(out3, nk, pc, 3)
    p2 = n.conv(out2, nk, pc, 3, ink="pass")
    fuse = g.op("Concat", [up(p5, 8), up(p4, 4), up(p3, 2), p2], axis=1)
    # head: the ink channel is concat index 3*pc
    hc = pc
    w = n._w((hc, 4 * pc, 3, 3), 4 * pc * 9)
    b = n._b(hc)
    w[0] = 0.0
    w[0, 3 * pc] = 1.0 / 9.0
    b[0] = 0.0
    x = n.conv(fuse, 4 * pc, hc, 3, w=w, b=b, bias=True)
    x = n.bn(x, hc, ink=True)
    x = g.op("Relu", [x])
    x = n.conv_transpose2x2(x, hc, hc, ink_gain=1.0)
    x = n.bn(x, hc, ink=True)
    x = g.op("Relu", [x])
    x = n.conv_transpose2x2(x, hc, 1, ink_gain=1.2, ink_bias=-4.0, noise=0.4) if soft else n.conv_transpose2x2(x, hc, 1, ink_gain=2.0, ink_bias=-7.0, noise=0.01)
    y = g.op("Sigmoid", [x])
    g.nodes.append(node("Identity", [y], ["prob"]))
    g.add_output("prob", ["N", 1, "H", "W"])
    return g.model(), {"params": g.n_params, "size": size}


def build_rec(size="tiny", vocab=6906, seed=1):
    """CRNN/SVTR CTC recognizer: in [n,3,48,W] -> out [n, W/8, vocab] softmax probabilities.
    size "svtrv2" (and its narrow test twin "svtrv2_small"): the SVTRv2 server recognizer's family, see build_rec_svtrv2 (T = W/4)."""
    if size in _SVTRV2_CFG:
        return build_rec_svtrv2(size, vocab=vocab, seed=seed)
    cfg = {
        #        stem b2  b3  b4   b5   b6   svtr_dim heads out
        "tiny": (16, 24, 48, 96, 192, 256, 64, 4, 64),
        # the parameter count of the file it stands for: pp-ocrv6_tiny_rec.onnx is 4 462 639 bytes ~ 1.116 M f32 parameters (reference registry.rs:84); "tiny" above
        # has 0.914 M.  The missing 0.2 M go where PP-OCR recognizers keep most of their parameters -- the CTC projection hidden x V behind the SVTR neck (PP-OCRv5
        # mobile: 120 x 18 385 = 2.2 M of 4.1 M; here 64 x 6906 = 0.44 M of 0.91 M): the neck's output width 64 -> 96 makes the head 96 x 6906 = 0.66 M (58 % of the
        # graph, the family's proportion) and the graph 1 135 700 parameters (+1.8 % against the file).  The first form of this graph (round 6, earlier) deepened the
        # backbone instead (two more 192 -> 192 5x5 blocks, one more 256-channel SE block: 1 103 284 parameters): "tiny_deep", kept for the comparison in DESIGN section 5
        "tiny_full": (16, 24, 48, 96, 192, 256, 64, 4, 96),
        "tiny_deep": (16, 24, 48, 96, 192, 256, 64, 4, 64),
        "server": (32, 64, 128, 256, 512, 768, 192, 6, 192),
    }[size]
    extra_b5, extra_b6 = (2, 1) if size == "tiny_deep" else (0, 0)
    stem, b2, b3, b4, b5, b6, dim, heads, outc = cfg
    n = _Net(f"synth_rec_{size}", seed, decomposed_hswish=False)
    g = n.g
    g.add_input("x", ["N", 3, 48, "W"])
    x = n.conv("x", 3, stem, 3, 2, act="hswish")                       # 24 x W/2
    x = n.ds_block(x, stem, b2, 3, 1)
    x = n.ds_block(x, b2, b3, 3, 1)
    x = n.ds_block(x, b3, b3, 3, 1)
    x = n.ds_block(x, b3, b4, 3, (2, 1))                               # 12 x W/2
    x = n.ds_block(x, b4, b4, 3, 1)
    x = n.ds_block(x, b4, b5, 3, (1, 2))                               # 12 x W/4
    x = n.ds_block(x, b5, b5, 5, 1)
    x = n.ds_block(x, b5, b5, 5, 1)
    for _ in range(extra_b5):
        x = n.ds_block(x, b5, b5, 5, 1)
    x = n.ds_block(x, b5, b6, 5, (2, 1), use_se=True)                  # 6 x W/4
    x = n.ds_block(x, b6, b6, 5, 1, use_se=True)
    for _ in range(extra_b6):
        x = n.ds_block(x, b6, b6, 5, 1, use_se=True)
    x = g.op("AveragePool", [x], kernel_shape=[6, 2], strides=[6, 2], pads=[0, 0, 0, 0])   # 1 x W/8
    # SVTR neck (EncoderWithSVTR)
    h = x
    z = n.conv(x, b6, b6 // 8, (1, 3), act="swish", pad=(0, 1))
    z = n.conv(z, b6 // 8, dim, 1, act="swish")
    z = g.op("Squeeze", [z, g.init(np.array([2], np.int64), "axes")])   # [n, dim, T]
    z = g.op("Transpose", [z], perm=[0, 2, 1])                           # [n, T, dim]
    hd = dim // heads

    def linear(t, cin, cout, gain=1.0):
        w = n._w((cin, cout), cin, gain)
        t = g.op("MatMul", [t, g.init(w)])
        return g.op("Add", [t, g.init(n._b(cout), "b")])

    def layernorm(t, c):
        gamma = (1.0 + 0.1 * n.rng.standard_normal(c)).astype(np.float32)
        beta = (0.05 * n.rng.standard_normal(c)).astype(np.float32)
        return g.op("LayerNormalization", [t, g.init(gamma), g.init(beta)], axis=-1, epsilon=1e-5)

    for _ in range(2):
        y = layernorm(z, dim)
        qkv = linear(y, dim, 3 * dim)
        qkv = g.op("Reshape", [qkv, g.init(np.array([0, -1, 3, heads, hd], np.int64), "shape")])
        qkv = g.op("Transpose", [qkv], perm=[2, 0, 3, 1, 4])            # [3, n, heads, T, hd]
        q, k, v = g.op("Split", [qkv], n_out=3, axis=0)
        ax0 = g.init(np.array([0], np.int64), "axes")
        q = g.op("Squeeze", [q, ax0])
        k = g.op("Squeeze", [k, ax0])
        v = g.op("Squeeze", [v, ax0])
        q = g.op("Mul", [q, g.init(np.array(hd ** -0.5, np.float32), "scale")])
        kt = g.op("Transpose", [k], perm=[0, 1, 3, 2])
        att = g.op("MatMul", [q, kt])                                    # [n, heads, T, T]
        att = g.op("Softmax", [att], axis=-1)
        o = g.op("MatMul", [att, v])                                     # [n, heads, T, hd]
        o = g.op("Transpose", [o], perm=[0, 2, 1, 3])
        o = g.op("Reshape", [o, g.init(np.array([0, -1, dim], np.int64), "shape")])
        o = linear(o, dim, dim)
        z = g.op("Add", [z, o])
        y = layernorm(z, dim)
        y = linear(y, dim, 2 * dim)
        y = n.act(y, "swish")
        y = linear(y, 2 * dim, dim)
        z = g.op("Add", [z, y])
    z = layernorm(z, dim)
    z = g.op("Transpose", [z], perm=[0, 2, 1])                           # [n, dim, T]
    z = g.op("Unsqueeze", [z, g.init(np.array([2], np.int64), "axes")])  # [n, dim, 1, T]
    z = n.conv(z, dim, b6, 1, act="swish")
    z = g.op("Concat", [h, z], axis=1)
    z = n.conv(z, 2 * b6, b6 // 8, (1, 3), act="swish", pad=(0, 1))
    z = n.conv(z, b6 // 8, outc, 1, act="swish")
    z = g.op("Squeeze", [z, g.init(np.array([2], np.int64), "axes")])
    z = g.op("Transpose", [z], perm=[0, 2, 1])                           # [n, T, outc]
    logits = linear(z, outc, vocab, gain=float(outc) * 4.0)               # spread logits: distinct argmax
    g.nodes.append(node("Softmax", [logits], ["probs"], axis=2))
    g.add_output("probs", ["N", "T", vocab])
    return g.model(), {"params": g.n_params, "size": size, "vocab": vocab}


# ---------------------------------------------------------------------------------------------- BASELINE C3: the server graphs
# pp-ocrv5_server_det.onnx is 88 116 836 bytes ~ 22.0 M f32 parameters (reference registry.rs:77): PP-HGNetV2-B4 backbone (stem with a 2x2 side
# branch, four stages of HG blocks -- a run of `layers` convolutions whose outputs are ALL concatenated with the block input and aggregated by two
# 1x1 convolutions; stages 1-2 use dense 3x3 convolutions, stages 3-4 the "light" form 1x1 + depthwise 5x5), LK-PAN neck (1x1 laterals to 256
# channels, top-down adds, 9x9 convolutions 256 -> 64 on every level, a bottom-up path of 3x3 stride-2 convolutions, 9x9 convolutions 64 -> 64,
# concat at stride 4) and the DB head.  Channel / depth numbers are PaddleOCR's published configuration of that detector [NOT in the reference: the
# reference only pins the file's byte size]; weights are random apart from the ink path.  An effective-squeeze-excite gate (GlobalAveragePool ->
# 1x1 -> Sigmoid -> Mul; PP-HGNet v1's, SURVEY Appendix B names it) closes the two dense stages: 0.28 M parameters (on all four stages its C x C
# weights would be 5.5 M and put the graph 25 % over the file it stands for).
_HGNET_CFG = {
    #                 stem (c1, c2a, c3, c4)   stages: (mid, out, blocks, light, k, layers)                                              neck  pan
    "server_hgnet": ((32, 16, 32, 48), ((48, 128, 1, False, 3, 6), (96, 512, 1, False, 3, 6), (192, 1024, 3, True, 5, 6), (384, 2048, 1, True, 5, 6)), 256, 64),
    # the same topology at 1/4 of the widths (1.4 M parameters): what the CPU oracle can run on several pages inside a test
    "hgnet_small": ((16, 8, 16, 16), ((16, 32, 1, False, 3, 6), (24, 128, 1, False, 3, 6), (48, 256, 3, True, 5, 6), (96, 512, 1, True, 5, 6)), 64, 16),
}


def build_det_hgnet(size="server_hgnet", seed=0, soft=False, ese=True):
    """PP-OCRv5-server-class DB detector (see the comment above).  Returns (onnx_bytes, info)."""
    (s1, s2a, s3, s4), stages, nk, pc = _HGNET_CFG[size]
    n = _Net(f"synth_db_{size}", seed)
    g = n.g
    g.add_input("x", ["N", 3, "H", "W"])
    same2 = [0, 0, 1, 1]   # "SAME" padding of a 2x2 kernel: bottom / right
    # ---- stem (stride 4).  Channel 0 carries the ink signal as in build_det
    w = n._w((s1, 3, 3, 3), 27)
    b = n._b(s1)
    for c in range(3):
        w[0, c] = -INK_AMPL * DB_STD[c] / 27.0
    b[0] = INK_AMPL * (1.0 - float(DB_MEAN.mean()))
    x1 = n.conv("x", 3, s1, 3, 2, act="relu", w=w, b=b)
    x2 = n.conv(x1, s1, s2a, 2, 1, act="relu", ink="pass", pads4=same2)
    x2 = n.conv(x2, s2a, s1, 2, 1, act="relu", ink="pass", pads4=same2)
    xp = g.op("MaxPool", [x1], kernel_shape=[2, 2], strides=[1, 1], pads=same2)
    x = g.op("Concat", [xp, x2], axis=1)
    x = n.conv(x, 2 * s1, s3, 3, 2, act="relu", ink="pass")
    x = n.conv(x, s3, s4, 1, 1, act="relu", ink="pass")

    def hg_block(x, cin, mid, cout, k, layers, light, identity, ink):
        outs = [x]
        t, c = x, cin
        for _ in range(layers):
            if light:   # LightConvBNAct: 1x1 without activation, then depthwise k x k with ReLU
                t = n.conv(t, c, mid, 1, 1, gain=1.0)       # (no activation behind it: a variance-preserving draw, there is no BatchNorm to rescale)
                t = n.conv(t, mid, mid, k, 1, groups=mid, act="relu")
            else:
                t = n.conv(t, c, mid, k, 1, act="relu")
            outs.append(t)
            c = mid
        total = cin + layers * mid
        y = g.op("Concat", outs, axis=1)
        y = n.conv(y, total, cout // 2, 1, 1, act="relu", ink=ink)     # aggregation squeeze: concat channel 0 is the block input's channel 0
        y = n.conv(y, cout // 2, cout, 1, 1, act="relu", ink=ink)      # aggregation excitation
        if identity:
            y = g.op("Add", [y, x])
        return y

    def ese_gate(x, c, ink):
        p = g.op("GlobalAveragePool", [x])
        w, b = n._w((c, c, 1, 1), c, gain=1.0), n._b(c)
        if ink:   # sigmoid(12) = 1 - 6e-6: the ink channel passes the gate unscaled
            w[0] = 0.0
            b[0] = 12.0
        h = g.op("Conv", [p, g.init(w), g.init(b, "b")], kernel_shape=[1, 1], strides=[1, 1], pads=[0, 0, 0, 0], group=1, dilations=[1, 1])
        return g.op("Mul", [x, g.op("Sigmoid", [h])])

    feats = []
    cin = s4
    for si, (mid, cout, blocks, light, k, layers) in enumerate(stages):
        ink = "pass" if si == 0 else None
        if si > 0:   # downsample: depthwise 3x3 stride 2 without activation
            x = n.conv(x, cin, cin, 3, 2, groups=cin, gain=1.0)
        for bi in range(blocks):
            x = hg_block(x, cin if bi == 0 else cout, mid, cout, k, layers, light, identity=bi > 0, ink=ink)
        if ese and not light:
            x = ese_gate(x, cout, ink=si == 0)
        feats.append((x, cout))
        cin = cout
    (f2, c2), (f3, c3), (f4, c4), (f5, c5) = feats
    # ---- LK-PAN
    in5 = n.conv(f5, c5, nk, 1, ink="zero", bias=False, gain=0.5)
    in4 = n.conv(f4, c4, nk, 1, ink="zero", bias=False, gain=0.5)
    in3 = n.conv(f3, c3, nk, 1, ink="zero", bias=False, gain=0.5)
    in2 = n.conv(f2, c2, nk, 1, ink="pass", bias=False, gain=0.5)

    def up(t, s):
        return g.op("Resize", [t, "", g.init(np.array([1, 1, s, s], np.float32), "scales")], mode="nearest",
                    coordinate_transformation_mode="asymmetric", nearest_mode="floor")

    out4 = g.op("Add", [in4, up(in5, 2)])
    out3 = g.op("Add", [in3, up(out4, 2)])
    out2 = g.op("Add", [in2, up(out3, 2)])
    q5 = n.conv(in5, nk, pc, 9, bias=False, gain=0.5)
    q4 = n.conv(out4, nk, pc, 9, bias=False, gain=0.5)
    q3 = n.conv(out3, nk, pc, 9, bias=False, gain=0.5)
    q2 = n.conv(out2, nk, pc, 9, ink="center", bias=False, gain=0.5)
    pan3 = g.op("Add", [q3, n.conv(q2, pc, pc, 3, 2, bias=False, gain=0.5)])
    pan4 = g.op("Add", [q4, n.conv(pan3, pc, pc, 3, 2, bias=False, gain=0.5)])
    pan5 = g.op("Add", [q5, n.conv(pan4, pc, pc, 3, 2, bias=False, gain=0.5)])
    p2 = n.conv(q2, pc, pc, 9, ink="center", bias=False, gain=0.5)
    p3 = n.conv(pan3, pc, pc, 9, bias=False, gain=0.5)
    p4 = n.conv(pan4, pc, pc, 9, bias=False, gain=0.5)
    p5 = n.conv(pan5, pc, pc, 9, bias=False, gain=0.5)
    fuse = g.op("Concat", [up(p5, 8), up(p4, 4), up(p3, 2), p2], axis=1)
    # ---- DB head; the ink channel is concat index 3 * pc
    hc = nk // 4
    w = n._w((hc, 4 * pc, 3, 3), 4 * pc * 9)
    b = n._b(hc)
    w[0] = 0.0
    w[0, 3 * pc] = 1.0 / 9.0
    b[0] = 0.0
    x = n.conv(fuse, 4 * pc, hc, 3, w=w, b=b, bias=True)
    x = n.bn(x, hc, ink=True)
    x = g.op("Relu", [x])
    x = n.conv_transpose2x2(x, hc, hc, ink_gain=1.0)
    x = n.bn(x, hc, ink=True)
    x = g.op("Relu", [x])
    x = n.conv_transpose2x2(x, hc, 1, ink_gain=1.2, ink_bias=-4.0, noise=0.4) if soft else n.conv_transpose2x2(x, hc, 1, ink_gain=2.0, ink_bias=-7.0, noise=0.01)
    y = g.op("Sigmoid", [x])
    g.nodes.append(node("Identity", [y], ["prob"]))
    g.add_output("prob", ["N", 1, "H", "W"])
    return g.model(), {"params": g.n_params, "size": size}


# ch_svtrv2_rec.onnx is 84 196 641 bytes ~ 21.0 M f32 parameters (reference registry.rs:25); it is used with ppocr_keys_v1.txt (6623 lines ->
# V = 6625).  SVTRv2-B as OpenOCR publishes it [NOT in the reference]: conv stem (two 3x3 stride-2 convolutions, GELU) to [n, 128, 12, W/4]; three
# stages of six mixing blocks at dims 128 / 256 / 384 -- "Conv" blocks (5x5 grouped convolution, groups = heads, post-norm, MLP ratio 4) in stage 1 and
# the first two blocks of stage 2, global self-attention blocks (heads 8 / 12, head dim 32) everywhere else -- with a 3x3 stride-(2,1) convolution +
# LayerNorm between stages; mean over the 3 remaining rows -> [n, W/4, 384] -> Linear -> Softmax.  T = W/4: the pipeline reads T from the output.
_SVTRV2_CFG = {
    #            dims              depths     heads       mixers per stage (C = conv, G = global)
    "svtrv2": ((128, 256, 384), (6, 6, 6), (4, 8, 12), ("CCCCCC", "CCGGGG", "GGGGGG")),
    "svtrv2_small": ((32, 64, 96), (2, 3, 2), (2, 4, 6), ("CC", "CGG", "GG")),
}


def build_rec_svtrv2(size="svtrv2", vocab=6625, seed=1):
    """SVTRv2-class CTC recognizer: in [n,3,48,W] (W % 4 == 0) -> out [n, W/4, vocab] softmax probabilities (see the comment above)."""
    dims, depths, heads, mixers = _SVTRV2_CFG[size]
    n = _Net(f"synth_rec_{size}", seed, decomposed_hswish=False)
    g = n.g
    i64 = np.int64
    g.add_input("x", ["N", 3, 48, "W"])
    x = n.conv("x", 3, dims[0] // 2, 3, 2, act="gelu")
    x = n.conv(x, dims[0] // 2, dims[0], 3, 2, act="gelu")                      # [n, d0, 12, W/4]

    def linear(t, cin, cout, gain=1.0):
        w = n._w((cin, cout), cin, gain)
        t = g.op("MatMul", [t, g.init(w)])
        return g.op("Add", [t, g.init(n._b(cout), "b")])

    def layernorm(t, c):
        gamma = (1.0 + 0.1 * n.rng.standard_normal(c)).astype(np.float32)
        beta = (0.05 * n.rng.standard_normal(c)).astype(np.float32)
        return g.op("LayerNormalization", [t, g.init(gamma), g.init(beta)], axis=-1, epsilon=1e-6)

    def to_tokens(m, c):      # [n, c, h, w] -> [n, h*w, c]
        t = g.op("Reshape", [m, g.init(np.array([0, c, -1], i64), "shape")])
        return g.op("Transpose", [t], perm=[0, 2, 1])

    def to_map(t, c, h):      # [n, h*w, c] -> [n, c, h, w]
        m = g.op("Transpose", [t], perm=[0, 2, 1])
        return g.op("Reshape", [m, g.init(np.array([0, c, h, -1], i64), "shape")])

    def mlp(t, c):
        y = linear(t, c, 4 * c)
        y = n.act(y, "gelu")
        return linear(y, 4 * c, c, gain=0.5)

    def conv_block(t, c, h, nh):
        m = n.conv(to_map(t, c, h), c, c, 5, 1, groups=nh)
        t = layernorm(g.op("Add", [t, to_tokens(m, c)]), c)
        return layernorm(g.op("Add", [t, mlp(t, c)]), c)

    def global_block(t, c, nh):
        hd = c // nh
        qkv = linear(t, c, 3 * c)
        qkv = g.op("Reshape", [qkv, g.init(np.array([0, -1, 3, nh, hd], i64), "shape")])
        qkv = g.op("Transpose", [qkv], perm=[2, 0, 3, 1, 4])
        q, k, v = g.op("Split", [qkv], n_out=3, axis=0)
        ax0 = g.init(np.array([0], i64), "axes")
        q = g.op("Squeeze", [q, ax0])
        k = g.op("Squeeze", [k, ax0])
        v = g.op("Squeeze", [v, ax0])
        q = g.op("Mul", [q, g.init(np.array(hd ** -0.5, np.float32), "scale")])
        att = g.op("MatMul", [q, g.op("Transpose", [k], perm=[0, 1, 3, 2])])
        att = g.op("Softmax", [att], axis=-1)
        o = g.op("MatMul", [att, v])
        o = g.op("Transpose", [o], perm=[0, 2, 1, 3])
        o = g.op("Reshape", [o, g.init(np.array([0, -1, c], i64), "shape")])
        o = linear(o, c, c, gain=0.5)
        t = layernorm(g.op("Add", [t, o]), c)
        return layernorm(g.op("Add", [t, mlp(t, c)]), c)

    h = 12
    t = to_tokens(x, dims[0])
    for si in range(3):
        c = dims[si]
        for kind in mixers[si][:depths[si]]:
            t = conv_block(t, c, h, heads[si]) if kind == "C" else global_block(t, c, heads[si])
        if si < 2:   # sub-sampling: 3x3 convolution, stride (2, 1), then LayerNorm over the channels
            m = n.conv(to_map(t, c, h), c, dims[si + 1], 3, (2, 1))
            h //= 2
            t = layernorm(to_tokens(m, dims[si + 1]), dims[si + 1])
    c = dims[2]
    m = to_map(t, c, h)                                                          # [n, c, 3, W/4]
    m = g.op("ReduceMean", [m], axes=[2], keepdims=0)                            # [n, c, W/4]
    z = g.op("Transpose", [m], perm=[0, 2, 1])                                   # [n, T, c]
    logits = linear(z, c, vocab, gain=float(c) * 4.0)                             # spread logits: distinct argmax
    g.nodes.append(node("Softmax", [logits], ["probs"], axis=2))
    g.add_output("probs", ["N", "T", vocab])
    return g.model(), {"params": g.n_params, "size": size, "vocab": vocab}


def synth_dict(n_chars=6904):
    """A dictionary file body with n_chars distinct single-character lines (CJK block + ASCII)."""
    chars = []
    for c in range(0x21, 0x7F):
        chars.append(chr(c))
    c = 0x4E00
    while len(chars) < n_chars:
        chars.append(chr(c))
        c += 1
    return "\n".join(chars[:n_chars]) + "\n"


# ---------------------------------------------------------------------------------------------- config 5 graphs
def build_cls(n_classes=4, seed=5, width=(16, 32, 64, 128)):
    """PP-LCNet-style image classifier (doc orientation: 4 classes on 224x224; text-line orientation: 2 classes on
    80x160): stem s2, depthwise-separable stages with SE on the last two, GAP -> 1x1 conv (hswish) -> Flatten ->
    Gemm -> Softmax.  Input "x" [n,3,H,W] f32, output [n, n_classes] probabilities
    (oar-ocr-core/src/models/classification/pp_lcnet.rs:207-240 reads output[0] as a 2-D array)."""
    net = _Net("synth_cls", seed)
    g = net.g
    g.add_input("x", ["N", 3, "H", "W"])
    c0, c1, c2, c3 = width
    x = net.conv("x", 3, c0, 3, 2, act="hswish")
    x = net.ds_block(x, c0, c1, 3, 2)
    x = net.ds_block(x, c1, c2, 3, 2)
    x = net.ds_block(x, c2, c2, 5, 1, use_se=True)
    x = net.ds_block(x, c2, c3, 5, 2, use_se=True)
    x = g.op("GlobalAveragePool", [x])
    x = net.conv(x, c3, 256, 1, 1, act="hswish")
    x = g.op("Flatten", [x], axis=1)
    w = net._w((256, n_classes), 256, gain=8.0)
    x = g.op("Gemm", [x, g.init(w), g.init(net._b(n_classes, 0.5), "b")])
    y = g.op("Softmax", [x], axis=-1)
    g.add_output(y, ["N", n_classes])
    return g.model(), {"params": g.param_count() if hasattr(g, "param_count") else None, "classes": n_classes}


def build_uvdoc(seed=6, width=(16, 32, 64), head="grid", size=512):
    """UVDoc-style rectifier stand-in: input "image" [n,3,512,512] BGR in [0,1], output [n,3,512,512] BGR in [0,1]
    (oar-ocr-core/src/models/rectification/uvdoc.rs:291-293 input name; :166-207 consumes output[0] as 4-D).
    head="grid" (default, the UVDoc topology): encoder (s2, s2, s2, PRelu blocks) -> bottleneck -> 2-channel sampling grid
    at 1/8 resolution (identity + a small predicted displacement) -> bilinear Resize to full size (align_corners) ->
    GridSample of the input image.  head="image": nearest x2 decoders with skip adds -> 3-channel Sigmoid head."""
    net = _Net("synth_uvdoc", seed)
    g = net.g
    g.add_input("image", ["N", 3, "H", "W"])
    c0, c1, c2 = width
    e0 = net.conv("image", 3, c0, 3, 2, act="relu")
    e1 = net.ds_block(e0, c0, c1, 3, 2, act="relu")
    e2 = net.ds_block(e1, c1, c2, 3, 2, act="relu")
    b = net.ds_block(e2, c2, c2, 5, 1, use_se=True, act="relu")
    if head == "grid":
        gs = size // 8
        t = net.conv(b, c2, c1, 3, 1, act=None)
        t = g.op("PRelu", [t, g.init((0.05 + 0.2 * net.rng.random((c1, 1, 1))).astype(np.float32), "slope")])
        disp = g.op("Tanh", [net.conv(t, c1, 2, 3, 1, act=None)])
        disp = g.op("Mul", [disp, g.init(np.array(0.02, np.float32), "gain")])          # +-2 % of the page: a mild warp
        lin = np.linspace(-1.0, 1.0, gs, dtype=np.float32)
        ident = np.stack([np.broadcast_to(lin[None, :], (gs, gs)), np.broadcast_to(lin[:, None], (gs, gs))])[None]   # [1,2,gs,gs]: x, y
        grid = g.op("Add", [disp, g.init(np.ascontiguousarray(ident, np.float32), "identity_grid")])
        grid = g.op("Resize", [grid, "", g.init(np.array([1.0, 1.0, 8.0, 8.0], np.float32), "scales")], mode="linear", coordinate_transformation_mode="align_corners")
        grid = g.op("Transpose", [grid], perm=[0, 2, 3, 1])
        y = g.op("GridSample", ["image", grid], mode="linear", padding_mode="border", align_corners=1)
        g.add_output(y, ["N", 3, "H", "W"])
        return g.model(), {"classes": 3}
    scales = g.init(np.array([1.0, 1.0, 2.0, 2.0], np.float32), "scales")

    def up(x):
        return g.op("Resize", [x, "", scales], mode="nearest", coordinate_transformation_mode="asymmetric", nearest_mode="floor")

    d1 = net.conv(up(b), c2, c1, 3, 1, act="relu")
    d1 = g.op("Add", [d1, e1])
    d0 = net.conv(up(d1), c1, c0, 3, 1, act="relu")
    d0 = g.op("Add", [d0, e0])
    y = net.conv(up(d0), c0, 3, 3, 1, act="sigmoid")
    g.add_output(y, ["N", 3, "H", "W"])
    return g.model(), {"classes": 3}


# ---------------------------------------------------------------------------------------------- exporter-shaped fixture
def build_p2o_fixture(seed=11, c=32, vocab=37):
    """A small recognizer-like graph written the way Paddle2ONNX / torch.onnx write real exports, i.e. WITHOUT the
    conveniences the other synthetic graphs use (first contact with a real file was likely
    OAR_UNSUPPORTED_OP):
      * Conv -> BatchNormalization kept as separate nodes; HardSwish decomposed to Add(3) / Clip(0, 6) / Mul / Div(6);
        SE gate from ReduceMean(axes 2, 3) + HardSigmoid decomposed to Mul / Add / Clip;
      * dynamic Resize: sizes = Concat(Slice(Shape(x), 0:2), Cast(Mul(Cast(Slice(Shape(x), 2:4), float), 2.0), int64));
      * Reshape targets built from Shape -> Gather -> Unsqueeze -> Concat, a `0`-entry resolved by Where(Equal(shape, 0), Shape, shape);
      * LayerNorm decomposed to ReduceMean / Sub / Pow / ReduceMean / Add / Sqrt / Div / Mul / Add;
      * a positional term from Range -> Cast -> Div -> Unsqueeze, a bias from ConstantOfShape -> Expand, a learned row Tile'd to
        [N, T, C] with repeats from Shape; leaky activation as Where(Greater(x, 0), x, 0.1 x); clamp as Min(Max(x, lo), hi);
      * softmax spelled Exp / ReduceSum / Div after a ReduceMax shift.
    Input "x" [N, 3, 32, W] (W % 8 == 0), output [N, T, vocab] with T = W / 4."""
    net = _Net("synth_p2o_fixture", seed, opset=17)
    g, rng = net.g, net.rng
    f32, i64 = np.float32, np.int64
    g.add_input("x", ["N", 3, 32, "W"])

    def conv_bn_hswish(x, cin, cout, k, stride, groups=1):
        w = net._w((cout, cin // groups, k, k), (cin // groups) * k * k)
        y = g.op("Conv", [x, g.init(w)], kernel_shape=[k, k], strides=list(stride), pads=[k // 2] * 4, group=groups, dilations=[1, 1])
        y = net.bn(y, cout)
        t = g.op("Add", [y, g.init(np.array([3.0], f32))])
        t = g.op("Clip", [t, g.init(np.array(0.0, f32)), g.init(np.array(6.0, f32))])
        t = g.op("Mul", [y, t])
        return g.op("Div", [t, g.init(np.array([6.0], f32))])

    x = conv_bn_hswish("x", 3, c // 2, 3, (2, 2))                 # 16 x W/2
    x = conv_bn_hswish(x, c // 2, c // 2, 3, (1, 1), groups=c // 2)
    x = conv_bn_hswish(x, c // 2, c, 1, (1, 1))
    # SE: ReduceMean over (2, 3), decomposed hard-sigmoid
    p = g.op("ReduceMean", [x], axes=[2, 3], keepdims=1)
    h = g.op("Relu", [g.op("Conv", [p, g.init(net._w((c // 4, c, 1, 1), c)), g.init(net._b(c // 4), "b")], kernel_shape=[1, 1], strides=[1, 1], pads=[0] * 4, group=1, dilations=[1, 1])])
    h = g.op("Conv", [h, g.init(net._w((c, c // 4, 1, 1), c // 4)), g.init(net._b(c), "b")], kernel_shape=[1, 1], strides=[1, 1], pads=[0] * 4, group=1, dilations=[1, 1])
    h = g.op("Clip", [g.op("Add", [g.op("Mul", [h, g.init(np.array([0.2], f32))]), g.init(np.array([0.5], f32))]),
                      g.init(np.array(0.0, f32)), g.init(np.array(1.0, f32))])
    x = g.op("Mul", [x, h])
    # down to 8 x W/4, then a dynamic nearest x2 back up to 16 x W/2 and a skip add (FPN-style)
    d = conv_bn_hswish(x, c, c, 3, (2, 2))
    shp = g.op("Shape", [d])
    nc = g.op("Slice", [shp, g.init(np.array([0], i64)), g.init(np.array([2], i64)), g.init(np.array([0], i64))])
    hw = g.op("Slice", [shp, g.init(np.array([2], i64)), g.init(np.array([4], i64)), g.init(np.array([0], i64))])
    hw2 = g.op("Cast", [g.op("Mul", [g.op("Cast", [hw], to=1), g.init(np.array([2.0, 2.0], f32))])], to=7)
    sizes = g.op("Concat", [nc, hw2], axis=0)
    up = g.op("Resize", [d, "", "", sizes], mode="nearest", coordinate_transformation_mode="asymmetric", nearest_mode="floor")
    x = g.op("Add", [x, up])
    x = g.op("AveragePool", [x, ], kernel_shape=[16, 2], strides=[16, 2], pads=[0, 0, 0, 0])      # [N, c, 1, W/4]
    # [N, c, 1, T] -> [N, c, T] with a Paddle-style reshape: target [0, c, -1] whose 0 is resolved by Where(Equal(.., 0), Shape, ..)
    tgt = g.init(np.array([0, c, -1], i64))
    s3 = g.op("Slice", [g.op("Shape", [x]), g.init(np.array([0], i64)), g.init(np.array([3], i64)), g.init(np.array([0], i64))])
    tgt = g.op("Where", [g.op("Equal", [tgt, g.init(np.array([0, 0, 0], i64))]), s3, tgt])
    x = g.op("Reshape", [x, tgt])
    x = g.op("Transpose", [x], perm=[0, 2, 1])                                                      # [N, T, c]
    # shape scalars
    xs = g.op("Shape", [x])
    n_ = g.op("Unsqueeze", [g.op("Gather", [xs, g.init(np.array(0, i64))], axis=0), g.init(np.array([0], i64))])
    t_ = g.op("Gather", [xs, g.init(np.array(1, i64))], axis=0)
    t1 = g.op("Unsqueeze", [t_, g.init(np.array([0], i64))])
    # positional ramp: Range(0, T, 1) / T -> [1, T, 1]
    ramp = g.op("Div", [g.op("Cast", [g.op("Range", [g.init(np.array(0, i64)), t_, g.init(np.array(1, i64))])], to=1), g.op("Cast", [t_], to=1)])
    ramp = g.op("Unsqueeze", [ramp, g.init(np.array([0, 2], i64))])
    x = g.op("Add", [x, ramp])
    # bias: ConstantOfShape([T, c]) = 0.25 -> Expand to [N, T, c]
    cos = g.op("ConstantOfShape", [g.op("Concat", [t1, g.init(np.array([c], i64))], axis=0)], value=np.array([0.25], f32))
    x = g.op("Sub", [x, g.op("Expand", [cos, g.op("Concat", [n_, t1, g.init(np.array([c], i64))], axis=0)])])
    # learned row, Tile'd to [N, T, c]
    row = g.init((0.1 * rng.standard_normal((1, 1, c))).astype(f32))
    x = g.op("Add", [x, g.op("Tile", [row, g.op("Concat", [n_, t1, g.init(np.array([1], i64))], axis=0)])])
    # decomposed LayerNorm over the last axis
    mu = g.op("ReduceMean", [x], axes=[-1], keepdims=1)
    xc = g.op("Sub", [x, mu])
    var = g.op("ReduceMean", [g.op("Pow", [xc, g.init(np.array(2.0, f32))])], axes=[-1], keepdims=1)
    x = g.op("Div", [xc, g.op("Sqrt", [g.op("Add", [var, g.init(np.array(1e-5, f32))])])])
    x = g.op("Add", [g.op("Mul", [x, g.init((1.0 + 0.1 * rng.standard_normal(c)).astype(f32))]), g.init((0.05 * rng.standard_normal(c)).astype(f32))])
    # MatMul + Add, leaky via Greater / Where, clamp via Max / Min
    x = g.op("Add", [g.op("MatMul", [x, g.init(net._w((c, c), c))]), g.init(net._b(c))])
    x = g.op("Where", [g.op("Greater", [x, g.init(np.array(0.0, f32))]), x, g.op("Mul", [x, g.init(np.array(0.1, f32))])])
    x = g.op("Min", [g.op("Max", [x, g.init(np.array(-4.0, f32))]), g.init(np.array(4.0, f32))])
    # head + softmax spelled out (ReduceMax shift, Exp, ReduceSum (opset-13 axes input), Div)
    z = g.op("Add", [g.op("MatMul", [x, g.init(net._w((c, vocab), c))]), g.init(net._b(vocab))])
    z = g.op("Sub", [z, g.op("ReduceMax", [z], axes=[-1], keepdims=1)])
    e = g.op("Exp", [z])
    y = g.op("Div", [e, g.op("ReduceSum", [e, g.init(np.array([-1], i64))], keepdims=1)])
    g.nodes.append(node("Identity", [y], ["probs"]))
    g.add_output("probs", ["N", "T", vocab])
    return g.model(), {"params": g.n_params}


# ---------------------------------------------------------------------------------------------- several inputs / integer outputs
def build_multi_io_fixture(seed=5, c=16, vocab=29):
    """Seam-A surface the single-input OCR graphs never touch (OrtInfer::infer with several named inputs and
    TensorOutput::I64, core/inference/ort_infer_execution.rs:121-219, tensor_output.rs:16-21), shaped like the
    reference's detection-style exports: inputs "image" [N, 3, H, W], "scale_factor" [N, 2], "im_shape" [N, 2];
    outputs "boxes" f32 [N, 2] (a head scaled by scale_factor and offset by im_shape), "ids" i64 [N, T = H/4] (ArgMax over
    the class axis of a per-row head), "dims" i64 [4] (Shape(image): a plan-time host value)."""
    net = _Net("synth_multi_io", seed, opset=17)
    g = net.g
    f32, i64 = np.float32, np.int64
    g.add_input("image", ["N", 3, "H", "W"])
    g.add_input("scale_factor", ["N", 2])
    g.add_input("im_shape", ["N", 2])
    x = net.conv("image", 3, c, 3, stride=2, act="relu")
    x = net.conv(x, c, c, 3, stride=2, act="hswish")
    p = g.op("Flatten", [g.op("GlobalAveragePool", [x])], axis=1)                       # [N, c]
    b = g.op("Gemm", [p, g.init(net._w((c, 2), c)), g.init(net._b(2))])                 # [N, 2]
    b = g.op("Add", [g.op("Mul", [b, "scale_factor"]), "im_shape"])
    g.nodes.append(node("Identity", [b], ["boxes"]))
    col = g.op("ReduceMean", [x], axes=[3], keepdims=0)                                 # [N, c, H/4]
    col = g.op("Transpose", [col], perm=[0, 2, 1])                                      # [N, T, c]
    z = g.op("Add", [g.op("MatMul", [col, g.init(net._w((c, vocab), c))]), g.init(net._b(vocab))])
    g.nodes.append(node("ArgMax", [z], ["ids"], axis=-1, keepdims=0))
    g.nodes.append(node("Shape", ["image"], ["dims"]))
    g.add_output("boxes", ["N", 2])
    g.add_output("ids", ["N", "T"], elem_type=7)
    g.add_output("dims", [4], elem_type=7)
    return g.model(), {"params": g.n_params}


# ---------------------------------------------------------------------------------------------- f4: layout detector graphs
def build_layout(kind="picodet", n_classes=5, seed=11, image_shape=(800, 608), feat=None):
    """A PicoDet / PP-DocLayout-shaped layout detector with synthetic weights.
    Inputs as the exported Paddle detection graphs declare them (models/detection/scale_aware_detector.rs:248-290): "image"
    [N,3,H,W], "scale_factor" [N,2] = (resized_h / orig_h, resized_w / orig_w), and for kind "pp-doclayout" also "im_shape" [N,2].
    Output: ONE 2-D tensor [N * K, feat] of rows (class_id, score, x1, y1, x2, y2[, col, row]) in ORIGINAL-image pixels -- what the
    reference reshapes to [N, K, 1, feat] (:325-339); K = anchors of the stride-32 and stride-16 grids.  The box decode inside the graph is
    the detectors' own: anchor centre -/+ stride * softplus-like distances, divided by the scale factor.  No NMS node: the suppression is
    LayoutPostProcess's (the reference runs its own NMS on whatever the graph returns, layout_postprocess.rs:482-548)."""
    feat = feat or (8 if kind == "pp-doclayout" else 6)
    H, W = image_shape
    n = _Net(f"synth_layout_{kind}", seed, decomposed_hswish=False)
    g = n.g
    g.add_input("image", ["N", 3, H, W])
    g.add_input("scale_factor", ["N", 2])
    if kind == "pp-doclayout":
        g.add_input("im_shape", ["N", 2])
    c1, c2, c3, c4, c5 = 16, 24, 48, 96, 128
    x = n.conv("image", 3, c1, 3, 2, act="hswish")
    x = n.ds_block(x, c1, c2, 3, 2)
    x = n.ds_block(x, c2, c3, 3, 2)
    x = n.ds_block(x, c3, c3, 3, 1)
    f16 = n.ds_block(x, c3, c4, 3, 2)                                  # stride 16
    f16 = n.ds_block(f16, c4, c4, 5, 1)
    f32_ = n.ds_block(f16, c4, c5, 5, 2, use_se=True)                  # stride 32
    f32_ = n.ds_block(f32_, c5, c5, 5, 1, use_se=True)
    rows = []
    for ft, cin, stride in ((f16, c4, 16), (f32_, c5, 32)):
        h, w = H // stride, W // stride
        k = h * w
        t = n.conv(ft, cin, 64, 3, act="hswish")
        cls = n.conv(t, 64, n_classes, 1, w=n._w((n_classes, 64, 1, 1), 64, gain=24.0), b=(n._b(n_classes) - 2.0).astype(np.float32))   # sparse confident anchors
        reg = n.conv(t, 64, 4, 1, w=n._w((4, 64, 1, 1), 64, gain=6.0), b=(n._b(4) + 1.0).astype(np.float32))
        cls = g.op("Reshape", [g.op("Transpose", [cls], perm=[0, 2, 3, 1]), g.init(np.array([0, k, n_classes], np.int64), "shape")])     # [N, K, C]
        reg = g.op("Reshape", [g.op("Transpose", [reg], perm=[0, 2, 3, 1]), g.init(np.array([0, k, 4], np.int64), "shape")])             # [N, K, 4]
        prob = g.op("Sigmoid", [cls])
        score = g.op("ReduceMax", [prob], axes=[2], keepdims=1)                                                                         # [N, K, 1]
        cid = g.op("Cast", [g.op("ArgMax", [prob], axis=2, keepdims=1)], to=1)
        dist = g.op("Mul", [g.op("Softplus", [reg]), g.init(np.array(float(stride) * 1.5, np.float32), "stride")])                      # l, t, r, b in resized pixels
        ys, xs = np.mgrid[0:h, 0:w]
        ctr = np.stack([(xs + 0.5) * stride, (ys + 0.5) * stride, (xs + 0.5) * stride, (ys + 0.5) * stride], -1).reshape(1, k, 4).astype(np.float32)
        sign = np.array([-1, -1, 1, 1], np.float32).reshape(1, 1, 4)
        box = g.op("Add", [g.init(ctr, "centres"), g.op("Mul", [dist, g.init(sign, "sign")])])                                          # x1 y1 x2 y2, resized pixels
        # / (scale_x, scale_y, scale_x, scale_y): scale_factor is (scale_y, scale_x)
        ax1 = g.init(np.array([1], np.int64), "axes")
        sy = g.op("Slice", ["scale_factor", g.init(np.array([0], np.int64), "starts"), g.init(np.array([1], np.int64), "ends"), ax1])    # [N, 1]
        sx = g.op("Slice", ["scale_factor", g.init(np.array([1], np.int64), "starts"), g.init(np.array([2], np.int64), "ends"), ax1])
        sf = g.op("Concat", [sx, sy, sx, sy], axis=1)                                                                                    # [N, 4]
        sf = g.op("Unsqueeze", [sf, g.init(np.array([1], np.int64), "axes")])                                                           # [N, 1, 4]
        box = g.op("Div", [box, sf])
        cols = [cid, score, box]
        if feat == 8:   # PP-DocLayoutV2's reading-order columns: (col, row) of the anchor grid cell, coarse
            order = np.stack([(xs // max(w // 2, 1)).astype(np.float32), ys.astype(np.float32) * w + xs], -1).reshape(1, k, 2).astype(np.float32)
            cols.append(g.op("Add", [g.op("Mul", [score, g.init(np.zeros((1, 1, 2), np.float32), "zero")]), g.init(order, "order")]))     # broadcast to [N, K, 2]
        rows.append(g.op("Concat", cols, axis=2))
    allr = g.op("Concat", rows, axis=1)                                                                                                  # [N, K, feat]
    g.nodes.append(node("Reshape", [allr, g.init(np.array([-1, feat], np.int64), "shape")], ["boxes"]))
    g.add_output("boxes", ["M", feat])
    return g.model(), {"params": g.n_params, "kind": kind, "classes": n_classes, "anchors": (H // 16) * (W // 16) + (H // 32) * (W // 32), "feat": feat}
