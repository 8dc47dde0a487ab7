"""The torch-CPU network oracle (oracle/onnx_ref.py) against a second, torch-free evaluation of the same graphs (oracle/onnx_np.py: numpy, float64,
operator semantics restated from the ONNX specification).  VERDICT r3 "missing" #2: ONNX Runtime is not installable here, and every GPU network-parity
test compares the HIP engine with onnx_ref alone -- a wrong attribute mapping in onnx_ref would sit on both sides of those tests.  Here it cannot: the
two evaluators share the protobuf reader and nothing else.  Tolerance: float32 torch kernels vs float64 sums, 2e-5 of the output's scale."""
import numpy as np
import pytest

from oar_ocr_amd.synth import models, pages
from oar_ocr_amd.synth.onnx_writer import GraphBuilder
from oracle import cpu_ref as R
from oracle import onnx_np, onnx_ref


def _agree(model_bytes, feeds, tol=2e-5):
    m = onnx_ref.parse_model(model_bytes)
    a = onnx_ref.run(m, feeds)
    b = onnx_np.run(m, feeds)
    assert len(a) == len(b)
    for x, y in zip(a, b):
        x, y = np.asarray(x), np.asarray(y)
        assert x.shape == y.shape, (x.shape, y.shape)
        if x.dtype.kind in "iub":
            assert np.mean(x != y) < 0.005, "integer outputs (argmax ties only)"      # a tie between two classes may break differently in f32 / f64
        else:
            assert np.abs(x.astype(np.float64) - y).max() <= tol * max(1.0, float(np.abs(y).max())), float(np.abs(x - y).max())
    return a


def test_detector_and_recognizer_graphs():
    det, _ = models.build_det("tiny", seed=0)
    x, _ = R.det_preprocess(pages.make_page(3, (96, 128), lines=2))
    (p,) = _agree(det, {"x": x[None]})
    assert p.shape == (1, 1, 96, 128) and 0.0 <= p.min() and p.max() <= 1.0
    rec, _ = models.build_rec("tiny", vocab=97, seed=1)
    xr = R.rec_preprocess([pages.make_crop(i, w, 48) for i, w in enumerate((120, 77))])
    (q,) = _agree(rec, {"x": xr})
    assert q.shape[0] == 2 and q.shape[2] == 97 and np.allclose(q.sum(-1), 1.0, atol=1e-4)


def test_classifier_rectifier_and_layout_graphs():
    rng = np.random.default_rng(4)
    cls, _ = models.build_cls(4, seed=5)
    _agree(cls, {onnx_ref.parse_model(cls)["inputs"][0]: rng.standard_normal((2, 3, 64, 64)).astype(np.float32)})
    uv, _ = models.build_uvdoc(seed=6, size=128)          # (the identity grid is baked in for one input size)
    _agree(uv, {onnx_ref.parse_model(uv)["inputs"][0]: rng.random((1, 3, 128, 128)).astype(np.float32)}, tol=1e-4)
    for kind in ("picodet", "pp-doclayout"):
        m, info = models.build_layout(kind, image_shape=(64, 64))
        pm = onnx_ref.parse_model(m)
        feeds = {"image": rng.standard_normal((2, 3, 64, 64)).astype(np.float32), "scale_factor": np.array([[0.5, 0.25], [1.0, 2.0]], np.float32)}
        if "im_shape" in pm["inputs"]:
            feeds["im_shape"] = np.array([[64, 64]] * 2, np.float32)
        _agree(m, feeds, tol=1e-4)


@pytest.mark.parametrize("case", ["conv_variants", "convtranspose_pool_resize", "sequence"])
def test_operator_attribute_corners(case):
    """attribute combinations the model graphs do not reach: asymmetric pads, dilation, groups, strides; ConvTranspose with output_padding; pooling
    with count_include_pad / ceil_mode; every Resize coordinate mode the interpreter accepts; LayerNormalization / Softmax on inner axes."""
    rng = np.random.default_rng(11)
    g = GraphBuilder("ops")
    if case == "conv_variants":
        g.add_input("x", ["N", 8, "H", "W"])
        outs = []
        for k, s, p, d, grp in ((3, (1, 1), [1, 1, 1, 1], (1, 1), 1), (3, (2, 1), [0, 1, 2, 1], (1, 2), 2), (5, (2, 2), [2, 2, 2, 2], (1, 1), 8), (1, (1, 2), [0, 0, 0, 0], (1, 1), 4)):
            w = (rng.standard_normal((16, 8 // grp, k, k)) * 0.2).astype(np.float32)
            outs.append(g.op("Conv", ["x", g.init(w), g.init(rng.standard_normal(16).astype(np.float32))], kernel_shape=[k, k], strides=list(s), pads=p, group=grp, dilations=list(d)))
        for o in outs:
            g.add_output(o, ["N", 16, "H", "W"])
        _agree(g.model(), {"x": rng.standard_normal((2, 8, 13, 17)).astype(np.float32)})
    elif case == "convtranspose_pool_resize":
        g.add_input("x", ["N", 6, "H", "W"])
        w2 = (rng.standard_normal((6, 4, 2, 2)) * 0.3).astype(np.float32)
        w3 = (rng.standard_normal((6, 2, 3, 3)) * 0.3).astype(np.float32)
        a = g.op("ConvTranspose", ["x", g.init(w2), g.init(rng.standard_normal(4).astype(np.float32))], kernel_shape=[2, 2], strides=[2, 2], pads=[0, 0, 0, 0], group=1, dilations=[1, 1])
        b = g.op("ConvTranspose", ["x", g.init(w3)], kernel_shape=[3, 3], strides=[2, 2], pads=[1, 1, 1, 1], output_padding=[1, 1], group=3, dilations=[1, 1])
        c = g.op("AveragePool", ["x"], kernel_shape=[3, 3], strides=[2, 2], pads=[1, 1, 1, 1], count_include_pad=1)
        d = g.op("AveragePool", ["x"], kernel_shape=[3, 2], strides=[2, 2], pads=[1, 0, 1, 0])
        e = g.op("MaxPool", ["x"], kernel_shape=[3, 3], strides=[2, 2], pads=[1, 1, 1, 1])
        # ceil_mode (round 6, found by tools/op_fuzz.py): the window hanging over the padded extent -- with count_include_pad the divisor counts padding, not the overhang
        e2 = [g.op("AveragePool", ["x"], kernel_shape=[3, 2], strides=[3, 2], pads=[0, 0, 0, 0], ceil_mode=1, count_include_pad=1),
              g.op("AveragePool", ["x"], kernel_shape=[3, 3], strides=[2, 2], pads=[1, 1, 1, 1], ceil_mode=1, count_include_pad=1),
              g.op("AveragePool", ["x"], kernel_shape=[4, 3], strides=[3, 2], pads=[2, 1, 2, 1], ceil_mode=1, count_include_pad=0),
              g.op("MaxPool", ["x"], kernel_shape=[3, 4], strides=[3, 2], pads=[0, 2, 0, 2], ceil_mode=1)]
        sc = lambda fy, fx: g.init(np.array([1, 1, fy, fx], np.float32))
        r = [g.op("Resize", ["x", "", sc(2, 2)], mode="nearest", coordinate_transformation_mode="asymmetric", nearest_mode="floor"),
             g.op("Resize", ["x", "", sc(1.5, 2.5)], mode="nearest", coordinate_transformation_mode="half_pixel", nearest_mode="round_prefer_floor"),
             g.op("Resize", ["x", "", sc(2, 3)], mode="linear", coordinate_transformation_mode="half_pixel"),
             g.op("Resize", ["x", "", sc(0.5, 0.5)], mode="linear", coordinate_transformation_mode="align_corners"),
             # scales whose floor(in * scale) / in differs from the scale (the coordinate follows the SCALE, as ONNX Runtime does), and linear + asymmetric
             g.op("Resize", ["x", "", sc(0.7, 1.3)], mode="linear", coordinate_transformation_mode="half_pixel"),
             g.op("Resize", ["x", "", sc(1.3, 0.45)], mode="linear", coordinate_transformation_mode="pytorch_half_pixel"),
             g.op("Resize", ["x", "", sc(2.5, 1.7)], mode="linear", coordinate_transformation_mode="asymmetric")]
        for o in [a, b, c, d, e] + e2 + r:
            g.add_output(o, ["N", "C", "H", "W"])
        _agree(g.model(), {"x": rng.standard_normal((2, 6, 10, 14)).astype(np.float32)})
    else:
        g.add_input("x", ["N", "T", 24])
        ln = g.op("LayerNormalization", ["x", g.init(rng.standard_normal(24).astype(np.float32)), g.init(rng.standard_normal(24).astype(np.float32))], axis=-1, epsilon=1e-5)
        mm = g.op("MatMul", [ln, g.init((rng.standard_normal((24, 10)) * 0.3).astype(np.float32))])
        sm = g.op("Softmax", [mm], axis=1)
        tr = g.op("Transpose", [sm], perm=[0, 2, 1])
        am = g.op("ArgMax", [mm], axis=2, keepdims=0)
        g.add_output(tr, ["N", 10, "T"])
        g.add_output(am, ["N", "T"])
        _agree(g.model(), {"x": rng.standard_normal((3, 7, 24)).astype(np.float32)})
