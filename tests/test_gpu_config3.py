"""BASELINE C3 on graphs of the size and kind it names: the PP-HGNetV2 / LK-PAN detector (21.7 M parameters, dense 3x3 and
9x9 convolutions) and the SVTRv2 recognizer (20.5 M parameters, grouped 5x5 mixing + global attention over 480 ... 2400 tokens) of
oar_ocr_amd/synth/models.py, against the torch-CPU oracle; plus the two pieces the engine grew for them: the streaming bf16x6 attention kernel
(csrc/attention_x6.hip) and the decomposed-GELU rewrite (engine.cc pass 2b).  Tolerance as in test_gpu_engine.py: 2e-4 (north_star allows 1e-3)."""
import numpy as np
import pytest

from oar_ocr_amd import api
from oar_ocr_amd.synth import models, pages
from oar_ocr_amd.synth.onnx_writer import GraphBuilder
from oracle import cpu_ref as R
from oracle import onnx_ref, pipeline_ref

pytestmark = pytest.mark.gpu
TOL = 2e-4


def _check(model_bytes, x, tol=TOL):
    eng = api.OrtInfer(model_bytes)
    got = eng.infer(x)
    ref = onnx_ref.run(model_bytes, {eng.input_name(): x})
    assert len(got) == len(ref)
    for (name, g), r in zip(got, ref):
        assert g.shape == r.shape, (name, g.shape, r.shape)
        d = np.abs(g - r).max() if g.size else 0.0
        scale = max(1.0, float(np.abs(r).max())) if r.size else 1.0
        assert d <= tol * scale, (name, d, scale)
    return got, ref


def _attention_graph(dim, heads, seed, gain=1.0):
    g = GraphBuilder("attn")
    rng = np.random.default_rng(seed)
    hd = dim // heads
    g.add_input("x", ["N", "T", dim])
    w = (gain * rng.standard_normal((dim, 3 * dim)) / np.sqrt(dim)).astype(np.float32)
    qkv = g.op("MatMul", ["x", g.init(w)])
    qkv = g.op("Add", [qkv, g.init((0.1 * rng.standard_normal(3 * dim)).astype(np.float32))])
    qkv = g.op("Reshape", [qkv, g.init(np.array([0, -1, 3, heads, hd], np.int64), "shape")])
    qkv = g.op("Transpose", [qkv], perm=[2, 0, 3, 1, 4])
    q, k, v = g.op("Split", [qkv], n_out=3, axis=0)
    ax0 = g.init(np.array([0], np.int64), "axes")
    q, k, v = g.op("Squeeze", [q, ax0]), g.op("Squeeze", [k, ax0]), g.op("Squeeze", [v, ax0])
    q = g.op("Mul", [q, g.init(np.array(hd ** -0.5, np.float32), "scale")])
    att = g.op("Softmax", [g.op("MatMul", [q, g.op("Transpose", [k], perm=[0, 1, 3, 2])])], axis=-1)
    o = g.op("Transpose", [g.op("MatMul", [att, v])], perm=[0, 2, 1, 3])
    o = g.op("Reshape", [o, g.init(np.array([0, -1, dim], np.int64), "shape")])
    g.add_output(o, ["N", "T", dim])
    return g.model()


@pytest.mark.parametrize("case", [
    ("one block and a bit", 64, 2, 3, 33, 1.0),
    ("one query tile", 64, 2, 2, 128, 1.0),
    ("svtrv2 stage 2 at W = 320", 256, 8, 2, 480, 1.0),
    ("svtrv2 stage 3, ragged tails", 384, 12, 1, 1001, 1.0),
    ("wide crop, peaked soft-max", 64, 2, 1, 2400, 4.0),     # scores of +-40: the running maximum moves block after block
])
def test_streaming_attention_matches_oracle(case):
    """head dim 32 -> attention_x6.hip (K and V of a head no longer have to fit LDS: T = 2400 is 600 KB of them)."""
    name, dim, heads, n, T, gain = case
    model = _attention_graph(dim, heads, seed=len(name), gain=gain)
    x = np.random.default_rng(n + T).standard_normal((n, T, dim)).astype(np.float32)
    api.prof_enable(True); api.prof_reset()
    _check(model, x)
    snap = {e["name"]: e["launches"] for e in api.prof_snapshot()}
    api.prof_enable(False)
    assert snap.get("attention_x6", 0) >= 1, snap


@pytest.mark.parametrize("form", ["(x * (1 + erf)) * 0.5", "(x * 0.5) * (1 + erf)", "x * 0.70711 inside"])
def test_decomposed_gelu_is_one_epilogue(form):
    """Pass 2b: Div / Erf / Add / Mul / Mul behind a Linear -> the Linear's activation; same numbers as the op-by-op oracle."""
    g = GraphBuilder("gelu")
    rng = np.random.default_rng(len(form))
    c = 64
    g.add_input("x", ["N", "T", c])
    f32 = np.float32
    y = g.op("Add", [g.op("MatMul", ["x", g.init((rng.standard_normal((c, 2 * c)) / 8).astype(f32))]), g.init((0.1 * rng.standard_normal(2 * c)).astype(f32))])
    pre = g.op("Mul", [y, g.init(np.array(0.70710678, f32))]) if "0.70711" in form else g.op("Div", [y, g.init(np.array(np.sqrt(2.0), f32))])
    t = g.op("Add", [g.op("Erf", [pre]), g.init(np.array(1.0, f32))])
    if form.startswith("(x * 0.5)"):
        o = g.op("Mul", [g.op("Mul", [y, g.init(np.array(0.5, f32))]), t])
    else:
        o = g.op("Mul", [g.op("Mul", [y, t]), g.init(np.array(0.5, f32))])
    o = g.op("Add", [g.op("MatMul", [o, g.init((rng.standard_normal((2 * c, c)) / 11).astype(f32))]), g.init((0.1 * rng.standard_normal(c)).astype(f32))])
    g.add_output(o, ["N", "T", c])
    m = g.model()
    x = rng.standard_normal((2, 50, c)).astype(f32)
    _check(m, x)
    assert api.OrtInfer(m).cost((2, 50, c))[2] <= 3       # two Linears (+ output copy); seven launches op by op


def test_hgnet_detector_graph_matches_oracle():
    """The narrow twin of the C3 detector (same topology: 2x2 stem branch with bottom / right padding, MaxPool, HG blocks with 7-way concats,
    light blocks, ESE gates, LK-PAN 9x9 convolutions, bottom-up path) on two page shapes."""
    det, info = models.build_det("hgnet_small", seed=0)
    for seed, hw in ((1, (160, 224)), (2, (96, 320))):
        x, _ = R.det_preprocess(pages.make_page(seed, hw, lines=3))
        (name, g), = _check(det, x[None])[0]
        assert g.shape == (1, 1) + hw and g.min() >= 0.0 and g.max() <= 1.0
    _check(det, np.random.default_rng(0).standard_normal((3, 3, 64, 96)).astype(np.float32))


def test_svtrv2_recognizer_graph_matches_oracle():
    """The narrow twin of the C3 recognizer: conv stem with GELU, grouped 5x5 mixing blocks, sub-sampling convolutions, global attention
    (head dim 16: the LDS-resident kernel) -- at three widths, T = W / 4."""
    rec, _ = models.build_rec("svtrv2_small", vocab=97, seed=1)
    for ws in ((320, 200, 96), (640,)):
        xr = R.rec_preprocess([pages.make_crop(i, w, 48) for i, w in enumerate(ws)])
        (name, g), = _check(rec, xr)[0]
        assert g.shape == (len(ws), xr.shape[3] // 4, 97)
        ref = onnx_ref.run(rec, {"x": xr})[0]
        assert np.array_equal(g.argmax(-1), ref.argmax(-1))


def test_server_graphs_full_width_on_a_small_page():
    """The 21.7 M / 20.5 M parameter graphs themselves (bench.py --config 2) through OAROCR::predict on one 320 x 480 page and on a page
    with a 1000-pixel line (Wt = 1600 -> 2400 tokens in stage 2): boxes bit-exact, texts equal, scores within 1e-3 of the oracle."""
    det, di = models.build_det("server_hgnet", seed=0)
    rec, ri = models.build_rec("svtrv2", vocab=6625, seed=1)
    assert 21.0e6 < di["params"] < 22.5e6 and 20.0e6 < ri["params"] < 21.5e6
    chars = api.read_dict(models.synth_dict(6623))
    imgs = [pages.make_page(0, (320, 480), lines=6), pages.make_page(3, (256, 1280), lines=4)]
    cfg = api.TextDetectionConfig(score_threshold=0.3, box_threshold=0.6, unclip_ratio=1.5, limit_side_len=1280)
    ocr = api.OAROCRBuilder(det, rec, chars).text_detection_config(cfg).image_batch_size(2).region_batch_size(8).build()
    got = ocr.predict(imgs)
    oracle = pipeline_ref.OracleOCR(det, rec, chars, thresh=0.3, box_thresh=0.6, unclip=1.5, region_batch_size=8, limit_side_len=1280)
    ref = oracle.predict(imgs)
    for g, r in zip(got, ref):
        rep = pipeline_ref.compare_results(g, r)
        assert rep["ok"] and rep["n_regions"][0] >= 3 and rep["text_equal"] >= rep["n_regions"][0] - 1, rep
    ocr.close()


@pytest.mark.parametrize("case", [("head dim 64, K and V past LDS", 192, 3, 2, 650, 1.0), ("head dim 16, 1892 tokens", 176, 11, 1, 1892, 3.0), ("head dim 40 (padded to 64)", 80, 2, 1, 700, 1.0),
                                  ("head dim 64, fits LDS (whole-head kernel)", 128, 2, 2, 300, 1.0)])
def test_attention_of_any_head_dim_runs_at_any_length(case):
    """Round 6 (found by tools/op_fuzz.py): a fused Attention whose head dim is not the streaming bf16x6 kernel's 32 and whose K / V do not fit LDS used to be REFUSED
    at plan time; it now runs on attention_stream_kernel (key blocks of 128 rows through LDS, online soft-max).  Against torch-CPU, 2e-4."""
    name, dim, heads, n, T, gain = case
    model = _attention_graph(dim, heads, seed=len(name), gain=gain)
    x = np.random.default_rng(n + T).standard_normal((n, T, dim)).astype(np.float32)
    api.prof_enable(True); api.prof_reset()
    _check(model, x)
    snap = {e["name"]: e["launches"] for e in api.prof_snapshot()}
    api.prof_enable(False)
    assert snap.get("attention", 0) >= 1 and not snap.get("attention_x6", 0), snap


@pytest.mark.parametrize("case", [("svtrv2 stage 1", 128, 4, 8, 12, 704, False), ("svtrv2 stage 2 + residual", 256, 8, 6, 6, 2000, True), ("few tiles: output-stationary", 128, 4, 2, 12, 1400, False), ("10 rows: 12-row tiles hang over", 64, 2, 9, 10, 530, True)])
def test_grouped_mixing_conv_runs_per_group_on_the_matrix_pipe(case):
    """SVTRv2's local mixing: 5 x 5 convolution, 32 channels per group.  Each group is one implicit GEMM (K = 800, N = 32) on the
    output-stationary bf16x6 kernel reading its channels out of the full tensor (ConvP::x_ld); the direct kernel it replaces ran at 6.7 TFLOP/s."""
    name, c, groups, n, h, w, with_res = case
    g = GraphBuilder("gconv")
    rng = np.random.default_rng(len(name))
    g.add_input("x", ["N", c, "H", "W"])
    wt = (rng.standard_normal((c, c // groups, 5, 5)) * np.sqrt(1.0 / (25 * c // groups))).astype(np.float32)
    y = g.op("Conv", ["x", g.init(wt), g.init((0.1 * rng.standard_normal(c)).astype(np.float32))], kernel_shape=[5, 5], strides=[1, 1], pads=[2, 2, 2, 2], group=groups, dilations=[1, 1])
    if with_res:
        y = g.op("Add", [y, "x"])
    g.add_output(y, ["N", c, "H", "W"])
    m = g.model()
    x = rng.standard_normal((n, c, h, w)).astype(np.float32)
    api.prof_enable(True); api.prof_reset()
    _check(m, x)
    snap = {e["name"]: e["launches"] for e in api.prof_snapshot()}
    api.prof_enable(False)
    # (round 6, later: groups of 32 channels run on the LDS-tiled kernel of igemm_lk_x6.hip when the launch has enough tiles, else on the output-stationary one)
    assert snap.get("conv_lk_x6", 0) + snap.get("conv_igemm_os_x6", 0) == groups and not snap.get("conv_direct", 0), snap


@pytest.mark.parametrize("case", [("lk-pan 256 -> 64", 256, 64, 4, 80, 128, True), ("lk-pan 64 -> 64, ragged tiles", 64, 64, 8, 45, 70, False), ("48 couts, ragged", 32, 48, 6, 60, 70, False)])
def test_large_kernel_conv_from_an_lds_staged_halo_tile(case):
    """igemm_lk_x6.hip: the 9 x 9 convolutions of the LK-PAN neck -- halo tile split once into bf16 planes in LDS, taps as address offsets -- against
    torch-CPU conv2d; tiles hanging over the right / bottom edge, a cout count that is not a multiple of 64, bias + activation in the epilogue."""
    name, cin, cout, n, h, w, with_bias = case
    g = GraphBuilder("lk")
    rng = np.random.default_rng(len(name))
    g.add_input("x", ["N", cin, "H", "W"])
    wt = (rng.standard_normal((cout, cin, 9, 9)) * np.sqrt(1.0 / (81 * cin))).astype(np.float32)
    ins = ["x", g.init(wt)] + ([g.init((0.2 * rng.standard_normal(cout)).astype(np.float32))] if with_bias else [])
    y = g.op("Conv", ins, kernel_shape=[9, 9], strides=[1, 1], pads=[4, 4, 4, 4], group=1, dilations=[1, 1])
    if with_bias:
        y = g.op("Relu", [y])
    g.add_output(y, ["N", cout, "H", "W"])
    m = g.model()
    x = rng.standard_normal((n, cin, h, w)).astype(np.float32)
    api.prof_enable(True); api.prof_reset()
    _check(m, x)
    snap = {e["name"]: e["launches"] for e in api.prof_snapshot()}
    api.prof_enable(False)
    assert snap.get("conv_lk_x6", 0) == 1, snap


def test_aggregation_conv_reads_a_concat_that_is_never_written():
    """PP-HGNetV2's block: seven maps concatenated, then a 1x1 convolution.  The concat is deferred and the output-stationary bf16x6 kernel reads its K dimension
    source by source (ConvP::msrc) -- no gather / copy launch; a second graph whose concat has TWO readers must still materialise it.  Both against torch-CPU."""
    rng = np.random.default_rng(5)

    def graph(two_readers):
        g = GraphBuilder("agg")
        g.add_input("x", ["N", 8, "H", "W"])
        w0 = (rng.standard_normal((48, 8, 3, 3)) * np.sqrt(2.0 / 72)).astype(np.float32)
        t = g.op("Relu", [g.op("Conv", ["x", g.init(w0)], kernel_shape=[3, 3], strides=[1, 1], pads=[1, 1, 1, 1], group=1, dilations=[1, 1])])   # (a channels-last producer, like the stage in front of a block)
        outs = [t]
        for i in range(6):
            w = (rng.standard_normal((40, 48 if i == 0 else 40, 3, 3)) * np.sqrt(2.0 / (9 * 48))).astype(np.float32)
            t = g.op("Relu", [g.op("Conv", [t, g.init(w)], kernel_shape=[3, 3], strides=[1, 1], pads=[1, 1, 1, 1], group=1, dilations=[1, 1])])
            outs.append(t)
        cat = g.op("Concat", outs, axis=1)                       # 48 + 6 * 40 = 288 channels
        wa = (rng.standard_normal((64, 288, 1, 1)) * np.sqrt(2.0 / 288)).astype(np.float32)
        y = g.op("Relu", [g.op("Conv", [cat, g.init(wa), g.init((0.1 * rng.standard_normal(64)).astype(np.float32))], kernel_shape=[1, 1], strides=[1, 1], pads=[0, 0, 0, 0], group=1, dilations=[1, 1])])
        if two_readers:
            z = g.op("GlobalAveragePool", [cat])
            g.add_output(z, ["N", 288, 1, 1])
        g.add_output(y, ["N", 64, "H", "W"])
        return g.model()
    x = rng.standard_normal((2, 8, 160, 240)).astype(np.float32)      # 76 800 pixels: the output-stationary kernel's launch is large enough
    for two in (False, True):
        m = graph(two)
        api.prof_enable(True); api.prof_reset()
        _check(m, x)
        snap = {e["name"]: e["launches"] for e in api.prof_snapshot()}
        api.prof_enable(False)
        copies = snap.get("resize", 0) + snap.get("copy2d", 0)
        assert (copies >= 1) if two else (copies == 0 and snap.get("conv_igemm_os_x6", 0) >= 1), snap
