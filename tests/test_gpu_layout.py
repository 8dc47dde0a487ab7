"""Layout detection on the GPU through the C ABI against the oracle: filtered resize and LayoutPostProcess bit-exact,
the preprocessed tensor bit-exact, the whole adapter on a PicoDet-shaped and a PP-DocLayoutV2-shaped graph."""
import numpy as np
import pytest

from oar_ocr_amd import api
from oar_ocr_amd.synth import models, pages
from oracle import cpu_ref as R
from oracle import pipeline_ref

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("filt", ["triangle", "catmullrom", "lanczos3"])
def test_resize_filter_kernel_bit_exact(filt):
    rng = np.random.default_rng(5)
    for (w, h, nw, nh) in [(130, 90, 64, 48), (64, 48, 200, 150), (960, 700, 608, 800), (33, 200, 32, 32), (5, 7, 40, 3)]:
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        assert np.array_equal(api.k_resize_filter(a, nw, nh, filt), R.resize_filter(a, nw, nh, filt)), (w, h, nw, nh)


def _random_predictions(rng, n, rows, feat, num_classes, fmt, nan_ok=True):
    if feat < 6:
        return rng.uniform(0, 1, (n, rows, feat)).astype(np.float32)      # under-width rows: the processors return nothing
    p = np.zeros((n, rows, feat), np.float32)
    x1, y1 = rng.uniform(0, 500, (n, rows)), rng.uniform(0, 700, (n, rows))
    bw, bh = rng.uniform(-5, 250, (n, rows)), rng.uniform(-5, 250, (n, rows))
    cls = rng.integers(-1, num_classes + 7, (n, rows)).astype(np.float32) + rng.choice([0.0, 0.3, -0.3], (n, rows))
    sc = np.round(rng.uniform(-0.1, 1.1, (n, rows)), 2).astype(np.float32)       # rounded: plenty of exact ties
    box = np.stack([x1, y1, x1 + bw, y1 + bh], -1)
    norm = rng.random((n, rows)) < 0.2
    box[norm] = rng.uniform(-0.04, 1.04, (int(norm.sum()), 4))
    box[norm, 2:] = np.maximum(box[norm, 2:], box[norm, :2] + 0.01)
    if fmt == "scores":            # [x1 y1 x2 y2 scores...]
        p[..., :4] = box
        p[..., 4:] = rng.uniform(-0.2, 1.0, (n, rows, feat - 4))
    else:
        order = {"csb": (0, 1, 2), "bsc": (5, 4, 0), "scb": (1, 0, 2)}[fmt]
        p[..., order[0]] = cls
        p[..., order[1]] = sc
        p[..., order[2]:order[2] + 4] = box
        if feat > 6:
            p[..., 6:] = rng.integers(0, 4, (n, rows, feat - 6))
    bad = rng.random((n, rows)) < 0.03
    # pp-doclayout rows are not validated before the sort (layout_postprocess.rs:277-283): a NaN score there makes the reference's own
    # sort_by comparator non-total (unspecified order), so that mode gets infinities only
    p[bad, 1] = rng.choice([np.nan, np.inf, -np.inf] if nan_ok else [np.inf, -np.inf], int(bad.sum()))
    return p


@pytest.mark.parametrize("model_type,feat,fmt", [("picodet", 9, "scores"), ("picodet", 6, "csb"), ("picodet", 6, "bsc"), ("picodet", 7, "scb"), ("rtdetr", 6, "csb"),
                                                 ("pp-doclayout", 6, "csb"), ("pp-doclayout", 7, "csb"), ("pp-doclayout", 8, "csb"), ("pp-doclayout", 4, "csb")])
def test_layout_postprocess_kernel_equals_the_oracle(model_type, feat, fmt):
    rng = np.random.default_rng(hash((model_type, feat, fmt)) % (1 << 31))
    for (n, rows, max_det, nms) in [(3, 300, 100, 0.5), (1, 1, 100, 0.5), (2, 900, 40, 0.3), (2, 17, 5, 0.9)]:
        pred = _random_predictions(rng, n, rows, feat, 5, fmt, nan_ok=model_type != "pp-doclayout")
        wh = np.stack([rng.integers(300, 700, n), rng.integers(300, 900, n)], -1).astype(np.float32)
        got = api.k_layout_postprocess(pred, wh, 5, 0.5, nms, max_det, model_type)
        kept = 0
        for i in range(n):
            rb, rc, rs = R.layout_postprocess(pred[i], wh[i, 0], wh[i, 1], 5, 0.5, nms, max_det, model_type)
            gb, gc, gs = got[i]
            assert np.array_equal(gc, rc) and np.array_equal(gs, rs, equal_nan=True) and np.array_equal(gb, rb), (model_type, feat, fmt, n, rows, i)
            kept += len(rb)
        if rows >= 300 and feat >= 6:
            assert kept > 10


def test_preprocessed_tensor_bit_exact_both_families():
    page = pages.make_page(7, (700, 520), 14)
    tiny = np.random.default_rng(1).integers(0, 256, (20, 30, 3), dtype=np.uint8)      # h + w < 64: padded to 32 x 32 first
    for kind, shape in (("picodet", (320, 256)), ("pp-doclayout", (256, 256))):
        m, _ = models.build_layout(kind, image_shape=shape)
        mc = api.LayoutModelConfig(f"synthetic_{kind}", 5, {k: f"c{k}" for k in range(5)}, kind, shape)
        pred = api.LayoutDetectionPredictor(m, mc)
        orc = pipeline_ref.OracleLayoutDetector(m, 5, kind, shape)
        for im in (page, tiny, page[:shape[0], :shape[1]]):
            assert np.array_equal(pred.preprocess(im), orc.preprocess(im)[0]), (kind, im.shape)
        pred.close()


@pytest.mark.parametrize("kind,shape", [("picodet", (320, 256)), ("pp-doclayout", (256, 256))])
def test_layout_adapter_matches_oracle(kind, shape):
    """The network's prediction rows agree with the torch-CPU oracle within 1e-3 (scores) / 5e-2 px (boxes); LayoutPostProcess applied by
    the oracle to THE SAME rows (taken from the HIP engine through Seam A) equals the adapter's output exactly."""
    m, info = models.build_layout(kind, image_shape=shape)
    imgs = [pages.make_page(20 + i, (480 + 40 * i, 360), 10 + i) for i in range(3)]
    mc = api.LayoutModelConfig(f"synthetic_{kind}", 5, {k: f"c{k}" for k in range(5)}, kind, shape)
    thr = 0.5 if kind == "picodet" else 0.18    # the synthetic PP-DocLayout head sees un-normalised RGB: scores bunch around 0.18
    cfg = api.LayoutDetectionConfig(score_threshold=thr, max_elements=60, nms_threshold=0.5)
    pred = api.LayoutDetectionPredictor(m, mc, cfg)
    raw, feat = pred.detect_raw(imgs)
    assert feat == info["feat"]
    orc = pipeline_ref.OracleLayoutDetector(m, 5, kind, shape, thr, 0.5, 60)
    ref, y_ref = orc.detect(imgs)
    # the same graph through Seam A on the same preprocessed tensors: the rows the adapter's kernels saw
    eng = api.OrtInfer(m)
    x = np.stack([orc.preprocess(im)[0] for im in imgs])
    feeds = [("image", x), ("scale_factor", np.array([[np.float32(shape[0]) / np.float32(im.shape[0]), np.float32(shape[1]) / np.float32(im.shape[1])] for im in imgs], np.float32))]
    if kind == "pp-doclayout":
        feeds.append(("im_shape", np.array([[shape[0], shape[1]]] * len(imgs), np.float32)))
    y = eng.infer(feeds)[0][1].reshape(len(imgs), -1, feat)
    assert np.abs(y[..., 1] - y_ref[..., 1]).max() <= 1e-3 and np.abs(y[..., 2:6] - y_ref[..., 2:6]).max() <= 5e-2
    assert np.array_equal(y[..., 0], y_ref[..., 0]) or np.mean(y[..., 0] != y_ref[..., 0]) < 0.01      # argmax ties only
    total = 0
    for i, im in enumerate(imgs):
        rb, rc, rs = R.layout_postprocess(y[i], im.shape[1], im.shape[0], 5, thr, 0.5, 60, kind)
        gb, gc, gs = raw[i]
        assert np.array_equal(gc, rc) and np.array_equal(gs, rs) and np.array_equal(gb, rb), (kind, i)
        total += len(rb)
        # and against the all-oracle run: same detections unless a score / IoU sits within the network tolerance of a threshold
        if len(rb) and len(ref[i][0]) == len(rb):
            assert np.abs(ref[i][0] - rb).max() <= 5e-2 and np.array_equal(ref[i][1], rc)
    assert total > (20 if kind == "picodet" else 0), total
    els = pred.predict(imgs)
    if kind == "pp-doclayout":
        # model_type "pp-doclayout" takes the adapter's own post-processing (postprocess_pp_doclayout, layout_detection_adapter.rs:631-846) instead of
        # LayoutPostProcess: the elements are the oracle's restatement of it applied to the same rows
        for i, im in enumerate(imgs):
            rb, rc, rs = R.pp_doclayout_postprocess(y[i], im.shape[1], im.shape[0], 5, thr)
            assert len(els[i]) == min(len(rb), 60)
            for e, b, c, sc in zip(els[i], rb, rc, rs):
                assert e.element_type == f"c{c}" and e.score == float(sc) and np.array_equal(e.bbox[0], b[:2]) and np.array_equal(e.bbox[2], b[2:])
    else:
        assert [len(e) for e in els] == [len(r[0]) for r in raw]
    assert all(e.element_type.startswith("c") for p_ in els for e in p_)
    assert pred.is_reading_order_sorted == (feat in (7, 8))
    pred.close()
    eng.close()


@pytest.mark.parametrize("feat", [6, 7, 8])
def test_pp_doclayout_adapter_postprocess_kernel_equals_the_oracle(feat):
    """LayoutDetectionAdapter::postprocess_pp_doclayout as one HIP kernel (layout.hip ppdoc_post_kernel) against the oracle's restatement on identical
    prediction rows: per-class thresholds, paddlex_layout_nms (the '+ 1' IoU, 0.6 same class / 0.98 across), filter_large_image_boxes,
    apply_paddlex_merge_modes (Large / Small / Union, the formula rule), reading-order sort by total_cmp -- with score ties, page-sized image boxes,
    normalised coordinates, nested boxes, infinite scores, every switch on and off."""
    rng = np.random.default_rng(90 + feat)
    labels = {0: "text", 1: "image", 2: "formula", 3: "table", 4: "chart", 5: "title"}
    nc = len(labels)
    for trial, (n, rows) in enumerate([(3, 300), (2, 40), (1, 1), (2, 700), (1, 0)]):
        pred = _random_predictions(rng, n, rows, feat, nc, "csb", nan_ok=False) if rows else np.zeros((n, 0, feat), np.float32)
        wh = np.stack([rng.integers(300, 700, n), rng.integers(300, 900, n)], -1).astype(np.float32)
        if rows >= 40:
            for i in range(n):
                # nested boxes (containment >= 0.9) of assorted classes, and page-sized image boxes
                for k in range(0, rows // 4, 2):
                    x0, y0 = rng.uniform(0, 250), rng.uniform(0, 400)
                    w, h = rng.uniform(60, 200), rng.uniform(60, 200)
                    pred[i, k, 2:6] = [x0, y0, x0 + w, y0 + h]
                    pred[i, k + 1, 2:6] = [x0 + 0.02 * w, y0 + 0.02 * h, x0 + rng.uniform(0.5, 0.99) * w, y0 + rng.uniform(0.5, 0.99) * h]
                    pred[i, k, 0], pred[i, k + 1, 0] = rng.integers(0, nc), rng.integers(0, nc)
                    pred[i, k, 1], pred[i, k + 1, 1] = np.round(rng.uniform(0.3, 1.0, 2), 2)
                pred[i, rows // 2, :6] = [1, 0.9, 0, 0, wh[i, 0], wh[i, 1]]
                pred[i, rows // 2 + 1, :6] = [1, 0.8, 2, 2, wh[i, 0] * 0.5, wh[i, 1] * 0.5]
        variants = [
            dict(),
            dict(class_thresholds={"text": 0.4, "table": 0.75, "title": 0.0}),
            dict(layout_nms=False),
            dict(class_merge_modes={"text": "large", "image": "union", "table": "small", "chart": "large"}),
            dict(class_merge_modes={k: "small" for k in labels.values()}, class_thresholds={"formula": 0.2}),
            dict(class_merge_modes={"formula": "large", "title": "small"}, layout_nms=False),
        ]
        for kw in variants:
            thr = 0.5 if trial % 2 == 0 else 0.3
            got = api.k_ppdoc_postprocess(pred, wh, nc, labels, thr, kw.get("class_thresholds"), kw.get("layout_nms", True), kw.get("class_merge_modes"))
            by = {v: k for k, v in labels.items()}
            kept = 0
            for i in range(n):
                rb, rc, rs = R.pp_doclayout_postprocess(pred[i], wh[i, 0], wh[i, 1], nc, thr, {by[k]: v for k, v in kw.get("class_thresholds", {}).items()} or None,
                                                        kw.get("layout_nms", True), by["image"], by["formula"], {by[k]: v for k, v in kw.get("class_merge_modes", {}).items()} or None)
                gb, gc, gs = got[i]
                assert np.array_equal(gc, rc) and np.array_equal(gs, rs) and np.array_equal(gb, rb), (feat, trial, kw, i, len(gc), len(rc))
                kept += len(rb)
            if rows >= 300:
                assert kept > 10
