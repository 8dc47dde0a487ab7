"""Layout detection, CPU side: the oracle restatement of LayoutPostProcess against the reference's own inline tests
(processors/layout_postprocess.rs:886-957) and hand-derived cases, the resize filters, the adapter-level helpers of the Python mirror."""
import numpy as np

from oar_ocr_amd import api
from oracle import cpu_ref as R


def test_reference_inline_tests_of_layout_postprocess():
    # test_pp_doclayout_under_width_output_does_not_panic (:919-936)
    assert len(R.layout_postprocess(np.zeros((3, 4), np.float32), 100, 100, 17, 0.5, 0.5, 100, "pp-doclayout")[0]) == 0
    assert len(R.layout_postprocess(np.zeros((0, 8), np.float32), 100, 100, 17, 0.5, 0.5, 100, "pp-doclayout")[0]) == 0
    # test_picodet_argmax_handles_non_positive_scores (:938-957)
    b, c, s = R.layout_postprocess(np.array([[10, 10, 50, 50, -0.9, -0.2, -0.5]], np.float32), 100, 100, 3, -1.0, 1.0, 100, "picodet")
    assert len(b) == 1 and c[0] == 1 and abs(s[0] + 0.2) < 1e-6 and np.array_equal(b[0], [10, 10, 50, 50])
    # test_iou_calculation (:894-917) through the suppression itself: identical boxes (IoU 1) of one class -> one survives; disjoint -> both
    rows = np.array([[0, 0.9, 0, 0, 100, 100], [0, 0.8, 0, 0, 100, 100], [0, 0.7, 200, 200, 300, 300]], np.float32)
    b, c, s = R.layout_postprocess(rows, 400, 400, 5, 0.5, 0.5, 100, "picodet")
    assert np.allclose(s, [0.9, 0.7]) and np.array_equal(b, [[0, 0, 100, 100], [200, 200, 300, 300]])


def test_layout_postprocess_rules():
    # class-aware: the same box in another class is kept; the three column orders; normalised coordinates; threshold is `<`
    rows = np.array([[1, 0.9, 10, 10, 60, 60],        # (class, score, box)
                     [2, 0.9, 10, 10, 60, 60],        # other class: not suppressed
                     [12, 12, 58, 58, 0.85, 1],       # (box, score, class): overlaps row 0's class -> suppressed (IoU 0.846 > 0.5)
                     [0.5, 3, 100, 20, 150, 160],     # (score, class, box): score exactly at the threshold is kept; y2 clamped to the page
                     [4, 0.6, 0.1, 0.2, 0.5, 0.9],    # normalised box -> scaled by the image size
                     [1, 0.4, 0, 0, 5, 5],            # below the threshold
                     [1, 0.95, 30, 30, 30, 40]], np.float32)   # degenerate (x2 == x1)
    b, c, s = R.layout_postprocess(rows, 200, 100, 5, 0.5, 0.5, 100, "picodet")
    assert c.tolist() == [1, 2, 4, 3] and np.allclose(s, [0.9, 0.9, 0.6, 0.5])
    assert np.allclose(b[2], [20, 20, 100, 90]) and np.array_equal(b[3], [100, 20, 150, 100])   # clamped to the page (200 x 100)
    # stable order of equal scores, max_detections cap
    rows = np.array([[0, 0.7, 10 * k, 0, 10 * k + 5, 5] for k in range(6)], np.float32)
    b, c, s = R.layout_postprocess(rows, 100, 100, 5, 0.5, 0.5, 4, "picodet")
    assert np.array_equal(b[:, 0], [0, 10, 20, 30])
    # pp-doclayout 8 columns: reading order (col, row) after NMS; `as i32` truncation of the class column
    rows = np.array([[1.9, 0.9, 0, 0, 10, 10, 1, 5], [0.2, 0.8, 20, 0, 30, 10, 0, 9], [3, 0.95, 40, 0, 50, 10, 0, 2], [-1, 0.99, 60, 0, 70, 10, 0, 0]], np.float32)
    b, c, s = R.layout_postprocess(rows, 100, 100, 5, 0.5, 0.5, 100, "pp-doclayout")
    assert c.tolist() == [3, 0, 1] and np.allclose(s, [0.95, 0.8, 0.9])


def test_resize_filters_are_sane_and_exact_on_identity():
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (40, 56, 3), dtype=np.uint8)
    for f in ("triangle", "catmullrom", "lanczos3"):
        assert np.array_equal(R.resize_filter(a, 56, 40, f), a)            # same size: a copy
        flat = np.full((30, 44, 3), 137, np.uint8)
        assert np.all(R.resize_filter(flat, 91, 17, f) == 137)             # weights sum to one
    assert np.array_equal(R.resize_filter(a, 31, 23, "triangle"), R.resize_triangle(a, 31, 23))
    from PIL import Image
    g = np.clip(np.add.outer(np.arange(64) * 3, np.arange(80) * 2)[..., None] % 256 + np.zeros(3), 0, 255).astype(np.uint8)
    for f, pf in (("lanczos3", Image.LANCZOS), ("catmullrom", Image.BICUBIC)):
        d = np.abs(R.resize_filter(g, 40, 32, f).astype(int) - np.asarray(Image.fromarray(g).resize((40, 32), pf)).astype(int))
        assert d.mean() < 1.0                                                 # same filter family as PIL's (pass order / rounding differ)


def test_adapter_level_helpers():
    mc = api.LayoutModelConfig.picodet_layout_1x()
    assert mc.num_classes == 5 and mc.class_labels[3] == "table" and mc.input_size == (800, 608) and mc.preprocess()[:2] == ("lanczos3", True)
    cfg = api.LayoutDetectionConfig(class_thresholds={"text": 0.4})
    assert cfg.get_class_threshold("text") == 0.4 and cfg.get_class_threshold("table") == 0.5
    boxes = np.array([[10, 20, 30, 60], [0, 0, 10, 10]], np.float32)
    out = api.unclip_boxes(boxes, [0, 1], (1.5, 0.5))                      # layout_postprocess.rs:636-690: centre fixed
    assert np.allclose(out, [[5, 30, 35, 50], [-2.5, 2.5, 12.5, 7.5]])
    assert np.array_equal(api.unclip_boxes(boxes, [0, 1], 1.0), boxes)
    assert np.allclose(api.unclip_boxes(boxes, [0, 1], {1: (2.0, 2.0)}), [[10, 20, 30, 60], [-5, -5, 15, 15]])
