"""Pins the oracle (oracle/oar_oracle.c) against the known-answer vectors of the reference's own
inline tests.  Inputs are the closed-form generators written in those
tests; expected values are the constants the reference asserts."""
import math

import numpy as np
import pytest

from oracle import cpu_ref as R


def make_rgb(w, h):
    n = w * h * 3
    return ((np.arange(n) * 37 + 11) % 256).astype(np.uint8).reshape(h, w, 3)


# K1 processors/simd.rs:356-372
@pytest.mark.parametrize("src", [(0, 1, 2), (2, 1, 0)])
def test_k1_chw_normalize(src):
    w, h = 37, 19
    rgb = make_rgb(w, h)
    alpha = np.array([1.0 / 255.0, 0.5, 2.0], np.float32)
    beta = np.array([-0.485, 0.1, -1.0], np.float32)
    got = R.normalize(rgb, alpha, beta, src, "chw")
    flat = rgb.reshape(-1, 3)
    for c in range(3):
        exp = flat[:, src[c]].astype(np.float32) * alpha[c] + beta[c]  # numpy f32: mul then add
        assert np.array_equal(got[c].reshape(-1), exp)


# K2 processors/simd.rs:374-387
@pytest.mark.parametrize("src", [(0, 1, 2), (2, 1, 0)])
def test_k2_hwc_normalize(src):
    w, h = 23, 7
    rgb = make_rgb(w, h)
    alpha = np.array([1.0 / 255.0, 0.5, 2.0], np.float32)
    beta = np.array([-0.485, 0.1, -1.0], np.float32)
    got = R.normalize(rgb, alpha, beta, src, "hwc")
    exp = rgb[..., list(src)].astype(np.float32) * alpha + beta
    assert np.array_equal(got, exp)


# K3 processors/simd.rs:389-403
def test_k3_argmax_last_index_wins():
    row = np.array([((i * 17) % 13) * 0.5 for i in range(101)], np.float32)
    idx, p = R.argmax_rows(row[None, :])
    m = row.max()
    assert p[0] == m and idx[0] == max(i for i in range(101) if row[i] == m)
    idx, p = R.argmax_rows(np.array([[1, 2, 5, 9, 4, 8, 9, 0]], np.float32))
    assert (idx[0], p[0]) == (6, 9.0)
    idx, p = R.argmax_rows(np.array([[42.0]], np.float32))
    assert (idx[0], p[0]) == (0, 42.0)


# K4 processors/simd.rs:405-429
def test_k4_crnn_normalize_padding():
    rw, ih, tw = 21, 32, 40
    rgb = make_rgb(rw, ih)
    got = R.crnn_normalize(rgb, tw)
    exp = np.zeros((3, ih, tw), np.float32)
    for c in range(3):
        v = rgb[:, :, 2 - c].astype(np.float32)
        exp[c, :, :rw] = (v / np.float32(255.0) - np.float32(0.5)) / np.float32(0.5)
    assert np.array_equal(got, exp)
    assert np.all(got[:, :, rw:] == 0.0)


# K6 processors/normalization.rs:498-555
def test_k6_color_order_and_mean_std_in_output_order():
    px = np.array([[[10, 20, 30]]], np.uint8)
    a, b = R.alpha_beta(1.0, [0, 0, 0], [1, 1, 1])
    assert R.normalize(px, a, b, (0, 1, 2)).reshape(-1).tolist() == [10, 20, 30]
    assert R.normalize(px, a, b, (2, 1, 0)).reshape(-1).tolist() == [30, 20, 10]
    px = np.array([[[11, 22, 33]]], np.uint8)
    a, b = R.alpha_beta(1.0, [1, 2, 3], [2, 4, 5])
    assert R.normalize(px, a, b, (0, 1, 2)).reshape(-1).tolist() == [5, 5, 6]
    a, b = R.alpha_beta(1.0, [3, 2, 1], [5, 4, 2])
    assert R.normalize(px, a, b, (2, 1, 0)).reshape(-1).tolist() == [6, 5, 5]


# K7 processors/normalization.rs:630-683
def test_k7_layout_element_order():
    imgs = []
    for off in (0, 20):
        im = np.zeros((2, 2, 3), np.uint8)
        for y in range(2):
            for x in range(2):
                base = (2 * y + x) * 3 + 1 + off
                im[y, x] = [base, base + 1, base + 2]
        imgs.append(im)
    a, b = R.alpha_beta(1.0, [0, 0, 0], [1, 1, 1])
    chw = np.stack([R.normalize(i, a, b, (0, 1, 2), "chw") for i in imgs])
    assert chw.shape == (2, 3, 2, 2)
    assert chw.reshape(-1).tolist() == [1, 4, 7, 10, 2, 5, 8, 11, 3, 6, 9, 12,
                                        21, 24, 27, 30, 22, 25, 28, 31, 23, 26, 29, 32]
    hwc = np.stack([R.normalize(i, a, b, (0, 1, 2), "hwc") for i in imgs])
    assert hwc.reshape(-1).tolist() == list(range(1, 13)) + list(range(21, 33))


# K8 processors/normalization.rs:685-709 (values follow K1's formula with the real DB constants)
def test_k8_db_constants():
    w, h = 96, 64
    ys, xs = np.mgrid[0:h, 0:w]
    A = np.stack([xs % 251, ys % 241, (xs + ys) % 239], -1).astype(np.uint8)
    got = R.db_normalize(A)
    scale = np.float32(1.0) / np.float32(255.0)
    mean = np.array(R.DB_MEAN, np.float32)
    std = np.array(R.DB_STD, np.float32)
    alpha = scale / std
    beta = -mean / std
    for c in range(3):
        exp = A[:, :, 2 - c].astype(np.float32) * alpha[c] + beta[c]
        assert np.array_equal(got[c], exp)


# K9 processors/decode.rs:679-745
def test_k9_ctc_decode():
    winners = [[(0, .9), (1, .8), (1, .7), (0, .6), (1, .5), (2, .4), (2, .3)],
               [(3, .95), (3, .85), (4, .75), (3, .65), (0, .55), (2, .45), (0, .35)]]
    logits = np.full((2, 7, 5), -10.0, np.float32)
    for b, seq in enumerate(winners):
        for t, (i, p) in enumerate(seq):
            logits[b, t, i] = p
    idx, prob = R.argmax_rows(logits)
    charset = R.ctc_charset(["a", "b", "c"], use_space_char=False)
    assert charset == ["\0", "a", "b", "c"]
    texts, scores, pos, cols, lens = R.ctc_decode(idx, prob, 2, 7, charset)
    assert texts == ["aab", "ccb"]
    f = np.float32
    assert scores[0] == float((f(.8) + f(.5) + f(.4)) / f(3))
    assert scores[1] == float((f(.95) + f(.65) + f(.45)) / f(3))
    assert cols == [[1, 4, 5], [0, 3, 5]]
    assert pos[0] == [float(f(c) / f(7)) for c in (1, 4, 5)]
    assert lens == [7, 7]


# K10 processors/decode.rs:747-757
def test_k10_ctc_empty():
    texts, scores, pos, cols, lens = R.ctc_decode(np.zeros(0), np.zeros(0), 2, 0, ["\0", "a"])
    assert texts == [] and scores == [] and lens == []


# K11 processors/db_bitmap.rs:375-390
def test_k11_minibox_point_order():
    mb, _ = R.mini_box(np.array([[20, 20], [10, 10], [20, 10], [10, 20]], np.float32))
    assert np.allclose(mb, [[10, 10], [20, 10], [20, 20], [10, 20]], atol=1e-4)


# K12 processors/db_bitmap.rs:392-406
def test_k12_min_side():
    _, ms = R.mini_box(np.array([[0, 0], [10, 0], [10, 5], [0, 5]], np.float32))
    assert abs(ms - 5.0) < 1e-3


# K13 processors/db_bitmap.rs:408-423
def test_k13_simplify_chain():
    pts = np.array([[0, 0], [1, 0], [2, 0], [2, 1], [2, 2], [1, 2], [0, 2], [0, 1]], np.float32)
    assert R.simplify_chain(pts).shape[0] == 4


# K16 utils/transform.rs:579-608 -- bicubic against the straightforward 16-tap formula (:543-577)
def test_k16_bicubic_matches_reference_formula():
    w, h = 17, 11
    i = np.arange(w * h)
    img = np.stack([(37 * i + 11) % 256, (59 * i + 7) % 256, (101 * i + 3) % 256], -1).astype(np.uint8).reshape(h, w, 3)

    def ck(t):
        A = np.float32(-0.5)
        a = np.float32(abs(t))
        if a <= 1:
            return (A + np.float32(2)) * a * a * a - (A + np.float32(3)) * a * a + np.float32(1)
        if a < 2:
            return A * a * a * a - np.float32(5) * A * a * a + np.float32(8) * A * a - np.float32(4) * A
        return np.float32(0)

    def ref(x, y):
        x, y = np.float32(x), np.float32(y)
        xi, yi = int(math.floor(x)), int(math.floor(y))
        dx, dy = x - np.float32(xi), y - np.float32(yi)
        res = np.zeros(3, np.float32)
        for j in range(-1, 3):
            for k in range(-1, 3):
                px = min(max(xi + k, 0), w - 1)
                py = min(max(yi + j, 0), h - 1)
                wt = np.float32(ck(dx - np.float32(k)) * ck(dy - np.float32(j)))
                res = res + wt * img[py, px].astype(np.float32)
        r = np.where(res >= 0, np.floor(res + np.float32(0.5)), np.ceil(res - np.float32(0.5)))
        return np.clip(r, 0, 255).astype(np.uint8)

    for yi in range(-3, 14):
        for fy in (0.0, 0.33, 0.66):
            for xi in range(-3, 20, 2):
                for fx in (0.0, 0.25, 0.5, 0.75):
                    x, y = xi + fx, yi + fy
                    assert np.array_equal(R.bicubic_sample(img, x, y), ref(x, y)), (x, y)


# K17 utils/transform.rs:699-716 via the crop planner: strict axis-aligned predicate
def test_k17_axis_aligned_fast_path():
    img = np.zeros((40, 60, 3), np.uint8)
    plan_mode = lambda b: R.lib().orc_crop_plan(60, 40, R._p(np.asarray(b, np.float32)), R._p(np.zeros(7, np.int32)), R._p(np.zeros(9, np.float32)))
    assert plan_mode([[0, 0], [50, 0], [50, 30], [0, 30]]) == 1
    assert plan_mode([[0, 0], [50, 0.001], [50, 30], [0, 30]]) == 2
    assert plan_mode([[0.5, 0], [50, 0], [50, 30], [0, 30]]) == 2


# K18 src/oarocr/processors.rs:283-302
def test_k18_sixteen_axis_aligned_crops():
    img = np.zeros((4, 64, 3), np.uint8)
    img[:, :, 0] = (np.arange(64) // 4)[None, :]
    for i in range(16):
        x1, x2 = 4 * i, 4 * i + 4
        crop = R.rotate_crop(img, np.array([[x1, 0], [x2, 0], [x2, 4], [x1, 4]], np.float32))
        assert crop.shape == (4, 4, 3)
        assert crop[0, 0].tolist() == [i, 0, 0]


# K19 processors/sorting.rs:740-783 (exact vectors of the reference tests)
def test_k19_sort_quad_boxes():
    def bc(x1, y1, x2, y2):
        return [x1, y1, x2, y1, x2, y2, x1, y2]
    boxes = np.array([bc(10, 50, 50, 70), bc(10, 10, 50, 30), bc(10, 30, 50, 50)], np.float32)
    assert R.sort_quad_boxes(boxes).tolist() == [1, 2, 0]
    boxes = np.array([bc(60, 10, 100, 30), bc(10, 12, 50, 32)], np.float32)   # same line: left first
    assert R.sort_quad_boxes(boxes).tolist() == [1, 0]
    boxes = np.array([bc(60, 10, 100, 30), bc(10, 11, 50, 31), bc(10, 50, 50, 70), bc(60, 52, 100, 72)], np.float32)
    assert R.sort_quad_boxes(boxes).tolist() == [1, 0, 2, 3]
    assert R.sort_quad_boxes(np.zeros((0, 8), np.float32)).tolist() == []


# K21 src/oarocr/builder_utils.rs:132-175
def test_k21_batch_policy():
    assert R.resolve_device_batch_sizes(None, None, False, None) == (1, 4)
    assert R.resolve_device_batch_sizes(3, 7, False, None) == (3, 7)
    assert R.resolve_device_batch_sizes(None, None, True, None) == (None, None)
    assert R.default_cpu_region_batch_size("pp-ocrv6_tiny_rec") == 16
    assert R.default_cpu_region_batch_size("pp-ocrv6_small_rec") == 4
    assert R.default_cpu_region_batch_size(None) == 4
    assert R.default_cpu_region_batch_size("PP-OCRv6_tiny_rec") == 16
