"""GPU parity tests, kernel by kernel: each HIP kernel is called through the C ABI on the same input bits as
the oracle (oracle/oar_oracle.c) and must agree BIT-EXACTLY (byte / index / f32 stages of the path)."""
import numpy as np
import pytest

from oar_ocr_amd import api
from oar_ocr_amd.synth import pages
from oracle import cpu_ref as R

pytestmark = pytest.mark.gpu


def make_rgb(w, h):
    n = w * h * 3
    return ((np.arange(n) * 37 + 11) % 256).astype(np.uint8).reshape(h, w, 3)


@pytest.mark.parametrize("src", [(0, 1, 2), (2, 1, 0)])
@pytest.mark.parametrize("layout", ["chw", "hwc"])
def test_normalize_k1_k2(src, layout):
    # processors/simd.rs:356-387 vectors
    rgb = make_rgb(37, 19)
    alpha = np.array([1.0 / 255.0, 0.5, 2.0], np.float32)
    beta = np.array([-0.485, 0.1, -1.0], np.float32)
    assert np.array_equal(api.k_normalize(rgb, alpha, beta, src, layout), R.normalize(rgb, alpha, beta, src, layout))


def test_normalize_db_constants_full_page():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (960, 960, 3), dtype=np.uint8)
    a, b = R.alpha_beta(np.float32(1.0) / np.float32(255.0), R.DB_MEAN, R.DB_STD)
    got = api.k_normalize(img, a, b, (2, 1, 0), "chw")
    assert np.array_equal(got, R.db_normalize(img))
    # odd sizes: tails of the 4-pixel vector path
    img = rng.integers(0, 256, (7, 13, 3), dtype=np.uint8)
    assert np.array_equal(api.k_normalize(img, a, b, (2, 1, 0), "hwc"), R.normalize(img, a, b, (2, 1, 0), "hwc"))


def test_threshold_strict_greater():
    rng = np.random.default_rng(1)
    p = rng.random((123, 77)).astype(np.float32)
    p[0, :5] = 0.3   # == thresh must NOT pass (db_postprocess.rs:202)
    t = np.float32(0.3)
    assert np.array_equal(api.k_threshold(p, t), R.threshold_mask(p, t))


def test_ctc_argmax_last_index_wins():
    rng = np.random.default_rng(2)
    x = rng.random((50, 6906)).astype(np.float32)
    x[3, 100] = x[3, 5000] = 2.0          # tie -> last
    x[4, 0] = x[4, 6905] = 3.0
    x[5, :] = 0.25                         # all equal -> last index
    gi, gp = api.k_ctc_argmax(x)
    ri, rp = R.argmax_rows(x)
    assert np.array_equal(gi, ri) and np.array_equal(gp, rp)
    assert gi[3] == 5000 and gi[4] == 6905 and gi[5] == 6905
    # K3 vectors (simd.rs:389-403)
    gi, gp = api.k_ctc_argmax(np.array([[1, 2, 5, 9, 4, 8, 9, 0]], np.float32))
    assert (gi[0], gp[0]) == (6, 9.0)
    gi, gp = api.k_ctc_argmax(np.array([[42.0]], np.float32))
    assert (gi[0], gp[0]) == (0, 42.0)


@pytest.mark.parametrize("size", [(100, 37, 64, 48), (320, 48, 320, 48), (91, 33, 200, 48), (640, 120, 213, 48), (33, 17, 33, 17), (1000, 410, 200, 80), (50, 20, 333, 97), (1300, 9, 100, 3)])
def test_resize_triangle(size):
    w, h, nw, nh = size
    rng = np.random.default_rng(w * 7 + h)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    assert np.array_equal(api.k_resize_triangle(img, nw, nh), R.resize_triangle(img, nw, nh))


def test_rec_preprocess_batch():
    rng = np.random.default_rng(3)
    crops = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for (w, h) in [(120, 30), (400, 41), (64, 64), (33, 20), (900, 25)]]
    got = api.k_rec_preprocess(crops)
    ref = R.rec_preprocess(crops)
    assert got.shape == ref.shape
    assert np.array_equal(got, ref)
    # K4-style: padding stays exactly 0
    tw, rws = R.rec_tensor_width([(c.shape[1], c.shape[0]) for c in crops])
    for i, rw in enumerate(rws):
        assert np.all(got[i, :, :, rw:] == 0.0)


@pytest.mark.parametrize("flip", [False, True])
def test_rec_preprocess_over_the_scale_range(flip):
    """The recognizer resize against the oracle, bit for bit, over the scale range a page can produce: 4x enlargement to 6x reduction, the identity copy, crops of a
    few pixels, widths capped at the tensor width (horizontal ratio != vertical ratio), and the rotate180 read."""
    rng = np.random.default_rng(31)
    sizes = [(96, 48), (97, 48), (31, 48), (3, 9), (200, 12), (333, 24), (500, 61), (260, 70), (410, 100), (300, 139), (640, 150), (700, 300), (5000, 40), (3300, 96), (64, 7)]
    crops = [rng.integers(0, 256, (h, w, 3), dtype=np.uint8) for (w, h) in sizes]
    flips = [flip] * len(crops)
    want = R.rec_preprocess([R.rotate_rgb(c, 2) if flip else c for c in crops])
    got = api.k_rec_preprocess(crops, flips=flips)
    assert got.shape == want.shape and np.array_equal(got, want)


def test_box_scores():
    rng = np.random.default_rng(4)
    pred = rng.random((240, 320)).astype(np.float32)
    boxes = []
    for _ in range(40):
        cx, cy = rng.uniform(20, 300), rng.uniform(20, 220)
        w, h, a = rng.uniform(5, 120), rng.uniform(4, 60), rng.uniform(-0.5, 0.5)
        c, s = np.cos(a), np.sin(a)
        pts = np.array([[-w / 2, -h / 2], [w / 2, -h / 2], [w / 2, h / 2], [-w / 2, h / 2]]) @ np.array([[c, s], [-s, c]]) + [cx, cy]
        boxes.append(pts.astype(np.float32))
    boxes.append(np.array([[-5, -5], [400, -5], [400, 300], [-5, 300]], np.float32))   # clipped, > 8000 px branch
    boxes = np.stack(boxes)
    got = api.k_box_scores(pred, boxes)
    ref = np.array([R.box_score_fast(pred, b) for b in boxes], np.float32)
    assert np.array_equal(got, ref)


def test_rotate_crop_axis_aligned_and_perspective():
    page = pages.make_page(5, (320, 480), lines=6)
    rng = np.random.default_rng(5)
    boxes = [np.array([[10, 20], [200, 20], [200, 60], [10, 60]], np.float32),            # exact axis aligned -> fast path
             np.array([[10.5, 20], [200, 22], [199, 61], [9, 58]], np.float32),           # perspective
             np.array([[300, 10], [330, 12], [328, 200], [298, 198]], np.float32),        # tall -> rotate270
             np.array([[-5, -3], [100, -4], [101, 30], [-4, 31]], np.float32),            # clipped at the border
             np.array([[470, 300], [500, 300], [500, 330], [470, 330]], np.float32)]      # partially outside
    for _ in range(20):
        cx, cy = rng.uniform(40, 440), rng.uniform(30, 290)
        w, h, a = rng.uniform(20, 200), rng.uniform(8, 50), rng.uniform(-0.08, 0.08)
        c, s = np.cos(a), np.sin(a)
        pts = np.array([[-w / 2, -h / 2], [w / 2, -h / 2], [w / 2, h / 2], [-w / 2, h / 2]]) @ np.array([[c, s], [-s, c]]) + [cx, cy]
        boxes.append(np.round(pts).astype(np.float32))
    for b in boxes:
        got, ref = api.k_rotate_crop(page, b), R.rotate_crop(page, b)
        assert (got is None) == (ref is None), b
        if ref is not None:
            assert got.shape == ref.shape, (got.shape, ref.shape, b)
            assert np.array_equal(got, ref), (b, np.abs(got.astype(int) - ref.astype(int)).max())
    # degenerate: reference errors -> dropped
    assert api.k_rotate_crop(page, np.array([[5, 5], [5, 5], [5, 5], [5, 5]], np.float32)) is None


def test_k18_sixteen_crops_order_preserved():
    # src/oarocr/processors.rs:283-302
    img = np.zeros((4, 64, 3), np.uint8)
    img[:, :, 0] = (np.arange(64) // 4)[None, :]
    for i in range(16):
        crop = api.k_rotate_crop(img, np.array([[4 * i, 0], [4 * i + 4, 0], [4 * i + 4, 4], [4 * i, 4]], np.float32))
        assert crop.shape == (4, 4, 3) and crop[0, 0].tolist() == [i, 0, 0]


# ------------------------------------------------------------------------------------------------ DBPostProcess options
def test_dilate_matches_oracle():
    rng = np.random.default_rng(3)
    for h, w in [(1, 1), (5, 7), (64, 33), (257, 129)]:
        m = (rng.random((h, w)) > 0.93).astype(np.uint8) * 255
        m[0, 0] = 255; m[-1, -1] = 255          # corners: the neighbourhood is clipped to the image
        assert np.array_equal(api.k_dilate(m), R.dilate3x3(m))
    import scipy.ndimage as ndi
    m = (rng.random((40, 50)) > 0.9).astype(np.uint8) * 255
    assert np.array_equal(api.k_dilate(m) > 0, ndi.binary_dilation(m > 0, structure=np.ones((3, 3), bool)))   # independent pin: LInf ball of radius 1


def test_poly_scores_match_oracle_on_contours_and_boxes():
    """ScoreMode::Slow's polygon is the traced contour (hundreds of points); the same kernel on 4-point boxes must equal
    box_score_fast bit for bit, including the >= 8000-pixel branch."""
    rng = np.random.default_rng(5)
    import scipy.ndimage as ndi
    pred = ndi.gaussian_filter(rng.random((200, 320)), 3).astype(np.float32)
    pred = (pred - pred.min()) / (pred.max() - pred.min())
    mask = (pred > 0.55).astype(np.uint8) * 255
    polys = [p.astype(np.float32) for p, _, _ in R.find_contours(mask)][:80]
    assert len(polys) > 5 and max(len(p) for p in polys) > 100
    boxes = [np.array([[10, 10], [300, 12], [298, 150], [8, 148]], np.float32), np.array([[50.5, 20.25], [90.75, 30.5], [80.25, 70.75], [40.5, 60.5]], np.float32),
             np.array([[0, 0], [319, 0], [319, 199], [0, 199]], np.float32), np.array([[5, 5], [5, 5], [5, 5], [5, 5]], np.float32)]
    allp = polys + boxes
    got = api.k_poly_scores(pred, allp)
    want = np.array([R.box_score_fast(pred, p) for p in allp], np.float32)
    assert np.array_equal(got, want)
    assert np.array_equal(api.k_box_scores(pred, np.stack(boxes)), want[len(polys):])


# ------------------------------------------------------------------------------------------------ a8 on the GPU (contours.hip)
def _same_contours(mask, max_contours=100000):
    want = R.find_contours(mask)[:max_contours]
    got = api.k_contours(mask, max_contours)
    assert len(got) == len(want), (mask.shape, len(got), len(want))
    for (pw, tw, _), (pg, tg) in zip(want, got):
        assert tw == tg and np.array_equal(pw, pg)
    return len(want)


def _noise_masks(n, seed, lo=4, hi=200):
    from scipy import ndimage as ndi
    rng = np.random.default_rng(seed)
    for i in range(n):
        h, w = int(rng.integers(lo, hi)), int(rng.integers(lo, hi))
        sigma = float(rng.uniform(0.4, 2.5))
        m = (ndi.gaussian_filter(rng.random((h, w)), sigma) > rng.uniform(0.42, 0.58)).astype(np.uint8) * 255
        if i % 5 == 0:
            m[:, 0] = 255 * (rng.random(h) > 0.4)        # components on the x == 0 column (imageproc's `x > 0` quirk)
        if i % 7 == 0:
            m[rng.integers(0, h)] = 255                   # a full-width line
        if i % 3 == 0:
            m[rng.random((h, w)) > 0.97] = 255            # isolated pixels and thin diagonal bridges
        if i % 4 == 0:
            m[rng.integers(0, h, 3)] = 0                  # blank rows -> several bands
        yield m


@pytest.mark.gpu
def test_gpu_contours_equal_the_oracle_chain_for_chain():
    """The wavefront border follower against the oracle's find_contours (imageproc semantics at db_bitmap.rs:100): same
    contours, same order, same points in the same tracing order, same Outer / Hole typing -- on blob noise of every density
    with thin bridges, single pixels, x == 0 components, full-width lines, odd widths (not a multiple of 16) and blank rows."""
    total = 0
    for mask in _noise_masks(120, 5):
        total += _same_contours(mask)
    assert total > 3000


@pytest.mark.gpu
def test_gpu_contours_special_shapes():
    m = np.zeros((12, 9), np.uint8)
    assert api.k_contours(m) == []                                   # blank mask: no band at all
    m[3:6] = 255
    assert api.k_contours(m) == []                                   # full-width band: no start pixel (the quirk)
    m[4, 4] = 0
    _same_contours(m)                                                # ... but the hole inside is found
    m = np.full((40, 70), 255, np.uint8)
    _same_contours(m)                                                # everything foreground
    m = np.zeros((64, 64), np.uint8); m[::2, ::2] = 255
    assert _same_contours(m) == 32 * 32                              # isolated pixels (those on x == 0 come out typed Hole)
    m = np.zeros((33, 47), np.uint8); m[np.arange(33), np.arange(33)] = 255; m[np.arange(33), 46 - np.arange(33)] = 255
    _same_contours(m)                                                # two crossing 1-pixel diagonals: pixels visited twice
    m = np.zeros((50, 50), np.uint8)
    for r in range(2, 24, 4):
        m[r:50 - r, r:50 - r] = 255 if (r // 4) % 2 == 0 else 0     # nested rings: holes inside holes
        m[r + 2:48 - r, r + 2:48 - r] = 0 if (r // 4) % 2 == 0 else 255
    _same_contours(m)
    m = np.zeros((30, 1000), np.uint8); m[5:25, 3:997] = 255; m[10:20, 100:900:7] = 0
    _same_contours(m)                                                # one wide component, many holes, > 64 px groups
    assert len(api.k_contours(np.pad(np.full((3, 3), 255, np.uint8), 1), max_contours=0)) == 0


@pytest.mark.gpu
def test_gpu_contours_text_page_and_host_fallback():
    """A 960 x 960 detector-like mask (text-line blobs in ~40 bands) and a mask whose single band is taller than the LDS
    capacity (a vertical rule joins every line): that band is flagged by the kernel and followed on the host -- same result."""
    rng = np.random.default_rng(0)
    m = np.zeros((960, 960), np.uint8)
    for y in range(12, 940, 24):
        x = 20
        while x < 900:
            w = int(rng.integers(30, 200))
            m[y:y + int(rng.integers(9, 15)), x:min(x + w, 940)] = 255
            x += w + int(rng.integers(6, 40))
    n = _same_contours(m)
    assert n > 150
    assert len(api.k_contours(m, max_contours=25)) == 25 and _same_contours(m, 25) == 25      # take(max_candidates)
    m2 = m.copy(); m2[5:955, 5:8] = 255                                                        # one band of 950 rows
    _same_contours(m2)
    holes = m.copy(); holes[::24, :] = 0; holes[14::24, 30:900:11] = 0                          # punch holes into the lines
    _same_contours(holes)
    m3 = m.copy(); m3[300:620, 10:950] = 255; m3[340:600:9, 40:900:13] = 0                     # a 320-row block: its segment exceeds the LDS classes
    _same_contours(m3)                                                                          # -> followed on the host, merged with the GPU's segments
    m4 = np.zeros((64, 3000), np.uint8); m4[8:40, 4:2990:3] = 255; m4[20, 4:2990] = 255          # ~1000 column segments joined by one row -> one wide segment
    _same_contours(m4)
    m5 = np.zeros((40, 9000), np.uint8); m5[5:30, ::2] = 255                                     # 4500 segments in one band: more than the table holds
    _same_contours(m5)


def test_gpu_unclip_equals_the_oracle_on_random_boxes():
    """a11 as a HIP kernel (pp::unclip_quads, run next to the box scores): vertex for vertex what the ORACLE's unclip (oracle/oar_oracle.c,
    the restatement of db_bitmap.rs:279-368) produces -- compared directly, not through the product's own host routine
    -- on 20 000 random rectangles of every orientation, size and winding, at several ratios; degenerate boxes are
    dropped by both."""
    rng = np.random.default_rng(77)
    n = 20000
    cx, cy = rng.uniform(20, 940, n), rng.uniform(20, 940, n)
    w, h = rng.uniform(3, 600, n), rng.uniform(3, 90, n)
    ang = np.where(rng.random(n) < 0.4, 0.0, rng.uniform(-np.pi, np.pi, n))
    ca, sa = np.cos(ang), np.sin(ang)
    corners = np.array([[-0.5, -0.5], [0.5, -0.5], [0.5, 0.5], [-0.5, 0.5]])
    boxes = np.zeros((n, 4, 2), np.float32)
    for k, (ux, uy) in enumerate(corners):
        boxes[:, k, 0] = cx + ux * w * ca - uy * h * sa
        boxes[:, k, 1] = cy + ux * w * sa + uy * h * ca
    boxes[::3] = np.round(boxes[::3])                   # integer corners, as mini boxes of pixel contours often are
    boxes[1::7] = boxes[1::7, ::-1]                     # the other winding
    boxes[5] = boxes[5, [0, 0, 1, 1]]                   # zero area
    boxes[6, :] = boxes[6, 0]                           # a point
    for ratio in (1.5, 2.0, 0.5):
        got = api.k_unclip(boxes, ratio)
        bad = 0
        for i in range(n):
            ref = R.unclip(boxes[i], ratio)
            g = got[i]
            assert g is not None
            if g.shape != ref.shape or not np.array_equal(g, ref):
                bad += 1
        assert bad == 0, (ratio, bad)
    assert len(got[5]) == 0 and len(got[6]) == 0
