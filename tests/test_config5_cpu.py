"""Config-5 stages: the oracle restatement against the reference's own inline tests and
closed-form cases.  CPU only."""
import numpy as np

from oracle import cpu_ref as R


def _from_coords(x1, y1, x2, y2):   # BoundingBox::from_coords (processors/geometry.rs): tl, tr, br, bl
    return np.array([[x1, y1], [x2, y1], [x2, y2], [x1, y2]], np.float32)


def test_rotate_back_to_original_reference_vectors():
    # oar-ocr-core/src/processors/geometry.rs:1269-1330
    b = _from_coords(0.0, 1.0, 2.0, 3.0)
    assert np.array_equal(R.rotate_back_points(b, 0.0, 10, 20), b)
    assert R.rotate_back_points(_from_coords(0, 0, 1, 1), 90.0, 3, 4).tolist() == [[4, 0], [4, 1], [3, 1], [3, 0]]
    assert R.rotate_back_points(_from_coords(1, 1, 2, 2), 180.0, 4, 3).tolist() == [[3, 2], [2, 2], [2, 1], [3, 1]]
    assert R.rotate_back_points(_from_coords(0, 0, 1, 1), 270.0, 3, 4).tolist() == [[0, 3], [0, 2], [1, 2], [1, 3]]


def test_apply_orientation_from_class_id_reference_cases():
    # src/oarocr/preprocess.rs:286-380: 100x200 (w x h) page
    img = np.zeros((200, 100, 3), np.uint8)
    out, corr = R.correct_orientation(img, None)
    assert out is img and corr is None
    out, corr = R.correct_orientation(img, 0)
    assert out is img and corr == (0.0, 100, 200)
    out, corr = R.correct_orientation(img, 1)      # 90 deg detected -> rotate270
    assert out.shape[:2] == (100, 200) and corr == (90.0, 200, 100)
    out, corr = R.correct_orientation(img, 2)
    assert out.shape[:2] == (200, 100) and corr == (180.0, 100, 200)
    out, corr = R.correct_orientation(img, 3)      # 270 deg detected -> rotate90
    assert out.shape[:2] == (100, 200) and corr == (270.0, 200, 100)
    out, corr = R.correct_orientation(img, 7)      # unknown class: no rotation, metadata kept
    assert out is img and corr == (630.0, 100, 200)


def test_rotations_are_clockwise_quarter_turns_and_compose():
    rng = np.random.default_rng(0)
    im = rng.integers(0, 256, (5, 7, 3), dtype=np.uint8)
    r90 = R.rotate_rgb(im, 1)
    assert r90.shape == (7, 5, 3)
    assert np.array_equal(r90[0, 0], im[4, 0]) and np.array_equal(r90[0, 4], im[0, 0])    # top row = first column, bottom-up
    assert np.array_equal(R.rotate_rgb(r90, 1), R.rotate_rgb(im, 2))
    assert np.array_equal(R.rotate_rgb(R.rotate_rgb(im, 2), 2), im)
    assert np.array_equal(R.rotate_rgb(R.rotate_rgb(im, 1), 3), im)
    assert np.array_equal(R.rotate_rgb(im, 2), im[::-1, ::-1])


def test_rotate_back_inverts_the_correction_on_pixel_centres():
    # a point (x, y) of the ORIGINAL page lands at p' in the corrected page; rotate_back_to_original maps the box of
    # that pixel back onto the pixel's own box (corner order aside)
    rng = np.random.default_rng(1)
    im = rng.integers(0, 256, (6, 9, 3), dtype=np.uint8)
    for cls in (1, 2, 3):
        cor, (angle, rw, rh) = R.correct_orientation(im, cls)
        for (y, x) in [(0, 0), (2, 5), (5, 8)]:
            ys, xs = np.nonzero((cor == im[y, x]).all(-1))
            hit = [(yy, xx) for yy, xx in zip(ys, xs)]
            box = None
            for yy, xx in hit:
                back = R.rotate_back_points(_from_coords(xx, yy, xx + 1, yy + 1), angle, rw, rh)
                if back[:, 0].min() == x and back[:, 1].min() == y:
                    box = back
            assert box is not None


def test_classifier_resize_rule():
    # pp_lcnet.rs:158-170: scale = 256 / short, round, max(crop), centre crop with integer halving
    assert R.cls_resize_dims(100, 200, 256, 224, 224) == (256, 512, 16, 144)
    assert R.cls_resize_dims(640, 480, 256, 224, 224) == (341, 256, 58, 16)
    assert R.cls_resize_dims(3000, 100, 256, 224, 224) == (7680, 256, 3728, 16)
    assert R.cls_resize_dims(301, 40, 0, 160, 80) == (160, 80, 0, 0)       # text-line mode: direct resize
    x = R.cls_preprocess(np.full((50, 70, 3), 255, np.uint8))
    assert x.shape == (3, 224, 224)
    assert np.allclose(x[:, 0, 0], (1.0 - np.array(R.IMAGENET_MEAN)) / np.array(R.IMAGENET_STD), atol=1e-6)   # RGB order


def test_topk_is_a_stable_descending_sort():
    ids, sc = R.topk(np.array([0.1, 0.5, 0.5, 0.2], np.float32), 4)
    assert ids.tolist() == [1, 2, 3, 0] and sc.tolist() == [0.5, 0.5, 0.20000000298023224, 0.10000000149011612]
    ids, _ = R.topk(np.array([0.25, 0.25, 0.25, 0.25], np.float32), 1)
    assert ids.tolist() == [0]                                             # first index wins ties (A.7)


def test_uvdoc_post_truncates_and_swaps_to_rgb():
    pred = np.zeros((3, 1, 4), np.float32)
    pred[0, 0] = [0.0, 0.5, 1.0, 2.0]            # B
    pred[1, 0] = [-1.0, 0.999, 0.00392, 0.3]     # G
    pred[2, 0] = [np.nan, 1.0 / 255.0, 0.9961, 0.1]   # R
    out = R.uvdoc_postprocess(pred, (4, 1))
    assert out[0, :, 2].tolist() == [0, 127, 255, 255]          # B plane lands in channel 2; 127.5 -> 127 (truncation)
    assert out[0, :, 1].tolist() == [0, 254, 0, 76]
    assert out[0, :, 0].tolist() == [0, 1, 254, 25]             # NaN -> 0 (`as u8`)
    x = R.uvdoc_preprocess(np.dstack([np.full((2, 2), 10, np.uint8), np.full((2, 2), 20, np.uint8), np.full((2, 2), 30, np.uint8)]), (2, 2))
    assert np.allclose(x[:, 0, 0], [30 / 255.0, 20 / 255.0, 10 / 255.0])   # BGR planes, no mean shift
