"""Host helpers of the overall-OCR row: the reference's own known answers for the box / crop utilities,
checked against BOTH the product helpers (oar_ocr_amd/structure.py, no GPU involved) and the oracle restatement
(oracle/structure_ref.py)."""
import numpy as np
import pytest

from oar_ocr_amd import structure as S
from oracle import structure_ref as O


def _img(w=100, h=100):
    y, x = np.mgrid[0:h, 0:w]
    return np.stack([x % 256, y % 256, (x + y) % 256], -1).astype(np.uint8)


CROP_CASES = [   # utils/bbox_crop.rs:163-283 -> (points, (width, height) | None)
    ([(10, 10), (50, 10), (50, 40), (10, 40)], (40, 30)),
    ([], None),
    ([(50, 50)], None),
    ([(-10, -5), (30, -5), (30, 25), (-10, 25)], (30, 25)),
    ([(80, 80), (150, 80), (150, 120), (80, 120)], (20, 20)),
    ([(20, 30), (60, 10), (80, 50), (40, 70), (10, 40)], (70, 60)),
]


@pytest.mark.parametrize("crop", [S.crop_bounding_box, O.cut])
@pytest.mark.parametrize("pts,want", CROP_CASES)
def test_crop_bounding_box_reference_cases(crop, pts, want):
    got = crop(_img(), np.asarray(pts, np.float32).reshape(-1, 2))
    if want is None:
        assert got is None
    else:
        assert (got.shape[1], got.shape[0]) == want
        x0, y0 = max(int(min(p[0] for p in pts)), 0), max(int(min(p[1] for p in pts)), 0)
        assert np.array_equal(got, _img()[y0:y0 + want[1], x0:x0 + want[0]])


@pytest.mark.parametrize("iou,rect", [(S.aabb_iou, S.from_coords), (O.iou, O.rect)])
def test_iou_reference_cases(iou, rect):
    a, b = rect(0, 0, 10, 10), rect(5, 5, 15, 15)                     # geometry.rs:1181-1200
    assert abs(float(iou(a, b)) - 25.0 / 175.0) < 1e-6
    assert float(iou(a, a)) == 1.0
    assert float(iou(a, rect(20, 20, 30, 30))) == 0.0
    assert float(iou(rect(40, 40, 150, 150), rect(50, 50, 200, 200))) > 0.3   # geometry.rs:1225-1240


@pytest.mark.parametrize("area,rect", [(S.polygon_area, S.from_coords), (O.shoelace, O.rect)])
def test_area(area, rect):
    assert float(area(rect(10, 20, 100, 80))) == 90.0 * 60.0
    assert float(area(np.zeros((2, 2), np.float32))) == 0.0
    tri = np.array([(0, 0), (4, 0), (0, 3)], np.float32)
    assert float(area(tri)) == 6.0


@pytest.mark.parametrize("mask,rect", [(S.mask_regions, S.from_coords), (O.paint, O.rect)])
def test_mask_regions(mask, rect):
    im = _img(200, 100)
    ref = im.copy()
    mask(im, [rect(10.7, 10.2, 50.9, 30.1), rect(150, 80, 400, 300), rect(-20, -20, 5, 5), rect(60, 60, 60, 90), rect(300, 10, 400, 20)])
    ref[10:30, 10:50] = 255            # `as u32` truncation (utils/image.rs:772-775)
    ref[80:100, 150:200] = 255         # clamped to the image (utils/image.rs:720-723)
    ref[0:5, 0:5] = 255                # negative coordinates saturate to 0
    assert np.array_equal(im, ref)     # the empty and the out-of-image rectangles change nothing


def test_split_boxes_by_containers_matches_oracle_rule():
    left, right = S.from_coords(0, 0, 100, 200), S.from_coords(100, 0, 200, 200)
    wide = np.array([(20, 50), (180, 52), (180, 72), (20, 70)], np.float32)        # spans both containers
    narrow = np.array([(10, 100), (60, 100), (60, 120), (10, 120)], np.float32)    # inside one
    sliver = np.array([(60, 150), (103, 150), (103, 170), (60, 170)], np.float32)  # 3 px into the right container: < 0.3 of its area
    got = S.split_boxes_by_containers([wide, narrow, sliver], [left, right])
    assert len(got) == 4
    assert np.array_equal(got[0], S.from_coords(20, 50, 100, 72)) and np.array_equal(got[1], S.from_coords(100, 50, 180, 72))
    assert got[2] is narrow and got[3] is sliver
    assert S.split_boxes_by_containers([wide], []) == [wide]
