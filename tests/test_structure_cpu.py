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


# ------------------------------------------------------------------------------------------------ the cross-page control flow, no GPU
class _Det:
    """Stub text detector: one box per dark row band of the page's red channel; `fail_batches` makes every multi-page call raise."""
    def __init__(self, fail_batches=False, fail_pages=()):
        self.calls, self.fail_batches, self.fail_pages = [], fail_batches, set(fail_pages)

    @staticmethod
    def recommended_batch_size():
        return 8

    def predict(self, images):
        from oar_ocr_amd import api
        self.calls.append(len(images))
        if self.fail_batches and len(images) > 1:
            raise api.OCRError(6, "batched detection failed")
        out = []
        for im in images:
            if int(im[0, 0, 2]) in self.fail_pages:
                raise api.OCRError(6, "page detection failed")
            rows = np.nonzero(im[:, 0, 0] == 0)[0]
            boxes, start = [], None
            for y in range(im.shape[0] + 1):
                dark = y < im.shape[0] and im[y, 0, 0] == 0
                if dark and start is None:
                    start = y
                if not dark and start is not None:
                    boxes.append(type("D", (), {"bbox": S.from_coords(4, start, im.shape[1] - 4 - 10 * len(boxes), y)})())
                    start = None
            out.append(boxes)
        return out


class _Rec:
    """Stub recognizer: text = "<w>x<h>" of the crop; the batch whose first crop is `poison` wide raises."""
    def __init__(self, poison=None):
        self.batches, self.poison = [], poison

    @staticmethod
    def recommended_batch_size():
        return 64

    def predict(self, crops):
        from oar_ocr_amd import api
        self.batches.append([c.shape[1] for c in crops])
        if self.poison is not None and crops and crops[0].shape[1] == self.poison:
            raise api.OCRError(6, "recognition batch failed")
        return type("R", (), {"texts": [f"{c.shape[1]}x{c.shape[0]}" for c in crops], "scores": [0.9] * len(crops)})()


def _banded_page(tag, w, bands):
    im = np.full((120, w, 3), 255, np.uint8)
    for y0, y1 in bands:
        im[y0:y1] = 0
    im[0, 0, 2] = tag
    return im


def test_cross_page_overall_ocr_control_flow(monkeypatch):
    """precompute_overall_ocr_across_pages (structure.rs:2859-3260) over stub adapters: detection batches of image_batch_size with per-page fallback
    after a failed batch, a page whose own detection fails carries the error alone, one document-wide queue sorted by crop ratio and cut into
    region_batch_size chunks, a failed recognition chunk leaves its slots empty, texts return to (page, detection) slots, seal stand-down."""
    from oar_ocr_amd import api
    monkeypatch.setattr(api, "k_rotate_crop", lambda page, b: S.crop_bounding_box(page, b))
    monkeypatch.setattr(api, "host_sort_quad_boxes", lambda boxes: sorted(range(len(boxes)), key=lambda i: (float(boxes[i][:, 1].min()), float(boxes[i][:, 0].min()))))
    pages_ = [_banded_page(1, 200, [(10, 30), (50, 70)]), _banded_page(2, 300, [(20, 40)]), _banded_page(3, 100, [(5, 25), (40, 60), (80, 100)])]

    def prepared():
        return [S.PreparedPage(p, []) for p in pages_]

    # plain run: one detection call of 3 pages; six crops in ONE queue sorted by ratio, chunks of 4
    det, rec = _Det(), _Rec()
    pp = prepared()
    assert S.OverallOCR(det, rec, region_batch_size=4, image_batch_size=8).precompute_across_pages(pp) is True
    assert det.calls == [3]
    widths = [w for b in rec.batches for w in b]
    assert len(rec.batches) == 2 and len(rec.batches[0]) == 4 and widths == sorted(widths)            # every band is 20 px tall: ratio order = width order
    assert [[r.text for r in p.precomputed_text_regions] for p in pp] == [["192x20", "182x20"], ["292x20"], ["92x20", "82x20", "72x20"]]
    # image_batch_size 2 -> calls of 2 + 1; a failing batch falls back to per-page calls for ITS pages only
    det = _Det(fail_batches=True)
    pp = prepared()
    S.OverallOCR(det, _Rec(), region_batch_size=64, image_batch_size=2).precompute_across_pages(pp)
    assert det.calls == [2, 1, 1, 1] and all(p.error is None and p.precomputed_text_regions for p in pp)
    # a page whose own detection fails (after its batch failed) keeps the error; the others are complete
    det = _Det(fail_batches=True, fail_pages=(2,))
    pp = prepared()
    S.OverallOCR(det, _Rec(), region_batch_size=64, image_batch_size=3).precompute_across_pages(pp)
    assert isinstance(pp[1].error, api.OCRError) and pp[1].precomputed_text_regions is None
    assert len(pp[0].precomputed_text_regions) == 2 and len(pp[2].precomputed_text_regions) == 3
    # a failed recognition chunk: its crops stay unrecognised (no region), later chunks still run
    rec = _Rec(poison=72)
    pp = prepared()
    S.OverallOCR(_Det(), rec, region_batch_size=2, image_batch_size=8).precompute_across_pages(pp)
    assert [[r.text for r in p.precomputed_text_regions] for p in pp] == [["192x20", "182x20"], ["292x20"], ["92x20"]]
    # pages already in error are skipped; seal-enabled pipelines stand down before touching anything
    pp = prepared()
    pp[0].error = RuntimeError("decode")
    det = _Det()
    S.OverallOCR(det, _Rec(), image_batch_size=8).precompute_across_pages(pp)
    assert det.calls == [2] and pp[0].precomputed_text_regions is None
    pp = prepared()
    det = _Det()
    assert S.OverallOCR(det, _Rec(), seal_text_detection=True).precompute_across_pages(pp) is False and det.calls == [] and pp[0].precomputed_text_regions is None
