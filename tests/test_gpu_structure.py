"""GPU parity of OARStructure's overall OCR: `oar_ocr_amd.structure.OverallOCR` over the C-ABI adapters
against `oracle.structure_ref.OracleOverallOCR` on the same page and layout.  Boxes bit-exact, texts equal, scores <= 1e-3."""
import numpy as np
import pytest

from oar_ocr_amd import api, structure
from oar_ocr_amd.synth import models, pages
from oracle import structure_ref

pytestmark = pytest.mark.gpu

SIZE = 640


@pytest.fixture(scope="module")
def world():
    det, _ = models.build_det("tiny", seed=0)
    rec, _ = models.build_rec("tiny", vocab=6906, seed=1)
    line, _ = models.build_cls(2, seed=9)
    chars = api.read_dict(models.synth_dict(6904))
    page = pages.make_page(11, (SIZE, SIZE), 18)
    cfg = api.TextDetectionConfig(score_threshold=0.3, box_threshold=0.6, unclip_ratio=1.5)
    d = api.TextDetectionPredictor(det, cfg)
    r = api.TextRecognitionPredictor(rec, chars)
    lo = api.ImageClassifier(line, input_hw=(80, 160), resize_short=0)
    boxes = [b.bbox for b in d.predict([page])[0]]
    assert len(boxes) >= 6, "the synthetic page must produce text boxes"
    return dict(det=det, rec=rec, line=line, chars=chars, page=page, d=d, r=r, lo=lo, boxes=boxes)


def _layout(boxes):
    """Left / right text columns (wide lines get split), a formula and a table over two detections, an image and a
    footer over blank strips (the footer triggers the fallback recognition, the image must not)."""
    f, t = structure.aabb(boxes[2]), structure.aabb(boxes[4])
    return [
        (structure.from_coords(0, 0, SIZE / 2, SIZE - 40), "text"),
        (structure.from_coords(SIZE / 2, 0, SIZE, SIZE - 40), "text"),
        (structure.from_coords(f[0] - 2, f[1] - 2, f[2] + 2, f[3] + 2), "formula"),
        (structure.from_coords(t[0], t[1], t[2], t[3]), "table"),
        (structure.from_coords(4, SIZE - 30, 200, SIZE - 4), "image"),
        (structure.from_coords(300, SIZE - 30.5, 620.7, SIZE - 2), "footer"),
        (structure.from_coords(SIZE + 10, 10, SIZE + 50, 40), "aside_text"),      # outside the page: the crop fails, nothing is added
    ]


def _compare(got, ref):
    assert len(got) == len(ref) and len(got) > 0
    for g, r in zip(got, ref):
        assert np.array_equal(np.asarray(g.bounding_box, np.float32), np.asarray(r["box"], np.float32))
        assert g.text == r["text"]
        if r["text"] is not None:
            assert abs(g.confidence - r["score"]) <= 1e-3


@pytest.mark.parametrize("mode", ["layout containers + line orientation", "region blocks + formula masking"])
def test_overall_ocr_matches_oracle(world, mode):
    layout = _layout(world["boxes"])
    elems = [structure.LayoutElement(b, t) for b, t in layout]
    if mode.startswith("layout"):
        prod = structure.OverallOCR(world["d"], world["r"], world["lo"], region_batch_size=8)
        orc = structure_ref.OracleOverallOCR(world["det"], world["rec"], world["chars"], line_orientation=world["line"], region_batch_size=8)
        got, ref = prod.run(world["page"], elems), orc.run(world["page"], layout)
    else:
        blocks = [structure.from_coords(0, 0, SIZE, SIZE / 3), structure.from_coords(0, SIZE / 3, SIZE, SIZE)]
        prod = structure.OverallOCR(world["d"], world["r"], None, region_batch_size=5, formula_recognition=True)
        orc = structure_ref.OracleOverallOCR(world["det"], world["rec"], world["chars"], region_batch_size=5, formula_recognition=True)
        got = prod.run(world["page"], elems, [structure.RegionBlock(b) for b in blocks])
        ref = orc.run(world["page"], layout, blocks)
    _compare(got, ref)
    n_plain = len(world["boxes"])
    assert len(got) != n_plain or any(not np.array_equal(g.bounding_box, b) for g, b in zip(got, world["boxes"])), \
        "the layout must change the plain detection result (splits / fallback / masking)"


def test_overall_ocr_without_layout_is_plain_det_rec(world):
    prod = structure.OverallOCR(world["d"], world["r"], None, region_batch_size=8)
    orc = structure_ref.OracleOverallOCR(world["det"], world["rec"], world["chars"], region_batch_size=8)
    _compare(prod.run(world["page"], []), orc.run(world["page"], []))


def test_overall_ocr_across_pages_matches_oracle(world):
    """`precompute_overall_ocr_across_pages` (structure.rs:2859-3260; VERDICT r5 missing #2): four pages of three different sizes (two detector shape
    groups in one batch), one slot already in error, detection batches of 3, ONE width-sorted recognition queue over the whole document cut into
    batches of 8 (a crop's batch -- hence its padding width -- depends on the other pages), formula masking, text-line orientation, per-page
    layout refinement.  Then the seal rule: a seal-enabled pipeline leaves every page untouched."""
    sizes = [(SIZE, SIZE), (480, 800), (SIZE, SIZE), (320, 480)]
    imgs = [pages.make_page(40 + i, hw, 6 + 4 * i) for i, hw in enumerate(sizes)]
    layouts = []
    for img, (h, w) in zip(imgs, sizes):
        boxes = [b.bbox for b in world["d"].predict([img])[0]]
        f = structure.aabb(boxes[1])
        layouts.append([
            (structure.from_coords(0, 0, w / 2, h - 30), "text"),
            (structure.from_coords(w / 2, 0, w, h - 30), "text"),
            (structure.from_coords(f[0] - 2, f[1] - 2, f[2] + 2, f[3] + 2), "formula"),
            (structure.from_coords(10, h - 26.5, w - 10.25, h - 2), "footer"),
        ])
    blocks2 = [structure.from_coords(0, 0, sizes[2][1], sizes[2][0] / 2), structure.from_coords(0, sizes[2][0] / 2, sizes[2][1], sizes[2][0])]
    prepared, oracle_pages = [], []
    for i, (img, lay) in enumerate(zip(imgs, layouts)):
        rb = [structure.RegionBlock(b) for b in blocks2] if i == 2 else None
        prepared.append(structure.PreparedPage(img, [structure.LayoutElement(b, t) for b, t in lay], rb))
        oracle_pages.append({"image": img, "layout": lay, "region_blocks": blocks2 if i == 2 else None})
    prepared.insert(1, structure.PreparedPage(imgs[0], [], error=RuntimeError("load failed")))     # a slot that already holds Err
    oracle_pages.insert(1, None)
    prod = structure.OverallOCR(world["d"], world["r"], world["lo"], region_batch_size=8, formula_recognition=True, image_batch_size=3)
    orc = structure_ref.OracleOverallOCR(world["det"], world["rec"], world["chars"], line_orientation=world["line"], region_batch_size=8, formula_recognition=True)
    assert prod.precompute_across_pages(prepared) is True
    ref = orc.precompute(oracle_pages, image_batch_size=3)
    assert prepared[1].precomputed_text_regions is None and ref[1] is None and isinstance(prepared[1].error, RuntimeError)
    total = 0
    for pg, r in zip(prepared, ref):
        if r is None:
            continue
        assert pg.error is None
        _compare(pg.precomputed_text_regions, r)
        total += len(r)
    assert total > 30
    # the cross-page queue is not the per-page one: at least one page's single-page result differs in a score (another padding width) or is equal
    # only because its crops happened to share batches -- what must hold is that both paths agree with THEIR oracles; here: the seal rule
    sealed = structure.OverallOCR(world["d"], world["r"], None, region_batch_size=8, seal_text_detection=True)
    fresh = [structure.PreparedPage(imgs[0], [])]
    assert sealed.precompute_across_pages(fresh) is False and fresh[0].precomputed_text_regions is None
    assert orc.precompute(oracle_pages, seal_enabled=True) is None
