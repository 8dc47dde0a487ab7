"""Overall OCR of OARStructure -- the second caller of the text-detection / text-recognition adapters
(SURVEY 8f rank 1).

Mirrors `OARStructure::run_overall_ocr` (src/oarocr/structure.rs:2208-2540),
`precompute_overall_ocr_across_pages` (src/oarocr/structure.rs:2859-3260: detection batched over the pages of a document, the crops of ALL pages in
one width-sorted recognition queue, per-page results scattered back) and
`refine_overall_ocr_with_layout` (src/oarocr/structure.rs:1438-1660): formula masking, text detection on the masked
page, splitting of text boxes that span several layout containers, reading-order sort, cropping from the unmasked
page, optional text-line orientation, width-sorted recognition batches, then the two layout-guided refinements
(re-recognition per overlapped layout block, fallback recognition of text-less blocks).

Host orchestration only: every pixel / tensor step runs through the C-ABI adapters of `api.py` (detector, recognizer,
classifier, crop and rotate kernels).  The layout elements themselves come from the caller -- the layout detectors are
outside this path.  All box arithmetic is f32, in the reference's operation order."""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import api

F = np.float32

TEXT_BOX_SPLIT_IOA_THRESHOLD = F(0.3)   # src/oarocr/structure.rs:49

# containers that split a text box when no region blocks are given (src/oarocr/structure.rs:2281-2303)
SPLIT_CONTAINER_TYPES = frozenset({
    "doc_title", "paragraph_title", "text", "content", "abstract", "header", "footer", "footnote", "number", "reference",
    "reference_content", "algorithm", "aside_text", "list", "figure_title", "table_title", "chart_title",
    "figure_table_chart_title"})
EXCLUDED_FROM_MATCHING = frozenset({"formula", "formula_number", "table", "seal"})   # structure.rs:1469-1477
NO_FALLBACK_TYPES = frozenset({"image", "chart"})                                    # structure.rs:1590-1595
FORMULA_TYPES = frozenset({"formula", "formula_number"})                             # domain/structure.rs:2266-2271


@dataclass
class LayoutElement:
    """domain/structure.rs LayoutElement: `bbox` is [n, 2] f32 points, `element_type` the reference's `as_str()` name."""
    bbox: np.ndarray
    element_type: str


@dataclass
class RegionBlock:
    bbox: np.ndarray


@dataclass
class PreparedPage:
    """The fields of the reference's PreparedPage that the overall OCR reads and writes (src/oarocr/structure.rs: current_image, layout_elements,
    detected_region_blocks, precomputed_text_regions).  `error` stands for the page's slot holding Err(..): such pages are skipped, and a page whose
    own detection / cropping / refinement fails gets its error recorded here instead of aborting the document."""
    current_image: np.ndarray
    layout_elements: Sequence[LayoutElement]
    detected_region_blocks: Optional[Sequence[RegionBlock]] = None
    precomputed_text_regions: Optional[List["api.TextRegion"]] = None
    error: Optional[Exception] = None


def from_coords(x1, y1, x2, y2) -> np.ndarray:
    """BoundingBox::from_coords (processors/geometry.rs:98-106)."""
    return np.array([[x1, y1], [x2, y1], [x2, y2], [x1, y2]], np.float32)


def aabb(box) -> tuple:
    b = np.asarray(box, np.float32).reshape(-1, 2)
    return F(b[:, 0].min()), F(b[:, 1].min()), F(b[:, 0].max()), F(b[:, 1].max())


def polygon_area(box) -> np.float32:
    """BoundingBox::area (processors/geometry.rs:141-154): shoelace in f32, one product at a time."""
    b = np.asarray(box, np.float32).reshape(-1, 2)
    n = b.shape[0]
    if n < 3:
        return F(0.0)
    area = F(0.0)
    for i in range(n):
        j = (i + 1) % n
        area = F(area + F(b[i, 0] * b[j, 1]))
        area = F(area - F(b[j, 0] * b[i, 1]))
    return F(np.abs(area) / F(2.0))


def aabb_iou(a, b) -> np.float32:
    """BoundingBox::iou (processors/geometry.rs:688-717)."""
    ax0, ay0, ax1, ay1 = aabb(a)
    bx0, by0, bx1, by1 = aabb(b)
    ix0, iy0, ix1, iy1 = max(ax0, bx0), max(ay0, by0), min(ax1, bx1), min(ay1, by1)
    if ix0 >= ix1 or iy0 >= iy1:
        return F(0.0)
    inter = F(F(ix1 - ix0) * F(iy1 - iy0))
    if inter <= 0:
        return F(0.0)
    union = F(F(F(F(ax1 - ax0) * F(ay1 - ay0)) + F(F(bx1 - bx0) * F(by1 - by0))) - inter)
    if union <= 0:
        return F(0.0)
    return F(inter / union)


def _as_u32(v) -> int:
    """Rust `f32 as u32`: truncation toward zero, saturating, NaN -> 0."""
    v = float(v)
    if v != v or v <= 0.0:
        return 0
    return min(int(v), 0xFFFFFFFF)


def mask_regions(image: np.ndarray, boxes: Sequence[np.ndarray], fill=(255, 255, 255)) -> None:
    """utils/image.rs:709-780: fills the AABB of each box in place; empty / out-of-image rectangles are skipped."""
    h, w = image.shape[:2]
    for b in boxes:
        x0, y0, x1, y1 = aabb(b)
        x0, y0, x1, y1 = min(_as_u32(x0), w), min(_as_u32(y0), h), min(_as_u32(x1), w), min(_as_u32(y1), h)
        if x0 >= x1 or y0 >= y1:
            continue
        image[y0:y1, x0:x1] = np.asarray(fill, np.uint8)


def crop_bounding_box(image: np.ndarray, box) -> Optional[np.ndarray]:
    """BBoxCrop::crop_bounding_box (utils/bbox_crop.rs:26-71); None where the reference returns Err."""
    b = np.asarray(box, np.float32).reshape(-1, 2)
    if b.shape[0] == 0:
        return None
    h, w = image.shape[:2]
    x0, y0, x1, y1 = aabb(b)
    x0, y0 = max(x0, F(0.0)), max(y0, F(0.0))
    cx0, cy0 = min(_as_u32(x0), max(w - 1, 0)), min(_as_u32(y0), max(h - 1, 0))
    cx1, cy1 = min(_as_u32(x1), w), min(_as_u32(y1), h)
    if cx1 <= cx0 or cy1 <= cy0:
        return None
    return np.ascontiguousarray(image[cy0:cy1, cx0:cx1])


def _overlaps(a, b, min_pixels) -> bool:
    ax0, ay0, ax1, ay1 = aabb(a)
    bx0, by0, bx1, by1 = aabb(b)
    return F(min(ax1, bx1) - max(ax0, bx0)) > min_pixels and F(min(ay1, by1) - max(ay0, by0)) > min_pixels


def split_boxes_by_containers(boxes: Sequence[np.ndarray], containers: Sequence[np.ndarray]) -> List[np.ndarray]:
    """Cross-layout splitting (src/oarocr/structure.rs:2263-2358)."""
    if not containers:
        return list(boxes)
    out = []
    for box in boxes:
        self_area = polygon_area(box)
        if self_area <= 0:
            out.append(box)
            continue
        bx0, by0, bx1, by1 = aabb(box)
        inter = []
        for c in containers:
            cx0, cy0, cx1, cy1 = aabb(c)
            ix0, iy0, ix1, iy1 = max(bx0, cx0), max(by0, cy0), min(bx1, cx1), min(by1, cy1)
            if F(ix1 - ix0) <= F(2.0) or F(iy1 - iy0) <= F(2.0):
                continue
            ib = from_coords(ix0, iy0, ix1, iy1)
            ia = polygon_area(ib)
            if ia <= 0:
                continue
            if F(ia / self_area) >= TEXT_BOX_SPLIT_IOA_THRESHOLD:
                inter.append(ib)
        if len(inter) >= 2:
            out.extend(inter)
        else:
            out.append(box)
    return out


class OverallOCR:
    """`OARStructure::run_overall_ocr` over this package's adapters.

    det / rec / text_line_orientation are `api.TextDetectionPredictor`, `api.TextRecognitionPredictor` and (optional)
    `api.ImageClassifier(input_hw=(80, 160), resize_short=0)`; `formula_recognition` says whether the structure pipeline
    has a formula recogniser attached (only then are formula regions masked before detection, structure.rs:2228-2241)."""

    def __init__(self, det, rec, text_line_orientation=None, region_batch_size: Optional[int] = None, formula_recognition: bool = False,
                 image_batch_size: Optional[int] = None, seal_text_detection: bool = False):
        self.det, self.rec, self.line_ori = det, rec, text_line_orientation
        self.region_batch_size = region_batch_size
        self.formula_recognition = formula_recognition
        self.image_batch_size = image_batch_size          # pipeline.image_batch_size (cross-page detection batches)
        self.seal_text_detection = seal_text_detection    # a seal detector is attached: the cross-page path stands down (structure.rs:2874-2878)

    def _batch_size(self) -> int:
        return max(self.region_batch_size if self.region_batch_size is not None else self.rec.recommended_batch_size(), 1)

    def _recognize(self, crops: Sequence[np.ndarray]):
        texts, scores = [], []
        bs = self._batch_size()
        for s in range(0, len(crops), bs):
            r = self.rec.predict(crops[s:s + bs])
            texts.extend(r.texts)
            scores.extend(r.scores)
        return texts, scores

    def run(self, page: np.ndarray, layout_elements: Sequence[LayoutElement], region_blocks: Optional[Sequence[RegionBlock]] = None) -> List[api.TextRegion]:
        page = np.ascontiguousarray(page, np.uint8)
        ocr_image = page
        if self.formula_recognition:
            masks = [e.bbox for e in layout_elements if e.element_type in FORMULA_TYPES]
            if masks:
                ocr_image = page.copy()
                mask_regions(ocr_image, masks)
        boxes = [d.bbox for d in self.det.predict([ocr_image])[0]]

        if boxes:
            if region_blocks is not None:
                containers = [r.bbox for r in region_blocks]
            else:
                containers = [e.bbox for e in layout_elements if e.element_type in SPLIT_CONTAINER_TYPES]
            boxes = split_boxes_by_containers(boxes, containers)
            order = api.host_sort_quad_boxes(np.stack(boxes))          # PaddleX reading order before cropping
            boxes = [boxes[i] for i in order]

        regions: List[api.TextRegion] = []
        if boxes:
            crops, valid = [], []
            for i, b in enumerate(boxes):                              # TextCroppingProcessor::new(true), crops from the UNMASKED page
                c = api.k_rotate_crop(page, b) if np.asarray(b).reshape(-1, 2).shape[0] == 4 else crop_bounding_box(page, b)
                if c is not None:
                    crops.append(c)
                    valid.append(i)
            if crops:
                if self.line_ori is not None:
                    for i, cls in enumerate(self.line_ori.predict(crops)):
                        if cls and cls[0].class_id == 1:
                            crops[i] = api.k_rotate_rgb(crops[i], 2)
                ratios = [F(c.shape[1]) / F(max(c.shape[0], 1)) for c in crops]
                by_ratio = sorted(range(len(crops)), key=lambda i: ratios[i])      # stable, like sort_by(partial_cmp)
                recognized = [None] * len(boxes)
                bs = self._batch_size()
                for s in range(0, len(by_ratio), bs):
                    chunk = by_ratio[s:s + bs]
                    r = self.rec.predict([crops[i] for i in chunk])
                    for i, text, score in zip(chunk, r.texts, r.scores):
                        if text:
                            recognized[valid[i]] = (text, float(score))
                for i, rec in enumerate(recognized):                   # original detection order
                    if rec is not None:
                        regions.append(api.TextRegion(bounding_box=boxes[i], text=rec[0], confidence=rec[1], dt_poly=boxes[i], rec_poly=boxes[i]))
        self._refine(regions, layout_elements, page)
        return regions

    # ------------------------------------------------------------------ the cross-page path
    def _masked(self, pg: PreparedPage) -> np.ndarray:
        """The image text detection sees: formula regions painted white when a formula recogniser is attached (structure.rs:2917-2928)."""
        img = np.ascontiguousarray(pg.current_image, np.uint8)
        if self.formula_recognition:
            masks = [e.bbox for e in pg.layout_elements if e.element_type in FORMULA_TYPES]
            if masks:
                img = img.copy()
                mask_regions(img, masks)
        return img

    def precompute_across_pages(self, prepared_pages: Sequence[PreparedPage]) -> bool:
        """`precompute_overall_ocr_across_pages` (src/oarocr/structure.rs:2859-3260): fills `precomputed_text_regions` of every page it can.
        Returns False when the path stands down (seal-enabled pipeline) and nothing was touched.

        Detection runs over the pages in batches of `image_batch_size` (a failed batch falls back to per-page detection of its pages); every page's
        boxes are split by its containers, sorted and cropped from the unmasked page; the crops of ALL pages form one queue, stably sorted by
        width / height and recognised in batches of `region_batch_size`; texts go back to (page, detection index); each page is then refined against
        its own layout.  A page whose own step fails carries the error; the others are unaffected."""
        if self.seal_text_detection:
            return False
        n = len(prepared_pages)
        live = [i for i, pg in enumerate(prepared_pages) if pg.error is None]
        det_bs = max(self.image_batch_size if self.image_batch_size is not None else self.det.recommended_batch_size(), 1)
        # 1. batched detection on the masked pages
        page_boxes: List[Optional[List[np.ndarray]]] = [None] * n
        seen = {i: self._masked(prepared_pages[i]) for i in live}
        for s in range(0, len(live), det_bs):
            idx = live[s:s + det_bs]
            try:
                dets = self.det.predict([seen[i] for i in idx])
            except api.OCRError:
                continue                                               # these pages are detected one by one below
            for i, d in zip(idx, dets):
                page_boxes[i] = [q.bbox for q in d]
        # 2. per page: containers, reading order, crops into the document-wide queue
        boxes_of: List[Optional[List[np.ndarray]]] = [None] * n
        recognized: List[Optional[list]] = [None] * n
        queue = []                                                     # (page, detection index, wh ratio, crop)
        for i in live:
            pg = prepared_pages[i]
            boxes = page_boxes[i]
            if boxes is None:
                try:
                    boxes = [q.bbox for q in self.det.predict([seen[i]])[0]]
                except api.OCRError as e:
                    pg.error = e
                    continue
            if boxes:
                if pg.detected_region_blocks is not None:
                    containers = [r.bbox for r in pg.detected_region_blocks]
                else:
                    containers = [e.bbox for e in pg.layout_elements if e.element_type in SPLIT_CONTAINER_TYPES]
                boxes = split_boxes_by_containers(boxes, containers)
                order = api.host_sort_quad_boxes(np.stack(boxes))
                boxes = [boxes[k] for k in order]
            page = np.ascontiguousarray(pg.current_image, np.uint8)
            try:
                for k, b in enumerate(boxes):
                    c = api.k_rotate_crop(page, b) if np.asarray(b).reshape(-1, 2).shape[0] == 4 else crop_bounding_box(page, b)
                    if c is not None:
                        queue.append((i, k, F(c.shape[1]) / F(max(c.shape[0], 1)), c))
            except api.OCRError as e:
                pg.error = e
                queue = [q for q in queue if q[0] != i]
                continue
            boxes_of[i], recognized[i] = boxes, [None] * len(boxes)
        # 3. one queue for the whole document: optional line orientation, stable sort by ratio, recognition batches
        if queue:
            if self.line_ori is not None:
                try:
                    for k, cls in enumerate(self.line_ori.predict([q[3] for q in queue])):
                        if cls and cls[0].class_id == 1:
                            queue[k] = queue[k][:3] + (api.k_rotate_rgb(queue[k][3], 2),)
                except api.OCRError:
                    pass                                               # "proceeding without rotation"
            queue.sort(key=lambda q: q[2])
            bs = self._batch_size()
            for s in range(0, len(queue), bs):
                chunk = queue[s:s + bs]
                try:
                    r = self.rec.predict([q[3] for q in chunk])
                except api.OCRError:
                    continue                                           # the batch is skipped, its slots stay empty
                for (i, k, _, _), text, score in zip(chunk, r.texts, r.scores):
                    if text and recognized[i] is not None:
                        recognized[i][k] = (text, float(score))
        # 4. per page: regions in detection order, layout-guided refinement
        for i in live:
            pg = prepared_pages[i]
            if pg.error is not None or recognized[i] is None:
                continue
            regions = [api.TextRegion(bounding_box=boxes_of[i][k], text=t[0], confidence=t[1], dt_poly=boxes_of[i][k], rec_poly=boxes_of[i][k])
                       for k, t in enumerate(recognized[i]) if t is not None]
            try:
                self._refine(regions, pg.layout_elements, np.ascontiguousarray(pg.current_image, np.uint8))
            except api.OCRError as e:
                pg.error = e
                continue
            pg.precomputed_text_regions = regions
        return True

    def _refine(self, regions: List[api.TextRegion], layout_elements: Sequence[LayoutElement], page: np.ndarray) -> None:
        """refine_overall_ocr_with_layout (src/oarocr/structure.rs:1438-1660)."""
        if not regions or not layout_elements:
            return
        min_pixels = F(3.0)
        matched = [[li for li, e in enumerate(layout_elements)
                    if e.element_type not in EXCLUDED_FROM_MATCHING and _overlaps(r.bounding_box, e.bbox, min_pixels)] for r in regions]
        appended = []
        for oi in range(len(regions)):
            ids = matched[oi]
            if len(ids) <= 1:
                continue
            ocr_box = regions[oi].bounding_box
            ox0, oy0, ox1, oy1 = aabb(ocr_box)
            crops, crop_boxes = [], []
            for j, li in enumerate(ids):
                lx0, ly0, lx1, ly1 = aabb(layout_elements[li].bbox)
                x1, y1, x2, y2 = max(ox0, lx0), max(oy0, ly0), min(ox1, lx1), min(oy1, ly1)
                if F(x2 - x1) <= F(1.0) or F(y2 - y1) <= F(1.0):
                    continue
                crop_box = from_coords(x1, y1, x2, y2)
                for k, other in enumerate(regions):                    # text fully covered by this crop is dropped
                    if k != oi and aabb_iou(other.bounding_box, crop_box) > F(0.8):
                        other.text = None
                c = crop_bounding_box(page, crop_box)
                if c is not None:
                    crops.append(c)
                    crop_boxes.append((crop_box, j == 0))
            if not crops:
                continue
            texts, scores = self._recognize(crops)
            for (crop_box, first), text, score in zip(crop_boxes, texts, scores):
                if not text:
                    continue
                if first:
                    r = regions[oi]
                    r.bounding_box, r.dt_poly, r.rec_poly, r.text, r.confidence = crop_box, crop_box, crop_box, text, float(score)
                else:
                    appended.append(api.TextRegion(bounding_box=crop_box, text=text, confidence=float(score), dt_poly=crop_box, rec_poly=crop_box))
        regions.extend(appended)

        for e in layout_elements:                                      # fallback recognition of blocks without text
            if e.element_type in EXCLUDED_FROM_MATCHING or e.element_type in NO_FALLBACK_TYPES:
                continue
            if any(r.text and _overlaps(r.bounding_box, e.bbox, min_pixels) for r in regions):
                continue
            c = crop_bounding_box(page, e.bbox)
            if c is None:
                continue
            r = self.rec.predict([c])
            if r.texts and r.texts[0]:
                box = np.asarray(e.bbox, np.float32).reshape(-1, 2).copy()
                regions.append(api.TextRegion(bounding_box=box, text=r.texts[0], confidence=float(r.scores[0]), dt_poly=box, rec_poly=box))
