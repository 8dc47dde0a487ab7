"""Oracle for OARStructure's overall OCR: `run_overall_ocr` (src/oarocr/structure.rs:2208-2540),
`precompute_overall_ocr_across_pages` (src/oarocr/structure.rs:2859-3260; `precompute` below, written from the Rust statement by statement -- its
PageOcrState / RecItem records, its four phases -- not from oar_ocr_amd/structure.py) and
`refine_overall_ocr_with_layout` (src/oarocr/structure.rs:1438-1660) restated over the CPU oracle components
(C restatement of pre/post + torch-CPU network interpreter).

TEST INFRASTRUCTURE ONLY.  Layout elements are (points[n,2] f32, type-name) pairs; regions are dicts
{"box", "text", "score"} in the reference's output order."""
from __future__ import annotations

import numpy as np

from . import cpu_ref as R
from .pipeline_ref import OracleClassifier, OracleDetector, OracleRecognizer

f32 = np.float32
SPLIT_IOA = f32(0.3)                       # structure.rs:49
TEXTUAL = {"doc_title", "paragraph_title", "text", "content", "abstract", "header", "footer", "footnote", "number", "reference",
           "reference_content", "algorithm", "aside_text", "list", "figure_title", "table_title", "chart_title",
           "figure_table_chart_title"}     # structure.rs:2281-2303
SPECIALISED = {"formula", "formula_number", "table", "seal"}   # structure.rs:1469-1477


def rect(x1, y1, x2, y2):                  # geometry.rs:98-106
    return np.array([(x1, y1), (x2, y1), (x2, y2), (x1, y2)], np.float32)


def extent(pts):                           # geometry.rs:179-215,569-606
    p = np.asarray(pts, np.float32).reshape(-1, 2)
    xs, ys = p[:, 0], p[:, 1]
    return f32(xs.min()), f32(ys.min()), f32(xs.max()), f32(ys.max())


def shoelace(pts):                         # geometry.rs:141-154
    p = np.asarray(pts, np.float32).reshape(-1, 2)
    if len(p) < 3:
        return f32(0)
    acc = f32(0)
    for k in range(len(p)):
        a, b = p[k], p[(k + 1) % len(p)]
        acc = f32(acc + f32(a[0] * b[1]))
        acc = f32(acc - f32(b[0] * a[1]))
    return f32(abs(acc) / f32(2))


def iou(a, b):                             # geometry.rs:688-717
    a0, a1, a2, a3 = extent(a)
    b0, b1, b2, b3 = extent(b)
    l, t, r, bt = max(a0, b0), max(a1, b1), min(a2, b2), min(a3, b3)
    if l >= r or t >= bt:
        return f32(0)
    inter = f32(f32(r - l) * f32(bt - t))
    if inter <= 0:
        return f32(0)
    union = f32(f32(f32(f32(a2 - a0) * f32(a3 - a1)) + f32(f32(b2 - b0) * f32(b3 - b1))) - inter)
    return f32(inter / union) if union > 0 else f32(0)


def u32(v):                                # Rust `as u32`
    v = float(v)
    if not (v > 0.0):
        return 0
    return int(min(v, 4294967295.0))


def paint(image, boxes, colour=(255, 255, 255)):   # utils/image.rs:709-780
    H, W = image.shape[:2]
    for b in boxes:
        l, t, r, bt = extent(b)
        l, t, r, bt = min(u32(l), W), min(u32(t), H), min(u32(r), W), min(u32(bt), H)
        if l < r and t < bt:
            image[t:bt, l:r, :] = colour


def cut(image, box):                       # utils/bbox_crop.rs:26-71
    p = np.asarray(box, np.float32).reshape(-1, 2)
    if len(p) == 0:
        return None
    H, W = image.shape[:2]
    l, t, r, bt = extent(p)
    l, t = max(l, f32(0)), max(t, f32(0))
    x1, y1 = min(u32(l), max(W - 1, 0)), min(u32(t), max(H - 1, 0))
    x2, y2 = min(u32(r), W), min(u32(bt), H)
    if x2 <= x1 or y2 <= y1:
        return None
    return image[y1:y2, x1:x2].copy()


def touches(a, b, px):
    a0, a1, a2, a3 = extent(a)
    b0, b1, b2, b3 = extent(b)
    return f32(min(a2, b2) - max(a0, b0)) > px and f32(min(a3, b3) - max(a1, b1)) > px


class OracleOverallOCR:
    def __init__(self, det, rec, character_list, line_orientation=None, region_batch_size=64, formula_recognition=False,
                 thresh=0.3, box_thresh=0.6, unclip=1.5, **det_kw):
        self.det = OracleDetector(det, **det_kw)
        self.rec = OracleRecognizer(rec, character_list)
        self.line = OracleClassifier(line_orientation, (80, 160), None) if line_orientation else None
        self.bs = max(region_batch_size, 1)
        self.formula = formula_recognition
        self.p = (thresh, box_thresh, unclip)

    def _read(self, crops):
        texts, scores = [], []
        for s in range(0, len(crops), self.bs):
            r = self.rec.recognize(crops[s:s + self.bs])
            texts += list(r["texts"])
            scores += [float(x) for x in r["scores"]]
        return texts, scores

    def run(self, page, layout, region_blocks=None):
        page = np.ascontiguousarray(page, np.uint8)
        seen = page
        if self.formula:
            formulas = [b for b, t in layout if t in ("formula", "formula_number")]
            if formulas:
                seen = page.copy()
                paint(seen, formulas)
        boxes, _, _ = self.det.detect([seen], *self.p)[0]
        boxes = [np.asarray(b, np.float32).reshape(4, 2) for b in boxes]

        if boxes:
            holders = list(region_blocks) if region_blocks is not None else [b for b, t in layout if t in TEXTUAL]
            if holders:
                pieces = []
                for q in boxes:
                    own = shoelace(q)
                    if own <= 0:
                        pieces.append(q)
                        continue
                    q0, q1, q2, q3 = extent(q)
                    parts = []
                    for h in holders:
                        h0, h1, h2, h3 = extent(h)
                        l, t, r, b = max(q0, h0), max(q1, h1), min(q2, h2), min(q3, h3)
                        if f32(r - l) <= 2 or f32(b - t) <= 2:
                            continue
                        part = rect(l, t, r, b)
                        a = shoelace(part)
                        if a > 0 and f32(a / own) >= SPLIT_IOA:
                            parts.append(part)
                    pieces += parts if len(parts) >= 2 else [q]
                boxes = pieces
            boxes = [boxes[i] for i in R.sort_quad_boxes(np.stack(boxes))]

        out = []
        if boxes:
            got = [(i, R.rotate_crop(page, b)) for i, b in enumerate(boxes)]
            got = [(i, c) for i, c in got if c is not None]
            if got:
                if self.line is not None:
                    for k, (ids, _) in enumerate(self.line.classify([c for _, c in got])):
                        if int(ids[0]) == 1:
                            got[k] = (got[k][0], R.rotate_rgb(got[k][1], 2))
                queue = sorted(got, key=lambda ic: f32(ic[1].shape[1]) / f32(max(ic[1].shape[0], 1)))
                found = {}
                for s in range(0, len(queue), self.bs):
                    part = queue[s:s + self.bs]
                    r = self.rec.recognize([c for _, c in part])
                    for (i, _), text, score in zip(part, r["texts"], r["scores"]):
                        if text != "":
                            found[i] = (text, float(score))
                out = [{"box": boxes[i], "text": found[i][0], "score": found[i][1]} for i in range(len(boxes)) if i in found]
        self.refine(out, layout, page)
        return out

    def precompute(self, prepared_pages, image_batch_size=8, seal_enabled=False):
        """structure.rs:2859-3260.  prepared_pages: list of dicts {"image", "layout", "region_blocks" (None or list of boxes)} or None for a page
        whose slot already holds Err.  Returns a list, per page None (Err / not precomputed) or its regions -- or None altogether when the path
        stands down (:2874-2878, seal detector attached)."""
        if seal_enabled:                                                        # :2874-2878
            return None
        image_batch_size = max(image_batch_size, 1)                             # :2880-2884
        n_pages = len(prepared_pages)
        page_states = [None] * n_pages                                          # Vec<Option<PageOcrState>>  :2901-2902
        rec_items = []                                                          # Vec<RecItem>: dicts page_idx / det_idx / wh_ratio / image
        batched_detection_boxes = [None] * n_pages                              # :2905-2906
        out = [None] * n_pages

        def ocr_image_of(prepared):                                             # :2916-2928 (and again :2969-2981)
            img = np.ascontiguousarray(prepared["image"], np.uint8)
            if self.formula:
                mask_bboxes = [b for b, t in prepared["layout"] if t in ("formula", "formula_number")]
                if mask_bboxes:
                    img = img.copy()
                    paint(img, mask_bboxes)
            return img

        # phase 1 (:2908-2961): detection over chunks of image_batch_size pages
        det_page_indices, det_images = [], []
        for page_idx, prepared in enumerate(prepared_pages):
            if prepared is None:
                continue
            det_page_indices.append(page_idx)
            det_images.append(ocr_image_of(prepared))
        for start in range(0, len(det_page_indices), image_batch_size):
            batch_page_indices = det_page_indices[start:start + image_batch_size]
            det_result = self.det.detect(det_images[start:start + image_batch_size], *self.p)
            for offset, (boxes, _scores, _prob) in enumerate(det_result):
                batched_detection_boxes[batch_page_indices[offset]] = [np.asarray(b, np.float32).reshape(4, 2) for b in boxes]

        # phase 2 (:2963-3128): per page -- split by containers, sort, crop; crops join the document's RecItem list
        for page_idx in range(n_pages):
            prepared = prepared_pages[page_idx]
            if prepared is None:
                continue
            detection_boxes = batched_detection_boxes[page_idx]
            batched_detection_boxes[page_idx] = None                            # .take()
            if detection_boxes is None:                                         # per-page fallback of a failed batch (:2968-3007)
                boxes, _, _ = self.det.detect([ocr_image_of(prepared)], *self.p)[0]
                detection_boxes = [np.asarray(b, np.float32).reshape(4, 2) for b in boxes]
            if detection_boxes:                                                 # :3009-3092
                split_boxes = []
                if prepared.get("region_blocks") is not None:
                    container_boxes = list(prepared["region_blocks"])
                else:
                    container_boxes = [b for b, t in prepared["layout"] if t in TEXTUAL]
                if container_boxes:
                    for bbox in detection_boxes:
                        intersections = []
                        self_area = shoelace(bbox)
                        if self_area <= 0:
                            split_boxes.append(bbox)
                            continue
                        b0, b1, b2, b3 = extent(bbox)
                        for container in container_boxes:
                            c0, c1, c2, c3 = extent(container)
                            ix0, iy0, ix1, iy1 = max(b0, c0), max(b1, c1), min(b2, c2), min(b3, c3)
                            if f32(ix1 - ix0) <= 2 or f32(iy1 - iy0) <= 2:
                                continue
                            inter_bbox = rect(ix0, iy0, ix1, iy1)
                            inter_area = shoelace(inter_bbox)
                            if inter_area <= 0:
                                continue
                            if f32(inter_area / self_area) >= SPLIT_IOA:
                                intersections.append(inter_bbox)
                        if len(intersections) >= 2:
                            split_boxes.extend(intersections)
                        else:
                            split_boxes.append(bbox)
                    detection_boxes = split_boxes
            if detection_boxes:                                                 # :3094-3096
                detection_boxes = [detection_boxes[i] for i in R.sort_quad_boxes(np.stack(detection_boxes))]
            state = {"recognized": [None] * len(detection_boxes), "detection_boxes": detection_boxes}   # :3098-3101
            if state["detection_boxes"]:                                        # :3103-3126, TextCroppingProcessor::new(true) on the UNMASKED page
                page = np.ascontiguousarray(prepared["image"], np.uint8)
                for det_idx, bbox in enumerate(state["detection_boxes"]):
                    img = R.rotate_crop(page, bbox)
                    if img is None:
                        continue
                    wh_ratio = f32(img.shape[1]) / f32(max(img.shape[0], 1))
                    rec_items.append({"page_idx": page_idx, "det_idx": det_idx, "wh_ratio": wh_ratio, "image": img})
            page_states[page_idx] = state

        # phase 3 (:3131-3210): text-line orientation over ALL items, stable sort by wh_ratio, recognition chunks
        if rec_items:
            if self.line is not None:
                for item, (ids, _) in zip(rec_items, self.line.classify([it["image"] for it in rec_items])):
                    if int(ids[0]) == 1:
                        item["image"] = R.rotate_rgb(item["image"], 2)
            rec_items.sort(key=lambda it: it["wh_ratio"])                       # sort_by(partial_cmp): stable
            start = 0
            while start < len(rec_items):
                end = min(start + self.bs, len(rec_items))
                chunk = rec_items[start:end]
                rec_result = self.rec.recognize([it["image"] for it in chunk])
                for i, item in enumerate(chunk):
                    text = rec_result["texts"][i]
                    if text == "":
                        continue
                    page_states[item["page_idx"]]["recognized"][item["det_idx"]] = (text, float(rec_result["scores"][i]))
                start = end

        # phase 4 (:3218-3262): regions in detection order, refined against the page's own layout
        for page_idx in range(n_pages):
            state = page_states[page_idx]
            page_states[page_idx] = None
            if state is None or prepared_pages[page_idx] is None:
                continue
            prepared = prepared_pages[page_idx]
            text_regions = []
            for det_idx, rec in enumerate(state["recognized"]):
                if rec is None:
                    continue
                text_regions.append({"box": state["detection_boxes"][det_idx], "text": rec[0], "score": rec[1]})
            self.refine(text_regions, prepared["layout"], np.ascontiguousarray(prepared["image"], np.uint8))
            out[page_idx] = text_regions
        return out

    def refine(self, regions, layout, page):
        if not regions or not layout:
            return
        three = f32(3)
        eligible = [k for k, (_, t) in enumerate(layout) if t not in SPECIALISED]
        hits = [[k for k in eligible if touches(r["box"], layout[k][0], three)] for r in regions]
        extra = []
        for n in range(len(regions)):
            if len(hits[n]) < 2:
                continue
            b0, b1, b2, b3 = extent(regions[n]["box"])
            crops, where = [], []
            for pos, k in enumerate(hits[n]):
                e0, e1, e2, e3 = extent(layout[k][0])
                l, t, r, b = max(b0, e0), max(b1, e1), min(b2, e2), min(b3, e3)
                if f32(r - l) <= 1 or f32(b - t) <= 1:
                    continue
                piece = rect(l, t, r, b)
                for m, other in enumerate(regions):
                    if m != n and iou(other["box"], piece) > f32(0.8):
                        other["text"] = None
                img = cut(page, piece)
                if img is not None:
                    crops.append(img)
                    where.append((piece, pos == 0))
            if not crops:
                continue
            texts, scores = self._read(crops)
            for (piece, first), text, score in zip(where, texts, scores):
                if text == "":
                    continue
                if first:
                    regions[n].update(box=piece, text=text, score=score)
                else:
                    extra.append({"box": piece, "text": text, "score": score})
        regions += extra
        for box, kind in layout:
            if kind in SPECIALISED or kind in ("image", "chart"):
                continue
            if any(r["text"] and touches(r["box"], box, three) for r in regions):
                continue
            img = cut(page, box)
            if img is None:
                continue
            texts, scores = self._read([img])
            if texts and texts[0] != "":
                regions.append({"box": np.asarray(box, np.float32).reshape(-1, 2).copy(), "text": texts[0], "score": scores[0]})
