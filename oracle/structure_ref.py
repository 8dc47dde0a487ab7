"""Oracle for OARStructure's overall OCR: `run_overall_ocr` (src/oarocr/structure.rs:2208-2540) and
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
