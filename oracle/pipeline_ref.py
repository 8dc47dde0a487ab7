"""Oracle for the whole path: the reference's OAROCR::predict (src/oarocr/ocr.rs:518-659) orchestrated in
Python over the C restatement (oracle/oar_oracle.c) and the torch-CPU network interpreter (oracle/onnx_ref.py).

TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import numpy as np

from . import cpu_ref as R
from . import onnx_ref


class OracleDetector:
    """TextDetectionAdapter::execute -> DBModel::forward (models/detection/db.rs:281-335)."""

    def __init__(self, onnx_bytes, limit_side_len=None, limit_type=None, max_side_limit=4000, max_candidates=1000, text_type=None):
        self.model = onnx_ref.parse_model(onnx_bytes)
        self.input = self.model["inputs"][0]
        # text type "seal": 736 / min and BoxType::Poly (text_detection_adapter.rs:131-150, preprocessing.rs:44-62)
        self.seal = (text_type or "").lower() == "seal"
        self.cfg = (limit_side_len or (736 if self.seal else 960), limit_type or ("min" if self.seal else "max"), max_side_limit)
        self.max_candidates = max_candidates

    def prob_maps(self, images):
        """Returns per image (prob[H,W], (src_h, src_w)) with shape grouping as db.rs:297-309."""
        pre = [R.det_preprocess(im, *self.cfg) for im in images]
        out = [None] * len(images)
        groups = {}
        for i, (t, _) in enumerate(pre):
            groups.setdefault(t.shape, []).append(i)
        for shape, idx in groups.items():
            x = np.stack([pre[i][0] for i in idx])
            y = onnx_ref.run(self.model, {self.input: x})[0]
            for k, i in enumerate(idx):
                out[i] = (y[k, 0], pre[i][1])
        return out

    def detect(self, images, thresh=0.3, box_thresh=0.6, unclip=1.5):
        res = []
        for prob, (sh, sw) in self.prob_maps(images):
            if self.seal:
                from . import poly_ref
                boxes, scores = poly_ref.db_postprocess_poly(prob, sh, sw, thresh, box_thresh, unclip, self.max_candidates)
            else:
                boxes, scores = R.db_postprocess(prob, sh, sw, thresh, box_thresh, unclip, self.max_candidates)
            res.append((boxes, scores, prob))
        return res


class OracleRecognizer:
    """TextRecognitionAdapter::execute -> CRNNModel::forward_refs (models/recognition/crnn.rs:247-293)."""

    def __init__(self, onnx_bytes, character_list, rec_image_shape=(3, 48, 320), max_img_w=3200):
        self.model = onnx_ref.parse_model(onnx_bytes)
        self.input = self.model["inputs"][0]
        self.charset = R.ctc_charset([s[0] for s in character_list if len(s) > 0], use_space_char=True)
        self.shape = rec_image_shape
        self.max_img_w = max_img_w

    pool = None   # optional concurrent.futures executor: per-crop preprocessing in parallel (bench.py's cpu_baseline leg)

    def probs(self, crops, batch_max_wh_ratio=None):
        x = R.rec_preprocess(crops, self.shape[1], self.shape[2], self.max_img_w, batch_max_wh_ratio, pool=self.pool)
        return onnx_ref.run(self.model, {self.input: x})[0], x

    def recognize(self, crops, batch_max_wh_ratio=None):
        p, x = self.probs(crops, batch_max_wh_ratio)
        n, T, V = p.shape
        idx, pr = R.argmax_rows(p)
        texts, scores, pos, cols, lens = R.ctc_decode(idx, pr, n, T, self.charset)
        return {"texts": texts, "scores": scores, "cols": cols, "idx": idx.reshape(n, T), "prob": pr.reshape(n, T), "probs_full": p, "Wt": x.shape[3]}


class OracleClassifier:
    """DocumentOrientationAdapter / TextLineOrientationAdapter -> PPLCNetModel::forward_refs (pp_lcnet.rs:139-330)."""

    def __init__(self, onnx_bytes, input_hw=(224, 224), resize_short=256, topk=1):
        self.model = onnx_ref.parse_model(onnx_bytes)
        self.input = self.model["inputs"][0]
        self.input_hw, self.resize_short, self.topk = input_hw, resize_short, topk

    def preprocess(self, images):
        return np.stack([R.cls_preprocess(im, self.input_hw, self.resize_short) for im in images])

    def probs(self, images):
        return onnx_ref.run(self.model, {self.input: self.preprocess(images)})[0]

    def classify(self, images):
        """[(class_ids[topk], scores[topk])] per image (utils/topk.rs:181-199)."""
        return [R.topk(row, min(self.topk, row.size)) for row in self.probs(images)]


class OracleRectifier:
    """UVDocRectifierAdapter -> UVDocModel::forward (uvdoc.rs:82-109,166-207)."""

    def __init__(self, onnx_bytes, target_hw=(512, 512)):
        self.model = onnx_ref.parse_model(onnx_bytes)
        self.input = self.model["inputs"][0]
        self.target_hw = target_hw

    def rectify(self, images):
        out = []
        for im in images:
            x = R.uvdoc_preprocess(im, self.target_hw)[None]
            y = onnx_ref.run(self.model, {self.input: x})[0]
            out.append(R.uvdoc_postprocess(y[0], (im.shape[1], im.shape[0])))
        return out


class OracleLayoutDetector:
    """LayoutDetectionAdapter's model half for the PicoDet / RT-DETR / PP-DocLayout graphs: ScaleAwareDetectorModel::forward
    (models/detection/scale_aware_detector.rs:169-440) + LayoutPostProcess::apply (processors/layout_postprocess.rs:60-97)."""

    def __init__(self, onnx_bytes, num_classes=5, model_type="picodet", image_shape=(800, 608), score_threshold=0.5, nms_threshold=0.5, max_detections=100):
        self.model = onnx_ref.parse_model(onnx_bytes)
        self.inputs = self.model["inputs"]
        self.kw = dict(filter="catmullrom", bgr=False, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0)) if model_type == "pp-doclayout" else dict(filter="lanczos3", bgr=True)
        self.image_shape, self.num_classes, self.model_type = image_shape, num_classes, model_type
        self.post = (score_threshold, nms_threshold, max_detections)

    def preprocess(self, image):
        return R.layout_preprocess(image, self.image_shape, **self.kw)

    def predictions(self, images):
        pre = [self.preprocess(im) for im in images]
        x = np.stack([p[0] for p in pre])
        feeds = {"image": x, "scale_factor": np.array([[np.float32(p[2][0]) / np.float32(im.shape[0]), np.float32(p[2][1]) / np.float32(im.shape[1])] for p, im in zip(pre, images)], np.float32)}
        if "im_shape" in self.inputs:
            feeds["im_shape"] = np.array([[p[2][0], p[2][1]] for p in pre], np.float32)
        y = onnx_ref.run(self.model, feeds)[0]
        n = len(images)
        return y.reshape(n, -1, y.shape[-1])            # [n, boxes, feat] (the reference's [n, boxes, 1, feat])

    def detect(self, images):
        y = self.predictions(images)
        return [R.layout_postprocess(y[i], images[i].shape[1], images[i].shape[0], self.num_classes, *self.post, model_type=self.model_type) for i in range(len(images))], y


class OracleOCR:
    def __init__(self, det, rec, character_list, thresh=0.3, box_thresh=0.6, unclip=2.0, image_batch_size=8, region_batch_size=64,
                 max_pooled_crops=4096, doc_orientation=None, rectifier=None, line_orientation=None, threads=0, **det_kw):
        """threads > 0: crops are cut and recognizer inputs packed on a thread pool of that size -- the places where the reference's own CPU
        path fans out over its rayon pool (TextCroppingProcessor >= 16 boxes, src/oarocr/processors.rs:113-131; CRNN preprocess per crop,
        crnn.rs:98-121).  Results are identical; only bench.py's cpu_baseline leg sets it."""
        self.det = OracleDetector(det, **det_kw)
        self.rec = OracleRecognizer(rec, character_list)
        self.p = (thresh, box_thresh, unclip)
        self.region_bs = region_batch_size
        self.max_pool = max_pooled_crops
        self.doc_ori = OracleClassifier(doc_orientation) if doc_orientation else None
        self.rect = OracleRectifier(rectifier) if rectifier else None
        self.line_ori = OracleClassifier(line_orientation, (80, 160), None) if line_orientation else None
        self.page_meta = []
        self.pool = None
        if threads > 0:
            from concurrent.futures import ThreadPoolExecutor
            self.pool = ThreadPoolExecutor(max_workers=threads)
            self.rec.pool = self.pool

    def preprocess(self, image):
        """DocumentPreprocessor::preprocess (src/oarocr/preprocess.rs:59-97): (image, angle | None, rotation | None, rectified)."""
        cur, angle, rotation = image, None, None
        if self.doc_ori is not None:
            ids, _ = self.doc_ori.classify([image])[0]
            cur, rotation = R.correct_orientation(image, int(ids[0]))
            angle = rotation[0]
        rectified = False
        if self.rect is not None:
            cur = self.rect.rectify([cur])[0]
            rectified = True
            rotation = None
        return cur, angle, rotation, rectified

    def predict(self, images):
        """With optional stages attached: slots carry boxes mapped back to the input page unless it was rectified
        (ocr.rs:644-646, 898-925); page_meta[i] = (orientation_angle | None, rectified)."""
        if len(images) == 0:   # ocr.rs:525-532: validation error "images must be a non-empty slice" (round 6: tools/edge_pages.py found the oracle laxer than the reference)
            raise ValueError("OCR Pipeline: images must be a non-empty slice")
        pre = [self.preprocess(im) for im in images]
        self.page_meta = [(p[1], p[3]) for p in pre]
        res = self._predict_core([p[0] for p in pre])
        for slots, (_, _, rotation, _) in zip(res, pre):
            if rotation is None:
                continue
            for s in slots:
                s["box"] = R.rotate_back_points(s["box"], rotation[0], rotation[1], rotation[2]).reshape(-1, 2)
        return res

    def _predict_core(self, images):
        dets = self.det.detect(images, *self.p)
        per_image = []
        pool = []
        for img_idx, (boxes, scores, _) in enumerate(dets):
            if self.det.seal:   # sort_detection_boxes / crop_single for polygons (ocr.rs:699-716, processors.rs:96-102)
                from . import poly_ref
                order = poly_ref.sort_poly_boxes(boxes)
            else:
                order = R.sort_quad_boxes(boxes)
            slots = []

            def cut(o, img_idx=img_idx, boxes=boxes):
                if self.det.seal and len(boxes[o]) != 4:
                    return poly_ref.crop_bounding_box(images[img_idx], boxes[o])
                return R.rotate_crop(images[img_idx], boxes[o])
            crops = list(self.pool.map(cut, order)) if self.pool is not None and len(order) >= 16 else [cut(o) for o in order]
            for k, o in enumerate(order):
                crop = crops[k]
                slots.append({"box": boxes[o].copy(), "det_score": float(scores[o]), "filled": False})
                if crop is None:
                    continue
                slots[-1]["crop_wh"] = (crop.shape[1], crop.shape[0])
                pool.append((img_idx, k, crop, np.float32(crop.shape[1]) / np.float32(max(crop.shape[0], 1))))
                if len(pool) >= self.max_pool:
                    self._flush(pool, per_image + [slots])
                    pool = []
            per_image.append(slots)
        self._flush(pool, per_image)
        return [[s for s in slots if s["filled"]] for slots in per_image]

    def _flush(self, pool, per_image):
        if not pool:
            return
        line_angle = [None] * len(pool)
        if self.line_ori is not None:   # classify_line_orientations (ocr.rs:757-790): class 1 => rotate180 of the crop
            cls = self.line_ori.classify([c[2] for c in pool])
            for i, (ids, _) in enumerate(cls):
                line_angle[i] = float(ids[0]) * 180.0
                if int(ids[0]) == 1:
                    pool[i] = (pool[i][0], pool[i][1], R.rotate_rgb(pool[i][2], 2), pool[i][3])
            for i, (img_idx, k, _, _) in enumerate(pool):
                per_image[img_idx][k]["line_angle"] = line_angle[i]
        order = sorted(range(len(pool)), key=lambda i: pool[i][3])   # stable, like Rust sort_by
        for c0 in range(0, len(order), self.region_bs):
            chunk = [pool[i] for i in order[c0:c0 + self.region_bs]]
            r = self.rec.recognize([c[2] for c in chunk])
            # chunk_max_wh_ratio as ocr.rs:828-831 folds it: from the recognizer's base ratio (rec_image_shape w / h) over the chunk's crops, in f32
            base = np.float32(self.rec.shape[2]) / np.float32(self.rec.shape[1])
            chunk_max = max([base] + [c[3] for c in chunk])
            for j, (img_idx, k, _, wh) in enumerate(chunk):
                s = per_image[img_idx][k]
                s.update(filled=True, text=r["texts"][j], score=r["scores"][j], idx=r["idx"][j], prob=r["prob"][j],
                         probs_full=r["probs_full"][j], cols=r["cols"][j], wh_ratio=wh, max_wh_ratio=chunk_max)


def compare_results(got, ref, prob_tol=1e-3, tie_tol=1e-5):
    """got: api.OAROCRResult; ref: list of oracle slots for the same image.
    Boxes must be bit-exact; CTC indices must match except where the oracle's own top-2 probabilities are within
    tie_tol of each other (a genuine tie under the 1e-3 float budget); max-probabilities within prob_tol."""
    rep = {"ok": True, "n_regions": (len(got.text_regions), len(ref)), "box_mismatch": 0, "idx_mismatch": 0, "idx_tie": 0, "max_prob_diff": 0.0,
           "text_equal": 0}
    if len(got.text_regions) != len(ref):
        rep["ok"] = False
        return rep
    for g, r in zip(got.text_regions, ref):
        if not np.array_equal(np.asarray(g.bounding_box, np.float32), np.asarray(r["box"], np.float32)):
            rep["box_mismatch"] += 1
        if g.text == r["text"]:
            rep["text_equal"] += 1
        rep["max_prob_diff"] = max(rep["max_prob_diff"], abs(g.confidence - r["score"]))
    rep["ok"] = rep["box_mismatch"] == 0 and rep["max_prob_diff"] <= prob_tol
    return rep
