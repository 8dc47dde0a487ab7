"""ctypes front-end for the oracle's C restatement (oracle/oar_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  Nothing under oar_ocr_amd/ may import this module.

Every helper mirrors one reference function; the citation (file:line under
/root/reference) lives on the C function it calls.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_LIB = None


def build(force: bool = False) -> Path:
    so = _HERE / "liboar_oracle.so"
    src = _HERE / "oar_oracle.c"
    if force or not so.exists() or (src.exists() and so.stat().st_mtime < src.stat().st_mtime):
        subprocess.check_call(["make", "-C", str(_HERE), "-s", "liboar_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(str(build()))
        L = _LIB
        L.orc_find_contours.restype = C.c_void_p
        L.orc_find_contours.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_contours_count.restype = C.c_long
        L.orc_contours_count.argtypes = [C.c_void_p]
        L.orc_contours_npts.restype = C.c_long
        L.orc_contours_npts.argtypes = [C.c_void_p]
        L.orc_contours_copy.argtypes = [C.c_void_p] * 5
        L.orc_contours_free.argtypes = [C.c_void_p]
        L.orc_box_score_fast.restype = C.c_float
        L.orc_box_score_fast.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int]
        L.orc_unclip.restype = C.c_int
        L.orc_unclip.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_int]
        L.orc_boxes_from_bitmap.restype = C.c_int
        L.orc_boxes_from_bitmap.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_uint32, C.c_uint32,
                                            C.c_float, C.c_float, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_int]
        L.orc_threshold_mask.argtypes = [C.c_void_p, C.c_long, C.c_float, C.c_void_p]
        L.orc_argmax_rows.argtypes = [C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_void_p]
        L.orc_ctc_collapse.restype = C.c_int
        L.orc_ctc_collapse.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_normalize_chw.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_normalize_hwc.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_alpha_beta.argtypes = [C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_normalize_crnn_chw.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_det_resize_dims.restype = C.c_int
        L.orc_det_resize_dims.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_resize_triangle_rgb.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_convex_hull.restype = C.c_int
        L.orc_convex_hull.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_mini_box_from_points.restype = C.c_int
        L.orc_mini_box_from_points.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_simplify_chain.restype = C.c_int
        L.orc_simplify_chain.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_sort_quad_boxes.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.orc_crop_plan.restype = C.c_int
        L.orc_crop_plan.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_crop_exec.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_bicubic_sample.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p]
        L.orc_rec_tensor_width.restype = C.c_int
        L.orc_rec_tensor_width.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_perspective_transform.restype = C.c_int
        L.orc_perspective_transform.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_inverse3.restype = C.c_int
        L.orc_inverse3.argtypes = [C.c_void_p, C.c_void_p]
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---------------------------------------------------------------- normalize (a4)
DB_MEAN = (0.485, 0.456, 0.406)
DB_STD = (0.229, 0.224, 0.225)


def alpha_beta(scale, mean, std):
    a = np.zeros(3, np.float32)
    b = np.zeros(3, np.float32)
    lib().orc_alpha_beta(C.c_float(scale), _p(_f32(mean)), _p(_f32(std)), _p(a), _p(b))
    return a, b


def normalize(rgb: np.ndarray, alpha, beta, src=(0, 1, 2), layout="chw") -> np.ndarray:
    """rgb: [H,W,3] u8.  Returns [3,H,W] (chw) or [H,W,3] (hwc) f32."""
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    src_a = np.asarray(src, dtype=np.int32)
    alpha, beta = _f32(alpha), _f32(beta)
    if layout == "chw":
        out = np.empty((3, h, w), np.float32)
        lib().orc_normalize_chw(_p(rgb), w, h, _p(src_a), _p(alpha), _p(beta), _p(out))
    else:
        out = np.empty((h, w, 3), np.float32)
        lib().orc_normalize_hwc(_p(rgb), w, h, _p(src_a), _p(alpha), _p(beta), _p(out))
    return out


def db_normalize(rgb: np.ndarray) -> np.ndarray:
    """DB detector input: scale 1/255, ImageNet mean/std applied in OUTPUT (BGR) order
    (models/detection/db.rs:404-415). Returns [3,H,W]."""
    a, b = alpha_beta(np.float32(1.0) / np.float32(255.0), DB_MEAN, DB_STD)
    return normalize(rgb, a, b, src=(2, 1, 0), layout="chw")


def crnn_normalize(resized_rgb: np.ndarray, tensor_w: int) -> np.ndarray:
    resized_rgb = np.ascontiguousarray(resized_rgb, dtype=np.uint8)
    h, rw, _ = resized_rgb.shape
    out = np.zeros((3, h, tensor_w), np.float32)
    lib().orc_normalize_crnn_chw(_p(resized_rgb), rw, h, tensor_w, _p(out))
    return out


# ---------------------------------------------------------------- resize (a3, a16)
def det_resize_dims(w, h, limit_side_len=960, limit_type="max", max_side_limit=4000):
    lt = {"max": 0, "min": 1, "resize_long": 2}[limit_type]
    hw = np.zeros(2, np.uint32)
    ratios = np.zeros(2, np.float32)
    need = lib().orc_det_resize_dims(w, h, limit_side_len, lt, max_side_limit, _p(hw), _p(ratios))
    return bool(need), int(hw[0]), int(hw[1]), float(ratios[0]), float(ratios[1])


def resize_triangle(rgb: np.ndarray, nw: int, nh: int) -> np.ndarray:
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    out = np.empty((nh, nw, 3), np.uint8)
    lib().orc_resize_triangle_rgb(_p(rgb), w, h, nw, nh, _p(out))
    return out


def det_preprocess(rgb: np.ndarray, limit_side_len=960, limit_type="max", max_side_limit=4000):
    """resize_detection.rs:162-221 + db.rs normalize. Returns (tensor[3,H,W], (src_h, src_w))."""
    h, w, _ = rgb.shape
    img = rgb
    if h + w < 64:  # image_padding
        nw_, nh_ = max(w, 32), max(h, 32)
        pad = np.zeros((nh_, nw_, 3), np.uint8)
        pad[:h, :w] = rgb
        img = pad
    ih, iw, _ = img.shape
    need, rh, rw, _, _ = det_resize_dims(iw, ih, limit_side_len, limit_type, max_side_limit)
    if need:
        img = resize_triangle(img, rw, rh)
    return db_normalize(img), (h, w)


def rec_tensor_width(sizes, img_h=48, img_w=320, max_img_w=3200):
    ws = np.asarray([s[0] for s in sizes], np.int32)
    hs = np.asarray([s[1] for s in sizes], np.int32)
    rw = np.zeros(len(sizes), np.int32)
    tw = lib().orc_rec_tensor_width(_p(ws), _p(hs), len(sizes), img_h, img_w, max_img_w, _p(rw))
    return tw, rw


def rec_preprocess(crops, img_h=48, img_w=320, max_img_w=3200, batch_max_wh_ratio=None, pool=None) -> np.ndarray:
    """models/recognition/crnn.rs:71-125. crops: list of [h,w,3] u8. Returns [n,3,img_h,Wt] f32.
    batch_max_wh_ratio: the crops are PART of a larger batch whose widest member has this w/h ratio (the tensor width of a
    batch is set by its widest crop, crnn.rs:80-87) -- used to check a few pages of a big pooled run."""
    if not crops:
        return np.zeros((0, 0, 0, 0), np.float32)
    tw, rws = rec_tensor_width([(c.shape[1], c.shape[0]) for c in crops], img_h, img_w, max_img_w)
    if batch_max_wh_ratio is not None:
        tw = min(int(np.float32(img_h) * np.float32(batch_max_wh_ratio)), max_img_w)   # `(img_h as f32 * max_wh) as usize` (crnn.rs:87)
        rws = np.array([min(int(np.ceil(np.float32(img_h) * (np.float32(c.shape[1]) / np.float32(c.shape[0])))), tw) for c in crops], np.int32)
    out = np.zeros((len(crops), 3, img_h, tw), np.float32)

    def one(i):
        out[i] = crnn_normalize(resize_triangle(crops[i], int(rws[i]), img_h), tw)
    if pool is not None:   # the reference resizes / normalises the crops of a batch on its rayon pool (crnn.rs:98-121); ctypes calls release the GIL
        list(pool.map(one, range(len(crops))))
    else:
        for i in range(len(crops)):
            one(i)
    return out


# ---------------------------------------------------------------- DB postprocess (a7..a12)
def threshold_mask(pred: np.ndarray, thresh: float) -> np.ndarray:
    pred = _f32(pred)
    mask = np.empty(pred.shape, np.uint8)
    lib().orc_threshold_mask(_p(pred), pred.size, C.c_float(thresh), _p(mask))
    return mask


def find_contours(mask: np.ndarray):
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    h, w = mask.shape
    L = lib()
    cs = L.orc_find_contours(_p(mask), w, h)
    n, npts = L.orc_contours_count(cs), L.orc_contours_npts(cs)
    offs = np.zeros(n + 1, np.int64)
    pts = np.zeros((max(npts, 1), 2), np.int32)
    bt = np.zeros(max(n, 1), np.int32)
    par = np.zeros(max(n, 1), np.int32)
    L.orc_contours_copy(cs, _p(offs), _p(pts), _p(bt), _p(par))
    L.orc_contours_free(cs)
    return [(pts[offs[i]:offs[i + 1]].copy(), int(bt[i]), int(par[i])) for i in range(n)]


def box_score_fast(pred: np.ndarray, box: np.ndarray) -> float:
    pred = _f32(pred)
    box = _f32(box).reshape(-1, 2)
    h, w = pred.shape
    return float(lib().orc_box_score_fast(_p(pred), h, w, _p(box), box.shape[0]))


def unclip(box: np.ndarray, ratio: float) -> np.ndarray:
    box = _f32(box).reshape(-1, 2)
    out = np.zeros((1024, 2), np.float32)
    n = lib().orc_unclip(_p(box), box.shape[0], C.c_float(ratio), _p(out), 1024)
    return out[:max(n, 0)].copy()


def mini_box(points: np.ndarray):
    pts = _f32(points).reshape(-1, 2)
    out = np.zeros((4, 2), np.float32)
    ms = C.c_float(0)
    ok = lib().orc_mini_box_from_points(_p(pts), pts.shape[0], _p(out), C.byref(ms))
    return (out, float(ms.value)) if ok else None


def simplify_chain(points: np.ndarray) -> np.ndarray:
    pts = _f32(points).reshape(-1, 2)
    out = np.zeros_like(pts)
    n = lib().orc_simplify_chain(_p(pts), pts.shape[0], _p(out))
    return out[:n].copy()


def convex_hull(points: np.ndarray) -> np.ndarray:
    pts = _f32(points).reshape(-1, 2)
    out = np.zeros_like(pts)
    n = lib().orc_convex_hull(_p(pts), pts.shape[0], _p(out))
    return out[:n].copy()


def dilate3x3(mask: np.ndarray) -> np.ndarray:
    """processors/db_mask.rs:11"""
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    out = np.empty_like(mask)
    lib().orc_dilate3x3(_p(mask), mask.shape[0], mask.shape[1], _p(out))
    return out


def boxes_from_bitmap(pred, mask, dest_w, dest_h, box_thresh=0.6, unclip_ratio=1.5, max_candidates=1000, min_size=3.0, score_mode="fast"):
    pred = _f32(pred)
    mask = np.ascontiguousarray(mask, dtype=np.uint8)
    h, w = pred.shape
    cap = max_candidates
    boxes = np.zeros((cap, 4, 2), np.float32)
    scores = np.zeros(cap, np.float32)
    n = lib().orc_boxes_from_bitmap_ex(_p(pred), _p(mask), h, w, dest_w, dest_h, C.c_float(box_thresh),
                                       C.c_float(unclip_ratio), max_candidates, C.c_float(min_size), 1 if score_mode == "slow" else 0, _p(boxes), _p(scores), cap)
    return boxes[:n].copy(), scores[:n].copy()


def db_postprocess(pred, src_h, src_w, thresh=0.3, box_thresh=0.6, unclip_ratio=1.5, max_candidates=1000, score_mode="fast", use_dilation=False):
    """processors/db_postprocess.rs:134-183 (BoxType::Quad). pred: [H,W] f32."""
    mask = threshold_mask(pred, thresh)
    if use_dilation:
        mask = dilate3x3(mask)
    return boxes_from_bitmap(pred, mask, int(src_w), int(src_h), box_thresh, unclip_ratio, max_candidates, score_mode=score_mode)


def sort_quad_boxes(boxes: np.ndarray) -> np.ndarray:
    """Returns the permutation (processors/sorting.rs:35-84)."""
    boxes = _f32(boxes).reshape(-1, 8)
    order = np.zeros(boxes.shape[0], np.int32)
    if boxes.shape[0]:
        lib().orc_sort_quad_boxes(_p(boxes), boxes.shape[0], _p(order))
    return order


# ---------------------------------------------------------------- crop (a14)
def rotate_crop(img: np.ndarray, box: np.ndarray):
    """utils/transform.rs:76-191. Returns the crop [h,w,3] u8 or None when the reference errors."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, _ = img.shape
    box = _f32(box).reshape(4, 2)
    plan = np.zeros(7, np.int32)
    inv = np.zeros(9, np.float32)
    mode = lib().orc_crop_plan(w, h, _p(box), _p(plan), _p(inv))
    if mode == 0:
        return None
    out = np.empty((int(plan[5]), int(plan[4]), 3), np.uint8)
    lib().orc_crop_exec(_p(img), w, h, mode, _p(plan), _p(inv), _p(out))
    return out


def bicubic_sample(img: np.ndarray, x: float, y: float) -> np.ndarray:
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, _ = img.shape
    out = np.zeros(3, np.uint8)
    lib().orc_bicubic_sample(_p(img), w, h, C.c_float(x), C.c_float(y), _p(out))
    return out


# ---------------------------------------------------------------- CTC (a18, a19)
def argmax_rows(probs: np.ndarray):
    probs = _f32(probs)
    v = probs.shape[-1]
    rows = probs.size // v if v else 0
    idx = np.zeros(rows, np.int64)
    p = np.zeros(rows, np.float32)
    lib().orc_argmax_rows(_p(probs), rows, v, _p(idx), _p(p))
    return idx, p


def read_dict_lines(text: str):
    """decode.rs:120 + ocr.rs:277-291: one entry per line, first char only, empty lines vanish."""
    parts = text.split("\n")   # Rust str::lines(): "\n" / "\r\n" terminate a line, nothing else does (ocr.rs:386)
    lines = [p[:-1] if p.endswith("\r") else p for p in parts[:-1]] + ([parts[-1]] if parts[-1] != "" else [])
    return [ln[0] for ln in lines if len(ln) > 0]


def ctc_charset(dict_chars, use_space_char=True):
    """decode.rs:391-421 (has_explicit_blank=false): ['\\0'] + chars + [' ']"""
    chars = list(dict_chars)
    if use_space_char:
        chars.append(" ")
    return ["\0"] + chars


def ctc_decode(idx: np.ndarray, prob: np.ndarray, n: int, T: int, charset):
    """decode.rs:505-614. Returns texts, scores, positions, cols, seq_lens."""
    idx = np.ascontiguousarray(idx, np.int64).reshape(n, T) if n * T else np.zeros((0, 0), np.int64)
    prob = _f32(prob).reshape(n, T) if n * T else np.zeros((0, 0), np.float32)
    texts, scores, positions, cols, lens = [], [], [], [], []
    if n == 0 or T == 0:
        return texts, scores, positions, cols, lens
    kc = np.zeros(T, np.int32)
    ki = np.zeros(T, np.int64)
    for b in range(n):
        sc = C.c_float(0)
        k = lib().orc_ctc_collapse(_p(idx[b]), _p(prob[b]), T, len(charset), _p(kc), _p(ki), C.byref(sc))
        texts.append("".join(charset[int(i)] for i in ki[:k]))
        scores.append(float(np.float32(sc.value)))
        cols.append([int(c) for c in kc[:k]])
        positions.append([float(np.float32(c) / np.float32(T)) for c in kc[:k]])
        lens.append(T)
    return texts, scores, positions, cols, lens


# ---------------------------------------------------------------- host policy (a2, a21)
def default_cpu_region_batch_size(model_name: str | None) -> int:
    """src/oarocr/builder_utils.rs:111-125"""
    return 16 if (model_name and "tiny" in model_name.lower()) else 4


def resolve_device_batch_sizes(user_image, user_region, accelerated: bool, model_name):
    """src/oarocr/builder_utils.rs:86-102"""
    if accelerated:
        return user_image, user_region
    return (user_image if user_image is not None else 1,
            user_region if user_region is not None else default_cpu_region_batch_size(model_name))


def is_cjk(ch: str) -> bool:
    c = ord(ch)
    return (0x4E00 <= c <= 0x9FFF) or (0x3400 <= c <= 0x4DBF) or (0x3040 <= c <= 0x30FF) or (0xAC00 <= c <= 0xD7AF)


# ---------------------------------------------------------------------------------------------- config 5 (a22 / a23)
def cls_resize_dims(w, h, resize_short, crop_w, crop_h):
    out = np.zeros(4, np.uint32)
    lib().orc_cls_resize_dims(int(w), int(h), int(resize_short), int(crop_w), int(crop_h), _p(out))
    return tuple(int(v) for v in out)


IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def cls_preprocess(rgb: np.ndarray, input_hw=(224, 224), resize_short=256) -> np.ndarray:
    """pp_lcnet.rs:139-196: (short-edge resize + centre crop) or direct resize, Triangle; ImageNet normalisation, RGB, CHW."""
    h, w, _ = rgb.shape
    ch, cw = input_hw
    nw, nh, x1, y1 = cls_resize_dims(w, h, resize_short or 0, cw, ch)
    img = resize_triangle(rgb, nw, nh) if (nw, nh) != (w, h) else rgb
    img = np.ascontiguousarray(img[y1:y1 + ch, x1:x1 + cw])
    a, b = alpha_beta(1.0 / 255.0, IMAGENET_MEAN, IMAGENET_STD)
    return normalize(img, a, b, src=(0, 1, 2), layout="chw")


def topk(scores: np.ndarray, k: int):
    scores = _f32(scores)
    idx = np.zeros(k, np.int32); sc = np.zeros(k, np.float32)
    lib().orc_topk(_p(scores), scores.size, k, _p(idx), _p(sc))
    return idx, sc


def rotate_rgb(rgb: np.ndarray, quarter: int) -> np.ndarray:
    """image::imageops::rotate90/180/270: clockwise by quarter * 90 degrees."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w, _ = rgb.shape
    out = np.zeros((w, h, 3) if quarter in (1, 3) else (h, w, 3), np.uint8)
    lib().orc_rotate_rgb(_p(rgb), w, h, int(quarter), _p(out))
    return out


def correct_orientation(rgb: np.ndarray, class_id):
    """src/oarocr/preprocess.rs:109-141: class 1 -> rotate270, 2 -> rotate180, 3 -> rotate90; returns (image, (angle, rw, rh))."""
    if class_id is None:
        return rgb, None
    q = {1: 3, 2: 2, 3: 1}.get(int(class_id), 0)
    out = rotate_rgb(rgb, q) if q else rgb
    return out, (float(class_id) * 90.0, out.shape[1], out.shape[0])


def rotate_back_points(pts: np.ndarray, angle: float, rotated_w: int, rotated_h: int) -> np.ndarray:
    p = np.ascontiguousarray(pts, np.float32).copy()
    lib().orc_rotate_back_points(_p(p), p.size // 2, C.c_float(angle), int(rotated_w), int(rotated_h))
    return p


def uvdoc_preprocess(rgb: np.ndarray, target_hw=(512, 512)) -> np.ndarray:
    """uvdoc.rs:82-109 + :296-303: Triangle resize to the target, then v/255 in BGR plane order (no mean shift), CHW."""
    h, w, _ = rgb.shape
    th, tw = target_hw
    img = resize_triangle(rgb, tw, th) if (th > 0 and tw > 0 and (w, h) != (tw, th)) else rgb
    a, b = alpha_beta(1.0 / 255.0, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0))
    return normalize(img, a, b, src=(2, 1, 0), layout="chw")


def uvdoc_postprocess(pred_chw: np.ndarray, orig_wh) -> np.ndarray:
    """uvdoc.rs:166-207: BGR planes * 255, clamp, truncate -> RGB8; Triangle resize back to the original size."""
    c, h, w = pred_chw.shape
    pl = [np.ascontiguousarray(pred_chw[i], np.float32) for i in range(3)]
    out = np.zeros((h, w, 3), np.uint8)
    lib().orc_bgr_planes_to_rgb(_p(pl[0]), _p(pl[1]), _p(pl[2]), h * w, C.c_float(255.0), _p(out))
    ow, oh = orig_wh
    if ow and oh and (ow, oh) != (w, h):
        out = resize_triangle(out, ow, oh)
    return out


# ------------------------------------------------------------------------------------------------ a21 word boxes
def _is_cjk(ch: str) -> bool:
    """OAROCR::is_cjk (src/oarocr/ocr.rs:1075-1082)"""
    u = ord(ch)
    return (0x4E00 <= u <= 0x9FFF) or (0x3400 <= u <= 0x4DBF) or (0x20000 <= u <= 0x2A6DF) or (0x2A700 <= u <= 0x2B73F) or (0x2B740 <= u <= 0x2B81F)


def ctc_word_boxes(line_bbox: np.ndarray, text: str, col_indices, seq_len: int, wh_ratio: float, max_wh_ratio: float):
    """OAROCR::ctc_word_boxes, src/oarocr/ocr.rs:949-1020 (f32 arithmetic, same operation order); checker for oar_ctc_word_boxes."""
    if not col_indices or seq_len == 0 or not text:
        return []
    f = np.float32
    EPS = f(1.1920929e-7)
    eff = f(f(seq_len) * f(f(wh_ratio) / f(max_wh_ratio)))
    if eff <= EPS:
        return []
    xs, ys = np.asarray(line_bbox, np.float32)[:, 0], np.asarray(line_bbox, np.float32)[:, 1]
    x_min, y_min, x_max, y_max = f(xs.min()), f(ys.min()), f(xs.max()), f(ys.max())
    width = f(x_max - x_min)
    cell = f(width / max(eff, EPS))
    chars = list(text)
    avg_w = f(width / f(max(len(chars), 1)))
    centers = [f(x_min + f(f(f(i) + f(0.5)) * cell)) for i in col_indices]
    boxes = []
    n = len(col_indices)
    for i in range(n):
        ch = chars[i] if i < len(chars) else "?"
        c = centers[i]
        if _is_cjk(ch):
            half = f(avg_w / f(2.0))
            l, r = max(f(c - half), x_min), min(f(c + half), x_max)
        else:
            l = max(x_min if i == 0 else f(f(centers[i - 1] + c) / f(2.0)), x_min)
            r = min(x_max if i == n - 1 else f(f(c + centers[i + 1]) / f(2.0)), x_max)
        boxes.append(np.array([[l, y_min], [r, y_min], [r, y_max], [l, y_max]], np.float32))
    return boxes



def char_positions_to_word_boxes(line_bbox: np.ndarray, char_positions, char_count: int):
    """OAROCR::char_positions_to_word_boxes, src/oarocr/ocr.rs:1036-1072."""
    if len(char_positions) == 0 or char_count == 0:
        return []
    f = np.float32
    xs, ys = np.asarray(line_bbox, np.float32)[:, 0], np.asarray(line_bbox, np.float32)[:, 1]
    x_min, y_min, x_max, y_max = f(xs.min()), f(ys.min()), f(xs.max()), f(ys.max())
    width = f(x_max - x_min)
    cw = f(width / f(char_count))
    boxes = []
    for pos in char_positions:
        c = f(x_min + f(f(pos) * width))
        l, r = max(f(c - f(cw / f(2.0))), x_min), min(f(c + f(cw / f(2.0))), x_max)
        boxes.append(np.array([[l, y_min], [r, y_min], [r, y_max], [l, y_max]], np.float32))
    return boxes


# ------------------------------------------------------------------------------------------------ f4 layout detection
FILTERS = {"triangle": 0, "catmullrom": 1, "lanczos3": 2}


def resize_filter(rgb: np.ndarray, nw: int, nh: int, filter="lanczos3") -> np.ndarray:
    """image 0.25.6 imageops::resize with FilterType::{Triangle, CatmullRom, Lanczos3} (DetResizeForTest::with_filter,
    processors/resize_detection.rs:104-108 -> resize_exact, :362)."""
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w, _ = rgb.shape
    out = np.empty((nh, nw, 3), np.uint8)
    lib().orc_resize_filter_rgb(_p(rgb), w, h, nw, nh, FILTERS[filter], _p(out))
    return out


def layout_preprocess(rgb: np.ndarray, image_shape=(800, 608), filter="lanczos3", scale=1.0 / 255.0, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225),
                      bgr=True):
    """ScaleAwareDetectorModel::preprocess (models/detection/scale_aware_detector.rs:169-199): DetResizeForTest Type1 (keep_ratio = false:
    resize_exact to image_shape, resize_detection.rs:337-365; pages with h + w < 64 are padded to >= 32 x 32 first, :174-176) and
    NormalizeImage::with_color_order_from_rgb_stats (normalization.rs:241-273: the RGB statistics are permuted into the output
    channel order).  Returns (tensor [3, H, W], (src_h, src_w, ratio_h, ratio_w), (resized_h, resized_w))."""
    src_h, src_w = rgb.shape[:2]
    img = rgb
    if src_h + src_w < 64:
        ph, pw = max(32, src_h), max(32, src_w)
        pad = np.zeros((ph, pw, 3), np.uint8)
        pad[:src_h, :src_w] = rgb
        img = pad
    h, w = img.shape[:2]
    th, tw = image_shape
    if (th, tw) == (h, w):
        res, rh, rw = img, np.float32(1.0), np.float32(1.0)
    else:
        rh, rw = np.float32(th) / np.float32(h), np.float32(tw) / np.float32(w)
        res = resize_filter(img, tw, th, filter)
    m = [mean[2], mean[1], mean[0]] if bgr else list(mean)
    s = [std[2], std[1], std[0]] if bgr else list(std)
    alpha, beta = alpha_beta(scale, m, s)
    t = normalize(res, alpha, beta, src=(2, 1, 0) if bgr else (0, 1, 2), layout="chw")
    return t, (np.float32(src_h), np.float32(src_w), rh, rw), (res.shape[0], res.shape[1])


MODEL_TYPES = {"picodet": 0, "standard": 0, "rtdetr": 1, "pp-doclayout": 2}


def layout_postprocess(pred: np.ndarray, src_w, src_h, num_classes, score_threshold=0.5, nms_threshold=0.5, max_detections=100, model_type="picodet"):
    """LayoutPostProcess::apply for one image (processors/layout_postprocess.rs:60-634).  pred: [rows, feat] (the image's slab of the
    [batch, boxes, 1, feat] prediction tensor).  Returns (boxes [k, 4] x1 y1 x2 y2, classes [k], scores [k])."""
    pred = np.ascontiguousarray(pred, np.float32).reshape(-1, pred.shape[-1]) if pred.size else np.zeros((0, max(pred.shape[-1], 1)), np.float32)
    rows, feat = pred.shape
    ob, oc, os_ = np.zeros((max(rows, 1), 4), np.float32), np.zeros(max(rows, 1), np.int32), np.zeros(max(rows, 1), np.float32)
    L = lib()
    L.orc_layout_postprocess.restype = C.c_int
    n = L.orc_layout_postprocess(_p(pred), rows, feat, C.c_float(src_w), C.c_float(src_h), int(num_classes), C.c_float(score_threshold), C.c_float(nms_threshold),
                                 int(max_detections), MODEL_TYPES[model_type], _p(ob), _p(oc), _p(os_))
    return ob[:n].copy(), oc[:n].copy(), os_[:n].copy()


MERGE_MODES = {"large": 0, "union": 1, "small": 2}


def paddlex_layout_nms(boxes, classes, scores):
    """LayoutDetectionAdapter::paddlex_layout_nms (layout_detection_adapter.rs:884-935) on [n, 4] accessor values (x_min, y_min, x_max, y_max):
    indices of the selected boxes in selection order.  Restated in the compacting form of the reference's own test (:1668-1697)."""
    b = np.ascontiguousarray(boxes, np.float32).reshape(-1, 4)
    c = np.ascontiguousarray(classes, np.int32)
    s = np.ascontiguousarray(scores, np.float32)
    sel = np.zeros(max(len(s), 1), np.int32)
    L = lib()
    L.orc_paddlex_layout_nms.restype = C.c_int
    n = L.orc_paddlex_layout_nms(_p(b), _p(c), _p(s), len(s), _p(sel))
    return sel[:n].copy()


def pp_doclayout_postprocess(pred, src_w, src_h, num_classes, score_threshold=0.5, class_thresholds=None, layout_nms=True, image_class=-1, formula_class=-1,
                             merge_modes=None):
    """LayoutDetectionAdapter::postprocess_pp_doclayout (:631-846) for one image, up to and including the reading-order sort (labels,
    layout_unclip_ratio and max_elements are the caller's).  class_thresholds / merge_modes: {class_id: value}; modes "large" | "union" | "small"."""
    pred = np.ascontiguousarray(pred, np.float32).reshape(-1, pred.shape[-1]) if pred.size else np.zeros((0, max(pred.shape[-1], 1)), np.float32)
    rows, feat = pred.shape
    thr = np.full(max(num_classes, 1), np.nan, np.float32)
    for k, v in (class_thresholds or {}).items():
        if 0 <= int(k) < num_classes:
            thr[int(k)] = v
    mm = np.full(max(num_classes, 1), -1, np.int32)
    for k, v in (merge_modes or {}).items():
        if 0 <= int(k) < num_classes:
            mm[int(k)] = MERGE_MODES[v]
    ob, oc, os_ = np.zeros((max(rows, 1), 4), np.float32), np.zeros(max(rows, 1), np.int32), np.zeros(max(rows, 1), np.float32)
    L = lib()
    L.orc_pp_doclayout_postprocess.restype = C.c_int
    n = L.orc_pp_doclayout_postprocess(_p(pred), rows, feat, C.c_float(src_w), C.c_float(src_h), int(num_classes), C.c_float(score_threshold),
                                       _p(thr) if class_thresholds else None, int(bool(layout_nms)), int(image_class), int(formula_class),
                                       _p(mm) if merge_modes else None, _p(ob), _p(oc), _p(os_))
    return ob[:n].copy(), oc[:n].copy(), os_[:n].copy()


def apply_nms_with_merge(boxes, classes, scores, mode_of_class, nms_threshold=0.5, max_detections=100):
    """processors/layout_postprocess.rs:743-841 (class_merge_modes of the PicoDet / RT-DETR adapters).  mode_of_class: per class id "large" | "union" | "small"."""
    b = np.ascontiguousarray(boxes, np.float32).reshape(-1, 4)
    c = np.ascontiguousarray(classes, np.int32)
    s = np.ascontiguousarray(scores, np.float32)
    modes = np.ascontiguousarray([MERGE_MODES[m] for m in mode_of_class], np.int32)
    n = len(s)
    ob, oc, os_ = np.zeros((max(n, 1), 4), np.float32), np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.float32)
    L = lib()
    L.orc_apply_nms_with_merge.restype = C.c_int
    k = L.orc_apply_nms_with_merge(_p(b), _p(c), _p(s), n, _p(modes), len(modes), C.c_float(nms_threshold), int(max_detections), _p(ob), _p(oc), _p(os_))
    return ob[:k].copy(), oc[:k].copy(), os_[:k].copy()
