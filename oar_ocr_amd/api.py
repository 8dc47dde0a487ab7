"""ctypes binding of libOarMi355x.so plus a thin Python mirror of the reference's user-facing API for the
det+rec path (same names / argument meaning / error behaviour as the Rust reference):

    OAROCRBuilder(det, rec, dict).image_batch_size(..)...build() -> OAROCR      src/oarocr/ocr.rs:105,249-417
    OAROCR.predict(images) -> [OAROCRResult]                                    src/oarocr/ocr.rs:518-659
    TextDetectionPredictor / TextRecognitionPredictor                           oar-ocr-core/src/predictors/*.rs
    CTCLabelDecode.decode_argmax                                                processors/decode.rs:505-614

All compute goes through the C ABI (include/oar_mi355x.h).  There is NO CPU fallback: loading fails loudly
when the HIP library is missing, and every call fails with OCRError(code=OAR_DEVICE) when no GPU is visible.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from pathlib import Path
from typing import List, Optional, Sequence

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "lib" / "libOarMi355x.so"
_lib = None

OAR_OK, OAR_INVALID_INPUT, OAR_MODEL_LOAD, OAR_UNSUPPORTED_OP, OAR_SHAPE_MISMATCH, OAR_DEVICE, OAR_OOM, OAR_INTERNAL = range(8)


class OCRError(RuntimeError):
    """Mirror of `OCRError` (oar-ocr-core/src/core/errors/types.rs:112-214); `.code` is the oar_status."""

    def __init__(self, code: int, message: str):
        super().__init__(f"[oar_status {code}] {message}")
        self.code = code
        self.message = message


class EngineCfg(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("use_hip_graph", C.c_int32), ("profile", C.c_int32), ("precision", C.c_int32), ("stream", C.c_void_p)]


PRECISION_F32 = 0   # oar_precision: the only arithmetic mode (f32 FMA / f32 MFMA / bf16x6); anything else is refused by oar_engine_create


class Tensor(C.Structure):
    _fields_ = [("rank", C.c_int32), ("dims", C.c_int64 * 8), ("data", C.POINTER(C.c_float)), ("name", C.c_char * 64),
                ("dtype", C.c_int32), ("reserved", C.c_int32), ("data_i64", C.POINTER(C.c_int64))]


class Input(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.POINTER(C.c_float)), ("dims", C.POINTER(C.c_int64)), ("rank", C.c_int32),
                ("reserved", C.c_int32)]


class IoInfo(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("dtype", C.c_int32), ("rank", C.c_int32), ("dims", C.c_int64 * 8)]


OUTPUT_VIEW_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(C.c_int64), C.c_int32, C.POINTER(C.c_float))


class DetCfg(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("limit_side_len", C.c_uint32), ("limit_type", C.c_int32), ("max_side_limit", C.c_uint32),
                ("max_candidates", C.c_uint32), ("use_hip_graph", C.c_int32), ("profile", C.c_int32), ("host_threads", C.c_int32),
                ("box_type", C.c_int32), ("score_mode", C.c_int32), ("use_dilation", C.c_int32), ("gpu_contours", C.c_int32)]


class DetResult(C.Structure):
    _fields_ = [("n_images", C.c_uint32), ("n_boxes", C.c_uint32), ("box_offsets", C.POINTER(C.c_uint32)),
                ("points", C.POINTER(C.c_float)), ("scores", C.POINTER(C.c_float)), ("n_points", C.c_uint32), ("point_offsets", C.POINTER(C.c_uint32))]


class RecCfg(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("rec_image_shape", C.c_uint32 * 3), ("max_img_w", C.c_uint32), ("use_hip_graph", C.c_int32),
                ("profile", C.c_int32), ("reserved", C.c_int32)]


class RecResult(C.Structure):
    _fields_ = [("batch", C.c_uint32), ("seq_len", C.c_uint32), ("vocab", C.c_uint32), ("tensor_width", C.c_uint32),
                ("indices", C.POINTER(C.c_int64)), ("probs", C.POINTER(C.c_float))]


class OcrCfg(C.Structure):
    _fields_ = [("det", DetCfg), ("rec", RecCfg), ("det_thresh", C.c_float), ("det_box_thresh", C.c_float), ("det_unclip_ratio", C.c_float),
                ("image_batch_size", C.c_uint32), ("region_batch_size", C.c_uint32), ("max_pooled_crops", C.c_uint32), ("box_sort", C.c_int32), ("lanes", C.c_uint32)]


class OcrResult(C.Structure):
    _fields_ = [("n_images", C.c_uint32), ("n_regions", C.c_uint32), ("region_offsets", C.POINTER(C.c_uint32)),
                ("points", C.POINTER(C.c_float)), ("det_scores", C.POINTER(C.c_float)), ("crop_wh", C.POINTER(C.c_uint32)),
                ("seq_len", C.POINTER(C.c_uint32)), ("max_wh_ratio", C.POINTER(C.c_float)), ("ctc_offsets", C.POINTER(C.c_uint64)),
                ("ctc_indices", C.POINTER(C.c_int64)), ("ctc_probs", C.POINTER(C.c_float)),
                ("page_angle", C.POINTER(C.c_float)), ("page_rectified", C.POINTER(C.c_uint8)), ("line_angle", C.POINTER(C.c_float)),
                ("n_points", C.c_uint32), ("point_offsets", C.POINTER(C.c_uint32))]


class ClsCfg(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("input_h", C.c_uint32), ("input_w", C.c_uint32), ("resize_short", C.c_uint32),
                ("topk", C.c_uint32), ("batch", C.c_uint32)]


class ClsResult(C.Structure):
    _fields_ = [("n_images", C.c_uint32), ("topk", C.c_uint32), ("n_classes", C.c_uint32), ("class_ids", C.POINTER(C.c_int32)),
                ("scores", C.POINTER(C.c_float))]


class RectCfg(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("target_h", C.c_uint32), ("target_w", C.c_uint32)]


class TextResult(C.Structure):
    _fields_ = [("n", C.c_uint32), ("text_offsets", C.POINTER(C.c_uint64)), ("utf8", C.POINTER(C.c_char)), ("scores", C.POINTER(C.c_float)),
                ("char_offsets", C.POINTER(C.c_uint64)), ("char_cols", C.POINTER(C.c_uint32)), ("char_positions", C.POINTER(C.c_float)),
                ("seq_len", C.POINTER(C.c_uint32)), ("kept", C.POINTER(C.c_uint8))]


class WordBoxes(C.Structure):
    _fields_ = [("n_regions", C.c_uint32), ("box_offsets", C.POINTER(C.c_uint64)), ("boxes", C.POINTER(C.c_float))]


class LayoutCfg(C.Structure):
    _fields_ = [("device_id", C.c_int32), ("input_h", C.c_uint32), ("input_w", C.c_uint32), ("resize_filter", C.c_int32), ("color_bgr", C.c_int32), ("scale", C.c_float),
                ("mean", C.c_float * 3), ("std", C.c_float * 3), ("num_classes", C.c_uint32), ("model_type", C.c_int32), ("score_threshold", C.c_float),
                ("nms_threshold", C.c_float), ("max_detections", C.c_uint32)]


class LayoutResult(C.Structure):
    _fields_ = [("n_images", C.c_uint32), ("n_boxes", C.c_uint32), ("box_offsets", C.POINTER(C.c_uint32)), ("boxes", C.POINTER(C.c_float)),
                ("classes", C.POINTER(C.c_int32)), ("scores", C.POINTER(C.c_float)), ("feature_dim", C.c_uint32)]


class ProfEntry(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint64), ("total_ms", C.c_double), ("alg_bytes", C.c_double), ("alg_flops", C.c_double)]


EXPORTS = [
    "oar_last_error", "oar_version", "oar_device_count", "oar_engine_create", "oar_engine_destroy", "oar_engine_input_name",
    "oar_engine_run", "oar_engine_run_named", "oar_engine_run_first_f32", "oar_engine_io", "oar_tensor_free", "oar_engine_cost", "oar_det_create", "oar_det_destroy", "oar_det_run", "oar_det_result_free",
    "oar_db_postprocess", "oar_rec_create", "oar_rec_destroy", "oar_rec_run", "oar_rec_result_free", "oar_ocr_create", "oar_ocr_destroy",
    "oar_ocr_predict", "oar_ocr_predict_device", "oar_ocr_result_free", "oar_dev_alloc", "oar_dev_upload", "oar_dev_download",
    "oar_dev_free", "oar_dev_synchronize", "oar_k_normalize", "oar_k_rec_preprocess", "oar_k_resize_triangle", "oar_k_threshold",
    "oar_k_ctc_argmax", "oar_k_box_scores", "oar_k_rotate_crop", "oar_prof_reset", "oar_prof_enable", "oar_prof_filter", "oar_prof_sampling", "oar_prof_snapshot",
    "oar_host_candidates", "oar_host_unclip", "oar_host_mini_box", "oar_host_convex_hull", "oar_host_sort_quad_boxes", "oar_host_pool_selftest", "oar_host_plan_crop",
    "oar_cls_create", "oar_cls_destroy", "oar_cls_run", "oar_cls_result_free", "oar_cls_preprocess", "oar_rect_create", "oar_rect_destroy",
    "oar_rect_run", "oar_ocr_attach", "oar_k_rotate_rgb", "oar_k_bgr_planes_to_rgb", "oar_host_rotate_back_points",
    "oar_engine_cache_stats", "oar_onnx_inspect", "oar_host_contours", "oar_ctc_dict_create", "oar_ctc_dict_destroy", "oar_ctc_dict_classes",
    "oar_ctc_decode", "oar_ocr_decode", "oar_text_result_free", "oar_db_postprocess_ex", "oar_k_dilate", "oar_k_poly_scores", "oar_debug_inject_failure", "oar_k_contours", "oar_host_contours_bits",
    "oar_k_unclip", "oar_k_rec_preprocess_flip", "oar_layout_create", "oar_layout_destroy", "oar_layout_run", "oar_layout_result_free", "oar_layout_preprocess", "oar_k_resize_filter", "oar_k_layout_postprocess", "oar_layout_run_ppdoc", "oar_k_ppdoc_postprocess", "oar_host_nms_with_merge", "oar_image_decode_device", "oar_ocr_predict_async", "oar_ocr_wait", "oar_ctc_word_boxes", "oar_char_positions_to_word_boxes", "oar_ocr_word_boxes", "oar_word_boxes_free", "oar_image_decode", "oar_image_free", "oar_host_approx_poly_dp", "oar_host_perimeter", "oar_host_unclip_poly", "oar_host_offset_ring", "oar_host_ring_outline", "oar_host_sort_poly_boxes",
]


def lib():
    """Loads the HIP library. Raises (never falls back) when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise OCRError(OAR_DEVICE, f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                   "(the MI355X path has no CPU fallback)")
    L = C.CDLL(str(LIB_PATH))
    vp, u8pp, u32p, f32p = C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.c_float)
    L.oar_last_error.restype = C.c_size_t
    L.oar_last_error.argtypes = [C.c_char_p, C.c_size_t]
    L.oar_version.restype = C.c_size_t
    L.oar_version.argtypes = [C.c_char_p, C.c_size_t]
    L.oar_device_count.restype = C.c_int
    L.oar_engine_create.argtypes = [vp, C.c_size_t, C.POINTER(EngineCfg), C.POINTER(vp)]
    L.oar_engine_destroy.argtypes = [vp]
    L.oar_engine_destroy.restype = None
    L.oar_engine_input_name.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.oar_engine_run.argtypes = [vp, vp, C.POINTER(C.c_int64), C.c_int32, C.POINTER(Tensor), C.c_int32, C.POINTER(C.c_int32)]
    L.oar_engine_run_named.argtypes = [vp, C.POINTER(Input), C.c_int32, C.POINTER(Tensor), C.c_int32, C.POINTER(C.c_int32)]
    L.oar_engine_run_first_f32.argtypes = [vp, C.POINTER(Input), C.c_int32, OUTPUT_VIEW_FN, vp]
    L.oar_engine_io.argtypes = [vp, C.POINTER(IoInfo), C.c_int32, C.POINTER(C.c_int32), C.POINTER(IoInfo), C.c_int32, C.POINTER(C.c_int32)]
    L.oar_tensor_free.argtypes = [C.POINTER(Tensor)]
    L.oar_tensor_free.restype = None
    L.oar_engine_cost.argtypes = [vp, C.POINTER(C.c_int64), C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32)]
    L.oar_engine_cache_stats.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.oar_onnx_inspect.argtypes = [vp, C.c_size_t, C.c_char_p, C.c_size_t]
    L.oar_det_create.argtypes = [vp, C.c_size_t, C.POINTER(DetCfg), C.POINTER(vp)]
    L.oar_det_destroy.argtypes = [vp]
    L.oar_det_destroy.restype = None
    L.oar_det_run.argtypes = [vp, u8pp, u32p, u32p, C.c_uint32, C.c_float, C.c_float, C.c_float, C.POINTER(DetResult)]
    L.oar_det_result_free.argtypes = [C.POINTER(DetResult)]
    L.oar_det_result_free.restype = None
    L.oar_db_postprocess.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_uint32, C.POINTER(DetResult)]
    L.oar_db_postprocess_ex.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_int32, C.c_int32,
                                        C.c_int32, C.POINTER(DetResult)]
    L.oar_debug_inject_failure.argtypes = [C.c_char_p, C.c_int32]
    L.oar_k_dilate.argtypes = [vp, C.c_uint32, C.c_uint32, vp]
    L.oar_k_poly_scores.argtypes = [vp, C.c_uint32, C.c_uint32, vp, u32p, C.c_uint32, vp]
    L.oar_rec_create.argtypes = [vp, C.c_size_t, C.POINTER(RecCfg), C.POINTER(vp)]
    L.oar_rec_destroy.argtypes = [vp]
    L.oar_rec_destroy.restype = None
    L.oar_rec_run.argtypes = [vp, u8pp, u32p, u32p, C.c_uint32, C.POINTER(RecResult)]
    L.oar_rec_result_free.argtypes = [C.POINTER(RecResult)]
    L.oar_rec_result_free.restype = None
    L.oar_ocr_create.argtypes = [vp, C.c_size_t, vp, C.c_size_t, C.POINTER(OcrCfg), C.POINTER(vp)]
    L.oar_ocr_destroy.argtypes = [vp]
    L.oar_ocr_destroy.restype = None
    L.oar_ocr_predict.argtypes = [vp, u8pp, u32p, u32p, C.c_uint32, C.POINTER(OcrResult)]
    L.oar_ocr_predict_device.argtypes = [vp, u8pp, u32p, u32p, C.c_uint32, C.POINTER(OcrResult)]
    L.oar_ocr_result_free.argtypes = [C.POINTER(OcrResult)]
    L.oar_ocr_result_free.restype = None
    L.oar_dev_alloc.argtypes = [C.c_int32, C.c_size_t, C.POINTER(vp)]
    L.oar_dev_upload.argtypes = [vp, vp, C.c_size_t]
    L.oar_dev_download.argtypes = [vp, vp, C.c_size_t]
    L.oar_dev_free.argtypes = [vp]
    L.oar_dev_free.restype = None
    L.oar_dev_synchronize.argtypes = [C.c_int32]
    L.oar_k_normalize.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(C.c_int32), f32p, f32p, C.c_int32, vp]
    L.oar_k_rec_preprocess.argtypes = [u8pp, u32p, u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, u32p]
    L.oar_ctc_word_boxes.argtypes = [vp, C.c_uint32, C.c_char_p, C.c_size_t, vp, C.c_uint32, C.c_uint32, C.c_float, C.c_float, vp, C.c_uint32, u32p]
    L.oar_char_positions_to_word_boxes.argtypes = [vp, C.c_uint32, vp, C.c_uint32, C.c_uint32, vp, C.c_uint32, u32p]
    L.oar_ocr_word_boxes.argtypes = [vp, vp, vp]
    L.oar_word_boxes_free.argtypes = [vp]
    L.oar_word_boxes_free.restype = None
    L.oar_k_rec_preprocess_flip.argtypes = [u8pp, u32p, u32p, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp, u32p]
    L.oar_k_resize_triangle.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, vp]
    L.oar_k_threshold.argtypes = [vp, C.c_size_t, C.c_float, vp]
    L.oar_k_ctc_argmax.argtypes = [vp, C.c_size_t, C.c_size_t, vp, vp]
    L.oar_k_box_scores.argtypes = [vp, C.c_uint32, C.c_uint32, vp, C.c_uint32, vp]
    L.oar_k_unclip.argtypes = [vp, C.c_uint32, C.c_float, vp, vp, C.c_uint32]
    L.oar_image_decode.argtypes = [vp, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.oar_image_free.argtypes = [C.POINTER(C.c_uint8)]
    L.oar_layout_create.argtypes = [vp, C.c_size_t, vp, C.POINTER(C.c_void_p)]
    L.oar_layout_destroy.argtypes = [vp]
    L.oar_layout_destroy.restype = None
    L.oar_layout_run.argtypes = [vp, u8pp, u32p, u32p, C.c_uint32, vp]
    L.oar_layout_result_free.argtypes = [vp]
    L.oar_layout_result_free.restype = None
    L.oar_layout_preprocess.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp]
    L.oar_k_resize_filter.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, vp]
    L.oar_k_layout_postprocess.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, vp, C.c_uint32, C.c_int32, C.c_float, C.c_float, C.c_uint32, vp]
    L.oar_ocr_predict_async.argtypes = [vp, u8pp, u32p, u32p, C.c_uint32, C.c_int32, C.POINTER(C.c_uint64)]
    L.oar_ocr_wait.argtypes = [vp, C.c_uint64, vp]
    L.oar_image_decode_device.argtypes = [vp, C.c_size_t, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.oar_image_free.restype = None
    L.oar_k_rotate_crop.argtypes = [vp, C.c_uint32, C.c_uint32, f32p, vp, C.c_size_t, u32p, u32p]
    L.oar_host_candidates.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, vp, C.c_int32]
    L.oar_host_candidates.restype = C.c_int32
    L.oar_host_contours.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, vp, vp, vp, C.c_int64]
    L.oar_host_contours.restype = C.c_int32
    L.oar_host_unclip.argtypes = [vp, C.c_float, vp, C.c_int32]
    L.oar_host_unclip.restype = C.c_int32
    L.oar_host_approx_poly_dp.argtypes = [vp, C.c_int32, C.c_float, vp, C.c_int32]
    L.oar_host_approx_poly_dp.restype = C.c_int32
    L.oar_host_perimeter.argtypes = [vp, C.c_int32]
    L.oar_host_perimeter.restype = C.c_float
    L.oar_host_unclip_poly.argtypes = [vp, C.c_int32, C.c_float, vp, C.c_int32]
    L.oar_host_unclip_poly.restype = C.c_int32
    L.oar_host_offset_ring.argtypes = [vp, C.c_int32, C.c_double, vp, C.c_int32]
    L.oar_host_offset_ring.restype = C.c_int32
    L.oar_host_ring_outline.argtypes = [vp, C.c_int32, C.c_int32, vp, C.c_int32]
    L.oar_host_ring_outline.restype = C.c_int32
    L.oar_host_sort_poly_boxes.argtypes = [vp, vp, C.c_int32, vp]
    L.oar_host_sort_poly_boxes.restype = None
    L.oar_host_mini_box.argtypes = [vp, C.c_int32, vp, f32p]
    L.oar_host_mini_box.restype = C.c_int32
    L.oar_host_convex_hull.argtypes = [vp, C.c_int32, vp, C.c_int32]
    L.oar_host_convex_hull.restype = C.c_int32
    L.oar_host_sort_quad_boxes.argtypes = [vp, C.c_int32, vp]
    L.oar_host_sort_quad_boxes.restype = None
    L.oar_host_plan_crop.argtypes = [C.c_uint32, C.c_uint32, vp, vp, vp]
    L.oar_host_plan_crop.restype = None
    L.oar_cls_create.argtypes = [vp, C.c_size_t, C.POINTER(ClsCfg), C.POINTER(vp)]
    L.oar_cls_destroy.argtypes = [vp]
    L.oar_cls_destroy.restype = None
    L.oar_cls_run.argtypes = [vp, u8pp, u32p, u32p, C.c_uint32, C.POINTER(ClsResult)]
    L.oar_cls_result_free.argtypes = [C.POINTER(ClsResult)]
    L.oar_cls_result_free.restype = None
    L.oar_cls_preprocess.argtypes = [vp, u8pp, u32p, u32p, C.c_uint32, vp]
    L.oar_rect_create.argtypes = [vp, C.c_size_t, C.POINTER(RectCfg), C.POINTER(vp)]
    L.oar_rect_destroy.argtypes = [vp]
    L.oar_rect_destroy.restype = None
    L.oar_rect_run.argtypes = [vp, vp, C.c_uint32, C.c_uint32, vp]
    L.oar_ocr_attach.argtypes = [vp, vp, vp, vp]
    L.oar_k_rotate_rgb.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_int32, vp]
    L.oar_k_bgr_planes_to_rgb.argtypes = [vp, C.c_uint64, C.c_float, vp]
    L.oar_host_rotate_back_points.argtypes = [vp, C.c_uint32, C.c_float, C.c_uint32, C.c_uint32]
    L.oar_ctc_dict_create.argtypes = [C.c_char_p, C.c_size_t, C.c_int32, C.POINTER(vp)]
    L.oar_ctc_dict_destroy.argtypes = [vp]
    L.oar_ctc_dict_destroy.restype = None
    L.oar_ctc_dict_classes.argtypes = [vp]
    L.oar_ctc_dict_classes.restype = C.c_uint32
    L.oar_ctc_decode.argtypes = [vp, vp, vp, C.c_uint32, C.c_uint32, C.c_float, C.POINTER(TextResult)]
    L.oar_ocr_decode.argtypes = [vp, C.POINTER(OcrResult), C.c_float, C.POINTER(TextResult)]
    L.oar_text_result_free.argtypes = [C.POINTER(TextResult)]
    L.oar_text_result_free.restype = None
    L.oar_prof_reset.restype = None
    L.oar_prof_enable.argtypes = [C.c_int32]
    L.oar_prof_enable.restype = None
    L.oar_prof_filter.argtypes = [C.c_char_p]
    L.oar_prof_filter.restype = None
    L.oar_prof_sampling.argtypes = [C.c_int32, C.c_int32]
    L.oar_prof_sampling.restype = None
    L.oar_prof_snapshot.argtypes = [C.POINTER(ProfEntry), C.c_int32]
    L.oar_prof_snapshot.restype = C.c_int32
    _lib = L
    return L


def _check(st: int):
    if st != OAR_OK:
        buf = C.create_string_buffer(4096)
        lib().oar_last_error(buf, 4096)
        raise OCRError(st, buf.value.decode(errors="replace"))


# ------------------------------------------------------------------------------------------------ image loading
DEFAULT_PARALLEL_THRESHOLD = 4   # core/constants.rs:18


def load_image_from_memory(data: bytes) -> np.ndarray:
    """utils/image.rs:65-68: encoded bytes -> [H, W, 3] u8 (RgbImage).  PNG is decoded by the library to the bytes image 0.25.6
    yields, JPEG to libjpeg's default-path bytes (= PIL's; unpinned against zune-jpeg), BMP / PNM (maxval 255) / baseline TIFF / the first frame of a GIF
    (round 4, image_misc_decode.cc; == PIL); WebP and the variants those decoders refuse raise OCRError with OAR_UNSUPPORTED_OP (the message
    names the format)."""
    buf = (C.c_char * len(data)).from_buffer_copy(data) if len(data) else (C.c_char * 1)()
    out = C.POINTER(C.c_uint8)()
    w, h = C.c_uint32(0), C.c_uint32(0)
    _check(lib().oar_image_decode(C.cast(buf, C.c_void_p), len(data), C.byref(out), C.byref(w), C.byref(h)))
    try:
        return np.ctypeslib.as_array(out, shape=(h.value, w.value, 3)).copy()
    finally:
        lib().oar_image_free(out)


def load_image_to_device(data: bytes, device_id: int = 0):
    """oar_image_decode_device: the decoded page in HBM (JPEG: Huffman on the host, IDCT / upsampling / colour as HIP kernels).
    Returns (DeviceBuffer-like pointer holder, width, height); pass .ptr to predict_device, free() when done."""
    buf = (C.c_char * len(data)).from_buffer_copy(data) if len(data) else (C.c_char * 1)()
    ptr = C.c_void_p()
    w, h = C.c_uint32(0), C.c_uint32(0)
    _check(lib().oar_image_decode_device(C.cast(buf, C.c_void_p), len(data), device_id, C.byref(ptr), C.byref(w), C.byref(h)))
    holder = DeviceBuffer.__new__(DeviceBuffer)
    holder.ptr, holder.nbytes = ptr, w.value * h.value * 3
    return holder, w.value, h.value


def load_image(path) -> np.ndarray:
    """utils/image.rs:87-92"""
    with open(path, "rb") as f:
        return load_image_from_memory(f.read())


def load_images(paths, parallel_threshold: Optional[int] = None) -> List[np.ndarray]:
    """utils/image.rs:299-345: sequential up to the threshold, in parallel above it (the decoder holds no lock and ctypes
    releases the GIL, so the threads really overlap, like the reference's rayon pool); any failure fails the call."""
    paths = list(paths)
    thr = DEFAULT_PARALLEL_THRESHOLD if parallel_threshold is None else parallel_threshold
    if len(paths) > thr:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=min(len(paths), os.cpu_count() or 4)) as ex:
            return list(ex.map(load_image, paths))
    return [load_image(p) for p in paths]


def version() -> str:
    buf = C.create_string_buffer(256)
    lib().oar_version(buf, 256)
    return buf.value.decode()


def onnx_inspect(model: bytes) -> str:
    """Host-only: parse + validate a model like oar_engine_create does; returns the op-histogram summary."""
    buf = C.create_string_buffer(8192)
    b = (C.c_char * max(len(model), 1)).from_buffer_copy(model or b"\0")
    st = lib().oar_onnx_inspect(C.cast(b, C.c_void_p), len(model), buf, 8192)
    _check(st)
    return buf.value.decode()


def debug_inject_failure(site: str, count: int):
    """Test hook (oar_debug_inject_failure): the next `count` occurrences of `site` fail with OAR_DEVICE."""
    _check(lib().oar_debug_inject_failure(site.encode(), count))


def device_count() -> int:
    return int(lib().oar_device_count())


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def _img_arrays(images: Sequence[np.ndarray]):
    imgs = [np.ascontiguousarray(im, dtype=np.uint8) for im in images]
    for im in imgs:
        if im.ndim != 3 or im.shape[2] != 3:
            raise OCRError(OAR_INVALID_INPUT, "images must be HxWx3 uint8 (RgbImage)")
    ptrs = (C.c_void_p * max(len(imgs), 1))(*[im.ctypes.data for im in imgs])
    ws = (C.c_uint32 * max(len(imgs), 1))(*[im.shape[1] for im in imgs])
    hs = (C.c_uint32 * max(len(imgs), 1))(*[im.shape[0] for im in imgs])
    return imgs, ptrs, ws, hs


# ------------------------------------------------------------------------------------------------ Seam A
class OrtInfer:
    """Drop-in for `OrtInfer` (core/inference/mod.rs:31-115): `.onnx` bytes in, f32 tensors in/out."""

    def __init__(self, model: bytes, device_id: int = 0, profile: bool = False, precision: int = PRECISION_F32, stream: Optional[int] = None):
        """stream: a hipStream_t of the caller (integer handle, e.g. torch.cuda.Stream().cuda_stream) on `device_id`; every copy and kernel of
        `infer` is then enqueued on it.  None: the engine creates its own stream."""
        self._h = C.c_void_p()
        cfg = EngineCfg(device_id, 0, int(profile), int(precision), C.c_void_p(stream) if stream else None)
        buf = (C.c_char * len(model)).from_buffer_copy(model)
        _check(lib().oar_engine_create(C.cast(buf, C.c_void_p), len(model), C.byref(cfg), C.byref(self._h)))

    def input_name(self) -> str:
        buf = C.create_string_buffer(256)
        _check(lib().oar_engine_input_name(self._h, buf, 256))
        return buf.value.decode()

    @staticmethod
    def _collect(outs, n):
        res = []
        for i in range(n):
            t = outs[i]
            shape = tuple(t.dims[k] for k in range(t.rank))
            cnt = int(np.prod(shape)) if shape else 1
            src = t.data_i64 if t.dtype == 7 else t.data     # TensorOutput::I64 / ::F32 (tensor_output.rs:16-21)
            arr = (np.ctypeslib.as_array(src, shape=(cnt,)).copy() if cnt else np.zeros(0, np.int64 if t.dtype == 7 else np.float32)).reshape(shape)
            res.append((t.name.decode(), arr))
            lib().oar_tensor_free(C.byref(outs[i]))
        return res

    @staticmethod
    def _inputs(inputs):
        keep, arr = [], (Input * len(inputs))()
        for i, (name, x) in enumerate(inputs):
            x = np.ascontiguousarray(x, dtype=np.float32)
            dims = (C.c_int64 * x.ndim)(*x.shape)
            keep += [x, dims]
            arr[i] = Input(name.encode() if name else None, x.ctypes.data_as(C.POINTER(C.c_float)), dims, x.ndim, 0)
        return arr, keep

    def infer(self, x, *more):
        """Returns [(name, ndarray)] for every graph output (ort_infer_execution.rs:121-219).  `x` is either one f32 array
        (bound to the primary input) or a list of (name, array) pairs -- the reference's `&[(&str, TensorInput)]`."""
        outs = (Tensor * 16)()
        n = C.c_int32(0)
        if isinstance(x, (list, tuple)) and x and isinstance(x[0], (list, tuple)):
            arr, keep = self._inputs(list(x))
            _check(lib().oar_engine_run_named(self._h, arr, len(arr), outs, 16, C.byref(n)))
            return self._collect(outs, n.value)
        if isinstance(x, (list, tuple)) and not x:
            _check(lib().oar_engine_run_named(self._h, None, 0, outs, 16, C.byref(n)))
        x = np.ascontiguousarray(x, dtype=np.float32)
        dims = (C.c_int64 * x.ndim)(*x.shape)
        _check(lib().oar_engine_run(self._h, _p(x), dims, x.ndim, outs, 16, C.byref(n)))
        return self._collect(outs, n.value)

    def infer_first_output_f32(self, inputs, fn):
        """`fn(shape, data)` sees the first output as a borrowed f32 view, valid only inside the call
        (OrtInfer::infer_first_output_f32, ort_infer_execution.rs:234-306); returns fn's result."""
        if isinstance(inputs, np.ndarray):
            inputs = [("", inputs)]
        arr, keep = self._inputs(list(inputs))
        box = {}

        def tramp(_user, dims, rank, data):
            try:
                shape = tuple(dims[k] for k in range(rank))
                cnt = int(np.prod(shape)) if shape else 1
                view = np.ctypeslib.as_array(data, shape=(cnt,)).reshape(shape) if cnt else np.zeros(shape, np.float32)
                box["r"] = fn(shape, view)
                return 0
            except Exception as ex:      # noqa: BLE001 -- must not unwind through the C frame
                box["e"] = ex
                return 1

        cb = OUTPUT_VIEW_FN(tramp)
        st = lib().oar_engine_run_first_f32(self._h, arr, len(arr), cb, None)
        if "e" in box:
            raise box["e"]
        _check(st)
        return box.get("r")

    def _io(self):
        ins, outs = (IoInfo * 32)(), (IoInfo * 32)()
        ni, no = C.c_int32(0), C.c_int32(0)
        _check(lib().oar_engine_io(self._h, ins, 32, C.byref(ni), outs, 32, C.byref(no)))
        conv = lambda a, n: [(a[i].name.decode(), a[i].dtype, None if a[i].rank < 0 else [a[i].dims[k] for k in range(a[i].rank)]) for i in range(n)]
        return conv(ins, ni.value), conv(outs, no.value)

    def input_names_from_model(self):
        """core/inference/mod.rs:66-79"""
        return [n for n, _, _ in self._io()[0]]

    def primary_input_shape(self):
        """Declared shape of the first input, dynamic dimensions as -1; None when undeclared (mod.rs:81-92)."""
        return self._io()[0][0][2]

    def output_shapes(self):
        """[(name, declared shape)] of the outputs that declare one (mod.rs:94-112)."""
        return [(n, sh) for n, _, sh in self._io()[1] if sh is not None]

    def cache_stats(self):
        """(cached plans, evicted plans): plans are per input shape, LRU-bounded (OAR_PLAN_CACHE, default 256)."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        _check(lib().oar_engine_cache_stats(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def cost(self, shape):
        dims = (C.c_int64 * len(shape))(*shape)
        fl, by, nk = C.c_double(0), C.c_double(0), C.c_int32(0)
        _check(lib().oar_engine_cost(self._h, dims, len(shape), C.byref(fl), C.byref(by), C.byref(nk)))
        return fl.value, by.value, nk.value

    def close(self):
        """Releases the native handle (streams, HBM). Safe to call twice."""
        if getattr(self, "_h", None) and _lib is not None:
            _lib.oar_engine_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------ host-side types
@dataclass
class TextDetectionConfig:
    """domain/tasks/text_detection.rs:34-66"""
    score_threshold: float = 0.3
    box_threshold: float = 0.6
    unclip_ratio: float = 1.5
    max_candidates: int = 1000
    limit_side_len: Optional[int] = None
    limit_type: Optional[str] = None   # "max" | "min" | "resize_long"
    max_side_len: Optional[int] = None
    # DBPostProcess options (processors/db_postprocess.rs:60-98; the adapters set them at build time)
    box_type: str = "quad"             # "quad" | "poly" (seal / curved text: Detection.bbox is [n, 2] with n >= 3)
    score_mode: str = "fast"           # "fast" | "slow"
    use_dilation: bool = False
    gpu_contours: bool = False         # backend option (not in the reference): follow the mask borders on the GPU (oar_det_cfg.gpu_contours)

    def validate(self):
        if self.box_type not in ("quad", "poly") or self.score_mode not in ("fast", "slow"):
            raise OCRError(OAR_INVALID_INPUT, "box_type must be quad|poly and score_mode fast|slow")
        for name, v, lo, hi in (("score_threshold", self.score_threshold, 0.0, 1.0), ("box_threshold", self.box_threshold, 0.0, 1.0)):
            if not (lo <= v <= hi):
                raise OCRError(OAR_INVALID_INPUT, f"{name} must be in [{lo}, {hi}]")
        if self.unclip_ratio <= 0:
            raise OCRError(OAR_INVALID_INPUT, "unclip_ratio must be > 0")
        if self.max_candidates < 1:
            raise OCRError(OAR_INVALID_INPUT, "max_candidates must be >= 1")


@dataclass
class Detection:
    bbox: np.ndarray   # [4,2] f32 (BoxType::Poly: [n,2])
    score: float


@dataclass
class TextRegion:
    """domain/text_region.rs:10-41"""
    bounding_box: np.ndarray
    text: Optional[str]
    confidence: Optional[float]
    dt_poly: Optional[np.ndarray] = None
    rec_poly: Optional[np.ndarray] = None
    word_boxes: Optional[List[np.ndarray]] = None
    det_score: float = 0.0
    crop_wh: tuple = (0, 0)
    orientation_angle: Optional[float] = None   # text-line orientation (0 / 180), ocr.rs:888
    rec_max_wh_ratio: float = 0.0               # chunk_max_wh_ratio of the recognition batch the crop was in (ocr.rs:828-831)
    rec_seq_len: int = 0


@dataclass
class OAROCRResult:
    """src/oarocr/result.rs:34-49"""
    input_path: str
    index: int
    text_regions: List[TextRegion] = field(default_factory=list)
    orientation_angle: Optional[float] = None
    rectified: bool = False   # rectified_img is Some: boxes are in rectified space (preprocess.rs:84-89)


def dict_lines(text: str) -> List[str]:
    """Rust `str::lines()` (src/oarocr/ocr.rs:386): split at '\n', a '\r' right before it belongs to the terminator; no
    trailing empty line.  (Python's splitlines() also splits at \x0b, \x0c, \x1c-\x1e, \x85, \u2028, \u2029 and a bare \r.)"""
    parts = text.split("\n")
    lines = [p[:-1] if p.endswith("\r") else p for p in parts[:-1]]
    if parts[-1] != "":
        lines.append(parts[-1])
    return lines


def read_dict(text: str) -> List[str]:
    """Dictionary file -> entries (src/oarocr/ocr.rs:277-291,386, decode.rs:120): first char of each non-empty line."""
    return [ln[0] for ln in dict_lines(text) if len(ln) > 0]


@dataclass
class DecodedTexts:
    """What oar_ctc_decode / oar_ocr_decode return (TextRecognitionOutput after the adapter's score filter)."""
    texts: List[str]
    scores: np.ndarray              # [n] f32
    char_cols: List[np.ndarray]     # per sequence: time step of each character
    char_positions: List[np.ndarray]
    seq_len: np.ndarray             # [n]
    kept: np.ndarray                # [n] bool: score >= threshold
    utf8: bytes = b""               # the concatenated texts as the library produced them
    text_offsets: Optional[np.ndarray] = None
    word_boxes: Optional[list] = None   # per region: list of [4, 2] boxes or None (oar_ocr_word_boxes; only when asked for)


class CtcDict:
    """The library-side CTCLabelDecode (processors/decode.rs:391-421,549-614): collapse, text, mean score, positions and
    the adapter's score filter in C (microseconds per region; the Python CTCLabelDecode below is the readable mirror)."""

    def __init__(self, dict_text: str, use_space_char: bool = True):
        raw = dict_text.encode("utf-8")
        self._h = C.c_void_p()
        _check(lib().oar_ctc_dict_create(raw, len(raw), int(use_space_char), C.byref(self._h)))

    @classmethod
    def from_entries(cls, entries: Sequence[str], use_space_char: bool = True) -> "CtcDict":
        return cls("".join(e + "\n" for e in entries), use_space_char)

    @property
    def classes(self) -> int:
        return int(lib().oar_ctc_dict_classes(self._h))

    @staticmethod
    def _unpack(res: TextResult, want_positions: bool = True) -> DecodedTexts:
        n = int(res.n)
        to = np.ctypeslib.as_array(res.text_offsets, shape=(n + 1,)).copy()
        raw = C.string_at(res.utf8, int(to[n]))
        sc = np.ctypeslib.as_array(res.scores, shape=(max(n, 1),)).copy()[:n]
        sl = np.ctypeslib.as_array(res.seq_len, shape=(max(n, 1),)).copy()[:n]
        kept = np.ctypeslib.as_array(res.kept, shape=(max(n, 1),)).copy()[:n].astype(bool)
        texts = [raw[to[i]:to[i + 1]].decode("utf-8", errors="replace") for i in range(n)]
        cols, pos = [], []
        if want_positions:
            co = np.ctypeslib.as_array(res.char_offsets, shape=(n + 1,)).copy()
            nc = int(co[n])
            cc = np.ctypeslib.as_array(res.char_cols, shape=(max(nc, 1),)).copy()[:nc]
            cp = np.ctypeslib.as_array(res.char_positions, shape=(max(nc, 1),)).copy()[:nc]
            cols = [cc[co[i]:co[i + 1]] for i in range(n)]
            pos = [cp[co[i]:co[i + 1]] for i in range(n)]
        return DecodedTexts(texts, sc, cols, pos, sl, kept, raw, to)

    def decode(self, indices: np.ndarray, probs: np.ndarray, batch: int, T: int, score_threshold: float = 0.0) -> DecodedTexts:
        idx = np.ascontiguousarray(indices, np.int64)
        pr = np.ascontiguousarray(probs, np.float32)
        res = TextResult()
        _check(lib().oar_ctc_decode(self._h, _p(idx) if idx.size else None, _p(pr) if pr.size else None, batch, T, C.c_float(score_threshold), C.byref(res)))
        out = self._unpack(res)
        lib().oar_text_result_free(C.byref(res))
        return out

    def decode_ocr(self, res: "OcrResult", score_threshold: float = 0.0, want_positions: bool = True, word_boxes: bool = False, want_blob: bool = False) -> DecodedTexts:
        """word_boxes: also run oar_ocr_word_boxes (return_word_box, src/oarocr/ocr.rs:860-877) on the decoded texts; the per-region
        lists of [4, 2] boxes (None where the reference yields none) are returned as DecodedTexts.word_boxes."""
        tr = TextResult()
        _check(lib().oar_ocr_decode(self._h, C.byref(res), C.c_float(score_threshold), C.byref(tr)))
        out = self._unpack(tr, want_positions)
        if word_boxes:
            wb = WordBoxes()
            try:
                _check(lib().oar_ocr_word_boxes(C.byref(res), C.byref(tr), C.byref(wb)))
                n = int(wb.n_regions)
                offs = np.ctypeslib.as_array(wb.box_offsets, shape=(n + 1,)).copy()
                flat = np.ctypeslib.as_array(wb.boxes, shape=(max(int(offs[n]), 1) * 8,)).copy()[:int(offs[n]) * 8].reshape(-1, 4, 2)
                out.word_boxes = [[b for b in flat[offs[k]:offs[k + 1]]] if offs[k + 1] > offs[k] else None for k in range(n)]
            finally:
                lib().oar_word_boxes_free(C.byref(wb))
        if want_blob:   # oar_ocr_pack: the rank's final results as one blob for the host's transport (header: "multi-process hosts")
            blob, ln = C.POINTER(C.c_uint8)(), C.c_size_t(0)
            try:
                _check(lib().oar_ocr_pack(C.byref(res), C.byref(tr), C.byref(blob), C.byref(ln)))
                out.blob = C.string_at(blob, ln.value)
            finally:
                lib().oar_blob_free(blob)
        lib().oar_text_result_free(C.byref(tr))
        return out

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.oar_ctc_dict_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class CTCLabelDecode:
    """processors/decode.rs:391-421 (from_string_list, has_explicit_blank=false) and :505-614 (decode_argmax)."""

    def __init__(self, character_list: Sequence[str], use_space_char: bool = True):
        chars = [s[0] for s in character_list if len(s) > 0]
        if use_space_char:
            chars.append(" ")
        self.character = ["\0"] + chars
        self.blank_index = 0

    def decode_argmax(self, indices: np.ndarray, probs: np.ndarray, batch: int, T: int, with_positions: bool = True):
        texts, scores, positions, cols, lens = [], [], [], [], []
        if batch == 0 or T == 0:
            return texts, scores, positions, cols, lens
        idx = np.asarray(indices, np.int64).reshape(batch, T)
        pr = np.asarray(probs, np.float32).reshape(batch, T)
        nchar = len(self.character)
        for b in range(batch):
            prev = self.blank_index
            chars, kept, ts = [], [], []
            for t in range(T):
                i = int(idx[b, t])
                if i != self.blank_index and i != prev and 0 <= i < nchar:
                    chars.append(self.character[i])
                    kept.append(pr[b, t])
                    ts.append(t)
                prev = i
            s = np.float32(0.0)
            for v in kept:                      # sequential f32 sum (decode.rs:531-535)
                s = np.float32(s + v)
            scores.append(float(s / np.float32(len(kept))) if kept else 0.0)
            texts.append("".join(chars))
            cols.append(ts)
            positions.append([float(np.float32(t) / np.float32(T)) for t in ts])
            lens.append(T)
        return texts, scores, positions, cols, lens


# ------------------------------------------------------------------------------------------------ adapters / predictors
_LIMIT = {"max": 0, "min": 1, "resize_long": 2}


class TextDetectionPredictor:
    """predictors/text_detection.rs + domain/adapters/text_detection_adapter.rs:36-79"""

    def __init__(self, model: bytes, config: Optional[TextDetectionConfig] = None, device_id: int = 0, limit_side_len: Optional[int] = None,
                 limit_type: Optional[str] = None, max_side_limit: int = 4000, host_threads: int = 0, profile: bool = False,
                 text_type: Optional[str] = None):
        self.config = config or TextDetectionConfig()
        self.config.validate()
        # text_type "seal": 736 / min preprocessing and polygon boxes (text_detection_adapter.rs:131-150, preprocessing.rs:44-62)
        seal = (text_type or "").lower() == "seal"
        limit_side_len = limit_side_len or (736 if seal else 960)
        limit_type = limit_type or ("min" if seal else "max")
        cfg = DetCfg(device_id, self.config.limit_side_len or limit_side_len, _LIMIT[self.config.limit_type or limit_type],
                     self.config.max_side_len or max_side_limit, self.config.max_candidates, 0, int(profile), host_threads,
                     int(self.config.box_type == "poly" or seal), int(self.config.score_mode == "slow"), int(self.config.use_dilation), int(self.config.gpu_contours))
        self._h = C.c_void_p()
        buf = (C.c_char * len(model)).from_buffer_copy(model)
        _check(lib().oar_det_create(C.cast(buf, C.c_void_p), len(model), C.byref(cfg), C.byref(self._h)))

    @staticmethod
    def recommended_batch_size() -> int:
        return 8   # text_detection_adapter.rs:85-87

    def predict(self, images: Sequence[np.ndarray], config: Optional[TextDetectionConfig] = None) -> List[List[Detection]]:
        if len(images) == 0:
            raise OCRError(OAR_INVALID_INPUT, "images must be a non-empty slice")   # predictors/core.rs:58-69 validate_input
        c = config or self.config
        imgs, ptrs, ws, hs = _img_arrays(images)
        res = DetResult()
        _check(lib().oar_det_run(self._h, ptrs, ws, hs, len(imgs), c.score_threshold, c.box_threshold, c.unclip_ratio, C.byref(res)))
        out = _unpack_det(res)
        lib().oar_det_result_free(C.byref(res))
        return out

    def close(self):
        """Releases the native handle (streams, HBM). Safe to call twice."""
        if getattr(self, "_h", None) and _lib is not None:
            _lib.oar_det_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class SealTextDetectionConfig:
    """domain/tasks/seal_text_detection.rs:15-43"""
    score_threshold: float = 0.2
    box_threshold: float = 0.6
    unclip_ratio: float = 0.5
    max_candidates: int = 1000


class SealTextDetectionPredictor(TextDetectionPredictor):
    """domain/adapters/seal_text_detection_adapter.rs:20-160: the DB model with seal preprocessing (736 / min), BoxType::Poly,
    ScoreMode::Fast, no dilation.  Detection.bbox is an [n, 2] polygon."""

    def __init__(self, model: bytes, config: Optional[SealTextDetectionConfig] = None, device_id: int = 0, host_threads: int = 0):
        c = config or SealTextDetectionConfig()
        super().__init__(model, TextDetectionConfig(c.score_threshold, c.box_threshold, c.unclip_ratio, c.max_candidates, box_type="poly"),
                         device_id=device_id, host_threads=host_threads, text_type="seal")


def _unpack_det(res: DetResult) -> List[List[Detection]]:
    n, nb = res.n_images, res.n_boxes
    offs = np.ctypeslib.as_array(res.box_offsets, shape=(n + 1,)).copy()
    sc = np.ctypeslib.as_array(res.scores, shape=(max(nb, 1),)).copy()[:nb]
    if res.point_offsets:   # BoxType::Poly: polygons of any size
        po = np.ctypeslib.as_array(res.point_offsets, shape=(nb + 1,)).copy()
        npt = int(res.n_points)
        flat = np.ctypeslib.as_array(res.points, shape=(max(npt, 1) * 2,)).copy()[:npt * 2].reshape(npt, 2)
        return [[Detection(flat[po[k]:po[k + 1]].copy(), float(sc[k])) for k in range(offs[i], offs[i + 1])] for i in range(n)]
    pts = np.ctypeslib.as_array(res.points, shape=(max(nb, 1) * 8,)).copy()[:nb * 8].reshape(nb, 4, 2)
    return [[Detection(pts[k].copy(), float(sc[k])) for k in range(offs[i], offs[i + 1])] for i in range(n)]


def db_postprocess(pred: np.ndarray, src_w: int, src_h: int, thresh=0.3, box_thresh=0.6, unclip_ratio=1.5, max_candidates=1000,
                   score_mode="fast", use_dilation=False, box_type="quad"):
    """Test hook: DB post-processing (a7..a12) on a host probability map through the HIP kernels."""
    pred = np.ascontiguousarray(pred, np.float32)
    h, w = pred.shape
    res = DetResult()
    _check(lib().oar_db_postprocess_ex(_p(pred), h, w, src_w, src_h, thresh, box_thresh, unclip_ratio, max_candidates, int(box_type == "poly"), int(score_mode == "slow"),
                                       int(use_dilation), C.byref(res)))
    out = _unpack_det(res)[0]
    lib().oar_det_result_free(C.byref(res))
    return out


@dataclass
class TextRecognitionOutput:
    """domain/tasks/text_recognition.rs:33-47"""
    texts: List[str]
    scores: List[float]
    char_positions: List[List[float]]
    char_col_indices: List[List[int]]
    sequence_lengths: List[int]
    indices: Optional[np.ndarray] = None
    probs: Optional[np.ndarray] = None
    tensor_width: int = 0


class TextRecognitionPredictor:
    """predictors/text_recognition.rs:33-106 + domain/adapters/text_recognition_adapter.rs:35-111"""

    def __init__(self, model: bytes, character_list: Sequence[str], score_threshold: float = 0.0, device_id: int = 0,
                 rec_image_shape=(3, 48, 320), max_img_w: int = 3200, profile: bool = False):
        self.score_threshold = score_threshold
        self.decoder = CTCLabelDecode(character_list, use_space_char=True)   # crnn.rs:384-385
        cfg = RecCfg(device_id, (C.c_uint32 * 3)(*rec_image_shape), max_img_w, 0, int(profile), 0)
        self._h = C.c_void_p()
        buf = (C.c_char * len(model)).from_buffer_copy(model)
        _check(lib().oar_rec_create(C.cast(buf, C.c_void_p), len(model), C.byref(cfg), C.byref(self._h)))

    @staticmethod
    def recommended_batch_size() -> int:
        return 256  # reference adapter reports 64 (text_recognition_adapter.rs:117-127); this backend recommends 256

    def predict(self, images: Sequence[np.ndarray]) -> TextRecognitionOutput:
        if len(images) == 0:
            raise OCRError(OAR_INVALID_INPUT, "images must be a non-empty slice")
        imgs, ptrs, ws, hs = _img_arrays(images)
        res = RecResult()
        _check(lib().oar_rec_run(self._h, ptrs, ws, hs, len(imgs), C.byref(res)))
        n, T = res.batch, res.seq_len
        idx = np.ctypeslib.as_array(res.indices, shape=(max(n * T, 1),)).copy()[:n * T]
        pr = np.ctypeslib.as_array(res.probs, shape=(max(n * T, 1),)).copy()[:n * T]
        tw = res.tensor_width
        lib().oar_rec_result_free(C.byref(res))
        texts, scores, pos, cols, lens = self.decoder.decode_argmax(idx, pr, n, T)
        for i, s in enumerate(scores):
            if not (0.0 <= s <= 1.0):   # validate_output: domain/tasks/text_recognition.rs:114-120
                raise OCRError(OAR_INVALID_INPUT, f"recognition score {s} outside [0,1]")
            if not (s >= self.score_threshold):   # adapter filter keeps the slot (text_recognition_adapter.rs:70-102)
                texts[i], pos[i], cols[i] = "", [], []
        return TextRecognitionOutput(texts, scores, pos, cols, lens, idx.reshape(n, T), pr.reshape(n, T), tw)

    def close(self):
        """Releases the native handle (streams, HBM). Safe to call twice."""
        if getattr(self, "_h", None) and _lib is not None:
            _lib.oar_rec_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ------------------------------------------------------------------------------------------------ pipeline
class OAROCRBuilder:
    """src/oarocr/ocr.rs:105-417"""

    def __init__(self, det_model: bytes, rec_model: bytes, character_dict: Sequence[str]):
        self._det, self._rec, self._dict = det_model, rec_model, list(character_dict)
        self._image_bs = None
        self._region_bs = None
        self._lanes = 0
        self._det_cfg: Optional[TextDetectionConfig] = None
        self._score_thr = 0.0
        self._device = 0
        self._profile = False
        self._host_threads = 0
        self._doc_ori = self._rectifier = self._line_ori = None
        self._text_type: Optional[str] = None

    def text_type(self, text_type: str):
        """ocr.rs:218-229: "seal" selects polygon boxes, sort_poly_boxes and bounding-rectangle crops (curved text)."""
        self._text_type = text_type
        return self

    def with_document_image_orientation_classification(self, model: bytes):
        """PP-LCNet_x1_0_doc_ori: 224x224, resize_short 256, 4 classes (domain/tasks/document_orientation.rs:46-53)"""
        self._doc_ori = model
        return self

    def with_document_image_rectification(self, model: bytes):
        """UVDoc (models/rectification/uvdoc.rs)"""
        self._rectifier = model
        return self

    def with_text_line_orientation_classification(self, model: bytes):
        """PP-LCNet_x1_0_textline_ori: direct resize to 160x80 (w x h), 2 classes (text_line_orientation.rs:25-32)"""
        self._line_ori = model
        return self

    def image_batch_size(self, n: int):
        self._image_bs = n
        return self

    def region_batch_size(self, n: int):
        self._region_bs = n
        return self

    def lanes(self, n: int):
        """This backend's addition: n complete pipelines behind the handle for submit() / wait() (oar_ocr_cfg.lanes)."""
        self._lanes = int(n)
        return self

    def text_detection_config(self, cfg: TextDetectionConfig):
        self._det_cfg = cfg
        return self

    def text_rec_score_threshold(self, t: float):
        self._score_thr = t
        return self

    def device(self, device_id: int):
        self._device = device_id
        return self

    def profile(self, on: bool = True):
        self._profile = on
        return self

    def host_threads(self, n: int):
        self._host_threads = n
        return self

    def build(self) -> "OAROCR":
        for name, v in (("image_batch_size", self._image_bs), ("region_batch_size", self._region_bs)):
            if v is not None and not (1 <= v <= 4096):   # ocr.rs:250-255,419-430
                raise OCRError(OAR_INVALID_INPUT, f"{name} must be in 1..=4096")
        # the adapter decides by the lower-cased text type (text_detection_adapter.rs:131-136), the builder's default table by the
        # exact string (ocr.rs:322)
        seal = (self._text_type or "").lower() == "seal"
        pre_lsl, pre_lt = (736, "min") if seal else (960, "max")   # db_preprocess_for_text_type (preprocessing.rs:44-62)
        if self._det_cfg is not None:
            d = self._det_cfg
            d.validate()
            thresh, box_thresh, unclip, maxc = d.score_threshold, d.box_threshold, d.unclip_ratio, d.max_candidates
            lsl, lt, msl = d.limit_side_len or pre_lsl, d.limit_type or pre_lt, d.max_side_len or 4000
            opts = (int(d.box_type == "poly" or seal), int(d.score_mode == "slow"), int(d.use_dilation), int(d.gpu_contours))
        else:   # builder defaults (ocr.rs:319-366)
            tt = self._text_type or "general"
            thresh, box_thresh, unclip = (0.3, 0.4, 2.0) if tt == "table" else (0.2, 0.6, 0.5) if tt == "seal" else (0.3, 0.6, 2.0)
            maxc, msl = 1000, 4000
            lsl, lt = (736, "min") if tt == "seal" else (960, "max")
            opts = (int(seal), 0, 0, 0)
        cfg = OcrCfg()
        cfg.det = DetCfg(self._device, lsl, _LIMIT[lt], msl, maxc, 0, int(self._profile), self._host_threads, *opts)
        cfg.rec = RecCfg(self._device, (C.c_uint32 * 3)(3, 48, 320), 3200, 0, int(self._profile), 0)
        cfg.det_thresh, cfg.det_box_thresh, cfg.det_unclip_ratio = thresh, box_thresh, unclip
        cfg.image_batch_size = self._image_bs or 0      # accelerator: adapter defaults 8 / 64 (builder_utils.rs:86-102)
        cfg.region_batch_size = self._region_bs or 0
        cfg.max_pooled_crops = 0
        # sort_detection_boxes keys on the text type (ocr.rs:699-716): 2 = sort_poly_boxes for "seal", 1 = sort_quad_boxes otherwise
        cfg.box_sort = 2 if (self._text_type or "").lower() == "seal" else 1
        cfg.lanes = self._lanes
        ocr = OAROCR(self._det, self._rec, self._dict, cfg, self._score_thr)
        if self._doc_ori or self._rectifier or self._line_ori:
            ocr.attach(ImageClassifier(self._doc_ori, device_id=self._device) if self._doc_ori else None,
                       DocumentRectifier(self._rectifier, device_id=self._device) if self._rectifier else None,
                       ImageClassifier(self._line_ori, input_hw=(80, 160), resize_short=0, device_id=self._device) if self._line_ori else None)
        return ocr


class OAROCR:
    def __init__(self, det: bytes, rec: bytes, character_dict, cfg: OcrCfg, score_threshold: float):
        self.decoder = CTCLabelDecode(character_dict, use_space_char=True)
        self.ctc = CtcDict.from_entries([s[0] for s in character_dict if len(s) > 0], use_space_char=True)
        self.score_threshold = score_threshold
        self.return_word_box = False
        self._h = C.c_void_p()
        b1 = (C.c_char * len(det)).from_buffer_copy(det)
        b2 = (C.c_char * len(rec)).from_buffer_copy(rec)
        _check(lib().oar_ocr_create(C.cast(b1, C.c_void_p), len(det), C.cast(b2, C.c_void_p), len(rec), C.byref(cfg), C.byref(self._h)))
        self._stages = (None, None, None)

    def attach(self, doc_orientation=None, rectifier=None, line_orientation=None):
        """Optional stages of OAROCR::predict (preprocess.rs:59-141, ocr.rs:757-790). The adapters stay referenced here."""
        self._stages = (doc_orientation, rectifier, line_orientation)
        h = [x._h if x is not None else None for x in self._stages]
        _check(lib().oar_ocr_attach(self._h, h[0], h[1], h[2]))

    def predict(self, images: Sequence[np.ndarray]) -> List[OAROCRResult]:
        if len(images) == 0:
            raise OCRError(OAR_INVALID_INPUT, "OCR Pipeline: images must be a non-empty slice")   # ocr.rs:525-532
        imgs, ptrs, ws, hs = _img_arrays(images)
        res = OcrResult()
        _check(lib().oar_ocr_predict(self._h, ptrs, ws, hs, len(imgs), C.byref(res)))
        out = self._assemble(res)
        lib().oar_ocr_result_free(C.byref(res))
        return out

    def predict_device(self, dev_ptrs: Sequence[int], widths: Sequence[int], heights: Sequence[int], raw: bool = False):
        """Pages already resident in HBM (pointers from DeviceBuffer). raw=True skips string assembly and returns
        (n_regions, n_ctc)."""
        n = len(dev_ptrs)
        ptrs = (C.c_void_p * n)(*dev_ptrs)
        ws = (C.c_uint32 * n)(*widths)
        hs = (C.c_uint32 * n)(*heights)
        res = OcrResult()
        _check(lib().oar_ocr_predict_device(self._h, ptrs, ws, hs, n, C.byref(res)))
        if raw:
            r = (int(res.n_regions), int(res.ctc_offsets[res.n_regions]) if res.n_regions else 0)
            lib().oar_ocr_result_free(C.byref(res))
            return r
        out = self._assemble(res)
        lib().oar_ocr_result_free(C.byref(res))
        return out

    def predict_packed(self, ptrs, ws, hs, n: int, device: bool = False, want_blob: bool = False) -> "PackedPages":
        """The metric path of bench.py (u8 pages in host memory -> sorted boxes + texts + scores on the host):
        ONE oar_ocr_predict + ONE oar_ocr_decode, results as flat arrays (no per-region Python objects).  ptrs / ws / hs are
        prepared ctypes arrays (the caller owns the page buffers, as the Rust caller owns its `RgbImage`s)."""
        res = OcrResult()
        fn = lib().oar_ocr_predict_device if device else lib().oar_ocr_predict
        _check(fn(self._h, ptrs, ws, hs, n, C.byref(res)))
        try:
            if res.point_offsets:
                raise OCRError(OAR_INVALID_INPUT, "predict_packed carries quad boxes only: use predict() for seal text")
            d = self.ctc.decode_ocr(res, self.score_threshold, want_positions=False, want_blob=want_blob)
            nr = int(res.n_regions)
            offs = np.ctypeslib.as_array(res.region_offsets, shape=(n + 1,)).copy()
            pts = np.ctypeslib.as_array(res.points, shape=(max(nr, 1) * 8,)).copy()[:nr * 8].reshape(nr, 4, 2)
        finally:
            lib().oar_ocr_result_free(C.byref(res))
        pk = PackedPages(offs, pts, d.scores, d.utf8, d.text_offsets)
        pk.blob = getattr(d, "blob", None)   # want_blob: oar_ocr_pack's wire format (== to_bytes())
        return pk

    # ---- calls in flight (oar_ocr_predict_async / oar_ocr_wait; oar_ocr_cfg.lanes)
    def submit_packed(self, ptrs, ws, hs, n: int, device: bool = False) -> int:
        """Queues one predict on the next lane and returns its ticket (the page buffers must stay alive until wait_packed)."""
        t = C.c_uint64(0)
        _check(lib().oar_ocr_predict_async(self._h, ptrs, ws, hs, n, int(device), C.byref(t)))
        return int(t.value)

    def wait_packed(self, ticket: int, n: int) -> "PackedPages":
        res = OcrResult()
        _check(lib().oar_ocr_wait(self._h, C.c_uint64(ticket), C.byref(res)))
        try:
            d = self.ctc.decode_ocr(res, self.score_threshold, want_positions=False)
            nr = int(res.n_regions)
            offs = np.ctypeslib.as_array(res.region_offsets, shape=(n + 1,)).copy()
            pts = np.ctypeslib.as_array(res.points, shape=(max(nr, 1) * 8,)).copy()[:nr * 8].reshape(nr, 4, 2)
        finally:
            lib().oar_ocr_result_free(C.byref(res))
        return PackedPages(offs, pts, d.scores, d.utf8, d.text_offsets)

    def submit(self, images: Sequence[np.ndarray]) -> int:
        """OAROCR::predict queued on the next lane; wait(ticket) returns what predict(images) would."""
        imgs, ptrs, ws, hs = _img_arrays(images)
        t = self.submit_packed(ptrs, ws, hs, len(imgs))
        self._inflight = getattr(self, "_inflight", {})
        self._inflight[t] = (imgs, ptrs, ws, hs)   # keeps the page buffers alive
        return t

    def wait(self, ticket: int) -> List[OAROCRResult]:
        res = OcrResult()
        try:
            _check(lib().oar_ocr_wait(self._h, C.c_uint64(ticket), C.byref(res)))
            return self._assemble(res)
        finally:
            getattr(self, "_inflight", {}).pop(ticket, None)
            lib().oar_ocr_result_free(C.byref(res))

    def _assemble(self, res: OcrResult) -> List[OAROCRResult]:
        n, nr = res.n_images, res.n_regions
        offs = np.ctypeslib.as_array(res.region_offsets, shape=(n + 1,)).copy()
        pang = np.ctypeslib.as_array(res.page_angle, shape=(n,)).copy()
        prect = np.ctypeslib.as_array(res.page_rectified, shape=(n,)).copy()
        page_kw = [dict(orientation_angle=(float(pang[i]) if pang[i] >= 0 else None), rectified=bool(prect[i])) for i in range(n)]
        if nr == 0:
            return [OAROCRResult(f"image_{i}", i, **page_kw[i]) for i in range(n)]
        if res.point_offsets:   # seal text: polygons of any size
            po = np.ctypeslib.as_array(res.point_offsets, shape=(nr + 1,)).copy()
            flat = np.ctypeslib.as_array(res.points, shape=(max(int(res.n_points), 1) * 2,)).copy()[:int(res.n_points) * 2].reshape(-1, 2)
            pts = [flat[po[k]:po[k + 1]] for k in range(nr)]
        else:
            pts = np.ctypeslib.as_array(res.points, shape=(nr * 8,)).copy().reshape(nr, 4, 2)
        dsc = np.ctypeslib.as_array(res.det_scores, shape=(nr,)).copy()
        cwh = np.ctypeslib.as_array(res.crop_wh, shape=(nr * 2,)).copy().reshape(nr, 2)
        sl = np.ctypeslib.as_array(res.seq_len, shape=(nr,)).copy()
        mwh = np.ctypeslib.as_array(res.max_wh_ratio, shape=(nr,)).copy()
        lang = np.ctypeslib.as_array(res.line_angle, shape=(nr,)).copy()
        dec = self.ctc.decode_ocr(res, self.score_threshold, word_boxes=self.return_word_box)   # collapse + text + score filter (+ word boxes) inside the library
        results = []
        for i in range(n):
            regions = []
            for k in range(offs[i], offs[i + 1]):
                T = int(sl[k])
                text, score, col = dec.texts[k], float(dec.scores[k]), dec.char_cols[k]
                wb = dec.word_boxes[k] if self.return_word_box else None   # oar_ocr_word_boxes (row a21)
                regions.append(TextRegion(pts[k].copy(), text, score, pts[k].copy(), pts[k].copy(), wb, float(dsc[k]), (int(cwh[k, 0]), int(cwh[k, 1])),
                                          float(lang[k]) if lang[k] >= 0 else None, float(mwh[k]), T))
            results.append(OAROCRResult(f"image_{i}", i, regions, **page_kw[i]))
        return results

    def close(self):
        """Releases the native handle (streams, HBM). Safe to call twice."""
        if getattr(self, "_h", None) and _lib is not None:
            _lib.oar_ocr_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class PackedPagesC(C.Structure):   # oar_packed_pages
    _fields_ = [("n_images", C.c_uint32), ("n_regions", C.c_uint32), ("region_offsets", C.POINTER(C.c_uint32)), ("points", C.POINTER(C.c_float)),
                ("scores", C.POINTER(C.c_float)), ("text_offsets", C.POINTER(C.c_uint64)), ("utf8", C.c_void_p)]


def shard_range(n_items: int, world_size: int, rank: int):
    """oar_shard_range: the static block partition of SURVEY 8e (rank r owns items [begin, end))"""
    a, b = C.c_uint64(0), C.c_uint64(0)
    if world_size <= 0 or rank < 0:
        raise OCRError(OAR_INVALID_INPUT, "oar_shard_range: bad world_size / rank")
    _check(lib().oar_shard_range(C.c_uint64(n_items), C.c_uint32(world_size), C.c_uint32(rank), C.byref(a), C.byref(b)))
    return int(a.value), int(b.value)


@dataclass
class PackedPages:
    """Flat result of OAROCR.predict_packed: page i owns regions [region_offsets[i], region_offsets[i+1]) in reading order;
    region k has box points[k] (4 x 2 f32, original-image coordinates), score[k] and text utf8[text_offsets[k]:text_offsets[k+1]]."""
    region_offsets: np.ndarray
    points: np.ndarray
    scores: np.ndarray
    utf8: bytes
    text_offsets: np.ndarray
    blob: Optional[bytes] = None   # predict_packed(want_blob=True): oar_ocr_pack's blob of these arrays

    def text(self, k: int) -> str:
        return self.utf8[self.text_offsets[k]:self.text_offsets[k + 1]].decode("utf-8", errors="replace")

    def to_bytes(self) -> bytes:
        """one contiguous blob (what a rank ships to rank 0): header of 3 counts, then the arrays"""
        n, nr = len(self.region_offsets) - 1, len(self.scores)
        head = np.array([n, nr, len(self.utf8)], np.int64)
        return b"".join([head.tobytes(), self.region_offsets.astype(np.uint32).tobytes(), self.points.astype(np.float32).tobytes(),
                         self.scores.astype(np.float32).tobytes(), self.text_offsets.astype(np.uint64).tobytes(), self.utf8])

    @classmethod
    def merge(cls, blobs: Sequence[bytes]) -> "PackedPages":
        """oar_packed_merge: the ranks' blobs, in rank order, as one flat result (a block partition makes that page order)"""
        n = len(blobs)
        keep = [C.create_string_buffer(b, len(b)) for b in blobs]
        ptrs = (C.POINTER(C.c_uint8) * max(n, 1))(*[C.cast(k, C.POINTER(C.c_uint8)) for k in keep])
        lens = (C.c_size_t * max(n, 1))(*[len(b) for b in blobs])
        out = PackedPagesC()
        _check(lib().oar_packed_merge(ptrs, lens, n, C.byref(out)))
        try:
            ni, nr = int(out.n_images), int(out.n_regions)
            ro = np.ctypeslib.as_array(out.region_offsets, shape=(ni + 1,)).copy()
            pts = np.ctypeslib.as_array(out.points, shape=(max(nr, 1) * 8,)).copy()[:nr * 8].reshape(nr, 4, 2)
            sc = np.ctypeslib.as_array(out.scores, shape=(max(nr, 1),)).copy()[:nr]
            to = np.ctypeslib.as_array(out.text_offsets, shape=(nr + 1,)).copy()
            utf8 = C.string_at(out.utf8, int(to[nr]))
        finally:
            lib().oar_packed_pages_free(C.byref(out))
        return cls(ro, pts, sc, utf8, to)

    @classmethod
    def from_bytes(cls, blob: bytes) -> "PackedPages":
        n, nr, nb = (int(v) for v in np.frombuffer(blob, np.int64, 3))
        o = 24
        ro = np.frombuffer(blob, np.uint32, n + 1, o); o += 4 * (n + 1)
        pts = np.frombuffer(blob, np.float32, nr * 8, o).reshape(nr, 4, 2); o += 32 * nr
        sc = np.frombuffer(blob, np.float32, nr, o); o += 4 * nr
        to = np.frombuffer(blob, np.uint64, nr + 1, o); o += 8 * (nr + 1)
        return cls(ro, pts, sc, bytes(blob[o:o + nb]), to)


def ctc_word_boxes(line_bbox: np.ndarray, text: str, col_indices, seq_len: int, wh_ratio: float, max_wh_ratio: float):
    """OAROCR::ctc_word_boxes (src/oarocr/ocr.rs:949-1020) through the C ABI (oar_ctc_word_boxes): one [4, 2] box per character."""
    pts = np.ascontiguousarray(line_bbox, np.float32).reshape(-1, 2)
    cols = np.ascontiguousarray(col_indices, np.uint32)
    raw = text.encode("utf-8")
    n = C.c_uint32(0)
    args = (_p(pts) if pts.size else None, pts.shape[0], raw, len(raw), _p(cols) if cols.size else None, cols.size, int(seq_len), C.c_float(wh_ratio), C.c_float(max_wh_ratio))
    _check(lib().oar_ctc_word_boxes(*args, None, 0, C.byref(n)))
    out = np.empty((n.value, 4, 2), np.float32)
    if n.value:
        _check(lib().oar_ctc_word_boxes(*args, _p(out), n.value, C.byref(n)))
    return [b for b in out]


def char_positions_to_word_boxes(line_bbox: np.ndarray, char_positions, char_count: int):
    """OAROCR::char_positions_to_word_boxes (src/oarocr/ocr.rs:1036-1072) through the C ABI."""
    pts = np.ascontiguousarray(line_bbox, np.float32).reshape(-1, 2)
    pos = np.ascontiguousarray(char_positions, np.float32)
    n = C.c_uint32(0)
    args = (_p(pts) if pts.size else None, pts.shape[0], _p(pos) if pos.size else None, pos.size, int(char_count))
    _check(lib().oar_char_positions_to_word_boxes(*args, None, 0, C.byref(n)))
    out = np.empty((n.value, 4, 2), np.float32)
    if n.value:
        _check(lib().oar_char_positions_to_word_boxes(*args, _p(out), n.value, C.byref(n)))
    return [b for b in out]


# ------------------------------------------------------------------------------------------------ layout detection
class PpDocCfg(C.Structure):
    _fields_ = [("score_threshold", C.c_float), ("class_thresholds", C.POINTER(C.c_float)), ("layout_nms", C.c_int32), ("image_class_id", C.c_int32),
                ("formula_class_id", C.c_int32), ("class_merge_modes", C.POINTER(C.c_int32))]


MERGE_MODES = {"large": 0, "union": 1, "small": 2}   # MergeBboxMode (domain/tasks/layout_detection.rs:17-25)


def _ppdoc_cfg(num_classes, class_labels, score_threshold, class_thresholds, layout_nms, class_merge_modes):
    """LayoutDetectionConfig + LayoutModelConfig -> oar_ppdoc_cfg (label-keyed maps become class-id-indexed arrays, as the adapter builds them, :645-675).
    Returns (cfg, keepalive)."""
    cfg = PpDocCfg()
    cfg.score_threshold = score_threshold
    cfg.layout_nms = int(bool(layout_nms))
    by_label = {v: k for k, v in class_labels.items()}
    cfg.image_class_id = by_label.get("image", -1)
    cfg.formula_class_id = by_label.get("formula", -1)
    keep = []
    if class_thresholds:
        thr = np.full(max(num_classes, 1), np.nan, np.float32)
        for cid, label in class_labels.items():
            if label in class_thresholds and 0 <= cid < num_classes:
                thr[cid] = class_thresholds[label]
        keep.append(thr)
        cfg.class_thresholds = thr.ctypes.data_as(C.POINTER(C.c_float))
    if class_merge_modes:
        mm = np.full(max(num_classes, 1), -1, np.int32)
        for cid, label in class_labels.items():
            if label in class_merge_modes and 0 <= cid < num_classes:
                mm[cid] = MERGE_MODES[class_merge_modes[label]]
        keep.append(mm)
        cfg.class_merge_modes = mm.ctypes.data_as(C.POINTER(C.c_int32))
    return cfg, keep


LAYOUT_FILTERS = {"triangle": 0, "catmullrom": 1, "lanczos3": 2}
LAYOUT_MODEL_TYPES = {"picodet": 0, "rtdetr": 1, "pp-doclayout": 2}


@dataclass
class LayoutModelConfig:
    """domain/adapters/layout_detection_adapter.rs:38-52 + the model's ScaleAwareDetectorPreprocessConfig (scale_aware_detector.rs:49-75)"""
    model_name: str
    num_classes: int
    class_labels: dict
    model_type: str = "picodet"
    input_size: Optional[tuple] = (800, 608)

    @staticmethod
    def picodet_layout_1x():                       # layout_detection_adapter.rs:56-71
        return LayoutModelConfig("picodet_layout_1x", 5, {0: "text", 1: "title", 2: "list", 3: "table", 4: "figure"}, "picodet", (800, 608))

    @staticmethod
    def picodet_layout_1x_table():                 # :74-85
        return LayoutModelConfig("picodet_layout_1x_table", 1, {0: "table"}, "picodet", (800, 608))

    _PPDOC23 = ["paragraph_title", "image", "text", "number", "abstract", "content", "figure_title", "formula", "table", "table_title", "reference", "doc_title",
                "footnote", "header", "algorithm", "footer", "seal", "chart_title", "chart", "formula_number", "header_image", "footer_image", "aside_text"]
    _PPDOCV2 = ["abstract", "algorithm", "aside_text", "chart", "content", "display_formula", "doc_title", "figure_title", "footer", "footer_image", "footnote",
                "formula_number", "header", "header_image", "image", "inline_formula", "number", "paragraph_title", "reference", "reference_content", "seal", "table",
                "text", "vertical_text", "vision_footnote"]

    @staticmethod
    def pp_doclayout_s():                          # layout_detection_adapter.rs:242-275 (23 classes, 480 x 480)
        return LayoutModelConfig("pp-doclayout-s", 23, dict(enumerate(LayoutModelConfig._PPDOC23)), "pp-doclayout", (480, 480))

    @staticmethod
    def pp_doclayoutv2():                          # :383-418 (25 classes, 800 x 800; rows carry (col, row) reading-order columns)
        return LayoutModelConfig("pp-doclayoutv2", 25, dict(enumerate(LayoutModelConfig._PPDOCV2)), "pp-doclayout", (800, 800))

    def preprocess(self):
        """(filter, bgr, mean, std) of the model family"""
        if self.model_type == "pp-doclayout":
            return "catmullrom", False, (0.0, 0.0, 0.0), (1.0, 1.0, 1.0)
        return "lanczos3", True, (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


@dataclass
class LayoutDetectionConfig:
    """domain/tasks/layout_detection.rs:47-78"""
    score_threshold: float = 0.5
    max_elements: int = 100
    class_thresholds: Optional[dict] = None
    nms_threshold: float = 0.5
    layout_unclip_ratio: Optional[object] = None    # float | (w, h) | {class_id: (w, h)}
    class_merge_modes: Optional[dict] = None        # {label: "large" | "union" | "small"}
    layout_nms: bool = True

    def get_class_threshold(self, name: str) -> float:   # :287-292
        return (self.class_thresholds or {}).get(name, self.score_threshold)

    @staticmethod
    def with_pp_doclayoutv2_defaults():                  # domain/tasks/layout_detection.rs:140-205
        low = {"display_formula", "doc_title", "inline_formula", "paragraph_title", "text", "vertical_text"}
        large = {"chart", "display_formula", "doc_title", "inline_formula", "paragraph_title"}
        thr = {n: (0.45 if n == "seal" else 0.4 if n in low else 0.5) for n in LayoutModelConfig._PPDOCV2}
        modes = {n: ("large" if n in large else "union") for n in LayoutModelConfig._PPDOCV2}
        return LayoutDetectionConfig(score_threshold=0.4, max_elements=100, class_thresholds=thr, nms_threshold=0.5, layout_unclip_ratio=(1.0, 1.0),
                                     class_merge_modes=modes, layout_nms=True)


@dataclass
class LayoutDetectionElement:
    bbox: np.ndarray          # [4, 2]
    element_type: str
    score: float


def unclip_boxes(boxes: np.ndarray, classes, ratio) -> np.ndarray:
    """processors/layout_postprocess.rs:636-690 on [k, 4] x1 y1 x2 y2 boxes (f32, the reference's operation order)"""
    f = np.float32
    out = np.asarray(boxes, np.float32).copy()
    for i, c in enumerate(classes):
        if isinstance(ratio, dict):
            wr, hr = ratio.get(int(c), (1.0, 1.0))
        elif isinstance(ratio, (tuple, list)):
            wr, hr = ratio
        else:
            wr = hr = ratio
        wr, hr = f(wr), f(hr)
        if abs(wr - f(1.0)) < f(1e-6) and abs(hr - f(1.0)) < f(1e-6):
            continue
        x0, y0, x1, y1 = out[i]
        w, h = f(x1 - x0), f(y1 - y0)
        cx, cy = f(x0 + f(w * f(0.5))), f(y0 + f(h * f(0.5)))
        hw, hh = f(f(w * wr) * f(0.5)), f(f(h * hr) * f(0.5))
        out[i] = [f(cx - hw), f(cy - hh), f(cx + hw), f(cy + hh)]
    return out


class LayoutDetectionPredictor:
    """LayoutDetectionAdapter (domain/adapters/layout_detection_adapter.rs) for the PicoDet / RT-DETR families: the model half
    (resize, normalise, graph, LayoutPostProcess) is ONE C call into HBM-resident kernels (oar_layout_run); class labels, per-class
    thresholds, layout_unclip_ratio and max_elements are applied here as the adapter's postprocess does (:540-629)."""

    def __init__(self, onnx_bytes: bytes, model_config: Optional[LayoutModelConfig] = None, config: Optional[LayoutDetectionConfig] = None, device_id: int = 0):
        self.model_config = model_config or LayoutModelConfig.picodet_layout_1x()
        self.config = config or LayoutDetectionConfig()
        filt, bgr, mean, std = self.model_config.preprocess()
        c = LayoutCfg()
        c.device_id = device_id
        c.input_h, c.input_w = self.model_config.input_size or (800, 800)
        c.resize_filter, c.color_bgr, c.scale = LAYOUT_FILTERS[filt], int(bgr), 1.0 / 255.0
        c.mean, c.std = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
        c.num_classes, c.model_type = self.model_config.num_classes, LAYOUT_MODEL_TYPES.get(self.model_config.model_type, 0)
        c.score_threshold, c.nms_threshold, c.max_detections = self.config.score_threshold, self.config.nms_threshold, self.config.max_elements
        buf = (C.c_char * len(onnx_bytes)).from_buffer_copy(onnx_bytes)
        self._h = C.c_void_p()
        _check(lib().oar_layout_create(C.cast(buf, C.c_void_p), len(onnx_bytes), C.byref(c), C.byref(self._h)))
        self.input_hw = (int(c.input_h), int(c.input_w))

    def preprocess(self, image: np.ndarray) -> np.ndarray:
        img = np.ascontiguousarray(image, np.uint8)
        out = np.empty((3, self.input_hw[0], self.input_hw[1]), np.float32)
        _check(lib().oar_layout_preprocess(self._h, _p(img), img.shape[1], img.shape[0], _p(out)))
        return out

    def detect_raw(self, images: Sequence[np.ndarray]):
        """LayoutPostProcess::apply's output: per image (boxes [k, 4], classes [k], scores [k]); plus the prediction width"""
        imgs, ptrs, ws, hs = _img_arrays(images)
        res = LayoutResult()
        _check(lib().oar_layout_run(self._h, ptrs, ws, hs, len(imgs), C.byref(res)))
        try:
            return _unpack_layout(res), int(res.feature_dim)
        finally:
            lib().oar_layout_result_free(C.byref(res))

    def detect_raw_ppdoc(self, images: Sequence[np.ndarray], cfg: "LayoutDetectionConfig"):
        """postprocess_pp_doclayout up to the reading-order sort (oar_layout_run_ppdoc): per image (boxes, classes, scores) in final order"""
        imgs, ptrs, ws, hs = _img_arrays(images)
        pc, keep = _ppdoc_cfg(self.model_config.num_classes, self.model_config.class_labels, cfg.score_threshold, cfg.class_thresholds, cfg.layout_nms, cfg.class_merge_modes)
        res = LayoutResult()
        _check(lib().oar_layout_run_ppdoc(self._h, ptrs, ws, hs, len(imgs), C.byref(pc), C.byref(res)))
        del keep
        try:
            return _unpack_layout(res), int(res.feature_dim)
        finally:
            lib().oar_layout_result_free(C.byref(res))

    def predict(self, images: Sequence[np.ndarray], config: Optional[LayoutDetectionConfig] = None):
        cfg = config or self.config
        if self.model_config.model_type == "pp-doclayout":        # layout_detection_adapter.rs:545-547: the adapter's own post-processing
            per_image, feat = self.detect_raw_ppdoc(images, cfg)
            out = []
            for boxes, classes, scores in per_image:
                if cfg.layout_unclip_ratio is not None:
                    boxes = unclip_boxes(boxes, classes, cfg.layout_unclip_ratio)
                els = []
                for b, c, s in zip(boxes, classes, scores):          # (no second threshold here: :813-831)
                    name = self.model_config.class_labels.get(int(c), "unknown")
                    els.append(LayoutDetectionElement(np.array([[b[0], b[1]], [b[2], b[1]], [b[2], b[3]], [b[0], b[3]]], np.float32), name, float(s)))
                    if len(els) >= cfg.max_elements:
                        break
                out.append(els)
            self.is_reading_order_sorted = feat in (7, 8)
            return out
        per_image, feat = self.detect_raw(images)
        out = []
        for boxes, classes, scores in per_image:
            if cfg.layout_unclip_ratio is not None:
                boxes = unclip_boxes(boxes, classes, cfg.layout_unclip_ratio)
            if cfg.class_merge_modes is not None:                # apply_nms_with_merge (:577-587)
                modes = [cfg.class_merge_modes.get(self.model_config.class_labels.get(c, "unknown"), "large") for c in range(self.model_config.num_classes)]
                boxes, classes, scores = host_nms_with_merge(boxes, classes, scores, modes, cfg.nms_threshold, cfg.max_elements)
            els = []
            for b, c, s in zip(boxes, classes, scores):
                name = self.model_config.class_labels.get(int(c), "unknown")
                if s >= np.float32(cfg.get_class_threshold(name)):
                    els.append(LayoutDetectionElement(np.array([[b[0], b[1]], [b[2], b[1]], [b[2], b[3]], [b[0], b[3]]], np.float32), name, float(s)))
                    if len(els) >= cfg.max_elements:
                        break
            out.append(els)
        self.is_reading_order_sorted = feat in (7, 8)
        return out

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.oar_layout_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _unpack_layout(res: "LayoutResult"):
    n, nb = int(res.n_images), int(res.n_boxes)
    offs = np.ctypeslib.as_array(res.box_offsets, shape=(n + 1,)).copy()
    boxes = np.ctypeslib.as_array(res.boxes, shape=(max(nb, 1) * 4,)).copy()[:nb * 4].reshape(nb, 4)
    cls = np.ctypeslib.as_array(res.classes, shape=(max(nb, 1),)).copy()[:nb]
    sc = np.ctypeslib.as_array(res.scores, shape=(max(nb, 1),)).copy()[:nb]
    return [(boxes[offs[i]:offs[i + 1]], cls[offs[i]:offs[i + 1]], sc[offs[i]:offs[i + 1]]) for i in range(n)]


def k_ppdoc_postprocess(pred, src_wh, num_classes, class_labels, score_threshold=0.5, class_thresholds=None, layout_nms=True, class_merge_modes=None):
    """pred: [n_images, rows, feat]; the HIP PP-DocLayout post-processing on caller-supplied predictions (oar_k_ppdoc_postprocess)."""
    pred = np.ascontiguousarray(pred, np.float32)
    n, rows, feat = pred.shape
    wh = np.ascontiguousarray(src_wh, np.float32).reshape(n, 2)
    pc, keep = _ppdoc_cfg(num_classes, class_labels, score_threshold, class_thresholds, layout_nms, class_merge_modes)
    res = LayoutResult()
    _check(lib().oar_k_ppdoc_postprocess(_p(pred) if pred.size else None, n, rows, feat, _p(wh), num_classes, C.byref(pc), C.byref(res)))
    del keep
    try:
        return _unpack_layout(res)
    finally:
        lib().oar_layout_result_free(C.byref(res))


def host_nms_with_merge(boxes, classes, scores, mode_of_class, nms_threshold=0.5, max_detections=100):
    """apply_nms_with_merge (processors/layout_postprocess.rs:743-841) through the C ABI; mode_of_class: per class id "large" | "union" | "small"."""
    b = np.ascontiguousarray(boxes, np.float32).reshape(-1, 4)
    c = np.ascontiguousarray(classes, np.int32)
    s = np.ascontiguousarray(scores, np.float32)
    modes = np.ascontiguousarray([MERGE_MODES[m] for m in mode_of_class], np.int32)
    n = len(s)
    ob, oc, os_ = np.zeros((max(n, 1), 4), np.float32), np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.float32)
    L = lib()
    L.oar_host_nms_with_merge.restype = C.c_int32
    k = L.oar_host_nms_with_merge(_p(b), _p(c), _p(s), n, _p(modes), len(modes), C.c_float(nms_threshold), int(max_detections), _p(ob), _p(oc), _p(os_))
    if k < 0:
        raise OCRError(OAR_INVALID_INPUT, "oar_host_nms_with_merge failed")
    return ob[:k].copy(), oc[:k].copy(), os_[:k].copy()


def k_resize_filter(rgb, nw, nh, filter="lanczos3"):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w, _ = rgb.shape
    out = np.empty((nh, nw, 3), np.uint8)
    _check(lib().oar_k_resize_filter(_p(rgb), w, h, nw, nh, LAYOUT_FILTERS[filter], _p(out)))
    return out


def k_layout_postprocess(pred, src_wh, num_classes, score_threshold=0.5, nms_threshold=0.5, max_detections=100, model_type="picodet"):
    """pred: [n_images, rows, feat]; src_wh: [n_images, 2] (width, height).  The HIP LayoutPostProcess on caller-supplied predictions."""
    pred = np.ascontiguousarray(pred, np.float32)
    n, rows, feat = pred.shape
    wh = np.ascontiguousarray(src_wh, np.float32).reshape(n, 2)
    res = LayoutResult()
    _check(lib().oar_k_layout_postprocess(_p(pred) if pred.size else None, n, rows, feat, _p(wh), num_classes, LAYOUT_MODEL_TYPES.get(model_type, 0), C.c_float(score_threshold),
                                          C.c_float(nms_threshold), max_detections, C.byref(res)))
    try:
        return _unpack_layout(res)
    finally:
        lib().oar_layout_result_free(C.byref(res))


# ------------------------------------------------------------------------------------------------ device buffers / profiling
class DeviceBuffer:
    def __init__(self, data: np.ndarray, device_id: int = 0):
        data = np.ascontiguousarray(data)
        self.nbytes = data.nbytes
        self.ptr = C.c_void_p()
        _check(lib().oar_dev_alloc(device_id, self.nbytes, C.byref(self.ptr)))
        _check(lib().oar_dev_upload(self.ptr, _p(data), self.nbytes))

    def free(self):
        if self.ptr:
            lib().oar_dev_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def prof_enable(on: bool = True):
    lib().oar_prof_enable(int(on))


def prof_filter(name: str = ""):
    lib().oar_prof_filter(name.encode() if name else None)


def prof_sampling(stride: int = 1, phase: int = 0):
    """Time launch i (counted from this call) iff i % stride == phase; phase < 0: none."""
    lib().oar_prof_sampling(stride, phase)


def prof_reset():
    lib().oar_prof_reset()


def prof_snapshot(max_entries: int = 512):
    arr = (ProfEntry * max_entries)()
    n = lib().oar_prof_snapshot(arr, max_entries)
    return [{"name": arr[i].name.decode(), "launches": int(arr[i].launches), "total_ms": arr[i].total_ms,
             "alg_bytes": arr[i].alg_bytes, "alg_flops": arr[i].alg_flops} for i in range(min(n, max_entries))]


# stand-alone kernels (parity hooks)
def k_normalize(rgb, alpha, beta, src=(0, 1, 2), layout="chw"):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w, _ = rgb.shape
    out = np.empty((3, h, w) if layout == "chw" else (h, w, 3), np.float32)
    a, b = np.ascontiguousarray(alpha, np.float32), np.ascontiguousarray(beta, np.float32)
    s = (C.c_int32 * 3)(*src)
    _check(lib().oar_k_normalize(_p(rgb), w, h, s, a.ctypes.data_as(C.POINTER(C.c_float)), b.ctypes.data_as(C.POINTER(C.c_float)),
                                 0 if layout == "chw" else 1, _p(out)))
    return out


def k_rec_preprocess(crops, img_h=48, img_w=320, max_img_w=3200, flips=None):
    """flips[i] truthy: crop i is packed as its rotate180 (text-line orientation class 1) without a rotated copy."""
    imgs, ptrs, ws, hs = _img_arrays(crops)
    tw = C.c_uint32(0)
    fl = None if flips is None else np.ascontiguousarray(np.asarray(flips, bool).astype(np.uint8))
    _check(lib().oar_k_rec_preprocess_flip(ptrs, ws, hs, None if fl is None else _p(fl), len(imgs), img_h, img_w, max_img_w, None, C.byref(tw)))
    out = np.empty((len(imgs), 3, img_h, tw.value), np.float32)
    _check(lib().oar_k_rec_preprocess_flip(ptrs, ws, hs, None if fl is None else _p(fl), len(imgs), img_h, img_w, max_img_w, _p(out), C.byref(tw)))
    return out


def k_resize_triangle(rgb, nw, nh):
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w, _ = rgb.shape
    out = np.empty((nh, nw, 3), np.uint8)
    _check(lib().oar_k_resize_triangle(_p(rgb), w, h, nw, nh, _p(out)))
    return out


def k_threshold(pred, thresh):
    pred = np.ascontiguousarray(pred, np.float32)
    out = np.empty(pred.shape, np.uint8)
    _check(lib().oar_k_threshold(_p(pred), pred.size, thresh, _p(out)))
    return out


def k_dilate(mask):
    mask = np.ascontiguousarray(mask, np.uint8)
    out = np.empty_like(mask)
    _check(lib().oar_k_dilate(_p(mask), mask.shape[0], mask.shape[1], _p(out)))
    return out


def k_poly_scores(pred, polys):
    """polys: list of [n_i, 2] point arrays"""
    pred = np.ascontiguousarray(pred, np.float32)
    h, w = pred.shape
    pts = np.ascontiguousarray(np.concatenate([np.asarray(p, np.float32).reshape(-1, 2) for p in polys]) if polys else np.zeros((0, 2), np.float32))
    counts = (C.c_uint32 * max(len(polys), 1))(*[len(p) for p in polys])
    out = np.zeros(len(polys), np.float32)
    _check(lib().oar_k_poly_scores(_p(pred), h, w, _p(pts), counts, len(polys), _p(out)))
    return out


def k_ctc_argmax(probs):
    probs = np.ascontiguousarray(probs, np.float32)
    v = probs.shape[-1]
    rows = probs.size // v if v else 0
    idx = np.zeros(rows, np.int64)
    p = np.zeros(rows, np.float32)
    _check(lib().oar_k_ctc_argmax(_p(probs), rows, v, _p(idx), _p(p)))
    return idx, p


def k_box_scores(pred, boxes):
    pred = np.ascontiguousarray(pred, np.float32)
    boxes = np.ascontiguousarray(boxes, np.float32).reshape(-1, 8)
    h, w = pred.shape
    out = np.zeros(boxes.shape[0], np.float32)
    _check(lib().oar_k_box_scores(_p(pred), h, w, _p(boxes), boxes.shape[0], _p(out)))
    return out


def k_rotate_crop(img, box):
    img = np.ascontiguousarray(img, np.uint8)
    h, w, _ = img.shape
    box = np.ascontiguousarray(box, np.float32).reshape(8)
    cap = 3 * (w + h + 8) ** 2
    out = np.empty(cap, np.uint8)
    ow, oh = C.c_uint32(0), C.c_uint32(0)
    _check(lib().oar_k_rotate_crop(_p(img), w, h, box.ctypes.data_as(C.POINTER(C.c_float)), _p(out), cap, C.byref(ow), C.byref(oh)))
    if ow.value == 0:
        return None
    return out[:ow.value * oh.value * 3].reshape(oh.value, ow.value, 3).copy()


# host-side geometry hooks (no GPU needed)
def host_candidates(mask, max_candidates=1000, max_bands=1):
    mask = np.ascontiguousarray(mask, np.uint8)
    h, w = mask.shape
    out = np.zeros((max_candidates, 4, 2), np.float32)
    n = lib().oar_host_candidates(_p(mask), w, h, max_candidates, max_bands, _p(out), max_candidates)
    if n < 0:
        raise OCRError(OAR_INTERNAL, "oar_host_candidates failed")
    return out[:n].copy()


def host_contours(mask, max_contours=100000, max_bands=1, bits=False):
    """[(points [n,2] int32 (x, y) in tracing order, type 0 outer / 1 hole)] in discovery order (a8).  bits: through the
    bit-plane read-back format of the detector."""
    mask = np.ascontiguousarray(mask, np.uint8)
    h, w = mask.shape
    cap = 4 * h * w + 16
    offs = np.zeros(max_contours + 1, np.int64)
    pts = np.zeros((cap, 2), np.int32)
    types = np.zeros(max_contours, np.int32)
    fn = lib().oar_host_contours_bits if bits else lib().oar_host_contours
    n = fn(_p(mask), w, h, max_contours, max_bands, _p(offs), _p(pts), _p(types), cap)
    if n < 0:
        raise OCRError(OAR_INTERNAL, "oar_host_contours failed")
    return [(pts[offs[i]:offs[i + 1]].copy(), int(types[i])) for i in range(n)]


def k_contours(mask, max_contours=100000):
    """a8 through the GPU border follower (contours.hip): same result format as host_contours."""
    mask = np.ascontiguousarray(mask, np.uint8)
    h, w = mask.shape
    cap = 4 * h * w + 16
    offs = np.zeros(max_contours + 1, np.int64)
    pts = np.zeros((cap, 2), np.int32)
    types = np.zeros(max_contours, np.int32)
    n = C.c_int32(0)
    _check(lib().oar_k_contours(_p(mask), w, h, max_contours, C.byref(n), _p(offs), _p(pts), _p(types), cap))
    return [(pts[offs[i]:offs[i + 1]].copy(), int(types[i])) for i in range(n.value)]


def k_unclip(boxes, ratio, cap_points=96):
    """pp::unclip_quads on n boxes [n, 4, 2]: list of [m, 2] polygons; None where the kernel leaves the box to the host."""
    b = np.ascontiguousarray(boxes, np.float32).reshape(-1, 8)
    n = len(b)
    counts = np.zeros(max(n, 1), np.int32)
    pts = np.zeros((max(n, 1), cap_points, 2), np.float32)
    _check(lib().oar_k_unclip(_p(b), n, C.c_float(ratio), _p(counts), _p(pts), cap_points))
    return [None if counts[i] < 0 else pts[i, :counts[i]].copy() for i in range(n)]


def host_unclip(box, ratio):
    box = np.ascontiguousarray(box, np.float32).reshape(8)
    out = np.zeros((1024, 2), np.float32)
    n = lib().oar_host_unclip(_p(box), ratio, _p(out), 1024)
    return out[:max(n, 0)].copy()


def host_approx_poly_dp(points, epsilon):
    """geometry.rs:453-561 on the open chain `points`."""
    pts = np.ascontiguousarray(points, np.float32).reshape(-1, 2)
    out = np.zeros((max(len(pts), 1), 2), np.float32)
    n = lib().oar_host_approx_poly_dp(_p(pts), pts.shape[0], C.c_float(epsilon), _p(out), out.shape[0])
    return out[:max(n, 0)].copy()


def host_perimeter(points) -> float:
    pts = np.ascontiguousarray(points, np.float32).reshape(-1, 2)
    return float(lib().oar_host_perimeter(_p(pts), pts.shape[0]))


def host_unclip_poly(points, ratio):
    """db_bitmap.rs:279-368 for any polygon; an empty array where the reference drops the box."""
    pts = np.ascontiguousarray(points, np.float32).reshape(-1, 2)
    cap = 64 * len(pts) + 1024
    out = np.zeros((cap, 2), np.float32)
    n = lib().oar_host_unclip_poly(_p(pts), pts.shape[0], C.c_float(ratio), _p(out), cap)
    if n < 0:
        raise RuntimeError("oar_host_unclip_poly failed")
    return out[:n].copy()


def host_offset_ring(ring, radius):
    r = np.ascontiguousarray(ring, np.int64).reshape(-1, 2)
    cap = 64 * len(r) + 1024
    out = np.zeros((cap, 2), np.int64)
    n = lib().oar_host_offset_ring(_p(r), r.shape[0], C.c_double(radius), _p(out), cap)
    return out[:max(n, 0)].copy()


def host_ring_outline(raw, negative=False):
    """None when the outline is not exactly one loop."""
    r = np.ascontiguousarray(raw, np.int64).reshape(-1, 2)
    cap = 4 * len(r) + 64
    out = np.zeros((cap, 2), np.int64)
    n = lib().oar_host_ring_outline(_p(r), r.shape[0], 1 if negative else 0, _p(out), cap)
    return out[:n].copy() if n > 0 else None


def host_sort_poly_boxes(polys):
    if not polys:
        return np.zeros(0, np.int32)
    offs = np.zeros(len(polys) + 1, np.uint32)
    offs[1:] = np.cumsum([len(p) for p in polys])
    pts = np.ascontiguousarray(np.concatenate([np.asarray(p, np.float32).reshape(-1, 2) for p in polys] + [np.zeros((0, 2), np.float32)]), np.float32)
    if len(pts) == 0:
        pts = np.zeros((1, 2), np.float32)
    order = np.zeros(len(polys), np.int32)
    lib().oar_host_sort_poly_boxes(_p(pts), _p(offs), len(polys), _p(order))
    return order


def host_mini_box(points):
    pts = np.ascontiguousarray(points, np.float32).reshape(-1, 2)
    out = np.zeros((4, 2), np.float32)
    ms = C.c_float(0)
    ok = lib().oar_host_mini_box(_p(pts), pts.shape[0], _p(out), C.byref(ms))
    return (out, float(ms.value)) if ok == 1 else None


def host_convex_hull(points):
    """convex_hull (processors/geometry.rs:226-271) of [n, 2] points through the C ABI (oar_host_convex_hull): hull vertices in scan order."""
    pts = np.ascontiguousarray(points, np.float32).reshape(-1, 2)
    out = np.zeros((max(pts.shape[0], 1), 2), np.float32)
    n = lib().oar_host_convex_hull(_p(pts), pts.shape[0], _p(out), out.shape[0])
    if n < 0:
        raise OCRError(1, "oar_host_convex_hull failed")
    return out[:n].copy()


def host_sort_quad_boxes(boxes):
    b = np.ascontiguousarray(boxes, np.float32).reshape(-1, 8)
    order = np.zeros(b.shape[0], np.int32)
    if b.shape[0]:
        lib().oar_host_sort_quad_boxes(_p(b), b.shape[0], _p(order))
    return order


def host_plan_crop(img_w, img_h, box):
    b = np.ascontiguousarray(box, np.float32).reshape(8)
    plan = np.zeros(8, np.int32)
    inv = np.zeros(9, np.float32)
    lib().oar_host_plan_crop(img_w, img_h, _p(b), _p(plan), _p(inv))
    return plan, inv


# ---------------------------------------------------------------------------------------------- config-5 adapters
@dataclass
class Classification:
    class_id: int
    score: float


class ImageClassifier:
    """DocumentOrientationAdapter / TextLineOrientationAdapter over PPLCNetModel (pp_lcnet.rs:139-330)."""

    def __init__(self, model: bytes, input_hw=(224, 224), resize_short: int = 256, topk: int = 1, device_id: int = 0, batch: int = 0):
        self._h = C.c_void_p()
        self.input_hw = tuple(input_hw)
        cfg = ClsCfg(device_id, input_hw[0], input_hw[1], resize_short or 0, topk, batch)
        buf = (C.c_char * len(model)).from_buffer_copy(model)
        _check(lib().oar_cls_create(C.cast(buf, C.c_void_p), len(model), C.byref(cfg), C.byref(self._h)))

    def predict(self, images: Sequence[np.ndarray]) -> List[List[Classification]]:
        if len(images) == 0:
            return []
        imgs, ptrs, ws, hs = _img_arrays(images)
        res = ClsResult()
        _check(lib().oar_cls_run(self._h, ptrs, ws, hs, len(imgs), C.byref(res)))
        n, k = res.n_images, res.topk
        ids = np.ctypeslib.as_array(res.class_ids, shape=(n * k,)).copy().reshape(n, k)
        sc = np.ctypeslib.as_array(res.scores, shape=(n * k,)).copy().reshape(n, k)
        lib().oar_cls_result_free(C.byref(res))
        return [[Classification(int(ids[i, j]), float(sc[i, j])) for j in range(k) if ids[i, j] >= 0] for i in range(n)]

    def preprocess(self, images: Sequence[np.ndarray]) -> np.ndarray:
        imgs, ptrs, ws, hs = _img_arrays(images)
        out = np.zeros((len(imgs), 3, self.input_hw[0], self.input_hw[1]), np.float32)
        _check(lib().oar_cls_preprocess(self._h, ptrs, ws, hs, len(imgs), _p(out)))
        return out

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.oar_cls_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DocumentRectifier:
    """UVDocRectifierAdapter (models/rectification/uvdoc.rs:82-109,166-207)."""

    def __init__(self, model: bytes, target_hw=(512, 512), device_id: int = 0):
        self._h = C.c_void_p()
        cfg = RectCfg(device_id, target_hw[0], target_hw[1])
        buf = (C.c_char * len(model)).from_buffer_copy(model)
        _check(lib().oar_rect_create(C.cast(buf, C.c_void_p), len(model), C.byref(cfg), C.byref(self._h)))

    def predict(self, images: Sequence[np.ndarray]) -> List[np.ndarray]:
        outs = []
        for im in images:
            im = np.ascontiguousarray(im, np.uint8)
            out = np.zeros_like(im)
            _check(lib().oar_rect_run(self._h, _p(im), im.shape[1], im.shape[0], _p(out)))
            outs.append(out)
        return outs

    def close(self):
        if getattr(self, "_h", None) and _lib is not None:
            _lib.oar_rect_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def k_rotate_rgb(rgb: np.ndarray, quarter: int) -> np.ndarray:
    rgb = np.ascontiguousarray(rgb, np.uint8)
    h, w, _ = rgb.shape
    out = np.zeros((w, h, 3) if quarter in (1, 3) else (h, w, 3), np.uint8)
    _check(lib().oar_k_rotate_rgb(_p(rgb), w, h, quarter, _p(out)))
    return out


def k_bgr_planes_to_rgb(planes: np.ndarray, scale: float = 255.0) -> np.ndarray:
    planes = np.ascontiguousarray(planes, np.float32)
    _, h, w = planes.shape
    out = np.zeros((h, w, 3), np.uint8)
    _check(lib().oar_k_bgr_planes_to_rgb(_p(planes), h * w, C.c_float(scale), _p(out)))
    return out


def host_rotate_back_points(pts: np.ndarray, angle: float, rotated_w: int, rotated_h: int) -> np.ndarray:
    p = np.ascontiguousarray(pts, np.float32).copy()
    _check(lib().oar_host_rotate_back_points(_p(p), p.size // 2, C.c_float(angle), rotated_w, rotated_h))
    return p
