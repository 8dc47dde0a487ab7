//! Raw FFI bindings to `libOarMi355x.so` (the MI355X / gfx950 drop-in for the det+rec hot path of oar-ocr).
//!
//! GENERATED from `include/oar_mi355x.h` by `tools/gen_rust_sys.py` -- do not edit by hand; the header carries the
//! documentation of every item, including the reference interface (file:line) each entry point replaces.
//! `tests/test_rust_bindings_cpu.py` fails when this file and the header (or the symbols the built library exports)
//! drift apart.
#![allow(non_camel_case_types, non_snake_case, clippy::too_many_arguments)]
#![no_std]

use core::ffi::{c_char, c_int, c_void};

pub type oar_status = c_int;
pub const OAR_OK: oar_status = 0;
pub const OAR_INVALID_INPUT: oar_status = 1;
pub const OAR_MODEL_LOAD: oar_status = 2;
pub const OAR_UNSUPPORTED_OP: oar_status = 3;
pub const OAR_SHAPE_MISMATCH: oar_status = 4;
pub const OAR_DEVICE: oar_status = 5;
pub const OAR_OOM: oar_status = 6;
pub const OAR_INTERNAL: oar_status = 7;

pub type oar_precision = c_int;
pub const OAR_PRECISION_F32: oar_precision = 0;

pub type oar_dtype = c_int;
pub const OAR_DTYPE_F32: oar_dtype = 1;
pub const OAR_DTYPE_I64: oar_dtype = 7;

pub const OAR_RECT_NATIVE_SIZE: u32 = 0xFFFFFFFF;

#[repr(C)]
pub struct oar_engine {
    _private: [u8; 0],
    _marker: core::marker::PhantomData<(*mut u8, core::marker::PhantomPinned)>,
}

#[repr(C)]
pub struct oar_det {
    _private: [u8; 0],
    _marker: core::marker::PhantomData<(*mut u8, core::marker::PhantomPinned)>,
}

#[repr(C)]
pub struct oar_rec {
    _private: [u8; 0],
    _marker: core::marker::PhantomData<(*mut u8, core::marker::PhantomPinned)>,
}

#[repr(C)]
pub struct oar_ctc_dict {
    _private: [u8; 0],
    _marker: core::marker::PhantomData<(*mut u8, core::marker::PhantomPinned)>,
}

#[repr(C)]
pub struct oar_ocr {
    _private: [u8; 0],
    _marker: core::marker::PhantomData<(*mut u8, core::marker::PhantomPinned)>,
}

#[repr(C)]
pub struct oar_cls {
    _private: [u8; 0],
    _marker: core::marker::PhantomData<(*mut u8, core::marker::PhantomPinned)>,
}

#[repr(C)]
pub struct oar_rect {
    _private: [u8; 0],
    _marker: core::marker::PhantomData<(*mut u8, core::marker::PhantomPinned)>,
}

#[repr(C)]
pub struct oar_layout {
    _private: [u8; 0],
    _marker: core::marker::PhantomData<(*mut u8, core::marker::PhantomPinned)>,
}

pub type oar_output_view_fn = Option<unsafe extern "C" fn(user: *mut c_void, dims: *const i64, rank: i32, data: *const f32) -> i32>;

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct oar_engine_cfg {
    pub device_id: i32,
    pub use_hip_graph: i32,
    pub profile: i32,
    pub precision: i32,
    pub stream: *mut c_void,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct oar_tensor {
    pub rank: i32,
    pub dims: [i64; 8],
    pub data: *mut f32,
    pub name: [c_char; 64],
    pub dtype: i32,
    pub reserved: i32,
    pub data_i64: *mut i64,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct oar_input {
    pub name: *const c_char,
    pub data: *const f32,
    pub dims: *const i64,
    pub rank: i32,
    pub reserved: i32,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct oar_io_info {
    pub name: [c_char; 64],
    pub dtype: i32,
    pub rank: i32,
    pub dims: [i64; 8],
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct oar_det_cfg {
    pub device_id: i32,
    pub limit_side_len: u32,
    pub limit_type: i32,
    pub max_side_limit: u32,
    pub max_candidates: u32,
    pub use_hip_graph: i32,
    pub profile: i32,
    pub host_threads: i32,
    pub box_type: i32,
    pub score_mode: i32,
    pub use_dilation: i32,
    pub gpu_contours: i32,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct oar_det_result {
    pub n_images: u32,
    pub n_boxes: u32,
    pub box_offsets: *mut u32,
    pub points: *mut f32,
    pub scores: *mut f32,
    pub n_points: u32,
    pub point_offsets: *mut u32,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct oar_rec_cfg {
    pub device_id: i32,
    pub rec_image_shape: [u32; 3],
    pub max_img_w: u32,
    pub use_hip_graph: i32,
    pub profile: i32,
    pub reserved: i32,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct oar_rec_result {
    pub batch: u32,
    pub seq_len: u32,
    pub vocab: u32,
    pub tensor_width: u32,
    pub indices: *mut i64,
    pub probs: *mut f32,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct oar_text_result {
    pub n: u32,
    pub text_offsets: *mut u64,
    pub utf8: *mut c_char,
    pub scores: *mut f32,
    pub char_offsets: *mut u64,
    pub char_cols: *mut u32,
    pub char_positions: *mut f32,
    pub seq_len: *mut u32,
    pub kept: *mut u8,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct oar_ocr_cfg {
    pub det: oar_det_cfg,
    pub rec: oar_rec_cfg,
    pub det_thresh: f32,
    pub det_box_thresh: f32,
    pub det_unclip_ratio: f32,
    pub image_batch_size: u32,
    pub region_batch_size: u32,
    pub max_pooled_crops: u32,
    pub box_sort: i32,
    pub lanes: u32,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct oar_ocr_result {
    pub n_images: u32,
    pub n_regions: u32,
    pub region_offsets: *mut u32,
    pub points: *mut f32,
    pub det_scores: *mut f32,
    pub crop_wh: *mut u32,
    pub seq_len: *mut u32,
    pub max_wh_ratio: *mut f32,
    pub ctc_offsets: *mut u64,
    pub ctc_indices: *mut i64,
    pub ctc_probs: *mut f32,
    pub page_angle: *mut f32,
    pub page_rectified: *mut u8,
    pub line_angle: *mut f32,
    pub n_points: u32,
    pub point_offsets: *mut u32,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct oar_word_boxes {
    pub n_regions: u32,
    pub box_offsets: *mut u64,
    pub boxes: *mut f32,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct oar_packed_pages {
    pub n_images: u32,
    pub n_regions: u32,
    pub region_offsets: *mut u32,
    pub points: *mut f32,
    pub scores: *mut f32,
    pub text_offsets: *mut u64,
    pub utf8: *mut c_char,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct oar_cls_cfg {
    pub device_id: i32,
    pub input_h: u32,
    pub input_w: u32,
    pub resize_short: u32,
    pub topk: u32,
    pub batch: u32,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct oar_cls_result {
    pub n_images: u32,
    pub topk: u32,
    pub n_classes: u32,
    pub class_ids: *mut i32,
    pub scores: *mut f32,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct oar_rect_cfg {
    pub device_id: i32,
    pub target_h: u32,
    pub target_w: u32,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct oar_layout_cfg {
    pub device_id: i32,
    pub input_h: u32,
    pub input_w: u32,
    pub resize_filter: i32,
    pub color_bgr: i32,
    pub scale: f32,
    pub mean: [f32; 3],
    pub std: [f32; 3],
    pub num_classes: u32,
    pub model_type: i32,
    pub score_threshold: f32,
    pub nms_threshold: f32,
    pub max_detections: u32,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct oar_layout_result {
    pub n_images: u32,
    pub n_boxes: u32,
    pub box_offsets: *mut u32,
    pub boxes: *mut f32,
    pub classes: *mut i32,
    pub scores: *mut f32,
    pub feature_dim: u32,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct oar_ppdoc_cfg {
    pub score_threshold: f32,
    pub class_thresholds: *const f32,
    pub layout_nms: i32,
    pub image_class_id: i32,
    pub formula_class_id: i32,
    pub class_merge_modes: *const i32,
}

#[repr(C)]
#[derive(Debug, Clone, Copy)]
pub struct oar_prof_entry {
    pub name: [c_char; 48],
    pub launches: u64,
    pub total_ms: f64,
    pub alg_bytes: f64,
    pub alg_flops: f64,
}

#[link(name = "OarMi355x")]
unsafe extern "C" {
    pub fn oar_last_error(buf: *mut c_char, cap: usize) -> usize;
    pub fn oar_version(buf: *mut c_char, cap: usize) -> usize;
    pub fn oar_device_count() -> c_int;
    pub fn oar_engine_create(onnx: *const u8, onnx_len: usize, cfg: *const oar_engine_cfg, out: *mut *mut oar_engine) -> oar_status;
    pub fn oar_engine_destroy(e: *mut oar_engine);
    pub fn oar_engine_input_name(e: *const oar_engine, buf: *mut c_char, cap: usize) -> oar_status;
    pub fn oar_engine_run(e: *mut oar_engine, input: *const f32, dims: *const i64, rank: i32, outs: *mut oar_tensor, max_out: i32, n_out: *mut i32) -> oar_status;
    pub fn oar_tensor_free(t: *mut oar_tensor);
    pub fn oar_engine_run_named(e: *mut oar_engine, inputs: *const oar_input, n_in: i32, outs: *mut oar_tensor, max_out: i32, n_out: *mut i32) -> oar_status;
    pub fn oar_engine_run_first_f32(e: *mut oar_engine, inputs: *const oar_input, n_in: i32, view: oar_output_view_fn, user: *mut c_void) -> oar_status;
    pub fn oar_engine_io(e: *const oar_engine, inputs: *mut oar_io_info, max_in: i32, n_in: *mut i32, outputs: *mut oar_io_info, max_out: i32, n_out: *mut i32) -> oar_status;
    pub fn oar_engine_cost(e: *mut oar_engine, dims: *const i64, rank: i32, flops: *mut f64, bytes: *mut f64, n_kernels: *mut i32) -> oar_status;
    pub fn oar_engine_cache_stats(e: *mut oar_engine, cached_plans: *mut u64, evicted_plans: *mut u64) -> oar_status;
    pub fn oar_onnx_inspect(onnx: *const u8, onnx_len: usize, summary: *mut c_char, cap: usize) -> oar_status;
    pub fn oar_det_create(onnx: *const u8, onnx_len: usize, cfg: *const oar_det_cfg, out: *mut *mut oar_det) -> oar_status;
    pub fn oar_det_destroy(d: *mut oar_det);
    pub fn oar_det_run(d: *mut oar_det, rgb: *const *const u8, widths: *const u32, heights: *const u32, n_images: u32, thresh: f32, box_thresh: f32, unclip_ratio: f32, out: *mut oar_det_result) -> oar_status;
    pub fn oar_det_result_free(r: *mut oar_det_result);
    pub fn oar_db_postprocess(pred: *const f32, height: u32, width: u32, src_w: u32, src_h: u32, thresh: f32, box_thresh: f32, unclip_ratio: f32, max_candidates: u32, out: *mut oar_det_result) -> oar_status;
    pub fn oar_db_postprocess_ex(pred: *const f32, height: u32, width: u32, src_w: u32, src_h: u32, thresh: f32, box_thresh: f32, unclip_ratio: f32, max_candidates: u32, box_type: i32, score_mode: i32, use_dilation: i32, out: *mut oar_det_result) -> oar_status;
    pub fn oar_rec_create(onnx: *const u8, onnx_len: usize, cfg: *const oar_rec_cfg, out: *mut *mut oar_rec) -> oar_status;
    pub fn oar_rec_destroy(r: *mut oar_rec);
    pub fn oar_rec_run(r: *mut oar_rec, rgb: *const *const u8, widths: *const u32, heights: *const u32, n_crops: u32, out: *mut oar_rec_result) -> oar_status;
    pub fn oar_rec_result_free(r: *mut oar_rec_result);
    pub fn oar_ctc_dict_create(dict_utf8: *const c_char, len: usize, use_space_char: i32, out: *mut *mut oar_ctc_dict) -> oar_status;
    pub fn oar_ctc_dict_destroy(d: *mut oar_ctc_dict);
    pub fn oar_ctc_dict_classes(d: *const oar_ctc_dict) -> u32;
    pub fn oar_ctc_decode(dict: *const oar_ctc_dict, indices: *const i64, probs: *const f32, batch: u32, seq_len: u32, score_threshold: f32, out: *mut oar_text_result) -> oar_status;
    pub fn oar_text_result_free(r: *mut oar_text_result);
    pub fn oar_ocr_create(det_onnx: *const u8, det_len: usize, rec_onnx: *const u8, rec_len: usize, cfg: *const oar_ocr_cfg, out: *mut *mut oar_ocr) -> oar_status;
    pub fn oar_ocr_destroy(o: *mut oar_ocr);
    pub fn oar_ocr_predict(o: *mut oar_ocr, rgb: *const *const u8, widths: *const u32, heights: *const u32, n_images: u32, out: *mut oar_ocr_result) -> oar_status;
    pub fn oar_ocr_predict_device(o: *mut oar_ocr, d_rgb: *const *const u8, widths: *const u32, heights: *const u32, n_images: u32, out: *mut oar_ocr_result) -> oar_status;
    pub fn oar_ocr_result_free(r: *mut oar_ocr_result);
    pub fn oar_ocr_predict_async(o: *mut oar_ocr, rgb: *const *const u8, widths: *const u32, heights: *const u32, n_images: u32, device_pages: i32, ticket: *mut u64) -> oar_status;
    pub fn oar_ocr_wait(o: *mut oar_ocr, ticket: u64, out: *mut oar_ocr_result) -> oar_status;
    pub fn oar_ocr_decode(dict: *const oar_ctc_dict, res: *const oar_ocr_result, score_threshold: f32, out: *mut oar_text_result) -> oar_status;
    pub fn oar_ctc_word_boxes(line_pts_xy: *const f32, n_points: u32, text_utf8: *const c_char, text_len: usize, col_indices: *const u32, n_cols: u32, seq_len: u32, wh_ratio: f32, max_wh_ratio: f32, boxes: *mut f32, cap_boxes: u32, n_boxes: *mut u32) -> oar_status;
    pub fn oar_char_positions_to_word_boxes(line_pts_xy: *const f32, n_points: u32, char_positions: *const f32, n_positions: u32, char_count: u32, boxes: *mut f32, cap_boxes: u32, n_boxes: *mut u32) -> oar_status;
    pub fn oar_ocr_word_boxes(res: *const oar_ocr_result, txt: *const oar_text_result, out: *mut oar_word_boxes) -> oar_status;
    pub fn oar_word_boxes_free(w: *mut oar_word_boxes);
    pub fn oar_shard_range(n_items: u64, world_size: u32, rank: u32, begin: *mut u64, end: *mut u64) -> oar_status;
    pub fn oar_ocr_pack(res: *const oar_ocr_result, txt: *const oar_text_result, blob: *mut *mut u8, len: *mut usize) -> oar_status;
    pub fn oar_blob_free(blob: *mut u8);
    pub fn oar_packed_merge(blobs: *const *const u8, lens: *const usize, n_blobs: u32, out: *mut oar_packed_pages) -> oar_status;
    pub fn oar_packed_pages_free(p: *mut oar_packed_pages);
    pub fn oar_cls_create(onnx: *const u8, onnx_len: usize, cfg: *const oar_cls_cfg, out: *mut *mut oar_cls) -> oar_status;
    pub fn oar_cls_destroy(c: *mut oar_cls);
    pub fn oar_cls_run(c: *mut oar_cls, rgb: *const *const u8, widths: *const u32, heights: *const u32, n_images: u32, out: *mut oar_cls_result) -> oar_status;
    pub fn oar_cls_result_free(r: *mut oar_cls_result);
    pub fn oar_cls_preprocess(c: *mut oar_cls, rgb: *const *const u8, widths: *const u32, heights: *const u32, n_images: u32, out_nchw: *mut f32) -> oar_status;
    pub fn oar_rect_create(onnx: *const u8, onnx_len: usize, cfg: *const oar_rect_cfg, out: *mut *mut oar_rect) -> oar_status;
    pub fn oar_rect_destroy(r: *mut oar_rect);
    pub fn oar_rect_run(r: *mut oar_rect, rgb: *const u8, width: u32, height: u32, out_rgb: *mut u8) -> oar_status;
    pub fn oar_ocr_attach(o: *mut oar_ocr, doc_orientation: *mut oar_cls, rectifier: *mut oar_rect, line_orientation: *mut oar_cls) -> oar_status;
    pub fn oar_k_rotate_rgb(rgb: *const u8, w: u32, h: u32, quarter: i32, out: *mut u8) -> oar_status;
    pub fn oar_k_bgr_planes_to_rgb(planes: *const f32, plane: u64, scale: f32, out: *mut u8) -> oar_status;
    pub fn oar_host_rotate_back_points(pts: *mut f32, n_points: u32, angle: f32, rotated_w: u32, rotated_h: u32) -> oar_status;
    pub fn oar_layout_create(onnx: *const u8, onnx_len: usize, cfg: *const oar_layout_cfg, out: *mut *mut oar_layout) -> oar_status;
    pub fn oar_layout_destroy(l: *mut oar_layout);
    pub fn oar_layout_run(l: *mut oar_layout, rgb: *const *const u8, widths: *const u32, heights: *const u32, n_images: u32, out: *mut oar_layout_result) -> oar_status;
    pub fn oar_layout_result_free(r: *mut oar_layout_result);
    pub fn oar_layout_preprocess(l: *mut oar_layout, rgb: *const u8, width: u32, height: u32, out_chw: *mut f32) -> oar_status;
    pub fn oar_k_resize_filter(rgb: *const u8, w: u32, h: u32, nw: u32, nh: u32, filter: i32, out: *mut u8) -> oar_status;
    pub fn oar_k_layout_postprocess(pred: *const f32, n_images: u32, rows: u32, feat: u32, src_wh: *const f32, num_classes: u32, model_type: i32, score_threshold: f32, nms_threshold: f32, max_detections: u32, out: *mut oar_layout_result) -> oar_status;
    pub fn oar_layout_run_ppdoc(l: *mut oar_layout, rgb: *const *const u8, widths: *const u32, heights: *const u32, n_images: u32, cfg: *const oar_ppdoc_cfg, out: *mut oar_layout_result) -> oar_status;
    pub fn oar_k_ppdoc_postprocess(pred: *const f32, n_images: u32, rows: u32, feat: u32, src_wh: *const f32, num_classes: u32, cfg: *const oar_ppdoc_cfg, out: *mut oar_layout_result) -> oar_status;
    pub fn oar_host_nms_with_merge(boxes: *const f32, classes: *const i32, scores: *const f32, n: u32, mode_of_class: *const i32, num_classes: u32, nms_threshold: f32, max_detections: u32, out_boxes: *mut f32, out_classes: *mut i32, out_scores: *mut f32) -> i32;
    pub fn oar_dev_alloc(device_id: i32, bytes: usize, out: *mut *mut c_void) -> oar_status;
    pub fn oar_dev_upload(dst: *mut c_void, src: *const c_void, bytes: usize) -> oar_status;
    pub fn oar_dev_download(dst: *mut c_void, src: *const c_void, bytes: usize) -> oar_status;
    pub fn oar_dev_free(p: *mut c_void);
    pub fn oar_dev_synchronize(device_id: i32) -> oar_status;
    /// fixed-length arrays: src_channels: [i32; 3], alpha: [f32; 3], beta: [f32; 3]
    pub fn oar_k_normalize(rgb: *const u8, w: u32, h: u32, src_channels: *const i32, alpha: *const f32, beta: *const f32, hwc_layout: i32, out: *mut f32) -> oar_status;
    pub fn oar_k_rec_preprocess(rgb: *const *const u8, widths: *const u32, heights: *const u32, n: u32, img_h: u32, img_w: u32, max_img_w: u32, out_nchw: *mut f32, tensor_width: *mut u32) -> oar_status;
    pub fn oar_k_rec_preprocess_flip(rgb: *const *const u8, widths: *const u32, heights: *const u32, flips: *const u8, n: u32, img_h: u32, img_w: u32, max_img_w: u32, out_nchw: *mut f32, tensor_width: *mut u32) -> oar_status;
    pub fn oar_k_resize_triangle(rgb: *const u8, w: u32, h: u32, nw: u32, nh: u32, out: *mut u8) -> oar_status;
    pub fn oar_k_threshold(pred: *const f32, n: usize, thresh: f32, mask: *mut u8) -> oar_status;
    pub fn oar_k_dilate(mask: *const u8, height: u32, width: u32, out: *mut u8) -> oar_status;
    pub fn oar_k_poly_scores(pred: *const f32, height: u32, width: u32, pts_xy: *const f32, counts: *const u32, n_polys: u32, scores: *mut f32) -> oar_status;
    pub fn oar_k_contours(mask: *const u8, width: u32, height: u32, max_contours: u32, n_contours: *mut i32, offsets: *mut i64, pts_xy: *mut i32, types: *mut i32, cap_points: i64) -> oar_status;
    pub fn oar_k_ctc_argmax(probs: *const f32, rows: usize, vocab: usize, idx: *mut i64, prob: *mut f32) -> oar_status;
    pub fn oar_k_unclip(boxes: *const f32, n_boxes: u32, ratio: f32, counts: *mut i32, pts_xy: *mut f32, cap_points: u32) -> oar_status;
    pub fn oar_k_box_scores(pred: *const f32, height: u32, width: u32, boxes: *const f32, n_boxes: u32, scores: *mut f32) -> oar_status;
    /// fixed-length arrays: box_: [f32; 8]
    pub fn oar_k_rotate_crop(rgb: *const u8, w: u32, h: u32, box_: *const f32, out: *mut u8, cap: usize, out_w: *mut u32, out_h: *mut u32) -> oar_status;
    pub fn oar_image_decode(bytes: *const u8, len: usize, rgb: *mut *mut u8, width: *mut u32, height: *mut u32) -> oar_status;
    pub fn oar_image_free(rgb: *mut u8);
    pub fn oar_image_decode_device(bytes: *const u8, len: usize, device_id: i32, dev_rgb: *mut *mut c_void, width: *mut u32, height: *mut u32) -> oar_status;
    pub fn oar_host_candidates(mask: *const u8, width: u32, height: u32, max_candidates: u32, max_bands: i32, boxes8: *mut f32, cap: i32) -> i32;
    pub fn oar_host_contours(mask: *const u8, width: u32, height: u32, max_contours: u32, max_bands: i32, offsets: *mut i64, pts_xy: *mut i32, types: *mut i32, cap_points: i64) -> i32;
    pub fn oar_host_contours_bits(mask: *const u8, width: u32, height: u32, max_contours: u32, max_bands: i32, offsets: *mut i64, pts_xy: *mut i32, types: *mut i32, cap_points: i64) -> i32;
    /// fixed-length arrays: box8: [f32; 8]
    pub fn oar_host_unclip(box8: *const f32, ratio: f32, out_xy: *mut f32, cap_points: i32) -> i32;
    pub fn oar_host_approx_poly_dp(xy: *const f32, n_points: i32, epsilon: f32, out_xy: *mut f32, cap_points: i32) -> i32;
    pub fn oar_host_perimeter(xy: *const f32, n_points: i32) -> f32;
    pub fn oar_host_unclip_poly(xy: *const f32, n_points: i32, ratio: f32, out_xy: *mut f32, cap_points: i32) -> i32;
    pub fn oar_host_offset_ring(xy: *const i64, n_points: i32, radius: f64, out_xy: *mut i64, cap_points: i32) -> i32;
    pub fn oar_host_ring_outline(xy: *const i64, n_points: i32, negative: i32, out_xy: *mut i64, cap_points: i32) -> i32;
    pub fn oar_host_sort_poly_boxes(pts_xy: *const f32, offsets: *const u32, n: i32, order: *mut i32);
    /// fixed-length arrays: box8: [f32; 8]
    pub fn oar_host_mini_box(xy: *const f32, n_points: i32, box8: *mut f32, min_side: *mut f32) -> i32;
    pub fn oar_host_convex_hull(xy: *const f32, n_points: i32, out_xy: *mut f32, cap_points: i32) -> i32;
    pub fn oar_host_sort_quad_boxes(boxes8: *const f32, n: i32, order: *mut i32);
    pub fn oar_host_pool_selftest(threads: i32, jobs: i32) -> i32;
    /// fixed-length arrays: box8: [f32; 8], plan: [i32; 8], inv: [f32; 9]
    pub fn oar_host_plan_crop(img_w: u32, img_h: u32, box8: *const f32, plan: *mut i32, inv: *mut f32);
    pub fn oar_debug_inject_failure(site: *const c_char, count: i32) -> oar_status;
    pub fn oar_prof_reset();
    pub fn oar_prof_enable(on: i32);
    pub fn oar_prof_filter(class_name: *const c_char);
    pub fn oar_prof_sampling(stride: i32, phase: i32);
    pub fn oar_prof_snapshot(entries: *mut oar_prof_entry, cap: i32) -> i32;
}
