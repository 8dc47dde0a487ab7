//! The whole det -> sort -> crop -> rec path behind one call: stands where `OAROCR::predict` stands
//! (src/oarocr/ocr.rs:518-659).
//!
//! Going through the four adapters reproduces the reference pipeline stage by stage, with every stage's result
//! crossing PCIe.  `Mi355xOcr` instead keeps everything between the u8 pages and the per-region (box, CTC indices)
//! in HBM: detect -> `sort_quad_boxes` (`sort_poly_boxes` for seal text) -> rotate-crop -> width-ratio pooled recognition batches -> CTC argmax
//! (`oar_ocr_predict`), then the collapse / text assembly on the host (`oar_ocr_decode`).  Optional stages attach
//! exactly like the builder methods of the reference (`with_document_image_orientation_classification`,
//! `with_document_image_rectification`, `with_text_line_orientation_classification`).

use crate::error::{Mi355xError, check};
use crate::ffi_util::{ImageBatch, device_id_from_ort_config, model_bytes, slice_or_empty};
use crate::orientation::{ClsHandle, Mi355xDocumentOrientationAdapter, Mi355xTextLineOrientationAdapter};
use crate::rectification::RectHandle;
use crate::text_recognition::{DictHandle, TextResultGuard};
use image::RgbImage;
use oar_mi355x_sys as sys;
use oar_ocr_core::core::OCRError;
use oar_ocr_core::core::config::OrtSessionConfig;
use oar_ocr_core::core::inference::ModelSource;
use oar_ocr_core::domain::TextRegion;
use oar_ocr_core::processors::{BoundingBox, Point};
use std::ptr::NonNull;
use std::sync::Arc;

#[derive(Debug)]
struct OcrHandle(NonNull<sys::oar_ocr>);
// SAFETY: handles are usable from any thread; calls on one handle serialise inside the library.
unsafe impl Send for OcrHandle {}
unsafe impl Sync for OcrHandle {}
impl Drop for OcrHandle {
    fn drop(&mut self) {
        // SAFETY: created by oar_ocr_create, destroyed once.
        unsafe { sys::oar_ocr_destroy(self.0.as_ptr()) }
    }
}

struct OcrResultGuard(sys::oar_ocr_result);
impl Drop for OcrResultGuard {
    fn drop(&mut self) {
        // SAFETY: filled by oar_ocr_predict or all-NULL.
        unsafe { sys::oar_ocr_result_free(&mut self.0) }
    }
}

/// One detected + recognised region; `region` is the reference's `TextRegion` (domain/text_region.rs:10-41) filled the
/// way `recognize_global` fills it (ocr.rs:880-890), the rest is what the reference keeps in its crop metadata.
#[derive(Debug, Clone)]
pub struct Mi355xOcrRegion {
    pub region: TextRegion,
    /// detector confidence of the box (not part of `TextRegion`)
    pub det_score: f32,
    /// (width, height) of the rectified crop the recognizer saw
    pub crop_size: (u32, u32),
    /// CTC time steps of the recognition batch and the time step of every kept character (for word boxes)
    pub sequence_length: usize,
    pub char_col_indices: Vec<usize>,
    /// `chunk_max_wh_ratio` of the batch the crop was recognised in (ocr.rs:828-831)
    pub max_wh_ratio: f32,
}

/// One page of `predict`'s answer: the fields of `OAROCRResult` (src/oarocr/result.rs:34-49) that the hot path
/// produces.  `rectified` says whether `rectified_img` would be `Some` (boxes then live in rectified-page space).
#[derive(Debug, Clone)]
pub struct Mi355xOcrPage {
    pub index: usize,
    pub text_regions: Vec<Mi355xOcrRegion>,
    pub orientation_angle: Option<f32>,
    pub rectified: bool,
}

struct WordBoxesGuard(sys::oar_word_boxes);
impl Drop for WordBoxesGuard {
    fn drop(&mut self) {
        // SAFETY: filled by oar_ocr_word_boxes or all-NULL.
        unsafe { sys::oar_word_boxes_free(&mut self.0) }
    }
}

/// Builder with the surface of `OAROCRBuilder` (src/oarocr/ocr.rs:105-417) for the stages of the hot path.
#[derive(Clone)]
pub struct Mi355xOcrBuilder {
    det_model: ModelSource,
    rec_model: ModelSource,
    character_dict: Vec<String>,
    doc_orientation_model: Option<ModelSource>,
    rectification_model: Option<ModelSource>,
    line_orientation_model: Option<ModelSource>,
    det_thresh: f32,
    det_box_thresh: f32,
    det_unclip_ratio: f32,
    rec_score_thresh: f32,
    image_batch_size: u32,
    region_batch_size: u32,
    limit_side_len: u32,
    device_id: i32,
    text_type: Option<String>,
    explicit_det_thresholds: bool,
    return_word_box: bool,
}

impl Mi355xOcrBuilder {
    /// `OAROCRBuilder::new(det_model, rec_model, char_dict_path)` (ocr.rs:105); the dictionary is given as its lines
    /// (`char_dict.lines()`, ocr.rs:386).
    pub fn new(det_model: impl Into<ModelSource>, rec_model: impl Into<ModelSource>, character_dict: Vec<String>) -> Self {
        Self {
            det_model: det_model.into(),
            rec_model: rec_model.into(),
            character_dict,
            doc_orientation_model: None,
            rectification_model: None,
            line_orientation_model: None,
            det_thresh: 0.3,       // builder defaults, ocr.rs:319-366
            det_box_thresh: 0.6,
            det_unclip_ratio: 2.0,
            rec_score_thresh: 0.0,
            image_batch_size: 0,   // 0 => adapter recommended size
            region_batch_size: 0,
            limit_side_len: 0,
            device_id: 0,
            text_type: None,
            explicit_det_thresholds: false,
            return_word_box: false,
        }
    }

    /// `OAROCRBuilder::text_type` (ocr.rs:218-229): `"seal"` = curved text -- 736 / min preprocessing, polygon boxes
    /// (`BoxType::Poly`), `sort_poly_boxes`, bounding-rectangle crops, and 0.2 / 0.6 / 0.5 as detection defaults.
    pub fn text_type(mut self, text_type: impl Into<String>) -> Self {
        self.text_type = Some(text_type.into());
        self
    }

    pub fn with_document_image_orientation_classification(mut self, model: impl Into<ModelSource>) -> Self {
        self.doc_orientation_model = Some(model.into());
        self
    }

    pub fn with_document_image_rectification(mut self, model: impl Into<ModelSource>) -> Self {
        self.rectification_model = Some(model.into());
        self
    }

    pub fn with_text_line_orientation_classification(mut self, model: impl Into<ModelSource>) -> Self {
        self.line_orientation_model = Some(model.into());
        self
    }

    pub fn text_det_threshold(mut self, v: f32) -> Self {
        self.explicit_det_thresholds = true;
        self.det_thresh = v;
        self
    }

    pub fn text_det_box_threshold(mut self, v: f32) -> Self {
        self.explicit_det_thresholds = true;
        self.det_box_thresh = v;
        self
    }

    pub fn text_det_unclip_ratio(mut self, v: f32) -> Self {
        self.explicit_det_thresholds = true;
        self.det_unclip_ratio = v;
        self
    }

    pub fn text_det_limit_side_len(mut self, v: u32) -> Self {
        self.limit_side_len = v;
        self
    }

    pub fn text_rec_score_threshold(mut self, v: f32) -> Self {
        self.rec_score_thresh = v;
        self
    }

    pub fn image_batch_size(mut self, v: usize) -> Self {
        self.image_batch_size = v as u32;
        self
    }

    pub fn region_batch_size(mut self, v: usize) -> Self {
        self.region_batch_size = v as u32;
        self
    }

    pub fn device_id(mut self, device_id: i32) -> Self {
        self.device_id = device_id;
        self
    }

    /// `OAROCRBuilder::ort_session` (src/oarocr/ocr.rs:135-141): the session configuration of every model of the pipeline.
    /// Here it selects the device (see `device_id_from_ort_config`); the ONNX Runtime knobs do not apply.
    pub fn ort_session(mut self, config: OrtSessionConfig) -> Self {
        if let Some(device_id) = device_id_from_ort_config(&config) {
            self.device_id = device_id;
        }
        self
    }

    /// `OAROCRBuilder::return_word_box` (src/oarocr/ocr.rs): fill `TextRegion::word_boxes` (row a21, `oar_ocr_word_boxes`).
    pub fn return_word_box(mut self, enable: bool) -> Self {
        self.return_word_box = enable;
        self
    }

    pub fn build(self) -> Result<Mi355xOcr, OCRError> {
        let dict = DictHandle::new(Some(&self.character_dict))?;
        // the adapter decides by the lower-cased text type (text_detection_adapter.rs:131-136); the default thresholds by the exact
        // string (ocr.rs:322-352)
        let is_seal_text = self.text_type.as_ref().map(|t| t.to_lowercase() == "seal").unwrap_or(false);
        let (det_thresh, det_box_thresh, det_unclip_ratio) = if self.explicit_det_thresholds {
            (self.det_thresh, self.det_box_thresh, self.det_unclip_ratio)
        } else {
            match self.text_type.as_deref().unwrap_or("general") {
                "table" => (0.3, 0.4, 2.0),
                "seal" => (0.2, 0.6, 0.5),
                _ => (0.3, 0.6, 2.0),
            }
        };
        let det_cfg = sys::oar_det_cfg {
            device_id: self.device_id,
            limit_side_len: if self.limit_side_len == 0 && is_seal_text { 736 } else { self.limit_side_len }, // 0 => 960
            limit_type: if is_seal_text { 1 } else { 0 },
            max_side_limit: 0,
            max_candidates: 0,
            use_hip_graph: 0,
            profile: 0,
            host_threads: 0,
            box_type: if is_seal_text { 1 } else { 0 },
            score_mode: 0,
            use_dilation: 0,
            gpu_contours: 0,
        };
        let rec_cfg = sys::oar_rec_cfg {
            device_id: self.device_id,
            rec_image_shape: [0, 0, 0], // => [3, 48, 320]
            max_img_w: 0,
            use_hip_graph: 0,
            profile: 0,
            reserved: 0,
        };
        let cfg = sys::oar_ocr_cfg {
            det: det_cfg,
            rec: rec_cfg,
            det_thresh,
            det_box_thresh,
            det_unclip_ratio,
            image_batch_size: self.image_batch_size,
            region_batch_size: self.region_batch_size,
            max_pooled_crops: 0,
            // sort_detection_boxes keys on the text type, not on the detector's box type (ocr.rs:699-716)
            box_sort: if is_seal_text { 2 } else { 1 },
            lanes: 0,
        };
        let (det_bytes, det_shown) = model_bytes(&self.det_model)?;
        let (rec_bytes, _rec_shown) = model_bytes(&self.rec_model)?;
        let mut raw: *mut sys::oar_ocr = std::ptr::null_mut();
        // SAFETY: both byte ranges are valid; cfg / raw valid for the call.
        let status = unsafe {
            sys::oar_ocr_create(det_bytes.as_ptr(), det_bytes.len(), rec_bytes.as_ptr(), rec_bytes.len(), &cfg, &mut raw)
        };
        check(status).map_err(|e: Mi355xError| e.into_model_load(&det_shown))?;
        let handle = OcrHandle(NonNull::new(raw).ok_or_else(|| OCRError::ConfigError {
            message: "oar_ocr_create returned OAR_OK with a null handle".to_string(),
        })?);

        // optional stages: handles are borrowed by the pipeline and must outlive it -> owned by Mi355xOcr, dropped after it
        let doc_orientation = match self.doc_orientation_model {
            Some(m) => Some(ClsHandle::create(
                m,
                Mi355xDocumentOrientationAdapter::DEFAULT_INPUT_SHAPE,
                Some(256),
                1,
                self.device_id,
            )?),
            None => None,
        };
        let line_orientation = match self.line_orientation_model {
            Some(m) => Some(ClsHandle::create(
                m,
                Mi355xTextLineOrientationAdapter::DEFAULT_INPUT_SHAPE,
                None,
                1,
                self.device_id,
            )?),
            None => None,
        };
        let rectifier = match self.rectification_model {
            Some(m) => {
                let (bytes, shown) = model_bytes(&m)?;
                let rcfg = sys::oar_rect_cfg { device_id: self.device_id, target_h: 0, target_w: 0 };
                let mut r: *mut sys::oar_rect = std::ptr::null_mut();
                // SAFETY: bytes valid for bytes.len(); rcfg / r valid for the call.
                let status = unsafe { sys::oar_rect_create(bytes.as_ptr(), bytes.len(), &rcfg, &mut r) };
                check(status).map_err(|e: Mi355xError| e.into_model_load(&shown))?;
                Some(RectHandle(NonNull::new(r).ok_or_else(|| OCRError::ConfigError {
                    message: "oar_rect_create returned a null handle".to_string(),
                })?))
            }
            None => None,
        };
        let null_cls = std::ptr::null_mut::<sys::oar_cls>();
        // SAFETY: every non-null handle is alive and owned by the struct built below.
        let status = unsafe {
            sys::oar_ocr_attach(
                handle.0.as_ptr(),
                doc_orientation.as_ref().map_or(null_cls, |h| h.0.as_ptr()),
                rectifier.as_ref().map_or(std::ptr::null_mut(), |h| h.0.as_ptr()),
                line_orientation.as_ref().map_or(null_cls, |h| h.0.as_ptr()),
            )
        };
        check(status).map_err(|e| OCRError::ConfigError { message: e.to_string() })?;

        Ok(Mi355xOcr {
            handle,
            dict,
            rec_score_thresh: self.rec_score_thresh,
            return_word_box: self.return_word_box,
            _doc_orientation: doc_orientation,
            _rectifier: rectifier,
            _line_orientation: line_orientation,
        })
    }
}

/// `OAROCR` for the det+rec hot path.  Field order matters: `handle` is declared (hence dropped) before the stage
/// handles it borrows.
#[derive(Debug)]
pub struct Mi355xOcr {
    handle: OcrHandle,
    dict: DictHandle,
    rec_score_thresh: f32,
    return_word_box: bool,
    _doc_orientation: Option<ClsHandle>,
    _rectifier: Option<RectHandle>,
    _line_orientation: Option<ClsHandle>,
}

impl Mi355xOcr {
    /// One rank's share of a multi-process job (`crate::shard`): `predict` + decode on `images`, the final results (boxes, text
    /// scores, texts) as ONE blob in `oar_ocr_pack`'s wire format -- what the host ships to rank 0 for `shard::merge_packed`.
    pub fn predict_packed_blob(&self, images: &[Arc<RgbImage>]) -> Result<Vec<u8>, OCRError> {
        let batch = ImageBatch::new(images.iter().map(AsRef::as_ref));
        // SAFETY: an all-zero oar_ocr_result (NULL arrays, zero counts) is a valid "empty" value for the free function.
        let mut res = OcrResultGuard(unsafe { std::mem::zeroed() });
        if !batch.is_empty() {
            // SAFETY: three arrays of batch.len() entries; page buffers outlive the call.
            let status = unsafe {
                sys::oar_ocr_predict(self.handle.0.as_ptr(), batch.ptrs.as_ptr(), batch.widths.as_ptr(), batch.heights.as_ptr(), batch.len() as u32, &mut res.0)
            };
            check(status).map_err(|e| e.into_adapter_error("OAROCR", format!("predict (pages={})", batch.len())))?;
        }
        let mut texts = TextResultGuard::empty();
        // SAFETY: res is empty or was filled by oar_ocr_predict; texts is a valid out-parameter.
        let status = unsafe { sys::oar_ocr_decode(self.dict.0.as_ptr(), &res.0, self.rec_score_thresh, &mut texts.0) };
        check(status).map_err(|e| e.into_adapter_error("OAROCR", "decode".to_string()))?;
        let (mut blob, mut len) = (std::ptr::null_mut::<u8>(), 0usize);
        // SAFETY: res / texts describe the same regions; blob / len are valid out-parameters.
        let status = unsafe { sys::oar_ocr_pack(&res.0, &texts.0, &mut blob, &mut len) };
        check(status).map_err(|e| e.into_adapter_error("OAROCR", "pack".to_string()))?;
        // SAFETY: oar_ocr_pack returned len bytes at blob; copied out before the library's buffer is released.
        let out = unsafe { crate::ffi_util::slice_or_empty(blob as *const u8, len) }.to_vec();
        // SAFETY: blob came from oar_ocr_pack (or is NULL).
        unsafe { sys::oar_blob_free(blob) };
        Ok(out)
    }

    /// `OAROCR::predict(images)` (ocr.rs:518-659): one entry per input page, regions in reading order.
    pub fn predict(&self, images: &[Arc<RgbImage>]) -> Result<Vec<Mi355xOcrPage>, OCRError> {
        if images.is_empty() {
            return Ok(Vec::new());
        }
        let batch = ImageBatch::new(images.iter().map(AsRef::as_ref));
        // SAFETY: an all-zero oar_ocr_result (NULL arrays, zero counts) is a valid "empty" value for the free function.
        let mut res = OcrResultGuard(unsafe { std::mem::zeroed() });
        // SAFETY: three arrays of batch.len() entries; page buffers outlive the call.
        let status = unsafe {
            sys::oar_ocr_predict(
                self.handle.0.as_ptr(),
                batch.ptrs.as_ptr(),
                batch.widths.as_ptr(),
                batch.heights.as_ptr(),
                batch.len() as u32,
                &mut res.0,
            )
        };
        check(status).map_err(|e| e.into_adapter_error("OAROCR", format!("predict (pages={})", batch.len())))?;

        let mut texts = TextResultGuard::empty();
        // SAFETY: res was filled by oar_ocr_predict; texts is a valid out-parameter.
        let status = unsafe { sys::oar_ocr_decode(self.dict.0.as_ptr(), &res.0, self.rec_score_thresh, &mut texts.0) };
        check(status).map_err(|e| e.into_adapter_error("OAROCR", "decode".to_string()))?;
        let decoded = texts.to_output(true);
        // return_word_box (ocr.rs:860-877): OAROCR::ctc_word_boxes of every region, inside the library (row a21)
        let mut word_boxes = WordBoxesGuard(sys::oar_word_boxes { n_regions: 0, box_offsets: std::ptr::null_mut(), boxes: std::ptr::null_mut() });
        if self.return_word_box {
            // SAFETY: res / texts were filled by the two calls above and describe the same regions; word_boxes is a valid out-parameter.
            let status = unsafe { sys::oar_ocr_word_boxes(&res.0, &texts.0, &mut word_boxes.0) };
            check(status).map_err(|e| e.into_adapter_error("OAROCR", "word boxes".to_string()))?;
        }

        let r = &res.0;
        let (n, nr) = (r.n_images as usize, r.n_regions as usize);
        // SAFETY: lengths as documented for oar_ocr_result.
        let n_points = r.n_points as usize;
        let (offsets, points, det_scores, crop_wh, max_wh, page_angle, page_rect, line_angle, point_offsets) = unsafe {
            (
                slice_or_empty(r.region_offsets, n + 1),
                slice_or_empty(r.points, n_points * 2),
                slice_or_empty(r.det_scores, nr),
                slice_or_empty(r.crop_wh, nr * 2),
                slice_or_empty(r.max_wh_ratio, nr),
                slice_or_empty(r.page_angle, n),
                slice_or_empty(r.page_rectified, n),
                slice_or_empty(r.line_angle, nr),
                // seal text: region k owns points [point_offsets[k], point_offsets[k + 1]); quads: the 4 points at k * 4
                if r.point_offsets.is_null() { None } else { Some(slice_or_empty(r.point_offsets, nr + 1)) },
            )
        };
        // SAFETY: lengths as documented for oar_word_boxes (both arrays NULL when word boxes were not requested).
        let (wb_offsets, wb_boxes) = unsafe {
            let offs = slice_or_empty(word_boxes.0.box_offsets, if self.return_word_box { nr + 1 } else { 0 });
            let total = offs.last().copied().unwrap_or(0) as usize;
            (offs, slice_or_empty(word_boxes.0.boxes, total * 8))
        };
        let mut pages = Vec::with_capacity(n);
        for i in 0..n {
            let (lo, hi) = (offsets[i] as usize, offsets[i + 1] as usize);
            let mut regions = Vec::with_capacity(hi - lo);
            for k in lo..hi {
                let (p0, p1) = match point_offsets {
                    Some(po) => (po[k] as usize, po[k + 1] as usize),
                    None => (k * 4, k * 4 + 4),
                };
                let bbox = BoundingBox::new((p0..p1).map(|q| Point::new(points[q * 2], points[q * 2 + 1])).collect());
                let region = TextRegion {
                    bounding_box: bbox.clone(),
                    dt_poly: Some(bbox.clone()),
                    rec_poly: Some(bbox),
                    text: Some(Arc::from(decoded.texts[k].as_str())),
                    confidence: Some(decoded.scores[k]),
                    orientation_angle: if line_angle[k] >= 0.0 { Some(line_angle[k]) } else { None },
                    // Some(..) exactly when the reference's branch at ocr.rs:860 is taken: word boxes requested and the region has characters
                    word_boxes: if self.return_word_box && wb_offsets[k + 1] > wb_offsets[k] {
                        Some(
                            (wb_offsets[k] as usize..wb_offsets[k + 1] as usize)
                                .map(|b| BoundingBox::new((0..4).map(|q| Point::new(wb_boxes[b * 8 + q * 2], wb_boxes[b * 8 + q * 2 + 1])).collect()))
                                .collect(),
                        )
                    } else {
                        None
                    },
                    label: None,
                };
                regions.push(Mi355xOcrRegion {
                    region,
                    det_score: det_scores[k],
                    crop_size: (crop_wh[k * 2], crop_wh[k * 2 + 1]),
                    sequence_length: decoded.sequence_lengths[k],
                    char_col_indices: decoded.char_col_indices[k].clone(),
                    max_wh_ratio: max_wh[k],
                });
            }
            pages.push(Mi355xOcrPage {
                index: i,
                text_regions: regions,
                orientation_angle: if page_angle[i] >= 0.0 { Some(page_angle[i]) } else { None },
                rectified: page_rect[i] != 0,
            });
        }
        Ok(pages)
    }
}
