//! Text detection on the MI355X: stands where `TextDetectionAdapter` stands
//! (oar-ocr-core/src/domain/adapters/text_detection_adapter.rs:19-87).
//!
//! `execute` = `DBModel::forward(images, score_threshold, box_threshold, unclip_ratio)` (models/detection/db.rs:281-335):
//! detection resize -> normalise -> DB network -> threshold -> contours -> mini boxes -> fast score -> unclip -> map back.
//! Here that is one `oar_det_run` call; boxes come back per image in contour discovery order, exactly as the reference
//! adapter returns them (the pipeline sorts them later, src/oarocr/ocr.rs:681).

use crate::error::{Mi355xError, check};
use crate::ffi_util::{ImageBatch, model_bytes, slice_or_empty};
use crate::ffi_util::device_id_from_ort_config;
use oar_mi355x_sys as sys;
use oar_ocr_core::core::OCRError;
use oar_ocr_core::core::config::ConfigValidator;
use oar_ocr_core::core::inference::ModelSource;
use oar_ocr_core::core::config::OrtSessionConfig;
use oar_ocr_core::core::traits::adapter::{AdapterBuilder, AdapterInfo, ModelAdapter, OrtConfigurable};
use oar_ocr_core::core::traits::task::{Task, TaskType};
use oar_ocr_core::domain::tasks::{Detection, TextDetectionConfig, TextDetectionOutput, TextDetectionTask};
use oar_ocr_core::processors::{BoundingBox, LimitType, Point};
use std::ptr::NonNull;

/// Owning handle of an `oar_det`.
#[derive(Debug)]
pub(crate) struct DetHandle(pub(crate) NonNull<sys::oar_det>);

// SAFETY: the library documents that a handle may be used from any thread and that calls on one handle serialise on an
// internal mutex (include/oar_mi355x.h, "Conventions"), which is what `ModelAdapter: Send + Sync` needs.
unsafe impl Send for DetHandle {}
unsafe impl Sync for DetHandle {}

impl Drop for DetHandle {
    fn drop(&mut self) {
        // SAFETY: the pointer came from oar_det_create and is destroyed exactly once.
        unsafe { sys::oar_det_destroy(self.0.as_ptr()) }
    }
}

/// Frees an `oar_det_result` on every exit path.
struct DetResultGuard(sys::oar_det_result);

impl Drop for DetResultGuard {
    fn drop(&mut self) {
        // SAFETY: the struct was filled by oar_det_run (or is all-NULL, which the library accepts).
        unsafe { sys::oar_det_result_free(&mut self.0) }
    }
}

/// `TextDetectionAdapter` with the DB model on the GPU.
#[derive(Debug)]
pub struct Mi355xTextDetectionAdapter {
    handle: DetHandle,
    info: AdapterInfo,
    config: TextDetectionConfig,
}

/// One `oar_det_run` call -> `Vec<Vec<Detection>>`, shared by the text and the seal text adapters.
///
/// Quad boxes come back as 4 points; with `box_type = 1` (BoxType::Poly) `point_offsets` delimits polygons of any size, exactly
/// the `BoundingBox::points` `polygons_from_bitmap` builds (processors/db_bitmap.rs:66-78).
pub(crate) fn run_detection(
    handle: *mut sys::oar_det,
    images: &[&image::RgbImage],
    score_threshold: f32,
    box_threshold: f32,
    unclip_ratio: f32,
    adapter_name: &'static str,
) -> Result<Vec<Vec<Detection>>, OCRError> {
    let batch = ImageBatch::new(images.iter().copied());
    if batch.is_empty() {
        return Ok(Vec::new());
    }
    let mut result = DetResultGuard(sys::oar_det_result {
        n_images: 0,
        n_boxes: 0,
        box_offsets: std::ptr::null_mut(),
        points: std::ptr::null_mut(),
        scores: std::ptr::null_mut(),
        n_points: 0,
        point_offsets: std::ptr::null_mut(),
    });
    // SAFETY: the three arrays hold batch.len() entries; every image pointer is valid for width * height * 3 bytes while
    // `images` is alive; `result` is a valid out-parameter.
    let status = unsafe {
        sys::oar_det_run(
            handle,
            batch.ptrs.as_ptr(),
            batch.widths.as_ptr(),
            batch.heights.as_ptr(),
            batch.len() as u32,
            score_threshold,
            box_threshold,
            unclip_ratio,
            &mut result.0,
        )
    };
    check(status).map_err(|e| {
        e.into_adapter_error(
            adapter_name,
            format!(
                "failed to detect text (score_threshold={score_threshold}, box_threshold={box_threshold}, unclip_ratio={unclip_ratio})"
            ),
        )
    })?;

    // CSR -> Vec<Vec<Detection>>: image i owns boxes [box_offsets[i], box_offsets[i + 1]); box b owns points
    // [point_offsets[b], point_offsets[b + 1]) when point_offsets is there, else the 4 points at b * 4
    let r = &result.0;
    let n_images = r.n_images as usize;
    let n_boxes = r.n_boxes as usize;
    let n_points = r.n_points as usize;
    // SAFETY: array lengths are the ones the header documents for oar_det_result.
    let (offsets, points, scores, point_offsets) = unsafe {
        (
            slice_or_empty(r.box_offsets, n_images + 1),
            slice_or_empty(r.points, n_points * 2),
            slice_or_empty(r.scores, n_boxes),
            if r.point_offsets.is_null() { None } else { Some(slice_or_empty(r.point_offsets, n_boxes + 1)) },
        )
    };
    let mut detections = Vec::with_capacity(n_images);
    for i in 0..n_images {
        let (lo, hi) = (offsets[i] as usize, offsets[i + 1] as usize);
        let mut per_image = Vec::with_capacity(hi - lo);
        for b in lo..hi {
            let (p0, p1) = match point_offsets {
                Some(po) => (po[b] as usize, po[b + 1] as usize),
                None => (b * 4, b * 4 + 4),
            };
            let pts = (p0..p1).map(|k| Point::new(points[k * 2], points[k * 2 + 1])).collect();
            per_image.push(Detection::new(BoundingBox::new(pts), scores[b]));
        }
        detections.push(per_image);
    }
    Ok(detections)
}

impl ModelAdapter for Mi355xTextDetectionAdapter {
    type Task = TextDetectionTask;

    fn info(&self) -> AdapterInfo {
        self.info.clone()
    }

    fn execute(
        &self,
        input: <Self::Task as Task>::Input,
        config: Option<&<Self::Task as Task>::Config>,
    ) -> Result<<Self::Task as Task>::Output, OCRError> {
        let effective_config = config.unwrap_or(&self.config);
        let images: Vec<&image::RgbImage> = input.images.iter().map(AsRef::as_ref).collect();
        let detections = run_detection(
            self.handle.0.as_ptr(),
            &images,
            effective_config.score_threshold,
            effective_config.box_threshold,
            effective_config.unclip_ratio,
            "TextDetectionAdapter",
        )?;
        Ok(TextDetectionOutput { detections })
    }

    fn supports_batching(&self) -> bool {
        true
    }

    fn recommended_batch_size(&self) -> usize {
        8 // text_detection_adapter.rs:85-87; the library splits a larger batch into sub-batches of 8 itself
    }
}

/// Builder with the surface of `TextDetectionAdapterBuilder` (text_detection_adapter.rs:89-187).
#[derive(Debug, Clone)]
pub struct Mi355xTextDetectionAdapterBuilder {
    config: TextDetectionConfig,
    text_type: Option<String>,
    model_name_override: Option<String>,
    device_id: i32,
    host_threads: i32,
    gpu_contours: bool,
}

impl Default for Mi355xTextDetectionAdapterBuilder {
    fn default() -> Self {
        Self::new()
    }
}

impl Mi355xTextDetectionAdapterBuilder {
    pub fn new() -> Self {
        Self {
            config: TextDetectionConfig::default(),
            text_type: None,
            model_name_override: None,
            device_id: 0,
            host_threads: 0,
            gpu_contours: false,
        }
    }

    /// `"seal"` selects the seal-text preprocessing and polygon boxes in the reference.
    pub fn text_type(mut self, text_type: impl Into<String>) -> Self {
        self.text_type = Some(text_type.into());
        self
    }

    pub fn model_name(mut self, model_name: impl Into<String>) -> Self {
        self.model_name_override = Some(model_name.into());
        self
    }

    /// HIP device ordinal (one adapter per GPU; one process per GPU when scaling out).
    pub fn device_id(mut self, device_id: i32) -> Self {
        self.device_id = device_id;
        self
    }

    /// Worker threads of the host-side contour / geometry stage (0 = all hardware threads).
    pub fn host_threads(mut self, host_threads: i32) -> Self {
        self.host_threads = host_threads;
        self
    }

    /// Follow the mask borders on the GPU instead of the host thread pool (`oar_det_cfg.gpu_contours`): identical boxes;
    /// worthwhile when many GPU ranks share the host's cores.
    pub fn gpu_contours(mut self, enable: bool) -> Self {
        self.gpu_contours = enable;
        self
    }

    fn base_adapter_info() -> AdapterInfo {
        AdapterInfo::new(
            "text_detection",
            TaskType::TextDetection,
            "Detects text regions in images with bounding boxes (MI355X backend)",
        )
    }
}

impl AdapterBuilder for Mi355xTextDetectionAdapterBuilder {
    type Config = TextDetectionConfig;
    type Adapter = Mi355xTextDetectionAdapter;

    fn build(self, model_source: impl Into<ModelSource>) -> Result<Self::Adapter, OCRError> {
        self.config.validate().map_err(|err| OCRError::ConfigError { message: err.to_string() })?;
        let task_config = self.config;

        let is_seal_text = self.text_type.as_ref().map(|t| t.to_lowercase() == "seal").unwrap_or(false);
        // db_preprocess_for_text_type (domain/adapters/preprocessing.rs:44-62), then the task-config overrides
        // (text_detection_adapter.rs:131-140)
        let (mut limit_side_len, mut limit_type, mut max_side_limit) =
            if is_seal_text { (736u32, LimitType::Min, 4000u32) } else { (960u32, LimitType::Max, 4000u32) };
        if let Some(limit) = task_config.limit_side_len {
            limit_side_len = limit;
        }
        if let Some(lt) = task_config.limit_type.clone() {
            limit_type = lt;
        }
        if let Some(max_limit) = task_config.max_side_len {
            max_side_limit = max_limit;
        }

        let cfg = sys::oar_det_cfg {
            device_id: self.device_id,
            limit_side_len,
            limit_type: match limit_type {
                LimitType::Max => 0,
                LimitType::Min => 1,
                LimitType::ResizeLong => 2,
            },
            max_side_limit,
            max_candidates: task_config.max_candidates as u32,
            use_hip_graph: 0,
            profile: 0,
            host_threads: self.host_threads,
            // BoxType::Poly for seal text (text_detection_adapter.rs:144-148): polygons come back through point_offsets
            box_type: if is_seal_text { 1 } else { 0 },
            score_mode: 0,   // ScoreMode::Fast  (text_detection_adapter.rs:155)
            use_dilation: 0, // use_dilation: false (text_detection_adapter.rs:154)
            gpu_contours: i32::from(self.gpu_contours),
        };

        let source: ModelSource = model_source.into();
        let (bytes, shown) = model_bytes(&source)?;
        let mut raw: *mut sys::oar_det = std::ptr::null_mut();
        // SAFETY: bytes is valid for bytes.len(); cfg and raw are valid for the duration of the call.
        let status = unsafe { sys::oar_det_create(bytes.as_ptr(), bytes.len(), &cfg, &mut raw) };
        check(status).map_err(|e: Mi355xError| e.into_model_load(&shown))?;
        let handle = DetHandle(NonNull::new(raw).ok_or_else(|| OCRError::ConfigError {
            message: "oar_det_create returned OAR_OK with a null handle".to_string(),
        })?);

        let mut info = Self::base_adapter_info();
        if let Some(model_name) = self.model_name_override {
            info.model_name = model_name;
        }
        Ok(Mi355xTextDetectionAdapter { handle, info, config: task_config })
    }

    fn with_config(mut self, config: Self::Config) -> Self {
        self.config = config;
        self
    }

    fn adapter_type(&self) -> &str {
        "text_detection"
    }
}

/// `OrtConfigurable` (core/traits/adapter.rs:126-129): lets the reference's generic construction path
/// (`build_optional_adapter`, src/oarocr/builder_utils.rs:60-80; `OAROCRBuilder::build`, src/oarocr/ocr.rs:311,393) configure
/// this builder unchanged.  Only the device ordinal of the session configuration applies (see `device_id_from_ort_config`).
impl OrtConfigurable for Mi355xTextDetectionAdapterBuilder {
    fn with_ort_config(mut self, config: OrtSessionConfig) -> Self {
        if let Some(device_id) = device_id_from_ort_config(&config) {
            self.device_id = device_id;
        }
        self
    }
}
