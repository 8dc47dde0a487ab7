//! Seal (curved) text detection on the MI355X: stands where `SealTextDetectionAdapter` stands
//! (oar-ocr-core/src/domain/adapters/seal_text_detection_adapter.rs:20-160).
//!
//! The reference builds the same `DBModel` as text detection with the seal preprocessing (`limit_side_len = 736`,
//! `LimitType::Min`, preprocessing.rs:44-62) and `BoxType::Poly`, `ScoreMode::Fast`, no dilation
//! (seal_text_detection_adapter.rs:131-139): `DBPostProcess::polygons_from_bitmap` (processors/db_bitmap.rs:16-82) then yields one
//! polygon of any size per text region.  Here that is `oar_det_create` with `box_type = 1` and one `oar_det_run` per batch; the
//! polygons come back through `oar_det_result::point_offsets`.

use crate::error::{Mi355xError, check};
use crate::ffi_util::model_bytes;
use crate::text_detection::{DetHandle, run_detection};
use crate::ffi_util::device_id_from_ort_config;
use oar_mi355x_sys as sys;
use oar_ocr_core::core::OCRError;
use oar_ocr_core::core::config::ConfigValidator;
use oar_ocr_core::core::inference::ModelSource;
use oar_ocr_core::core::config::OrtSessionConfig;
use oar_ocr_core::core::traits::adapter::{AdapterBuilder, AdapterInfo, ModelAdapter, OrtConfigurable};
use oar_ocr_core::core::traits::task::{Task, TaskType};
use oar_ocr_core::domain::tasks::{SealTextDetectionConfig, SealTextDetectionOutput, SealTextDetectionTask};
use std::ptr::NonNull;

/// `SealTextDetectionAdapter` with the DB model on the GPU.
#[derive(Debug)]
pub struct Mi355xSealTextDetectionAdapter {
    handle: DetHandle,
    info: AdapterInfo,
    config: SealTextDetectionConfig,
}

impl ModelAdapter for Mi355xSealTextDetectionAdapter {
    type Task = SealTextDetectionTask;

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
            "SealTextDetectionAdapter",
        )?;
        Ok(SealTextDetectionOutput { detections })
    }

    fn supports_batching(&self) -> bool {
        true
    }

    fn recommended_batch_size(&self) -> usize {
        8 // seal_text_detection_adapter.rs:100-102
    }
}

/// Builder with the surface of `SealTextDetectionAdapterBuilder` (seal_text_detection_adapter.rs:105-160).
#[derive(Debug, Clone)]
pub struct Mi355xSealTextDetectionAdapterBuilder {
    config: SealTextDetectionConfig,
    device_id: i32,
    host_threads: i32,
}

impl Default for Mi355xSealTextDetectionAdapterBuilder {
    fn default() -> Self {
        Self::new()
    }
}

impl Mi355xSealTextDetectionAdapterBuilder {
    pub fn new() -> Self {
        Self { config: SealTextDetectionConfig::default(), device_id: 0, host_threads: 0 }
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
}

impl AdapterBuilder for Mi355xSealTextDetectionAdapterBuilder {
    type Config = SealTextDetectionConfig;
    type Adapter = Mi355xSealTextDetectionAdapter;

    fn build(self, model_source: impl Into<ModelSource>) -> Result<Self::Adapter, OCRError> {
        self.config.validate().map_err(|err| OCRError::ConfigError { message: err.to_string() })?;
        let task_config = self.config;
        let cfg = sys::oar_det_cfg {
            device_id: self.device_id,
            limit_side_len: 736, // db_preprocess_for_text_type(Some("seal")) (preprocessing.rs:44-62)
            limit_type: 1,       // LimitType::Min
            max_side_limit: 4000,
            max_candidates: task_config.max_candidates as u32,
            use_hip_graph: 0,
            profile: 0,
            host_threads: self.host_threads,
            box_type: 1,     // BoxType::Poly  (seal_text_detection_adapter.rs:138)
            score_mode: 0,   // ScoreMode::Fast (:137)
            use_dilation: 0, // use_dilation: false (:136)
            gpu_contours: 0,
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
        let info = AdapterInfo::new(
            "seal_text_detection",
            TaskType::SealTextDetection,
            "Detects curved seal text with polygon bounding boxes (MI355X backend)",
        );
        Ok(Mi355xSealTextDetectionAdapter { handle, info, config: task_config })
    }

    fn with_config(mut self, config: Self::Config) -> Self {
        self.config = config;
        self
    }

    fn adapter_type(&self) -> &str {
        "seal_text_detection"
    }
}

/// `OrtConfigurable` (core/traits/adapter.rs:126-129): lets the reference's generic construction path
/// (`build_optional_adapter`, src/oarocr/builder_utils.rs:60-80; `OAROCRBuilder::build`, src/oarocr/ocr.rs:311,393) configure
/// this builder unchanged.  Only the device ordinal of the session configuration applies (see `device_id_from_ort_config`).
impl OrtConfigurable for Mi355xSealTextDetectionAdapterBuilder {
    fn with_ort_config(mut self, config: OrtSessionConfig) -> Self {
        if let Some(device_id) = device_id_from_ort_config(&config) {
            self.device_id = device_id;
        }
        self
    }
}
