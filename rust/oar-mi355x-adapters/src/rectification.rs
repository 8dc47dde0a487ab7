//! UVDoc rectification on the MI355X: stands where `UVDocRectifierAdapter` stands
//! (oar-ocr-core/src/domain/adapters/document_rectification_adapter.rs:17-64).
//!
//! `execute` = `UVDocModel::forward_refs` (models/rectification/uvdoc.rs:82-207) per image: Triangle resize to the
//! configured input shape, BGR v/255, the network (with its GridSample un-warp), u8 conversion, Triangle resize back to
//! the page size -- one `oar_rect_run` call per page, the page never leaves HBM in between.

use crate::error::{Mi355xError, check};
use crate::ffi_util::model_bytes;
use image::RgbImage;
use crate::ffi_util::device_id_from_ort_config;
use oar_mi355x_sys as sys;
use oar_ocr_core::core::OCRError;
use oar_ocr_core::core::inference::ModelSource;
use oar_ocr_core::core::config::OrtSessionConfig;
use oar_ocr_core::core::traits::adapter::{AdapterBuilder, AdapterInfo, ModelAdapter, OrtConfigurable};
use oar_ocr_core::core::traits::task::{Task, TaskType};
use oar_ocr_core::domain::tasks::{
    DocumentRectificationConfig, DocumentRectificationOutput, DocumentRectificationTask,
};
use std::ptr::NonNull;

#[derive(Debug)]
pub(crate) struct RectHandle(pub(crate) NonNull<sys::oar_rect>);
// SAFETY: handles are usable from any thread; calls on one handle serialise inside the library.
unsafe impl Send for RectHandle {}
unsafe impl Sync for RectHandle {}
impl Drop for RectHandle {
    fn drop(&mut self) {
        // SAFETY: created by oar_rect_create, destroyed once.
        unsafe { sys::oar_rect_destroy(self.0.as_ptr()) }
    }
}

/// `UVDocRectifierAdapter` on the GPU.
#[derive(Debug)]
pub struct Mi355xRectifierAdapter {
    pub(crate) handle: RectHandle,
    info: AdapterInfo,
    _config: DocumentRectificationConfig,
}

impl ModelAdapter for Mi355xRectifierAdapter {
    type Task = DocumentRectificationTask;

    fn info(&self) -> AdapterInfo {
        self.info.clone()
    }

    fn execute(
        &self,
        input: <Self::Task as Task>::Input,
        _config: Option<&<Self::Task as Task>::Config>,
    ) -> Result<<Self::Task as Task>::Output, OCRError> {
        let batch_len = input.images.len();
        let mut rectified_images = Vec::with_capacity(batch_len);
        for img in &input.images {
            let (w, h) = (img.width(), img.height());
            let mut out = vec![0u8; w as usize * h as usize * 3];
            // SAFETY: img's buffer is w * h * 3 bytes; out has the same size, as oar_rect_run requires.
            let status = unsafe { sys::oar_rect_run(self.handle.0.as_ptr(), img.as_raw().as_ptr(), w, h, out.as_mut_ptr()) };
            check(status).map_err(|e: Mi355xError| {
                e.into_adapter_error("UVDocRectifierAdapter", format!("model forward (batch_size={})", batch_len))
            })?;
            let rectified = RgbImage::from_raw(w, h, out).ok_or_else(|| OCRError::InvalidInput {
                message: "UVDocRectifierAdapter: rectified buffer does not match the page size".to_string(),
            })?;
            rectified_images.push(rectified);
        }
        Ok(DocumentRectificationOutput { rectified_images })
    }

    fn supports_batching(&self) -> bool {
        true
    }

    fn recommended_batch_size(&self) -> usize {
        8
    }
}

/// Builder with the surface of `UVDocRectifierAdapterBuilder` (document_rectification_adapter.rs:66-131).
#[derive(Debug, Clone)]
pub struct Mi355xRectifierAdapterBuilder {
    config: DocumentRectificationConfig,
    /// `UVDocPreprocessConfig::rec_image_shape`, default [3, 512, 512] (uvdoc.rs:21-27)
    rec_image_shape: [usize; 3],
    model_name_override: Option<String>,
    device_id: i32,
}

impl Default for Mi355xRectifierAdapterBuilder {
    fn default() -> Self {
        Self::new()
    }
}

impl Mi355xRectifierAdapterBuilder {
    pub fn new() -> Self {
        Self {
            config: DocumentRectificationConfig::default(),
            rec_image_shape: [3, 512, 512],
            model_name_override: None,
            device_id: 0,
        }
    }

    pub fn model_name(mut self, model_name: impl Into<String>) -> Self {
        self.model_name_override = Some(model_name.into());
        self
    }

    /// [channels, height, width]; updates the task config as well (document_rectification_adapter.rs:133-141).
    pub fn input_shape(mut self, shape: [usize; 3]) -> Self {
        self.rec_image_shape = shape;
        self.config.rec_image_shape = shape;
        self
    }

    pub fn device_id(mut self, device_id: i32) -> Self {
        self.device_id = device_id;
        self
    }
}

impl AdapterBuilder for Mi355xRectifierAdapterBuilder {
    type Config = DocumentRectificationConfig;
    type Adapter = Mi355xRectifierAdapter;

    fn build(self, model_source: impl Into<ModelSource>) -> Result<Self::Adapter, OCRError> {
        let [_, h, w] = self.rec_image_shape;
        // height or width 0 = "feed pages at their own size" (uvdoc.rs:84-88), which is what
        // DocumentRectificationConfig::default()'s [3, 0, 0] becomes once with_config copies it into the preprocess config
        let (target_h, target_w) =
            if h == 0 || w == 0 { (sys::OAR_RECT_NATIVE_SIZE, sys::OAR_RECT_NATIVE_SIZE) } else { (h as u32, w as u32) };
        let cfg = sys::oar_rect_cfg { device_id: self.device_id, target_h, target_w };
        let source: ModelSource = model_source.into();
        let (bytes, shown) = model_bytes(&source)?;
        let mut raw: *mut sys::oar_rect = std::ptr::null_mut();
        // SAFETY: bytes valid for bytes.len(); cfg / raw valid for the call.
        let status = unsafe { sys::oar_rect_create(bytes.as_ptr(), bytes.len(), &cfg, &mut raw) };
        check(status).map_err(|e: Mi355xError| e.into_model_load(&shown))?;
        let handle = RectHandle(NonNull::new(raw).ok_or_else(|| OCRError::ConfigError {
            message: "oar_rect_create returned OAR_OK with a null handle".to_string(),
        })?);
        let mut info = AdapterInfo::new(
            "uvdoc_rectifier",
            TaskType::DocumentRectification,
            "Corrects geometric distortions in document images (MI355X backend)",
        );
        if let Some(model_name) = self.model_name_override {
            info.model_name = model_name;
        }
        Ok(Mi355xRectifierAdapter { handle, info, _config: self.config })
    }

    /// Like the reference's override: the task config's shape also becomes the preprocess shape
    /// (document_rectification_adapter.rs:84-91).
    fn with_config(mut self, config: Self::Config) -> Self {
        self.rec_image_shape = config.rec_image_shape;
        self.config = config;
        self
    }

    fn adapter_type(&self) -> &str {
        "uvdoc_rectifier"
    }
}

/// `OrtConfigurable` (core/traits/adapter.rs:126-129): lets the reference's generic construction path
/// (`build_optional_adapter`, src/oarocr/builder_utils.rs:60-80; `OAROCRBuilder::build`, src/oarocr/ocr.rs:311,393) configure
/// this builder unchanged.  Only the device ordinal of the session configuration applies (see `device_id_from_ort_config`).
impl OrtConfigurable for Mi355xRectifierAdapterBuilder {
    fn with_ort_config(mut self, config: OrtSessionConfig) -> Self {
        if let Some(device_id) = device_id_from_ort_config(&config) {
            self.device_id = device_id;
        }
        self
    }
}
