//! PP-LCNet orientation classifiers on the MI355X: stand where `DocumentOrientationAdapter`
//! (oar-ocr-core/src/domain/adapters/document_orientation_adapter.rs:17-125) and `TextLineOrientationAdapter`
//! (text_line_orientation_adapter.rs:20-122) stand.
//!
//! `execute` = `PPLCNetModel::forward_refs(images, postprocess{labels, topk})` (models/classification/pp_lcnet.rs:139-330):
//! Triangle resize (+ centre crop when `resize_short` is set), ImageNet normalisation, the network, top-k -- one
//! `oar_cls_run` call.  The handle is created with topk = number of classes; a per-call `config.topk` truncates the rows,
//! which is what the reference's stable descending sort followed by `take(topk)` yields (utils/topk.rs:181-199).

use crate::error::{Mi355xError, check};
use crate::ffi_util::{ImageBatch, model_bytes, slice_or_empty};
use image::RgbImage;
use crate::ffi_util::device_id_from_ort_config;
use oar_mi355x_sys as sys;
use oar_ocr_core::core::OCRError;
use oar_ocr_core::core::config::ConfigValidator;
use oar_ocr_core::core::inference::ModelSource;
use oar_ocr_core::core::config::OrtSessionConfig;
use oar_ocr_core::core::traits::adapter::{AdapterBuilder, AdapterInfo, ModelAdapter, OrtConfigurable};
use oar_ocr_core::core::traits::task::{Task, TaskType};
use oar_ocr_core::domain::tasks::{
    Classification, DocumentOrientationConfig, DocumentOrientationOutput, DocumentOrientationTask,
    TextLineOrientationConfig, TextLineOrientationOutput, TextLineOrientationTask,
};
use std::ptr::NonNull;
use std::sync::Arc;

#[derive(Debug)]
pub(crate) struct ClsHandle(pub(crate) NonNull<sys::oar_cls>);
// SAFETY: handles are usable from any thread; calls on one handle serialise inside the library.
unsafe impl Send for ClsHandle {}
unsafe impl Sync for ClsHandle {}
impl Drop for ClsHandle {
    fn drop(&mut self) {
        // SAFETY: created by oar_cls_create, destroyed once.
        unsafe { sys::oar_cls_destroy(self.0.as_ptr()) }
    }
}

struct ClsResultGuard(sys::oar_cls_result);
impl Drop for ClsResultGuard {
    fn drop(&mut self) {
        // SAFETY: filled by oar_cls_run or all-NULL.
        unsafe { sys::oar_cls_result_free(&mut self.0) }
    }
}

impl ClsHandle {
    /// `input_shape` is (height, width) like `PPLCNetPreprocessConfig::input_shape`; `resize_short` = `Some(256)` for the
    /// document classifier (pp_lcnet.rs:40-53), `None` for the text-line classifier (text_line_orientation_adapter.rs:161-165).
    pub(crate) fn create(
        source: ModelSource,
        input_shape: (u32, u32),
        resize_short: Option<u32>,
        n_classes: usize,
        device_id: i32,
    ) -> Result<Self, OCRError> {
        let cfg = sys::oar_cls_cfg {
            device_id,
            input_h: input_shape.0,
            input_w: input_shape.1,
            resize_short: resize_short.unwrap_or(0),
            topk: n_classes as u32,
            batch: 0,
        };
        let (bytes, shown) = model_bytes(&source)?;
        let mut raw: *mut sys::oar_cls = std::ptr::null_mut();
        // SAFETY: bytes valid for bytes.len(); cfg / raw valid for the call.
        let status = unsafe { sys::oar_cls_create(bytes.as_ptr(), bytes.len(), &cfg, &mut raw) };
        check(status).map_err(|e: Mi355xError| e.into_model_load(&shown))?;
        NonNull::new(raw)
            .map(ClsHandle)
            .ok_or_else(|| OCRError::ConfigError { message: "oar_cls_create returned a null handle".to_string() })
    }

    /// Per image: up to `topk` (class_id, label, score), best first.
    pub(crate) fn classify(
        &self,
        images: &[Arc<RgbImage>],
        labels: &[String],
        topk: usize,
        angle_step: usize,
    ) -> Result<Vec<Vec<Classification>>, Mi355xError> {
        let batch = ImageBatch::new(images.iter().map(AsRef::as_ref));
        if batch.is_empty() {
            return Ok(Vec::new());
        }
        let mut res = ClsResultGuard(sys::oar_cls_result {
            n_images: 0,
            topk: 0,
            n_classes: 0,
            class_ids: std::ptr::null_mut(),
            scores: std::ptr::null_mut(),
        });
        // SAFETY: three arrays of batch.len() entries; image buffers outlive the call.
        let status = unsafe {
            sys::oar_cls_run(
                self.0.as_ptr(),
                batch.ptrs.as_ptr(),
                batch.widths.as_ptr(),
                batch.heights.as_ptr(),
                batch.len() as u32,
                &mut res.0,
            )
        };
        check(status)?;
        let (n, k) = (res.0.n_images as usize, res.0.topk as usize);
        // SAFETY: n_images * topk entries each (oar_cls_result).
        let (ids, scores) = unsafe { (slice_or_empty(res.0.class_ids, n * k), slice_or_empty(res.0.scores, n * k)) };
        let keep = topk.min(k);
        let mut out = Vec::with_capacity(n);
        for i in 0..n {
            let mut row = Vec::with_capacity(keep);
            for j in 0..keep {
                let class_id = ids[i * k + j].max(0) as usize;
                // labels from the postprocess config when the class has one, else the angle itself
                // (document_orientation_adapter.rs:80-86 / text_line_orientation_adapter.rs:92-98)
                let label = labels.get(class_id).cloned().unwrap_or_else(|| format!("{}", class_id * angle_step));
                row.push(Classification::new(class_id, label, scores[i * k + j]));
            }
            out.push(row);
        }
        Ok(out)
    }
}

// ------------------------------------------------------------------------------------------------ document orientation

/// `DocumentOrientationAdapter` on the GPU.
#[derive(Debug)]
pub struct Mi355xDocumentOrientationAdapter {
    pub(crate) handle: ClsHandle,
    info: AdapterInfo,
    config: DocumentOrientationConfig,
}

impl Mi355xDocumentOrientationAdapter {
    pub const DEFAULT_INPUT_SHAPE: (u32, u32) = (224, 224);

    pub fn labels() -> Vec<String> {
        vec!["0".to_string(), "90".to_string(), "180".to_string(), "270".to_string()]
    }
}

impl ModelAdapter for Mi355xDocumentOrientationAdapter {
    type Task = DocumentOrientationTask;

    fn info(&self) -> AdapterInfo {
        self.info.clone()
    }

    fn execute(
        &self,
        input: <Self::Task as Task>::Input,
        config: Option<&<Self::Task as Task>::Config>,
    ) -> Result<<Self::Task as Task>::Output, OCRError> {
        let effective_config = config.unwrap_or(&self.config);
        let classifications = self
            .handle
            .classify(&input.images, &Self::labels(), effective_config.topk, 90)
            .map_err(|e| {
                e.into_adapter_error(
                    "DocumentOrientationAdapter",
                    format!("failed to classify document orientation (topk={})", effective_config.topk),
                )
            })?;
        Ok(DocumentOrientationOutput { classifications })
    }

    fn supports_batching(&self) -> bool {
        true
    }

    fn recommended_batch_size(&self) -> usize {
        32
    }
}

/// Builder with the surface of `DocumentOrientationAdapterBuilder` (document_orientation_adapter.rs:127-190).
#[derive(Debug, Clone)]
pub struct Mi355xDocumentOrientationAdapterBuilder {
    config: DocumentOrientationConfig,
    input_shape: (u32, u32),
    model_name_override: Option<String>,
    device_id: i32,
}

impl Default for Mi355xDocumentOrientationAdapterBuilder {
    fn default() -> Self {
        Self::new()
    }
}

impl Mi355xDocumentOrientationAdapterBuilder {
    pub fn new() -> Self {
        Self {
            config: DocumentOrientationConfig::default(),
            input_shape: Mi355xDocumentOrientationAdapter::DEFAULT_INPUT_SHAPE,
            model_name_override: None,
            device_id: 0,
        }
    }

    pub fn input_shape(mut self, input_shape: (u32, u32)) -> Self {
        self.input_shape = input_shape;
        self
    }

    pub fn model_name(mut self, model_name: impl Into<String>) -> Self {
        self.model_name_override = Some(model_name.into());
        self
    }

    pub fn device_id(mut self, device_id: i32) -> Self {
        self.device_id = device_id;
        self
    }
}

impl AdapterBuilder for Mi355xDocumentOrientationAdapterBuilder {
    type Config = DocumentOrientationConfig;
    type Adapter = Mi355xDocumentOrientationAdapter;

    fn build(self, model_source: impl Into<ModelSource>) -> Result<Self::Adapter, OCRError> {
        self.config.validate().map_err(|err| OCRError::ConfigError { message: err.to_string() })?;
        // pp_lcnet_preprocess(input_shape): resize_short stays at the default Some(256) (preprocessing.rs:12-17)
        let handle = ClsHandle::create(
            model_source.into(),
            self.input_shape,
            Some(256),
            Mi355xDocumentOrientationAdapter::labels().len(),
            self.device_id,
        )?;
        let mut info = AdapterInfo::new(
            "document_orientation",
            TaskType::DocumentOrientation,
            "Classifies document image orientation (0°, 90°, 180°, 270°) (MI355X backend)",
        );
        if let Some(model_name) = self.model_name_override {
            info.model_name = model_name;
        }
        Ok(Mi355xDocumentOrientationAdapter { handle, info, config: self.config })
    }

    fn with_config(mut self, config: Self::Config) -> Self {
        self.config = config;
        self
    }

    fn adapter_type(&self) -> &str {
        "document_orientation"
    }
}

// ------------------------------------------------------------------------------------------------ text-line orientation

/// `TextLineOrientationAdapter` on the GPU.
#[derive(Debug)]
pub struct Mi355xTextLineOrientationAdapter {
    pub(crate) handle: ClsHandle,
    info: AdapterInfo,
    config: TextLineOrientationConfig,
}

impl Mi355xTextLineOrientationAdapter {
    /// (height, width)
    pub const DEFAULT_INPUT_SHAPE: (u32, u32) = (80, 160);

    pub fn labels() -> Vec<String> {
        vec!["0".to_string(), "180".to_string()]
    }
}

impl ModelAdapter for Mi355xTextLineOrientationAdapter {
    type Task = TextLineOrientationTask;

    fn info(&self) -> AdapterInfo {
        self.info.clone()
    }

    fn execute(
        &self,
        input: <Self::Task as Task>::Input,
        config: Option<&<Self::Task as Task>::Config>,
    ) -> Result<<Self::Task as Task>::Output, OCRError> {
        let effective_config = config.unwrap_or(&self.config);
        let classifications = self
            .handle
            .classify(&input.images, &Self::labels(), effective_config.topk, 180)
            .map_err(|e| {
                e.into_adapter_error(
                    "TextLineOrientationAdapter",
                    format!("failed to classify text line orientation (topk={})", effective_config.topk),
                )
            })?;
        Ok(TextLineOrientationOutput { classifications })
    }

    fn supports_batching(&self) -> bool {
        true
    }

    fn recommended_batch_size(&self) -> usize {
        64
    }
}

/// Builder with the surface of `TextLineOrientationAdapterBuilder` (text_line_orientation_adapter.rs:124-195).
#[derive(Debug, Clone)]
pub struct Mi355xTextLineOrientationAdapterBuilder {
    config: TextLineOrientationConfig,
    input_shape: (u32, u32),
    model_name_override: Option<String>,
    device_id: i32,
}

impl Default for Mi355xTextLineOrientationAdapterBuilder {
    fn default() -> Self {
        Self::new()
    }
}

impl Mi355xTextLineOrientationAdapterBuilder {
    pub fn new() -> Self {
        Self {
            config: TextLineOrientationConfig::default(),
            input_shape: Mi355xTextLineOrientationAdapter::DEFAULT_INPUT_SHAPE,
            model_name_override: None,
            device_id: 0,
        }
    }

    pub fn input_shape(mut self, input_shape: (u32, u32)) -> Self {
        self.input_shape = input_shape;
        self
    }

    pub fn model_name(mut self, model_name: impl Into<String>) -> Self {
        self.model_name_override = Some(model_name.into());
        self
    }

    pub fn device_id(mut self, device_id: i32) -> Self {
        self.device_id = device_id;
        self
    }
}

impl AdapterBuilder for Mi355xTextLineOrientationAdapterBuilder {
    type Config = TextLineOrientationConfig;
    type Adapter = Mi355xTextLineOrientationAdapter;

    fn build(self, model_source: impl Into<ModelSource>) -> Result<Self::Adapter, OCRError> {
        self.config.validate().map_err(|err| OCRError::ConfigError { message: err.to_string() })?;
        // direct resize to (h, w): preprocess_config.resize_short = None (text_line_orientation_adapter.rs:161-165)
        let handle = ClsHandle::create(
            model_source.into(),
            self.input_shape,
            None,
            Mi355xTextLineOrientationAdapter::labels().len(),
            self.device_id,
        )?;
        let mut info = AdapterInfo::new(
            "text_line_orientation",
            TaskType::TextLineOrientation,
            "Classifies text line orientation (0°, 180°) (MI355X backend)",
        );
        if let Some(model_name) = self.model_name_override {
            info.model_name = model_name;
        }
        Ok(Mi355xTextLineOrientationAdapter { handle, info, config: self.config })
    }

    fn with_config(mut self, config: Self::Config) -> Self {
        self.config = config;
        self
    }

    fn adapter_type(&self) -> &str {
        "text_line_orientation"
    }
}

/// `OrtConfigurable` (core/traits/adapter.rs:126-129): lets the reference's generic construction path
/// (`build_optional_adapter`, src/oarocr/builder_utils.rs:60-80; `OAROCRBuilder::build`, src/oarocr/ocr.rs:311,393) configure
/// this builder unchanged.  Only the device ordinal of the session configuration applies (see `device_id_from_ort_config`).
impl OrtConfigurable for Mi355xDocumentOrientationAdapterBuilder {
    fn with_ort_config(mut self, config: OrtSessionConfig) -> Self {
        if let Some(device_id) = device_id_from_ort_config(&config) {
            self.device_id = device_id;
        }
        self
    }
}

/// `OrtConfigurable` (core/traits/adapter.rs:126-129): lets the reference's generic construction path
/// (`build_optional_adapter`, src/oarocr/builder_utils.rs:60-80; `OAROCRBuilder::build`, src/oarocr/ocr.rs:311,393) configure
/// this builder unchanged.  Only the device ordinal of the session configuration applies (see `device_id_from_ort_config`).
impl OrtConfigurable for Mi355xTextLineOrientationAdapterBuilder {
    fn with_ort_config(mut self, config: OrtSessionConfig) -> Self {
        if let Some(device_id) = device_id_from_ort_config(&config) {
            self.device_id = device_id;
        }
        self
    }
}
