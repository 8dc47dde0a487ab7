//! Text recognition on the MI355X: stands where `TextRecognitionAdapter` stands
//! (oar-ocr-core/src/domain/adapters/text_recognition_adapter.rs:18-127).
//!
//! `execute` = `CRNNModel::forward_refs(images, return_word_box)` (models/recognition/crnn.rs:247-293) -- resize to the
//! batch tensor width + normalise, the SVTR/CRNN network, per-step argmax -- as one `oar_rec_run` call, followed by the
//! CTC collapse / text assembly (`decode_argmax_with_positions`, processors/decode.rs:549-614) and the adapter's own
//! `score >= threshold` filter, both inside `oar_ctc_decode`.

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
use oar_ocr_core::domain::tasks::{TextRecognitionConfig, TextRecognitionOutput, TextRecognitionTask};
use std::ptr::NonNull;

#[derive(Debug)]
struct RecHandle(NonNull<sys::oar_rec>);
// SAFETY: handles are usable from any thread; calls on one handle serialise inside the library (oar_mi355x.h, "Conventions").
unsafe impl Send for RecHandle {}
unsafe impl Sync for RecHandle {}
impl Drop for RecHandle {
    fn drop(&mut self) {
        // SAFETY: created by oar_rec_create, destroyed once.
        unsafe { sys::oar_rec_destroy(self.0.as_ptr()) }
    }
}

/// The decoder's character table (`CTCLabelDecode`, processors/decode.rs:352-421), kept inside the library.
#[derive(Debug)]
pub(crate) struct DictHandle(pub(crate) NonNull<sys::oar_ctc_dict>);
// SAFETY: an oar_ctc_dict is immutable after creation.
unsafe impl Send for DictHandle {}
unsafe impl Sync for DictHandle {}
impl Drop for DictHandle {
    fn drop(&mut self) {
        // SAFETY: created by oar_ctc_dict_create, destroyed once.
        unsafe { sys::oar_ctc_dict_destroy(self.0.as_ptr()) }
    }
}

impl DictHandle {
    /// `CTCLabelDecode::from_string_list(Some(lines), true, false)` (crnn.rs:384-389): only the first char of each entry
    /// counts, empty entries vanish, blank sits at index 0 and a space is appended.  `None` = the default alphabet of
    /// `CTCLabelDecode::new(None, true)` (decode.rs:83-88).
    pub(crate) fn new(character_dict: Option<&[String]>) -> Result<Self, OCRError> {
        let text: String = match character_dict {
            Some(lines) => {
                // one entry per line; an entry is represented by its first char, exactly what from_string_list keeps
                let mut s = String::new();
                for entry in lines {
                    if let Some(c) = entry.chars().next() {
                        s.push(c);
                    }
                    s.push('\n');
                }
                s
            }
            None => "0123456789abcdefghijklmnopqrstuvwxyz".chars().flat_map(|c| [c, '\n']).collect(),
        };
        let mut raw: *mut sys::oar_ctc_dict = std::ptr::null_mut();
        // SAFETY: text is valid UTF-8 of text.len() bytes; raw is a valid out-parameter.
        let status = unsafe { sys::oar_ctc_dict_create(text.as_ptr().cast(), text.len(), 1, &mut raw) };
        check(status).map_err(|e| OCRError::ConfigError { message: format!("character dictionary rejected: {e}") })?;
        NonNull::new(raw)
            .map(DictHandle)
            .ok_or_else(|| OCRError::ConfigError { message: "oar_ctc_dict_create returned a null handle".to_string() })
    }
}

struct RecResultGuard(sys::oar_rec_result);
impl Drop for RecResultGuard {
    fn drop(&mut self) {
        // SAFETY: filled by oar_rec_run or all-NULL.
        unsafe { sys::oar_rec_result_free(&mut self.0) }
    }
}

pub(crate) struct TextResultGuard(pub(crate) sys::oar_text_result);
impl Drop for TextResultGuard {
    fn drop(&mut self) {
        // SAFETY: filled by oar_ctc_decode / oar_ocr_decode or all-NULL.
        unsafe { sys::oar_text_result_free(&mut self.0) }
    }
}

impl TextResultGuard {
    pub(crate) fn empty() -> Self {
        TextResultGuard(sys::oar_text_result {
            n: 0,
            text_offsets: std::ptr::null_mut(),
            utf8: std::ptr::null_mut(),
            scores: std::ptr::null_mut(),
            char_offsets: std::ptr::null_mut(),
            char_cols: std::ptr::null_mut(),
            char_positions: std::ptr::null_mut(),
            seq_len: std::ptr::null_mut(),
            kept: std::ptr::null_mut(),
        })
    }

    /// The reference's `TextRecognitionOutput` (domain/tasks/text_recognition.rs:33-47).  With `with_positions == false`
    /// the three position fields are what the adapter produces from `decode_argmax` (crnn.rs:214-223 and the
    /// `chain(repeat(..))` in text_recognition_adapter.rs:70-84): empty vectors and a sequence length of 0.
    pub(crate) fn to_output(&self, with_positions: bool) -> TextRecognitionOutput {
        let r = &self.0;
        let n = r.n as usize;
        // SAFETY: lengths as documented for oar_text_result.
        let (text_off, scores, char_off, seq_len) = unsafe {
            (
                slice_or_empty(r.text_offsets, n + 1),
                slice_or_empty(r.scores, n),
                slice_or_empty(r.char_offsets, n + 1),
                slice_or_empty(r.seq_len, n),
            )
        };
        let total_bytes = text_off.last().copied().unwrap_or(0) as usize;
        let total_chars = char_off.last().copied().unwrap_or(0) as usize;
        // SAFETY: as above.
        let (utf8, cols, positions) = unsafe {
            (
                slice_or_empty(r.utf8.cast::<u8>(), total_bytes),
                slice_or_empty(r.char_cols, total_chars),
                slice_or_empty(r.char_positions, total_chars),
            )
        };
        let mut out = TextRecognitionOutput {
            texts: Vec::with_capacity(n),
            scores: Vec::with_capacity(n),
            char_positions: Vec::with_capacity(n),
            char_col_indices: Vec::with_capacity(n),
            sequence_lengths: Vec::with_capacity(n),
        };
        for i in 0..n {
            let (t0, t1) = (text_off[i] as usize, text_off[i + 1] as usize);
            out.texts.push(String::from_utf8_lossy(&utf8[t0..t1]).into_owned());
            out.scores.push(scores[i]);
            if with_positions {
                let (c0, c1) = (char_off[i] as usize, char_off[i + 1] as usize);
                out.char_positions.push(positions[c0..c1].to_vec());
                out.char_col_indices.push(cols[c0..c1].iter().map(|&c| c as usize).collect());
                out.sequence_lengths.push(seq_len[i] as usize);
            } else {
                out.char_positions.push(Vec::new());
                out.char_col_indices.push(Vec::new());
                out.sequence_lengths.push(0);
            }
        }
        out
    }
}

/// `TextRecognitionAdapter` with the recognizer on the GPU.
#[derive(Debug)]
pub struct Mi355xTextRecognitionAdapter {
    handle: RecHandle,
    dict: DictHandle,
    info: AdapterInfo,
    config: TextRecognitionConfig,
    return_word_box: bool,
}

impl ModelAdapter for Mi355xTextRecognitionAdapter {
    type Task = TextRecognitionTask;

    fn info(&self) -> AdapterInfo {
        self.info.clone()
    }

    fn execute(
        &self,
        input: <Self::Task as Task>::Input,
        config: Option<&<Self::Task as Task>::Config>,
    ) -> Result<<Self::Task as Task>::Output, OCRError> {
        let effective_config = config.unwrap_or(&self.config);
        let batch = ImageBatch::new(input.images.iter().map(AsRef::as_ref));
        let batch_len = batch.len();
        let context = || format!("forward (batch_size={}, return_word_box={})", batch_len, self.return_word_box);
        if batch.is_empty() {
            return Ok(TextResultGuard::empty().to_output(self.return_word_box));
        }

        let mut rec = RecResultGuard(sys::oar_rec_result {
            batch: 0,
            seq_len: 0,
            vocab: 0,
            tensor_width: 0,
            indices: std::ptr::null_mut(),
            probs: std::ptr::null_mut(),
        });
        // SAFETY: three arrays of batch_len entries; image buffers stay alive with `input`.
        let status = unsafe {
            sys::oar_rec_run(
                self.handle.0.as_ptr(),
                batch.ptrs.as_ptr(),
                batch.widths.as_ptr(),
                batch.heights.as_ptr(),
                batch_len as u32,
                &mut rec.0,
            )
        };
        check(status).map_err(|e| e.into_adapter_error("TextRecognitionAdapter", context()))?;

        let mut texts = TextResultGuard::empty();
        // SAFETY: indices / probs hold batch * seq_len entries (oar_rec_result); texts is a valid out-parameter.
        let status = unsafe {
            sys::oar_ctc_decode(
                self.dict.0.as_ptr(),
                rec.0.indices,
                rec.0.probs,
                rec.0.batch,
                rec.0.seq_len,
                effective_config.score_threshold,
                &mut texts.0,
            )
        };
        check(status).map_err(|e| e.into_adapter_error("TextRecognitionAdapter", context()))?;
        Ok(texts.to_output(self.return_word_box))
    }

    fn supports_batching(&self) -> bool {
        true
    }

    fn recommended_batch_size(&self) -> usize {
        // The reference reports 64 (text_recognition_adapter.rs:117-127), tuned for ONNX Runtime.  On 256 CUs the small
        // recognizers are launch-bound at 64 crops; 256 is this backend's sweet spot.  Batch composition changes the
        // tensor width a crop is padded to, hence its text in rare cases -- a caller that needs outputs identical to the
        // reference's default passes `region_batch_size(64)` to the pipeline builder.
        256
    }
}

/// Builder with the surface of `TextRecognitionAdapterBuilder` (text_recognition_adapter.rs:129-207).
#[derive(Debug, Clone)]
pub struct Mi355xTextRecognitionAdapterBuilder {
    config: TextRecognitionConfig,
    model_input_shape: [usize; 3],
    max_img_w: Option<usize>,
    character_dict: Option<Vec<String>>,
    return_word_box: bool,
    model_name_override: Option<String>,
    device_id: i32,
}

impl Default for Mi355xTextRecognitionAdapterBuilder {
    fn default() -> Self {
        Self::new()
    }
}

impl Mi355xTextRecognitionAdapterBuilder {
    pub fn new() -> Self {
        Self {
            config: TextRecognitionConfig::default(),
            model_input_shape: [3, 48, 320], // CRNNPreprocessConfig::default (crnn.rs:305-311)
            max_img_w: None,
            character_dict: None,
            return_word_box: false,
            model_name_override: None,
            device_id: 0,
        }
    }

    pub fn model_input_shape(mut self, shape: [usize; 3]) -> Self {
        self.model_input_shape = shape;
        self
    }

    pub fn model_name(mut self, model_name: impl Into<String>) -> Self {
        self.model_name_override = Some(model_name.into());
        self
    }

    pub fn character_dict(mut self, character_dict: Vec<String>) -> Self {
        self.character_dict = Some(character_dict);
        self
    }

    pub fn score_thresh(mut self, score_thresh: f32) -> Self {
        self.config.score_threshold = score_thresh;
        self
    }

    pub fn max_img_w(mut self, max_img_w: usize) -> Self {
        self.max_img_w = Some(max_img_w);
        self
    }

    pub fn return_word_box(mut self, enable: bool) -> Self {
        self.return_word_box = enable;
        self
    }

    pub fn device_id(mut self, device_id: i32) -> Self {
        self.device_id = device_id;
        self
    }

    fn base_adapter_info() -> AdapterInfo {
        AdapterInfo::new(
            "text_recognition",
            TaskType::TextRecognition,
            "Recognizes text content from image regions (MI355X backend)",
        )
    }
}

impl AdapterBuilder for Mi355xTextRecognitionAdapterBuilder {
    type Config = TextRecognitionConfig;
    type Adapter = Mi355xTextRecognitionAdapter;

    fn build(self, model_source: impl Into<ModelSource>) -> Result<Self::Adapter, OCRError> {
        self.config.validate().map_err(|err| OCRError::ConfigError { message: err.to_string() })?;
        let [c, h, w] = self.model_input_shape;
        let cfg = sys::oar_rec_cfg {
            device_id: self.device_id,
            rec_image_shape: [c as u32, h as u32, w as u32],
            max_img_w: self.max_img_w.unwrap_or(0) as u32, // 0 => DEFAULT_MAX_IMG_WIDTH 3200 (core/constants.rs:8)
            use_hip_graph: 0,
            profile: 0,
            reserved: 0,
        };
        let dict = DictHandle::new(self.character_dict.as_deref())?;

        let source: ModelSource = model_source.into();
        let (bytes, shown) = model_bytes(&source)?;
        let mut raw: *mut sys::oar_rec = std::ptr::null_mut();
        // SAFETY: bytes valid for bytes.len(); cfg / raw valid for the call.
        let status = unsafe { sys::oar_rec_create(bytes.as_ptr(), bytes.len(), &cfg, &mut raw) };
        check(status).map_err(|e: Mi355xError| e.into_model_load(&shown))?;
        let handle = RecHandle(NonNull::new(raw).ok_or_else(|| OCRError::ConfigError {
            message: "oar_rec_create returned OAR_OK with a null handle".to_string(),
        })?);

        let mut info = Self::base_adapter_info();
        if let Some(model_name) = self.model_name_override {
            info.model_name = model_name;
        }
        Ok(Mi355xTextRecognitionAdapter {
            handle,
            dict,
            info,
            config: self.config,
            return_word_box: self.return_word_box,
        })
    }

    fn with_config(mut self, config: Self::Config) -> Self {
        self.config = config;
        self
    }

    fn adapter_type(&self) -> &str {
        "text_recognition"
    }
}

/// `OrtConfigurable` (core/traits/adapter.rs:126-129): lets the reference's generic construction path
/// (`build_optional_adapter`, src/oarocr/builder_utils.rs:60-80; `OAROCRBuilder::build`, src/oarocr/ocr.rs:311,393) configure
/// this builder unchanged.  Only the device ordinal of the session configuration applies (see `device_id_from_ort_config`).
impl OrtConfigurable for Mi355xTextRecognitionAdapterBuilder {
    fn with_ort_config(mut self, config: OrtSessionConfig) -> Self {
        if let Some(device_id) = device_id_from_ort_config(&config) {
            self.device_id = device_id;
        }
        self
    }
}
