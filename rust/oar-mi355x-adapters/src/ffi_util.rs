//! Helpers shared by the adapters: model bytes from a `ModelSource`, image batches as the C ABI wants them.

use image::RgbImage;
use oar_ocr_core::core::OCRError;
use oar_ocr_core::core::config::{OrtExecutionProvider, OrtSessionConfig};
use oar_ocr_core::core::inference::ModelSource;
use std::path::PathBuf;
use std::sync::Arc;

/// `.onnx` bytes of a `ModelSource::{Path, Memory}` (core/inference/model_source.rs:21-28) plus the path used in
/// error messages.  The C ABI always takes bytes, so `Path` sources are read here.
pub fn model_bytes(source: &ModelSource) -> Result<(Arc<[u8]>, PathBuf), OCRError> {
    let shown = source.display_path();
    match source {
        ModelSource::Memory(bytes) => Ok((Arc::clone(bytes), shown)),
        ModelSource::Path(path) => {
            let bytes = std::fs::read(path).map_err(|e| {
                OCRError::model_load_error(path, format!("cannot read model file: {e}"), Some("check the model path"), Some(e))
            })?;
            Ok((Arc::from(bytes.into_boxed_slice()), shown))
        }
    }
}

/// Borrowed view of a batch of tightly packed RGB8 images in the three parallel arrays every Seam-B entry point takes
/// (`rgb`, `widths`, `heights`).  `RgbImage`'s buffer is exactly width * height * 3 bytes, row-major, no padding.
pub struct ImageBatch<'a> {
    pub ptrs: Vec<*const u8>,
    pub widths: Vec<u32>,
    pub heights: Vec<u32>,
    _images: std::marker::PhantomData<&'a RgbImage>,
}

impl<'a> ImageBatch<'a> {
    pub fn new<I>(images: I) -> Self
    where
        I: IntoIterator<Item = &'a RgbImage>,
    {
        let mut ptrs = Vec::new();
        let mut widths = Vec::new();
        let mut heights = Vec::new();
        for img in images {
            ptrs.push(img.as_raw().as_ptr());
            widths.push(img.width());
            heights.push(img.height());
        }
        Self { ptrs, widths, heights, _images: std::marker::PhantomData }
    }

    pub fn len(&self) -> usize {
        self.ptrs.len()
    }

    pub fn is_empty(&self) -> bool {
        self.ptrs.is_empty()
    }
}

/// `slice::from_raw_parts` that tolerates the NULL / 0 pair an empty result carries.
///
/// # Safety
/// When `len > 0`, `ptr` must be valid for reads of `len` elements for the lifetime `'a`.
pub unsafe fn slice_or_empty<'a, T>(ptr: *const T, len: usize) -> &'a [T] {
    if ptr.is_null() || len == 0 {
        &[]
    } else {
        // SAFETY: guaranteed by the caller.
        unsafe { std::slice::from_raw_parts(ptr, len) }
    }
}

/// What an `OrtSessionConfig` means to this backend (`OrtConfigurable::with_ort_config`, core/traits/adapter.rs:126-129).
/// The reference's generic construction path hands every adapter builder the pipeline's session configuration
/// (`build_optional_adapter`, src/oarocr/builder_utils.rs:60-80; `OAROCRBuilder::build`, src/oarocr/ocr.rs:311,393).  Of
/// that configuration only the DEVICE is meaningful here: the first execution provider that names a device id (CUDA,
/// TensorRT, DirectML -- the accelerator entries a caller already has in its config) selects the MI355X with the same
/// ordinal; thread counts, arena / optimisation levels and provider options configure ONNX Runtime and are ignored.
/// `None`: the config names no device (CPU-only / OpenVINO / CoreML / WebGPU lists, or no list) -- the builder keeps its own.
pub fn device_id_from_ort_config(config: &OrtSessionConfig) -> Option<i32> {
    config.execution_providers.as_ref()?.iter().find_map(|ep| match ep {
        OrtExecutionProvider::CUDA { device_id, .. }
        | OrtExecutionProvider::TensorRT { device_id, .. }
        | OrtExecutionProvider::DirectML { device_id } => Some(device_id.unwrap_or(0)),
        _ => None,
    })
}
