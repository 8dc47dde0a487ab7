//! `ModelAdapter` / `AdapterBuilder` implementations for oar-ocr backed by `libOarMi355x.so` (AMD MI355X, gfx950).
//!
//! Each adapter stands exactly where the ONNX-Runtime-backed adapter of the same task stands in the reference
//! (`oar-ocr-core/src/domain/adapters/*.rs`): same `Task`, same `execute(input, config)` contract, same
//! error wrapping (`OCRError::adapter_execution_error`), same builder surface.  What differs is what happens
//! underneath -- pre-processing, the network and post-processing all run on the GPU behind one C call
//! ("Seam B" of `include/oar_mi355x.h`), so the host only sees `RgbImage`s going in and boxes / texts coming out.
//!
//! | reference adapter (domain/adapters/...)                     | this crate                          | C entry points            |
//! |--------------------------------------------------------------|-------------------------------------|---------------------------|
//! | `text_detection_adapter.rs`  `TextDetectionAdapter`          | [`Mi355xTextDetectionAdapter`]      | `oar_det_*`               |
//! | `seal_text_detection_adapter.rs` `SealTextDetectionAdapter`  | [`Mi355xSealTextDetectionAdapter`]  | `oar_det_*` (box_type 1)  |
//! | `text_recognition_adapter.rs` `TextRecognitionAdapter`       | [`Mi355xTextRecognitionAdapter`]    | `oar_rec_*`, `oar_ctc_*`  |
//! | `document_orientation_adapter.rs`                            | [`Mi355xDocumentOrientationAdapter`]| `oar_cls_*`               |
//! | `text_line_orientation_adapter.rs`                           | [`Mi355xTextLineOrientationAdapter`]| `oar_cls_*`               |
//! | `document_rectification_adapter.rs` `UVDocRectifierAdapter`  | [`Mi355xRectifierAdapter`]          | `oar_rect_*`              |
//! | `layout_detection_adapter.rs` `LayoutDetectionAdapter` (PicoDet / RT-DETR) | [`Mi355xLayoutDetectionAdapter`]   | `oar_layout_*`            |
//! | `core/inference/ort_infer_execution.rs` `OrtInfer` (Seam A)  | [`Mi355xInfer`]                     | `oar_engine_*`            |
//! | `src/oarocr/ocr.rs` `OAROCR::predict`                        | [`Mi355xOcr`]                       | `oar_ocr_*`               |
//! | (no counterpart: one process per GPU)             | [`shard`]                           | `oar_shard_range`, `oar_ocr_pack`, `oar_packed_merge` |
//!
//! The build image of the backend repository has no Rust toolchain: this crate is source-only there, checked
//! lexically against the `-sys` crate (every `sys::` item it names exists) by `tests/test_rust_bindings_cpu.py`.

pub mod error;
pub mod ffi_util;
pub mod infer;
pub mod layout_detection;
pub mod orientation;
pub mod pipeline;
pub mod rectification;
pub mod seal_text_detection;
pub mod shard;
pub mod text_detection;
pub mod text_recognition;

pub use error::{Mi355xError, check};
pub use infer::Mi355xInfer;
pub use layout_detection::{Mi355xLayoutDetectionAdapter, Mi355xLayoutDetectionAdapterBuilder};
pub use orientation::{
    Mi355xDocumentOrientationAdapter, Mi355xDocumentOrientationAdapterBuilder,
    Mi355xTextLineOrientationAdapter, Mi355xTextLineOrientationAdapterBuilder,
};
pub use pipeline::{Mi355xOcr, Mi355xOcrBuilder, Mi355xOcrPage, Mi355xOcrRegion};
pub use rectification::{Mi355xRectifierAdapter, Mi355xRectifierAdapterBuilder};
pub use seal_text_detection::{Mi355xSealTextDetectionAdapter, Mi355xSealTextDetectionAdapterBuilder};
pub use shard::{PackedPages, merge_packed, shard_range};
pub use text_detection::{Mi355xTextDetectionAdapter, Mi355xTextDetectionAdapterBuilder};
pub use text_recognition::{Mi355xTextRecognitionAdapter, Mi355xTextRecognitionAdapterBuilder};

/// Number of MI355X devices the library can see (0: every `build()` fails with `OAR_DEVICE` -- there is no CPU
/// fallback behind these adapters; keep the ONNX Runtime adapters for hosts without a GPU).
pub fn device_count() -> usize {
    // SAFETY: no arguments, no preconditions.
    let n = unsafe { oar_mi355x_sys::oar_device_count() };
    n.max(0) as usize
}

/// `"libOarMi355x <ver> gfx950 <device name> CUs=<n>"`
pub fn version() -> String {
    let mut buf = vec![0u8; 256];
    // SAFETY: buf is writable for buf.len() bytes; the library NUL-terminates within cap.
    let n = unsafe { oar_mi355x_sys::oar_version(buf.as_mut_ptr().cast(), buf.len()) };
    buf.truncate(n.min(buf.len().saturating_sub(1)));
    String::from_utf8_lossy(&buf).into_owned()
}
