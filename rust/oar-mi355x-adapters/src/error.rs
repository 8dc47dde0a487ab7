//! Status codes of the C ABI -> Rust errors.
//!
//! The library keeps a thread-local message for the last failure (`oar_last_error`); [`check`] turns a
//! non-zero `oar_status` into a [`Mi355xError`] carrying both.  The adapters then wrap it exactly the way
//! the reference adapters wrap a failed `model.forward` (`OCRError::adapter_execution_error`,
//! core/errors/types.rs:369-379), so callers see the same `OCRError::Processing { kind: AdapterExecution, .. }`.

use oar_mi355x_sys as sys;
use oar_ocr_core::core::OCRError;
use std::path::Path;

/// A failed call into `libOarMi355x.so`.
#[derive(Debug, Clone, thiserror::Error)]
#[error("libOarMi355x status {status} ({name}): {message}")]
pub struct Mi355xError {
    /// the `oar_status` value
    pub status: i32,
    /// its symbolic name
    pub name: &'static str,
    /// the library's message for this thread's last failure
    pub message: String,
}

fn status_name(status: sys::oar_status) -> &'static str {
    match status {
        sys::OAR_OK => "OAR_OK",
        sys::OAR_INVALID_INPUT => "OAR_INVALID_INPUT",
        sys::OAR_MODEL_LOAD => "OAR_MODEL_LOAD",
        sys::OAR_UNSUPPORTED_OP => "OAR_UNSUPPORTED_OP",
        sys::OAR_SHAPE_MISMATCH => "OAR_SHAPE_MISMATCH",
        sys::OAR_DEVICE => "OAR_DEVICE",
        sys::OAR_OOM => "OAR_OOM",
        sys::OAR_INTERNAL => "OAR_INTERNAL",
        _ => "unknown status",
    }
}

/// `Ok(())` for `OAR_OK`, otherwise the status plus this thread's last error message.
pub fn check(status: sys::oar_status) -> Result<(), Mi355xError> {
    if status == sys::OAR_OK {
        return Ok(());
    }
    let mut buf = vec![0u8; 4096];
    // SAFETY: buf is writable for buf.len() bytes; the library truncates and NUL-terminates within cap.
    let n = unsafe { sys::oar_last_error(buf.as_mut_ptr().cast(), buf.len()) };
    buf.truncate(n.min(buf.len() - 1));
    Err(Mi355xError {
        status,
        name: status_name(status),
        message: String::from_utf8_lossy(&buf).into_owned(),
    })
}

impl Mi355xError {
    /// The error a builder returns when the model cannot be loaded: the reference builders end in
    /// `OCRError::model_load_error(path, reason, suggestion, source)` (core/errors/constructors.rs:481-500).
    pub fn into_model_load(self, model_path: impl AsRef<Path>) -> OCRError {
        let suggestion = match self.status {
            sys::OAR_DEVICE => Some("no gfx950 device is visible to this process; this backend has no CPU fallback"),
            sys::OAR_UNSUPPORTED_OP => Some("the graph uses an operator the MI355X engine does not implement; `oar_onnx_inspect` lists them"),
            _ => None,
        };
        let reason = self.message.clone();
        OCRError::model_load_error(model_path, reason, suggestion, Some(self))
    }

    /// Invalid caller input keeps its reference type (`OCRError::InvalidInput`, core/errors/types.rs:112-118);
    /// everything else becomes the adapter-execution error the reference adapters produce.
    pub fn into_adapter_error(self, adapter: &str, context: String) -> OCRError {
        if self.status == sys::OAR_INVALID_INPUT {
            return OCRError::InvalidInput { message: format!("{adapter}: {context}: {}", self.message) };
        }
        OCRError::adapter_execution_error(adapter, context, self)
    }
}
