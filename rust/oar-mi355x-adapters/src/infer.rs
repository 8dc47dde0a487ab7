//! Seam A: an `OrtInfer`-shaped engine (oar-ocr-core/src/core/inference/mod.rs:31-115,
//! ort_infer_execution.rs:121-306) for code that keeps the reference's own pre- and post-processing on the CPU and only
//! wants the network on the MI355X.  `.onnx` bytes in; named f32 tensors in; `TensorOutput::{F32, I64}` out.
//!
//! A model struct of the reference that holds an `OrtInfer` (`DBModel`, `CRNNModel`, `PPLCNetModel`, `UVDocModel`, ...)
//! can hold a `Mi355xInfer` instead: the method names, argument meaning and error behaviour are the same.

use crate::error::{Mi355xError, check};
use crate::ffi_util::{model_bytes, slice_or_empty};
use oar_mi355x_sys as sys;
use oar_ocr_core::core::OCRError;
use oar_ocr_core::core::inference::{ModelSource, TensorInput, TensorOutput};
use std::borrow::Cow;
use std::ffi::{CString, c_void};
use std::ptr::NonNull;

#[derive(Debug)]
struct EngineHandle(NonNull<sys::oar_engine>);
// SAFETY: handles are usable from any thread; calls on one handle serialise on an internal mutex, which mirrors the
// `Mutex<Session>` the reference serialises on (core/inference/mod.rs:31-37).
unsafe impl Send for EngineHandle {}
unsafe impl Sync for EngineHandle {}
impl Drop for EngineHandle {
    fn drop(&mut self) {
        // SAFETY: created by oar_engine_create, destroyed once.
        unsafe { sys::oar_engine_destroy(self.0.as_ptr()) }
    }
}

/// One named input with its data in row-major order (borrowed when the ndarray already is, copied otherwise).
struct PreparedInput<'a> {
    name: CString,
    dims: Vec<i64>,
    data: Cow<'a, [f32]>,
}

fn prepare<'a>(name: &str, input: &TensorInput<'a>) -> Result<PreparedInput<'a>, OCRError> {
    fn flat<'b, D: ndarray::Dimension>(a: &'b ndarray::Array<f32, D>) -> Cow<'b, [f32]> {
        match a.as_slice() {
            Some(s) => Cow::Borrowed(s),
            None => Cow::Owned(a.iter().copied().collect()), // logical (row-major) order
        }
    }
    let (dims, data): (Vec<i64>, Cow<'a, [f32]>) = match input {
        TensorInput::Array2(a) => (a.shape().iter().map(|&d| d as i64).collect(), flat(*a)),
        TensorInput::Array3(a) => (a.shape().iter().map(|&d| d as i64).collect(), flat(*a)),
        TensorInput::Array4(a) => (a.shape().iter().map(|&d| d as i64).collect(), flat(*a)),
    };
    let name = CString::new(name).map_err(|_| OCRError::InvalidInput {
        message: format!("input name {name:?} contains a NUL byte"),
    })?;
    Ok(PreparedInput { name, dims, data })
}

/// The MI355X engine behind the reference's `OrtInfer` interface.
#[derive(Debug)]
pub struct Mi355xInfer {
    handle: EngineHandle,
    input_name: String,
    model_name: String,
}

impl Mi355xInfer {
    /// `OrtInfer::new(model_source, input_name)` (core/inference/ort_infer_builders.rs:9-70).  `input_name: None` takes
    /// the graph's first non-initializer input, like the reference's default "x" resolution does for Paddle exports.
    pub fn new(model_source: impl Into<ModelSource>, input_name: Option<&str>, device_id: i32) -> Result<Self, OCRError> {
        let source: ModelSource = model_source.into();
        let (bytes, shown) = model_bytes(&source)?;
        let cfg = sys::oar_engine_cfg { device_id, use_hip_graph: 0, profile: 0, precision: sys::OAR_PRECISION_F32 as i32, stream: std::ptr::null_mut() };
        let mut raw: *mut sys::oar_engine = std::ptr::null_mut();
        // SAFETY: bytes valid for bytes.len(); cfg / raw valid for the call.
        let status = unsafe { sys::oar_engine_create(bytes.as_ptr(), bytes.len(), &cfg, &mut raw) };
        check(status).map_err(|e: Mi355xError| e.into_model_load(&shown))?;
        let handle = EngineHandle(NonNull::new(raw).ok_or_else(|| OCRError::ConfigError {
            message: "oar_engine_create returned OAR_OK with a null handle".to_string(),
        })?);

        let input_name = match input_name {
            Some(n) => n.to_string(),
            None => {
                let mut buf = vec![0u8; 256];
                // SAFETY: buf writable for buf.len() bytes.
                let status = unsafe { sys::oar_engine_input_name(handle.0.as_ptr(), buf.as_mut_ptr().cast(), buf.len()) };
                check(status).map_err(|e| e.into_model_load(&shown))?;
                let end = buf.iter().position(|&b| b == 0).unwrap_or(buf.len());
                String::from_utf8_lossy(&buf[..end]).into_owned()
            }
        };
        let model_name = shown.file_stem().map(|s| s.to_string_lossy().into_owned()).unwrap_or_else(|| "unknown_model".to_string());
        Ok(Self { handle, input_name, model_name })
    }

    /// core/inference/mod.rs:52-54
    pub fn input_name(&self) -> &str {
        &self.input_name
    }

    fn io(&self) -> (Vec<sys::oar_io_info>, Vec<sys::oar_io_info>) {
        const CAP: usize = 32;
        let blank = sys::oar_io_info { name: [0; 64], dtype: 0, rank: -1, dims: [0; 8] };
        let (mut ins, mut outs) = (vec![blank; CAP], vec![blank; CAP]);
        let (mut n_in, mut n_out) = (0i32, 0i32);
        // SAFETY: both arrays hold CAP entries; the counters are valid out-parameters.
        let status = unsafe {
            sys::oar_engine_io(self.handle.0.as_ptr(), ins.as_mut_ptr(), CAP as i32, &mut n_in, outs.as_mut_ptr(), CAP as i32, &mut n_out)
        };
        if check(status).is_err() {
            return (Vec::new(), Vec::new());
        }
        ins.truncate(n_in.max(0) as usize);
        outs.truncate(n_out.max(0) as usize);
        (ins, outs)
    }

    fn io_name(info: &sys::oar_io_info) -> String {
        let bytes: Vec<u8> = info.name.iter().take_while(|&&c| c != 0).map(|&c| c as u8).collect();
        String::from_utf8_lossy(&bytes).into_owned()
    }

    fn io_shape(info: &sys::oar_io_info) -> Option<Vec<i64>> {
        if info.rank < 0 { None } else { Some(info.dims[..info.rank as usize].to_vec()) }
    }

    /// core/inference/mod.rs:66-79
    pub fn input_names_from_model(&self) -> Vec<String> {
        self.io().0.iter().map(Self::io_name).collect()
    }

    /// core/inference/mod.rs:81-92: dynamic dimensions are -1.
    pub fn primary_input_shape(&self) -> Option<Vec<i64>> {
        self.io().0.first().and_then(Self::io_shape)
    }

    /// core/inference/mod.rs:94-112
    pub fn output_shapes(&self) -> Vec<(String, Vec<i64>)> {
        self.io().1.iter().filter_map(|o| Self::io_shape(o).map(|s| (Self::io_name(o), s))).collect()
    }

    fn error_context(&self, inputs: &[(&str, TensorInput)]) -> String {
        let names: Vec<&str> = inputs.iter().map(|(n, _)| *n).collect();
        format!("inputs {:?}, primary input shape {:?}", names, inputs.first().map(|(_, t)| t.shape().to_vec()))
    }

    fn wrap(&self, e: Mi355xError, context: String) -> OCRError {
        if e.status == sys::OAR_INVALID_INPUT {
            OCRError::InvalidInput { message: format!("Model '{}': {}", self.model_name, e.message) }
        } else {
            OCRError::inference_error(&self.model_name, &context, e)
        }
    }

    /// `OrtInfer::infer` (ort_infer_execution.rs:121-219): every declared graph input must be given by name.
    pub fn infer(&self, inputs: &[(&str, TensorInput)]) -> Result<Vec<(String, TensorOutput)>, OCRError> {
        if inputs.is_empty() {
            return Err(OCRError::InvalidInput { message: "No inputs provided for inference".to_string() });
        }
        let prepared: Vec<PreparedInput> = inputs.iter().map(|(n, t)| prepare(n, t)).collect::<Result<_, _>>()?;
        let raw_inputs: Vec<sys::oar_input> = prepared
            .iter()
            .map(|p| sys::oar_input {
                name: p.name.as_ptr(),
                data: p.data.as_ptr(),
                dims: p.dims.as_ptr(),
                rank: p.dims.len() as i32,
                reserved: 0,
            })
            .collect();

        const MAX_OUT: usize = 16;
        let blank = sys::oar_tensor {
            rank: 0,
            dims: [0; 8],
            data: std::ptr::null_mut(),
            name: [0; 64],
            dtype: 0,
            reserved: 0,
            data_i64: std::ptr::null_mut(),
        };
        let mut outs = vec![blank; MAX_OUT];
        let mut n_out = 0i32;
        // SAFETY: raw_inputs borrows from `prepared`, which outlives the call; outs has MAX_OUT entries.
        let status = unsafe {
            sys::oar_engine_run_named(
                self.handle.0.as_ptr(),
                raw_inputs.as_ptr(),
                raw_inputs.len() as i32,
                outs.as_mut_ptr(),
                MAX_OUT as i32,
                &mut n_out,
            )
        };
        check(status).map_err(|e| self.wrap(e, self.error_context(inputs)))?;

        let mut result = Vec::with_capacity(n_out.max(0) as usize);
        for t in outs.iter_mut().take(n_out.max(0) as usize) {
            let shape: Vec<i64> = t.dims[..t.rank.clamp(0, 8) as usize].to_vec();
            let count: usize = shape.iter().map(|&d| d.max(0) as usize).product();
            let name_bytes: Vec<u8> = t.name.iter().take_while(|&&c| c != 0).map(|&c| c as u8).collect();
            let name = String::from_utf8_lossy(&name_bytes).into_owned();
            // SAFETY: the tensor owns `count` elements of its dtype (oar_tensor).
            let out = unsafe {
                if t.dtype == sys::OAR_DTYPE_I64 {
                    TensorOutput::I64 { shape, data: slice_or_empty(t.data_i64, count).to_vec() }
                } else {
                    TensorOutput::F32 { shape, data: slice_or_empty(t.data, count).to_vec() }
                }
            };
            // SAFETY: t was filled by oar_engine_run_named; freed exactly once.
            unsafe { sys::oar_tensor_free(t) };
            result.push((name, out));
        }
        Ok(result)
    }

    /// `OrtInfer::infer_first_output_f32` (ort_infer_execution.rs:234-306): `f` sees `(shape, data)` of the first
    /// output as a borrowed row-major slice (a pinned staging buffer inside the engine) and returns a compact owned
    /// result; nothing of the logits tensor is copied into Rust-owned memory.
    pub fn infer_first_output_f32<R>(
        &self,
        inputs: &[(&str, TensorInput)],
        f: impl FnOnce(&[usize], &[f32]) -> Result<R, OCRError>,
    ) -> Result<R, OCRError> {
        if inputs.is_empty() {
            return Err(OCRError::InvalidInput { message: "No inputs provided for inference".to_string() });
        }
        let prepared: Vec<PreparedInput> = inputs.iter().map(|(n, t)| prepare(n, t)).collect::<Result<_, _>>()?;
        let raw_inputs: Vec<sys::oar_input> = prepared
            .iter()
            .map(|p| sys::oar_input {
                name: p.name.as_ptr(),
                data: p.data.as_ptr(),
                dims: p.dims.as_ptr(),
                rank: p.dims.len() as i32,
                reserved: 0,
            })
            .collect();

        // the closure and its result travel through the C callback's `user` pointer
        struct Slot<F, R> {
            f: Option<F>,
            result: Option<Result<R, OCRError>>,
        }
        unsafe extern "C" fn trampoline<F, R>(user: *mut c_void, dims: *const i64, rank: i32, data: *const f32) -> i32
        where
            F: FnOnce(&[usize], &[f32]) -> Result<R, OCRError>,
        {
            // SAFETY: `user` is the &mut Slot passed below; dims has `rank` entries; data has prod(dims) floats and is
            // valid until this function returns (include/oar_mi355x.h, oar_output_view_fn).
            let slot = unsafe { &mut *(user as *mut Slot<F, R>) };
            let shape: Vec<usize> = unsafe { slice_or_empty(dims, rank.max(0) as usize) }.iter().map(|&d| d.max(0) as usize).collect();
            let count: usize = shape.iter().product();
            let view = unsafe { slice_or_empty(data, count) };
            let Some(f) = slot.f.take() else { return 2 };
            // a panic must not unwind into the C frame
            match std::panic::catch_unwind(std::panic::AssertUnwindSafe(|| f(&shape, view))) {
                Ok(r) => {
                    let failed = r.is_err();
                    slot.result = Some(r);
                    i32::from(failed)
                }
                Err(_) => 3,
            }
        }

        fn run<F, R>(this: &Mi355xInfer, raw_inputs: &[sys::oar_input], f: F) -> (sys::oar_status, Option<Result<R, OCRError>>)
        where
            F: FnOnce(&[usize], &[f32]) -> Result<R, OCRError>,
        {
            let mut slot: Slot<F, R> = Slot { f: Some(f), result: None };
            // SAFETY: raw_inputs is valid for the call; `slot` outlives it; the trampoline matches oar_output_view_fn.
            let status = unsafe {
                sys::oar_engine_run_first_f32(
                    this.handle.0.as_ptr(),
                    raw_inputs.as_ptr(),
                    raw_inputs.len() as i32,
                    Some(trampoline::<F, R> as unsafe extern "C" fn(*mut c_void, *const i64, i32, *const f32) -> i32),
                    (&mut slot as *mut Slot<F, R>).cast(),
                )
            };
            (status, slot.result)
        }

        let (status, result) = run(self, &raw_inputs, f);
        match result {
            Some(r) => r, // the closure ran: its own Ok / Err wins (a closure error makes the C call report failure too)
            None => {
                check(status).map_err(|e| self.wrap(e, self.error_context(inputs)))?;
                Err(OCRError::InvalidInput {
                    message: format!("Model '{}': the output view callback was never invoked", self.model_name),
                })
            }
        }
    }
}
