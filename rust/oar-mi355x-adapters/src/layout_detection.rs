//! `LayoutDetectionAdapter` (oar-ocr-core/src/domain/adapters/layout_detection_adapter.rs) for the PicoDet / RT-DETR layout
//! families, with the model half on the GPU (`oar_layout_*`).
//!
//! The reference adapter = `ScaleAwareDetectorModel::forward` (resize to the model's `image_shape` with its filter, normalise,
//! graph with `image` + `scale_factor` [+ `im_shape`]) -> `LayoutPostProcess::apply` (row parsing, score filter, coordinate
//! conversion, class-aware NMS) -> the adapter's own configuration pass (`layout_unclip_ratio`, class labels, per-class
//! thresholds, `max_elements`; layout_detection_adapter.rs:540-629).  Everything up to `LayoutPostProcess`'s output is one C call
//! whose stages are HIP kernels; the configuration pass stays here, in the reference's own words (`unclip_boxes` is the
//! reference's function, called as the reference calls it).
//!
//! Round 4: a `LayoutModelConfig` of type "pp-doclayout" takes the adapter's own PaddleX-style post-processing
//! (`postprocess_pp_doclayout`, :631-846: per-class thresholds, `paddlex_layout_nms`, `filter_large_image_boxes`,
//! `apply_paddlex_merge_modes`, reading-order sort) as ONE HIP kernel behind `oar_layout_run_ppdoc`; `class_merge_modes` of the
//! other families (`apply_nms_with_merge`, processors/layout_postprocess.rs:743-841) goes through `oar_host_nms_with_merge`.
//! Source-only, never compiled here (no Rust toolchain in the backend's build image): checked lexically by
//! tests/test_rust_bindings_cpu.py.

use crate::error::{Mi355xError, check};
use crate::ffi_util::{ImageBatch, device_id_from_ort_config, model_bytes, slice_or_empty};
use oar_mi355x_sys as sys;
use oar_ocr_core::core::OCRError;
use oar_ocr_core::core::config::OrtSessionConfig;
use oar_ocr_core::core::inference::ModelSource;
use oar_ocr_core::core::traits::adapter::{AdapterBuilder, AdapterInfo, ModelAdapter, OrtConfigurable};
use oar_ocr_core::core::traits::task::{Task, TaskType};
use oar_ocr_core::domain::adapters::LayoutModelConfig;
use oar_ocr_core::domain::tasks::{LayoutDetectionConfig, LayoutDetectionElement, LayoutDetectionOutput, LayoutDetectionTask, MergeBboxMode, UnclipRatio};
use oar_ocr_core::processors::{BoundingBox, unclip_boxes};
use std::ptr::NonNull;

/// Owning handle of an `oar_layout`.
#[derive(Debug)]
pub(crate) struct LayoutHandle(pub(crate) NonNull<sys::oar_layout>);

// SAFETY: handles may be used from any thread; calls on one handle serialise on an internal mutex (include/oar_mi355x.h).
unsafe impl Send for LayoutHandle {}
unsafe impl Sync for LayoutHandle {}

impl Drop for LayoutHandle {
    fn drop(&mut self) {
        // SAFETY: the pointer came from oar_layout_create and is destroyed exactly once.
        unsafe { sys::oar_layout_destroy(self.0.as_ptr()) }
    }
}

struct LayoutResultGuard(sys::oar_layout_result);

impl Drop for LayoutResultGuard {
    fn drop(&mut self) {
        // SAFETY: filled by oar_layout_run or all-NULL.
        unsafe { sys::oar_layout_result_free(&mut self.0) }
    }
}

/// `LayoutDetectionAdapter` with resize, normalisation, the detector graph and `LayoutPostProcess` on the GPU.
#[derive(Debug)]
pub struct Mi355xLayoutDetectionAdapter {
    handle: LayoutHandle,
    info: AdapterInfo,
    model_config: LayoutModelConfig,
    config: LayoutDetectionConfig,
}

impl ModelAdapter for Mi355xLayoutDetectionAdapter {
    type Task = LayoutDetectionTask;

    fn info(&self) -> AdapterInfo {
        self.info.clone()
    }

    fn execute(
        &self,
        input: <Self::Task as Task>::Input,
        config: Option<&<Self::Task as Task>::Config>,
    ) -> Result<<Self::Task as Task>::Output, OCRError> {
        let effective_config = config.unwrap_or(&self.config);
        let batch_len = input.images.len();
        let images: Vec<&image::RgbImage> = input.images.iter().map(AsRef::as_ref).collect();
        let batch = ImageBatch::new(images.iter().copied());
        let mut result = LayoutResultGuard(sys::oar_layout_result {
            n_images: 0,
            n_boxes: 0,
            box_offsets: std::ptr::null_mut(),
            boxes: std::ptr::null_mut(),
            classes: std::ptr::null_mut(),
            scores: std::ptr::null_mut(),
            feature_dim: 0,
        });
        let is_ppdoc = self.model_config.model_type == "pp-doclayout";
        // label-keyed maps -> class-id-indexed arrays, as postprocess_pp_doclayout builds them (:645-675)
        let num_classes = self.model_config.num_classes;
        let class_thresholds: Option<Vec<f32>> = effective_config.class_thresholds.as_ref().map(|t| {
            (0..num_classes).map(|c| self.model_config.class_labels.get(&c).and_then(|l| t.get(l)).copied().unwrap_or(f32::NAN)).collect()
        });
        let merge_code = |m: MergeBboxMode| match m { MergeBboxMode::Large => 0i32, MergeBboxMode::Union => 1, MergeBboxMode::Small => 2 };
        let class_merge_modes: Option<Vec<i32>> = effective_config.class_merge_modes.as_ref().map(|m| {
            (0..num_classes).map(|c| self.model_config.class_labels.get(&c).and_then(|l| m.get(l)).map(|v| merge_code(*v)).unwrap_or(-1)).collect()
        });
        let label_id = |name: &str| self.model_config.class_labels.iter().find_map(|(id, l)| (l == name).then_some(*id as i32)).unwrap_or(-1);
        let ppdoc_cfg = sys::oar_ppdoc_cfg {
            score_threshold: effective_config.score_threshold,
            class_thresholds: class_thresholds.as_ref().map_or(std::ptr::null(), |v| v.as_ptr()),
            layout_nms: effective_config.layout_nms as i32,
            image_class_id: label_id("image"),
            formula_class_id: label_id("formula"),
            class_merge_modes: if is_ppdoc { class_merge_modes.as_ref().map_or(std::ptr::null(), |v| v.as_ptr()) } else { std::ptr::null() },
        };
        // SAFETY: three arrays of batch.len() entries; page buffers and the configuration arrays outlive the call; result is a valid out-parameter.
        let status = unsafe {
            if is_ppdoc {
                sys::oar_layout_run_ppdoc(self.handle.0.as_ptr(), batch.ptrs.as_ptr(), batch.widths.as_ptr(), batch.heights.as_ptr(), batch.len() as u32, &ppdoc_cfg, &mut result.0)
            } else {
                sys::oar_layout_run(
                    self.handle.0.as_ptr(),
                    batch.ptrs.as_ptr(),
                    batch.widths.as_ptr(),
                    batch.heights.as_ptr(),
                    batch.len() as u32,
                    &mut result.0,
                )
            }
        };
        check(status).map_err(|e| {
            e.into_adapter_error("LayoutDetectionAdapter", format!("PicoDet forward (batch_size={batch_len})"))
        })?;

        let r = &result.0;
        let (n, nb) = (r.n_images as usize, r.n_boxes as usize);
        // SAFETY: lengths as documented for oar_layout_result.
        let (offsets, boxes, classes, scores) = unsafe {
            (slice_or_empty(r.box_offsets, n + 1), slice_or_empty(r.boxes, nb * 4), slice_or_empty(r.classes, nb), slice_or_empty(r.scores, nb))
        };
        // the adapter's configuration pass (layout_detection_adapter.rs:552-629)
        let mut elements = Vec::with_capacity(n);
        for i in 0..n {
            let (lo, hi) = (offsets[i] as usize, offsets[i + 1] as usize);
            let mut img_boxes: Vec<BoundingBox> =
                (lo..hi).map(|b| BoundingBox::from_coords(boxes[b * 4], boxes[b * 4 + 1], boxes[b * 4 + 2], boxes[b * 4 + 3])).collect();
            let mut img_classes: Vec<usize> = (lo..hi).map(|b| classes[b].max(0) as usize).collect();
            let mut img_scores: Vec<f32> = (lo..hi).map(|b| scores[b]).collect();
            if let Some(ref unclip_ratio) = effective_config.layout_unclip_ratio {
                let (width_ratio, height_ratio, per_class_ratios) = match unclip_ratio {
                    UnclipRatio::Uniform(r) => (*r, *r, None),
                    UnclipRatio::Separate(w, h) => (*w, *h, None),
                    UnclipRatio::PerClass(ratios) => (1.0, 1.0, Some(ratios)),
                };
                img_boxes = unclip_boxes(&img_boxes, &img_classes, width_ratio, height_ratio, per_class_ratios);
            }
            // class_merge_modes of the PicoDet / RT-DETR families: apply_nms_with_merge (:577-587) on this image's boxes
            if let (false, Some(modes)) = (is_ppdoc, effective_config.class_merge_modes.as_ref()) {
                let mode_of_class: Vec<i32> = (0..num_classes)
                    // a class id without a label is looked up as "unknown", as the reference does (layout_postprocess.rs:771-779) and api.py's .get(c, "unknown")
                    .map(|c| {
                        let label = self.model_config.class_labels.get(&c).map(|s| s.as_str()).unwrap_or("unknown");
                        modes.get(label).map(|v| merge_code(*v)).unwrap_or(0)
                    })
                    .collect();
                let flat: Vec<f32> = img_boxes.iter().flat_map(|b| { let (x0, y0, x1, y1) = b.aabb(); [x0, y0, x1, y1] }).collect();
                let cls32: Vec<i32> = img_classes.iter().map(|&c| c as i32).collect();
                let k = img_boxes.len();
                let (mut ob, mut oc, mut os) = (vec![0f32; k * 4], vec![0i32; k], vec![0f32; k]);
                // SAFETY: input arrays hold k entries, output arrays have room for k
                let m = unsafe {
                    sys::oar_host_nms_with_merge(flat.as_ptr(), cls32.as_ptr(), img_scores.as_ptr(), k as u32, mode_of_class.as_ptr(), num_classes as u32,
                                                 effective_config.nms_threshold, effective_config.max_elements as u32, ob.as_mut_ptr(), oc.as_mut_ptr(), os.as_mut_ptr())
                };
                if m < 0 {
                    return Err(OCRError::InvalidInput { message: "oar_host_nms_with_merge failed".to_string() });
                }
                let m = m as usize;
                img_boxes = (0..m).map(|b| BoundingBox::from_coords(ob[b * 4], ob[b * 4 + 1], ob[b * 4 + 2], ob[b * 4 + 3])).collect();
                img_classes = oc[..m].iter().map(|&c| c.max(0) as usize).collect();
                img_scores = os[..m].to_vec();
            }
            let mut img_elements = Vec::new();
            for (k, bbox) in img_boxes.iter().enumerate() {
                let score = img_scores[k];
                let element_type =
                    self.model_config.class_labels.get(&img_classes[k]).cloned().unwrap_or_else(|| "unknown".to_string());
                // PP-DocLayout applied its per-class thresholds inside the kernel (:700-706) and has no second filter here (:813-831)
                if is_ppdoc || score >= effective_config.get_class_threshold(&element_type) {
                    img_elements.push(LayoutDetectionElement { bbox: bbox.clone(), element_type, score });
                    if img_elements.len() >= effective_config.max_elements {
                        break;
                    }
                }
            }
            elements.push(img_elements);
        }
        // :1186-1192: 7 / 8 prediction columns carry the model's reading order
        Ok(LayoutDetectionOutput { elements, is_reading_order_sorted: r.feature_dim == 7 || r.feature_dim == 8 })
    }

    fn supports_batching(&self) -> bool {
        true
    }

    fn recommended_batch_size(&self) -> usize {
        4 // layout_detection_adapter.rs:1203-1205 (the library runs sub-batches of 8 itself)
    }
}

/// Builder with the surface of `LayoutDetectionAdapterBuilder` (layout_detection_adapter.rs:1210-1383).
#[derive(Debug, Clone)]
pub struct Mi355xLayoutDetectionAdapterBuilder {
    config: LayoutDetectionConfig,
    model_config: Option<LayoutModelConfig>,
    device_id: i32,
}

impl Default for Mi355xLayoutDetectionAdapterBuilder {
    fn default() -> Self {
        Self::new()
    }
}

impl Mi355xLayoutDetectionAdapterBuilder {
    pub fn new() -> Self {
        Self { config: LayoutDetectionConfig::default(), model_config: None, device_id: 0 }
    }

    /// `LayoutDetectionAdapterBuilder::model_config` (:1222-1225)
    pub fn model_config(mut self, config: LayoutModelConfig) -> Self {
        self.model_config = Some(config);
        self
    }

    /// `task_config` (:1228-1231)
    pub fn task_config(mut self, config: LayoutDetectionConfig) -> Self {
        self.config = config;
        self
    }

    pub fn score_threshold(mut self, threshold: f32) -> Self {
        self.config.score_threshold = threshold;
        self
    }

    pub fn max_elements(mut self, max: usize) -> Self {
        self.config.max_elements = max;
        self
    }

    pub fn device_id(mut self, device_id: i32) -> Self {
        self.device_id = device_id;
        self
    }
}

impl AdapterBuilder for Mi355xLayoutDetectionAdapterBuilder {
    type Config = LayoutDetectionConfig;
    type Adapter = Mi355xLayoutDetectionAdapter;

    fn build(self, model_source: impl Into<ModelSource>) -> Result<Self::Adapter, OCRError> {
        let model_config = self.model_config.unwrap_or_else(LayoutModelConfig::picodet_layout_1x);
        // ScaleAwareDetectorPreprocessConfig of the family (scale_aware_detector.rs:49-75) and the processor's model type
        let (model_type, resize_filter, color_bgr, mean, std) = match model_config.model_type.as_str() {
            "picodet" => (0, 2, 1, [0.485f32, 0.456, 0.406], [0.229f32, 0.224, 0.225]),
            "rtdetr" => (1, 2, 1, [0.485f32, 0.456, 0.406], [0.229f32, 0.224, 0.225]),
            // PPDocLayoutModel: CatmullRom resize, RGB tensor, no mean / std (scale_aware_detector.rs:62-75)
            "pp-doclayout" => (2, 1, 0, [0.0f32, 0.0, 0.0], [1.0f32, 1.0, 1.0]),
            other => {
                return Err(OCRError::InvalidInput {
                    message: format!("Mi355xLayoutDetectionAdapter: model type '{other}' is not carried by this backend (picodet, rtdetr, pp-doclayout are)"),
                });
            }
        };
        let (input_h, input_w) = model_config.input_size.unwrap_or((800, 608));
        let source: ModelSource = model_source.into();
        let (bytes, shown) = model_bytes(&source)?;
        let cfg = sys::oar_layout_cfg {
            device_id: self.device_id,
            input_h,
            input_w,
            resize_filter,
            color_bgr,
            scale: 1.0 / 255.0,
            mean,
            std,
            num_classes: model_config.num_classes as u32,
            model_type,
            score_threshold: self.config.score_threshold,
            nms_threshold: self.config.nms_threshold,
            max_detections: self.config.max_elements as u32,
        };
        let mut h: *mut sys::oar_layout = std::ptr::null_mut();
        // SAFETY: bytes is valid for bytes.len(); cfg and h are valid for the call.
        let status = unsafe { sys::oar_layout_create(bytes.as_ptr(), bytes.len(), &cfg, &mut h) };
        check(status).map_err(|e: Mi355xError| e.into_model_load(&shown))?;
        let handle = LayoutHandle(NonNull::new(h).ok_or_else(|| OCRError::ConfigError {
            message: "oar_layout_create returned a null handle".to_string(),
        })?);
        let info = AdapterInfo::new(
            format!("LayoutDetection_{}", model_config.model_name),
            TaskType::LayoutDetection,
            format!("Layout detection adapter for {} with {} classes", model_config.model_name, model_config.num_classes),
        );
        Ok(Mi355xLayoutDetectionAdapter { handle, info, model_config, config: self.config })
    }

    fn with_config(mut self, config: Self::Config) -> Self {
        self.config = config;
        self
    }

    fn adapter_type(&self) -> &str {
        "LayoutDetection"
    }
}

/// `OrtConfigurable` (core/traits/adapter.rs:126-129; layout_detection_adapter.rs:1385-1390): only the device ordinal applies.
impl OrtConfigurable for Mi355xLayoutDetectionAdapterBuilder {
    fn with_ort_config(mut self, config: OrtSessionConfig) -> Self {
        if let Some(device_id) = device_id_from_ort_config(&config) {
            self.device_id = device_id;
        }
        self
    }
}
