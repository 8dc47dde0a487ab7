/*
 * oar_mi355x.h -- C ABI of libOarMi355x.so: the MI355X-native (gfx950) drop-in for the det+rec hot path
 * of GreatV/oar-ocr.  Plain pointers and sizes only; no C++/torch types cross this boundary.
 *
 * Every entry point cites the reference interface it replaces (file:line under the reference tree).
 * Two seams are exported:
 *   Seam A (narrow) -- replaces `OrtInfer` (oar-ocr-core/src/core/inference/ort_infer_execution.rs:121-306):
 *       .onnx bytes in, f32 tensors in/out.
 *   Seam B (wide)   -- replaces `ModelAdapter::execute` for text detection / recognition
 *       (oar-ocr-core/src/core/traits/adapter.rs:42-81) and `OAROCR::predict` (src/oarocr/ocr.rs:518-659),
 *       so pre/post-processing also runs on the GPU.
 *
 * Conventions: every function returns an oar_status (0 = ok); on failure a thread-local message is
 * available through oar_last_error().  Nothing throws or aborts across the ABI.  Inputs are borrowed for
 * the duration of the call only; outputs are library-allocated and released with the matching *_free.
 * Handles may be used from any thread; calls on one handle serialise on an internal mutex (the reference
 * serialises on `Mutex<Session>`, core/inference/mod.rs:31-37).
 */
#ifndef OAR_MI355X_H
#define OAR_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    OAR_OK = 0,
    OAR_INVALID_INPUT = 1,   /* OCRError::InvalidInput / validation_error        (core/errors/types.rs:112-214) */
    OAR_MODEL_LOAD = 2,      /* OCRError::ModelLoad                                                             */
    OAR_UNSUPPORTED_OP = 3,  /* graph contains an operator the engine does not implement                        */
    OAR_SHAPE_MISMATCH = 4,  /* OCRError::Tensor                                                                */
    OAR_DEVICE = 5,          /* HIP runtime failure / no gfx950 device (the library never falls back to CPU)    */
    OAR_OOM = 6,
    OAR_INTERNAL = 7
} oar_status;

/* Copies the calling thread's last error message (NUL terminated, truncated to cap). Returns its length. */
size_t oar_last_error(char* buf, size_t cap);
/* Library / device identification: "libOarMi355x <ver> gfx950 <device name> CUs=<n>". */
size_t oar_version(char* buf, size_t cap);
/* Number of visible HIP devices (0 => every create call fails with OAR_DEVICE). */
int oar_device_count(void);

/* ------------------------------------------------------------------------------------------------ Seam A
 * OrtInfer::new / from_config (core/inference/ort_infer_builders.rs:9-70) -> oar_engine_create
 * OrtInfer::infer               (core/inference/ort_infer_execution.rs:121-219) -> oar_engine_run
 * OrtInfer::input_name / primary_input_shape (core/inference/mod.rs:52-115) -> oar_engine_input_name
 */
typedef struct oar_engine oar_engine;

/* Arithmetic mode of the network kernels (SURVEY 8b "Device selection": OarEngineCfg{device_id, precision, stream}; the reference's seam is
 * OrtSessionConfig, core/config/onnx.rs:88-178).  OAR_PRECISION_F32 is the only mode the library implements: every product is either an f32
 * FMA / f32-input MFMA or "bf16x6" -- both operands split exactly into three bf16 pieces, the six significant products accumulated in f32,
 * relative error of a product ~2^-24 like an f32 multiply.  Narrower modes (bf16x3, 16-bit activations) would not be the reference's arithmetic;
 * any other value is refused with OAR_INVALID_INPUT instead of being silently widened or narrowed. */
typedef enum { OAR_PRECISION_F32 = 0 } oar_precision;

typedef struct {
    int32_t device_id;      /* HIP device ordinal                                                    */
    int32_t use_hip_graph;  /* 1: capture each (shape-specialised) plan into a hipGraph and replay    */
    int32_t profile;        /* 1: record hipEvents around kernels (see oar_prof_*)                   */
    int32_t precision;      /* oar_precision; 0 = OAR_PRECISION_F32 (was `reserved`, always 0)       */
    void* stream;           /* hipStream_t of the caller on `device_id`, or NULL: the engine creates its own non-blocking stream.  With a caller's
                             * stream every copy and kernel of oar_engine_run is enqueued on it (in order with the caller's own work) and the call
                             * still returns after the outputs have reached the host.  The stream must outlive the engine; the engine never
                             * destroys it.                                                          */
} oar_engine_cfg;

/* TensorOutput::{F32, I64} (core/inference/tensor_output.rs:16-21) */
typedef enum { OAR_DTYPE_F32 = 1, OAR_DTYPE_I64 = 7 } oar_dtype;
typedef struct {
    int32_t rank;
    int64_t dims[8];
    float* data;            /* dtype F32: host memory owned by the library; free with oar_tensor_free */
    char name[64];
    int32_t dtype;          /* oar_dtype                                                             */
    int32_t reserved;
    int64_t* data_i64;      /* dtype I64 (token ids, ArgMax, shapes): data is NULL, this holds the values */
} oar_tensor;

/* One named f32 input of OrtInfer::infer (`&[(&str, TensorInput)]`, ort_infer_execution.rs:12-19, 121-135): Array2 / Array3 /
 * Array4 are rank 2 / 3 / 4 here.  name NULL or "" = the graph's primary input. */
typedef struct {
    const char* name;
    const float* data;      /* row-major, host memory                                               */
    const int64_t* dims;
    int32_t rank;
    int32_t reserved;
} oar_input;

/* Declared name / element type / shape of one graph input or output (dynamic dimensions are -1), as
 * OrtInfer::input_names_from_model / primary_input_shape / output_shapes report them (core/inference/mod.rs:66-112). */
typedef struct {
    char name[64];
    int32_t dtype;          /* onnx TensorProto.DataType as declared by the model file (1 = f32, 7 = i64, 0 = undeclared) */
    int32_t rank;           /* -1: the model file declares no shape                                  */
    int64_t dims[8];
} oar_io_info;

/* Borrowed view of the first output (OrtInfer::infer_first_output_f32's `FnOnce(&[i64], &[f32])`, ort_infer_execution.rs:234-306):
 * `data` points into a pinned staging buffer owned by the engine and is valid only until the callback returns.
 * A non-zero return value aborts with OAR_INVALID_INPUT and is reported through oar_last_error.  The callback runs under the
 * engine's lock (like the reference's closure under the session lock): it must not call back into the same engine. */
typedef int32_t (*oar_output_view_fn)(void* user, const int64_t* dims, int32_t rank, const float* data);

oar_status oar_engine_create(const uint8_t* onnx, size_t onnx_len, const oar_engine_cfg* cfg, oar_engine** out);
void oar_engine_destroy(oar_engine* e);
/* Name of the graph's first non-initializer input (DB/CRNN use "x": models/detection/db.rs:388-390). */
oar_status oar_engine_input_name(const oar_engine* e, char* buf, size_t cap);
/* One f32 input (row-major, ONNX/NCHW semantics), all graph outputs copied back to host.
 * outs must have room for max_out entries; *n_out receives the count. */
oar_status oar_engine_run(oar_engine* e, const float* input, const int64_t* dims, int32_t rank,
                          oar_tensor* outs, int32_t max_out, int32_t* n_out);
void oar_tensor_free(oar_tensor* t);
/* OrtInfer::infer (ort_infer_execution.rs:121-219): n_in named f32 inputs -- every declared graph input must be given exactly
 * once (OAR_INVALID_INPUT otherwise, naming the missing / unknown one) -- all graph outputs back, F32 or I64. */
oar_status oar_engine_run_named(oar_engine* e, const oar_input* inputs, int32_t n_in, oar_tensor* outs, int32_t max_out,
                                int32_t* n_out);
/* OrtInfer::infer_first_output_f32 (ort_infer_execution.rs:234-306): runs the graph and hands the FIRST output to `view`
 * without an owned copy.  OAR_SHAPE_MISMATCH when that output is not f32. */
oar_status oar_engine_run_first_f32(oar_engine* e, const oar_input* inputs, int32_t n_in, oar_output_view_fn view, void* user);
/* Declared inputs (initializers excluded; entry 0 is the primary input) and outputs of the model file.  Either array may
 * be NULL to query the counts only. */
oar_status oar_engine_io(const oar_engine* e, oar_io_info* inputs, int32_t max_in, int32_t* n_in, oar_io_info* outputs,
                         int32_t max_out, int32_t* n_out);
/* Analytic cost of the plan for a given input shape (what roofline.achieved is computed from):
 * flops = sum 2*MACs over conv/matmul steps; bytes = sum (activations in + out + weights once). */
oar_status oar_engine_cost(oar_engine* e, const int64_t* dims, int32_t rank, double* flops, double* bytes,
                           int32_t* n_kernels);

/* Plan-cache statistics: plans are specialised per input shape and kept in an LRU of OAR_PLAN_CACHE (default 256)
 * entries per engine, so a long-running server on heterogeneous pages has bounded host memory. */
oar_status oar_engine_cache_stats(oar_engine* e, uint64_t* cached_plans, uint64_t* evicted_plans);
/* Host-only model check (no GPU needed): parses the file exactly as oar_engine_create does and writes a one-line
 * summary "opset=.. input=.. nodes=.. | Op:count ..." (operators the engine does not implement are prefixed '!').
 * OAR_MODEL_LOAD for a malformed / truncated file (dims, payload sizes and ranks are validated before anything is
 * indexed), OAR_UNSUPPORTED_OP when an operator is missing.  Stands where `Session::builder().commit_from_file`
 * fails in the reference (core/inference/session.rs:30-44). */
oar_status oar_onnx_inspect(const uint8_t* onnx, size_t onnx_len, char* summary, size_t cap);

/* ------------------------------------------------------------------------------------------------ Seam B: detection
 * TextDetectionAdapter::execute (domain/adapters/text_detection_adapter.rs:36-79) -> DBModel::forward
 * (models/detection/db.rs:281-335): resize (processors/resize_detection.rs:243-319) -> normalize
 * (processors/normalization.rs:429-482) -> network -> DBPostProcess (processors/db_postprocess.rs:100-183).
 */
typedef struct oar_det oar_det;

typedef struct {
    int32_t device_id;
    uint32_t limit_side_len;   /* default 960  (core/constants.rs:15)                                   */
    int32_t limit_type;        /* 0 = Max, 1 = Min, 2 = ResizeLong (processors/types.rs LimitType)      */
    uint32_t max_side_limit;   /* default 4000 (core/constants.rs:11)                                   */
    uint32_t max_candidates;   /* default 1000 (processors/db_postprocess.rs:79)                        */
    int32_t use_hip_graph;
    int32_t profile;
    int32_t host_threads;      /* threads of the contour / geometry pool, the caller included (0 = the CPUs this process may use --
                                * affinity mask and container quota --, at most 16, and at most the affinity mask minus 2 when it has
                                * >= 6 CPUs: a call also runs one uploader and one enqueuer thread)                                  */
    /* DBPostProcess options the adapter exposes (processors/db_postprocess.rs:60-98, processors/types.rs):          */
    int32_t box_type;          /* 0 = BoxType::Quad (default), 1 = BoxType::Poly (seal / curved text: polygons_from_bitmap, db_bitmap.rs:16-82;
                                  results carry point_offsets) */
    int32_t score_mode;        /* 0 = ScoreMode::Fast (mini-box scanline mean), 1 = ScoreMode::Slow (contour scanline mean, db_score.rs:139-181) */
    int32_t use_dilation;      /* 1: dilate the mask (3 x 3, db_mask.rs:11) before contour tracing (db_postprocess.rs:163-168) */
    /* Where find_contours (db_bitmap.rs:100) runs.  0: on the host thread pool from the read-back mask (default: fastest on one GPU,
     * the host cores are otherwise idle).  1: on the GPU (contours.hip, one wavefront per mask segment), only the border chains
     * cross PCIe and the host keeps the per-contour geometry -- for hosts whose cores are shared by many GPU ranks.  With it the
     * mini boxes are also unclipped on the GPU (pp::unclip_quads, in the box-score round trip).  Identical results either way.
     * The environment variables OAR_GPU_CONTOURS=0|1 and OAR_GPU_UNCLIP=0|1 override this field. */
    int32_t gpu_contours;
} oar_det_cfg;

/* CSR result: image i owns boxes [box_offsets[i], box_offsets[i+1]); each box is 4 points (x,y) f32 in
 * original-image coordinates, contour discovery order (unsorted, as the adapter returns them).
 * BoxType::Poly: a box is a polygon of any size -- box b owns points [point_offsets[b], point_offsets[b+1]) of `points`
 * (BoundingBox::points of polygons_from_bitmap, db_bitmap.rs:66-78); point_offsets is NULL for quads. */
typedef struct {
    uint32_t n_images;
    uint32_t n_boxes;
    uint32_t* box_offsets;   /* n_images + 1 */
    float* points;           /* quads: n_boxes * 8; polygons: n_points * 2 */
    float* scores;           /* n_boxes      */
    uint32_t n_points;       /* quads: n_boxes * 4 */
    uint32_t* point_offsets; /* polygons: n_boxes + 1; quads: NULL */
} oar_det_result;

oar_status oar_det_create(const uint8_t* onnx, size_t onnx_len, const oar_det_cfg* cfg, oar_det** out);
void oar_det_destroy(oar_det* d);
/* images: n tightly packed RGB8 (HWC) host buffers. thresh/box_thresh/unclip = TextDetectionConfig
 * {score_threshold, box_threshold, unclip_ratio} (domain/tasks/text_detection.rs:34-66). */
oar_status oar_det_run(oar_det* d, const uint8_t* const* rgb, const uint32_t* widths, const uint32_t* heights,
                       uint32_t n_images, float thresh, float box_thresh, float unclip_ratio, oar_det_result* out);
void oar_det_result_free(oar_det_result* r);
/* Test hook (parity of a7..a12 in isolation): run only DB post-processing on a host probability map. */
oar_status oar_db_postprocess(const float* pred, uint32_t height, uint32_t width, uint32_t src_w, uint32_t src_h,
                              float thresh, float box_thresh, float unclip_ratio, uint32_t max_candidates,
                              oar_det_result* out);
/* Same, with the DBPostProcess options that oar_det_cfg carries: box_type, score_mode, use_dilation. */
oar_status oar_db_postprocess_ex(const float* pred, uint32_t height, uint32_t width, uint32_t src_w, uint32_t src_h,
                                 float thresh, float box_thresh, float unclip_ratio, uint32_t max_candidates,
                                 int32_t box_type, int32_t score_mode, int32_t use_dilation, oar_det_result* out);

/* ------------------------------------------------------------------------------------------------ Seam B: recognition
 * TextRecognitionAdapter::execute (domain/adapters/text_recognition_adapter.rs:35-111) -> CRNNModel::forward_refs
 * (models/recognition/crnn.rs:247-293): resize+normalize (crnn.rs:71-125) -> network -> CTC argmax
 * (processors/decode.rs:452-501).  The dictionary / string assembly stays on the caller's side exactly as
 * decode.rs:505-614 (the result is the reference's CTCArgmaxOutput, decode.rs:28-33).
 */
typedef struct oar_rec oar_rec;

typedef struct {
    int32_t device_id;
    uint32_t rec_image_shape[3];  /* default {3,48,320} (core/constants.rs:21)  */
    uint32_t max_img_w;           /* default 3200       (core/constants.rs:8)   */
    int32_t use_hip_graph;
    int32_t profile;
    int32_t reserved;
} oar_rec_cfg;

typedef struct {
    uint32_t batch;
    uint32_t seq_len;        /* T                              */
    uint32_t vocab;          /* V (output last dim)            */
    uint32_t tensor_width;   /* Wt the batch was padded to     */
    int64_t* indices;        /* batch * T, argmax (last max index wins) */
    float* probs;            /* batch * T, max probability      */
} oar_rec_result;

oar_status oar_rec_create(const uint8_t* onnx, size_t onnx_len, const oar_rec_cfg* cfg, oar_rec** out);
void oar_rec_destroy(oar_rec* r);
oar_status oar_rec_run(oar_rec* r, const uint8_t* const* rgb, const uint32_t* widths, const uint32_t* heights,
                       uint32_t n_crops, oar_rec_result* out);
void oar_rec_result_free(oar_rec_result* r);

/* CTC collapse + text assembly + score filter on the host, inside the library (rows a19 / a20):
 * CTCLabelDecode::from_string_list(dict lines, use_space_char, has_explicit_blank = false) (processors/decode.rs:391-421;
 * lines from `char_dict.lines()`, src/oarocr/ocr.rs:386; only the first char of a line counts, empty lines vanish,
 * decode.rs:120) -> oar_ctc_dict_create;  decode_argmax_with_positions (decode.rs:549-614) + the adapter's
 * `score >= threshold` filter that blanks text / positions but keeps slot and score
 * (domain/adapters/text_recognition_adapter.rs:60-102) -> oar_ctc_decode / oar_ocr_decode.  No GPU involved. */
typedef struct oar_ctc_dict oar_ctc_dict;
typedef struct {
    uint32_t n;               /* sequences (regions)                                                   */
    uint64_t* text_offsets;   /* n + 1 byte offsets into utf8                                         */
    char* utf8;               /* concatenated texts, text_offsets[n] bytes (+ one NUL)                */
    float* scores;            /* n: mean probability of the kept characters, 0.0 when none            */
    uint64_t* char_offsets;   /* n + 1 offsets into char_cols / char_positions                        */
    uint32_t* char_cols;      /* char_col_indices: time step of each character                        */
    float* char_positions;    /* time step / T                                                        */
    uint32_t* seq_len;        /* n: T                                                                 */
    uint8_t* kept;            /* n: 0 when score < threshold (text and positions were blanked)        */
} oar_text_result;
/* dict_utf8: the dictionary file's text (UTF-8, one entry per line). */
oar_status oar_ctc_dict_create(const char* dict_utf8, size_t len, int32_t use_space_char, oar_ctc_dict** out);
void oar_ctc_dict_destroy(oar_ctc_dict* d);
uint32_t oar_ctc_dict_classes(const oar_ctc_dict* d);   /* blank + entries (+ space) */
/* indices / probs: batch * seq_len (an oar_rec_result). */
oar_status oar_ctc_decode(const oar_ctc_dict* dict, const int64_t* indices, const float* probs, uint32_t batch, uint32_t seq_len,
                          float score_threshold, oar_text_result* out);
void oar_text_result_free(oar_text_result* r);

/* ------------------------------------------------------------------------------------------------ Seam B: whole pipeline
 * OAROCRBuilder::new(det, rec, dict)...build() (src/oarocr/ocr.rs:105,249-417) -> oar_ocr_create
 * OAROCR::predict(Vec<RgbImage>)               (src/oarocr/ocr.rs:518-659)     -> oar_ocr_predict
 * Everything between the u8 pages and the per-region (box, CTC indices, probs) stays in HBM:
 * detect -> sort_quad_boxes (processors/sorting.rs:35-84) -> get_rotate_crop_image (utils/transform.rs:76-191)
 * -> wh-ratio pooled recognition batches (ocr.rs:802-897) -> CTC argmax.
 */
typedef struct oar_ocr oar_ocr;

typedef struct {
    oar_det_cfg det;
    oar_rec_cfg rec;
    float det_thresh;          /* builder default 0.3 (src/oarocr/ocr.rs:319-366)           */
    float det_box_thresh;      /* builder default 0.6                                         */
    float det_unclip_ratio;    /* builder default 2.0; explicit TextDetectionConfig: 1.5      */
    uint32_t image_batch_size; /* 0 => adapter recommended 8  (text_detection_adapter.rs:85-87)   */
    uint32_t region_batch_size;/* 0 => this backend's recommended 256 (reference adapter: 64, text_recognition_adapter.rs:117-127) */
    uint32_t max_pooled_crops; /* 0 => 4096 (src/oarocr/ocr.rs:603)                            */
    int32_t box_sort;          /* OAROCR::sort_detection_boxes keys on text_type, NOT on the detector's box type (ocr.rs:699-716):
                                * 0 = derive from det.box_type (Poly -> sort_poly_boxes, Quad -> sort_quad_boxes: what text_type "seal" /
                                * anything else gives when it configured both), 1 = sort_quad_boxes, 2 = sort_poly_boxes            */
    uint32_t lanes;            /* 0 / 1 => one pipeline; n (<= 8) => n complete pipelines behind the handle (own engines, streams,
                                * staging; the geometry threads are split between them) for oar_ocr_predict_async            */
} oar_ocr_cfg;

/* One entry per detected region, grouped per image in sorted (reading) order; regions whose crop failed
 * are dropped like the reference does (ocr.rs:736-738). */
typedef struct {
    uint32_t n_images;
    uint32_t n_regions;
    uint32_t* region_offsets;  /* n_images + 1                                              */
    float* points;             /* n_regions * 8, original-image coordinates (polygons: see point_offsets) */
    float* det_scores;         /* n_regions (not part of OAROCRResult; kept for parity checks) */
    uint32_t* crop_wh;         /* n_regions * 2 (w,h) of the rectified crop                 */
    uint32_t* seq_len;         /* n_regions: T of the batch the region was recognised in    */
    float* max_wh_ratio;       /* n_regions: chunk_max_wh_ratio (ocr.rs:828-831), for ctc_word_boxes */
    uint64_t* ctc_offsets;     /* n_regions + 1 into ctc_indices / ctc_probs                */
    int64_t* ctc_indices;
    float* ctc_probs;
    /* optional stages (oar_ocr_attach); filled with -1 / 0 when a stage is not attached */
    float* page_angle;         /* n_images: OAROCRResult::orientation_angle (0/90/180/270), -1 = none (ocr.rs:653)      */
    uint8_t* page_rectified;   /* n_images: 1 when rectified_img would be Some (ocr.rs:654); boxes then stay in rectified space */
    float* line_angle;         /* n_regions: TextRegion::orientation_angle (0/180), -1 = none (ocr.rs:782-783,888)      */
    /* det.box_type = 1 (seal text): a region's box is a polygon -- region r owns points [point_offsets[r], point_offsets[r+1]) of
     * `points` (then n_points * 2 floats); regions are in sort_poly_boxes order and cropped by their bounding rectangle
     * (bbox_crop.rs:26-72) unless they have exactly 4 points.  point_offsets is NULL for quads. */
    uint32_t n_points;         /* quads: n_regions * 4 */
    uint32_t* point_offsets;
} oar_ocr_result;

oar_status oar_ocr_create(const uint8_t* det_onnx, size_t det_len, const uint8_t* rec_onnx, size_t rec_len,
                          const oar_ocr_cfg* cfg, oar_ocr** out);
void oar_ocr_destroy(oar_ocr* o);
oar_status oar_ocr_predict(oar_ocr* o, const uint8_t* const* rgb, const uint32_t* widths, const uint32_t* heights,
                           uint32_t n_images, oar_ocr_result* out);
/* Same, with the pages already resident in HBM (device pointers from oar_dev_alloc/oar_dev_upload). */
oar_status oar_ocr_predict_device(oar_ocr* o, const uint8_t* const* d_rgb, const uint32_t* widths,
                                  const uint32_t* heights, uint32_t n_images, oar_ocr_result* out);
void oar_ocr_result_free(oar_ocr_result* r);
/* Calls in flight.  oar_ocr_predict_async queues one OAROCR::predict (same arguments; device_pages != 0: the pointers are device
 * memory, as for oar_ocr_predict_device) on the next lane of the handle and returns a ticket at once; oar_ocr_wait blocks until
 * that call has finished and hands over its result (or its error: status + oar_last_error).  With oar_ocr_cfg.lanes = 2, call
 * k + 1 uploads and detects while call k recognises -- the GPU time a single synchronous predict cannot fill (its recogniser
 * must wait for the last page's boxes: crops are pooled over the pages of one call, src/oarocr/ocr.rs:594-634).  Every call is
 * still one predict on one lane, so its result is exactly the synchronous one.  The page buffers must stay valid until the
 * ticket has been waited for; every ticket must be waited for exactly once.  Thread-safe. */
oar_status oar_ocr_predict_async(oar_ocr* o, const uint8_t* const* rgb, const uint32_t* widths, const uint32_t* heights, uint32_t n_images,
                                 int32_t device_pages, uint64_t* ticket);
oar_status oar_ocr_wait(oar_ocr* o, uint64_t ticket, oar_ocr_result* out);
/* Texts / scores / character columns of every region of a pipeline result, region order = res's
 * (OAROCR::recognize_global's scatter, src/oarocr/ocr.rs:840-891). */
oar_status oar_ocr_decode(const oar_ctc_dict* dict, const oar_ocr_result* res, float score_threshold, oar_text_result* out);

/* Word (character) boxes, row a21: OAROCR::ctc_word_boxes (src/oarocr/ocr.rs:949-1020) -- the line box's extent cut at the
 * CTC columns: effective columns = T * (wh_ratio / max_wh_ratio), cell = width / effective, centres (col + 0.5) * cell; a CJK
 * character (is_cjk, ocr.rs:1075-1082) gets an average-width box around its centre, any other character the span between the
 * midpoints to its neighbours; all clamped to the line.  A box = BoundingBox::from_coords = 4 points (x1,y1)(x2,y1)(x2,y2)(x1,y2)
 * = 8 floats.  f32 arithmetic in the reference's order.  boxes may be NULL (count query); cap_boxes in boxes of 8 floats. */
oar_status oar_ctc_word_boxes(const float* line_pts_xy, uint32_t n_points, const char* text_utf8, size_t text_len, const uint32_t* col_indices,
                              uint32_t n_cols, uint32_t seq_len, float wh_ratio, float max_wh_ratio, float* boxes, uint32_t cap_boxes, uint32_t* n_boxes);
/* OAROCR::char_positions_to_word_boxes (ocr.rs:1036-1072): the fallback when a recogniser reports positions but no columns */
oar_status oar_char_positions_to_word_boxes(const float* line_pts_xy, uint32_t n_points, const float* char_positions, uint32_t n_positions,
                                            uint32_t char_count, float* boxes, uint32_t cap_boxes, uint32_t* n_boxes);
/* return_word_box for a whole pipeline result (ocr.rs:860-877): region k owns boxes [box_offsets[k], box_offsets[k+1]) (none when
 * its text was filtered out or it has no characters); wh_ratio = crop w / h (ocr.rs:739), max_wh_ratio = res->max_wh_ratio. */
typedef struct {
    uint32_t n_regions;
    uint64_t* box_offsets;   /* n_regions + 1 */
    float* boxes;            /* box_offsets[n_regions] * 8 */
} oar_word_boxes;
oar_status oar_ocr_word_boxes(const oar_ocr_result* res, const oar_text_result* txt, oar_word_boxes* out);
void oar_word_boxes_free(oar_word_boxes* w);

/* ------------------------------------------------------------------------------------------------ multi-process hosts
 * One process per GPU, image-parallel: pages are independent units, the crop pool is per shard (ocr.rs:594-634 pools within one
 * predict call), and there is no collective on the data path -- only the final results travel.  The library ships no transport:
 * a rank packs the results of its block of pages into ONE contiguous blob, the host moves blobs with what it has (RCCL / MPI /
 * sockets: bench.py uses torch.distributed's gather), and rank 0 merges them in rank order.  Host-only, no GPU involved.
 *   oar_shard_range : static block partition, rank r owns items [begin, end) = [r n / G, (r + 1) n / G) with the remainder spread
 *                     over the first ranks (SURVEY 8e "Partitioning")
 *   oar_ocr_pack    : (pipeline result, decoded texts) -> blob.  Wire format v1, little-endian, no padding:
 *                     int64 n_images, n_regions, utf8_bytes | u32 region_offsets[n_images + 1] | f32 points[n_regions * 8] |
 *                     f32 scores[n_regions] (the text scores) | u64 text_offsets[n_regions + 1] | utf8.  Quad boxes only.
 *   oar_packed_merge: blobs in rank order -> one flat result (offsets rebased; a block partition makes concatenation = page order) */
typedef struct {
    uint32_t n_images, n_regions;
    uint32_t* region_offsets;  /* n_images + 1 */
    float* points;             /* n_regions * 8 */
    float* scores;             /* n_regions */
    uint64_t* text_offsets;    /* n_regions + 1 */
    char* utf8;                /* text_offsets[n_regions] bytes (+ one NUL) */
} oar_packed_pages;
oar_status oar_shard_range(uint64_t n_items, uint32_t world_size, uint32_t rank, uint64_t* begin, uint64_t* end);
oar_status oar_ocr_pack(const oar_ocr_result* res, const oar_text_result* txt, uint8_t** blob, size_t* len);
void oar_blob_free(uint8_t* blob);
oar_status oar_packed_merge(const uint8_t* const* blobs, const size_t* lens, uint32_t n_blobs, oar_packed_pages* out);
void oar_packed_pages_free(oar_packed_pages* p);

/* ------------------------------------------------------------------------------------------------ Seam B: config-5 stages
 * PP-LCNet classifier adapters: DocumentOrientationAdapter / TextLineOrientationAdapter ->
 * PPLCNetModel::forward_refs (oar-ocr-core/src/models/classification/pp_lcnet.rs:139-330): Triangle resize
 * (short edge -> resize_short + centre crop, or direct resize when resize_short == 0), ImageNet RGB normalisation,
 * graph, Topk (utils/topk.rs:181-199: stable descending sort, first index wins ties).
 * Defaults: doc orientation 224x224 / resize_short 256 / 4 classes (domain/tasks/document_orientation.rs:46-53);
 * text-line orientation 80x160 (h x w) / direct resize / 2 classes (text_line_orientation.rs:25-32).            */
typedef struct oar_cls oar_cls;
typedef struct {
    int32_t device_id;
    uint32_t input_h, input_w;   /* 0,0 => 224,224                                      */
    uint32_t resize_short;       /* 0 => direct resize to (input_w, input_h)            */
    uint32_t topk;               /* 0 => 1                                              */
    uint32_t batch;              /* 0 => 64 images per inference                        */
} oar_cls_cfg;
typedef struct {
    uint32_t n_images, topk, n_classes;
    int32_t* class_ids;          /* n_images * topk (PPLCNetModelOutput::class_ids)     */
    float* scores;               /* n_images * topk                                     */
} oar_cls_result;
oar_status oar_cls_create(const uint8_t* onnx, size_t onnx_len, const oar_cls_cfg* cfg, oar_cls** out);
void oar_cls_destroy(oar_cls* c);
oar_status oar_cls_run(oar_cls* c, const uint8_t* const* rgb, const uint32_t* widths, const uint32_t* heights, uint32_t n_images,
                       oar_cls_result* out);
void oar_cls_result_free(oar_cls_result* r);
/* test hook: the preprocessed batch tensor [n,3,input_h,input_w] (NCHW) of PPLCNetModel::preprocess_refs */
oar_status oar_cls_preprocess(oar_cls* c, const uint8_t* const* rgb, const uint32_t* widths, const uint32_t* heights, uint32_t n_images,
                              float* out_nchw);

/* UVDoc rectifier adapter (row a23): UVDocModel::{preprocess_refs, postprocess}
 * (oar-ocr-core/src/models/rectification/uvdoc.rs:82-109,166-207): Triangle resize to target, BGR v/255, graph
 * ("image" -> [n,3,h,w] BGR in [0,1]), (v*255).clamp(0,255) as u8 -> RGB (processors/simd.rs:327-348), Triangle resize
 * back to the input size.  out_rgb: caller-allocated w*h*3 bytes.                                               */
typedef struct oar_rect oar_rect;
/* target_h / target_w: 0,0 => 512,512 (UVDocPreprocessConfig::default, uvdoc.rs:21-27); OAR_RECT_NATIVE_SIZE => pages go through the
 * graph at their own size, no resize either way -- what DocumentRectificationConfig::default()'s [3, 0, 0] means once it reaches
 * the model through with_config (document_rectification_adapter.rs, uvdoc.rs:84-88). */
#define OAR_RECT_NATIVE_SIZE 0xFFFFFFFFu
typedef struct { int32_t device_id; uint32_t target_h, target_w; } oar_rect_cfg;
oar_status oar_rect_create(const uint8_t* onnx, size_t onnx_len, const oar_rect_cfg* cfg, oar_rect** out);
void oar_rect_destroy(oar_rect* r);
oar_status oar_rect_run(oar_rect* r, const uint8_t* rgb, uint32_t width, uint32_t height, uint8_t* out_rgb);

/* OAROCRBuilder::with_document_image_orientation_classification / with_document_image_rectification /
 * with_text_line_orientation_classification (src/oarocr/ocr.rs): attaches the optional stages of OAROCR::predict --
 * DocumentPreprocessor::preprocess per page (src/oarocr/preprocess.rs:59-141), classify_line_orientations per crop
 * (ocr.rs:757-790), rotate_text_regions_back (ocr.rs:898-925).  Any handle may be NULL; handles are borrowed and must
 * outlive the pipeline's use of them.                                                                           */
oar_status oar_ocr_attach(oar_ocr* o, oar_cls* doc_orientation, oar_rect* rectifier, oar_cls* line_orientation);

/* config-5 stand-alone kernels / host hooks (parity tests) */
/* image::imageops::rotate90/180/270: clockwise by quarter*90 degrees; out is (h x w) for quarter 1, 3 */
oar_status oar_k_rotate_rgb(const uint8_t* rgb, uint32_t w, uint32_t h, int32_t quarter, uint8_t* out);
/* processors/simd.rs:327-348; planes = 3*plane f32 (B, G, R), out = plane*3 u8 RGB */
oar_status oar_k_bgr_planes_to_rgb(const float* planes, uint64_t plane, float scale, uint8_t* out);
/* BoundingBox::rotate_back_to_original (processors/geometry.rs:848-889), in place on n (x, y) pairs */
oar_status oar_host_rotate_back_points(float* pts, uint32_t n_points, float angle, uint32_t rotated_w, uint32_t rotated_h);

/* ------------------------------------------------------------------------------------------------ Seam B: layout detection
 * LayoutDetectionAdapter's model half (oar-ocr-core/src/domain/adapters/layout_detection_adapter.rs:1121-1197): PicoDet / RT-DETR /
 * PP-DocLayout graphs through ScaleAwareDetectorModel::forward (models/detection/scale_aware_detector.rs:169-440: resize_exact to
 * the model's image_shape with the model's filter, NormalizeImage, graph inputs "image" + "scale_factor" [+ "im_shape"]) and
 * LayoutPostProcess::apply (processors/layout_postprocess.rs:60-634: row parsing in the three column orders, score filter,
 * normalised / absolute coordinate conversion, class-aware greedy NMS, max_detections, PP-DocLayoutV2 reading order).  Resize,
 * normalisation, the network and the post-processing are HIP kernels; per image the result is LayoutPostprocessOutput's
 * (boxes, class ids, scores).  What the adapter does ABOVE that with configuration only -- class labels, per-class thresholds,
 * max_elements -- stays with the caller (rust/oar-mi355x-adapters/src/layout_detection.rs, api.LayoutDetectionPredictor).       */
typedef struct oar_layout oar_layout;
typedef struct {
    int32_t device_id;
    uint32_t input_h, input_w;   /* image_shape: 0,0 => 800 x 608 (PicoDet); PP-DocLayout 800 x 800                       */
    int32_t resize_filter;       /* 0 Triangle, 1 CatmullRom (PP-DocLayout), 2 Lanczos3 (PicoDet): scale_aware_detector.rs:49-75 */
    int32_t color_bgr;           /* 1: BGR tensor (PicoDet), 0: RGB (PP-DocLayout)                                          */
    float scale;                 /* 0 => 1/255                                                                              */
    float mean[3], std[3];       /* RGB statistics; std all 0 => ImageNet (0.485.. / 0.229..)                               */
    uint32_t num_classes;
    int32_t model_type;          /* 0 "picodet" (and any other name: process_standard), 1 "rtdetr", 2 "pp-doclayout"        */
    float score_threshold;       /* LayoutPostProcess::score_threshold                                                      */
    float nms_threshold;
    uint32_t max_detections;     /* 0 => 100                                                                                */
} oar_layout_cfg;
typedef struct {
    uint32_t n_images, n_boxes;
    uint32_t* box_offsets;       /* n_images + 1                                                                            */
    float* boxes;                /* n_boxes * 4: x1 y1 x2 y2 in original-image pixels (BoundingBox = (x1,y1)(x2,y1)(x2,y2)(x1,y2)) */
    int32_t* classes;            /* n_boxes                                                                                 */
    float* scores;               /* n_boxes                                                                                 */
    uint32_t feature_dim;        /* columns of the graph's prediction rows: 7 / 8 => is_reading_order_sorted (adapter :1186-1192) */
} oar_layout_result;
oar_status oar_layout_create(const uint8_t* onnx, size_t onnx_len, const oar_layout_cfg* cfg, oar_layout** out);
void oar_layout_destroy(oar_layout* l);
oar_status oar_layout_run(oar_layout* l, const uint8_t* const* rgb, const uint32_t* widths, const uint32_t* heights, uint32_t n_images, oar_layout_result* out);
void oar_layout_result_free(oar_layout_result* r);
/* parity hooks: the preprocessed tensor of one image ([3, input_h, input_w] f32); the filtered resize alone; LayoutPostProcess alone on
 * a caller-supplied prediction tensor [n_images, rows, feat] (src_wh: n_images x (width, height)) */
oar_status oar_layout_preprocess(oar_layout* l, const uint8_t* rgb, uint32_t width, uint32_t height, float* out_chw);
oar_status oar_k_resize_filter(const uint8_t* rgb, uint32_t w, uint32_t h, uint32_t nw, uint32_t nh, int32_t filter, uint8_t* out);
oar_status oar_k_layout_postprocess(const float* pred, uint32_t n_images, uint32_t rows, uint32_t feat, const float* src_wh, uint32_t num_classes, int32_t model_type,
                                    float score_threshold, float nms_threshold, uint32_t max_detections, oar_layout_result* out);
/* PP-DocLayout: the adapter's own PaddleX-style post-processing (LayoutDetectionAdapter::postprocess_pp_doclayout, layout_detection_adapter.rs:631-846)
 * replaces LayoutPostProcess for model_type "pp-doclayout": per-class thresholds, paddlex_layout_nms (:884-935), filter_large_image_boxes (:955-995),
 * apply_paddlex_merge_modes (:997-1100) and the reading-order sort, as one HIP kernel on the raw prediction rows (layout.hip ppdoc_post_kernel).
 * Above the ABI stay: class labels, layout_unclip_ratio, max_elements.                                                                              */
typedef struct {
    float score_threshold;              /* LayoutDetectionConfig::score_threshold (clamped at 0 as the adapter does)                                 */
    const float* class_thresholds;      /* [num_classes] by class id, NaN = not configured; NULL = none configured                                   */
    int32_t layout_nms;                 /* LayoutDetectionConfig::layout_nms                                                                         */
    int32_t image_class_id;             /* id of the label "image" or -1                                                                             */
    int32_t formula_class_id;           /* id of the label "formula" or -1                                                                           */
    const int32_t* class_merge_modes;   /* [num_classes] by class id: -1 not configured, 0 Large, 1 Union, 2 Small (MergeBboxMode); NULL = none      */
} oar_ppdoc_cfg;
oar_status oar_layout_run_ppdoc(oar_layout* l, const uint8_t* const* rgb, const uint32_t* widths, const uint32_t* heights, uint32_t n_images, const oar_ppdoc_cfg* cfg,
                                oar_layout_result* out);
oar_status oar_k_ppdoc_postprocess(const float* pred, uint32_t n_images, uint32_t rows, uint32_t feat, const float* src_wh, uint32_t num_classes, const oar_ppdoc_cfg* cfg,
                                   oar_layout_result* out);
/* class_merge_modes of the PicoDet / RT-DETR adapters: apply_nms_with_merge (processors/layout_postprocess.rs:692-841) on one image's kept boxes
 * (host code: a greedy, order-dependent merge of a few dozen boxes).  mode_of_class: [num_classes] 0 Large (the default of an unlisted class), 1 Union,
 * 2 Small.  out_*: room for n entries; returns the number written.                                                                                  */
int32_t oar_host_nms_with_merge(const float* boxes, const int32_t* classes, const float* scores, uint32_t n, const int32_t* mode_of_class, uint32_t num_classes,
                                float nms_threshold, uint32_t max_detections, float* out_boxes, int32_t* out_classes, float* out_scores);

/* ------------------------------------------------------------------------------------------------ device helpers */
oar_status oar_dev_alloc(int32_t device_id, size_t bytes, void** out);
oar_status oar_dev_upload(void* dst, const void* src, size_t bytes);
oar_status oar_dev_download(void* dst, const void* src, size_t bytes);
void oar_dev_free(void* p);
oar_status oar_dev_synchronize(int32_t device_id);

/* ------------------------------------------------------------------------------------------------ stand-alone kernels
 * Exposed so each HIP kernel can be parity-tested against the oracle on identical input bits. Host in/out.
 * a4  processors/simd.rs:28-45,87-123   */
oar_status oar_k_normalize(const uint8_t* rgb, uint32_t w, uint32_t h, const int32_t src_channels[3],
                           const float alpha[3], const float beta[3], int32_t hwc_layout, float* out);
/* a16 processors/simd.rs:248-308 + image Triangle resize (models/recognition/crnn.rs:98-121) */
oar_status oar_k_rec_preprocess(const uint8_t* const* rgb, const uint32_t* widths, const uint32_t* heights,
                                uint32_t n, uint32_t img_h, uint32_t img_w, uint32_t max_img_w,
                                float* out_nchw, uint32_t* tensor_width);
/* the same with flips[i] != 0 => crop i is read as its imageops::rotate180 (classify_line_orientations' class 1,
 * src/oarocr/ocr.rs:785-788) without a rotated copy being made; flips == NULL: none */
oar_status oar_k_rec_preprocess_flip(const uint8_t* const* rgb, const uint32_t* widths, const uint32_t* heights, const uint8_t* flips,
                                     uint32_t n, uint32_t img_h, uint32_t img_w, uint32_t max_img_w,
                                     float* out_nchw, uint32_t* tensor_width);
/* a3  image Triangle resize (processors/resize_detection.rs:314) */
oar_status oar_k_resize_triangle(const uint8_t* rgb, uint32_t w, uint32_t h, uint32_t nw, uint32_t nh, uint8_t* out);
/* a7  processors/db_postprocess.rs:185-221 */
oar_status oar_k_threshold(const float* pred, size_t n, float thresh, uint8_t* mask);
/* processors/db_mask.rs:11 (imageproc dilate, LInf, k = 1) on one height x width mask */
oar_status oar_k_dilate(const uint8_t* mask, uint32_t height, uint32_t width, uint8_t* out);
/* processors/db_score.rs:139-181 (and db_bitmap.rs:49): scanline mean over polygons of any size; polygon i has
 * counts[i] (x, y) points, stored back to back in pts_xy */
oar_status oar_k_poly_scores(const float* pred, uint32_t height, uint32_t width, const float* pts_xy, const uint32_t* counts,
                             uint32_t n_polys, float* scores);
/* a8 processors/db_bitmap.rs:100 (imageproc find_contours) through the GPU border follower the detector uses (contours.hip): same
 * outputs as oar_host_contours (offsets: max_contours + 1 entries; types 0 outer / 1 hole), the count through n_contours */
oar_status oar_k_contours(const uint8_t* mask, uint32_t width, uint32_t height, uint32_t max_contours, int32_t* n_contours,
                          int64_t* offsets, int32_t* pts_xy, int32_t* types, int64_t cap_points);
/* a18 processors/decode.rs:452-501 + simd.rs:72-81 */
oar_status oar_k_ctc_argmax(const float* probs, size_t rows, size_t vocab, int64_t* idx, float* prob);
/* a11 db_bitmap.rs:279-368 on the GPU (the kernel the detector runs next to the box scores): boxes = n_boxes * 8 floats;
 * counts[i] = vertices of box i (0 = dropped, -1 = left to the host routine); pts_xy: cap_points (x, y) pairs per box. */
oar_status oar_k_unclip(const float* boxes, uint32_t n_boxes, float ratio, int32_t* counts, float* pts_xy, uint32_t cap_points);
/* a10 processors/db_score.rs:34-134: boxes = n_boxes * 8 floats */
oar_status oar_k_box_scores(const float* pred, uint32_t height, uint32_t width, const float* boxes,
                            uint32_t n_boxes, float* scores);
/* a14 utils/transform.rs:76-191: returns crop dims through out_w/out_h (0,0 when the reference would error);
 * out must hold at least cap bytes. */
oar_status oar_k_rotate_crop(const uint8_t* rgb, uint32_t w, uint32_t h, const float box[8], uint8_t* out,
                             size_t cap, uint32_t* out_w, uint32_t* out_h);

/* ------------------------------------------------------------------------------------------------ image decode
 * load_image_from_memory (oar-ocr-core/src/utils/image.rs:65-68: image::load_from_memory + DynamicImage::to_rgb8) for the
 * formats this library decodes itself.  PNG: every colour type / bit depth / interlace mode, to the bytes the image crate
 * yields (palette and low-bit grey expanded, alpha dropped, 16-bit -> 8-bit as (v + 128) / 257).  JPEG: baseline / extended
 * sequential / progressive Huffman, 8-bit, grey or 3 components at any integral sampling ratio, restart intervals (jpeg_decode.cc):
 * JPEG decoding is not bit-specified, so the bytes are those of the de-facto standard -- libjpeg(-turbo)'s default path (islow IDCT,
 * fancy upsampling, its colour tables), exactly PIL's output -- and UNPINNED against zune-jpeg, the crate the reference links
 * (expected agreement +-1..2 levels at chroma edges).  *rgb receives width * height * 3
 * bytes owned by the library (release with oar_image_free).  Errors: OAR_INVALID_INPUT = OCRError::ImageLoad (corrupt /
 * truncated data, CRC mismatch); OAR_UNSUPPORTED_OP = a format of the image crate that is not decoded here (BMP, GIF, WebP, ..., arithmetic-coded / 12-bit / CMYK JPEG; the
 * message names it) -- keep the reference's loader for those.  Thread-safe, no lock: decode a batch from as many threads as the
 * reference's rayon pool would use (utils/image.rs:299-345). */
oar_status oar_image_decode(const uint8_t* bytes, size_t len, uint8_t** rgb, uint32_t* width, uint32_t* height);
void oar_image_free(uint8_t* rgb);
/* The same image decoded INTO HBM: *dev_rgb = width * height * 3 bytes on device_id (release with oar_dev_free), ready for
 * oar_ocr_predict_device / oar_det_run_device.  For JPEG only the Huffman stream is decoded on the host; dequantisation, IDCT,
 * chroma upsampling and colour conversion run as HIP kernels (jpeg.hip), bit-identical to oar_image_decode's host arithmetic.
 * Other decoded formats (PNG) are decoded on the host and uploaded. */
oar_status oar_image_decode_device(const uint8_t* bytes, size_t len, int32_t device_id, void** dev_rgb, uint32_t* width, uint32_t* height);

/* ------------------------------------------------------------------------------------------------ host-side geometry hooks
 * The serial per-contour stages of DB post-processing and crop planning run on the host.
 * These hooks expose them WITHOUT touching a GPU so the CPU test-suite can check them against the oracle.
 * a8+a9 db_bitmap.rs:100-113,153-277: mask -> mini-box candidates (8 floats each, discovery order). max_bands > 1
 * traces row bands cut at blank rows in the same order (identical result). Returns the count (<= cap written). */
int32_t oar_host_candidates(const uint8_t* mask, uint32_t width, uint32_t height, uint32_t max_candidates, int32_t max_bands,
                            float* boxes8, int32_t cap);
/* a8 alone: the contours find_contours yields (imageproc semantics at db_bitmap.rs:100: outer AND hole borders, raster
 * discovery order, border pixels in tracing order).  offsets: max_contours + 1 entries; pts_xy: (x, y) pairs, at most
 * cap_points of them are written (offsets still count every point); types: 0 outer / 1 hole.  Returns the count. */
int32_t oar_host_contours(const uint8_t* mask, uint32_t width, uint32_t height, uint32_t max_contours, int32_t max_bands,
                          int64_t* offsets, int32_t* pts_xy, int32_t* types, int64_t cap_points);
/* The same through the detector's read-back format: the mask is packed to a bit plane (pixel x = bit x & 7 of byte x >> 3) and the
 * host border follower reads bits (what crosses PCIe after a detector pass; 8x smaller than the byte mask). */
int32_t oar_host_contours_bits(const uint8_t* mask, uint32_t width, uint32_t height, uint32_t max_contours, int32_t max_bands,
                               int64_t* offsets, int32_t* pts_xy, int32_t* types, int64_t cap_points);
/* a11 db_bitmap.rs:279-368: unclip a 4-point box; returns the number of points (0 = dropped), x,y pairs in out. */
int32_t oar_host_unclip(const float box8[8], float ratio, float* out_xy, int32_t cap_points);
/* f2 (polygon / seal branch) processors/geometry.rs:453-561: Douglas-Peucker over the open chain xy[0 .. n_points); returns the
 * number of kept points (<= cap_points written). */
int32_t oar_host_approx_poly_dp(const float* xy, int32_t n_points, float epsilon, float* out_xy, int32_t cap_points);
/* f2 processors/geometry.rs:161-171: closed-ring perimeter, f32 accumulation. */
float oar_host_perimeter(const float* xy, int32_t n_points);
/* f2 db_bitmap.rs:279-368 for ANY polygon (Clipper2 round-join offset + the outline its closing union keeps); returns the number
 * of points, 0 = the reference drops the box (degenerate input or an offset that is not exactly one path), -1 = error. */
int32_t oar_host_unclip_poly(const float* xy, int32_t n_points, float ratio, float* out_xy, int32_t cap_points);
/* f2 the two halves of the above on 1/100 px grid coordinates, for the tests: the raw Clipper2 offset ring of a closed ring
 * (radius < 0 for a clockwise ring), and the outline of a raw ring (negative != 0: the ring runs clockwise).  Return the number of
 * vertices, 0 (outline: not exactly one loop), -1 when cap_points is too small. */
int32_t oar_host_offset_ring(const int64_t* xy, int32_t n_points, double radius, int64_t* out_xy, int32_t cap_points);
int32_t oar_host_ring_outline(const int64_t* xy, int32_t n_points, int32_t negative, int64_t* out_xy, int32_t cap_points);
/* f2 processors/sorting.rs:100-118: permutation that sorts n polygons (CSR: polygon i = points [offsets[i], offsets[i + 1]))
 * by their smallest y, stably. */
void oar_host_sort_poly_boxes(const float* pts_xy, const uint32_t* offsets, int32_t n, int32_t* order);
/* a9 mini box of an arbitrary point set (db_bitmap.rs:164-205): returns 1 and fills box8/min_side, or 0. */
int32_t oar_host_mini_box(const float* xy, int32_t n_points, float box8[8], float* min_side);
/* convex_hull (processors/geometry.rs:226-271: Graham scan from the lowest-then-leftmost point, atan2 / squared-distance keys, pop on cross <= 0) of
 * n_points (x, y) pairs -> hull vertices in scan order; returns their count, -1 on bad arguments or cap_points too small.  A test entry: the product's
 * row-extreme pre-filter for integer border pixels (db_host.cc) is checked through it against an exact integer-arithmetic scan. */
int32_t oar_host_convex_hull(const float* xy, int32_t n_points, float* out_xy, int32_t cap_points);
/* a13 processors/sorting.rs:35-84: permutation that sorts n quad boxes into reading order. */
void oar_host_sort_quad_boxes(const float* boxes8, int32_t n, int32_t* order);
/* Self-test of the geometry thread pool: `jobs` back-to-back parallel loops of varying length on `threads` workers;
   returns 0 when every index of every loop ran exactly once in its own loop, else 1 + the index of the first bad loop. */
int32_t oar_host_pool_selftest(int32_t threads, int32_t jobs);
/* a14 utils/transform.rs:76-191 planning half: plan[8] = {mode, left, top, cw, ch, out_w, out_h, rot}; inv[9]. */
void oar_host_plan_crop(uint32_t img_w, uint32_t img_h, const float box8[8], int32_t plan[8], float inv[9]);

/* Test hook: makes the next `count` occurrences of a failure site throw OAR_DEVICE.  Sites: "batched_detection" (a
 * detector run over more than one page) -- exercises OAROCR::predict's "batched text detection failed; falling back to
 * per-image detection" path (src/oarocr/ocr.rs:576-588), which oar_ocr_predict reproduces inside the call. */
oar_status oar_debug_inject_failure(const char* site, int32_t count);

/* ------------------------------------------------------------------------------------------------ profiling
 * Per-kernel-class accumulators filled from hipEvents recorded on the engine's own stream while cfg.profile
 * is set.  class_name e.g. "conv_igemm", "dwconv", "softmax".  alg_bytes / alg_flops are the algorithmic
 * totals of the timed launches. */
typedef struct {
    char name[48];
    uint64_t launches;
    double total_ms;
    double alg_bytes;
    double alg_flops;
} oar_prof_entry;
void oar_prof_reset(void);
void oar_prof_enable(int32_t on);
/* Restrict instrumentation to one kernel class (NULL or "" = all): keeps event overhead out of a timed region. */
void oar_prof_filter(const char* class_name);
/* Time only every stride-th launch of the instrumented class(es): launches are counted from this call, launch i is
   timed iff i % stride == phase (stride <= 1: every launch; phase < 0: none).  A caller that rotates phase over the
   steps of a region times every launch position equally often at 1/stride of the event overhead (an event-bracketed
   kernel costs ~11 us of idle queue around it). */
void oar_prof_sampling(int32_t stride, int32_t phase);
/* Fills up to cap entries, sorted by total_ms descending; returns the number of classes. */
int32_t oar_prof_snapshot(oar_prof_entry* entries, int32_t cap);

#ifdef __cplusplus
}
#endif
#endif /* OAR_MI355X_H */
