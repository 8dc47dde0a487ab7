"""ORT-CPU stand-in (SURVEY 8d "CPU baseline beside it", plan 1) -- TEST / MEASUREMENT INFRASTRUCTURE ONLY, nothing under oar_ocr_amd/ imports this.

Only reachable when an operator has supplied registry-verified model files (oar_ocr_amd/weights.py) AND `import onnxruntime` works on the box: the
two networks then run through ONNX Runtime's CPU execution provider -- what the reference's `ort` crate drives (core/inference/session.rs:30-44,
ort_infer_execution.rs:178,281) -- with this repo's CPU restatement of the pre / post stages around them.  Two results:

  * throughput of that pipeline on a bounded page sample (the closest thing to the reference's CPU path that can exist here), and
  * the network-level parity report SURVEY 8c calls "unpinned vs ORT": for the same preprocessed tensors, the GPU engine's outputs (Seam A,
    api.OrtInfer) against ORT's -- max |dprob| of the detector's probability maps with the number of pixels closer to the 0.3 threshold than that
    difference (the only pixels whose mask bit could differ), and max |dprob| / arg-max mismatches of the recognizer's soft-max rows.

Neither onnxruntime nor real weights exist in the build container, so this module is exercised only by its import and by the stubbed-session unit test."""
from __future__ import annotations

import time

import numpy as np

from . import cpu_ref as R
from . import onnx_ref, pipeline_ref


class OrtNet:
    """One InferenceSession with the interface pipeline_ref's adapters use (`onnx_ref.run(model, feeds)` -> list of outputs)."""

    def __init__(self, model_bytes, threads, session_factory=None):
        if session_factory is None:
            import onnxruntime as ort
            so = ort.SessionOptions()
            so.intra_op_num_threads = int(threads)
            self.sess = ort.InferenceSession(bytes(model_bytes), sess_options=so, providers=["CPUExecutionProvider"])
        else:
            self.sess = session_factory(model_bytes)
        self.input = self.sess.get_inputs()[0].name

    def run(self, feeds):
        return self.sess.run(None, {k: np.ascontiguousarray(v, np.float32) for k, v in feeds.items()})


def make_oracle_ocr(det_bytes, rec_bytes, chars, threads, session_factory=None, **det_kw):
    """OracleOCR whose two network evaluations go through ORT; everything else is the parity oracle's code."""
    oc = pipeline_ref.OracleOCR(det_bytes, rec_bytes, chars, 0.3, 0.6, 1.5, image_batch_size=1, region_batch_size=16, threads=threads, **det_kw)
    nets = {id(oc.det.model): OrtNet(det_bytes, threads, session_factory), id(oc.rec.model): OrtNet(rec_bytes, threads, session_factory)}
    torch_run = onnx_ref.run

    def run(model, feeds, want=None):
        net = nets.get(id(model))
        return net.run(feeds) if net is not None and want is None else torch_run(model, feeds, want)
    return oc, run


def network_parity(det_bytes, rec_bytes, pages, api, threads, limit_side_len=None, session_factory=None):
    """GPU engine (Seam A) vs ORT on identical input tensors.  Returns the report dict described in the module docstring."""
    det_net, rec_net = OrtNet(det_bytes, threads, session_factory), OrtNet(rec_bytes, threads, session_factory)
    det_eng, rec_eng = api.OrtInfer(det_bytes), api.OrtInfer(rec_bytes)
    rep = {"det_max_abs_dprob": 0.0, "det_threshold_marginal_pixels": 0, "det_pixels": 0, "rec_max_abs_dprob": 0.0, "rec_argmax_mismatches": 0, "rec_rows": 0}
    kw = (limit_side_len or 960, "max", 4000)
    crops = []
    for pg in pages:
        x, _ = R.det_preprocess(pg, *kw)
        ref = np.asarray(det_net.run({det_net.input: x[None]})[0], np.float32)
        got = det_eng.infer(x[None])[0][1]
        d = float(np.abs(got - ref).max())
        rep["det_max_abs_dprob"] = max(rep["det_max_abs_dprob"], d)
        rep["det_threshold_marginal_pixels"] += int((np.abs(ref - np.float32(0.3)) <= d).sum())
        rep["det_pixels"] += int(ref.size)
        boxes, _ = R.db_postprocess(ref[0, 0], pg.shape[0], pg.shape[1], 0.3, 0.6, 1.5, 1000)
        for b in boxes[:16]:
            c = R.rotate_crop(pg, b)
            if c is not None:
                crops.append(c)
    for s in range(0, len(crops), 16):
        xr = R.rec_preprocess(crops[s:s + 16])
        ref = np.asarray(rec_net.run({rec_net.input: xr})[0], np.float32)
        got = rec_eng.infer(xr)[0][1]
        rep["rec_max_abs_dprob"] = max(rep["rec_max_abs_dprob"], float(np.abs(got - ref).max()))
        rep["rec_argmax_mismatches"] += int((got.argmax(-1) != ref.argmax(-1)).sum())
        rep["rec_rows"] += int(ref.shape[0] * ref.shape[1])
    det_eng.close(); rec_eng.close()
    rep["budget"] = "north_star: float scores / logits within 1e-3; a mask bit can only differ at a threshold-marginal pixel"
    return rep


def time_and_compare(det_bytes, rec_bytes, chars, pages, api, limit_side_len=None, threads=8, session_factory=None):
    det_kw = dict(limit_side_len=limit_side_len) if limit_side_len else {}
    oc, run = make_oracle_ocr(det_bytes, rec_bytes, chars, threads, session_factory, **det_kw)
    saved = onnx_ref.run
    onnx_ref.run = run
    try:
        oc.predict(pages[:1])                      # warm (the reference excludes the first call too, docs/FAQ.md:30)
        t0 = time.perf_counter()
        regions = 0
        for pg in pages:
            regions += len(oc.predict([pg])[0])
        dt = time.perf_counter() - t0
    finally:
        onnx_ref.run = saved
    return {"images_per_sec": round(len(pages) / dt, 3), "threads": threads, "regions": regions,
            "sample": f"{len(pages)} page(s), one predict per page, det batch 1 / rec batch 16 (reference CPU defaults); networks on onnxruntime CPUExecutionProvider with "
                      f"{threads} intra-op threads, pre / post = this repo's C restatement",
            "parity": network_parity(det_bytes, rec_bytes, pages[:2], api, threads, limit_side_len, session_factory)}
