"""Diagnosis of a parity_fuzz seal mismatch (gpurun_out/fuzz_case_<n>.npz): is it the network's float noise at the bitmap threshold, or the
polygon post-process?  For every page: product detector network (engine through the C ABI) vs the oracle's torch-CPU network on the SAME
preprocessed tensor -> max |diff| of the probability maps, pixels whose `prob > thresh` bit differs; then the oracle's polygon post-process on
the PRODUCT's map against the product's own boxes for that page.  usage: python tools/seal_case_diag.py <npz> thresh box_thresh unclip"""
import sys
sys.path.insert(0, ".")
import numpy as np
from oar_ocr_amd import api
from oar_ocr_amd.synth import models
from oracle import pipeline_ref, poly_ref, cpu_ref as R

z = np.load(sys.argv[1]); thr, bthr, unclip = float(sys.argv[2]), float(sys.argv[3]), float(sys.argv[4])
imgs = [z[k] for k in sorted(z.files)]
det, _ = models.build_det("tiny", seed=0)
od = pipeline_ref.OracleDetector(det, text_type="seal")
eng = api.OrtInfer(det)
pred = api.TextDetectionPredictor(det, api.TextDetectionConfig(thr, bthr, unclip), text_type="seal")
# pages of one shape go through the network as ONE batch on both sides (db.rs:297-309), as they do in OAROCR::predict -- a 1e-6 difference may
# exist at batch 3 and not at batch 1 (other tile boundaries in the kernels)
groups = {}
for i, im in enumerate(imgs):
    groups.setdefault(im.shape, []).append(i)
for shape, idx in groups.items():
    batch = [imgs[i] for i in idx]
    maps_o = od.prob_maps(batch)
    x = np.stack([R.det_preprocess(im, *od.cfg)[0] for im in batch])
    maps_p = eng.infer(x)[0][1][:, 0]
    got_all = pred.predict(batch)
    for k, i in enumerate(idx):
        prob_o, (sh, sw) = maps_o[k]
        prob_p = maps_p[k]
        d = np.abs(prob_p - prob_o)
        flips = np.argwhere((prob_p > np.float32(thr)) != (prob_o > np.float32(thr)))
        print(f"page {i} {imgs[i].shape[:2]} (batch of {len(idx)}) -> map {prob_o.shape}: max |prob diff| {d.max():.3e}; threshold flips {len(flips)} {[(int(y), int(x_), float(prob_o[y, x_]), float(prob_p[y, x_])) for y, x_ in flips[:6]]}")
        bo, so = poly_ref.db_postprocess_poly(prob_o, sh, sw, thr, bthr, unclip, 1000)
        bp, sp = poly_ref.db_postprocess_poly(prob_p, sh, sw, thr, bthr, unclip, 1000)
        gb = [np.asarray(d_.bbox, np.float32).reshape(-1, 2) for d_ in got_all[k]]
        same_o = len(gb) == len(bo) and all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(gb, bo))
        same_p = len(gb) == len(bp) and all(a.shape == b.shape and np.array_equal(a, b) for a, b in zip(gb, bp))
        print(f"   product boxes == oracle post-process of the ORACLE map: {same_o}; == oracle post-process of the PRODUCT map: {same_p}")
