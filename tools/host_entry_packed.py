"""Where the host-entry step of bench.py spends its time outside oar_ocr_predict: predict (C call) / oar_ocr_decode (CTC collapse, text) / result copies."""
import sys, time
sys.path.insert(0, ".")
import ctypes as C
import numpy as np
from oar_ocr_amd import api
from oar_ocr_amd.synth import models, pages
det, _ = models.build_det("tiny", seed=0); rec, _ = models.build_rec("tiny", vocab=6906, seed=1)
chars = api.read_dict(models.synth_dict(6904))
P = [pages.make_page(i, (960, 960), 40) for i in range(32)]
ocr = api.OAROCRBuilder(det, rec, chars).text_detection_config(api.TextDetectionConfig(0.3, 0.6, 1.5)).image_batch_size(32).region_batch_size(int(sys.argv[1]) if len(sys.argv) > 1 else 256).build()
_, ptrs, ws, hs = api._img_arrays(P)
n = len(P)
for _ in range(3): ocr.predict_packed(ptrs, ws, hs, n)
ta = tb = tc = 0
R = 20
for _ in range(R):
    t0 = time.perf_counter()
    res = api.OcrResult()
    api._check(api.lib().oar_ocr_predict(ocr._h, ptrs, ws, hs, n, C.byref(res)))
    t1 = time.perf_counter()
    d = ocr.ctc.decode_ocr(res, ocr.score_threshold, want_positions=False, want_blob=False)
    t2 = time.perf_counter()
    nr = int(res.n_regions)
    offs = np.ctypeslib.as_array(res.region_offsets, shape=(n + 1,)).copy()
    pts = np.ctypeslib.as_array(res.points, shape=(max(nr, 1) * 8,)).copy()[:nr * 8].reshape(nr, 4, 2)
    api.lib().oar_ocr_result_free(C.byref(res))
    t3 = time.perf_counter()
    ta += t1 - t0; tb += t2 - t1; tc += t3 - t2
print(f"regions {nr}: oar_ocr_predict {ta / R * 1e3:.3f} ms  decode {tb / R * 1e3:.3f} ms  copies + free {tc / R * 1e3:.3f} ms  -> {n / ((ta + tb + tc) / R):.0f} images/s")
ocr.close()
