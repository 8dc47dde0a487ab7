import sys, time
sys.path.insert(0, ".")
import ctypes as C
from oar_ocr_amd import api
from oar_ocr_amd.synth import models, pages
det, _ = models.build_det("tiny", seed=0); rec, _ = models.build_rec("tiny", vocab=6906, seed=1)
chars = api.read_dict(models.synth_dict(6904))
P = [pages.make_page(i, (960, 960), 40) for i in range(32)]
ocr = api.OAROCRBuilder(det, rec, chars).text_detection_config(api.TextDetectionConfig(0.3, 0.6, 1.5)).image_batch_size(32).region_batch_size(256).build()
for _ in range(3): ocr.predict(P)
ta=tb=tc=0
for _ in range(10):
    t0=time.perf_counter()
    imgs, ptrs, ws, hs = api._img_arrays(P)
    t1=time.perf_counter()
    res = api.OcrResult()
    api._check(api.lib().oar_ocr_predict(ocr._h, ptrs, ws, hs, len(imgs), C.byref(res)))
    t2=time.perf_counter()
    out = ocr._assemble(res)
    api.lib().oar_ocr_result_free(C.byref(res))
    t3=time.perf_counter()
    ta+=t1-t0; tb+=t2-t1; tc+=t3-t2
print(f"img_arrays {ta*100:.2f} ms  C call {tb*100:.2f} ms  assemble {tc*100:.2f} ms")
ocr.close()
