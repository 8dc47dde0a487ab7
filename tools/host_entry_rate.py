"""PCIe-inclusive rate of the host-buffer entry point (oar_ocr_predict): same workload as bench.py, pages in host memory."""
import sys, time
sys.path.insert(0, ".")
from oar_ocr_amd import api
from oar_ocr_amd.synth import models, pages
det, _ = models.build_det("tiny", seed=0); rec, _ = models.build_rec("tiny", vocab=6906, seed=1)
chars = api.read_dict(models.synth_dict(6904))
P = [pages.make_page(i, (960, 960), 40) for i in range(32)]
ocr = api.OAROCRBuilder(det, rec, chars).text_detection_config(api.TextDetectionConfig(0.3, 0.6, 1.5)).image_batch_size(32).region_batch_size(256).build()
for _ in range(3): ocr.predict(P)
t = time.perf_counter()
for _ in range(10): r = ocr.predict(P)
dt = (time.perf_counter() - t) / 10
print(f"host-buffer entry: {dt*1e3:.2f} ms per 32 pages = {32/dt:.1f} images/s (includes H2D of 88 MB and the Python result objects)")
ocr.close()
