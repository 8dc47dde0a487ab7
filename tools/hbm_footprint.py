import sys; sys.path.insert(0, ".")
import torch
from oar_ocr_amd import api
from oar_ocr_amd.synth import models, pages
free0, total = torch.cuda.mem_get_info()
det, _ = models.build_det("tiny", seed=0); rec, _ = models.build_rec("tiny", vocab=6906, seed=1)
chars = api.read_dict(models.synth_dict(6904))
P = [pages.make_page(i, (960, 960), 40) for i in range(32)]
bufs = [api.DeviceBuffer(p) for p in P]
ocr = api.OAROCRBuilder(det, rec, chars).text_detection_config(api.TextDetectionConfig(0.3, 0.6, 1.5)).image_batch_size(32).region_batch_size(256).build()
ptrs = [int(b.ptr.value) for b in bufs]
for _ in range(3): ocr.predict_device(ptrs, [960]*32, [960]*32, raw=True)
free1, _ = torch.cuda.mem_get_info()
print(f"HBM total {total/2**30:.1f} GiB; used by the bench workload (pages + weights + arenas + pools): {(free0-free1)/2**30:.2f} GiB")
